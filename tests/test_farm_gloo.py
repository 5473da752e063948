"""CPU tests of the multi-GPU farm path with world_size 2 over gloo: sequence assignment, and the
record gather that replaces the reference's log scraping.  No GPU compute: the per-sequence job is replaced by a stub."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from das3r_amd import farm


def test_assignment_round_robin_and_lpt():
    assert farm.assign(8, 0, 8) == [0] and farm.assign(8, 7, 8) == [7]
    got = [farm.assign(14, r, 8) for r in range(8)]
    assert sorted(sum(got, [])) == list(range(14)) and max(len(g) for g in got) == 2
    costs = [50, 20, 20, 20, 50, 10, 10, 10, 30, 30, 40, 40, 25, 25]
    bins = [farm.assign(14, r, 4, costs) for r in range(4)]
    assert sorted(sum(bins, [])) == list(range(14))
    loads = [sum(costs[i] for i in b) for b in bins]
    assert max(loads) - min(loads) <= 20            # LPT keeps the ranks balanced
    assert bins == [farm.assign(14, r, 4, costs) for r in range(4)]   # deterministic


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_seq, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = farm.assign(n_seq, rank, world)
    recs = []
    for s in mine:
        if s == 3:   # a failed sequence must not take the farm down and must show up as ok = 0
            recs.append(dict(scene_id=s, psnr=float("nan"), l1=float("nan"), iters_per_s=0.0, n_splats=0, ok=0))
        else:
            recs.append(dict(scene_id=s, psnr=20.0 + s, l1=0.01 * s, iters_per_s=100.0 + rank, n_splats=1000 * (s + 1), ok=1))
    table = farm.gather_records(recs, n_seq, torch.device("cpu"))
    torch.save(table, os.path.join(out_dir, f"table_{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_gather_records_world2_gloo(tmp_path):
    n_seq, world = 5, 2
    mp.spawn(_worker, args=(world, _free_port(), n_seq, str(tmp_path)), nprocs=world, join=True)
    t0, t1 = torch.load(tmp_path / "table_0.pt"), torch.load(tmp_path / "table_1.pt")
    assert torch.equal(torch.nan_to_num(t0, nan=-7.0), torch.nan_to_num(t1, nan=-7.0)), "every rank must see the same table"
    assert t0.shape == (n_seq, len(farm.RECORD_FIELDS))
    assert t0[:, 0].tolist() == [0, 1, 2, 3, 4]
    assert t0[[0, 1, 2, 4], 1].tolist() == [20.0, 21.0, 22.0, 24.0] and t0[3, 5] == 0 and t0[[0, 1, 2, 4], 5].tolist() == [1, 1, 1, 1]
    assert t0[2, 3] == 100.0 and t0[1, 3] == 101.0          # sequence 2 ran on rank 0, sequence 1 on rank 1


def test_single_process_gather():
    recs = [dict(scene_id=i, psnr=30.0 - i, l1=0.0, iters_per_s=1.0, n_splats=10, ok=1) for i in range(3)]
    t = farm.gather_records(recs, 3, torch.device("cpu"))
    assert t[:, 1].tolist() == [30.0, 29.0, 28.0]


def test_heldout_split_rule():
    from das3r_amd.train import is_test_index
    assert [i for i in range(40) if is_test_index(i)] == [5, 15, 25, 35]
