"""CPU tests of the multi-GPU farm path with world_size 2 over gloo: sequence assignment, and the
record gather that replaces the reference's log scraping.  No GPU compute: the per-sequence job is replaced by a stub."""
import json
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


def _worker_skewed(rank, world, port, n_seq, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = farm.assign(n_seq, rank, world, costs=[10, 1, 1, 1, 1])
    recs = [dict(scene_id=s, psnr=30.0 + s, l1=0.0, iters_per_s=1.0, n_splats=1, ok=1) for s in mine]
    table = farm.gather_records(recs, n_seq, torch.device("cpu"))
    torch.save((mine, table), os.path.join(out_dir, f"skew_{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_gather_records_keeps_every_record_of_a_skewed_assignment(tmp_path):
    """the LPT assignment can give one rank more than ceil(n / world) sequences (costs [10,1,1,1,1] over 2 ranks: one
    rank gets four); the gather must size its buffer by the largest rank and drop nothing."""
    n_seq, world = 5, 2
    mp.spawn(_worker_skewed, args=(world, _free_port(), n_seq, str(tmp_path)), nprocs=world, join=True)
    (m0, t0), (m1, t1) = torch.load(tmp_path / "skew_0.pt"), torch.load(tmp_path / "skew_1.pt")
    assert sorted(m0 + m1) == list(range(5)) and max(len(m0), len(m1)) == 4
    assert torch.equal(t0, t1)
    assert t0[:, 5].tolist() == [1.0] * 5 and t0[:, 1].tolist() == [30.0, 31.0, 32.0, 33.0, 34.0]


def test_latex_rows_and_log_scraper(tmp_path):
    """The two rows scripts/get_testing_psnr_davis.py:19-22 prints, and its scraper (:8-17): last number of the last line."""
    from das3r_amd.train import latex_rows, scrape_test_logs
    for scene, lines in (("bear_1", ["[ITER 2000] Evaluating test: L1 0.02 PSNR 21.5", "[ITER 4000] Evaluating test: L1 0.01 PSNR 24.25"]),
                         ("car-turn", ["[ITER 4000] Evaluating test: L1 0.03 PSNR 19.75"])):
        d = tmp_path / scene / "testing_pnsr_4000"
        d.mkdir(parents=True)
        (d / "test_log.txt").write_text("\n".join(lines) + "\n")
    (tmp_path / "not_a_scene.txt").write_text("x")
    res = scrape_test_logs(str(tmp_path), "testing_pnsr_4000")
    assert res == {"bear_1": 24.25, "car-turn": 19.75}
    head, row = latex_rows(res)
    assert head == "Scene & bear-1 & car-turn& average"
    assert row == "PSNR & 24.25 & 19.75 & 22.00 "


def test_mask_resize_is_nearest_neighbour():
    """scene/cameras.py:60-67: the ground-truth dynamic mask is repeated to 3 channels and nearest-resized to the render size."""
    from das3r_amd.train import resize_mask_nearest
    m = torch.zeros(4, 6, dtype=torch.bool)
    m[1, 2] = m[3, 5] = True
    out = resize_mask_nearest(m, 8, 12)
    assert out.shape == (3, 8, 12) and out.dtype == torch.float32
    assert torch.equal(out[0], out[1]) and torch.equal(out[0], out[2])
    assert out[0].sum() == 8 and out[0, 2:4, 4:6].sum() == 4 and out[0, 6:8, 10:12].sum() == 4
    down = resize_mask_nearest(m, 2, 3)       # source pixel = floor(dst * scale)
    assert down[0].tolist() == [[0.0, 0.0, 0.0], [0.0, 0.0, 0.0]]


def _guarded_worker(rank, world, port, n_seq, out_dir, die):
    """farm.main()'s end game: publish, let rank 0 decide, gather by collective or from the files."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dist.barrier()
    rdv = farm.Rendezvous(os.path.join(out_dir, ".farm"), rank, world, beat_s=0.1)
    if die and rank == 1:
        os._exit(3)                                  # a HIP fault: no record file, no more heartbeats, no goodbye
    recs = [dict(scene_id=s, psnr=20.0 + s, l1=0.01 * s, iters_per_s=100.0 + rank, n_splats=1000 * (s + 1), ok=1)
            for s in farm.assign(n_seq, rank, world)]
    rdv.publish(recs)
    mode = rdv.decide(stale_s=1.0)
    names = [f"seq_{i}" for i in range(n_seq)]
    table = farm.gather_records(recs, n_seq, torch.device("cpu")) if mode == "collective" else farm.table_from_files(rdv, n_seq, names, out_dir)
    rdv.close()
    torch.save((mode, table), os.path.join(out_dir, f"guarded_{rank}.pt"))
    os._exit(0)                                      # (no destroy_process_group: with a dead peer it would wait for it)


@pytest.mark.timeout(120)
@pytest.mark.parametrize("die", [False, True])
def test_one_dead_rank_does_not_hang_the_gather(tmp_path, die):
    """a rank that dies leaves the others in all_gather until the RCCL timeout.  With the file guard rank 0
    notices the stale heartbeat within seconds, nobody enters a collective, the table comes from the record files and — for a
    sequence of the dead rank that had already written its test_log.txt — from the log the reference's scripts scrape."""
    import time
    n_seq, world = 4, 2
    os.makedirs(tmp_path / "seq_1")
    (tmp_path / "seq_1" / "test_log.txt").write_text("[ITER 4000] Evaluating test: L1 0.01 PSNR 31.25\n")
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_guarded_worker, args=(r, world, port, n_seq, str(tmp_path), die)) for r in range(world)]
    t0 = time.time()
    for p in procs:
        p.start()
    for p in procs:
        p.join(60)
    assert all(not p.is_alive() for p in procs), "a rank is still blocked"
    assert time.time() - t0 < 45
    mode, table = torch.load(tmp_path / "guarded_0.pt")
    if not die:
        assert mode == "collective" and procs[1].exitcode == 0
        assert table[:, 1].tolist() == [20.0, 21.0, 22.0, 23.0] and table[:, 5].tolist() == [1, 1, 1, 1]
    else:
        assert mode == "files" and procs[1].exitcode == 3
        assert table[[0, 2], 1].tolist() == [20.0, 22.0] and table[[0, 2], 5].tolist() == [1, 1]     # rank 0's own sequences
        assert table[1, 1] == 31.25 and table[1, 5] == 1                                              # from seq_1/test_log.txt
        assert table[3, 5] == 0                                                                       # lost with its rank


def _scenario_worker(rank, world, port, out_dir, scenario):
    """Rendezvous scenarios of ADVICE r4 (no collective is needed to tell them apart: only the decision is recorded)."""
    import time
    rdv = farm.Rendezvous(os.path.join(out_dir, ".farm"), rank, world, beat_s=0.1)
    recs = [dict(scene_id=rank, psnr=20.0 + rank, l1=0.0, iters_per_s=1.0, n_splats=1, ok=1)]
    if scenario == "slow_rank_is_waited_for" and rank == 1:
        for _ in range(12):          # 3 s of work — three times the hung timeout — with progress ticks: slower, not hung
            time.sleep(0.25)
            rdv.tick()
    if scenario == "rank0_silent_then_late" and rank == 0:
        time.sleep(3.0)              # alive (heartbeat) but no progress for three hung timeouts: rank 1 gives up on the gather first
    if scenario == "hung_rank" and rank == 1:
        time.sleep(4.0)              # alive, never ticks, never publishes in time
    rdv.publish(recs)
    mode = rdv.decide(stale_s=10.0, hung_s=1.0)
    with open(os.path.join(out_dir, ".farm", "decision.json")) as f:
        by = json.load(f)["by"]
    rdv.close()
    torch.save((mode, by), os.path.join(out_dir, f"scenario_{rank}.pt"))
    os._exit(0)


@pytest.mark.timeout(120)
@pytest.mark.parametrize("scenario,mode,by", [("slow_rank_is_waited_for", "collective", 0), ("rank0_silent_then_late", "files", 1),
                                              ("hung_rank", "files", 0)])
def test_rendezvous_decision_is_one_and_fair(tmp_path, scenario, mode, by):
    """ADVICE r4 (farm.py Rendezvous.decide): (1) a rank that is merely slower — it keeps ticking — is waited for, however long the
    faster rank has been done (round 4 declared it hung after the deciding rank's own duration + 120 s and dropped its sequences);
    (2) the decision is written ONCE by whoever takes it first and honoured by everybody: a non-zero rank that gives up on a silent
    rank 0 writes "files", and rank 0, arriving later with every record file present, reads that instead of deciding "collective"
    and entering a gather its peer will never join; (3) a rank that lives (heartbeat) without progress is hung: files."""
    world = 2
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_scenario_worker, args=(r, world, 0, str(tmp_path), scenario)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(60)
    assert all(not p.is_alive() and p.exitcode == 0 for p in procs)
    got = [torch.load(tmp_path / f"scenario_{r}.pt") for r in range(world)]
    assert got[0] == got[1] == (mode, by), got


def _three_rank_worker(rank, world, out_dir):
    """rank 2 dies without publishing, rank 1 is healthy but slow (works 3 s beyond the decision, ticking), rank 0 is done at once."""
    import time
    rdv = farm.Rendezvous(os.path.join(out_dir, ".farm"), rank, world, beat_s=0.1)
    if rank == 2:
        time.sleep(0.3)
        os._exit(9)                  # a HIP fault: no records, the heartbeat stops
    recs = [dict(scene_id=s, psnr=20.0 + s, l1=0.0, iters_per_s=1.0, n_splats=1, ok=1) for s in farm.assign(6, rank, world)]
    if rank == 1:
        for _ in range(16):          # 4 s: well past the moment rank 0 sees rank 2's heartbeat go stale (1 s) and decides "files"
            time.sleep(0.25)
            rdv.tick()
    rdv.publish(recs)
    t_decided = time.time()
    mode = rdv.decide(stale_s=1.0, hung_s=2.0)
    states = rdv.wait_for_working_ranks(stale_s=1.0, hung_s=2.0)
    table = farm.table_from_files(rdv, 6)
    rdv.close()
    torch.save((mode, states, table, time.time() - t_decided), os.path.join(out_dir, f"three_{rank}.pt"))
    os._exit(0)


@pytest.mark.timeout(120)
def test_files_mode_waits_for_a_healthy_slower_rank_when_another_rank_is_dead(tmp_path):
    """ADVICE r5 (farm.py Rendezvous): world 3 — one dead rank, one rank that is done, one that is healthy and slower.  The decision
    "files" stands as soon as the dead rank's heartbeat is stale; the rank that assembles the table must nevertheless wait for the
    slower rank's records (it keeps ticking) and lose only the dead rank's sequences."""
    world = 3
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_three_rank_worker, args=(r, world, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(60)
    assert all(not p.is_alive() for p in procs)
    assert [p.exitcode for p in procs] == [0, 0, 9]
    for r in (0, 1):
        mode, states, table, waited = torch.load(tmp_path / f"three_{r}.pt")
        assert mode == "files" and states[2] == "dead" and "working" not in states, (mode, states)
        mine0, mine1, mine2 = (farm.assign(6, k, world) for k in range(3))
        assert table[mine0, 5].tolist() == [1.0] * len(mine0) and table[mine1, 5].tolist() == [1.0] * len(mine1), table   # both living ranks' sequences
        assert table[mine2, 5].tolist() == [0.0] * len(mine2)                                                              # lost with their rank
        assert table[mine1, 1].tolist() == [20.0 + s for s in mine1]
    assert torch.load(tmp_path / "three_0.pt")[3] > 2.0, "rank 0 must have waited for the slower rank instead of printing a partial table"


def test_run_jobs_on_the_host():
    """farm.run_jobs without a GPU: K worker threads pull the rank's sequences; results in the order of the items, never more than K
    in flight, a job's exception re-raised in the caller."""
    import threading
    import time
    lock, state = threading.Lock(), dict(now=0, peak=0)

    def job(i):
        with lock:
            state["now"] += 1
            state["peak"] = max(state["peak"], state["now"])
        time.sleep(0.02)
        with lock:
            state["now"] -= 1
        return i * i

    assert farm.run_jobs(range(9), job, 3) == [i * i for i in range(9)]
    assert 2 <= state["peak"] <= 3
    assert farm.run_jobs([4, 5], job, 1) == [16, 25] and farm.run_jobs([], job, 2) == []

    def bad(i):
        if i == 1:
            raise KeyError("sequence 1")
        return i

    with pytest.raises(KeyError):
        farm.run_jobs(range(3), bad, 2)
