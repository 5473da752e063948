"""Multi-GPU sequence farm: independent sequences are optimised one per GPU; the only collective is the
gather of one fixed-size record per sequence at the end (RCCL on GPUs, gloo in the CPU tests).

The reference runs its per-scene loops serially on GPU 0 (/root/reference/scripts/testing_psnr_davis.sh:3,35-59) and
scrapes the PSNR out of log files (/root/reference/scripts/get_testing_psnr_davis.py:8-22); the slicing idiom follows
the reference's own farm for the predictor stage (/root/reference/dynamic_predictor/dust3r/pose_eval.py:53-68).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 -m das3r_amd.farm \
        --sequences 8 --iterations 200
"""
import argparse
import os

import torch
import torch.distributed as dist

RECORD_FIELDS = ("scene_id", "psnr", "l1", "iters_per_s", "n_splats", "ok")   # one float64 row per sequence


def assign(n_sequences, rank, world, costs=None):
    """Sequence indices for `rank`.  Without costs: round-robin (rank, rank+world, ...).  With costs (e.g. N_VIEWS*H*W):
    longest-processing-time-first greedy, deterministic, same result on every rank."""
    if costs is None:
        return list(range(rank, n_sequences, world))
    order = sorted(range(n_sequences), key=lambda i: (-costs[i], i))
    loads, bins = [0.0] * world, [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (loads[k], k))
        bins[r].append(i)
        loads[r] += costs[i]
    return sorted(bins[rank])


def gather_records(local_records, n_sequences, device):
    """all_gather of fixed-size records; returns an (n_sequences, len(RECORD_FIELDS)) float64 tensor on every rank, rows
    ordered by scene id; sequences nobody reported (failed before producing a record) have ok = 0."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    per_rank = (n_sequences + world - 1) // world
    buf = torch.full((per_rank, len(RECORD_FIELDS)), -1.0, dtype=torch.float64, device=device)
    for k, rec in enumerate(local_records[:per_rank]):
        buf[k] = torch.tensor([float(rec[f]) for f in RECORD_FIELDS], dtype=torch.float64)
    if world > 1:
        out = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(out, buf)
        allrec = torch.cat(out, 0)
    else:
        allrec = buf
    table = torch.zeros(n_sequences, len(RECORD_FIELDS), dtype=torch.float64)
    table[:, 0] = torch.arange(n_sequences, dtype=torch.float64)
    for row in allrec.cpu():
        sid = int(row[0])
        if 0 <= sid < n_sequences:
            table[sid] = row
    return table


def sequence_cost(seq_dir):
    """Relative cost of a preprocessed sequence directory for the LPT assignment: frames x pixels (one Gaussian per pixel)."""
    from .io_formats import read_colmap_cameras_text
    cams = read_colmap_cameras_text(os.path.join(seq_dir, "sparse/0/cameras.txt"))
    return float(sum(c["width"] * c["height"] for c in cams.values()))


def run_sequence_job(scene_id, iterations, device, frames=6, seq_dir=None, out_dir=None, fused=False):
    """One independent 'sequence': load a preprocessed DAS3R sequence directory (das3r_amd.io_formats.load_sequence) — or,
    without one, build a synthetic multi-frame scene —, optimise it with the train-step harness, report the held-out PSNR and,
    with out_dir, write what the reference writes (point_cloud/iteration_N/point_cloud.ply, pose/pose_N.npy:
    train_gui.py:467-480,523-528).  Failures are isolated per sequence (the reference's predictor farm does the same,
    pose_eval.py:209-222)."""
    from .model import OptimParams
    from .train import build_from_sequence, is_test_index, psnr_report, synthetic_sequence, train
    try:
        masks = None
        if seq_dir is not None:
            from .io_formats import load_sequence
            seq = load_sequence(seq_dir, device=device)
            masks = seq.get("dynamic_masks")
        else:
            seq = synthetic_sequence(frames=frames, seed=scene_id, device=device)
        model, cams = build_from_sequence(seq)
        opt = OptimParams(iterations=iterations)
        model.training_setup(opt, fused=fused)
        test = [c for c in cams if is_test_index(c.uid)] or cams[-1:]
        train_cams = [c for c in cams if c not in test] or cams
        stats = train(model, train_cams, opt, iterations, seed=scene_id, fused=fused)
        dyn = None
        if masks is not None and all(m is not None for m in masks):
            dyn = {i: torch.from_numpy(m).to(device) for i, m in enumerate(masks)}
        rep = psnr_report(model, test, dynamic_masks=dyn)
        if out_dir is not None:
            from .io_formats import save_model_ply, save_poses_npy
            save_model_ply(os.path.join(out_dir, "point_cloud", f"iteration_{iterations}", "point_cloud.ply"), model)
            save_poses_npy(os.path.join(out_dir, "pose", f"pose_{iterations}.npy"), [model.get_RT(i) for i in range(len(cams))])
        return dict(scene_id=scene_id, psnr=rep["psnr"], l1=rep["l1"], iters_per_s=stats["iters_per_s"],
                    n_splats=model.get_xyz.shape[0], ok=1)
    except Exception as ex:  # noqa: BLE001 - keep the farm alive, report the failure in the table
        print(f"[farm] sequence {scene_id} failed: {ex!r}")
        return dict(scene_id=scene_id, psnr=float("nan"), l1=float("nan"), iters_per_s=0.0, n_splats=0, ok=0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sequences", type=int, default=8)
    ap.add_argument("--iterations", type=int, default=200)
    ap.add_argument("--backend", default=None)
    ap.add_argument("--data", default=None, help="directory whose sub-directories are preprocessed DAS3R sequences")
    ap.add_argument("--out", default=None, help="where to write <sequence>/point_cloud/... and pose/...")
    ap.add_argument("--fused", action="store_true", help="use the fused pre-transform / Adam / loss kernels")
    args = ap.parse_args()
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    from .hostpin import pin_to_ccx
    pin_to_ccx(local)   # one core complex per rank (hostpin.py), before the first HIP call
    use_gpu = torch.cuda.is_available()
    device = torch.device("cuda", local) if use_gpu else torch.device("cpu")
    if use_gpu:
        torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=args.backend or ("nccl" if use_gpu else "gloo"))
    if args.data:   # real sequences: every rank lists the same sorted directory, longest first across ranks
        dirs = sorted(d for d in os.listdir(args.data) if os.path.isfile(os.path.join(args.data, d, "sparse/0/cameras.txt")))
        args.sequences = len(dirs)
        mine = assign(len(dirs), rank, world, costs=[sequence_cost(os.path.join(args.data, d)) for d in dirs])
        records = [run_sequence_job(s, args.iterations, device, seq_dir=os.path.join(args.data, dirs[s]),
                                    out_dir=os.path.join(args.out, dirs[s]) if args.out else None, fused=args.fused) for s in mine]
    else:
        mine = assign(args.sequences, rank, world)
        records = [run_sequence_job(s, args.iterations, device, fused=args.fused) for s in mine]
    table = gather_records(records, args.sequences, device)
    if rank == 0:
        good = table[table[:, 5] > 0]
        print(" & ".join(f"{p:.2f}" for p in table[:, 1].tolist()))     # the LaTeX row get_testing_psnr_davis.py prints
        print(f"mean PSNR {good[:, 1].mean().item():.2f} over {good.shape[0]}/{args.sequences} sequences")
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
