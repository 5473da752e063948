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
    ordered by scene id; sequences nobody reported (failed before producing a record) have ok = 0.  The per-rank buffer is
    sized by the LARGEST number of records any rank holds (one all_reduce MAX): an LPT assignment can give a rank more than
    ceil(n / world) sequences, and nothing is ever truncated."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    nloc = torch.tensor([len(local_records)], dtype=torch.int64, device=device)
    if world > 1:
        dist.all_reduce(nloc, op=dist.ReduceOp.MAX)
    per_rank = max(int(nloc.item()), 1)
    if len(local_records) > per_rank:
        raise RuntimeError("gather_records: more local records than the agreed buffer holds")
    buf = torch.full((per_rank, len(RECORD_FIELDS)), -1.0, dtype=torch.float64, device=device)
    for k, rec in enumerate(local_records):
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


def run_sequence_job(scene_id, iterations, device, frames=6, seq_dir=None, out_dir=None, fused=False, gt_mask_dir=None, dataset="sintel"):
    """One independent 'sequence': load a preprocessed DAS3R sequence directory (das3r_amd.io_formats.load_sequence) — or,
    without one, build a synthetic multi-frame scene —, optimise it with the train-step harness, report the held-out PSNR and,
    with out_dir, write what the reference writes (point_cloud/iteration_N/point_cloud.ply, pose/pose_N.npy:
    train_gui.py:467-480,523-528).  Failures are isolated per sequence (the reference's predictor farm does the same,
    pose_eval.py:209-222)."""
    from .model import OptimParams
    from .train import build_from_sequence, psnr_report, synthetic_sequence, train
    try:
        masks = None
        if seq_dir is not None:
            from .io_formats import load_sequence
            seq = load_sequence(seq_dir, device=device, gt_mask_dir=gt_mask_dir, dataset=dataset)
            masks = seq.get("gt_dynamic_masks")   # ground-truth masks only: the report skips views without one
        else:
            seq = synthetic_sequence(frames=frames, seed=scene_id, device=device)
        # Gaussians, training poses and conf_static from the TRAINING frames only; the held-out frames ((idx + 5) % 10 == 0) give
        # their poses and their images as ground truth (scene/__init__.py:88-93, dataset_readers.py:342-347)
        model, train_cams, test = build_from_sequence(seq, heldout=True)
        opt = OptimParams(iterations=iterations)
        model.training_setup(opt, fused=fused)
        dyn = None
        if masks is not None and any(m is not None for m in masks):   # keyed by the test camera's uid; views without a mask are skipped
            dyn = {c.uid: (torch.from_numpy(masks[c.frame_index]).to(device) if masks[c.frame_index] is not None else None) for c in test}
        stats = train(model, train_cams, opt, iterations, seed=scene_id, fused=fused, test_cameras=test, gt_dynamic_masks=dyn)
        rep = psnr_report(model, test, dynamic_masks=dyn, test_poses=True, iteration=iterations, log_dir=out_dir)
        cams = train_cams
        if out_dir is not None:
            from .io_formats import save_model_ply, save_poses_npy
            save_model_ply(os.path.join(out_dir, "point_cloud", f"iteration_{iterations}", "point_cloud.ply"), model)
            save_poses_npy(os.path.join(out_dir, "pose", f"pose_{iterations}.npy"), [model.get_RT(i) for i in range(len(cams))])
        # a report over zero views (ground-truth masks exist for the sequence but none of the held-out views has one) is no result:
        # ok = 0 keeps its NaN out of the table's mean
        import math
        return dict(scene_id=scene_id, psnr=rep["psnr"], l1=rep["l1"], iters_per_s=stats["iters_per_s"],
                    n_splats=model.get_xyz.shape[0], ok=int(rep["views"] > 0 and math.isfinite(rep["psnr"])))
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
    ap.add_argument("--gt-dynamic-mask", default=None, help="root of the ground-truth dynamic masks, <root>/<sequence>/... (train_test_psnr.py --gt_dynamic_mask)")
    ap.add_argument("--dataset", default="sintel", choices=("sintel", "davis"))
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
                                    out_dir=os.path.join(args.out, dirs[s]) if args.out else None, fused=args.fused,
                                    gt_mask_dir=os.path.join(args.gt_dynamic_mask, dirs[s]) if args.gt_dynamic_mask else None,
                                    dataset=args.dataset) for s in mine]
    else:
        mine = assign(args.sequences, rank, world)
        records = [run_sequence_job(s, args.iterations, device, fused=args.fused) for s in mine]
    table = gather_records(records, args.sequences, device)
    if rank == 0:
        from .train import latex_rows
        good = table[table[:, 5] > 0]
        names = dirs if args.data else [f"seq_{i}" for i in range(args.sequences)]
        head, row = latex_rows({names[int(r[0])]: float(r[1]) for r in good})   # the rows get_testing_psnr_davis.py:19-22 prints
        print(head)
        print(row)
        print(f"mean PSNR {good[:, 1].mean().item():.2f} over {good.shape[0]}/{args.sequences} sequences")
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
