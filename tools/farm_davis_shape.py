#!/usr/bin/env python3
"""The reference's REAL workload size end to end: a DAVIS-shaped sequence — 50 frames of 512x288
(scripts/testing_psnr_davis.sh:35-59 runs eight of them, dynamic_predictor/dust3r/utils/image.py:123-144 fixes the size) — written
to disk in the preprocessed layout (das3r_amd.io_formats.write_sequence_dir), loaded back, initialised with distCUDA2, optimised
for 4000 iterations with the fused kernels, held-out report, PLY + poses written: what `python -m das3r_amd.farm --data ... --fused
--iterations 4000` does for one sequence, with the clock and the allocator's high-water mark around it.

    python tools/farm_davis_shape.py [--frames 50 --width 512 --height 288 --iterations 4000] --out profiles/r03_farm_davis_shape.json
The scene is synthetic (a static random splat cloud seen from slowly moving poses): the datasets are not in the container; the
point is the SIZE: 45 training frames x 147 456 pixels = 6.6 M Gaussians (7.4 M if no frame were held out)."""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=50)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--height", type=int, default=288)
    ap.add_argument("--iterations", type=int, default=4000)
    ap.add_argument("--no-fused", action="store_true")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    from das3r_amd import io_formats as io
    from das3r_amd.farm import run_sequence_job, sequence_cost
    from das3r_amd.train import synthetic_sequence
    dev = torch.device("cuda:0")
    work = tempfile.mkdtemp(prefix="das3r_davis_shape_")
    res = dict(frames=args.frames, image=[args.width, args.height], iterations=args.iterations, fused=not args.no_fused)
    try:
        t0 = time.perf_counter()
        seq = synthetic_sequence(frames=args.frames, W=args.width, H=args.height, focal=1.2 * args.width, n_splats=60000, seed=11, device="cuda:0")
        d = os.path.join(work, "data", "davis_shape")
        io.write_sequence_dir(seq, d)
        del seq
        torch.cuda.empty_cache()
        res["write_sequence_s"] = round(time.perf_counter() - t0, 2)
        res["sequence_bytes_on_disk"] = sum(os.path.getsize(os.path.join(r, f)) for r, _, fs in os.walk(d) for f in fs)
        assert sequence_cost(d) == args.frames * args.width * args.height
        torch.cuda.reset_peak_memory_stats()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        out = os.path.join(work, "out")
        rec = run_sequence_job(0, args.iterations, dev, seq_dir=d, out_dir=out, fused=not args.no_fused)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t1
        res.update(ok=int(rec["ok"]), n_splats=int(rec["n_splats"]), heldout_psnr=rec["psnr"], heldout_l1=rec["l1"],
                   train_iters_per_s=round(rec["iters_per_s"], 2), train_step_ms=round(1e3 / max(rec["iters_per_s"], 1e-9), 3),
                   job_wall_s=round(wall, 2), peak_hbm_bytes=int(torch.cuda.max_memory_allocated()),
                   peak_hbm_reserved_bytes=int(torch.cuda.max_memory_reserved()),
                   what="load_sequence (PNG / npy / COLMAP text / TUM) -> distCUDA2 init -> 4000 fused train steps -> held-out pose pass + PSNR "
                        "report -> point_cloud.ply + pose npy (das3r_amd.farm.run_sequence_job)")
        ply = os.path.join(out, "point_cloud", f"iteration_{args.iterations}", "point_cloud.ply")
        res["ply_bytes"] = os.path.getsize(ply) if os.path.exists(ply) else 0
        log = os.path.join(out, "test_log.txt")
        res["test_log"] = open(log).read().strip() if os.path.exists(log) else None
    finally:
        shutil.rmtree(work, ignore_errors=True)
    print(json.dumps(res, indent=1))
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
