#!/usr/bin/env python3
"""train-step ms of the DAS3R hot loop (render + masked L1/SSIM loss + backward + two Adam steps), on a synthetic sequence
with the shape real DAS3R training has: every pixel of every frame is one Gaussian, frames 512x208.

    python tools/train_bench.py [--frames 20 --W 512 --H 208 --iters 50] [--fused-adam] [--breakdown]
Prints one JSON line: {"train_step_ms": ..., "splats": P, "iters_per_s": ..., "breakdown_ms": {...}}."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=20)
    ap.add_argument("--W", type=int, default=512)
    ap.add_argument("--H", type=int, default=208)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--fused-adam", action="store_true")
    ap.add_argument("--fused-loss", action="store_true")
    ap.add_argument("--fused-pre", action="store_true")
    ap.add_argument("--breakdown", action="store_true")
    ap.add_argument("--depth", default="noise", choices=("noise", "smooth"), help="depth maps of the synthetic sequence (das3r_amd.train.synthetic_sequence)")
    args = ap.parse_args()
    from types import SimpleNamespace
    from das3r_amd.model import OptimParams
    from das3r_amd.train import build_from_sequence, synthetic_sequence, train_step
    seq = synthetic_sequence(frames=args.frames, W=args.W, H=args.H, focal=600.0, n_splats=20000, seed=0, depth=args.depth)
    model, cams = build_from_sequence(seq)
    opt = OptimParams(iterations=4000)
    model.training_setup(opt, fused=args.fused_adam) if args.fused_adam else model.training_setup(opt)
    pipe = SimpleNamespace(debug=False, compute_cov3D_python=False, convert_SHs_python=False)
    bg = torch.zeros(3, device="cuda")
    kw = dict(fused=True) if args.fused_pre else {}
    for it in range(1, args.warmup + 1):
        train_step(model, cams[it % len(cams)], opt, it, pipe, bg, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(args.warmup + 1, args.warmup + args.iters + 1):
        train_step(model, cams[it % len(cams)], opt, it, pipe, bg, **kw)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.iters * 1e3
    out = {"train_step_ms": round(ms, 3), "splats": int(model.get_xyz.shape[0]), "frames": args.frames, "image": [args.W, args.H],
           "iters_per_s": round(1e3 / ms, 2), "fused_adam": bool(args.fused_adam), "fused_pre": bool(args.fused_pre)}
    if args.breakdown:
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            for it in range(1000, 1005):
                train_step(model, cams[it % len(cams)], opt, it, pipe, bg, **kw)
            torch.cuda.synchronize()
        rows = sorted(prof.key_averages(), key=lambda e: -e.device_time_total)[:40]
        out["breakdown_ms"] = {r.key[:60]: round(r.device_time_total / 5 / 1e3, 3) for r in rows}
        out["kernel_launches_per_step"] = sum(r.count for r in prof.key_averages()) / 5
        out["device_ms_per_step"] = round(sum(r.device_time_total for r in prof.key_averages()) / 5 / 1e3, 3)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
