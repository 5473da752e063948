#!/usr/bin/env python3
"""Held-out static-region PSNR of whole Sintel-shaped jobs on the self-consistent sequence with the PREDICTOR'S errors put back in
(train.consistent_sequence(depth_noise=, pose_noise=)): which noise lands a job in the regime the published numbers live in
(29.03 dB on Sintel market_2, 25.70 dB DAVIS mean: /root/reference index.html table 2).   python tools/noisy_psnr_sweep.py [--small]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--small", action="store_true")
    ap.add_argument("--grid", default="0:0,0.02:0,0.05:0,0:0.005,0:0.01,0:0.02,0.02:0.01,0.05:0.02,0.1:0.02,0.1:0.04")
    ap.add_argument("--iterations", type=int, default=4000)
    args = ap.parse_args()
    from das3r_amd.farm import run_sequence_job
    from das3r_amd.train import consistent_sequence
    dev = torch.device("cuda:0")
    shape = dict(frames=12, W=256, H=104, focal=300.0, n_splats=8000) if args.small else dict(frames=22, W=512, H=208, focal=600.0, n_splats=20000)
    rows = []
    for cell in args.grid.split(","):
        dn, pn = (float(x) for x in cell.split(":"))
        seq = consistent_sequence(seed=0, depth_noise=dn, pose_noise=pn, **shape)
        start = run_sequence_job(0, 20, dev, fused=True, seq=seq)
        rec = run_sequence_job(0, args.iterations, dev, fused=True, seq=seq)
        rows.append(dict(depth_noise=dn, pose_noise=pn, psnr_20=round(start["psnr"], 2), psnr=round(rec["psnr"], 2), ok=rec["ok"], iters_per_s=round(rec["iters_per_s"], 1)))
        print(rows[-1], flush=True)
    print(json.dumps(rows))


if __name__ == "__main__":
    main()
