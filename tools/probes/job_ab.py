#!/usr/bin/env python3
"""Whole Sintel-shaped job on the self-consistent sequence under two settings of an environment switch (A-B on one box).
    python tools/probes/job_ab.py DAS3R_RENDER=slices [iterations]  (the switch may also be one that turns a default OFF: DAS3R_TILE_LPT=0)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from das3r_amd import _lib
from das3r_amd.farm import run_sequence_job
from das3r_amd.train import consistent_sequence
kv = sys.argv[1]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
dev = torch.device("cuda:0")
seq = consistent_sequence(seed=0, frames=22, W=512, H=208, focal=600.0, n_splats=20000)
run_sequence_job(0, 60, dev, fused=True, seq=consistent_sequence(frames=12, W=128, H=80, focal=150.0, n_splats=3000, seed=99))
for setting in ("", kv, "", kv):
    k, _, v = setting.partition("=")
    if k:
        os.environ[k] = v
    _lib.reload_switches()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rec = run_sequence_job(0, iters, dev, fused=True, seq=seq)
    torch.cuda.synchronize()
    print(f"[{setting or 'default'}] wall {time.perf_counter() - t0:.2f} s  psnr {rec['psnr']:.3f}  it/s {rec['iters_per_s']:.1f}", flush=True)
    if k:
        del os.environ[k]
