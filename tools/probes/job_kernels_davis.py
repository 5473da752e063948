#!/usr/bin/env python3
"""Per-kernel table (the library's HIP-event profiler) of the direct iteration on the SELF-CONSISTENT Sintel-shaped sequence — the data a
farm job runs on — early in the job and after N iterations of training (scales, opacities and poses have moved by then).
    python tools/probes/job_kernels.py [iterations before the second table = 1500]"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from types import SimpleNamespace
from das3r_amd import _lib
from das3r_amd.model import OptimParams
from das3r_amd.train import build_from_sequence, consistent_sequence, train_step
dev = torch.device("cuda:0")
later = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
seq = consistent_sequence(frames=50, W=512, H=288, focal=614.4, n_splats=60000, seed=0)
model, cams, test = build_from_sequence(seq, heldout=True)
opt = OptimParams(iterations=4000); model.training_setup(opt, fused=True)
pipe = SimpleNamespace(debug=False, compute_cov3D_python=False, convert_SHs_python=False)
bg = torch.zeros(3, device=dev); rng = random.Random(0)
it = 0
def run(n, profile=False):
    global it
    if profile:
        _lib.profile_report(); _lib.profile_enable(True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        it += 1
        train_step(model, cams[rng.randint(0, len(cams) - 1)], opt, it, pipe, bg, fused=True)
    e1.record(); torch.cuda.synchronize()
    if profile:
        _lib.profile_enable(False)
        rep = _lib.profile_report()
        print(f"  iterations {it - n + 1}..{it}: kernels (ms per iteration):", {k: round(v[1] / n, 4) for k, v in sorted(rep.items(), key=lambda kv: -kv[1][1])[:12]})
    return e0.elapsed_time(e1) / n
run(40)
print("early: %.4f ms per iteration" % run(100)); run(40, True)
run(max(later - it, 0))
print("after %d iterations: %.4f ms per iteration" % (it, run(100))); run(40, True)
