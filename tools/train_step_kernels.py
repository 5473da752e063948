"""The DAS3R-shaped fused train step (noise and smooth depth maps): ms per step under bench.py's timing protocol and the per-kernel
table of the library's own profiler (HIP events around every launch).   python tools/train_step_kernels.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device('cuda:0'); torch.cuda.set_device(0)
rk = bench.Ranks(bench.parse_args(['--gpus', '1']))
from das3r_amd import _lib
for depth in ('noise', 'smooth'):
    step, n = bench.train_step_timer(dev, fused=True, depth=depth)
    for _ in range(30): step()
    t = rk.timed(step, 100, 10) / 100 * 1e3
    _lib.profile_enable(True)
    for _ in range(20): step()
    torch.cuda.synchronize(); rep = _lib.profile_report(); _lib.profile_enable(False)
    print(depth, 'train step ms', round(t, 4), {k: round(v[1] / 20, 4) for k, v in sorted(rep.items(), key=lambda kv: -kv[1][1])[:14]})
