"""Which binning path the forwards of a DAS3R-shaped train step take when the host is NOT synchronised with the device (the first 130 steps of
a shape, as bench.py times them): forwards on the global sort (depth_hist launches), partition passes per forward, ms per step.
    python tools/probes/binning_paths_async.py [smooth,noise] [repeats]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
dev = torch.device('cuda:0'); torch.cuda.set_device(0)
rk = bench.Ranks(bench.parse_args(['--gpus', '1']))
from das3r_amd import _lib
for depth in (sys.argv[1] if len(sys.argv) > 1 else "smooth").split(","):
    for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 3):
        _lib.forget_shapes()
        step, n = bench.train_step_timer(dev, fused=True, depth=depth)
        _lib.profile_report(); _lib.profile_enable(True)
        for _ in range(30): step()
        torch.cuda.synchronize(); warm = _lib.profile_report()
        t = rk.timed(step, 100, 0) / 100 * 1e3
        torch.cuda.synchronize(); k = _lib.profile_report(); _lib.profile_enable(False)
        print(f"{depth} run {rep}: warm-up 30 steps: radix forwards {warm.get('depth_hist_kernel', (0, 0))[0]}, passes {warm.get('onesweep_pass_kernel', (0, 0))[0]};"
              f" timed 100 steps: {t:.4f} ms per step, radix forwards {k.get('depth_hist_kernel', (0, 0))[0]}, passes {k.get('onesweep_pass_kernel', (0, 0))[0]}", flush=True)
