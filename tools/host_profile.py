#!/usr/bin/env python3
"""Host-side cost of one fwd+bwd step through the drop-in autograd surface (cProfile), workload c2."""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from das3r_amd import GaussianRasterizationSettings, GaussianRasterizer  # noqa: E402
from das3r_amd.synth import make_workload  # noqa: E402

dev = torch.device("cuda:0")
sc = make_workload(sys.argv[1] if len(sys.argv) > 1 else "c2").to(dev)
rs = GaussianRasterizationSettings(**sc.settings_kwargs())
rast = GaussianRasterizer(rs)
leaves = [t.clone().requires_grad_(True) for t in (sc.means3D, sc.shs, sc.opacities, sc.scales, sc.rotations)]
m2 = torch.zeros_like(sc.means3D, requires_grad=True)


def step():
    color, radii = rast(means3D=leaves[0], means2D=m2, shs=leaves[1], opacities=leaves[2], scales=leaves[3], rotations=leaves[4])
    color.backward(sc.dL_dpix)
    for t in leaves + [m2]:
        t.grad = None


for _ in range(20):
    step()
torch.cuda.synchronize()
K = 200
t0 = time.perf_counter()
for _ in range(K):
    step()
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"host enqueue {t_host / K * 1e3:.3f} ms/step, with final sync {t_all / K * 1e3:.3f} ms/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(K):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
