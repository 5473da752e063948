"""Host-side cost of the direct iteration (das3r_amd/fast_step.py): wall time per iteration with the GPU kept out of the way (a tiny model:
the kernels take microseconds, what is left is Python + ctypes + torch dispatch), and the same under cProfile."""
import os, sys, time, random, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from types import SimpleNamespace
from das3r_amd.model import OptimParams
from das3r_amd.train import build_from_sequence, consistent_sequence, train_step
dev = torch.device("cuda:0")
seq = consistent_sequence(frames=12, W=64, H=48, focal=80.0, n_splats=800, seed=0)
model, cams, test = build_from_sequence(seq, heldout=True)
opt = OptimParams(iterations=4000); model.training_setup(opt, fused=True)
pipe = SimpleNamespace(debug=False, compute_cov3D_python=False, convert_SHs_python=False)
bg = torch.zeros(3, device=dev); rng = random.Random(0)
def run(n, it0):
    for k in range(n):
        train_step(model, cams[rng.randint(0, len(cams) - 1)], opt, it0 + k, pipe, bg, fused=True)
run(200, 1); torch.cuda.synchronize()
t = time.perf_counter(); run(2000, 201); t_host = time.perf_counter() - t; torch.cuda.synchronize(); t_all = time.perf_counter() - t
print(f"host {t_host / 2000 * 1e3:.3f} ms per iteration to ENQUEUE, {t_all / 2000 * 1e3:.3f} ms until the GPU is done (tiny model)")
pr = cProfile.Profile(); pr.enable(); run(500, 2300); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
