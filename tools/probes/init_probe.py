import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from das3r_amd.model import OptimParams
from das3r_amd.train import build_from_sequence, consistent_sequence
import das3r_amd.train as T, das3r_amd.model as M
dev = torch.device("cuda:0")
seq = consistent_sequence(frames=22, W=512, H=208, focal=600.0, n_splats=20000, seed=0)
def tm(f, n=3):
    best = 1e9
    for _ in range(n):
        torch.cuda.synchronize(); t = time.perf_counter(); r = f(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
    return best, r
t, (model, cams, test) = tm(lambda: build_from_sequence(seq, heldout=True))
print("build_from_sequence %.3f s" % t)
import cProfile, pstats
pr = cProfile.Profile(); torch.cuda.synchronize(); pr.enable(); build_from_sequence(seq, heldout=True); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
opt = OptimParams(iterations=4000)
t, _ = tm(lambda: model.training_setup(opt, fused=True)); print("training_setup %.3f s" % t)
