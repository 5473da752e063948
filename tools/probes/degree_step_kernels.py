import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from das3r_amd import _lib
dev = torch.device('cuda:0'); torch.cuda.set_device(0)
rk = bench.Ranks(bench.parse_args(['--gpus', '1']))
from types import SimpleNamespace
from das3r_amd.model import OptimParams
from das3r_amd.train import build_from_sequence, consistent_sequence, train_step
seq = consistent_sequence(frames=22, W=512, H=208, focal=600.0, n_splats=20000, seed=0)
model, cams, test = build_from_sequence(seq, heldout=True)
opt = OptimParams(iterations=4000); model.training_setup(opt, fused=True)
pipe = SimpleNamespace(debug=False, compute_cov3D_python=False, convert_SHs_python=False)
bg = torch.zeros(3, device=dev)
for deg in (0, 1, 2):
    model.active_sh_degree = deg; model.optimizer.set_active_sh_degree(deg)
    it = [10 + 100 * deg]
    def step():
        it[0] += 1
        train_step(model, cams[it[0] % len(cams)], opt, it[0], pipe, bg, fused=True)
    for _ in range(20): step()
    t = rk.timed(step, 60, 5) / 60 * 1e3
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(10): step()
        torch.cuda.synchronize()
    rows = sorted(((e.key[:60], e.device_time_total / 10 / 1e3) for e in prof.key_averages() if e.device_time_total > 0), key=lambda r: -r[1])[:14]
    print("degree", deg, "step ms", round(t, 4), [(k, round(v, 4)) for k, v in rows])
