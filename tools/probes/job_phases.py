"""Where a whole Sintel-shaped job spends its time: ms per training iteration by phase of the schedule (SH degree 0 / 1), the held-out
pose passes, initialisation and report — wall clock around synchronised stretches of das3r_amd.train's own loop."""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from types import SimpleNamespace
from das3r_amd.model import OptimParams
from das3r_amd.train import build_from_sequence, consistent_sequence, psnr_report, test_pose_pass, train_step
dev = torch.device("cuda:0")
seq = consistent_sequence(frames=22, W=512, H=208, focal=600.0, n_splats=20000, seed=0)
torch.cuda.synchronize(); t0 = time.perf_counter()
model, cams, test = build_from_sequence(seq, heldout=True)
opt = OptimParams(iterations=4000); model.training_setup(opt, fused=True)
torch.cuda.synchronize(); t_init = time.perf_counter() - t0
pipe = SimpleNamespace(debug=False, compute_cov3D_python=False, convert_SHs_python=False)
bg = torch.zeros(3, device=dev); rng = random.Random(0)
dyn = {c.uid: torch.from_numpy(seq["gt_dynamic_masks"][c.frame_index]).to(dev) for c in test}
acc = {"deg0": [0.0, 0], "deg1": [0.0, 0], "test_pass": [0.0, 0]}
stack = []
def timed(key, fn, n=1):
    torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize(); acc[key][0] += time.perf_counter() - t; acc[key][1] += n
it = 0
while it < 4000:
    if not stack: stack = list(cams)
    chunk = min(len(stack), 4000 - it)
    def run():
        global it
        for _ in range(chunk):
            it += 1
            cam = stack.pop(rng.randint(0, len(stack) - 1))
            train_step(model, cam, opt, it, pipe, bg, fused=True)
    timed("deg0" if it < 3000 else "deg1", run, chunk)
    if not stack: timed("test_pass", lambda: test_pose_pass(model, test, dyn, opt, pipe, bg, rng, fused=True), len(test))
torch.cuda.synchronize(); t = time.perf_counter(); rep = psnr_report(model, test, dynamic_masks=dyn, test_poses=True); torch.cuda.synchronize(); t_rep = time.perf_counter() - t
print("init %.3f s, report %.3f s, PSNR %.2f" % (t_init, t_rep, rep["psnr"]))
for k, (s, n) in acc.items(): print("%-10s %6d units, %.3f s total, %.3f ms each" % (k, n, s, 1e3 * s / max(n, 1)))
