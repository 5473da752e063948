"""Which binning path the first iterations of a job take on a self-consistent sequence (passes, segment sort, global sort)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from types import SimpleNamespace
from das3r_amd import _lib
from das3r_amd.model import OptimParams
from das3r_amd.train import build_from_sequence, consistent_sequence, train_step
dev = torch.device("cuda:0")
seq = consistent_sequence(frames=22, W=512, H=208, focal=600.0, n_splats=20000, seed=0)
model, cams, test = build_from_sequence(seq, heldout=True)
opt = OptimParams(iterations=4000); model.training_setup(opt, fused=True)
pipe = SimpleNamespace(debug=False, compute_cov3D_python=False, convert_SHs_python=False)
bg = torch.zeros(3, device=dev)
for it in range(1, int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    _lib.profile_report(); _lib.profile_enable(True)
    _, _, pkg = train_step(model, cams[(3 * it) % len(cams)], opt, it, pipe, bg, fused=True)
    torch.cuda.synchronize(); _lib.profile_enable(False)
    k = _lib.profile_report()
    print(it, "passes", k.get("onesweep_pass_kernel", (0, 0))[0], "segsort ms", round(k.get("segment_sort_kernel", (0, 0.0))[1], 4), "radix" if "depth_hist_kernel" in k else "seg",
          "fwd", [n for n in k if n.startswith("render_forward")], "I", int((pkg["radii"] > 0).sum()))
