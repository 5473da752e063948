"""How many iterations of a whole job take which binning path (segmented with 2 / 3 partition passes, global radix sort), per shape.
    python tools/probes/job_binning_paths.py sintel|davis [iterations]"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from types import SimpleNamespace
from das3r_amd import _lib
from das3r_amd.model import OptimParams
from das3r_amd.train import build_from_sequence, consistent_sequence, train_step
shape = dict(sintel=dict(frames=22, W=512, H=208, focal=600.0, n_splats=20000), davis=dict(frames=50, W=512, H=288, focal=614.4, n_splats=60000))[sys.argv[1]]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
dev = torch.device("cuda:0")
seq = consistent_sequence(seed=0, **shape)
model, cams, test = build_from_sequence(seq, heldout=True)
opt = OptimParams(iterations=4000); model.training_setup(opt, fused=True)
pipe = SimpleNamespace(debug=False, compute_cov3D_python=False, convert_SHs_python=False)
bg = torch.zeros(3, device=dev); rng = random.Random(0)
_lib.forget_shapes()
count, segms, ms = {}, 0.0, {}
for it in range(1, iters + 1):
    _lib.profile_report(); _lib.profile_enable(True)
    train_step(model, cams[rng.randint(0, len(cams) - 1)], opt, it, pipe, bg, fused=True)
    torch.cuda.synchronize(); _lib.profile_enable(False)
    k = _lib.profile_report()
    path = "radix" if "depth_hist_kernel" in k else f"seg{k.get('onesweep_pass_kernel', (0, 0))[0]}"
    count[path] = count.get(path, 0) + 1
    b = sum(v[1] for n, v in k.items() if n in ("onesweep_pass_kernel", "segment_sort_kernel", "scan_emit_kernel", "depth_hist_kernel", "tile_ranges_kernel"))
    ms[path] = ms.get(path, 0.0) + b
    if it in (100, 500, 1000, 2000, 3000, 4000):
        print(it, count, {p: round(ms[p] / count[p], 4) for p in count}, flush=True)
