"""Tile list lengths of a training forward on a self-consistent Sintel-shaped sequence (the spread that bounds the one-workgroup-per-tile
forward), and the compositing kernels' times of that iteration."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from types import SimpleNamespace
from das3r_amd import _lib, fast_step
from das3r_amd.model import OptimParams
from das3r_amd.rasterizer import _forward_full
from das3r_amd.train import build_from_sequence, consistent_sequence, synthetic_sequence, train_step
dev = torch.device("cuda:0")
kind = sys.argv[1] if len(sys.argv) > 1 else "rendered"
seq = synthetic_sequence(frames=22, W=512, H=208, focal=600.0, n_splats=20000, seed=0, depth=kind)
model, cams, test = build_from_sequence(seq, heldout=True)
opt = OptimParams(iterations=4000); model.training_setup(opt, fused=True)
pipe = SimpleNamespace(debug=False, compute_cov3D_python=False, convert_SHs_python=False)
bg = torch.zeros(3, device=dev)
for it in range(1, 40): train_step(model, cams[(3 * it) % len(cams)], opt, it, pipe, bg, fused=True)
st = fast_step._state(model)
cam = cams[5]
import ctypes as C
from das3r_amd.render import das3r_render
with torch.no_grad():
    pkg = das3r_render(cam, model, pipe, bg, camera_pose=model.get_RT(cam.uid), fused=True)
# lengths through the inspection helper: same inputs as the fused render
from das3r_amd.render import rasterizer_inputs
torch.cuda.synchronize()
_lib.profile_report(); _lib.profile_enable(True)
loss, ps, pkg = train_step(model, cam, opt, 41, pipe, bg, fused=True)
torch.cuda.synchronize(); _lib.profile_enable(False)
k = _lib.profile_report()
print({n: round(v[1], 4) for n, v in k.items() if n.startswith("render_")})
# tile lengths: count instances per tile from radii / positions is not exposed; use the forward's saved buffers via a plain forward of the same tensors
rs = fast_step._settings(st, cam, model, bg)
P = st.P
lib = _lib.load(); _p = lambda t: C.c_void_p(t.data_ptr()); s = fast_step._stream(dev)
means3D, rotations = torch.empty_like(model._xyz), torch.empty_like(model._rotation)
scales, opac = torch.empty_like(model._scaling), torch.empty(P, 1, device=dev)
_lib.check(lib.das3r_pose_matrices_qt(_p(model.Q[cam.uid]), _p(model.T[cam.uid]), _p(st.mats), s), "pm")
_lib.check(lib.das3r_pretransform_forward(P, _p(model._xyz), _p(model._rotation), _p(model._scaling), _p(model._opacity), _p(model._conf_static.view(-1)), st.mask_ptr,
                                          _p(st.mats), C.c_void_p(st.mats.data_ptr() + 36), C.c_void_p(st.mats.data_ptr() + 48), _p(means3D), _p(rotations), _p(scales), _p(opac), s), "pf")
e = st.e
I, image, radii, geom, binning, img, cap = _forward_full(rs, means3D, model._features_dc.detach(), e, opac, scales, rotations, e, exact=True)
L = _lib.layout(P, I, 512, 208)
tiles = 32 * 13
rg = img[L["ranges"]:L["ranges"] + 8 * tiles].view(torch.int32).reshape(tiles, 2).cpu().numpy()
ln = (rg[:, 1] - rg[:, 0]).astype(np.int64)
print(kind, "I", I, "tiles", tiles, "mean", ln.mean().round(0), "max", ln.max(), "max/mean", round(ln.max() / ln.mean(), 2), "p90/mean", round(np.percentile(ln, 90) / ln.mean(), 2),
      "std/mean", round(ln.std() / ln.mean(), 2))
print("rows of tile lengths (k):", [int(ln.reshape(13, 32)[r].mean() / 1000) for r in range(13)])
