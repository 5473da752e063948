"""Shared helpers for the parity tests: scene variants that hit every branch of SURVEY.md A.9, oracle runners and the
stated tolerances."""
import json
import math
import os

import numpy as np
import torch

from das3r_amd.camera import projection_matrix
from das3r_amd.synth import Scene, make_scene

# ---- stated fp32 tolerances (north_star: "within a stated fp32 tolerance") ----
COLOR_TOL = 1e-4          # max |delta| on colours in [0,1] ...
FLIP_FRACTION = 1e-3      # ... except at most this fraction of pixels, explained by a threshold flip
FLIP_MAX = 2.5e-2         # (alpha<1/255 or T<1e-4 decided differently by a 1-ulp exp difference), each bounded by this
# Gradients, max |delta| relative to the tensor's max |grad|.  Round 6: the bars are set to what the kernels
# measure (DAS3R_TOL_REPORT, tools/tol_report.py; profiles/r06_tol_report.txt), not to the ceiling SURVEY.md section 8(a) gave a
# kernel with float atomics (1e-3): the shipped kernels have none, and measure <= 1.2e-5 on every fixture and 1.5e-6 at C4.
# A biased kernel — exp scaled by 1 + 1e-4 — fails them (tests/test_gpu_fullsize.py::test_mutated_backward_is_caught).
GRAD_REL_TOL = 5e-5       # default kernel selection (dpp / blk backward, every forward kernel)
GRAD_REL_TOL_SPLIT = 1e-4  # the superseded split-bf16 kernels (scan, mfma: 16 mantissa bits per factor, measured 2.4e-5) and stream
# Threshold flips.  A pair whose alpha sits within an ulp of 1/255 (or a pixel whose T reaches 1e-4 within an ulp) is decided one way
# by v_exp_f32 and the other by the oracle's libm expf.  On the small fixtures that happens to no element; at full size (c2: 100 k
# splats over 2 M pixels, ds: 5 M splats) a handful of ELEMENTS carry such a pair with a weight that shows: measured 6.6e-4 of the
# tensor maximum on one element of c2, 6.4e-4 on ds, none at C4 (1.5e-6).  Tests at those sizes name the allowance explicitly
# (`flips=`): at most max(GRAD_FLIP_MIN, GRAD_FLIP_FRACTION * elements) elements may exceed the bar, none by more than GRAD_FLIP_MAX.
GRAD_FLIP_FRACTION = 5e-6   # (measured: 9 of the 4.8 M elements of c2's dL/dshs — one splat's row —, 3 of ds's 20 M)
GRAD_FLIP_MIN = 3
GRAD_FLIP_MAX = 2e-3      # (the old max-norm bar)


def look_at_view(eye, target, up=(0.0, 1.0, 0.0)):
    """World->camera 4x4 (column convention) for a camera at `eye` looking at `target`, +z forward, +y down-ish."""
    eye, target, up = (np.asarray(v, dtype=np.float64) for v in (eye, target, up))
    f = target - eye
    f /= np.linalg.norm(f)
    r = np.cross(up, f)
    r /= np.linalg.norm(r)
    u = np.cross(f, r)
    R = np.stack([r, u, f], 0)
    M = np.eye(4)
    M[:3, :3] = R
    M[:3, 3] = -R @ eye
    return torch.tensor(M, dtype=torch.float32)


def scene_variant(name):
    """Returns (Scene, mode dict).  mode: colors_precomp / cov3D_precomp flags, scale_modifier."""
    mode = dict(colors_precomp=False, cov3D_precomp=False, scale_modifier=1.0)
    if name == "basic_deg3":
        sc = make_scene(P=1500, W=128, H=80, focal=100.0, sh_degree=3, seed=21, bg=(0.2, 0.1, 0.3))
    elif name == "deg0":
        sc = make_scene(P=1200, W=96, H=96, focal=90.0, sh_degree=0, seed=22)
    elif name == "deg1":
        sc = make_scene(P=900, W=112, H=64, focal=80.0, sh_degree=1, seed=23, bg=(1.0, 1.0, 1.0))
    elif name == "deg2":
        sc = make_scene(P=900, W=112, H=64, focal=80.0, sh_degree=2, seed=24)
    elif name == "ragged_image":   # W, H not multiples of 16 (1080 = 67.5 * 16 in the real config)
        sc = make_scene(P=1500, W=123, H=77, focal=95.0, sh_degree=3, seed=25, bg=(0.0, 0.5, 0.0))
    elif name == "colors_precomp":
        sc = make_scene(P=1000, W=96, H=64, focal=80.0, sh_degree=0, seed=26)
        mode["colors_precomp"] = True
    elif name == "cov3D_precomp":
        sc = make_scene(P=1000, W=96, H=64, focal=80.0, sh_degree=2, seed=27)
        mode["cov3D_precomp"] = True
    elif name == "scale_modifier":
        sc = make_scene(P=1000, W=96, H=64, focal=80.0, sh_degree=1, seed=28)
        mode["scale_modifier"] = 1.7
    elif name == "long_lists":     # big opaque-ish splats: > 256 instances per tile (multi-batch), T<1e-4 early stop, alpha clamp
        sc = make_scene(P=2500, W=64, H=48, focal=60.0, sh_degree=1, seed=29, s_px=(3.0, 12.0))
        sc.opacities[: sc.P // 2] = 0.999
    elif name == "deep":           # faint splats: hundreds of blended layers per pixel (n_contrib > 256: multi-batch backward)
        sc = make_scene(P=6000, W=48, H=32, focal=40.0, sh_degree=0, seed=34, s_px=(3.0, 8.0), opacity=0.02)
    elif name == "world_camera":   # vanilla-3DGS style: non-identity viewmatrix + campos (gaussian_renderer/__init__3dgs.py)
        sc = make_scene(P=1500, W=128, H=80, focal=100.0, sh_degree=3, seed=30)
        eye = (0.7, -0.4, -1.5)
        view_col = look_at_view(eye, (0.2, 0.1, 5.0))
        fovx, fovy = 2 * math.atan(sc.tanfovx), 2 * math.atan(sc.tanfovy)
        view = view_col.t().contiguous()                     # row-vector layout
        proj = (view_col.t() @ projection_matrix(0.01, 100.0, fovx, fovy).t()).contiguous()
        sc = Scene(**{**sc.__dict__, "viewmatrix": view, "projmatrix": proj, "campos": torch.tensor(eye, dtype=torch.float32)})
    elif name == "culled":         # near-plane culls, behind-camera, off-screen rects, clamped EWA coordinates
        sc = make_scene(P=1500, W=96, H=64, focal=80.0, sh_degree=1, seed=31)
        sc.means3D[:200, 2] = torch.linspace(-1.0, 0.002, 200)          # behind / at the near plane
        sc.means3D[200:400, 0] *= 4.0                                   # far off-screen -> empty rect or EWA clamp
        sc.means3D[400:420, 2] = 0.0011                                 # just past the near plane, huge footprint
    elif name == "depth_ties":     # duplicated splats: identical depth keys -> order by index
        sc = make_scene(P=600, W=64, H=48, focal=60.0, sh_degree=0, seed=32, s_px=(1.0, 5.0))
        for fld in ("means3D", "scales", "rotations"):
            getattr(sc, fld)[300:] = getattr(sc, fld)[:300]
    elif name == "single":
        sc = make_scene(P=1, W=48, H=32, focal=40.0, sh_degree=3, seed=33, s_px=(4.0, 4.0))
        sc.means3D[0] = torch.tensor([0.05, -0.02, 2.0])
    else:
        raise KeyError(name)
    return sc, mode


VARIANTS = ["basic_deg3", "deg0", "deg1", "deg2", "ragged_image", "colors_precomp", "cov3D_precomp", "scale_modifier",
            "long_lists", "deep", "world_camera", "culled", "depth_ties", "single"]


def cov3d_of(sc, scale_modifier=1.0):
    """Sigma = R diag((mod*s)^2) R^T as 6-vector, float64 math -> float32 (independent of both implementations:
    the formula of /root/reference/scene/gaussian_model.py:32-36 without the quaternion normalisation)."""
    q = sc.rotations.double()
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                     2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                     2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).reshape(-1, 3, 3)
    Mm = R * (scale_modifier * sc.scales.double())[:, None, :]
    S = Mm @ Mm.transpose(1, 2)
    return torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], 1).float().contiguous()


def raster_inputs(sc, mode):
    """dict of the tensor kwargs GaussianRasterizer.forward takes (CPU tensors)."""
    kw = dict(means3D=sc.means3D, opacities=sc.opacities)
    if mode["colors_precomp"]:
        kw["colors_precomp"] = torch.sigmoid(sc.shs[:, 0, :] * 1.3).contiguous()
    else:
        kw["shs"] = sc.shs
    if mode["cov3D_precomp"]:
        kw["cov3D_precomp"] = cov3d_of(sc, 1.0)
    else:
        kw["scales"], kw["rotations"] = sc.scales, sc.rotations
    return kw


def settings_kwargs(sc, mode):
    kw = sc.settings_kwargs()
    kw["scale_modifier"] = mode["scale_modifier"]
    return kw


def run_oracle(sc, mode, backward=True):
    from oracle import c_oracle
    o = c_oracle.RasterOracle(**settings_kwargs(sc, mode))
    kw = {k: v.numpy() for k, v in raster_inputs(sc, mode).items()}
    color, radii = o.forward(kw.pop("means3D"), kw.pop("opacities"), **kw)
    grads = o.backward(sc.dL_dpix.numpy()) if backward else None
    saved = o.saved()
    o.free()
    return color, radii, grads, saved


def assert_color_close(a, b, what=""):
    d = np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))
    frac = float((d > COLOR_TOL).mean())
    assert frac <= FLIP_FRACTION and d.max() <= FLIP_MAX, f"{what}: {frac:.2e} of values differ by > {COLOR_TOL}, max {d.max():.3e}"


def _report(kind, what, value, tol):
    """DAS3R_TOL_REPORT=<file>: every measured error beside its bar (how far under the bar the kernels are: tools/tol_report.py)."""
    path = os.environ.get("DAS3R_TOL_REPORT")
    if path:
        with open(path, "a") as f:
            f.write(json.dumps({"kind": kind, "what": what, "value": float(value), "tol": float(tol),
                                "test": os.environ.get("PYTEST_CURRENT_TEST", "")}) + "\n")


def assert_grad_close(a, b, what="", tol=None, flips=False):
    """Max-norm gradient check.  flips=True (full-size scenes only): a few elements may sit on a threshold flip — see GRAD_FLIP_*."""
    tol = GRAD_REL_TOL if tol is None else tol
    a, b = np.asarray(a, np.float64).reshape(-1), np.asarray(b, np.float64).reshape(-1)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    assert np.isfinite(a).all(), f"{what}: non-finite gradient"
    scale = np.abs(b).max()
    if scale == 0:
        assert np.abs(a).max() == 0, f"{what}: expected all-zero gradient"
        return
    d = np.abs(a - b) / scale
    rel = d.max()
    over = int((d > tol).sum())
    _report("max_norm", what, rel, tol)
    _report("over_bar", what, over, a.size)
    if flips:
        allowed = max(GRAD_FLIP_MIN, int(GRAD_FLIP_FRACTION * a.size))
        assert over <= allowed and rel <= GRAD_FLIP_MAX, (f"{what}: {over} elements (allowed: {allowed} threshold flips) beyond {tol} of max|ref|, "
                                                           f"the largest at {rel:.3e} (flip bound {GRAD_FLIP_MAX})")
        return
    assert rel <= tol, f"{what}: max |delta| / max|ref| = {rel:.3e} > {tol}"


def tolerances_for(kind):
    """Named bars per backward kernel selection (DAS3R_RENDER_BWD value or None): the shipped kernels get the tight ones, the superseded
    split-bf16 / stream kernels (selectable for A-B runs, experiments builds) their own."""
    split = kind is not None and (kind.startswith("scan") or kind.startswith("mfma") or kind.startswith("stream"))
    if split:
        return dict(tol=GRAD_REL_TOL_SPLIT, rtol=GRAD_ELEM_RTOL_SPLIT, floor=GRAD_ELEM_FLOOR_SPLIT, outliers=GRAD_ELEM_OUTLIERS_SPLIT)
    return dict(tol=GRAD_REL_TOL, rtol=GRAD_ELEM_RTOL, floor=GRAD_ELEM_FLOOR, outliers=GRAD_ELEM_OUTLIERS)


GRAD_ELEM_RTOL = 1e-4     # element-wise: |delta| <= 1e-4 |ref| + floor * max|ref| ...
GRAD_ELEM_FLOOR = 2e-6    # ... the floor covers fp32 summation-order noise on elements that are sums of cancelling terms
GRAD_ELEM_OUTLIERS = 1e-4  # ... for all but this fraction of the elements (threshold flips: a pair with alpha within 1 ulp of 1/255);
                           # measured with these bars: <= 3.1e-5 of the elements at full size, 0 at C4
GRAD_ELEM_RTOL_SPLIT, GRAD_ELEM_FLOOR_SPLIT, GRAD_ELEM_OUTLIERS_SPLIT = 1e-3, 2e-5, 1e-3   # (round 1 - 5's bars, for the split-bf16 kernels)


def assert_grad_elementwise(a, b, what="", rtol=None, floor=None, outliers=None):
    """Element-wise gradient check (the max-norm check leaves splats with small gradients unchecked)."""
    rtol = GRAD_ELEM_RTOL if rtol is None else rtol
    floor = GRAD_ELEM_FLOOR if floor is None else floor
    outliers = GRAD_ELEM_OUTLIERS if outliers is None else outliers
    a, b = np.asarray(a, np.float64).reshape(-1), np.asarray(b, np.float64).reshape(-1)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    scale = np.abs(b).max()
    if scale == 0:
        assert np.abs(a).max() == 0, f"{what}: expected all-zero gradient"
        return
    bad = np.abs(a - b) > rtol * np.abs(b) + floor * scale
    frac = float(bad.mean())
    if os.environ.get("DAS3R_TOL_REPORT"):   # the fraction that would fail bars ten and a hundred times tighter
        for div in (1.0, 10.0, 100.0):
            _report(f"elementwise/{div:g}", what, float((np.abs(a - b) > (rtol * np.abs(b) + floor * scale) / div).mean()), outliers)
    assert frac <= outliers, f"{what}: {frac:.2e} of the elements differ by more than {rtol} |ref| + {floor} max|ref|"
    # an element that the reference has exactly zero (culled / untouched) must be exactly zero
    zero = b == 0
    if zero.any():
        assert float(np.abs(a[zero]).max()) <= floor * scale, f"{what}: non-zero gradient where the reference has none"
