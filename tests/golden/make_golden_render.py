#!/usr/bin/env python3
"""Golden vectors from the reference's render() and covariance helpers, run on the CPU.

Run ONLY in the build container (needs /root/reference); the resulting ``ref_render.npz`` is committed and is the only
thing that travels.  No reference source is copied: the reference's own functions are imported and executed on seeded
inputs; the file holds their inputs and outputs.

The code in question hard-codes device="cuda" (utils/general_utils.py:65,83,102; gaussian_renderer/__init__.py:39-45,57,87),
so for the duration of the import/run `torch.zeros/ones/eye/zeros_like/ones_like/empty/tensor` drop a device="cuda"
argument and `Tensor.cuda()` is the identity — the arithmetic is untouched.

What each vector pins (SURVEY.md §8c "what can be imported here", VERDICT r1 "pin more of the oracle"):
  cov_*     utils/general_utils.py:62-110 build_rotation / build_scaling_rotation / strip_symmetric and
            scene/gaussian_model.py:32-36 build_covariance_from_scaling_rotation -> the Sigma = (R S)(R S)^T 6-vector
            (Appendix A.3: the kernel's formula for unit quaternions; the reference normalises, the kernel does not)
  render_*  gaussian_renderer/__init__.py:23-149 render(): the GaussianRasterizationSettings and the eight tensor arguments it
            hands to GaussianRasterizer (recorded by a stand-in rasterizer), for seeded models in three pipe modes
            (default / compute_cov3D_python / convert_SHs_python) -> the whole pre-transform of §8 a1; cases 4..6 the same for the
            three sibling renderers render_test (:152), render_no_soft (:279), render_confidence (:410)
  r3dgs*    gaussian_renderer/__init__3dgs.py:18-99 render(): the vanilla 3DGS renderer (world-space Gaussians, camera in the settings)
"""
import math
import os
import sys
import types
from types import SimpleNamespace
from unittest.mock import MagicMock

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_render.npz")
FIELDS = ("image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix", "projmatrix", "sh_degree",
          "campos", "prefiltered", "debug")


class _NoCuda:
    """device="cuda" -> default device, Tensor.cuda() -> self, for code that cannot otherwise run without a GPU."""
    NAMES = ("zeros", "ones", "eye", "zeros_like", "ones_like", "empty", "tensor", "rand", "randn", "full")

    def __enter__(self):
        self.saved = {n: getattr(torch, n) for n in self.NAMES}
        for n, f in self.saved.items():
            setattr(torch, n, self._wrap(f))
        self.cuda = torch.Tensor.cuda
        torch.Tensor.cuda = lambda self_, *a, **k: self_
        return self

    @staticmethod
    def _wrap(f):
        def g(*a, **k):
            if str(k.get("device", "")).startswith("cuda"):
                k.pop("device")
            return f(*a, **k)
        return g

    def __exit__(self, *exc):
        for n, f in self.saved.items():
            setattr(torch, n, f)
        torch.Tensor.cuda = self.cuda


def _stub(names):
    for n in names:
        parts = n.split(".")
        for i in range(1, len(parts) + 1):
            k = ".".join(parts[:i])
            if k not in sys.modules:
                sys.modules[k] = MagicMock()


class _Recorder:
    """Stand-in for diff_gaussian_rasterization: keeps what render() hands over."""
    calls = []

    class Settings(SimpleNamespace):
        pass

    class Rasterizer:
        def __init__(self, raster_settings):
            self.rs = raster_settings

        def __call__(self, **kw):
            _Recorder.calls.append((self.rs, kw))
            P = kw["means3D"].shape[0]
            return torch.zeros(3, int(self.rs.image_height), int(self.rs.image_width)), torch.ones(P, dtype=torch.int32)


def main():
    assert os.path.isdir(REF), "reference checkout not present: run in the build container"
    _stub(["open3d", "plyfile", "simple_knn", "simple_knn._C", "evo", "evo.core", "evo.core.trajectory", "evo.tools", "evo.core.metrics",
           "evo.tools.plot", "evo.core.geometry", "evo.main_ape", "evo.main_rpe", "evo.tools.file_interface", "cv2", "matplotlib",
           "matplotlib.pyplot", "roma", "imageio", "icecream", "torchvision", "torchvision.utils"])
    dgr = types.ModuleType("diff_gaussian_rasterization")
    dgr.GaussianRasterizationSettings = lambda **kw: _Recorder.Settings(**kw)
    dgr.GaussianRasterizer = lambda raster_settings: _Recorder.Rasterizer(raster_settings)
    sys.modules["diff_gaussian_rasterization"] = dgr
    sys.path.insert(0, REF)
    out = {}
    g = torch.Generator().manual_seed(20250927)
    with _NoCuda():
        from utils.general_utils import build_rotation, build_scaling_rotation, strip_symmetric
        from utils.graphics_utils import getProjectionMatrix
        from scene.gaussian_model import GaussianModel
        from gaussian_renderer import render, render_confidence, render_no_soft, render_test

        # ---- (1) covariance helpers
        N = 96
        q = torch.randn(N, 4, generator=g) * (0.5 + torch.rand(N, 1, generator=g))      # NOT unit: the helpers normalise
        s = torch.exp(torch.randn(N, 3, generator=g) * 0.7 - 2.0)
        out["cov_q"], out["cov_s"] = q.numpy(), s.numpy()
        out["cov_R"] = build_rotation(q).numpy()
        for mod in (1.0, 1.7):
            L = build_scaling_rotation(mod * s, q)
            out[f"cov_L_mod{mod}"] = L.numpy()
            out[f"cov_sym_mod{mod}"] = strip_symmetric(L @ L.transpose(1, 2)).numpy()
        gm = GaussianModel(3)
        out["cov_activation_mod1.7"] = gm.covariance_activation(s, 1.7, q).numpy()

        # ---- (2) render(): what reaches the rasterizer
        frames, H, W = 3, 6, 8
        P = frames * H * W - 17          # aggregated_mask drops 17 pixels
        # cases 4..6: the sibling renderers (render_test :152, render_no_soft :279, render_confidence :410), appended so that the
        # seeded stream of cases 0..3 is unchanged
        fns = {"render": render, "test": render_test, "no_soft": render_no_soft, "confidence": render_confidence}
        for case, (deg, cov_py, sh_py, mod, variant) in enumerate([(0, False, False, 1.0, "render"), (2, True, False, 1.3, "render"),
                                                                   (3, False, True, 1.0, "render"), (1, False, False, 0.8, "render"),
                                                                   (1, False, False, 1.0, "test"), (2, False, False, 1.2, "no_soft"),
                                                                   (0, True, False, 1.0, "confidence")]):
            pc = GaussianModel(3)
            pc.active_sh_degree = deg
            pc._xyz = torch.randn(P, 3, generator=g) * 2.0 + torch.tensor([0.0, 0.0, 5.0])
            pc._rotation = torch.randn(P, 4, generator=g) * (0.8 + 0.4 * torch.rand(P, 1, generator=g))
            pc._scaling = torch.randn(P, 3, generator=g) * 0.5 - 2.5
            pc._opacity = torch.randn(P, 1, generator=g)
            pc._features_dc = torch.randn(P, 1, 3, generator=g)
            pc._features_rest = torch.randn(P, 15, 3, generator=g) * 0.2
            pc._conf_static = torch.rand(frames, H, W, generator=g)
            mask = torch.ones(frames * H * W, dtype=torch.bool)
            mask[torch.randperm(frames * H * W, generator=g)[:17]] = False
            pc.aggregated_mask = mask
            pose = torch.randn(7, generator=g)
            pose[:4] = pose[:4] / pose[:4].norm() * 1.05                                  # near-unit, not unit
            fovx, fovy = 1.1 + 0.1 * case, 0.8
            cam = SimpleNamespace(FoVx=fovx, FoVy=fovy, image_height=40 + case, image_width=56,
                                  projection_matrix=getProjectionMatrix(0.01, 100.0, fovx, fovy).transpose(0, 1),
                                  camera_center=torch.randn(3, generator=g))
            pipe = SimpleNamespace(debug=False, compute_cov3D_python=cov_py, convert_SHs_python=sh_py)
            bg = torch.rand(3, generator=g)
            if variant == "test":          # render_test multiplies by conf_static as stored: one value per Gaussian
                pc._conf_static = torch.rand(P, 1, generator=g)
            if variant == "no_soft":       # field of view and projection from the model
                pc.FoVx, pc.FoVy = torch.tensor(fovx * 1.07), torch.tensor(fovy * 0.93)
                cam.get_projection_matrix = lambda fx, fy: getProjectionMatrix(0.01, 100.0, fx, fy).transpose(0, 1)
            if variant == "confidence":
                pc._conf = torch.rand(P, generator=g)
            _Recorder.calls.clear()
            pkg = fns[variant](cam, pc, pipe, bg, scaling_modifier=mod, camera_pose=pose)
            assert len(_Recorder.calls) == 1
            assert torch.is_tensor(pkg) if variant == "confidence" else sorted(pkg) == ["radii", "render", "viewspace_points", "visibility_filter"]
            out[f"render{case}_variant"] = np.array(variant)
            if variant == "no_soft":
                out[f"render{case}_pc_fov"] = np.array([float(pc.FoVx), float(pc.FoVy)])
            if variant == "confidence":
                out[f"render{case}_pc_conf"] = pc._conf.numpy()
            rs, kw = _Recorder.calls[0]
            pre = f"render{case}_"
            for name in ("_xyz", "_rotation", "_scaling", "_opacity", "_features_dc", "_features_rest", "_conf_static"):
                out[pre + "pc" + name] = getattr(pc, name).detach().numpy()
            out[pre + "pc_mask"] = mask.numpy()
            out[pre + "pose"], out[pre + "bg"] = pose.numpy(), bg.numpy()
            out[pre + "cam"] = np.array([fovx, fovy, cam.image_height, cam.image_width], dtype=np.float64)
            out[pre + "cam_center"] = cam.camera_center.numpy()
            out[pre + "cam_proj"] = cam.projection_matrix.numpy()
            out[pre + "mode"] = np.array([deg, int(cov_py), int(sh_py)], dtype=np.int64)
            out[pre + "mod"] = np.array(mod)
            for f in FIELDS:
                v = getattr(rs, f)
                out[pre + "rs_" + f] = v.detach().numpy() if torch.is_tensor(v) else np.array(v)
            for k, v in kw.items():
                out[pre + "kw_" + k] = v.detach().numpy() if v is not None else np.zeros(0, dtype=np.float32)
            out[pre + "kw_none"] = np.array([k for k, v in kw.items() if v is None])
    # ---- (3) the vanilla 3DGS renderer kept beside render(): gaussian_renderer/__init__3dgs.py:18-99 (world-space Gaussians, camera
    #          in the settings); appended after everything else so that the seeded stream above is unchanged
    with _NoCuda():
        import importlib
        render_3dgs = importlib.import_module("gaussian_renderer.__init__3dgs").render
        from scene.gaussian_model import GaussianModel
        from utils.graphics_utils import getProjectionMatrix, getWorld2View2
        for case, (deg, cov_py, sh_py, mod) in enumerate([(3, False, False, 1.0), (1, True, True, 1.4)]):
            P = 90
            pc = GaussianModel(3)
            pc.active_sh_degree = deg
            pc._xyz = torch.randn(P, 3, generator=g) * 1.5
            pc._rotation = torch.randn(P, 4, generator=g) * (0.6 + 0.8 * torch.rand(P, 1, generator=g))
            pc._scaling = torch.randn(P, 3, generator=g) * 0.5 - 2.5
            pc._opacity = torch.randn(P, 1, generator=g)
            pc._features_dc = torch.randn(P, 1, 3, generator=g)
            pc._features_rest = torch.randn(P, 15, 3, generator=g) * 0.2
            Rm = torch.linalg.qr(torch.randn(3, 3, generator=g))[0].numpy()
            tv = (torch.randn(3, generator=g) + torch.tensor([0.0, 0.0, 6.0])).numpy()
            fovx, fovy = 0.9 + 0.2 * case, 0.7
            wvt = torch.tensor(getWorld2View2(Rm, tv)).transpose(0, 1)
            proj = getProjectionMatrix(0.01, 100.0, fovx, fovy).transpose(0, 1)
            full = (wvt.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0)
            cam = SimpleNamespace(FoVx=fovx, FoVy=fovy, image_height=36, image_width=48 + case, world_view_transform=wvt,
                                  full_proj_transform=full, camera_center=wvt.inverse()[3, :3])
            pipe = SimpleNamespace(debug=False, compute_cov3D_python=cov_py, convert_SHs_python=sh_py)
            bg = torch.rand(3, generator=g)
            _Recorder.calls.clear()
            pkg = render_3dgs(cam, pc, pipe, bg, scaling_modifier=mod)
            assert sorted(pkg) == ["radii", "render", "viewspace_points", "visibility_filter"] and len(_Recorder.calls) == 1
            rs, kw = _Recorder.calls[0]
            pre = f"r3dgs{case}_"
            for name in ("_xyz", "_rotation", "_scaling", "_opacity", "_features_dc", "_features_rest"):
                out[pre + "pc" + name] = getattr(pc, name).detach().numpy()
            out[pre + "bg"] = bg.numpy()
            out[pre + "cam"] = np.array([fovx, fovy, cam.image_height, cam.image_width], dtype=np.float64)
            out[pre + "cam_wvt"], out[pre + "cam_full"], out[pre + "cam_center"] = wvt.numpy(), full.numpy(), cam.camera_center.numpy()
            out[pre + "cam_R"], out[pre + "cam_t"] = Rm, tv
            out[pre + "mode"] = np.array([deg, int(cov_py), int(sh_py)], dtype=np.int64)
            out[pre + "mod"] = np.array(mod)
            for f in FIELDS:
                v = getattr(rs, f)
                out[pre + "rs_" + f] = v.detach().numpy() if torch.is_tensor(v) else np.array(v)
            for k, v in kw.items():
                out[pre + "kw_" + k] = v.detach().numpy() if v is not None else np.zeros(0, dtype=np.float32)
            out[pre + "kw_none"] = np.array([k for k, v in kw.items() if v is None])
    np.savez_compressed(OUT, **out)
    print(f"wrote {OUT}: {len(out)} arrays, {os.path.getsize(OUT)} bytes")


if __name__ == "__main__":
    main()
