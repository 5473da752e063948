"""One line at the top of an unmodified DAS3R entry point —

    import das3r_amd.integrate; das3r_amd.integrate.patch()

— and its training iteration (/root/reference/train_gui.py:542-589, train_test_psnr.py) runs on the opt-in fused kernels of SURVEY.md
section 8(f) wherever their preconditions hold, without another edit (the drop-in modules alone give an unmodified
checkout the HIP rasterizer inside the reference's PyTorch glue — 13.5 ms per iteration at the Sintel shape — while the 1.2 ms iteration
needed edits to render() and the optimizer).  What is swapped, and when:

  gaussian_renderer.render   -> das3r_amd.render.das3r_render(..., fused=True): pose -> camera frame, quaternion product, exp / sigmoid
      x conf_static as ONE HIP kernel each way (gaussian_renderer/__init__.py:83-97,107), the SH tensor handed over as its active prefix.
      Preconditions (else the call goes to the original function untouched): the default pipeline flags (no compute_cov3D_python, no
      convert_SHs_python), no override_color, no filtering, use_conf, a camera pose, every parameter a dense fp32 tensor on a HIP device.
      Every module that had already done `from gaussian_renderer import render` is re-bound too.
  GaussianModel.training_setup -> the original, then `optimizer` / `optimizer_cam` re-built as das3r_amd.fused.FusedAdam over the SAME
      param_groups (names, learning rates, eps, betas; scene/gaussian_model.py:236-261): one HIP launch per step() instead of ~10 torch
      kernels per group, SH coefficients swept up to the active degree only; step() / zero_grad() / param_groups / state keep the
      surface the reference's loop, its update_learning_rate and its capture() use.  Precondition: dense fp32 device parameters.
  GaussianModel.oneupSHdegree -> also tells the fused optimizer the new active degree.

  utils.loss_utils.ssim        -> das3r_amd.fused.ssim_map behind the same signature (round 6): the SSIM map of two [3, H, W] fp32 device images
      with the 11 x 11 window, and its backward, as two HIP launches each way — the reference's twelve depthwise convolutions and their
      elementwise chains were 0.75 ms of MIOpen kernels of the 2.7 ms patched iteration.  Anything else (another window, a batch, CPU
      tensors) goes to the original.  Modules that had already imported the name are re-bound.

What is NOT swapped, because it is code inside train_gui.py's loop and not a function: how the loop combines its L1 and SSIM maps into the
loss (a few elementwise torch kernels) and the `if psnr_frame > threshold` gate (one host sync per iteration).  The direct iteration
without either is das3r_amd.fast_step.  `unpatch()` restores everything."""
import sys
import types

import torch

_saved = {}


def _dense_device_f32(*tensors):
    return all(torch.is_tensor(t) and t.device.type == "cuda" and t.dtype == torch.float32 and t.is_contiguous() for t in tensors)


def _fused_optimizer(optim, sh_rest_names=("f_rest",)):
    """torch.optim.Adam -> FusedAdam over the same groups; None when a precondition fails (the caller keeps the original)."""
    from .fused import FusedAdam
    if not isinstance(optim, torch.optim.Adam):
        return None
    groups = []
    for g in optim.param_groups:
        if not _dense_device_f32(*g["params"]) or g.get("amsgrad") or g.get("weight_decay", 0) or g.get("maximize"):
            return None
        ng = {k: v for k, v in g.items() if k in ("params", "lr", "name")}
        if g.get("name") in sh_rest_names and all(p.dim() == 3 for p in g["params"]):
            ng["sh_rest"] = True
        groups.append(ng)
    first = optim.param_groups[0]
    return FusedAdam(groups, lr=optim.defaults.get("lr", 0.0), betas=tuple(first.get("betas", (0.9, 0.999))), eps=first.get("eps", 1e-15))


def make_render(original):
    """The replacement for gaussian_renderer.render (same signature: gaussian_renderer/__init__.py:23-24)."""
    from .render import das3r_render

    def render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None, camera_pose=None, filtering=None,
               use_conf=True, **kw):
        ok = (not kw and override_color is None and filtering is None and use_conf and camera_pose is not None
              and not getattr(pipe, "compute_cov3D_python", False) and not getattr(pipe, "convert_SHs_python", False)
              and _dense_device_f32(pc._xyz, pc._rotation, pc._scaling, pc._opacity, pc._conf_static, pc._features_dc, pc._features_rest)
              and torch.is_tensor(getattr(pc, "aggregated_mask", None)))
        if not ok:
            return original(viewpoint_camera, pc, pipe, bg_color, scaling_modifier, override_color, camera_pose, filtering, use_conf, **kw) \
                if (kw or not use_conf or filtering is not None) else original(viewpoint_camera, pc, pipe, bg_color, scaling_modifier, override_color, camera_pose)
        return das3r_render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier, camera_pose=camera_pose, fused=True)

    render._das3r_original = original
    return render


def make_ssim(original):
    """The replacement for utils.loss_utils.ssim (same signature: utils/loss_utils.py:39)."""
    from .fused import ssim_map

    def ssim(img1, img2, window_size=11, size_average=True):
        ok = (window_size == 11 and torch.is_tensor(img1) and torch.is_tensor(img2) and img1.dim() == 3 and img1.shape[0] == 3 and img1.shape == img2.shape
              and img1.device.type == "cuda" and img2.device == img1.device and img1.dtype == torch.float32 and img2.dtype == torch.float32)
        if not ok:
            return original(img1, img2, window_size, size_average)
        m = ssim_map(img1, img2)
        return m.mean() if size_average else m   # (size_average = False: the reference returns the map, train_gui.py:568)

    ssim._das3r_original = original
    return ssim


def patch_model_class(cls):
    """GaussianModel (or anything with its training_setup / oneupSHdegree): fused optimizers behind the same attributes."""
    if getattr(cls.training_setup, "_das3r_original", None) is not None:
        return cls
    orig_setup, orig_oneup = cls.training_setup, cls.oneupSHdegree

    def training_setup(self, *a, **k):
        out = orig_setup(self, *a, **k)
        fused = _fused_optimizer(self.optimizer)
        if fused is not None:
            self.optimizer = fused
            fused.set_active_sh_degree(self.active_sh_degree)
        cam = getattr(self, "optimizer_cam", None)
        if cam is not None:
            # the two inert field-of-view groups (0-d tensors that never get a gradient: SURVEY.md C6) ride along: FusedAdam passes a
            # parameter without gradient by, exactly like torch.optim.Adam
            fused_cam = _fused_optimizer(cam, sh_rest_names=())
            if fused_cam is not None:
                self.optimizer_cam = fused_cam
        return out

    def oneupSHdegree(self, *a, **k):
        out = orig_oneup(self, *a, **k)
        if hasattr(getattr(self, "optimizer", None), "set_active_sh_degree"):
            self.optimizer.set_active_sh_degree(self.active_sh_degree)
        return out

    training_setup._das3r_original, oneupSHdegree._das3r_original = orig_setup, orig_oneup
    cls.training_setup, cls.oneupSHdegree = training_setup, oneupSHdegree
    _saved.setdefault("classes", []).append(cls)
    return cls


def patch(renderer_module=None, model_class=None, loss_module=None):
    """Default: the reference's own modules (`gaussian_renderer`, `scene.gaussian_model.GaussianModel`, `utils.loss_utils`), imported here.
    Tests pass stand-ins.  Idempotent.  -> dict of what was patched."""
    done = {}
    if loss_module is None:
        import importlib
        try:
            loss_module = importlib.import_module("utils.loss_utils")
        except ImportError:
            loss_module = None   # (a checkout without it keeps its loss)
    if loss_module is not None and getattr(loss_module.ssim, "_das3r_original", None) is None:
        original_ssim = loss_module.ssim
        new_ssim = make_ssim(original_ssim)
        loss_module.ssim = new_ssim
        _saved.setdefault("ssims", []).append((loss_module, original_ssim))
        for mod in list(sys.modules.values()):   # `from utils.loss_utils import ssim` happened before us
            if isinstance(mod, types.ModuleType) and mod is not loss_module and getattr(mod, "ssim", None) is original_ssim:
                mod.ssim = new_ssim
                _saved.setdefault("ssims", []).append((mod, original_ssim))
        done["ssim"] = loss_module.__name__
    if renderer_module is None:
        import importlib
        renderer_module = importlib.import_module("gaussian_renderer")
    original = getattr(renderer_module.render, "_das3r_original", None) or renderer_module.render
    if getattr(renderer_module.render, "_das3r_original", None) is None:
        new = make_render(original)
        renderer_module.render = new
        _saved.setdefault("renderers", []).append((renderer_module, original))
        for mod in list(sys.modules.values()):   # `from gaussian_renderer import render` happened before us
            if isinstance(mod, types.ModuleType) and mod is not renderer_module and getattr(mod, "render", None) is original:
                mod.render = new
                _saved.setdefault("rebound", []).append((mod, original))
        done["render"] = renderer_module.__name__
    if model_class is None:
        import importlib
        model_class = importlib.import_module("scene.gaussian_model").GaussianModel
    patch_model_class(model_class)
    done["model"] = model_class.__name__
    return done


def unpatch():
    for mod, original in _saved.pop("renderers", []) + _saved.pop("rebound", []):
        mod.render = original
    for mod, original in _saved.pop("ssims", []):
        mod.ssim = original
    for cls in _saved.pop("classes", []):
        cls.training_setup = cls.training_setup._das3r_original
        cls.oneupSHdegree = cls.oneupSHdegree._das3r_original
