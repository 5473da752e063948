"""The fused iteration without autograd (round 4): one pass of DAS3R's hot loop — render, masked L1 + SSIM
loss, backward, both Adam steps (/root/reference/train_gui.py:532-589) — as a straight sequence of calls into the C-ABI.

Round 3's fused iteration still went through torch.autograd and the reference's Python: 163 kernel launches per step of which 22
were this library's — boolean-mask indexing just to shape the dummy means2D (a nonzero + gather over 2 M splats), eye / bmm /
inverse for the settings of every render, cat / index for the pose and their index_put backward, select_backward + add over the
whole conf_static tensor for one frame's mask gradient, a dozen scalar kernels for loss / PSNR / the EMA, zero fills.  Here:

    pose row of Q, row of T  -> das3r_pose_matrices_qt -> das3r_pretransform_forward
    cached settings (identity view, projection, zero campos: built once per camera)
    das3r_raster_forward -> das3r_photometric_forward -> das3r_photometric_finish  {loss, mse, psnr_frame} on the device
    das3r_photometric_backward -> das3r_raster_backward -> das3r_pretransform_backward_adam -> das3r_pose_chain_qt
        (ABI 11: the chain rule through the pre-transform AND the Adam step of xyz / rotation / scaling / opacity in one pass — those
         four gradients never reach memory; model.fuse_geometry_adam = False keeps das3r_pretransform_backward + gradients)
    FusedAdam.step (one launch: f_dc, f_rest, conf_static), FusedAdam.step(gate = psnr_frame) for the poses

Gradients land where FusedAdam expects them (`p.grad`, or the compact SH gradient of fused._ShPrefix's contract); the pose gradient
is written into row `uid` of two dense, otherwise zero buffers (torch's index backward produces exactly that dense gradient, and
Adam's moments of the other rows decay with it).  Semantics are those of das3r_amd.train.train_step(fused=True) — the tests hold
both to the float64 trainer and to each other — and nothing here is reachable unless the caller opted into the fused kernels.
"""
import ctypes as C
import math
import os

import torch

from . import _lib
from .rasterizer import GaussianRasterizationSettings, _backward_impl, _forward_full, _on_device, _stream

_p = lambda t: C.c_void_p(t.data_ptr())


class _Pkg(dict):
    """render package; `visibility_filter` (radii > 0: one kernel over P) is computed when somebody asks for it."""

    def __missing__(self, k):
        if k == "visibility_filter":
            v = self["radii"] > 0
            self[k] = v
            return v
        raise KeyError(k)


def available(model, pipe):
    """The direct path covers the configuration the farm trains in: fused optimizers on both parameter sets, the default `pipe`."""
    if os.environ.get("DAS3R_FAST_STEP", "1") == "0":   # (A-B runs of whole jobs: tools/farm_davis_shape.py)
        return False
    return (getattr(model, "fast_step", True) and getattr(model.optimizer, "is_fused", False) and hasattr(model.optimizer_cam, "_gate_state")
            and model.optimizer.handles_compact_sh(model._features_rest)
            and not getattr(pipe, "compute_cov3D_python", False) and not getattr(pipe, "convert_SHs_python", False)
            and not getattr(pipe, "debug", False) and model._xyz.device.type == "cuda"
            # the C-ABI takes plain pointers: every parameter must be a dense fp32 tensor (anything else keeps the autograd form, which
            # goes through .contiguous())
            and all(t.is_contiguous() and t.dtype == torch.float32 for t in (model._xyz, model._rotation, model._scaling, model._opacity,
                                                                            model._features_dc, model._features_rest, model._conf_static, model.Q, model.T)))


class _State:
    """Per-model buffers that live across iterations (all zero at rest where zero matters)."""

    def __init__(self, model):
        dev = model._xyz.device
        P = model._xyz.shape[0]
        self.dev, self.P = dev, P
        self.mats = torch.empty(28, device=dev)
        self.g_small = torch.zeros(28, device=dev)            # re-armed by das3r_pose_chain_qt
        self.one = torch.ones(1, device=dev)
        self.Qg, self.Tg = torch.zeros_like(model.Q), torch.zeros_like(model.T)
        self.tQg, self.tTg = (torch.zeros_like(model.test_Q), torch.zeros_like(model.test_T)) if model.test_Q is not None else (None, None)
        self.means2D = torch.zeros(P, 3, device=dev, requires_grad=True)   # the reference's dummy leaf: only its .grad is ever used
        self.e = torch.empty(0, device=dev)
        idx = getattr(model, "_mask_index", None)
        if idx is None:
            idx = model._mask_index = torch.nonzero(model.aggregated_mask.reshape(-1), as_tuple=False).reshape(-1).contiguous()
        self.mask_index = idx
        self.mask_is_everything = int(idx.numel()) == int(model._conf_static.numel())
        # every pixel a Gaussian: the index is the identity, and the kernels take NULL for that (8 bytes per Gaussian less to read, twice per iteration)
        self.mask_ptr = None if self.mask_is_everything else C.c_void_p(idx.data_ptr())
        self.settings = {}


def _state(model):
    st = getattr(model, "_fast_state", None)
    if st is None or st.P != model._xyz.shape[0] or st.Qg.shape != model.Q.shape:
        st = model._fast_state = _State(model)
    return st


def _packed_sh(st, model, K):
    """The [P, K, 3] tensor the rasterizer reads above degree 0: DC + the first K - 1 rest coefficients of every Gaussian.  Concatenated once,
    then KEPT CURRENT BY THE OPTIMIZER: FusedAdam writes the updated values of f_dc and f_rest into it as it steps them
    (das3r_adam_tensor.mirror) — the parameters themselves stay the truth, this is a cache of them.  Rebuilt whenever it cannot be vouched
    for: another degree, other tensors, an in-place write torch knows of (_version), or a step of either parameter that did not write it."""
    dc, rest = model._features_dc, model._features_rest
    c = getattr(st, "sh_cache", None)
    key = (K, id(dc), id(rest), dc.data_ptr(), rest.data_ptr())
    if (c is None or c["key"] != key or c["ver"] != (dc._version, rest._version)
            or getattr(dc, "_das3r_mirror_ok", None) is not c["t"] or getattr(rest, "_das3r_mirror_ok", None) is not c["t"]):
        t = torch.cat((dc.detach(), rest.detach()[:, :K - 1]), dim=1)
        c = st.sh_cache = dict(key=key, ver=(dc._version, rest._version), t=t)
        dc._das3r_mirror, rest._das3r_mirror = (t, 0), (t, 3)
        dc._das3r_mirror_ok = rest._das3r_mirror_ok = t
    return c["t"]


def _dense_f32(obj, name):
    """obj.<name> as a contiguous fp32 tensor — the C-ABI takes plain pointers (a ground-truth image that came through the on-disk
    formats is a [3, H, W] VIEW of H x W x 3 memory: round 4's first version of this file read it as if it were dense and trained a
    whole job towards a scrambled image).  A copy is made once per tensor and kept on the object."""
    t = getattr(obj, name)
    if t.is_contiguous() and t.dtype == torch.float32:
        return t
    cache = getattr(obj, "_das3r_dense", None)
    if cache is None or cache[0] != (name, t.data_ptr(), tuple(t.shape), tuple(t.stride()), t._version):
        cache = ((name, t.data_ptr(), tuple(t.shape), tuple(t.stride()), t._version), t.detach().contiguous().float())
        obj._das3r_dense = cache
    return cache[1]


def _settings(st, cam, model, bg):
    """GaussianRasterizationSettings of render() for this camera (gaussian_renderer/__init__.py:53-78): identity view matrix,
    projmatrix = I @ P^T, campos = 0.  Built once per (image size, FoV, projection-matrix tensor, SH degree, background tensor) —
    keyed on the VALUES the settings are made of plus the identity of the two tensors they are built from, both of which the entry
    keeps alive (so neither id can be handed to another object while the entry exists); never on the camera object's own id
    (CPython hands the id of a freed camera to the next one: a camera made per iteration with another size or FoV was served the
    stale settings).  Rebuilt when the projection matrix was written since."""
    pm = cam.projection_matrix
    key = (int(cam.image_height), int(cam.image_width), float(cam.FoVx), float(cam.FoVy), model.active_sh_degree, id(bg), id(pm))
    hit = st.settings.get(key)
    if hit is not None and hit[1] is pm and hit[2] == pm._version and hit[0].bg is bg:
        return hit[0]
    dev = st.dev
    ident = torch.eye(4, device=dev)
    proj = ident @ pm.to(dev)
    rs = GaussianRasterizationSettings(
        image_height=int(cam.image_height), image_width=int(cam.image_width), tanfovx=math.tan(float(cam.FoVx) * 0.5),
        tanfovy=math.tan(float(cam.FoVy) * 0.5), bg=bg, scale_modifier=1.0, viewmatrix=ident.contiguous(), projmatrix=proj.contiguous(),
        sh_degree=model.active_sh_degree, campos=torch.zeros(3, device=dev), prefiltered=False, debug=False)
    if len(st.settings) > 4096:
        st.settings.clear()
    st.settings[key] = (rs, pm, pm._version)
    return rs


def forward_backward(model, cam, q_row, t_row, gq_row, gt_row, static_hw, lambda_dssim, bg, geometry="grads", rearm_rows=None):
    """Render `cam` with the pose (q_row, t_row: views of one row of Q / T), masked photometric loss against cam.original_image
    under `static_hw` [H, W], and the complete backward.  Gradients: model parameters' .grad (f_rest: compact or none, as in
    das3r_amd.render), the pose gradient into gq_row / gt_row, d loss / d static_hw returned.
    geometry: what becomes of the gradients of xyz / rotation / scaling / opacity, which exist in camera space when the rasterizer's backward
    returns — "grads": chain rule through the pre-transform, left on the parameters' .grad; "adam": the same chain rule with the Adam
    step of the four tensors taken in that very pass (das3r_pretransform_backward_adam: they never reach memory; model.optimizer.step()
    then finds the four without gradient and passes them by); "pose": the camera's sums alone, every per-Gaussian gradient dropped.
    -> (out8 = {loss, mse x 3, psnr_frame, ...} device tensor, d_static [H, W], package)"""
    st = _state(model)
    with _on_device(st.dev):   # (the library's per-device state and the raw stream belong to the model's GPU, current or not)
        return _forward_backward(st, model, cam, q_row, t_row, gq_row, gt_row, static_hw, lambda_dssim, bg, geometry, rearm_rows)


def _forward_backward(st, model, cam, q_row, t_row, gq_row, gt_row, static_hw, lambda_dssim, bg, geometry, rearm_rows=None):
    lib = _lib.load()
    dev, P = st.dev, st.P
    H, W = int(cam.image_height), int(cam.image_width)
    s = _stream(dev)
    # ---- pre-transform
    conf_flat = model._conf_static.view(-1)
    mats = st.mats
    _lib.check(lib.das3r_pose_matrices_qt(_p(q_row), _p(t_row), _p(mats), s), "das3r_pose_matrices_qt")
    pre = None
    if getattr(model, "fuse_pretransform", True):
        # round 6 (include/das3r_raster.h das3r_pretransform, ABI 14): the rasterizer's per-Gaussian kernels take the raw parameters and the pose —
        # the camera-frame means / rotations / scales / opacities never reach memory, and the pre-transform kernel is not launched.  The four
        # positional tensors below are placeholders of the right shapes (not read).  model.fuse_pretransform = False: the separate pass.
        pre = _lib.PreTransform()
        pre.xyz, pre.rot, pre.scaling, pre.opacity_raw = model._xyz.data_ptr(), model._rotation.data_ptr(), model._scaling.data_ptr(), model._opacity.data_ptr()
        pre.conf_flat, pre.mask_index = conf_flat.data_ptr(), (st.mask_ptr.value if st.mask_ptr is not None else None)
        pre.R, pre.t, pre.Lq = mats.data_ptr(), mats.data_ptr() + 36, mats.data_ptr() + 48
        means3D, rotations, scales, opac = model._xyz, model._rotation, model._scaling, model._opacity
    else:
        means3D, rotations = torch.empty_like(model._xyz), torch.empty_like(model._rotation)
        scales, opac = torch.empty_like(model._scaling), torch.empty(P, 1, device=dev)
        _lib.check(lib.das3r_pretransform_forward(P, _p(model._xyz), _p(model._rotation), _p(model._scaling), _p(model._opacity), _p(conf_flat),
                                                  st.mask_ptr, _p(mats), C.c_void_p(mats.data_ptr() + 36), C.c_void_p(mats.data_ptr() + 48),
                                                  _p(means3D), _p(rotations), _p(scales), _p(opac), s), "das3r_pretransform_forward")
    # ---- the SH tensor of the active degree (das3r_amd.render: DC alone at degree 0, the active prefix below the maximum)
    deg = model.active_sh_degree
    K = (deg + 1) ** 2
    if deg == 0:
        shs = model._features_dc
    elif deg < model.max_sh_degree and getattr(model, "sh_prefix", True):
        shs = _packed_sh(st, model, K)
    else:
        shs = _packed_sh(st, model, 1 + model._features_rest.shape[1])
    rs = _settings(st, cam, model, bg)
    e = st.e
    I, image, radii, geom, binning, img, cap = _forward_full(rs, means3D, shs, e, opac, scales, rotations, e, pre=pre)
    # ---- loss
    gt = _dense_f32(cam, "original_image")
    static_hw = static_hw if (static_hw.is_contiguous() and static_hw.dtype == torch.float32) else static_hw.contiguous().float()
    nb = int(lib.das3r_photometric_blocks(H, W))
    partials = torch.empty(nb, 8, device=dev)
    dmaps = torch.empty(4, 3, H, W, device=dev)
    out8 = torch.empty(8, device=dev)
    lam = float(lambda_dssim)
    _lib.check(lib.das3r_photometric_forward(H, W, _p(image), _p(gt), _p(static_hw), C.c_float(lam), _p(partials), _p(dmaps), s), "das3r_photometric_forward")
    d_render, d_static = torch.empty_like(image), torch.empty(H, W, device=dev)
    # (ABI 15: the loss / PSNR reduction of das3r_photometric_finish is the backward kernel's first workgroup's side duty — one launch less)
    _lib.check(lib.das3r_photometric_backward_finish(H, W, _p(image), _p(gt), _p(static_hw), C.c_float(lam), _p(dmaps), _p(st.one), _p(d_render),
                                                     _p(d_static), _p(partials), _p(out8), s), "das3r_photometric_backward_finish")
    # ---- rasterizer backward (examines the forward's binning self-check first: include/das3r_raster.h)
    # round 6 (include/das3r_raster.h das3r_chain): with geometry == "adam" the rasterizer's backward goes on through the pre-transform — chain
    # rule, the Adam step of xyz / rotation / scaling / opacity, dL/d(confidence), the pose sums — and the four camera-frame gradient tensors
    # are never written.  SH rows in an unstaged layout only (below 16 coefficients per Gaussian); model.fuse_backward_chain = False: two calls.
    chain, conf_grad = None, None
    if geometry == "adam" and pre is not None and getattr(model, "fuse_backward_chain", True) and shs.shape[1] < 16:
        opt_ = model.optimizer
        slots, keep = opt_.adam_slots([model._xyz, model._rotation, model._scaling, model._opacity])
        g_conf = torch.empty_like(conf_flat) if st.mask_is_everything else torch.zeros_like(conf_flat)
        chain = _lib.Chain()
        chain.g_conf_flat, chain.g_small, chain.slots = g_conf.data_ptr(), st.g_small.data_ptr(), slots
        chain.beta1, chain.beta2, chain.eps = opt_.betas[0], opt_.betas[1], opt_.eps
        conf_grad = g_conf
    g_means2D, _g_colors, g_opac, g_means3D, _g_cov, g_sh, g_scales, g_rot = _backward_impl(
        rs, I, d_render, means3D, shs, e, opac, scales, rotations, e, geom, binning, img, cap, pre=pre, chain=chain)
    # ---- pre-transform backward, pose chain rule
    if chain is not None:
        del keep
    elif geometry == "pose":
        _lib.check(lib.das3r_pretransform_pose_sums(P, _p(model._xyz), _p(model._rotation), _p(mats), C.c_void_p(mats.data_ptr() + 48), _p(g_means3D),
                                                    _p(g_rot), _p(st.g_small), s), "das3r_pretransform_pose_sums")
    else:
        g_conf = torch.empty_like(conf_flat) if st.mask_is_everything else torch.zeros_like(conf_flat)   # (every position is written when all pixels are Gaussians)
        if geometry == "adam":
            opt = model.optimizer
            slots, keep = opt.adam_slots([model._xyz, model._rotation, model._scaling, model._opacity])
            _lib.check(lib.das3r_pretransform_backward_adam(P, _p(conf_flat), st.mask_ptr, _p(mats), C.c_void_p(mats.data_ptr() + 48), _p(g_means3D),
                                                            _p(g_rot), _p(g_scales), _p(g_opac), _p(g_conf), _p(st.g_small), slots,
                                                            C.c_float(opt.betas[0]), C.c_float(opt.betas[1]), C.c_float(opt.eps), s),
                       "das3r_pretransform_backward_adam")
            del keep
        else:
            g_xyz, g_rotation, g_scaling = torch.empty_like(model._xyz), torch.empty_like(model._rotation), torch.empty_like(model._scaling)
            g_opacity_raw = torch.empty_like(model._opacity)
            _lib.check(lib.das3r_pretransform_backward(P, _p(model._xyz), _p(model._rotation), _p(model._scaling), _p(model._opacity), _p(conf_flat),
                                                       st.mask_ptr, _p(mats), C.c_void_p(mats.data_ptr() + 48), _p(g_means3D), _p(g_rot), _p(g_scales),
                                                       _p(g_opac), _p(g_xyz), _p(g_rotation), _p(g_scaling), _p(g_opacity_raw), _p(g_conf), _p(st.g_small), s),
                       "das3r_pretransform_backward")
            model._xyz.grad, model._rotation.grad, model._scaling.grad, model._opacity.grad = g_xyz, g_rotation, g_scaling, g_opacity_raw
        conf_grad = g_conf
    if rearm_rows is not None:   # (ABI 15: the previous view's rows of the dense pose gradients are zeroed by this launch, not by two fills)
        _lib.check(lib.das3r_pose_chain_qt_rearm(_p(q_row), _p(st.g_small), _p(gq_row), _p(gt_row), _p(rearm_rows[0]), _p(rearm_rows[1]), s),
                   "das3r_pose_chain_qt_rearm")
    else:
        _lib.check(lib.das3r_pose_chain_qt(_p(q_row), _p(st.g_small), _p(gq_row), _p(gt_row), s), "das3r_pose_chain_qt")
    # ---- hand the gradients over
    if geometry != "pose":
        if deg == 0:
            model._features_dc.grad = g_sh
        else:
            model._features_dc.grad = g_sh[:, :1]   # (column blocks of dL/dshs, read in place by FusedAdam with the row stride: no copies)
            rest = g_sh[:, 1:]
            if rest.shape[1] == K - 1 and K - 1 < model._features_rest.shape[1]:
                old = getattr(model._features_rest, "_das3r_compact_grad", None)
                model._features_rest._das3r_compact_grad = rest if old is None else old + rest
            else:
                model._features_rest.grad = rest
        model._conf_static.grad = conf_grad.view(model._conf_static.shape)
    st.means2D.grad = g_means2D
    pkg = _Pkg(render=image, viewspace_points=st.means2D, radii=radii)
    return out8, d_static, pkg


def train_step(model, cam, opt, iteration, pipe, background):
    """das3r_amd.train.train_step(fused=True) without autograd.  -> (loss, psnr_frame, package): 0-dim device tensors."""
    model.update_learning_rate(iteration)
    if iteration % 3000 == 0:
        model.oneupSHdegree()
    st = _state(model)
    uid = cam.uid
    with torch.no_grad(), _on_device(st.dev):   # (FusedAdam's launches too)
        # the dense pose gradients are zero outside the row of the view that stepped last; that row is zeroed by this step's pose chain launch
        prev = getattr(st, "dirty_uid", None)
        out8, d_static, pkg = forward_backward(model, cam, model.Q[uid], model.T[uid], st.Qg[uid], st.Tg[uid], model._conf_static[uid],
                                               opt.lambda_dssim, background, geometry="adam" if getattr(model, "fuse_geometry_adam", True) else "grads",
                                               rearm_rows=None if prev is None else (st.Qg[prev], st.Tg[prev]))
        st.dirty_uid = uid
        model._conf_static.grad[uid] += d_static             # the loss sees conf_static twice: as opacity factor and as the frame's mask
        model.optimizer.step()
        model.optimizer.zero_grad(set_to_none=True)
        model.Q.grad, model.T.grad = st.Qg, st.Tg
        model.optimizer_cam.step(gate=out8[4], threshold=opt.psnr_threshold)
        model.optimizer_cam.zero_grad(set_to_none=True)
    return out8[0], out8[4], pkg


def test_pose_step(model, cam, static_hw, opt, background):
    """One view of train_test_psnr.py's pass over the held-out views (das3r_amd.train.test_pose_pass): render with the test pose,
    loss, backward — and every gradient dropped: the Gaussian optimizer is zeroed without a step and optimizer_cam owns no
    gradient here.  Reproduced for its cost; nothing changes."""
    st = _state(model)
    uid = cam.uid
    if st.tQg is None:   # (the held-out poses were set after the first training step)
        st.tQg, st.tTg = torch.zeros_like(model.test_Q), torch.zeros_like(model.test_T)
    with torch.no_grad(), _on_device(st.dev):
        out8, _d_static, _pkg = forward_backward(model, cam, model.test_Q[uid], model.test_T[uid], st.tQg[uid], st.tTg[uid], static_hw,
                                                 opt.lambda_dssim, background, geometry="pose")
        model.optimizer.zero_grad(set_to_none=True)
        model.optimizer_cam.zero_grad(set_to_none=True)
    return out8
