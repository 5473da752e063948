"""Opt-in fused work around the rasterizer, host side.  Hand-written HIP behind the C-ABI
(include/das3r_raster.h: das3r_pretransform_*, das3r_adam_step); the default DAS3R-compatible path never needs this module.

  pretransform(...)  : fused counterpart of /root/reference/gaussian_renderer/__init__.py:83-97,107 (autograd.Function)
  FusedAdam          : torch.optim.Adam semantics for the reference's parameter groups
                       (/root/reference/scene/gaussian_model.py:236-261), one launch per step, SH-degree aware
"""
import ctypes as C
import math

import torch

from . import _lib


class AdamTensor(C.Structure):
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p), ("rows", C.c_int64),
                ("row_len", C.c_int32), ("active_len", C.c_int32), ("step_size", C.c_float), ("bc2_sqrt", C.c_float),
                ("head_len", C.c_int32), ("step_size_tail", C.c_float), ("grad_row_len", C.c_int32), ("state_row_len", C.c_int32),
                ("mirror", C.c_void_p), ("mirror_row_len", C.c_int32)]


def _stream(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _p(t):
    return C.c_void_p(t.data_ptr())


class _PreTransform(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz, rot, scaling, opacity_raw, conf, mask_index, pose):
        lib = _lib.load()
        dev = xyz.device
        if dev.type != "cuda":
            raise RuntimeError("das3r_amd.fused.pretransform: tensors must live on a HIP device; there is no CPU path")
        xyz, rot, scaling, opacity_raw = (a.contiguous() for a in (xyz, rot, scaling, opacity_raw))
        conf_flat = conf.contiguous().view(-1)
        pose = pose.contiguous().float()
        P = xyz.shape[0]
        mats = torch.empty(28, device=dev)    # R (9) | t (3) | Lq (16), built on the device from the 7-vector pose
        means3D, rotations = torch.empty_like(xyz), torch.empty_like(rot)
        scales, opac = torch.empty_like(scaling), torch.empty(P, 1, device=dev)
        with torch.cuda.device(dev):
            st = _stream(dev)
            _lib.check(lib.das3r_pose_matrices(_p(pose), _p(mats), st), "das3r_pose_matrices")
            rc = lib.das3r_pretransform_forward(P, _p(xyz), _p(rot), _p(scaling), _p(opacity_raw), _p(conf_flat),
                                                _p(mask_index) if mask_index is not None else None, _p(mats), C.c_void_p(mats.data_ptr() + 36),
                                                C.c_void_p(mats.data_ptr() + 48), _p(means3D), _p(rotations), _p(scales), _p(opac), st)
        _lib.check(rc, "das3r_pretransform_forward")
        ctx.save_for_backward(xyz, rot, scaling, opacity_raw, conf_flat, mats, pose)
        ctx.mask_index = mask_index
        ctx.conf_shape = conf.shape
        return means3D, rotations, scales, opac

    @staticmethod
    def backward(ctx, g_means3D, g_rot, g_scales, g_opac):
        lib = _lib.load()
        xyz, rot, scaling, opacity_raw, conf_flat, mats, pose = ctx.saved_tensors
        dev, P = xyz.device, xyz.shape[0]
        g_means3D = torch.zeros_like(xyz) if g_means3D is None else g_means3D.contiguous()
        g_rot = torch.zeros_like(rot) if g_rot is None else g_rot.contiguous()
        g_scales = torch.zeros_like(scaling) if g_scales is None else g_scales.contiguous()
        g_opac = torch.zeros(P, 1, device=dev) if g_opac is None else g_opac.contiguous()
        g_xyz, g_rotation, g_scaling = torch.empty_like(xyz), torch.empty_like(rot), torch.empty_like(scaling)
        g_opacity_raw = torch.empty_like(opacity_raw)
        g_conf = torch.zeros_like(conf_flat)
        g_small = torch.zeros(28, device=dev)
        g_pose = torch.empty(7, device=dev)
        mi = ctx.mask_index
        with torch.cuda.device(dev):
            st = _stream(dev)
            rc = lib.das3r_pretransform_backward(P, _p(xyz), _p(rot), _p(scaling), _p(opacity_raw), _p(conf_flat),
                                                 _p(mi) if mi is not None else None, _p(mats), C.c_void_p(mats.data_ptr() + 48), _p(g_means3D),
                                                 _p(g_rot), _p(g_scales), _p(g_opac), _p(g_xyz), _p(g_rotation), _p(g_scaling),
                                                 _p(g_opacity_raw), _p(g_conf), _p(g_small), st)
            _lib.check(rc, "das3r_pretransform_backward")
            _lib.check(lib.das3r_pose_chain(_p(pose), _p(g_small), _p(g_pose), st), "das3r_pose_chain")
        return g_xyz, g_rotation, g_scaling, g_opacity_raw, g_conf.view(ctx.conf_shape), None, g_pose


def pretransform(xyz, rot, scaling, opacity_raw, conf, mask_index, pose):
    """-> (means3D, rotations, scales, opacities) exactly as DAS3R's render() builds them from the raw parameters and the
    7-vector pose (qw,qx,qy,qz,tx,ty,tz): R from the normalised quaternion (get_camera_from_tensor), quadmultiply with the raw
    one.  Two launches each way (pose -> matrices / chain rule are one-lane kernels); no PyTorch glue."""
    return _PreTransform.apply(xyz, rot, scaling, opacity_raw, conf, mask_index, pose)


class _ShPrefix(torch.autograd.Function):
    """cat(f_dc, f_rest[:, :K - 1]) -> [P, K, 3]: the coefficients of the ACTIVE degree only (K = (degree + 1)^2).  Its backward
    hands f_dc its slice and leaves f_rest.grad None: the [P, K - 1, 3] gradient of the active prefix is parked on the parameter
    (`_das3r_compact_grad`) for FusedAdam, which reads it with its own row stride — a dense [P, 15, 3] gradient would be 180
    bytes per Gaussian of zeros written and read back every iteration."""

    @staticmethod
    def forward(ctx, f_dc, f_rest, K):
        ctx.rest = f_rest
        return torch.cat((f_dc, f_rest[:, :K - 1]), dim=1)

    @staticmethod
    def backward(ctx, g):
        # accumulates like .grad does: a second backward before the optimizer's step (gradient accumulation, two renders per
        # step, a retain_graph re-run as train_gui.py:579 permits) adds to the parked gradient instead of replacing it; a parked
        # gradient of another width (the degree went up in between without a step) cannot be added to and is an error
        new = g[:, 1:].contiguous()
        old = getattr(ctx.rest, "_das3r_compact_grad", None)
        if old is None:
            ctx.rest._das3r_compact_grad = new
        elif old.shape == new.shape:
            ctx.rest._das3r_compact_grad = old + new
        else:
            raise RuntimeError(f"fused._ShPrefix: a compact SH gradient of shape {tuple(old.shape)} is still parked on f_rest while one "
                               f"of shape {tuple(new.shape)} arrives: step() or zero_grad() the optimizer before changing the SH degree")
        return g[:, :1].contiguous(), None, None


def active_sh_prefix(f_dc, f_rest, degree):
    """The SH tensor the fused render hands the rasterizer while 0 < degree < max (das3r_amd.render); FusedAdam only."""
    return _ShPrefix.apply(f_dc, f_rest, (int(degree) + 1) ** 2)


def remap_state_dict(sd, optimizer):
    """A state_dict whose groups differ from `optimizer`'s in number (the fused camera optimizer holds the two pose groups, the plain
    one also the two inert field-of-view groups: model.py) re-keyed for it: groups are matched by their "name", parameters by their
    position inside the group; state of groups the target does not have is dropped, groups the checkpoint does not have start empty."""
    saved = {g.get("name", f"#{k}"): g for k, g in enumerate(sd["param_groups"])}
    groups, state, index = [], {}, 0
    for k, g in enumerate(optimizer.param_groups):
        src = saved.get(g.get("name", f"#{k}"))
        ids = list(range(index, index + len(g["params"])))
        if src is not None:
            for new, old in zip(ids, src["params"]):
                if old in sd["state"]:
                    state[new] = sd["state"][old]
        base = src if src is not None else {kk: v for kk, v in g.items() if kk != "params"}
        groups.append({**{kk: v for kk, v in base.items() if kk != "params"}, "params": ids})
        index += len(ids)
    return {**sd, "state": state, "param_groups": groups}


class FusedAdam:
    """torch.optim.Adam(lr=0.0, eps=1e-15)-compatible optimizer for lists of fp32 device tensors: same param_groups / step() /
    zero_grad() surface as the reference uses, one HIP launch per step.  A group may carry "sh_rest": True — its tensor is
    [P, K, 3] SH coefficients of which only those of the active degree are swept (set_active_sh_degree) — or "sh_all": True
    with "lr_rest": one [P, 1 + K, 3] tensor holding DC + rest (no torch.cat in the render path, no split in its backward),
    DC stepping with "lr", the rest with "lr_rest"."""

    is_fused = True   # das3r_amd.render: the DC-only SH shortcut needs an optimizer that counts the steps of a gradient-less f_rest

    def __init__(self, params, lr=0.0, betas=(0.9, 0.999), eps=1e-15):
        self.param_groups = []
        for g in params:
            g = dict(g)
            g.setdefault("lr", lr)
            g["params"] = list(g["params"])
            self.param_groups.append(g)
        self.betas, self.eps = betas, eps
        self.state = {}
        self.active_sh_degree = None
        self._gate_state = None   # device int32[2] of step(gate=...): steps actually taken, scratch flag

    def set_active_sh_degree(self, d):
        self.active_sh_degree = d

    # ---- checkpoints: torch.optim.Adam's state_dict layout, so that a checkpoint written with either optimizer loads into the other
    # (the reference writes optimizer.state_dict() into chkpnt<iteration>.pth: scene/gaussian_model.py:66-101, train_gui.py:626-628)
    def state_dict(self):
        index, groups, state = 0, [], {}
        for g in self.param_groups:
            ids = []
            for p in g["params"]:
                st = self.state.get(p)
                if st is not None:
                    sh = g.get("sh_rest") and p.dim() == 3   # (compact moments leave in the parameter's shape)
                    state[index] = dict(step=torch.tensor(float(st["step"])), exp_avg=self._full_moment(st["exp_avg"], p) if sh else st["exp_avg"],
                                        exp_avg_sq=self._full_moment(st["exp_avg_sq"], p) if sh else st["exp_avg_sq"])
                ids.append(index)
                index += 1
            groups.append({**{k: v for k, v in g.items() if k != "params"}, "params": ids, "betas": self.betas, "eps": self.eps})
        gate = None if self._gate_state is None else self._gate_state.clone()
        return dict(state=state, param_groups=groups, das3r=dict(active_sh_degree=self.active_sh_degree, gate_state=gate))

    def load_state_dict(self, sd):
        params = [p for g in self.param_groups for p in g["params"]]
        if [len(g["params"]) for g in sd["param_groups"]] != [len(g["params"]) for g in self.param_groups]:
            sd = remap_state_dict(sd, self)
        saved_groups = sd["param_groups"]
        if sum(len(g["params"]) for g in saved_groups) != len(params):
            raise ValueError("FusedAdam.load_state_dict: the checkpoint holds another number of parameters")
        self.state = {}
        for k, st in sd["state"].items():
            p = params[int(k)]
            if tuple(st["exp_avg"].shape) != tuple(p.shape):
                raise ValueError(f"FusedAdam.load_state_dict: parameter {k} has shape {tuple(p.shape)}, its moments {tuple(st['exp_avg'].shape)}")
            self.state[p] = dict(step=int(round(float(st["step"]))), exp_avg=st["exp_avg"].detach().to(p.device, torch.float32).clone().contiguous(),
                                 exp_avg_sq=st["exp_avg_sq"].detach().to(p.device, torch.float32).clone().contiguous())
        for g, sg in zip(self.param_groups, saved_groups):   # (the schedules set lr before every step; kept for completeness)
            if "lr" in sg:
                g["lr"] = sg["lr"]
        extra = sd.get("das3r") or {}
        if extra.get("active_sh_degree") is not None:
            self.active_sh_degree = extra["active_sh_degree"]
        for g in self.param_groups:   # moments of SH coefficients above the active degree that are all zero (they are, in any checkpoint of a run) are dropped
            for p in g["params"]:
                st = self.state.get(p)
                if st is not None and g.get("sh_rest") and p.dim() == 3:
                    cols = self._sh_cols(p)
                    if st["exp_avg"].shape[1] > cols and not bool(st["exp_avg"][:, cols:].any()) and not bool(st["exp_avg_sq"][:, cols:].any()):
                        st["exp_avg"], st["exp_avg_sq"] = st["exp_avg"][:, :cols].contiguous(), st["exp_avg_sq"][:, :cols].contiguous()
        gate = extra.get("gate_state")
        self._gate_state = None if gate is None else gate.detach().to(params[0].device).clone()
        if self._gate_state is not None:
            self._gate_state[1] = 0   # (scratch word: the arrival count of the gated kernel's workgroups, zero between launches — whatever an older checkpoint holds)

    # ---- compact moments of an "sh_rest" parameter (round 6).  Above the active degree gradient and moments are exactly zero; at degree 1 the
    # step used to stream 9 of every 45 floats of p, exp_avg and exp_avg_sq — six strided streams that cost what the whole tensors would
    # (adam_kernel 0.46 ms of a 1.99 ms iteration at 2.13 M Gaussians).  The moments are OURS to lay out: [P, active coefficients, 3], grown
    # with zeros when the degree goes up; state_dict() hands out the full shape torch.optim.Adam and the reference's checkpoints hold.
    def _sh_cols(self, p):
        return p.shape[1] if self.active_sh_degree is None else min(p.shape[1], (self.active_sh_degree + 1) ** 2 - 1)

    def _sh_state(self, p):
        cols = self._sh_cols(p)
        st = self.state.get(p)
        if st is None:
            z = lambda: torch.zeros(p.shape[0], cols, p.shape[2], dtype=torch.float32, device=p.device)
            st = self.state[p] = dict(step=0, exp_avg=z(), exp_avg_sq=z())
        elif st["exp_avg"].shape[1] < cols:
            for k in ("exp_avg", "exp_avg_sq"):
                pad = torch.zeros(p.shape[0], cols - st[k].shape[1], p.shape[2], dtype=torch.float32, device=p.device)
                st[k] = torch.cat((st[k], pad), dim=1).contiguous()
        return st

    @staticmethod
    def _full_moment(m, p):
        if tuple(m.shape) == tuple(p.shape):
            return m
        full = torch.zeros_like(p, dtype=torch.float32)
        full[:, :m.shape[1]] = m
        return full

    def handles_compact_sh(self, p):
        """True iff `p` sits in an "sh_rest" group of this optimizer, i.e. a gradient parked on it by fused._ShPrefix (or no
        gradient at all while only the DC tensor is rendered) is consumed / counted by step().  das3r_amd.render gates its SH
        shortcuts on this: with any other optimizer f_rest would silently never be updated."""
        return any(g.get("sh_rest") and any(q is p for q in g["params"]) for g in self.param_groups)

    def zero_grad(self, set_to_none=True):
        for g in self.param_groups:
            for p in g["params"]:
                if getattr(p, "_das3r_compact_grad", None) is not None:
                    p._das3r_compact_grad = None
                if set_to_none:
                    p.grad = None
                elif p.grad is not None:
                    p.grad.zero_()

    def adam_slots(self, params):
        """Counts one step of `params` (fp32 device tensors of this optimizer) and returns (_lib.AdamSlot array, keep-alive list) for a
        kernel that takes their Adam step itself (das3r_pretransform_backward_adam: das3r_amd/fast_step.py).  The parameters must not
        carry a gradient: step() then passes them by, as torch.optim.Adam passes a parameter without gradient."""
        b1, b2 = self.betas
        slots = (_lib.AdamSlot * len(params))()
        keep = []
        for k, p in enumerate(params):
            group = next((g for g in self.param_groups if any(q is p for q in g["params"])), None)
            if group is None:
                raise RuntimeError("FusedAdam.adam_slots: a tensor that is not a parameter of this optimizer")
            if p.grad is not None:
                raise RuntimeError("FusedAdam.adam_slots: the parameter already carries a gradient (its step would be taken twice)")
            if p.device.type != "cuda" or p.dtype != torch.float32 or not p.is_contiguous():
                raise RuntimeError("FusedAdam: dense fp32 tensors on a HIP device only (no CPU path)")
            st = self.state.get(p)
            if st is None:
                st = self.state[p] = dict(step=0, exp_avg=torch.zeros_like(p), exp_avg_sq=torch.zeros_like(p))
            st["step"] += 1
            t = st["step"]
            slots[k].param, slots[k].exp_avg, slots[k].exp_avg_sq = p.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
            slots[k].step_size, slots[k].bc2_sqrt = group["lr"] / (1.0 - b1 ** t), math.sqrt(1.0 - b2 ** t)
            keep += [p, st["exp_avg"], st["exp_avg_sq"]]
        return slots, keep

    @torch.no_grad()
    def step(self, gate=None, threshold=0.0):
        """gate: optional 0-dim device tensor — the step is taken iff gate > threshold, decided on the device (no host sync);
        the bias corrections then use a device-side count of the steps actually taken, exactly as if step() had only been
        called on those iterations."""
        lib = _lib.load()
        b1, b2 = self.betas
        entries, keep, dev = [], [], None
        for g in self.param_groups:
            for p in g["params"]:
                compact = getattr(p, "_das3r_compact_grad", None)   # (_ShPrefix: gradient of the active prefix)
                if compact is not None:
                    if not g.get("sh_rest"):
                        raise RuntimeError('FusedAdam: a parameter carries a compact SH gradient but its group is not marked "sh_rest"')
                    p._das3r_compact_grad = None   # consumed by this step (a stale one must not be applied again)
                    K1 = (self.active_sh_degree + 1) ** 2 - 1 if self.active_sh_degree is not None else -1
                    if compact.shape[1] != K1:
                        raise RuntimeError(f"FusedAdam: compact SH gradient has {compact.shape[1]} coefficients per Gaussian, the active "
                                           f"degree {self.active_sh_degree} needs {K1}")
                    if p.grad is not None:   # both a dense and a compact gradient arrived (mixed render paths): they add up
                        p.grad[:, :K1] += compact
                        compact = None
                if p.grad is None and compact is None:
                    # The higher-order SH coefficients get no gradient while the fused render hands the rasterizer the DC tensor
                    # alone (render.py).  In the reference they receive an all-zero gradient from iteration 1, so torch.optim.Adam
                    # counts those steps (moments stay 0): keep the same count, or the bias corrections would restart at 1 when
                    # the degree goes up at iteration 3000 (bc2 = 0.001 instead of 0.95: ~3x smaller first updates)
                    if g.get("sh_rest"):
                        st = self._sh_state(p) if p.dim() == 3 else self.state.get(p)
                        if st is None:
                            st = self.state[p] = dict(step=0, exp_avg=torch.zeros_like(p), exp_avg_sq=torch.zeros_like(p))
                        st["step"] += 1
                    continue
                if p.device.type != "cuda" or p.dtype != torch.float32:
                    raise RuntimeError("FusedAdam: fp32 tensors on a HIP device only (no CPU path)")
                dev = p.device
                sh_compact = bool(g.get("sh_rest")) and self.active_sh_degree is not None and p.dim() == 3
                st = self._sh_state(p) if sh_compact else self.state.get(p)
                if st is None:
                    st = self.state[p] = dict(step=0, exp_avg=torch.zeros_like(p), exp_avg_sq=torch.zeros_like(p))
                st["step"] += 1
                t = st["step"]
                if p.grad is None and (self.active_sh_degree is None or p.dim() != 3):
                    raise RuntimeError("FusedAdam: a compact SH gradient needs set_active_sh_degree() and a [P, K, 3] parameter")
                grad = p.grad if p.grad is not None else compact
                # a gradient that is a block of columns of a wider row-major tensor (round 6: the rasterizer's dL/dshs [P, K, 3], of which f_dc
                # takes column block 0 and f_rest the blocks behind it) is read in place, with its row stride — no copy per tensor and step
                row_strided = (grad.dim() >= 2 and not grad.is_contiguous() and grad.stride(-1) == 1 and grad.dtype == torch.float32
                               and all(grad.stride(i) == grad.stride(i + 1) * grad.shape[i + 1] for i in range(1, grad.dim() - 1)) and grad.stride(0) >= grad[0].numel())
                if not row_strided:
                    grad = grad.contiguous()
                keep.append(grad)
                e = AdamTensor()
                e.grad_row_len = 0
                e.param, e.grad, e.exp_avg, e.exp_avg_sq = p.data_ptr(), grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
                e.head_len, e.step_size_tail = 0, 0.0
                e.state_row_len = st["exp_avg"].shape[1] * st["exp_avg"].shape[2] if sh_compact else 0
                if g.get("sh_all") and p.dim() == 3:     # one [P, K, 3] tensor: DC coefficient (lr) + the rest (lr_rest)
                    e.rows, e.row_len = p.shape[0], p.shape[1] * p.shape[2]
                    deg = self.active_sh_degree if self.active_sh_degree is not None else 99
                    e.active_len = min(e.row_len, 3 * (deg + 1) ** 2)
                    e.head_len = 3
                    e.step_size_tail = g["lr_rest"] / (1.0 - b1 ** t)
                elif g.get("sh_rest") and self.active_sh_degree is not None and p.dim() == 3:
                    e.rows, e.row_len = p.shape[0], p.shape[1] * p.shape[2]
                    e.active_len = min(e.row_len, 3 * ((self.active_sh_degree + 1) ** 2 - 1))
                    if p.grad is None:   # the compact gradient [P, K - 1, 3] of _ShPrefix
                        e.grad_row_len = grad.shape[1] * grad.shape[2]
                        if e.grad_row_len < e.active_len:
                            raise RuntimeError("FusedAdam: the compact SH gradient is shorter than the active degree")
                else:
                    e.rows, e.row_len, e.active_len = 1, p.numel(), p.numel()
                    if p.numel() >= 2 ** 31:
                        raise RuntimeError("FusedAdam: tensor too large")
                    if (row_strided or getattr(p, "_das3r_mirror", None) is not None) and p.dim() >= 2:   # rows of p against rows of a wider tensor
                        e.rows, e.row_len = p.shape[0], p[0].numel()
                        e.active_len = e.row_len
                if row_strided:
                    e.grad_row_len = grad.stride(0)
                mirror = getattr(p, "_das3r_mirror", None)   # (tensor [P, K, 3], column offset in floats): das3r_amd/fast_step.py
                if mirror is not None:
                    p._das3r_mirror_ok = None   # (this step changes p: the mirror stays current only if the step writes it too)
                    mt, moff = mirror
                    if (p.dim() == 3 and mt.shape[0] == p.shape[0] and mt.is_contiguous() and mt.device == p.device and mt.dtype == torch.float32
                            and moff + e.active_len <= mt[0].numel() and gate is None):
                        e.mirror = mt.data_ptr() + 4 * moff
                        e.mirror_row_len = mt[0].numel()
                        keep.append(mt)
                        p._das3r_mirror_ok = mt
                if gate is None:
                    e.step_size = g["lr"] / (1.0 - b1 ** t)
                    e.bc2_sqrt = math.sqrt(1.0 - b2 ** t)
                else:           # plain learning rates: the kernel divides by the device-side bias correction
                    e.step_size, e.bc2_sqrt = g["lr"], 1.0
                    if e.head_len:
                        e.step_size_tail = g["lr_rest"]
                entries.append(e)
        for i in range(0, len(entries), 16):
            chunk = entries[i:i + 16]
            arr = (AdamTensor * len(chunk))(*chunk)
            with torch.cuda.device(dev):
                if gate is None:
                    rc = lib.das3r_adam_step(len(chunk), arr, C.c_float(b1), C.c_float(b2), C.c_float(self.eps), _stream(dev))
                else:
                    if len(entries) > 16:
                        raise RuntimeError("FusedAdam.step(gate=...): at most 16 tensors")
                    if self._gate_state is None:
                        self._gate_state = torch.zeros(2, dtype=torch.int32, device=dev)
                    gt = gate.detach().reshape(1).float().contiguous()
                    keep.append(gt)
                    rc = lib.das3r_adam_step_gated(len(chunk), arr, C.c_float(b1), C.c_float(b2), C.c_float(self.eps), _p(gt),
                                                   C.c_float(threshold), _p(self._gate_state), _stream(dev))
            _lib.check(rc, "das3r_adam_step")


class _Photometric(torch.autograd.Function):
    @staticmethod
    def forward(ctx, render, gt, static, lam):
        lib = _lib.load()
        dev = render.device
        if dev.type != "cuda":
            raise RuntimeError("das3r_amd.fused.masked_photometric_loss: tensors must live on a HIP device; there is no CPU path")
        if render.dim() != 3 or render.shape[0] != 3 or gt.shape != render.shape or static.shape != render.shape[1:]:
            raise ValueError("render / gt must be [3, H, W] and static [H, W]")
        render, gt, static = render.contiguous().float(), gt.contiguous().float(), static.contiguous().float()
        H, W = int(render.shape[1]), int(render.shape[2])
        nb = int(lib.das3r_photometric_blocks(H, W))
        partials = torch.empty(nb, 8, device=dev)
        dmaps = torch.empty(4, 3, H, W, device=dev)
        with torch.cuda.device(dev):
            rc = lib.das3r_photometric_forward(H, W, _p(render), _p(gt), _p(static), C.c_float(lam), _p(partials), _p(dmaps), _stream(dev))
        _lib.check(rc, "das3r_photometric_forward")
        sums = partials[:, :5].sum(0)
        n = float(H * W)
        loss = ((1.0 - lam) * sums[0] + lam * sums[1]) / (3.0 * n)
        mse = sums[2:5] / n
        ctx.save_for_backward(render, gt, static, dmaps)
        ctx.lam = float(lam)
        ctx.mark_non_differentiable(mse)
        return loss, mse

    @staticmethod
    def backward(ctx, g_loss, _g_mse):
        lib = _lib.load()
        render, gt, static, dmaps = ctx.saved_tensors
        dev = render.device
        H, W = int(render.shape[1]), int(render.shape[2])
        g = g_loss.reshape(1).contiguous().float()
        d_render, d_static = torch.empty_like(render), torch.empty_like(static)
        with torch.cuda.device(dev):
            rc = lib.das3r_photometric_backward(H, W, _p(render), _p(gt), _p(static), C.c_float(ctx.lam), _p(dmaps), _p(g), _p(d_render),
                                                _p(d_static), _stream(dev))
        _lib.check(rc, "das3r_photometric_backward")
        return d_render, None, d_static, None


class _SsimMap(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img1, img2):
        lib = _lib.load()
        dev = img1.device
        if dev.type != "cuda":
            raise RuntimeError("das3r_amd.fused.ssim_map: tensors must live on a HIP device; there is no CPU path")
        if img1.dim() != 3 or img1.shape[0] != 3 or img2.shape != img1.shape:
            raise ValueError("img1 / img2 must be [3, H, W]")
        a, b = img1.detach().contiguous().float(), img2.detach().contiguous().float()
        H, W = int(a.shape[1]), int(a.shape[2])
        m, dmaps = torch.empty_like(a), torch.empty(4, 3, H, W, device=dev)
        partials = torch.empty(int(lib.das3r_photometric_blocks(H, W)), 8, device=dev)
        with torch.cuda.device(dev):
            rc = lib.das3r_ssim_map_forward(H, W, _p(a), _p(b), _p(m), _p(dmaps), _p(partials), _stream(dev))
        _lib.check(rc, "das3r_ssim_map_forward")
        ctx.save_for_backward(a, b, dmaps)
        return m

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        a, b, dmaps = ctx.saved_tensors
        dev = a.device
        H, W = int(a.shape[1]), int(a.shape[2])
        g = g.contiguous().float()
        da, db = torch.empty_like(a), torch.empty_like(b)
        with torch.cuda.device(dev):
            rc = lib.das3r_ssim_map_backward(H, W, _p(a), _p(b), _p(dmaps), _p(g), _p(da), _p(db), _stream(dev))
        _lib.check(rc, "das3r_ssim_map_backward")
        return da, db


def ssim_map(img1, img2):
    """The SSIM map of two [3, H, W] images (utils/loss_utils.py:39-66 `ssim(..., size_average=False)`: 11 x 11 Gaussian window, sigma 1.5,
    zero padding, per channel), differentiable in both — two HIP launches each way instead of twelve depthwise convolutions and their
    elementwise chains.  For loops that compose the loss themselves (train_gui.py:566-571); das3r_amd.integrate.patch() puts it behind
    the reference's own `ssim`."""
    return _SsimMap.apply(img1, img2)


def masked_photometric_loss(render, gt, static, lambda_dssim):
    """-> (loss, mse[3]): DAS3R's iteration loss mean[(1 - lambda) |image - gt'| + lambda (1 - SSIM_map(image, gt'))] with
    image = render * static, gt' = gt * static, and the per-channel mean squared error of the same pair (for psnr_frame).
    Differentiable in `render` and `static`; two HIP kernels instead of PyTorch's convolution / elementwise chains."""
    return _Photometric.apply(render, gt, static, float(lambda_dssim))
