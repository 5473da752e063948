"""Host-side mirror of the `diff_gaussian_rasterization` Python surface on top of the
MI355X C-ABI library.

Same names, argument meaning, return values and error behaviour as the module DAS3R imports at
/root/reference/gaussian_renderer/__init__.py:14-17 and calls at :62-80,131-140
(upstream:diff_gaussian_rasterization/__init__.py): `GaussianRasterizationSettings` (12-field NamedTuple),
`GaussianRasterizer(nn.Module)` with `forward(...) -> (color[3,H,W], radii[P])` and `markVisible`.
"""
import ctypes as C
import threading
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _lib


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


def cpu_deep_copy_tuple(input_tuple):
    return tuple(item.cpu().clone() if isinstance(item, torch.Tensor) else item for item in input_tuple)


def _ptr(t):
    return None if t is None or t.numel() == 0 else C.c_void_p(t.data_ptr())


_EMPTY = {}


def _empty(device):
    """The 'not provided' placeholder (upstream passes torch.Tensor([])): one cached empty tensor per device."""
    t = _EMPTY.get(device)
    if t is None:
        t = _EMPTY[device] = torch.empty(0, dtype=torch.float32, device=device)
    return t


def _prep(t, device, name):
    """contiguous fp32 tensor on `device` (empty tensors mean 'not provided', as upstream)."""
    if t is None:
        return _empty(device)
    if t.dtype == torch.float32 and t.device == device and t.is_contiguous():   # the usual case, first
        return t
    if t.numel() == 0:  # 'not provided' placeholders (upstream passes CPU torch.Tensor([])) and P == 0 inputs: keep the shape
        return t.to(device=device, dtype=torch.float32)
    if t.numel() and t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32 (got {t.dtype})")
    if t.device != device:
        raise RuntimeError(f"{name} is on {t.device}, expected {device}")
    return t.contiguous()


def _small(t, device, n, name):
    if not (isinstance(t, torch.Tensor) and t.dtype == torch.float32 and t.device == device and t.is_contiguous()):
        t = torch.as_tensor(t, dtype=torch.float32, device=device).contiguous()
    if t.numel() != n:
        raise ValueError(f"{name} must have {n} elements")
    return t


class _Alloc:
    """Allocator callbacks handed to the library (upstream's resizeFunctional): torch owns the bytes.  One instance per
    device and host thread is kept alive (building ctypes callbacks costs tens of microseconds; threads must not share one:
    a forward on another thread would drop this thread's buffers); `take()` hands the buffers of the call that just finished
    to the caller and forgets them."""

    _local = threading.local()

    @classmethod
    def get(cls, device):
        per_device = getattr(cls._local, "per_device", None)
        if per_device is None:
            per_device = cls._local.per_device = {}
        a = per_device.get(device)
        if a is None:
            a = per_device[device] = cls(device)
        a.bufs = {}
        return a

    def take(self):
        b, self.bufs = self.bufs, {}
        return b

    def __init__(self, device):
        self.device = device
        self.bufs = {}
        self.fns = {k: _lib.ALLOC_FN(self._make(k)) for k in ("geom", "binning", "img")}

    def _make(self, key):
        def fn(_user, nbytes):
            try:
                t = torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=self.device)
                self.bufs[key] = t
                return t.data_ptr()
            except Exception:  # noqa: BLE001 - reported by the library as DAS3R_ERR_ALLOC
                return 0
        return fn


def _fill_args(rs, P, M, device, keep):
    a = _lib.RasterArgs()
    a.P, a.sh_degree, a.M = P, int(rs.sh_degree), M
    a.image_width, a.image_height = int(rs.image_width), int(rs.image_height)
    a.tanfovx, a.tanfovy, a.scale_modifier = float(rs.tanfovx), float(rs.tanfovy), float(rs.scale_modifier)
    bg = _small(rs.bg, device, 3, "bg")
    vm = _small(rs.viewmatrix, device, 16, "viewmatrix")
    pm = _small(rs.projmatrix, device, 16, "projmatrix")
    cp = _small(rs.campos, device, 3, "campos")
    keep.extend([bg, vm, pm, cp])
    a.bg, a.viewmatrix, a.projmatrix, a.campos = bg.data_ptr(), vm.data_ptr(), pm.data_ptr(), cp.data_ptr()
    a.prefiltered, a.debug = int(bool(rs.prefiltered)), int(bool(rs.debug))
    return a


def _fill_in(means3D, opacities, sh, colors_precomp, scales, rotations, cov3Ds_precomp, pre=None):
    i = _lib.RasterIn()
    i.means3D, i.opacities = _ptr(means3D), _ptr(opacities)
    i.shs, i.colors_precomp = _ptr(sh), _ptr(colors_precomp)
    i.scales, i.rotations, i.cov3D_precomp = _ptr(scales), _ptr(rotations), _ptr(cov3Ds_precomp)
    if pre is not None:   # a _lib.PreTransform the caller keeps alive: the kernels take the pose pre-transform on their way in (ABI 14)
        i.pre = C.pointer(pre)
    return i


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(device):
    if _raw_stream is not None and device.index is not None:
        return C.c_void_p(_raw_stream(device.index))
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class _on_device:
    """`with torch.cuda.device(device)` only when `device` is not already current (the context manager costs ~5 us)."""

    def __init__(self, device):
        self.ctx = None if device.index is None or torch.cuda.current_device() == device.index else torch.cuda.device(device)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *exc):
        if self.ctx is not None:
            return self.ctx.__exit__(*exc)
        return False


class _Capacity(int):
    """Instance capacity of a forward's binning buffer; also carries the forward's binning self-check ticket
    (das3r_raster_saved.check_word / check_tag) to the backward pass and to `check_forward`."""
    check_word = None
    check_tag = 0
    flags = 0   # das3r_raster_saved.flags of the forward (which compositing kernels its lists call for)


def check_forward(capacity, device):
    """Wait for the binning self-check of the forward that returned `capacity` (the 7th element of `_forward_full`'s result) and
    raise RuntimeError if that forward's image is invalid — for callers that render without a backward pass (evaluation);
    the backward pass does this itself before it launches anything (include/das3r_raster.h: das3r_raster_check)."""
    if not getattr(capacity, "check_tag", 0):
        return
    saved = _lib.RasterSaved()
    saved.check_word, saved.check_tag = capacity.check_word, capacity.check_tag
    with _on_device(device):
        rc = _lib.load().das3r_raster_check(C.byref(saved), _stream(device))
    _lib.check(rc, "das3r_raster_check")


def _forward_impl(rs, means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp):
    """-> (num_rendered, color, radii, geom, binning, img), the binning buffer laid out for exactly num_rendered instances
    (inspection helper of the tests and tools: `_lib.layout(P, num_rendered, W, H)` then describes the buffers)."""
    return _forward_full(rs, means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, exact=True)[:6]


def _forward_full(rs, means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, exact=False, pre=None):
    """-> (num_rendered, color, radii, geom, binning, img, capacity); capacity >= num_rendered is what the binning buffer
    was laid out for (`_lib.layout(P, capacity, W, H)`), == num_rendered when `exact`."""
    lib = _lib.load()
    device = means3D.device
    if device.type != "cuda":
        raise RuntimeError("das3r_amd rasterizer: tensors must live on a HIP device (torch device 'cuda'); "
                           "there is no CPU path")
    if means3D.dim() != 2 or means3D.shape[1] != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    P = means3D.shape[0]
    H, W = int(rs.image_height), int(rs.image_width)
    M = 0
    if sh.numel():
        if sh.dim() != 3 or sh.shape[0] != P or sh.shape[2] != 3:
            raise RuntimeError("shs must have dimensions (num_points, M, 3)")
        M = sh.shape[1]
    if P == 0:   # upstream: zero image, background not applied, empty radii
        e = torch.empty(0, dtype=torch.uint8, device=device)
        return (0, torch.zeros(3, H, W, dtype=torch.float32, device=device), torch.zeros(0, dtype=torch.int32, device=device),
                e, e, e, 0)
    # every pixel and every radii entry is written by the kernels: no memset needed
    color = torch.empty(3, H, W, dtype=torch.float32, device=device)
    radii = torch.empty(P, dtype=torch.int32, device=device)
    alloc = _Alloc.get(device)
    keep = []
    a = _fill_args(rs, P, M, device, keep)
    a.capacity_hint = -1 if exact else 0   # 0: the library may lay the binning buffer out with headroom (include/das3r_raster.h)
    i = _fill_in(means3D, opacities, sh, colors_precomp, scales, rotations, cov3Ds_precomp, pre)
    o = _lib.RasterOut()
    o.out_color, o.radii = color.data_ptr(), radii.data_ptr()
    saved = _lib.RasterSaved()
    with _on_device(device):
        rc = lib.das3r_raster_forward(C.byref(a), C.byref(i), C.byref(o), alloc.fns["geom"], alloc.fns["binning"],
                                      alloc.fns["img"], None, C.byref(saved), _stream(device))
    _lib.check(rc, "das3r_raster_forward")
    empty = torch.empty(0, dtype=torch.uint8, device=device)
    bufs = alloc.take()
    cap = _Capacity(saved.capacity)
    cap.check_word, cap.check_tag, cap.flags = saved.check_word, int(saved.check_tag), int(saved.flags)
    return (int(rc), color, radii, bufs.get("geom", empty), bufs.get("binning", empty), bufs.get("img", empty), cap)


def _backward_impl(rs, num_rendered, grad_out_color, means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                   geom, binning, img, capacity=None, _scratch_misalign=0, pre=None, chain=None):
    ticket = capacity
    capacity = int(num_rendered) if capacity is None else int(capacity)
    lib = _lib.load()
    device = means3D.device
    P = means3D.shape[0]
    M = sh.shape[1] if sh.numel() else 0
    z = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=device)  # fully written by the library
    has_sh, has_cov = sh.numel() > 0, cov3Ds_precomp.numel() > 0
    # chain (a _lib.Chain the caller keeps alive, with `pre`): the library goes on through the pose pre-transform and the Adam step of the four
    # geometry tensors itself — dL/d(camera-frame means, opacities, scales, rotations) are not produced (returned as None)
    g_means2D = z(P, 3)
    g_opac, g_means3D = (None, None) if chain is not None else (z(P, 1), z(P, 3))
    # only the gradients this call's inputs have (the others are returned as None by the autograd function)
    g_sh = z(P, M, 3) if has_sh else None
    g_colors = None if has_sh else z(P, 3)
    g_scales, g_rot = (None, None) if (has_cov or chain is not None) else (z(P, 3), z(P, 4))
    g_cov = z(P, 6) if has_cov else None
    if P == 0:
        return g_means2D, g_colors, g_opac, g_means3D, g_cov, g_sh, g_scales, g_rot
    # per-instance partial sums, written by the render backward and added per Gaussian (no atomics, no memset by the caller)
    # (_scratch_misalign: tests hand the library a scratch buffer that is only 4-byte aligned — a C caller may)
    scratch = torch.empty(int(lib.das3r_raster_backward_scratch_bytes(max(capacity, 1))) + int(_scratch_misalign), dtype=torch.uint8, device=device)
    keep = []
    a = _fill_args(rs, P, M, device, keep)
    i = _fill_in(means3D, opacities, sh, colors_precomp, scales, rotations, cov3Ds_precomp, pre)
    saved = _lib.RasterSaved()
    saved.geom, saved.binning, saved.img = _ptr(geom), _ptr(binning), _ptr(img)
    saved.num_rendered = int(num_rendered)
    saved.capacity = capacity
    if getattr(ticket, "check_tag", 0):   # the forward's binning self-check is examined before the backward launches anything
        saved.check_word, saved.check_tag = ticket.check_word, ticket.check_tag
    saved.flags = int(getattr(ticket, "flags", 0))
    g = _lib.RasterGrads()
    g.dL_dmeans2D, g.dL_dopacities, g.dL_dmeans3D = g_means2D.data_ptr(), _ptr(g_opac), _ptr(g_means3D)
    if chain is not None:
        g.chain = C.pointer(chain)
    g.dL_dshs = _ptr(g_sh)
    g.dL_dcolors_precomp = _ptr(g_colors)
    g.dL_dscales, g.dL_drotations = _ptr(g_scales), _ptr(g_rot)
    g.dL_dcov3D = _ptr(g_cov)
    g.scratch = scratch.data_ptr() + int(_scratch_misalign)
    dL = grad_out_color.contiguous()
    if dL.dtype != torch.float32:
        dL = dL.float()
    with _on_device(device):
        rc = lib.das3r_raster_backward(C.byref(a), C.byref(i), C.byref(saved), C.c_void_p(dL.data_ptr()), C.byref(g),
                                       _stream(device))
    _lib.check(rc, "das3r_raster_backward")
    return g_means2D, g_colors, g_opac, g_means3D, g_cov, g_sh, g_scales, g_rot


def count_live_pairs(rs, P, M, num_rendered, geom, binning, img, capacity):
    """Measurement aid (include/das3r_raster.h das3r_raster_count_live_pairs): -> (live pairs, (pixel, list position) pairs below the
    pixels' last contributors) of the forward that produced these buffers (`_forward_full`'s results)."""
    lib = _lib.load()
    device = geom.device
    keep = []
    a = _fill_args(rs, P, M, device, keep)
    saved = _lib.RasterSaved()
    saved.geom, saved.binning, saved.img = _ptr(geom), _ptr(binning), _ptr(img)
    saved.num_rendered, saved.capacity = int(num_rendered), int(capacity)
    out = (C.c_uint64 * 2)()
    lib.das3r_raster_count_live_pairs.restype = C.c_int
    lib.das3r_raster_count_live_pairs.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p]
    with _on_device(device):
        rc = lib.das3r_raster_count_live_pairs(C.byref(a), C.byref(saved), out, _stream(device))
    _lib.check(rc, "das3r_raster_count_live_pairs")
    return int(out[0]), int(out[1])


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                                     raster_settings)


_last = threading.local()


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings):
        device = means3D.device
        means3D = _prep(means3D, device, "means3D")
        sh = _prep(sh, device, "shs")
        colors_precomp = _prep(colors_precomp, device, "colors_precomp")
        opacities = _prep(opacities, device, "opacities")
        scales = _prep(scales, device, "scales")
        rotations = _prep(rotations, device, "rotations")
        cov3Ds_precomp = _prep(cov3Ds_precomp, device, "cov3D_precomp")
        args = (raster_settings, means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp)
        if raster_settings.debug:
            cpu_args = cpu_deep_copy_tuple(args)  # copy them before they can be corrupted
            try:
                num_rendered, color, radii, geomBuffer, binningBuffer, imgBuffer, capacity = _forward_full(*args)
            except Exception as ex:
                torch.save(cpu_args, "snapshot_fw.dump")
                print("\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.")
                raise ex
        else:
            num_rendered, color, radii, geomBuffer, binningBuffer, imgBuffer, capacity = _forward_full(*args)
        ctx.raster_settings = raster_settings
        ctx.num_rendered = num_rendered
        ctx.capacity = capacity
        _last.capacity = capacity   # GaussianRasterizer.forward: a render that no backward pass will follow checks itself
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, opacities, geomBuffer,
                              binningBuffer, imgBuffer)
        ctx.mark_non_differentiable(radii)
        ctx.set_materialize_grads(False)   # no zero-filled 'gradient' for radii on every backward (a fill kernel per step)
        return color, radii

    @staticmethod
    def backward(ctx, grad_out_color, _grad_radii):
        rs = ctx.raster_settings
        if grad_out_color is None:   # (grads are not materialised) nothing flows back
            return (None,) * 9
        (colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, opacities, geomBuffer, binningBuffer,
         imgBuffer) = ctx.saved_tensors
        args = (rs, ctx.num_rendered, grad_out_color, means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                geomBuffer, binningBuffer, imgBuffer, ctx.capacity)
        if rs.debug:
            cpu_args = cpu_deep_copy_tuple(args)
            try:
                out = _backward_impl(*args)
            except Exception as ex:
                torch.save(cpu_args, "snapshot_bw.dump")
                print("\nAn error occured in backward. Writing snapshot_bw.dump for debugging.\n")
                raise ex
        else:
            out = _backward_impl(*args)
        g_means2D, g_colors, g_opac, g_means3D, g_cov, g_sh, g_scales, g_rot = out
        return (g_means3D, g_means2D, g_sh if sh.numel() else None, g_colors if colors_precomp.numel() else None, g_opac,
                g_scales if scales.numel() else None, g_rot if rotations.numel() else None,
                g_cov if cov3Ds_precomp.numel() else None, None)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        with torch.no_grad():
            rs = self.raster_settings
            device = positions.device
            if device.type != "cuda":
                raise RuntimeError("das3r_amd rasterizer: positions must live on a HIP device; there is no CPU path")
            pos = _prep(positions, device, "positions")
            P = pos.shape[0]
            present = torch.zeros(P, dtype=torch.uint8, device=device)
            if P:
                vm = _small(rs.viewmatrix, device, 16, "viewmatrix")
                pm = _small(rs.projmatrix, device, 16, "projmatrix")
                with torch.cuda.device(device):
                    rc = _lib.load().das3r_mark_visible(P, _ptr(pos), _ptr(vm), _ptr(pm), _ptr(present), _stream(device))
                _lib.check(rc, "das3r_mark_visible")
            return present.bool()

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        raster_settings = self.raster_settings
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        e = _empty(means3D.device)   # upstream: torch.Tensor([]) placeholders
        if shs is None:
            shs = e
        if colors_precomp is None:
            colors_precomp = e
        if scales is None:
            scales = e
        if rotations is None:
            rotations = e
        if cov3D_precomp is None:
            cov3D_precomp = e
        _last.capacity = None
        color, radii = rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                           raster_settings)
        if not color.requires_grad and getattr(_last, "capacity", None) is not None:
            # evaluation (torch.no_grad, or no input that takes a gradient): no backward pass will examine this forward's binning
            # self-check, so it is examined here, before the image is used (include/das3r_raster.h: das3r_raster_check)
            check_forward(_last.capacity, means3D.device)
        _last.capacity = None
        return color, radii
