"""ctypes binding of das3r_amd/libdas3r_hip.so (C-ABI: include/das3r_raster.h).

The product path has NO fallback: if the HIP library is missing or a call fails, a RuntimeError is raised.
PyTorch is used only for device memory (allocations handed to the library through the allocator callbacks)
and for the current HIP stream.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdas3r_hip.so")
ABI_VERSION = 15

ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t)


class RasterArgs(C.Structure):
    _fields_ = [("P", C.c_int32), ("sh_degree", C.c_int32), ("M", C.c_int32), ("image_width", C.c_int32),
                ("image_height", C.c_int32), ("tanfovx", C.c_float), ("tanfovy", C.c_float), ("scale_modifier", C.c_float),
                ("bg", C.c_void_p), ("viewmatrix", C.c_void_p), ("projmatrix", C.c_void_p), ("campos", C.c_void_p),
                ("prefiltered", C.c_int32), ("debug", C.c_int32), ("capacity_hint", C.c_int64)]


class PreTransform(C.Structure):
    """include/das3r_raster.h das3r_pretransform (ABI 14): raw model parameters + pose, taken by the rasterizer's own kernels."""
    _fields_ = [("xyz", C.c_void_p), ("rot", C.c_void_p), ("scaling", C.c_void_p), ("opacity_raw", C.c_void_p), ("conf_flat", C.c_void_p),
                ("mask_index", C.c_void_p), ("R", C.c_void_p), ("t", C.c_void_p), ("Lq", C.c_void_p)]


class RasterIn(C.Structure):
    _fields_ = [("means3D", C.c_void_p), ("opacities", C.c_void_p), ("shs", C.c_void_p), ("colors_precomp", C.c_void_p),
                ("scales", C.c_void_p), ("rotations", C.c_void_p), ("cov3D_precomp", C.c_void_p), ("pre", C.POINTER(PreTransform))]


class RasterOut(C.Structure):
    _fields_ = [("out_color", C.c_void_p), ("radii", C.c_void_p)]


class RasterSaved(C.Structure):
    _fields_ = [("geom", C.c_void_p), ("binning", C.c_void_p), ("img", C.c_void_p), ("num_rendered", C.c_int64), ("capacity", C.c_int64),
                ("check_word", C.c_void_p), ("check_tag", C.c_uint32), ("flags", C.c_uint32)]


class AdamSlot(C.Structure):
    """include/das3r_raster.h das3r_adam_slot (ABI 11)."""
    _fields_ = [("param", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p), ("step_size", C.c_float), ("bc2_sqrt", C.c_float)]


class Chain(C.Structure):
    """include/das3r_raster.h das3r_chain (ABI 14): the rasterizer's backward goes on through the pose pre-transform and the Adam step."""
    _fields_ = [("g_conf_flat", C.c_void_p), ("g_small", C.c_void_p), ("slots", C.POINTER(AdamSlot)), ("beta1", C.c_float), ("beta2", C.c_float),
                ("eps", C.c_float)]


class RasterGrads(C.Structure):
    _fields_ = [("dL_dmeans2D", C.c_void_p), ("dL_dopacities", C.c_void_p), ("dL_dmeans3D", C.c_void_p),
                ("dL_dshs", C.c_void_p), ("dL_dcolors_precomp", C.c_void_p), ("dL_dscales", C.c_void_p),
                ("dL_drotations", C.c_void_p), ("dL_dcov3D", C.c_void_p), ("scratch", C.c_void_p), ("chain", C.POINTER(Chain))]


class RasterLayout(C.Structure):
    _fields_ = [(n, C.c_size_t) for n in
                ("geom_bytes", "binning_bytes", "img_bytes", "depth_key", "xy", "conic_opacity", "rgbd", "splat_stride", "clamped",
                 "tiles_touched", "sorted_idx", "offsets", "point_list", "final_T", "n_contrib", "ranges")]


# every symbol include/das3r_raster.h declares
EXPORTS = ("das3r_raster_forward", "das3r_raster_backward", "das3r_raster_check", "das3r_raster_backward_scratch_bytes", "das3r_mark_visible", "das3r_knn3_workspace_bytes",
           "das3r_knn3_mean_dist2", "das3r_raster_get_layout", "das3r_abi_version", "das3r_last_error", "das3r_reload_switches", "das3r_get_stats",
           "das3r_raster_forget_shapes", "das3r_raster_learning",
           "das3r_profile_enable", "das3r_profile_report", "das3r_pretransform_forward", "das3r_pretransform_backward", "das3r_pose_matrices", "das3r_pose_chain",
           "das3r_adam_step", "das3r_adam_step_gated", "das3r_photometric_blocks", "das3r_photometric_forward", "das3r_photometric_backward",
           "das3r_has_experiments", "das3r_pair_counters", "das3r_debug_poison_lds", "das3r_debug_inject_fault", "das3r_debug_mutate",
           "das3r_pose_matrices_qt", "das3r_pose_chain_qt", "das3r_photometric_finish", "das3r_pretransform_backward_adam", "das3r_pretransform_pose_sums",
           "das3r_raster_count_live_pairs", "das3r_photometric_backward_finish", "das3r_pose_chain_qt_rearm", "das3r_ssim_map_forward", "das3r_ssim_map_backward")

_lib = None


def load():
    """Load libdas3r_hip.so (once).  Raises RuntimeError — never falls back — when it is absent or stale."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"das3r_amd: HIP library not found at {LIB_PATH}. Build it with `python -c 'import __graft_entry__ as g; "
            f"g.build()'` or `make -C das3r_amd/csrc`. There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    for name in EXPORTS:
        if not hasattr(L, name):
            raise RuntimeError(f"das3r_amd: {LIB_PATH} does not export {name}; rebuild it")
    L.das3r_abi_version.restype = C.c_int
    if L.das3r_abi_version() != ABI_VERSION:
        raise RuntimeError("das3r_amd: ABI version mismatch between the Python host layer and libdas3r_hip.so; rebuild")
    L.das3r_last_error.restype = C.c_char_p
    L.das3r_raster_forward.restype = C.c_int64
    L.das3r_raster_forward.argtypes = [C.POINTER(RasterArgs), C.POINTER(RasterIn), C.POINTER(RasterOut), ALLOC_FN, ALLOC_FN,
                                       ALLOC_FN, C.c_void_p, C.POINTER(RasterSaved), C.c_void_p]
    L.das3r_raster_backward.restype = C.c_int
    L.das3r_raster_backward.argtypes = [C.POINTER(RasterArgs), C.POINTER(RasterIn), C.POINTER(RasterSaved), C.c_void_p,
                                        C.POINTER(RasterGrads), C.c_void_p]
    L.das3r_raster_backward_scratch_bytes.restype = C.c_size_t
    L.das3r_raster_backward_scratch_bytes.argtypes = [C.c_int64]
    L.das3r_raster_check.restype = C.c_int
    L.das3r_raster_check.argtypes = [C.POINTER(RasterSaved), C.c_void_p]
    L.das3r_mark_visible.restype = C.c_int
    L.das3r_mark_visible.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.das3r_knn3_workspace_bytes.restype = C.c_size_t
    L.das3r_knn3_workspace_bytes.argtypes = [C.c_int32]
    L.das3r_knn3_mean_dist2.restype = C.c_int
    L.das3r_knn3_mean_dist2.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.das3r_pose_matrices.restype = C.c_int
    L.das3r_pose_matrices.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.das3r_pose_chain.restype = C.c_int
    L.das3r_pose_chain.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.das3r_pretransform_forward.restype = C.c_int
    L.das3r_pretransform_forward.argtypes = [C.c_int32] + [C.c_void_p] * 13 + [C.c_void_p]
    L.das3r_pretransform_backward.restype = C.c_int
    L.das3r_pretransform_backward.argtypes = [C.c_int32] + [C.c_void_p] * 18 + [C.c_void_p]
    L.das3r_pretransform_backward_adam.restype = C.c_int
    L.das3r_pretransform_backward_adam.argtypes = [C.c_int32] + [C.c_void_p] * 10 + [C.POINTER(AdamSlot), C.c_float, C.c_float, C.c_float, C.c_void_p]
    L.das3r_pretransform_pose_sums.restype = C.c_int
    L.das3r_pretransform_pose_sums.argtypes = [C.c_int32] + [C.c_void_p] * 7 + [C.c_void_p]
    L.das3r_photometric_blocks.restype = C.c_int64
    L.das3r_photometric_blocks.argtypes = [C.c_int32, C.c_int32]
    L.das3r_photometric_forward.restype = C.c_int
    L.das3r_photometric_forward.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
    L.das3r_photometric_backward.restype = C.c_int
    L.das3r_photometric_backward.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_void_p]
    L.das3r_pose_matrices_qt.restype = C.c_int
    L.das3r_pose_matrices_qt.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.das3r_pose_chain_qt.restype = C.c_int
    L.das3r_pose_chain_qt.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.das3r_pose_chain_qt_rearm.restype = C.c_int
    L.das3r_pose_chain_qt_rearm.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.das3r_ssim_map_forward.restype = C.c_int
    L.das3r_ssim_map_forward.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.das3r_ssim_map_backward.restype = C.c_int
    L.das3r_ssim_map_backward.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.das3r_photometric_backward_finish.restype = C.c_int
    L.das3r_photometric_backward_finish.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p,
                                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.das3r_photometric_finish.restype = C.c_int
    L.das3r_photometric_finish.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]
    L.das3r_adam_step_gated.restype = C.c_int
    L.das3r_adam_step_gated.argtypes = [C.c_int32, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]
    L.das3r_adam_step.restype = C.c_int
    L.das3r_adam_step.argtypes = [C.c_int32, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_void_p]
    L.das3r_raster_get_layout.restype = C.c_int
    L.das3r_raster_get_layout.argtypes = [C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.POINTER(RasterLayout)]
    _lib = L
    return L


def reload_switches():
    """Have the library read its DAS3R_* experiment switches from the environment again (it reads them once, at first use)."""
    L = load()
    L.das3r_reload_switches.restype = None
    L.das3r_reload_switches.argtypes = []
    L.das3r_reload_switches()


def has_experiments():
    """True when the library was built with `make EXPERIMENTS=1` (superseded kernels behind DAS3R_RENDER_BWD=mfma | stream,
    DAS3R_SORT=classic)."""
    L = load()
    L.das3r_has_experiments.restype = C.c_int
    L.das3r_has_experiments.argtypes = []
    return bool(L.das3r_has_experiments())


def inject_fault(bits):
    """Test aid: OR `bits` into the binning self-check word of every forward from now on; 0 switches it off (include/das3r_raster.h)."""
    L = load()
    L.das3r_debug_inject_fault.restype = None
    L.das3r_debug_inject_fault.argtypes = [C.c_uint32]
    L.das3r_debug_inject_fault(int(bits))


def forget_shapes():
    """The calling thread's library state on the current device forgets what it has learnt about shapes (include/das3r_raster.h
    das3r_raster_forget_shapes): a job that must end bit-identical whatever its thread ran before calls this first."""
    L = load()
    L.das3r_raster_forget_shapes.restype = C.c_int
    L.das3r_raster_forget_shapes.argtypes = []
    check(L.das3r_raster_forget_shapes(), "das3r_raster_forget_shapes")


def learning(state=None):
    """include/das3r_raster.h das3r_raster_learning: state=None reads (forward kernel, forwards so far) of the calling thread's current shape;
    a tuple hands it to the next shape the thread meets (a resumed job)."""
    L = load()
    L.das3r_raster_learning.restype = C.c_int
    L.das3r_raster_learning.argtypes = [C.c_int32, C.POINTER(C.c_uint32)]
    buf = (C.c_uint32 * 2)(*(state if state is not None else (0, 0)))
    check(L.das3r_raster_learning(0 if state is None else 1, buf), "das3r_raster_learning")
    return int(buf[0]), int(buf[1])


def mutate(what):
    """Test aid: 1 = the block-walk backward evaluates exp(power) (1 + 1e-4), 0 = off (include/das3r_raster.h das3r_debug_mutate)."""
    L = load()
    L.das3r_debug_mutate.restype = None
    L.das3r_debug_mutate.argtypes = [C.c_uint32]
    L.das3r_debug_mutate(int(what))


def poison_lds(pattern=0x7FC00000):
    """Test aid: fill the LDS of every CU with `pattern` on torch's current stream (include/das3r_raster.h)."""
    import torch
    L = load()
    L.das3r_debug_poison_lds.restype = C.c_int
    L.das3r_debug_poison_lds.argtypes = [C.c_uint32, C.c_void_p]
    check(L.das3r_debug_poison_lds(int(pattern), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "das3r_debug_poison_lds")


def pair_counters(enable):
    """enable=True: zero the compositing kernels' pair counters and start counting; enable=False: stop and return
    dict(fwd_pairs, bwd_pairs, fwd_wave_iterations, bwd_wave_iterations)."""
    L = load()
    L.das3r_pair_counters.restype = C.c_int
    L.das3r_pair_counters.argtypes = [C.c_int, C.POINTER(C.c_uint64)]
    if enable:
        check(L.das3r_pair_counters(1, None), "das3r_pair_counters")
        return None
    out = (C.c_uint64 * 4)()
    check(L.das3r_pair_counters(0, out), "das3r_pair_counters")
    return dict(fwd_pairs=int(out[0]), bwd_pairs=int(out[1]), fwd_wave_iterations=int(out[2]), bwd_wave_iterations=int(out[3]))


def stats():
    """-> dict(forwards, checks_examined, rescued_polls, failed_checks): the library's process-wide counters."""
    L = load()
    out = (C.c_uint64 * 4)()
    L.das3r_get_stats.restype = None
    L.das3r_get_stats.argtypes = [C.POINTER(C.c_uint64)]
    L.das3r_get_stats(out)
    return dict(forwards=int(out[0]), checks_examined=int(out[1]), rescued_polls=int(out[2]), failed_checks=int(out[3]))


def profile_enable(on):
    L = load()
    L.das3r_profile_enable.restype = None
    L.das3r_profile_enable.argtypes = [C.c_int]
    L.das3r_profile_enable(int(bool(on)))


def profile_report(raw=False):
    """-> {kernel name: (launches, total_ms)}; clears the records.  Template arguments are stripped from names unless `raw` (the
    names are the launch sites' own text, e.g. "render_backward_blk_kernel<MBV, PIX, 0, OCC, true>": the instantiation's LAST flag)."""
    L = load()
    L.das3r_profile_report.restype = C.c_int
    L.das3r_profile_report.argtypes = [C.c_char_p, C.c_size_t]
    buf = C.create_string_buffer(1 << 16)
    check(L.das3r_profile_report(buf, len(buf)), "das3r_profile_report")
    out = {}
    for line in buf.value.decode().splitlines():
        name, n, ms = line.rsplit(" ", 2)
        name = name.strip("() ")
        if not raw:
            name = name.split("<")[0]
        c, t = out.get(name, (0, 0.0))
        out[name] = (c + int(n), t + float(ms))
    return out


def splat_field(geom, L, name, P):
    """[P, 4] float32 view of one field ("xy" / "conic_opacity" / "rgbd") of the per-Gaussian 64-byte records in `geom`."""
    import torch
    nbytes = (P - 1) * L["splat_stride"] + 16 if P > 0 else 0
    fl = geom[L[name]:L[name] + nbytes].view(torch.float32)
    return torch.as_strided(fl, (P, 4), (L["splat_stride"] // 4, 1))


def last_error():
    return load().das3r_last_error().decode("utf-8", "replace")


def check(rc, what):
    if rc < 0:
        raise RuntimeError(f"{what} failed (status {rc}): {last_error()}")
    return rc


def layout(P, capacity, W, H):
    out = RasterLayout()
    check(load().das3r_raster_get_layout(int(P), int(capacity), int(W), int(H), C.byref(out)), "das3r_raster_get_layout")
    return {n: getattr(out, n) for n, _ in RasterLayout._fields_}
