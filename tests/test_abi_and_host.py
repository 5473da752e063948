"""CPU tests of the drop-in boundary: the C-ABI library loads without a GPU and exports every symbol the header declares
(no compute calls here), and the host-side Python mirror keeps the reference's names, argument checks and error behaviour."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "das3r_raster.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(das3r_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(hip_lib):
    syms = _header_symbols()
    assert {"das3r_raster_forward", "das3r_raster_backward", "das3r_mark_visible", "das3r_knn3_mean_dist2",
            "das3r_knn3_workspace_bytes", "das3r_last_error"} <= set(syms)
    for s in syms:
        assert hasattr(hip_lib, s), f"libdas3r_hip.so does not export {s}"
    from das3r_amd import _lib
    assert set(_lib.EXPORTS) == set(syms), "das3r_amd/_lib.py EXPORTS out of sync with include/das3r_raster.h"


def test_abi_version_and_layout(hip_lib):
    from das3r_amd import _lib
    assert hip_lib.das3r_abi_version() == _lib.ABI_VERSION
    L = _lib.layout(1000, 5000, 1920, 1080)
    assert L["img_bytes"] >= 1920 * 1080 * 8 + 8160 * 8 and L["geom_bytes"] > 1000 * 60 and L["binning_bytes"] >= 5000 * 16
    offs = [L[k] for k in ("xy", "clamped", "tiles_touched", "offsets")]
    assert all(o % 256 == 0 for o in offs) and len(set(offs)) == len(offs)
    # xy / conic_opacity / rgbd: three float4 fields of one 64-byte record per Gaussian
    assert L["splat_stride"] == 64 and L["conic_opacity"] == L["xy"] + 16 and L["rgbd"] == L["xy"] + 32
    # struct sizes seen by ctypes must match what the header lays out (plain C ABI: ints, floats, pointers)
    assert ctypes.sizeof(_lib.RasterArgs) == 5 * 4 + 3 * 4 + 4 * 8 + 2 * 4 + 8
    assert ctypes.sizeof(_lib.RasterIn) == 7 * 8 + 8 and ctypes.sizeof(_lib.RasterGrads) == 9 * 8 + 8   # ABI 14: + pre (das3r_pretransform *), + chain (das3r_chain *)
    assert ctypes.sizeof(_lib.Chain) == 3 * 8 + 3 * 4 + 4 and ctypes.sizeof(_lib.AdamSlot) == 3 * 8 + 2 * 4
    assert ctypes.sizeof(_lib.PreTransform) == 9 * 8
    assert ctypes.sizeof(_lib.RasterSaved) == 5 * 8 + 8 + 8   # + check_word, check_tag and (ABI 14, in what was padding) flags
    assert _lib.RasterSaved.flags.offset == 52 and _lib.RasterSaved.check_tag.offset == 48
    assert hip_lib.das3r_raster_check(ctypes.byref(_lib.RasterSaved()), None) == 0   # no ticket: nothing to check


def test_invalid_arguments_are_reported_not_thrown(hip_lib):
    """No exception crosses the ABI: bad arguments give a negative status and a message."""
    from das3r_amd import _lib
    a, i, o, s = _lib.RasterArgs(), _lib.RasterIn(), _lib.RasterOut(), _lib.RasterSaved()
    a.P, a.image_width, a.image_height = 5, 0, 0
    cb = _lib.ALLOC_FN(lambda u, n: 0)
    rc = hip_lib.das3r_raster_forward(ctypes.byref(a), ctypes.byref(i), ctypes.byref(o), cb, cb, cb, None, ctypes.byref(s), None)
    assert rc == -1 and b"extents" in hip_lib.das3r_last_error()
    assert hip_lib.das3r_knn3_workspace_bytes(0) > 0 and hip_lib.das3r_knn3_workspace_bytes(100000) > 100000 * 30
    assert hip_lib.das3r_mark_visible(3, None, None, None, None, None) == -1


def test_settings_namedtuple_surface():
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    assert GaussianRasterizationSettings._fields == (
        "image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix", "projmatrix", "sh_degree",
        "campos", "prefiltered", "debug")
    rs = GaussianRasterizationSettings(image_height=8, image_width=8, tanfovx=1.0, tanfovy=1.0, bg=torch.zeros(3), scale_modifier=1.0,
                                       viewmatrix=torch.eye(4), projmatrix=torch.eye(4), sh_degree=0, campos=torch.zeros(3),
                                       prefiltered=False, debug=False)
    r = GaussianRasterizer(raster_settings=rs)
    assert isinstance(r, torch.nn.Module) and r.raster_settings is rs
    from simple_knn._C import distCUDA2
    assert callable(distCUDA2)


def _rast():
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    rs = GaussianRasterizationSettings(image_height=8, image_width=8, tanfovx=1.0, tanfovy=1.0, bg=torch.zeros(3), scale_modifier=1.0,
                                       viewmatrix=torch.eye(4), projmatrix=torch.eye(4), sh_degree=0, campos=torch.zeros(3),
                                       prefiltered=False, debug=False)
    return GaussianRasterizer(rs)


def test_argument_validation_messages_match_reference_wrapper():
    r = _rast()
    z = torch.zeros
    with pytest.raises(Exception, match="Please provide excatly one of either SHs or precomputed colors!"):
        r(means3D=z(2, 3), means2D=z(2, 3), opacities=z(2, 1), scales=z(2, 3), rotations=z(2, 4))
    with pytest.raises(Exception, match="Please provide excatly one of either SHs or precomputed colors!"):
        r(means3D=z(2, 3), means2D=z(2, 3), opacities=z(2, 1), shs=z(2, 1, 3), colors_precomp=z(2, 3), scales=z(2, 3), rotations=z(2, 4))
    with pytest.raises(Exception, match="exactly one of either scale/rotation pair or precomputed 3D covariance"):
        r(means3D=z(2, 3), means2D=z(2, 3), opacities=z(2, 1), shs=z(2, 1, 3), scales=z(2, 3))
    with pytest.raises(Exception, match="exactly one of either scale/rotation pair or precomputed 3D covariance"):
        r(means3D=z(2, 3), means2D=z(2, 3), opacities=z(2, 1), shs=z(2, 1, 3), scales=z(2, 3), rotations=z(2, 4), cov3D_precomp=z(2, 6))


def test_product_path_fails_loudly_without_gpu_tensors():
    """There is NO CPU fallback: CPU tensors are rejected with an error, never silently rendered on the host."""
    r = _rast()
    z = torch.zeros
    with pytest.raises(RuntimeError, match="no CPU path"):
        r(means3D=z(2, 3), means2D=z(2, 3), opacities=z(2, 1), shs=z(2, 1, 3), scales=z(2, 3), rotations=z(2, 4))
    from simple_knn._C import distCUDA2
    with pytest.raises(RuntimeError, match="no CPU path"):
        distCUDA2(z(10, 3))
    with pytest.raises(RuntimeError, match="no CPU path"):
        r.markVisible(z(4, 3))


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under the product packages may reference it."""
    bad = []
    for pkg in ("das3r_amd", "diff_gaussian_rasterization", "simple_knn"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, pkg)):
            for f in files:
                if f.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                    txt = open(os.path.join(dirpath, f), errors="replace").read()
                    if re.search(r"^\s*(from|import)\s+oracle\b|liboracle|raster_oracle|knn_oracle", txt, flags=re.M):
                        bad.append(os.path.join(dirpath, f))
    assert not bad, f"product files referencing the oracle: {bad}"


def test_missing_library_is_a_hard_error(monkeypatch, tmp_path):
    from das3r_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()


def test_host_pinning_round_trip(monkeypatch):
    """das3r_amd.hostpin: a worker is pinned to (at most) eight of the CPUs it was allowed, different ranks get different
    core complexes when there are enough of them, the previous mask comes back, and DAS3R_PIN=0 switches it off."""
    import os
    from das3r_amd.hostpin import pin_to_ccx, unpin
    if not hasattr(os, "sched_getaffinity"):
        pytest.skip("no sched_setaffinity on this platform")
    before = sorted(os.sched_getaffinity(0))
    try:
        p0 = pin_to_ccx(0)
        if p0 is None:
            assert len(before) < 8
            return
        assert p0[0] == before and len(p0[1]) == 8 and set(p0[1]) <= set(before)
        assert sorted(os.sched_getaffinity(0)) == p0[1]
        unpin(p0)
        assert sorted(os.sched_getaffinity(0)) == before
        p1 = pin_to_ccx(1)
        unpin(p1)
        if len(before) >= 32:
            assert set(p1[1]).isdisjoint(p0[1])
        monkeypatch.setenv("DAS3R_PIN", "0")
        assert pin_to_ccx(0) is None and sorted(os.sched_getaffinity(0)) == before
    finally:
        os.sched_setaffinity(0, before)
