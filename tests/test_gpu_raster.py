"""GPU parity tests proper: the HIP path (through the C-ABI, via the drop-in GaussianRasterizer surface) against the CPU
oracle on the same seeded inputs.  Integer / index work (radii, tile counts, per-tile splat order, ranges) must be
bit-exact; floating point within the tolerances stated in tests/util.py."""
import numpy as np
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


def _view(buf, off, dtype, count):
    item = torch.tensor([], dtype=dtype).element_size()
    return buf[off:off + count * item].view(dtype)


LANES_COLOR_TOL = 2e-6   # render_lanes.hip: the same colour terms in four partial sums per pixel (colours in [0, 1])


def _run_hip(sc, mode, backward=True):
    from das3r_amd import GaussianRasterizationSettings, GaussianRasterizer
    dev = _dev()
    kw = {k: v.to(dev).clone().requires_grad_(True) for k, v in util.raster_inputs(sc, mode).items()}
    skw = util.settings_kwargs(sc, mode)
    skw = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in skw.items()}
    rs = GaussianRasterizationSettings(**skw)
    means2D = torch.zeros(sc.P, 3, device=dev, requires_grad=True)
    color, radii = GaussianRasterizer(rs)(means2D=means2D, **kw)
    grads = None
    if backward:
        color.backward(sc.dL_dpix.to(dev))
        grads = {k: v.grad for k, v in kw.items()}
        grads["means2D"] = means2D.grad
    torch.cuda.synchronize()
    return color.detach(), radii, grads, color.grad_fn


@pytest.mark.parametrize("name", util.VARIANTS)
def test_forward_backward_vs_oracle(name):
    sc, mode = util.scene_variant(name)
    ref_color, ref_radii, ref_g, S = util.run_oracle(sc, mode)
    color, radii, g, fn = _run_hip(sc, mode)
    assert np.array_equal(radii.cpu().numpy(), ref_radii), "radii must match the oracle exactly"
    assert 0 < fn.num_rendered <= S["num_rendered"] or S["num_rendered"] == 0, "binned instances must be a subset of upstream's"
    util.assert_color_close(color.cpu().numpy(), ref_color, f"{name} colour")
    gmap = {"means3D": "means3D", "opacities": "opacities", "shs": "shs", "scales": "scales", "rotations": "rotations",
            "means2D": "means2D", "colors_precomp": "colors", "cov3D_precomp": "cov3D"}
    for k, t in g.items():
        assert t is not None, f"no gradient for {k}"
        util.assert_grad_close(t.cpu().numpy(), ref_g[gmap[k]], f"{name} dL/d{k}")
        util.assert_grad_elementwise(t.cpu().numpy(), ref_g[gmap[k]], f"{name} dL/d{k}")
    # viewspace gradient convention: z component stays zero
    assert float(g["means2D"][:, 2].abs().max()) == 0.0

@pytest.mark.parametrize("degree", [0, 1, 2])
def test_compact_sh_layouts_and_ragged_waves_vs_oracle(degree):
    """SH tensors that hold exactly (D + 1)^2 coefficients (M = 1, 4, 9: what the fused render path hands over while the active
    degree is 0, and what vanilla 3DGS checkpoints of lower degree hold) take the unstaged per-Gaussian kernels, whose 12-byte
    gradient rows leave through the wave-private LDS transpose (preprocess_bwd.hip wave_store_rows); P = 64 k + 37 leaves the
    last wave of the last workgroup partly empty."""
    from das3r_amd.synth import make_scene
    sc = make_scene(P=64 * 21 + 37, W=112, H=80, focal=90.0, sh_degree=degree, seed=50 + degree, max_sh_degree=degree)
    assert sc.shs.shape[1] == (degree + 1) ** 2
    mode = dict(colors_precomp=False, cov3D_precomp=False, scale_modifier=1.0)
    ref_color, ref_radii, ref_g, S = util.run_oracle(sc, mode)
    color, radii, g, fn = _run_hip(sc, mode)
    assert np.array_equal(radii.cpu().numpy(), ref_radii)
    util.assert_color_close(color.cpu().numpy(), ref_color, f"M={sc.shs.shape[1]} colour")
    gmap = {"means3D": "means3D", "opacities": "opacities", "shs": "shs", "scales": "scales", "rotations": "rotations", "means2D": "means2D"}
    for k, t in g.items():
        util.assert_grad_close(t.cpu().numpy(), ref_g[gmap[k]], f"M={sc.shs.shape[1]} dL/d{k}")
        util.assert_grad_elementwise(t.cpu().numpy(), ref_g[gmap[k]], f"M={sc.shs.shape[1]} dL/d{k}")



@pytest.mark.parametrize("binning_path", ["radix", "local", "seg", "seg3"])
@pytest.mark.parametrize("name", ["basic_deg3", "ragged_image", "long_lists", "deep", "culled", "depth_ties", "world_camera"])
def test_binning_bit_exact(name, binning_path, monkeypatch):
    """Per-Gaussian geometry, depth order, per-tile splat lists and tile ranges are integer / exactly-rounded fp32 work:
    they must equal the oracle's bit for bit (binning over upstream's 3-sigma square: DAS3R_RECT=upstream; the default
    opacity-aware clipped rectangle is covered by test_tight_rect_is_exact) — with the global depth sort and with the local
    depth order (lists emitted in index order and sorted by the compositing kernel: in LDS, or, "long_lists", in global
    memory when a list has more than 1024 entries), and with the segmented path of round 4 ("seg": partition by (tile, depth
    bucket), then segment_sort_kernel — segkey.h, segsort.hip; "seg3": the same with one more partition pass of bucket bits, what a
    shape takes after a forward had to rank a long segment)."""
    monkeypatch.setenv("DAS3R_RECT", "upstream")
    monkeypatch.setenv("DAS3R_BINNING", binning_path)
    from das3r_amd import _lib
    _lib.profile_report()   # drain
    _lib.profile_enable(True)
    sc, mode = util.scene_variant(name)
    L0_passes = 1 if ((sc.W + 15) // 16) * ((sc.H + 15) // 16) <= 256 else 2   # partition passes the tile ids alone need
    _, ref_radii, _, S = util.run_oracle(sc, mode, backward=False)
    from das3r_amd import GaussianRasterizationSettings
    from das3r_amd.rasterizer import _forward_impl
    dev = _dev()
    kw = {k: v.to(dev) for k, v in util.raster_inputs(sc, mode).items()}
    skw = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in util.settings_kwargs(sc, mode).items()}
    e = torch.empty(0, device=dev)
    I, color, radii, geom, binning, img = _forward_impl(
        GaussianRasterizationSettings(**skw), kw["means3D"], kw.get("shs", e), kw.get("colors_precomp", e), kw["opacities"],
        kw.get("scales", e), kw.get("rotations", e), kw.get("cov3D_precomp", e))
    torch.cuda.synchronize()
    _lib.profile_enable(False)
    kernels = _lib.profile_report()
    ran = lambda prefix: any(k.startswith(prefix) for k in kernels)
    assert ran("depth_hist") == (binning_path == "radix"), kernels   # the local order and the segmented path skip the global depth sort
    assert ran("segment_sort") == binning_path.startswith("seg"), kernels
    assert kernels["onesweep_pass_kernel"][0] == {"radix": 4 + L0_passes, "local": L0_passes, "seg": L0_passes, "seg3": L0_passes + 1}[binning_path], kernels
    assert np.array_equal(radii.cpu().numpy(), ref_radii)
    P, npix = sc.P, sc.W * sc.H
    L = _lib.layout(P, I, sc.W, sc.H)
    vis = ref_radii > 0
    tt = _view(geom, L["tiles_touched"], torch.int32, P).cpu().numpy().astype(np.uint32)
    assert np.array_equal(tt, S["tiles_touched"])
    xy = _lib.splat_field(geom, L, "xy", P).cpu().numpy()[:, :2]
    co = _lib.splat_field(geom, L, "conic_opacity", P).cpu().numpy()
    rgbd = _lib.splat_field(geom, L, "rgbd", P).cpu().numpy()
    assert np.array_equal(xy[vis], S["xy"][vis]), "pixel centres must be bit-exact (no FMA contraction in K1)"
    assert np.array_equal(co[vis], S["conic_opacity"][vis]), "conics must be bit-exact"
    assert np.array_equal(rgbd[vis, 3], S["depths"][vis])
    if not mode["colors_precomp"]:
        assert np.array_equal(rgbd[vis, :3], S["rgb"][vis]), "SH colours must be bit-exact"
    assert I == S["num_rendered"]
    if I:
        pl = _view(binning, L["point_list"], torch.int32, I).cpu().numpy().astype(np.uint32)
        assert np.array_equal(pl, S["point_list"]), "per-tile (depth, index) order must match exactly"
    tiles = S["ranges"].shape[0]
    rg = _view(img, L["ranges"], torch.int32, 2 * tiles).cpu().numpy().reshape(tiles, 2).astype(np.uint32)
    assert np.array_equal(rg, S["ranges"])
    nc = _view(img, L["n_contrib"], torch.int32, npix).cpu().numpy().reshape(sc.H, sc.W).astype(np.uint32)
    assert (nc != S["n_contrib"]).mean() <= util.FLIP_FRACTION


def test_empty_scene_returns_zero_image():
    """P == 0: zero image, background NOT applied, empty radii (upstream:rasterize_points.cu behaviour)."""
    from das3r_amd import GaussianRasterizationSettings, GaussianRasterizer
    dev = _dev()
    sc, mode = util.scene_variant("deg0")
    skw = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in sc.settings_kwargs().items()}
    skw["bg"] = torch.tensor([0.3, 0.4, 0.5], device=dev)
    rast = GaussianRasterizer(GaussianRasterizationSettings(**skw))
    z = lambda *s: torch.zeros(*s, device=dev)
    color, radii = rast(means3D=z(0, 3), means2D=z(0, 3), opacities=z(0, 1), shs=z(0, 16, 3), scales=z(0, 3), rotations=z(0, 4))
    assert color.shape == (3, sc.H, sc.W) and float(color.abs().max()) == 0.0 and radii.numel() == 0


def test_all_culled_gives_background():
    from das3r_amd import GaussianRasterizationSettings, GaussianRasterizer
    dev = _dev()
    sc, mode = util.scene_variant("deg0")
    sc.means3D[:, 2] = -1.0
    ref_color, ref_radii, _, S = util.run_oracle(sc, mode, backward=False)
    assert S["num_rendered"] == 0
    color, radii, g, fn = _run_hip(sc, mode)
    assert fn.num_rendered == 0 and int(radii.abs().max()) == 0
    assert np.array_equal(color.cpu().numpy(), ref_color)
    for k, t in g.items():
        assert float(t.abs().max()) == 0.0, k


def test_retain_graph_double_backward_call():
    """DAS3R calls loss.backward(retain_graph=True) (/root/reference/train_gui.py:579): the saved buffers must survive."""
    from das3r_amd import GaussianRasterizationSettings, GaussianRasterizer
    dev = _dev()
    sc, mode = util.scene_variant("basic_deg3")
    scd = sc.to(dev)
    rs = GaussianRasterizationSettings(**scd.settings_kwargs())
    m3 = scd.means3D.clone().requires_grad_()
    m2 = torch.zeros(sc.P, 3, device=dev, requires_grad=True)
    color, _ = GaussianRasterizer(rs)(means3D=m3, means2D=m2, opacities=scd.opacities, shs=scd.shs, scales=scd.scales,
                                      rotations=scd.rotations)
    loss = (color * scd.dL_dpix).sum()
    loss.backward(retain_graph=True)
    g1 = m3.grad.clone()
    m3.grad = None
    loss.backward()
    assert torch.allclose(g1, m3.grad, rtol=1e-4, atol=1e-9)


def test_mark_visible():
    from das3r_amd import GaussianRasterizationSettings, GaussianRasterizer
    from oracle import c_oracle
    dev = _dev()
    sc, _ = util.scene_variant("culled")
    scd = sc.to(dev)
    rast = GaussianRasterizer(GaussianRasterizationSettings(**scd.settings_kwargs()))
    got = rast.markVisible(scd.means3D).cpu().numpy()
    ref = c_oracle.mark_visible(sc.means3D.numpy(), sc.viewmatrix.numpy(), sc.projmatrix.numpy())
    assert got.dtype == bool and np.array_equal(got, ref)


def test_prefiltered_is_honoured():
    """GaussianRasterizationSettings.prefiltered: the caller's promise that no Gaussian fails the near-plane cull.
    Kept, the image is the one of prefiltered=False bit for bit (twice: exactly sized and speculative layout); broken, upstream prints
    "Point is filtered although prefiltered is set. This shouldn't happen!" and traps (upstream:auxiliary.h in_frustum) — here the
    forward raises with that message, and the next forward of the thread works."""
    from das3r_amd import GaussianRasterizationSettings, GaussianRasterizer
    dev = _dev()

    def render(sc, prefiltered):
        scd = sc.to(dev)
        kw = scd.settings_kwargs()
        kw["prefiltered"] = prefiltered
        with torch.no_grad():
            return GaussianRasterizer(GaussianRasterizationSettings(**kw))(
                means3D=scd.means3D, means2D=torch.zeros_like(scd.means3D), opacities=scd.opacities, shs=scd.shs, scales=scd.scales,
                rotations=scd.rotations)

    sc, _ = util.scene_variant("basic_deg3")            # every splat in front of the near plane
    assert float(sc.means3D[:, 2].min()) > 0.5
    ref, ref_radii = render(sc, False)
    for _ in range(2):
        c, r = render(sc, True)
        assert torch.equal(c, ref) and torch.equal(r, ref_radii)
    bad, _ = util.scene_variant("culled")               # 200 splats behind / at the near plane
    for _ in range(2):
        with pytest.raises(RuntimeError, match="Point is filtered although prefiltered is set"):
            render(bad, True)
    c, _ = render(bad, False)
    assert bool(torch.isfinite(c).all())
    c, r = render(sc, True)
    assert torch.equal(c, ref) and torch.equal(r, ref_radii)


def test_reduction_modes_agree(monkeypatch):
    """The DPP wave reduction of the backward kernel against its ds_bpermute reference reduction."""
    sc, mode = util.scene_variant("long_lists")
    monkeypatch.setenv("DAS3R_BWD_REDUCE", "shfl")
    _, _, g_ref, _ = _run_hip(sc, mode)
    monkeypatch.setenv("DAS3R_BWD_REDUCE", "dpp")
    _, _, g, _ = _run_hip(sc, mode)
    for k in g:
        util.assert_grad_close(g[k].cpu().numpy(), g_ref[k].cpu().numpy(), f"dpp vs shfl {k}", tol=1e-4)


def test_finite_difference_directional():
    """Directional derivative of the HIP forward (central differences) against the HIP backward, on a scene whose discrete
    decisions are disabled by construction: a few huge, faint splats whose alpha stays above 1/255 over the
    whole image (no alpha cut-off crossings, tile rect = whole grid whatever the radius ceil does, T never reaches 1e-4,
    means well inside the EWA clamp).  Geometry gradients through thresholds are covered by oracle parity + the float64
    finite-difference test of the oracle (tests/test_oracle.py)."""
    from das3r_amd import GaussianRasterizationSettings, GaussianRasterizer
    from das3r_amd.synth import make_scene
    dev = _dev()
    sc = make_scene(P=16, W=64, H=48, focal=60.0, sh_degree=2, seed=77, s_px=(100.0, 140.0), opacity=0.1)
    sc.means3D[:, 0] *= 0.5
    sc.means3D[:, 1] *= 0.5
    sc = sc.to(dev)
    rs = GaussianRasterizationSettings(**sc.settings_kwargs())
    rast = GaussianRasterizer(rs)
    w = torch.randn(3, sc.H, sc.W, device=dev, generator=torch.Generator(device=dev).manual_seed(1)).abs()
    names = ["means3D", "opacities", "shs", "scales", "rotations"]
    base = {k: getattr(sc, k).double() for k in names}

    def f(vals):
        kw = {k: v.float().contiguous() for k, v in vals.items()}
        c, _ = rast(means2D=torch.zeros(sc.P, 3, device=dev), **kw)
        return float((c.double() * w.double()).sum())

    leaves = {k: v.float().clone().requires_grad_() for k, v in base.items()}
    c, radii = rast(means2D=torch.zeros(sc.P, 3, device=dev, requires_grad=True), **leaves)
    assert int((radii > 0).sum()) == sc.P
    (c * w).sum().backward()
    g = torch.Generator(device=dev).manual_seed(5)
    for k in names:
        d = torch.randn(base[k].shape, device=dev, generator=g, dtype=torch.float64)
        d = d / d.norm() * base[k].norm() * 2e-3
        plus, minus = dict(base), dict(base)
        plus[k] = base[k] + d
        minus[k] = base[k] - d
        fd = (f(plus) - f(minus)) / 2.0
        an = float((leaves[k].grad.double() * d).sum())
        assert abs(fd - an) <= 2e-2 * max(abs(fd), abs(an)) + 2e-4, f"{k}: finite difference {fd:.6e} vs analytic {an:.6e}"


def test_repeated_forwards_agree_and_count_is_exact():
    """num_rendered reaches the host from the first kernel of the forward (a sum of tiles_touched reduced with one atomic per
    workgroup and a self re-arming arrival word): it must equal the device-side scan total on every call, the binning buffer
    is sized by it, and back-to-back forwards (which reuse the mailbox and the arrival ring) give identical results."""
    from das3r_amd import GaussianRasterizationSettings, rasterizer, _lib
    dev = _dev()
    sc, mode = util.scene_variant("long_lists")
    scd = sc.to(dev)
    rs = GaussianRasterizationSettings(**scd.settings_kwargs())
    e = torch.empty(0, device=dev)
    args = (rs, scd.means3D, scd.shs, e, scd.opacities, scd.scales, scd.rotations, e)
    I0, c0, r0, g0, b0, i0, cap0 = rasterizer._forward_full(*args)
    assert cap0 >= I0 and I0 > 10000
    L = _lib.layout(sc.P, cap0, sc.W, sc.H)
    tt = _view(g0, L["tiles_touched"], torch.int32, sc.P)
    assert int(tt.sum()) == I0
    grads0 = rasterizer._backward_impl(rs, I0, scd.dL_dpix, scd.means3D, scd.shs, e, scd.opacities, scd.scales, scd.rotations, e, g0, b0, i0, cap0)
    pl0 = _view(b0, L["point_list"], torch.int32, I0).clone()
    for _ in range(70):   # more calls than the arrival ring has slots
        I1, c1, r1, g1, b1, i1, cap1 = rasterizer._forward_full(*args)
        assert I1 == I0 and cap1 >= I0   # (repeated shape: the binning buffer is laid out speculatively, with headroom)
    # the speculative forwards (scan + emission fused into the preprocess kernel when the grid is resident) build the same lists
    L1 = _lib.layout(sc.P, cap1, sc.W, sc.H)
    assert torch.equal(_view(b1, L1["point_list"], torch.int32, I1), pl0)
    rg0 = _view(i0, L["ranges"], torch.int32, 2 * ((sc.W + 15) // 16) * ((sc.H + 15) // 16))
    rg1 = _view(i1, L1["ranges"], torch.int32, rg0.numel())
    assert torch.equal(rg0, rg1)
    assert torch.equal(c1, c0) and torch.equal(r1, r0)
    grads1 = rasterizer._backward_impl(rs, I1, scd.dL_dpix, scd.means3D, scd.shs, e, scd.opacities, scd.scales, scd.rotations, e, g1, b1, i1, cap1)
    for k, (a, b) in enumerate(zip(grads0, grads1)):   # only the order of the four per-wave LDS adds differs run to run
        if a is None:   # dL_dcolors_precomp / dL_dcov3D are not produced in SH + scale/rotation mode
            continue
        util.assert_grad_close(a.cpu().numpy(), b.cpu().numpy(), "repeated forward", tol=1e-5)


@pytest.mark.parametrize("name", ["basic_deg3", "long_lists", "deep", "culled"])
def test_tight_rect_is_exact(name, monkeypatch):
    """Binning over the opacity-aware clipped rectangle only drops (tile, splat) instances whose alpha is < 1/255 on every
    pixel of the tile: the image must be bit-identical to binning over upstream's square, gradients equal up to the order of
    LDS adds, and the instance count can only shrink."""
    sc, mode = util.scene_variant(name)
    # (bit-identity of the IMAGE across two different lists needs a forward kernel that adds a pixel's colour terms one by one: the
    #  four-lanes kernel groups them by fours, and a dropped entry shifts the groups — same terms, last-bit differences)
    monkeypatch.setenv("DAS3R_RENDER", "rows")
    monkeypatch.setenv("DAS3R_RECT", "upstream")
    c_up, r_up, g_up, fn_up = _run_hip(sc, mode)
    n_up = fn_up.num_rendered
    monkeypatch.delenv("DAS3R_RECT")
    c_t, r_t, g_t, fn_t = _run_hip(sc, mode)
    assert torch.equal(c_t, c_up) and torch.equal(r_t, r_up)
    assert fn_t.num_rendered <= n_up
    for k in g_up:
        util.assert_grad_close(g_t[k].cpu().numpy(), g_up[k].cpu().numpy(), f"tight vs upstream rect dL/d{k}", tol=1e-5)


@pytest.mark.parametrize("binning", ["radix", "seg"])
@pytest.mark.parametrize("name", ["basic_deg3", "long_lists", "deep", "culled", "ragged_image", "depth_ties"])
def test_forward_kernels_agree_bit_for_bit(name, binning, monkeypatch):
    """The forward compositing kernels — sub-list per 8x8 quadrant, per 4x4 block (render_rows.hip), four lanes per pixel
    (render_lanes.hip) — visit every pixel's splats in the same order with the same arithmetic for alpha and T: radii, final_T and
    n_contrib must be identical; so must the image of the first two, while the four-lanes kernel adds a pixel's colour terms in four
    partial sums (LANES_COLOR_TOL); and the gradients computed from their saved state agree (up to the order of the per-wave LDS adds)
    — on lists from the global sort and from the segmented path."""
    from das3r_amd import GaussianRasterizationSettings, _lib, rasterizer
    sc, mode = util.scene_variant(name)
    monkeypatch.setenv("DAS3R_BINNING", binning)
    dev = _dev()
    out, state = {}, {}
    for kind in ("quad", "rows", "lanes"):
        monkeypatch.setenv("DAS3R_RENDER", kind)
        _lib.profile_report()
        _lib.profile_enable(True)
        c, r, g, fn = _run_hip(sc, mode)
        _lib.profile_enable(False)
        ran = _lib.profile_report()
        assert any(k.startswith({"quad": "render_forward_kernel", "rows": "render_forward_rows", "lanes": "render_forward_lanes"}[kind]) for k in ran), (kind, list(ran))
        out[kind] = (c, r, g, fn.num_rendered)
        # the saved per-pixel state itself
        kw = {k: v.to(dev) for k, v in util.raster_inputs(sc, mode).items()}
        skw = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in util.settings_kwargs(sc, mode).items()}
        rs = GaussianRasterizationSettings(**skw)
        e = torch.empty(0, device=dev)
        I, _, _, _, _, img, _ = rasterizer._forward_full(rs, kw["means3D"], kw.get("shs", e), kw.get("colors_precomp", e), kw["opacities"],
                                                         kw.get("scales", e), kw.get("rotations", e), kw.get("cov3D_precomp", e), exact=True)
        torch.cuda.synchronize()
        L = _lib.layout(sc.P, I, sc.W, sc.H)
        npix = sc.W * sc.H
        state[kind] = (img[L["final_T"]:L["final_T"] + 4 * npix].clone(), img[L["n_contrib"]:L["n_contrib"] + 4 * npix].clone())
    cq, rq, gq, nq = out["quad"]
    for kind in ("rows", "lanes"):
        cr, rr, gr, nr = out[kind]
        assert nq == nr and torch.equal(rq, rr), kind
        assert torch.equal(state["quad"][0], state[kind][0]), f"{kind}: final_T"
        assert torch.equal(state["quad"][1], state[kind][1]), f"{kind}: n_contrib"
        if kind == "lanes":
            assert (cq - cr).abs().max().item() <= LANES_COLOR_TOL, (kind, (cq - cr).abs().max().item())
        else:
            assert torch.equal(cq, cr), kind
        for k in gq:   # the backward kernel consumes the forward's final_T / n_contrib / checkpoints: any difference there shows up here
            util.assert_grad_close(gr[k].cpu().numpy(), gq[k].cpu().numpy(), f"{kind} vs quad forward dL/d{k}", tol=1e-5)


@pytest.mark.parametrize("binning", ["radix", "seg"])
@pytest.mark.parametrize("name", ["basic_deg3", "long_lists", "deep", "culled", "ragged_image", "depth_ties", "deg0"])
@pytest.mark.parametrize("kernel", ["slices", "fine"])
def test_sliced_forward_against_the_four_lanes_kernel_and_the_oracle(name, binning, kernel, monkeypatch):
    """render_slices.hip (round 6): a block's list of a batch cut into chunks that any wave walks from T = 1, the block's owner composing
    them in list order and walking exactly the chunk in which a pixel may stop.  Every stop is decided by the exact walk, so n_contrib is
    the four-lanes kernel's except where a pixel's T came within rounding of 1e-4; final_T differs by the rounding of T (prod) against the
    running product (<= 1e-5 relative measured bar), the colour by the order of its sum; the gradients computed from its saved state
    agree with the four-lanes kernel's to 1e-5 of the tensor maximum and with the oracle's within the parity bars."""
    from das3r_amd import GaussianRasterizationSettings, _lib, rasterizer
    sc, mode = util.scene_variant(name)
    monkeypatch.setenv("DAS3R_BINNING", binning)
    dev = _dev()
    out, state = {}, {}
    for kind in ("lanes", kernel):
        monkeypatch.setenv("DAS3R_RENDER", kind)
        _lib.profile_report()
        _lib.profile_enable(True)
        c, r, g, fn = _run_hip(sc, mode)
        _lib.profile_enable(False)
        ran = _lib.profile_report()
        assert any(k.startswith("render_forward_" + {"fine": "regions"}.get(kind, kind)) for k in ran), (kind, list(ran))
        out[kind] = (c, r, g, fn.num_rendered)
        kw = {k: v.to(dev) for k, v in util.raster_inputs(sc, mode).items()}
        skw = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in util.settings_kwargs(sc, mode).items()}
        rs = GaussianRasterizationSettings(**skw)
        e = torch.empty(0, device=dev)
        I, _, _, _, _, img, _ = rasterizer._forward_full(rs, kw["means3D"], kw.get("shs", e), kw.get("colors_precomp", e), kw["opacities"],
                                                         kw.get("scales", e), kw.get("rotations", e), kw.get("cov3D_precomp", e), exact=True)
        torch.cuda.synchronize()
        L = _lib.layout(sc.P, I, sc.W, sc.H)
        npix = sc.W * sc.H
        state[kind] = (img[L["final_T"]:L["final_T"] + 4 * npix].view(torch.float32).clone(), img[L["n_contrib"]:L["n_contrib"] + 4 * npix].view(torch.int32).clone())
    (cl, rl, gl, nl), (cs, rs_, gs, ns) = out["lanes"], out[kernel]
    assert nl == ns and torch.equal(rl, rs_)
    same = state["lanes"][1] == state[kernel][1]
    assert float((~same).float().mean()) <= util.FLIP_FRACTION, "n_contrib: the stops are the exact walk's"
    tl, ts = state["lanes"][0][same], state[kernel][0][same]
    assert float(((tl - ts).abs() / tl.abs().clamp_min(1e-30)).max()) <= 1e-5, "final_T: the rounding of T x (product), nothing else"
    util.assert_color_close(cs.cpu().numpy(), cl.cpu().numpy(), f"{name} slices vs lanes colour")
    for k in gl:
        util.assert_grad_close(gs[k].cpu().numpy(), gl[k].cpu().numpy(), f"slices vs lanes forward dL/d{k}", tol=1e-5)
    ref_color, ref_radii, ref_g, S = util.run_oracle(sc, mode)
    util.assert_color_close(cs.cpu().numpy(), ref_color, f"{name} slices colour vs oracle")
    for k, t in gs.items():
        util.assert_grad_close(t.cpu().numpy(), ref_g[k], f"{name} slices dL/d{k}")
        util.assert_grad_elementwise(t.cpu().numpy(), ref_g[k], f"{name} slices dL/d{k}")


def test_scratch_alignment_does_not_matter(monkeypatch):
    """The per-Gaussian backward reads a workgroup's run of partial rows as 16-byte words when the caller's scratch buffer is 16-byte
    aligned and as 4-byte words otherwise (preprocess_bwd.hip): same rows, same order of the sums — the gradients are the same bits
    (deterministic backward, so that the comparison can be exact)."""
    from das3r_amd import GaussianRasterizationSettings, rasterizer
    monkeypatch.setenv("DAS3R_DETERMINISTIC", "1")
    for name in ("basic_deg3", "long_lists", "ragged_image"):
        sc, mode = util.scene_variant(name)
        dev = _dev()
        kw = {k: v.to(dev) for k, v in util.raster_inputs(sc, mode).items()}
        skw = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in util.settings_kwargs(sc, mode).items()}
        rs = GaussianRasterizationSettings(**skw)
        e = torch.empty(0, device=dev)
        args = (rs, kw["means3D"], kw["shs"], e, kw["opacities"], kw["scales"], kw["rotations"], e)
        I, c, r, g, b, i, cap = rasterizer._forward_full(*args, exact=True)
        dL = sc.dL_dpix.to(dev)
        ref = rasterizer._backward_impl(rs, I, dL, *args[1:], g, b, i, cap)
        for off in (4, 8, 12):
            got = rasterizer._backward_impl(rs, I, dL, *args[1:], g, b, i, cap, _scratch_misalign=off)
            for x, y in zip(ref, got):
                assert (x is None) == (y is None)
                if x is not None:
                    assert torch.equal(x, y), (name, off)


@pytest.mark.parametrize("chunk", ["0", "1", "8", "64"])
def test_tile_order_does_not_change_results(chunk, monkeypatch):
    """The block -> tile map of the compositing kernels (contiguous eighths per XCD / chunks dealt round robin: render_common.h
    xcd_tile, CPU model in tests/test_tile_map.py) decides WHEN a tile is composited, nothing else: image, radii and every
    gradient are the same bits as with the default map (short lists: quad + dpp kernels; long lists: rows + block walk)."""
    for name in ("ragged_image", "long_lists", "deep"):
        sc, mode = util.scene_variant(name)
        monkeypatch.setenv("DAS3R_DETERMINISTIC", "1")   # (the dpp backward's LDS float atomics are order-dependent on their own)
        monkeypatch.delenv("DAS3R_TILE_CHUNK", raising=False)
        c0, r0, g0, _ = _run_hip(sc, mode)
        monkeypatch.setenv("DAS3R_TILE_CHUNK", chunk)
        c1, r1, g1, _ = _run_hip(sc, mode)
        assert torch.equal(c0, c1) and torch.equal(r0, r1), (name, chunk)
        for k in g0:
            assert torch.equal(g0[k], g1[k]), (name, chunk, k)


@pytest.mark.parametrize("pattern", [0x7FC00000, 0xFFFFFFFF, 0x7F800000, 0x7FA00000, 0x00000000, 0x01010101])
def test_unwritten_lds_does_not_reach_the_image(pattern):
    """The compositing kernels leave parts of their LDS arrays unwritten (list tails, staged entries past a tile's list) and the
    row walk reads a staged entry for every position of a trip, also past the end of a row's list: whatever the LDS held before
    must not matter (r3: 0 x NaN from an uninitialised staged colour turned a few tiles into garbage, once in ~15 fresh
    processes; a stale list byte naming an EARLIER entry pulled a pixel's last contributor back, which only small bytes show:
    hence the zero and 0x01 patterns).  Every CU's LDS is filled with quiet NaNs / all-ones / +inf / signalling NaNs / zeros, then a scene whose second forward
    takes the speculative path (rows kernel on short lists: deg1 and deg2 share P, W, H) is rendered and held to the oracle."""
    from das3r_amd import _lib
    for name in ("deg1", "deg2", "ragged_image", "long_lists", "deep", "single"):
        sc, mode = util.scene_variant(name)
        ref_color, ref_radii, ref_g, S = util.run_oracle(sc, mode)
        for rep in range(3):
            _lib.poison_lds(pattern)
            color, radii, g, fn = _run_hip(sc, mode)
            util.assert_color_close(color.cpu().numpy(), ref_color, f"{name} after LDS pattern {pattern:#x}, forward {rep}")
            for k, t in g.items():
                assert bool(torch.isfinite(t).all()), (name, k)
                util.assert_grad_close(t.cpu().numpy(), ref_g[k], f"{name} after LDS pattern {pattern:#x} dL/d{k}")


def test_deterministic_switch_gives_bit_identical_gradients(monkeypatch):
    """a bit-deterministic mode for tests.  DAS3R_DETERMINISTIC=1 takes the block-walk backward for every
    list length (its sums have a fixed order; the pixel-per-lane kernel of short lists meets its four waves with LDS float atomics):
    two runs of the same scene give the same bits in every gradient — short lists (the default would be the dpp kernel), long
    lists and the bucket-parallel replay alike."""
    monkeypatch.setenv("DAS3R_DETERMINISTIC", "1")   # (conftest.py: the library re-reads its switches)
    for name in ("basic_deg3", "long_lists", "deep"):
        sc, mode = util.scene_variant(name)
        _, _, g0, _ = _run_hip(sc, mode)
        for _ in range(3):
            _, _, g1, _ = _run_hip(sc, mode)
            for k in g0:
                assert torch.equal(g0[k], g1[k]), (name, k)
        _, _, ref_g, _ = util.run_oracle(sc, mode)
        for k, t in g0.items():
            util.assert_grad_close(t.cpu().numpy(), ref_g[k], f"{name} deterministic dL/d{k}")


SCAN_KINDS = ("scan64", "scan128", "scan256", "scana256",   # render_bwd_scan.hip: entries per round, private / atomic flush
              "blk64", "blk128", "blk256",                  # render_bwd_blk.hip: 4x4 block per DPP row, entries per round
              "fine64", "fine128", "fine192")               # render_bwd_rgn.hip (round 6): 2x2 region per DPP row, entries per round
EXPERIMENT_KINDS = ("mfma", "stream")                        # superseded kernels: only in `make EXPERIMENTS=1` builds of the library


def _built_kinds(kinds):
    from das3r_amd import _lib
    return tuple(k for k in kinds if k not in EXPERIMENT_KINDS or _lib.has_experiments())



@pytest.mark.parametrize("name", ["basic_deg3", "long_lists", "deep", "culled", "ragged_image", "deg1", "single"])
def test_backward_kernels_agree(name, monkeypatch):
    """The backward compositing kernels against each other: pixel per lane with the cross-lane DPP reduction of the nine per-pair
    sums (render_bwd.hip), the same with the moments reduced on the matrix cores through an LDS slab (render_bwd_mfma.hip), and
    lanes = 4 pixels x 16 splats with DPP row scans for the recurrences and split-bf16 sums on the matrix cores
    (render_bwd_scan.hip: 64 .. 512 list entries per round, wave-private or atomically shared accumulators).  Each
    is also checked against the oracle when selected (the default one in every other test; DAS3R_RENDER_BWD=... python -m pytest
    tests -m gpu for the others).  fp32 tolerance: the matrix-core kernels sum moments about the tile centre."""
    sc, mode = util.scene_variant(name)
    out = {}
    others = _built_kinds(EXPERIMENT_KINDS + SCAN_KINDS)
    for kind in ("dpp",) + others:
        monkeypatch.setenv("DAS3R_RENDER_BWD", kind)
        c, r, g, fn = _run_hip(sc, mode)
        out[kind] = (c, g)
    for kind in others:
        assert torch.equal(out["dpp"][0], out[kind][0])
        for k in out["dpp"][1]:
            util.assert_grad_close(out[kind][1][k].cpu().numpy(), out["dpp"][1][k].cpu().numpy(), f"{kind} vs dpp backward dL/d{k}", tol=5e-5)


@pytest.mark.parametrize("kind", ["dpp", "mfma", "scan64", "scan128", "scan256", "scana256", "stream", "blk64", "blk128", "blk256", "fine64", "fine128", "fine160", "fine192", "fine256", "fine128q", "fine128s"])
@pytest.mark.parametrize("name", ["basic_deg3", "long_lists", "deep", "ragged_image", "culled"])
def test_every_backward_kernel_vs_oracle(name, kind, monkeypatch):
    """Each backward compositing kernel on its own against the CPU oracle (the default one is covered on all variants above)."""
    if not _built_kinds((kind,)):
        pytest.skip("superseded kernel: built with `make EXPERIMENTS=1` only")
    monkeypatch.setenv("DAS3R_RENDER_BWD", kind)
    sc, mode = util.scene_variant(name)
    _, _, ref_g, _ = util.run_oracle(sc, mode)
    _, _, g, _ = _run_hip(sc, mode)
    gmap = {"means3D": "means3D", "opacities": "opacities", "shs": "shs", "scales": "scales", "rotations": "rotations", "means2D": "means2D"}
    for k, t in g.items():
        bars = util.tolerances_for(kind)   # (tight for the shipped kernels, the split-bf16 kernels' own for scan / mfma / stream)
        util.assert_grad_close(t.cpu().numpy(), ref_g[gmap[k]], f"{name} [{kind}] dL/d{k}", tol=bars["tol"])
        util.assert_grad_elementwise(t.cpu().numpy(), ref_g[gmap[k]], f"{name} [{kind}] dL/d{k}", rtol=bars["rtol"], floor=bars["floor"], outliers=bars["outliers"])


@pytest.mark.parametrize("kind", ["scan128", "scan256", "blk128", "blk256", "fine128", "fine192"])
@pytest.mark.parametrize("slices", [2, 5])
@pytest.mark.parametrize("name", ["deep", "long_lists", "basic_deg3"])
def test_bucket_parallel_backward(name, kind, slices, monkeypatch):
    """Long tile lists replayed bucket by bucket in parallel workgroups, each from the pixel states the forward checkpointed at
    the bucket boundaries (common.h BUCKET = 1024 list positions; "deep" has ~4000 entries per tile, "long_lists" ~1500 and
    saturating pixels, "basic_deg3" none that long: every tile is its own last bucket) — against the oracle and against the
    sequential replay."""
    sc, mode = util.scene_variant(name)
    _, _, ref_g, S = util.run_oracle(sc, mode)
    monkeypatch.setenv("DAS3R_RENDER_BWD", kind)
    monkeypatch.setenv("DAS3R_BWD_BUCKETS", "0")
    _, _, g_seq, fn = _run_hip(sc, mode)
    monkeypatch.setenv("DAS3R_BWD_BUCKETS", str(slices))
    _, _, g, _ = _run_hip(sc, mode)
    if name != "basic_deg3":
        tiles = ((sc.W + 15) // 16) * ((sc.H + 15) // 16)
        assert fn.num_rendered > 1024 * tiles, "the scene is meant to have lists of several buckets"
    gmap = {"means3D": "means3D", "opacities": "opacities", "shs": "shs", "scales": "scales", "rotations": "rotations", "means2D": "means2D"}
    for k, t in g.items():
        bars = util.tolerances_for(kind)
        util.assert_grad_close(t.cpu().numpy(), ref_g[gmap[k]], f"{name} [{kind}, {slices} slices] dL/d{k}", tol=bars["tol"])
        util.assert_grad_elementwise(t.cpu().numpy(), ref_g[gmap[k]], f"{name} [{kind}, {slices} slices] dL/d{k}", rtol=bars["rtol"], floor=bars["floor"],
                                     outliers=bars["outliers"])
        util.assert_grad_close(t.cpu().numpy(), g_seq[k].cpu().numpy(), f"{name} bucket-parallel vs sequential dL/d{k}", tol=5e-5)


def test_failed_binning_is_reported_before_the_backward_pass(monkeypatch):
    """A forward whose binning kernels fail their self-check (here: a look-back timeout forced into the word the last binning
    kernel hands to the host, das3r_debug_inject_fault) must not get as far as a parameter update: the backward pass examines the
    forward's word before it launches anything and raises; `check_forward` does the same for callers without a backward; a debug
    forward raises by itself.  The next clean forward works again."""
    from das3r_amd import GaussianRasterizationSettings, GaussianRasterizer
    from das3r_amd.rasterizer import _forward_full, check_forward
    dev = _dev()
    sc, mode = util.scene_variant("basic_deg3")
    scd = sc.to(dev)
    rs = GaussianRasterizationSettings(**scd.settings_kwargs())
    e = torch.empty(0, device=dev)

    def fwd_bwd(settings):
        m3 = scd.means3D.clone().requires_grad_()
        m2 = torch.zeros(sc.P, 3, device=dev, requires_grad=True)
        color, _ = GaussianRasterizer(settings)(means3D=m3, means2D=m2, opacities=scd.opacities, shs=scd.shs, scales=scd.scales,
                                                rotations=scd.rotations)
        (color * scd.dL_dpix).sum().backward()
        return m3.grad

    from das3r_amd import _lib
    good = fwd_bwd(rs)
    monkeypatch.setenv("DAS3R_INJECT_FAULT", "1")    # the shipped library does not read this (round 4): nothing may happen
    assert torch.allclose(fwd_bwd(rs), good, rtol=1e-4, atol=1e-9)
    monkeypatch.delenv("DAS3R_INJECT_FAULT")
    _lib.inject_fault(1)                             # ERR_TIMEOUT
    try:
        with pytest.raises(RuntimeError, match="self-check"):
            fwd_bwd(rs)                                  # the forward returns, the backward refuses
        out = _forward_full(rs, scd.means3D, scd.shs, e, scd.opacities, scd.scales, scd.rotations, e)
        with pytest.raises(RuntimeError, match="self-check"):
            check_forward(out[6], dev)                   # explicit check of a forward without a backward
        with pytest.raises(RuntimeError, match="self-check"):
            fwd_bwd(rs._replace(debug=True))             # debug: the forward itself waits for the word
    finally:
        _lib.inject_fault(0)
    for _ in range(20):                              # every slot of the self-check ring is examined and reused cleanly
        again = fwd_bwd(rs)
    assert torch.allclose(again, good, rtol=1e-4, atol=1e-9)


def _at_depths(sc, z_new):
    """The scene with every splat moved along its ray to depth z_new (camera at the origin looking down +z): same pixel, same
    footprint in pixels."""
    from das3r_amd.synth import Scene
    f = (z_new / sc.means3D[:, 2]).to(torch.float32)
    return Scene(**{**sc.__dict__, "means3D": (sc.means3D * f[:, None]).contiguous(), "scales": (sc.scales * f[:, None]).contiguous()})


def _lists_of(sc, mode, dev):
    """-> (I, point_list, ranges, kernels that ran) of one forward through the C-ABI."""
    from das3r_amd import GaussianRasterizationSettings, _lib
    from das3r_amd.rasterizer import _forward_impl
    _lib.profile_report()
    _lib.profile_enable(True)
    kw = {k: v.to(dev) for k, v in util.raster_inputs(sc, mode).items()}
    skw = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in util.settings_kwargs(sc, mode).items()}
    e = torch.empty(0, device=dev)
    I, color, radii, geom, binning, img = _forward_impl(
        GaussianRasterizationSettings(**skw), kw["means3D"], kw.get("shs", e), kw.get("colors_precomp", e), kw["opacities"],
        kw.get("scales", e), kw.get("rotations", e), kw.get("cov3D_precomp", e))
    torch.cuda.synchronize()
    _lib.profile_enable(False)
    kernels = _lib.profile_report()
    L = _lib.layout(sc.P, I, sc.W, sc.H)
    pl = _view(binning, L["point_list"], torch.int32, I).cpu().numpy().astype(np.uint32)
    tiles = ((sc.W + 15) // 16) * ((sc.H + 15) // 16)
    rg = _view(img, L["ranges"], torch.int32, 2 * tiles).cpu().numpy().reshape(tiles, 2).astype(np.uint32)
    return I, pl, rg, kernels, color


@pytest.mark.parametrize("passes", ["seg", "seg3"])
@pytest.mark.parametrize("case", ["one_wall", "three_layers", "thin_slab", "ties_and_spread", "one_tile_wall"])
def test_segmented_binning_long_and_degenerate_segments(case, passes, monkeypatch):
    """The segmented path (segkey.h, segsort.hip) where its depth buckets cannot help: every splat at ONE depth (a wall parallel to
    the image plane: each tile's whole list is one segment, ties broken by the index), three exact layers (segments of a third of
    a list: a few thousand entries, sorted by rank in LDS or, past the LDS span, by the global-memory network), a slab 1e-5 thick
    (distinct depths in one histogram bin), exact ties mixed into a spread, and 20 000 equal depths in a single tile (one segment
    of 20 000).  Lists, ranges and num_rendered bit-identical to the C oracle's in every case."""
    from das3r_amd.synth import make_scene
    g = torch.Generator().manual_seed(41)
    if case == "one_tile_wall":
        sc = make_scene(P=20000, W=16, H=16, focal=30.0, sh_degree=0, seed=51, s_px=(0.5, 2.0), opacity=0.02)
        sc = _at_depths(sc, torch.full((sc.P,), 4.0))
    else:
        sc = make_scene(P=9000, W=48, H=32, focal=50.0, sh_degree=0, seed=52, s_px=(0.5, 3.0), opacity=0.05)
        if case == "one_wall":
            z = torch.full((sc.P,), 4.0)
        elif case == "three_layers":
            z = torch.tensor([3.0, 4.0, 5.5])[torch.randint(0, 3, (sc.P,), generator=g)]
        elif case == "thin_slab":
            z = 4.0 + 1e-5 * torch.rand(sc.P, generator=g)
        else:
            z = torch.where(torch.rand(sc.P, generator=g) < 0.5, torch.tensor(2.5), 1.0 + 9.0 * torch.rand(sc.P, generator=g))
        sc = _at_depths(sc, z)
    mode = dict(colors_precomp=False, cov3D_precomp=False, scale_modifier=1.0)
    monkeypatch.setenv("DAS3R_RECT", "upstream")
    monkeypatch.setenv("DAS3R_BINNING", passes)
    ref_color, _, _, S = util.run_oracle(sc, mode, backward=False)
    I, pl, rg, kernels, color = _lists_of(sc, mode, _dev())
    assert any(k.startswith("segment_sort") for k in kernels) and not any(k.startswith("depth_hist") for k in kernels), kernels
    assert I == S["num_rendered"] and I > 4 * sc.P // 5
    assert np.array_equal(rg, S["ranges"])
    assert np.array_equal(pl, S["point_list"]), "per-tile (depth, index) order must match exactly"
    util.assert_color_close(color.cpu().numpy(), ref_color, case)


def test_segmented_binning_is_chosen_for_long_lists_and_backs_off(monkeypatch):
    """Unforced: the first forward of a shape has no history (global sort), the next ones with long lists take the segmented path;
    a segment that does not fit in LDS (here: 20 000 equal depths in one tile) is still sorted exactly and makes the shape take one
    more partition pass of bucket bits; when that does not help either (equal depths share every bucket) the following forwards go
    back to the global sort for a while.  Every forward's list is the oracle's."""
    from das3r_amd.synth import make_scene
    monkeypatch.delenv("DAS3R_BINNING", raising=False)
    monkeypatch.setenv("DAS3R_RECT", "upstream")
    mode = dict(colors_precomp=False, cov3D_precomp=False, scale_modifier=1.0)
    # (P differs from every other test's at this image size: the choice of path is remembered per (P, W, H) and thread)
    spread = make_scene(P=20011, W=16, H=16, focal=30.0, sh_degree=0, seed=53, s_px=(0.5, 2.0), opacity=0.02)
    wall = _at_depths(spread, torch.full((spread.P,), 4.0))
    dev = _dev()
    want = {id(sc): util.run_oracle(sc, mode, backward=False)[3]["point_list"] for sc in (spread, wall)}
    took = []
    for sc in (spread, spread, spread, wall, wall, spread):
        I, pl, _, kernels, _ = _lists_of(sc, mode, dev)
        assert np.array_equal(pl, want[id(sc)])
        took.append(("seg%d" % kernels["onesweep_pass_kernel"][0]) if any(k.startswith("segment_sort") for k in kernels)
                    else ("radix" if any(k.startswith("depth_hist") for k in kernels) else "local"))
    assert took[:3] == ["radix", "seg1", "seg1"], took    # history, then the segmented path (one tile: one partition pass)
    assert took[3:] == ["seg1", "seg2", "radix"], took    # the wall: too long -> one more pass of bucket bits -> still one segment -> global sort


@pytest.mark.parametrize("render", ["quad", "rows"])
@pytest.mark.parametrize("name", ["basic_deg3", "long_lists", "deep", "depth_ties", "ragged_image"])
def test_depth_orders_agree(name, render, monkeypatch):
    """Global depth sort vs the local depth order (tile lists sorted by the compositing kernel, api.hip): the lists are
    identical, so image, radii and num_rendered must be too, and the gradients up to the order of the per-wave LDS adds
    (the emission slots — rows of the backward pass's partial sums — are numbered in index instead of depth order)."""
    sc, mode = util.scene_variant(name)
    monkeypatch.setenv("DAS3R_RENDER", render)
    out = {}
    for kind in ("radix", "local", "seg"):
        monkeypatch.setenv("DAS3R_BINNING", kind)
        c, r, g, fn = _run_hip(sc, mode)
        out[kind] = (c, r, g, fn.num_rendered)
    ca, ra, ga, na = out["radix"]
    for kind in ("local", "seg"):   # (seg: the segmented path of round 4 — same lists, emission slots numbered in index order like local)
        cb, rb, gb, nb = out[kind]
        assert na == nb and torch.equal(ca, cb) and torch.equal(ra, rb), kind
        for k in ga:
            util.assert_grad_close(gb[k].cpu().numpy(), ga[k].cpu().numpy(), f"{kind} vs global depth order dL/d{k}", tol=1e-5)


def test_speculative_capacity_is_redone_when_the_scene_grows(monkeypatch):
    """A forward of a shape seen before lays the binning buffer out for the previous count + 25 % and enqueues everything
    before it knows its own count (api.hip).  When the scene has grown past that (here: the scale modifier goes from 0.15
    to 1), the binning and the compositing are redone with the exact size: image, radii, count and gradients must equal those
    of a forward that never speculated (DAS3R_CAPACITY=exact)."""
    from das3r_amd import GaussianRasterizationSettings, rasterizer
    dev = _dev()
    sc, mode = util.scene_variant("long_lists")
    scd = sc.to(dev)
    kw = scd.settings_kwargs()
    e = torch.empty(0, device=dev)
    monkeypatch.setenv("DAS3R_BINNING", "local")

    def run(mod):
        rs = GaussianRasterizationSettings(**{**kw, "scale_modifier": mod})
        I, c, r, g, b, i, cap = rasterizer._forward_full(rs, scd.means3D, scd.shs, e, scd.opacities, scd.scales, scd.rotations, e)
        grads = rasterizer._backward_impl(rs, I, scd.dL_dpix, scd.means3D, scd.shs, e, scd.opacities, scd.scales, scd.rotations, e, g, b, i, cap)
        return I, cap, c, r, grads

    other, _ = util.scene_variant("basic_deg3")      # another shape: the library forgets what it knew about this one
    od = other.to(dev)
    rasterizer._forward_full(GaussianRasterizationSettings(**od.settings_kwargs()), od.means3D, od.shs, e, od.opacities, od.scales,
                             od.rotations, e)
    I_small, cap_small, *_ = run(0.15)  # first forward of the shape: sized exactly
    assert cap_small == I_small
    I, cap, c, r, g = run(1.0)          # speculates with I_small's capacity, overflows, redoes
    assert (I_small + I_small // 4 + 4096) < I, "the test scene must outgrow the speculative capacity"
    assert cap == I
    I2, cap2, c2, r2, g2 = run(1.0)     # speculates with headroom, fits
    assert I2 == I and cap2 > I
    monkeypatch.setenv("DAS3R_CAPACITY", "exact")
    I_ref, cap_ref, c_ref, r_ref, g_ref = run(1.0)
    assert cap_ref == I_ref == I
    assert torch.equal(c, c_ref) and torch.equal(r, r_ref) and torch.equal(c2, c_ref)
    for k, (a, b, b2) in enumerate(zip(g_ref, g, g2)):
        if a is None:
            continue
        util.assert_grad_close(b.cpu().numpy(), a.cpu().numpy(), f"redone forward, grad {k}", tol=1e-5)
        util.assert_grad_close(b2.cpu().numpy(), a.cpu().numpy(), f"speculative forward, grad {k}", tol=1e-5)


def test_two_host_threads_render_concurrently():
    """The library's per-call state (host mailbox, arrival counters, the last shape's count) is thread_local: two host threads,
    each on its own stream and its own scene, interleave hundreds of forwards + backwards and must reproduce the images and
    gradients they get when they run alone."""
    import threading
    from das3r_amd import GaussianRasterizationSettings, rasterizer
    dev = _dev()
    jobs = []
    for name in ("basic_deg3", "long_lists"):
        sc, mode = util.scene_variant(name)
        scd = sc.to(dev)
        rs = GaussianRasterizationSettings(**scd.settings_kwargs())
        jobs.append((scd, rs))
    e = torch.empty(0, device=dev)

    def run(scd, rs):
        I, c, r, g, b, i, cap = rasterizer._forward_full(rs, scd.means3D, scd.shs, e, scd.opacities, scd.scales, scd.rotations, e)
        grads = rasterizer._backward_impl(rs, I, scd.dL_dpix, scd.means3D, scd.shs, e, scd.opacities, scd.scales, scd.rotations, e, g, b, i, cap)
        return I, c, grads

    ref = [run(*j) for j in jobs]
    torch.cuda.synchronize()
    errors = []

    def worker(k):
        try:
            stream = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(stream):
                for _ in range(150):
                    I, c, grads = run(*jobs[k])
                    if I != ref[k][0]:
                        errors.append((k, "count", I))
                        return
                stream.synchronize()
                if not torch.equal(c, ref[k][1]):
                    errors.append((k, "image"))
                for a, b in zip(grads, ref[k][2]):
                    if a is not None:
                        util.assert_grad_close(a.cpu().numpy(), b.cpu().numpy(), "concurrent render", tol=1e-5)
        except Exception as ex:   # noqa: BLE001 - reported below
            errors.append((k, repr(ex)))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


@pytest.mark.parametrize("P", [1, 2, 3, 63, 64, 65, 255, 256, 257, 300, 511, 512, 513, 1000, 1023, 1024, 1025, 1700])
def test_local_order_at_every_list_length(P, monkeypatch):
    """One 16x16 image = one tile whose list has exactly P entries (every splat visible, inside, many exact depth ties): the
    compositing kernel's own sort — rank sort up to 256 entries, LDS bitonic network up to 1024, global-memory network beyond —
    must produce the list of the global radix sort, bit for bit, at and around every switch-over."""
    from das3r_amd import GaussianRasterizationSettings, _lib
    from das3r_amd.rasterizer import _forward_impl
    from das3r_amd.synth import make_scene
    dev = _dev()
    sc = make_scene(P=P, W=16, H=16, focal=20.0, sh_degree=1, seed=100 + P, s_px=(1.0, 2.5), opacity=0.6)
    g = torch.Generator(device="cpu").manual_seed(P)
    z = torch.rand(P, generator=g) * 4.0 + 1.0
    z[::5] = 2.0                                   # exact depth ties: the order falls back to the splat index
    u = torch.rand(P, 2, generator=g) - 0.5        # well inside the image
    sc.means3D[:, 2] = z
    sc.means3D[:, 0] = z * sc.tanfovx * u[:, 0]
    sc.means3D[:, 1] = z * sc.tanfovy * u[:, 1]
    scd = sc.to(dev)
    rs = GaussianRasterizationSettings(**scd.settings_kwargs())
    e = torch.empty(0, device=dev)
    out = {}
    monkeypatch.setenv("DAS3R_RENDER", "rows")     # (one forward kernel for both: the four-lanes kernel, which a single long globally sorted list would get, adds colours in another order)
    for kind in ("radix", "local"):
        monkeypatch.setenv("DAS3R_BINNING", kind)
        I, color, radii, geom, binning, img = _forward_impl(rs, scd.means3D, scd.shs, e, scd.opacities, scd.scales, scd.rotations, e)
        torch.cuda.synchronize()
        L = _lib.layout(P, I, 16, 16)
        pl = _view(binning, L["point_list"], torch.int32, I).cpu().numpy()
        rg = _view(img, L["ranges"], torch.int32, 2).cpu().numpy()
        out[kind] = (I, pl, rg, color.cpu(), radii.cpu())
    (Ia, pla, rga, ca, ra), (Ib, plb, rgb_, cb, rb) = out["radix"], out["local"]
    assert Ia == Ib == P and tuple(rga) == tuple(rgb_) == (0, P)
    depth = scd.means3D[:, 2].cpu().numpy().view(np.uint32)
    expect = np.lexsort((np.arange(P), depth))     # (depth bits, index) order, computed independently
    assert np.array_equal(pla, expect), "global sort"
    assert np.array_equal(plb, expect), "local order"
    assert torch.equal(ca, cb) and torch.equal(ra, rb)


@pytest.mark.parametrize("fused", ["1", "0"])
def test_speculative_forward_builds_the_same_lists(fused, monkeypatch):
    """A short-list scene seen before is rendered without waiting for its count: binning buffer with headroom, the count
    delivered by the scan, and — when the preprocess grid is resident as a whole — scan and emission inside the preprocess
    kernel itself (DAS3R_FUSED_EMIT=0: the separate kernel).  Lists, ranges, image, radii and gradients must equal those of
    the exactly sized forward, call after call (the control words of the fused emission live in a ring that every forward has
    to leave zeroed)."""
    from das3r_amd import GaussianRasterizationSettings, rasterizer, _lib
    from das3r_amd.synth import make_scene
    monkeypatch.setenv("DAS3R_FUSED_EMIT", fused)
    dev = _dev()
    sc = make_scene(P=20000, W=320, H=192, focal=250.0, sh_degree=2, seed=77, s_px=(0.7, 3.0))
    scd = sc.to(dev)
    rs = GaussianRasterizationSettings(**scd.settings_kwargs())
    e = torch.empty(0, device=dev)
    args = (rs, scd.means3D, scd.shs, e, scd.opacities, scd.scales, scd.rotations, e)
    tiles = ((sc.W + 15) // 16) * ((sc.H + 15) // 16)
    I0, c0, r0, g0, b0, i0, cap0 = rasterizer._forward_full(*args, exact=True)
    assert cap0 == I0 and 0 < I0 <= 384 * tiles, "the scene must take the local depth order"
    L0 = _lib.layout(sc.P, cap0, sc.W, sc.H)
    pl0 = _view(b0, L0["point_list"], torch.int32, I0).clone()
    rg0 = _view(i0, L0["ranges"], torch.int32, 2 * tiles).clone()
    grads0 = rasterizer._backward_impl(rs, I0, scd.dL_dpix, *args[1:], g0, b0, i0, cap0)
    for it in range(40):
        I1, c1, r1, g1, b1, i1, cap1 = rasterizer._forward_full(*args)
        assert I1 == I0 and cap1 > I0, "speculative capacity expected"
        if it % 13 == 0 or it == 39:
            L1 = _lib.layout(sc.P, cap1, sc.W, sc.H)
            assert torch.equal(_view(b1, L1["point_list"], torch.int32, I1), pl0), it
            assert torch.equal(_view(i1, L1["ranges"], torch.int32, 2 * tiles), rg0), it
            assert torch.equal(c1, c0) and torch.equal(r1, r0), it
            tt = _view(g1, L1["tiles_touched"], torch.int32, sc.P)
            assert int(tt.sum()) == I1
    grads1 = rasterizer._backward_impl(rs, I1, scd.dL_dpix, *args[1:], g1, b1, i1, cap1)
    for a, b in zip(grads0, grads1):
        if a is not None:
            util.assert_grad_close(b.cpu().numpy(), a.cpu().numpy(), "speculative forward", tol=1e-5)


def test_vanilla_3dgs_renderer_end_to_end_vs_oracle():
    """das3r_render_3dgs (gaussian_renderer/__init__3dgs.py counterpart) with a camera built by world_camera (scene/cameras.py
    counterpart): image and gradients through the drop-in rasterizer against the C oracle fed with the same settings."""
    from types import SimpleNamespace
    from das3r_amd.camera import world_camera
    from das3r_amd.render import das3r_render_3dgs
    from oracle import c_oracle
    dev = _dev()
    g = torch.Generator().manual_seed(77)
    P, H, W = 900, 72, 104
    xyz = torch.randn(P, 3, generator=g) * 1.2
    rot_raw = torch.randn(P, 4, generator=g)
    log_s = torch.randn(P, 3, generator=g) * 0.4 - 2.2
    op_raw = torch.randn(P, 1, generator=g)
    feats = torch.randn(P, 16, 3, generator=g) * 0.2
    Rm = torch.linalg.qr(torch.randn(3, 3, generator=g))[0]
    if torch.linalg.det(Rm) < 0:
        Rm[:, 0] = -Rm[:, 0]
    cam = world_camera(Rm.numpy(), np.array([0.1, -0.2, 6.0]), 0.9, 0.7, H, W, device=dev)
    leaves = {k: v.to(dev).requires_grad_(True) for k, v in dict(xyz=xyz, rot=rot_raw, log_s=log_s, op=op_raw, feats=feats).items()}
    pc = SimpleNamespace(active_sh_degree=2, max_sh_degree=3, get_xyz=leaves["xyz"], get_opacity=torch.sigmoid(leaves["op"]),
                         get_scaling=torch.exp(leaves["log_s"]), get_rotation=torch.nn.functional.normalize(leaves["rot"]),
                         get_features=leaves["feats"])
    pipe = SimpleNamespace(debug=False, compute_cov3D_python=False, convert_SHs_python=False)
    bg = torch.tensor([0.1, 0.2, 0.3], device=dev)
    pkg = das3r_render_3dgs(cam, pc, pipe, bg)
    dL = torch.randn(3, H, W, generator=g) / float(H * W)
    pkg["render"].backward(dL.to(dev))
    o = c_oracle.RasterOracle(image_height=H, image_width=W, tanfovx=float(np.tan(0.45)), tanfovy=float(np.tan(0.35)), bg=bg.cpu().numpy(),
                              scale_modifier=1.0, viewmatrix=cam.world_view_transform.cpu().numpy(), projmatrix=cam.full_proj_transform.cpu().numpy(),
                              sh_degree=2, campos=cam.camera_center.cpu().numpy(), prefiltered=False, debug=False)
    ref_color, ref_radii = o.forward(xyz.numpy(), torch.sigmoid(op_raw).numpy(), shs=feats.numpy(), scales=torch.exp(log_s).numpy(),
                                     rotations=torch.nn.functional.normalize(rot_raw).numpy())
    ref_g = o.backward(dL.numpy())
    o.free()
    assert np.array_equal(pkg["radii"].cpu().numpy(), ref_radii) and (ref_radii > 0).sum() > P // 4
    util.assert_color_close(pkg["render"].detach().cpu().numpy(), ref_color, "3dgs colour")
    util.assert_grad_close(leaves["xyz"].grad.cpu().numpy(), ref_g["means3D"], "3dgs dL/dxyz")
    util.assert_grad_close(leaves["feats"].grad.cpu().numpy(), ref_g["shs"], "3dgs dL/dshs")
    util.assert_grad_close(pkg["viewspace_points"].grad.cpu().numpy(), ref_g["means2D"], "3dgs dL/dmeans2D")
    # through the activations: chain rule of the oracle's gradients in float64
    go = torch.from_numpy(ref_g["opacities"]).double() * (torch.sigmoid(op_raw) * (1 - torch.sigmoid(op_raw))).double()
    util.assert_grad_close(leaves["op"].grad.cpu().numpy(), go.numpy(), "3dgs dL/dopacity_raw")
    gs = torch.from_numpy(ref_g["scales"]).double() * torch.exp(log_s).double()
    util.assert_grad_close(leaves["log_s"].grad.cpu().numpy(), gs.numpy(), "3dgs dL/dlog_scale")
