"""Oracle parity of ONE optimisation step at the shape DAS3R really trains at (BASELINE
configs[2]): a Sintel-sized sequence — 22 frames of 512 x 208, two of them held out, every pixel of the 20 training frames one
Gaussian = 2 129 920 Gaussians — through the fused path the job's time is quoted on (fused pre-transform -> HIP rasterizer ->
fused masked L1 + SSIM loss -> backward), against

    float64 torch on the host for everything around the rasterizer (oracle/dense_trainer.py: pose -> camera frame, quaternion
    product, sigmoid * conf_static, exp; the loss), and oracle/raster_oracle.c as the renderer in its place
    (/root/reference/gaussian_renderer/__init__.py:83-140, train_gui.py:559-589),

loss, frame PSNR and EVERY parameter gradient, at SH degree 0 (the first 3000 iterations: the DC tensor alone reaches the
rasterizer) and at degree 1 (the active prefix [P, 4, 3], whose gradient FusedAdam takes compact: das3r_amd/fused.py).  The datasets
themselves are not available offline; what this adds to the 32 x 24 stand-in of test_gpu_trainstep.py is the real P, the real image
size and the real list lengths for one step.

Tolerances (fp32 product vs float64 host + fp32 C renderer): loss 2e-5 relative; a gradient tensor within 2e-3 of its largest
element in max-norm, and element-wise within 1e-2 |ref| + 1e-4 max|ref| for all but 1e-3 of its elements (the renderer's inputs
are rounded from float64 on one side and computed in fp32 on the other: a Gaussian whose radius or alpha sits on a threshold
may land on the other side)."""
import math
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
PIPE = SimpleNamespace(debug=False, compute_cov3D_python=False, convert_SHs_python=False)
NAMES = {"xyz": "_xyz", "f_dc": "_features_dc", "opacity": "_opacity", "scaling": "_scaling", "rotation": "_rotation",
         "conf_static": "_conf_static", "Q": "Q", "T": "T"}


class _COracleRaster(torch.autograd.Function):
    """oracle/raster_oracle.c behind torch.autograd: float64 tensors in, fp32 arrays to the C code, float64 back."""

    @staticmethod
    def forward(ctx, means3D, means2D, opac, shs, scales, rotations, settings):
        from oracle import c_oracle
        o = c_oracle.RasterOracle(**settings)
        f32 = lambda t: t.detach().to(torch.float32).numpy()
        color, _ = o.forward(f32(means3D), f32(opac), shs=f32(shs), scales=f32(scales), rotations=f32(rotations))
        ctx.o = o
        return torch.from_numpy(color).to(torch.float64)

    @staticmethod
    def backward(ctx, g):
        gr = ctx.o.backward(g.to(torch.float32).numpy())
        ctx.o.free()
        t = lambda k: torch.from_numpy(gr[k]).to(torch.float64)
        return t("means3D"), t("means2D"), t("opacities"), t("shs"), t("scales"), t("rotations"), None


def _oracle_as_renderer(means3D, means2D, opac, shs=None, scales=None, rotations=None, image_height=0, image_width=0, tanfovx=0.0,
                        tanfovy=0.0, bg=None, scale_modifier=1.0, viewmatrix=None, projmatrix=None, sh_degree=0, campos=None, dtype=None):
    """Same signature as oracle.dense_oracle.rasterize_dense (what DenseTrainer.render calls)."""
    settings = dict(image_height=image_height, image_width=image_width, tanfovx=tanfovx, tanfovy=tanfovy, bg=bg.float().numpy(),
                    scale_modifier=scale_modifier, viewmatrix=viewmatrix.float().numpy(), projmatrix=projmatrix.float().numpy(),
                    sh_degree=sh_degree, campos=campos.float().numpy())
    return _COracleRaster.apply(means3D, means2D, opac, shs, scales, rotations, settings), None, None


def _sintel_model(degree, depth="noise"):
    from das3r_amd.model import OptimParams
    from das3r_amd.train import build_from_sequence, synthetic_sequence
    seq = synthetic_sequence(frames=22, W=512, H=208, focal=600.0, n_splats=20000, seed=2, depth=depth)
    model, cams, test = build_from_sequence(seq, heldout=True)
    P = model.get_xyz.shape[0]
    assert len(cams) == 20 and len(test) == 2 and P == 20 * 512 * 208
    opt = OptimParams(iterations=4000)
    model.training_setup(opt, fused=True)
    gen = torch.Generator().manual_seed(17)
    with torch.no_grad():   # leave the initial state (isotropic, identity quaternions, zero higher-order SH) for a generic one
        model._scaling += 0.3 * torch.randn(model._scaling.shape, generator=gen).cuda()
        model._rotation.copy_(torch.nn.functional.normalize(torch.randn(model._rotation.shape, generator=gen)).cuda())
        model._features_rest.copy_((0.05 * torch.randn(model._features_rest.shape, generator=gen)).cuda())
        model._conf_static.mul_(0.6 + 0.4 * torch.rand(model._conf_static.shape, generator=gen).cuda())
    model.active_sh_degree = degree
    model.optimizer.set_active_sh_degree(degree)
    return model, cams, opt, P


def _host_step(model, cams, uid, degree, monkeypatch):
    """float64 host restatement of the iteration with the C oracle as its renderer -> (host trainer after backward, loss, frame PSNR,
    means2D leaf)"""
    import oracle.dense_trainer as dt
    monkeypatch.setattr(dt, "rasterize_dense", _oracle_as_renderer)
    cpu = lambda t: t.detach().cpu()
    params = dict(xyz=cpu(model._xyz), f_dc=cpu(model._features_dc), f_rest=cpu(model._features_rest), opacity=cpu(model._opacity),
                  scaling=cpu(model._scaling), rotation=cpu(model._rotation), conf_static=cpu(model._conf_static), Q=cpu(model.Q),
                  T=cpu(model.T), mask=cpu(model.aggregated_mask))
    cameras = [dict(gt=cpu(c.original_image), fovx=c.FoVx, fovy=c.FoVy, proj_T=cpu(c.projection_matrix)) for c in cams]
    host = dt.DenseTrainer(params, cameras, iterations=4000)
    host.active_deg = degree
    d_loss, d_psnr, d_m2d = host.loss_of(uid, torch.zeros(3, dtype=torch.float64))
    d_loss.backward()
    return host, d_loss, d_psnr, d_m2d


FLIP_ELEMENTS = 1e-5   # fraction of a gradient tensor's elements that may miss the max-norm bound because of a threshold flip / order swap ...
FLIP_REL = 5e-2        # ... each of them still within this much of the tensor's largest element


def _compare_grads(pairs, what, strict=False):
    """strict (the noise-depth first iteration, which held the plain bound before the allowance existed — ADVICE r5): max-norm 2e-3 of
    the largest element for EVERY element, no flips.  Otherwise: max-norm 2e-3 of the largest element and the element-wise bound of the module docstring.  The max-norm bound may be missed by
    single elements (at most FLIP_ELEMENTS of the tensor, >= 3, each within FLIP_REL): one (pixel, Gaussian) pair whose alpha sits on
    1/255 or whose T sits on 1e-4 decides differently on the two sides, and at this shape — Gaussians of one or two pixels — that
    pair can be half of a Gaussian's gradient; when the Gaussian is one of the tensor's largest, the flip shows in the max norm.
    Measured with tools/probes/smooth_step_diag.py on the smooth-depth sequence with IDENTICAL fp32 inputs on both sides: one Gaussian
    of 2 129 920 off by 5 % (1.6e-3 .. 2.1e-3 of the maximum), everything else within the element-wise bound; with the float64 host's
    inputs (rounded to fp32 on one side, computed in fp32 on the other) 15 of the 6.4 M elements of dL/dxyz, up to 5.6e-3: besides the
    flips, two overlapping Gaussians of a surface whose depths agree to the last bit or two are blended in the other order on the two
    sides (on spatially coherent depth maps neighbours in depth ARE neighbours in the image) — on the first-forward path and the
    steady-state path alike (their gradients are bit-identical to each other)."""
    report = {}
    for k, g, r in pairs:
        assert g is not None and r is not None, k
        g, r = g.detach().double().cpu().reshape(-1), r.reshape(-1)
        scale = float(r.abs().max())
        assert scale > 0 and bool(torch.isfinite(g).all()), k
        d = (g - r).abs()
        rel = float(d.max()) / scale
        flips = int((d > 2e-3 * scale).sum())
        bad = float((d > 1e-2 * r.abs() + 1e-4 * scale).double().mean())
        report[k] = (rel, bad)
        if strict:
            assert rel <= 2e-3, (what, k, rel)
        else:
            assert flips <= max(3, int(FLIP_ELEMENTS * d.numel())) and rel <= FLIP_REL, (what, k, rel, flips)
        assert bad <= 1e-3, (what, k, bad)
    return report


@pytest.mark.parametrize("degree", [0, 1])
def test_sintel_shaped_step_vs_float64_host_and_c_oracle(degree, monkeypatch):
    """The FIRST iteration of a model (autograd form of the fused path): the library has not seen the shape, so the binning is the
    global depth sort and the forward the rows kernel; the steady state is the test below."""
    from das3r_amd.fused import masked_photometric_loss
    from das3r_amd.render import das3r_render
    model, cams, opt, P = _sintel_model(degree)
    uid = 7
    bg = torch.zeros(3, device="cuda")

    # ---- the product: one fused iteration's forward + backward
    pkg = das3r_render(cams[uid], model, PIPE, bg, camera_pose=model.get_RT(uid), fused=True)
    loss, mse = masked_photometric_loss(pkg["render"], cams[uid].original_image, model._conf_static[uid], opt.lambda_dssim)
    loss.backward()
    psnr_frame = float((20 * torch.log10(1.0 / torch.sqrt(mse))).mean())
    torch.cuda.synchronize()

    # ---- float64 host restatement with the C oracle as its renderer
    host, d_loss, d_psnr, d_m2d = _host_step(model, cams, uid, degree, monkeypatch)

    assert abs(float(loss) - float(d_loss)) <= 2e-5 * abs(float(d_loss)) + 1e-7, (float(loss), float(d_loss))
    assert abs(psnr_frame - float(d_psnr)) < 2e-3, (psnr_frame, float(d_psnr))
    pairs = [(k, getattr(model, a).grad, host.p[k].grad) for k, a in NAMES.items()]
    pairs.append(("means2D", pkg["viewspace_points"].grad, d_m2d.grad))
    K1 = (degree + 1) ** 2 - 1
    compact = getattr(model._features_rest, "_das3r_compact_grad", None)
    if degree == 0:   # the DC tensor alone was rendered: nothing reaches f_rest on either side
        assert model._features_rest.grad is None and compact is None
        assert float(host.p["f_rest"].grad.abs().max()) == 0.0
    else:             # the gradient of the active prefix, parked compact for FusedAdam; the host's is zero above it
        assert model._features_rest.grad is None and compact is not None and tuple(compact.shape) == (P, K1, 3)
        assert float(host.p["f_rest"].grad[:, K1:].abs().max()) == 0.0
        pairs.append(("f_rest", compact, host.p["f_rest"].grad[:, :K1]))
    report = _compare_grads(pairs, f"degree {degree}", strict=True)   # (noise depth, first iteration)
    print("degree", degree, "P", P, "loss", float(loss), "max-norm / outlier fraction per gradient:",
          {k: (f"{a:.1e}", f"{b:.1e}") for k, (a, b) in report.items()})


@pytest.mark.parametrize("depth,degree", [("noise", 0), ("smooth", 0), ("smooth", 1)])
def test_sintel_shaped_steady_state_direct_step_vs_float64_host_and_c_oracle(depth, degree, monkeypatch):
    """the iteration the train-step time is QUOTED on — the direct form (das3r_amd/fast_step.py: a straight sequence
    of C-ABI calls, no autograd) in its steady state, on both kinds of depth maps (depth="smooth": the spatially coherent maps of a depth
    predictor, bench.py's train_step_smooth_ms).  Two warm-up passes over the same view teach the library the shape (they drop every
    per-Gaussian gradient: geometry="pose", nothing changes), then the THIRD forward + backward is held to the float64 host
    restatement with oracle/raster_oracle.c as its renderer: loss, frame PSNR, every parameter gradient, the pose gradient, the mask
    gradient — and the kernels that ran are the ones the train-step profiles show: segmented binning without a global depth sort,
    render_forward_lanes_kernel + the bucket-parallel render_backward_blk_kernel with the block-level last contributor on noise depth maps,
    the 2x2-region kernels both ways on the smooth ones.
    Reference: /root/reference/train_gui.py:532-589 around gaussian_renderer/__init__.py:83-140."""
    from das3r_amd import _lib, fast_step
    model, cams, opt, P = _sintel_model(degree, depth)
    assert fast_step.available(model, PIPE)
    uid = 7
    cam = cams[uid]
    bg = torch.zeros(3, device="cuda")
    st = fast_step._state(model)
    Qg, Tg = torch.zeros_like(model.Q), torch.zeros_like(model.T)

    def run(geometry):
        with torch.no_grad():
            out = fast_step.forward_backward(model, cam, model.Q[uid], model.T[uid], Qg[uid], Tg[uid], model._conf_static[uid],
                                             opt.lambda_dssim, bg, geometry=geometry)
        torch.cuda.synchronize()
        return out

    run("pose")
    run("pose")
    Qg.zero_()
    Tg.zero_()
    _lib.profile_report()
    _lib.profile_enable(True)
    try:
        out8, d_static, pkg = run("grads")
    finally:
        _lib.profile_enable(False)
    kernels = _lib.profile_report(raw=True)
    n = lambda prefix, sub="": sum(c for k, (c, _) in kernels.items() if k.startswith(prefix) and sub in k)
    assert n("segment_sort_kernel") == 1 and n("depth_hist_kernel") == 0, kernels
    # (one workgroup per tile, or four where the tile lists are skewed — the smooth-depth maps: api.hip decides from the tile ranges)
    assert n("render_forward_lanes_kernel") + n("render_forward_regions_kernel") == 1 and n("render_forward_rows_kernel") == 0, kernels
    if depth == "noise":
        assert n("render_forward_lanes_kernel") == 1, kernels
        assert n("render_backward_blk_kernel", "true>") == 1 and n("render_backward_") == 1, kernels
    else:   # (round 6: the backward follows the forward — skewed or crowded lists take the 2x2-region walk, render_bwd_rgn.hip)
        assert n("render_forward_regions_kernel") == 1 and n("render_backward_regions_kernel") == 1 and n("render_backward_") == 1, kernels
    loss, psnr_frame = float(out8[0]), float(out8[4])

    host, d_loss, d_psnr, d_m2d = _host_step(model, cams, uid, degree, monkeypatch)
    d_loss, d_psnr = d_loss.detach(), d_psnr.detach()
    assert abs(loss - float(d_loss)) <= 2e-5 * abs(float(d_loss)) + 1e-7, (loss, float(d_loss))
    assert abs(psnr_frame - float(d_psnr)) < 2e-3, (psnr_frame, float(d_psnr))
    conf_grad = model._conf_static.grad.clone()
    conf_grad[uid] += d_static          # the loss sees conf_static twice: as opacity factor and as the frame's mask (train_step adds it)
    pairs = [(k, getattr(model, a).grad, host.p[k].grad) for k, a in NAMES.items() if k not in ("conf_static", "Q", "T")]
    pairs += [("conf_static", conf_grad, host.p["conf_static"].grad), ("Q", Qg, host.p["Q"].grad), ("T", Tg, host.p["T"].grad),
              ("means2D", pkg["viewspace_points"].grad, d_m2d.grad)]
    K1 = (degree + 1) ** 2 - 1
    compact = getattr(model._features_rest, "_das3r_compact_grad", None)
    if degree == 0:
        assert model._features_rest.grad is None and compact is None
    else:
        assert compact is not None and tuple(compact.shape) == (P, K1, 3)
        pairs.append(("f_rest", compact, host.p["f_rest"].grad[:, :K1]))
    report = _compare_grads(pairs, f"{depth} depth, degree {degree}")
    print("steady state,", depth, "depth, degree", degree, "loss", loss, "kernels", sorted(kernels),
          {k: (f"{a:.1e}", f"{b:.1e}") for k, (a, b) in report.items()})
