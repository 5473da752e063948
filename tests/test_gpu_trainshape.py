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


@pytest.mark.parametrize("degree", [0, 1])
def test_sintel_shaped_step_vs_float64_host_and_c_oracle(degree, monkeypatch):
    import oracle.dense_trainer as dt
    from das3r_amd.fused import masked_photometric_loss
    from das3r_amd.model import OptimParams
    from das3r_amd.render import das3r_render
    from das3r_amd.train import build_from_sequence, synthetic_sequence
    seq = synthetic_sequence(frames=22, W=512, H=208, focal=600.0, n_splats=20000, seed=2)
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
    uid = 7
    bg = torch.zeros(3, device="cuda")

    # ---- the product: one fused iteration's forward + backward
    pkg = das3r_render(cams[uid], model, PIPE, bg, camera_pose=model.get_RT(uid), fused=True)
    loss, mse = masked_photometric_loss(pkg["render"], cams[uid].original_image, model._conf_static[uid], opt.lambda_dssim)
    loss.backward()
    psnr_frame = float((20 * torch.log10(1.0 / torch.sqrt(mse))).mean())
    torch.cuda.synchronize()

    # ---- float64 host restatement with the C oracle as its renderer
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
    report = {}
    for k, g, r in pairs:
        assert g is not None and r is not None, k
        g, r = g.detach().double().cpu().reshape(-1), r.reshape(-1)
        scale = float(r.abs().max())
        assert scale > 0 and bool(torch.isfinite(g).all()), k
        rel = float((g - r).abs().max()) / scale
        bad = float(((g - r).abs() > 1e-2 * r.abs() + 1e-4 * scale).double().mean())
        report[k] = (rel, bad)
        assert rel <= 2e-3, (k, rel)
        assert bad <= 1e-3, (k, bad)
    print("degree", degree, "P", P, "loss", float(loss), "max-norm / outlier fraction per gradient:",
          {k: (f"{a:.1e}", f"{b:.1e}") for k, (a, b) in report.items()})
