"""das3r_amd.integrate.patch() on stand-ins of the reference's modules.  CPU part: what is swapped, re-bound and
restored, and that a call whose preconditions fail goes to the original function untouched.  GPU part: a mocked train_gui.py:542-589
iteration driven through the patch against the unpatched loop and against the direct fused iteration."""
import sys
import types

import pytest
import torch

from das3r_amd import integrate
from das3r_amd.model import OptimParams, SplatModel


def _stand_ins():
    calls = []

    def render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None, camera_pose=None, filtering=None, use_conf=True):
        """the reference's signature (gaussian_renderer/__init__.py:23-24); the body is the repo's unfused counterpart of it"""
        from das3r_amd.render import das3r_render
        calls.append("original")
        return das3r_render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier, override_color, camera_pose, filtering, use_conf)

    renderer = types.ModuleType("gaussian_renderer_standin")
    renderer.render = render
    consumer = types.ModuleType("train_gui_standin")      # did `from gaussian_renderer import render` before the patch
    consumer.render = render
    sys.modules[consumer.__name__] = consumer
    from das3r_amd import losses as repo_losses
    loss_utils = types.ModuleType("loss_utils_standin")   # utils/loss_utils.py: the repo's counterpart of its ssim (pinned by tests/golden)
    loss_utils.ssim = lambda img1, img2, window_size=11, size_average=True: (calls.append("original ssim"), repo_losses.ssim(img1, img2, window_size, size_average))[1]
    consumer.ssim = loss_utils.ssim                        # ... and `from utils.loss_utils import l1_loss, ssim`
    renderer.loss_utils = loss_utils

    class GaussianModel(SplatModel):                       # the reference's training_setup(training_args): plain torch.optim.Adam
        def training_setup(self, training_args):
            return SplatModel.training_setup(self, training_args, fused=False)

    return renderer, consumer, GaussianModel, calls


def test_patch_swaps_rebinds_and_restores():
    renderer, consumer, Model, calls = _stand_ins()
    original = renderer.render
    try:
        original_ssim = renderer.loss_utils.ssim
        done = integrate.patch(renderer, Model, renderer.loss_utils)
        assert done == {"ssim": "loss_utils_standin", "render": "gaussian_renderer_standin", "model": "GaussianModel"}
        assert renderer.render is not original and consumer.render is renderer.render and renderer.render._das3r_original is original
        assert renderer.loss_utils.ssim is not original_ssim and consumer.ssim is renderer.loss_utils.ssim and consumer.ssim._das3r_original is original_ssim
        assert integrate.patch(renderer, Model, renderer.loss_utils)["model"] == "GaussianModel" and renderer.render._das3r_original is original   # idempotent
        assert consumer.ssim._das3r_original is original_ssim
        # host images: the call goes to the original ssim, same value
        a, b = torch.rand(3, 20, 24), torch.rand(3, 20, 24)
        from das3r_amd import losses as repo_losses
        assert torch.equal(consumer.ssim(a, b, size_average=False), repo_losses.ssim(a, b, size_average=False)) and calls == ["original ssim"]
        calls.clear()
        # preconditions fail (host tensors): the call goes to the original function, arguments untouched
        pc = types.SimpleNamespace(**{n: torch.zeros(4, 3) for n in ("_xyz", "_rotation", "_scaling", "_opacity", "_conf_static", "_features_dc", "_features_rest")},
                                   aggregated_mask=torch.ones(4, dtype=torch.bool))
        with pytest.raises(Exception):   # (the original then fails on the host tensors further down: the product has no CPU path)
            renderer.render(None, pc, types.SimpleNamespace(compute_cov3D_python=False, convert_SHs_python=False), torch.zeros(3), camera_pose=torch.zeros(7))
        assert calls == ["original"]
        # host parameters: the optimizers stay torch's
        m = Model(0)
        m._xyz = torch.nn.Parameter(torch.zeros(4, 3))
        for n, shape in (("_features_dc", (4, 1, 3)), ("_features_rest", (4, 0, 3)), ("_opacity", (4, 1)), ("_scaling", (4, 3)), ("_rotation", (4, 4)),
                         ("_conf_static", (1, 2, 2)), ("Q", (1, 4)), ("T", (1, 3))):
            setattr(m, n, torch.nn.Parameter(torch.zeros(shape)))
        m.training_setup(OptimParams())
        assert isinstance(m.optimizer, torch.optim.Adam) and isinstance(m.optimizer_cam, torch.optim.Adam)
    finally:
        integrate.unpatch()
        sys.modules.pop(consumer.__name__, None)
    assert renderer.render is original and consumer.render is original and not hasattr(Model.training_setup, "_das3r_original")
    assert renderer.loss_utils.ssim is original_ssim and consumer.ssim is original_ssim


NAMES = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity", "scaling": "_scaling", "rotation": "_rotation",
         "conf_static": "_conf_static", "Q": "Q", "T": "T"}
PIPE = types.SimpleNamespace(debug=False, compute_cov3D_python=False, convert_SHs_python=False)


def _reference_loop(render, gaussians, cams, opt, iterations, bg, psnr_threshold, ssim=None):
    """train_gui.py:532-589, statement by statement, on the names it uses (the loss helpers are the repo's counterparts of
    utils/loss_utils.py, pinned by tests/golden/ref_helpers.npz; `ssim`: the name the loop's module holds — patched or not)."""
    from das3r_amd.losses import l1_loss, psnr
    if ssim is None:
        from das3r_amd.losses import ssim
    for iteration, uid in iterations:
        gaussians.update_learning_rate(iteration)
        if iteration % 3000 == 0:
            gaussians.oneupSHdegree()
        viewpoint_cam = cams[uid]
        pose = gaussians.get_RT(viewpoint_cam.uid)
        render_pkg = render(viewpoint_cam, gaussians, PIPE, bg, camera_pose=pose)
        image = render_pkg["render"]
        gt_image = viewpoint_cam.original_image.cuda()
        static = gaussians._conf_static[viewpoint_cam.uid]
        image = image * static
        gt_image = gt_image * static
        Ll1 = l1_loss(image, gt_image, reduce=False)
        Lssim = ssim(image, gt_image, size_average=False)
        psnr_frame = psnr(image, gt_image).mean()
        loss = (1.0 - opt.lambda_dssim) * Ll1 + opt.lambda_dssim * (1.0 - Lssim)
        loss = (loss).mean()
        loss.backward(retain_graph=True)
        with torch.no_grad():
            gaussians.optimizer.step()
            gaussians.optimizer.zero_grad(set_to_none=True)
            if psnr_frame > psnr_threshold:
                gaussians.optimizer_cam.step()
            gaussians.optimizer_cam.zero_grad(set_to_none=True)


@pytest.mark.gpu
def test_mocked_train_gui_iteration_through_the_patch_matches_the_direct_iteration():
    from das3r_amd import _lib
    from das3r_amd.fused import FusedAdam
    from das3r_amd.train import consistent_sequence, build_from_sequence, train_step
    renderer, consumer, Model, calls = _stand_ins()
    seq = consistent_sequence(frames=6, W=128, H=80, focal=150.0, n_splats=3000, seed=21)
    bg = torch.zeros(3, device="cuda")
    schedule = [(it, (it * 5) % 5) for it in range(2995, 3007)]   # twelve iterations across the SH-degree bump at 3000
    opt = OptimParams(iterations=4000)

    def fresh(cls):
        model, cams = build_from_sequence(seq)
        model.__class__ = cls
        g = torch.Generator().manual_seed(2)
        with torch.no_grad():   # a generic state: with the initial isotropic scales the rotation gradient is analytically zero — pure rounding noise, which
            model._features_rest.copy_((0.05 * torch.randn(model._features_rest.shape, generator=g)).cuda())   # Adam (eps = 1e-15) turns into full steps
            model._scaling += 0.3 * torch.randn(model._scaling.shape, generator=g).cuda()
            model._rotation.copy_(torch.nn.functional.normalize(torch.randn(model._rotation.shape, generator=g)).cuda())
        return model, cams

    # (a) the unpatched loop: torch.optim.Adam and the reference's PyTorch glue around the HIP rasterizer
    plain, cams = fresh(Model)
    plain.training_setup(opt)
    _reference_loop(consumer.render, plain, cams, opt, schedule, bg, opt.psnr_threshold, consumer.ssim)
    n_plain = len(calls)
    assert calls.count("original") == len(schedule) and calls.count("original ssim") == len(schedule)
    # (b) the same loop, same objects' classes, after the one-liner
    try:
        integrate.patch(renderer, Model, renderer.loss_utils)
        patched, cams_b = fresh(Model)
        patched.training_setup(opt)
        assert isinstance(patched.optimizer, FusedAdam) and isinstance(patched.optimizer_cam, FusedAdam)
        assert [g["name"] for g in patched.optimizer.param_groups] == [g["name"] for g in plain.optimizer.param_groups]
        assert [g["name"] for g in patched.optimizer_cam.param_groups] == ["pose_Q", "pose_T", "fovX", "fovY"]
        _lib.profile_enable(True)
        try:
            _reference_loop(consumer.render, patched, cams_b, opt, schedule, bg, opt.psnr_threshold, consumer.ssim)
            torch.cuda.synchronize()
        finally:
            _lib.profile_enable(False)
        kernels = _lib.profile_report(raw=True)
        assert len(calls) == n_plain, "every render and every ssim call of the patched loop took the fused path"
        launched = lambda prefix: sum(n for k, (n, _) in kernels.items() if k.startswith(prefix))
        assert launched("pretransform_forward_kernel") == len(schedule) and launched("pretransform_backward_kernel") == len(schedule), kernels
        assert launched("adam_kernel") >= len(schedule), kernels
        assert launched("photometric_forward_kernel") == len(schedule) and launched("photometric_backward_kernel") == len(schedule), kernels   # (round 6: the loop's ssim)
        assert patched.active_sh_degree == 1 and patched.optimizer.active_sh_degree == 1
    finally:
        integrate.unpatch()
        sys.modules.pop(consumer.__name__, None)
    # (c) the direct fused iteration (das3r_amd/fast_step.py) on the same schedule
    direct, cams_c = fresh(SplatModel)
    direct.training_setup(opt, fused=True)
    for it, uid in schedule:
        train_step(direct, cams_c[uid], opt, it, PIPE, bg, fused=True)
    torch.cuda.synchronize()
    # the direct iteration runs the same pre-transform / Adam kernels (another loss kernel); the unpatched loop is another ARITHMETIC of the
    # same step (torch's glue, torch.optim.Adam): with eps = 1e-15 an Adam step moves an element by its learning rate whatever the size of
    # its gradient, so a noise-level gradient whose sign differs shows as a full step (opacity: lr 0.05) — measured 0.6 % of the opacities
    # after twelve steps against the unpatched loop
    for ref_model, what, bar in ((plain, "unpatched loop", 1.5e-2), (direct, "direct iteration", 3e-3)):
        for k, a in NAMES.items():
            x, y = getattr(patched, a).detach(), getattr(ref_model, a).detach()
            far = (x - y).abs() > 1e-5 + 1e-4 * y.abs()   # (an Adam step moves an element by at most its learning rate: sign flips of noise-level gradients)
            assert float(far.double().mean()) <= bar, (what, k, float(far.double().mean()))
    assert patched.optimizer.state[patched._features_rest]["step"] == len(schedule)
