"""CPU tests: the repo's own counterparts of the reference helpers, and the oracle's SH / projection conventions,
against golden vectors captured from the reference's importable pure-PyTorch code (tests/golden/make_golden.py)."""
import math

import numpy as np
import torch

from das3r_amd import camera


def test_projection_matrix_matches_reference(golden):
    for (fx, fy), ref in zip(golden["proj_fovs"], golden["proj_T"]):
        got = camera.projection_matrix(0.01, 100.0, float(fx), float(fy)).transpose(0, 1).numpy()
        np.testing.assert_allclose(got, ref, rtol=0, atol=1e-7)
    f2f = [camera.fov2focal(float(golden["proj_fovs"][2, 0]), 512), camera.fov2focal(float(golden["proj_fovs"][2, 1]), 208)]
    np.testing.assert_allclose(f2f, golden["proj_fov2focal"], rtol=1e-12)


def test_world2view_matches_reference(golden):
    for R, t, ref in zip(golden["w2v_R"], golden["w2v_t"], golden["w2v_out"]):
        np.testing.assert_allclose(camera.world2view(R, t).numpy(), ref, rtol=0, atol=2e-6)


def test_pose_pretransform_matches_reference(golden):
    poses = torch.tensor(golden["pose_in"])
    for i in range(poses.shape[0]):
        np.testing.assert_allclose(camera.camera_from_tensor(poses[i]).numpy(), golden["pose_w2c"][i], rtol=0, atol=1e-6)
        got = camera.quat_multiply(poses[i, :4], torch.tensor(golden["pose_gq"][i])).numpy()
        np.testing.assert_allclose(got, golden["pose_quadmul"][i], rtol=0, atol=1e-6)


def test_oracle_sh_basis_matches_reference_eval_sh(golden):
    """The rasterizer's colour stage is clamp_min(eval_sh + 0.5, 0) (/root/reference/gaussian_renderer/__init__.py:123-124).
    Place one splat along each golden direction (campos = 0) and read the oracle's per-splat rgb."""
    from oracle import c_oracle
    sh = golden["sh_coeffs"]          # (N, 3, 16)  [channel, coeff]
    dirs = golden["sh_dirs"]
    N = sh.shape[0]
    shs = np.ascontiguousarray(sh.transpose(0, 2, 1))   # rasterizer layout (N, 16, 3)
    # looking down +z with a very wide fov; put every splat in front of the camera by flipping directions with z < 0
    flip = np.where(dirs[:, 2:3] < 0, -1.0, 1.0).astype(np.float32)
    for deg in range(4):
        W = H = 64
        fov = 2 * math.atan(50.0)
        view = np.eye(4, dtype=np.float32)
        proj = camera.projection_matrix(0.01, 100.0, fov, fov).t().numpy()
        # a flipped direction -d evaluates the basis at -d: compare against the reference evaluated at the SAME direction
        o = c_oracle.RasterOracle(image_height=H, image_width=W, tanfovx=50.0, tanfovy=50.0, bg=np.zeros(3), scale_modifier=1.0,
                                  viewmatrix=view, projmatrix=proj, sh_degree=deg, campos=np.zeros(3))
        means = (dirs * 3.0).astype(np.float32)
        keep = dirs[:, 2] > 0.05
        color, radii = o.forward(means[keep], np.full((keep.sum(), 1), 0.5, np.float32), shs=shs[keep],
                                 scales=np.full((keep.sum(), 3), 0.01, np.float32),
                                 rotations=np.tile(np.array([[1, 0, 0, 0]], np.float32), (keep.sum(), 1)))
        S = o.saved()
        vis = radii > 0
        assert vis.sum() >= 10
        ref = np.maximum(golden[f"sh_eval_deg{deg}"][keep][vis] + 0.5, 0.0)
        np.testing.assert_allclose(S["rgb"][vis], ref, rtol=0, atol=3e-6)
        o.free()
    del flip


def test_loss_counterparts_match_reference(golden):
    from das3r_amd import losses
    a, b = torch.tensor(golden["loss_a"]), torch.tensor(golden["loss_b"])
    for i in range(2):
        assert abs(float(losses.l1_loss(a[i], b[i])) - golden["loss_l1"][i]) < 1e-7
        np.testing.assert_allclose(losses.ssim(a[i], b[i], size_average=False).numpy(), golden["loss_ssim_map"][i], atol=2e-6)
        assert abs(float(losses.ssim(a[i], b[i])) - golden["loss_ssim"][i]) < 1e-6
        np.testing.assert_allclose(losses.psnr(a[i], b[i]).numpy(), golden["loss_psnr"][i], rtol=1e-6)


def test_lr_schedule_matches_reference(golden):
    from das3r_amd import losses
    steps = golden["lr_steps"]
    f_xyz = losses.expon_lr_func(lr_init=1.6e-4 * 3.7, lr_final=1.6e-6 * 3.7, lr_delay_mult=0.01, max_steps=30000)
    f_conf = losses.expon_lr_func(lr_init=3e-3, lr_final=3e-4, max_steps=4000)
    f_cam = losses.expon_lr_func(lr_init=3e-5, lr_final=3e-6, max_steps=1000)
    f_delay = losses.expon_lr_func(lr_init=1e-2, lr_final=1e-4, lr_delay_steps=500, lr_delay_mult=0.01, max_steps=4000)
    np.testing.assert_allclose([f_xyz(int(s)) for s in steps], golden["lr_xyz"], rtol=1e-12)
    np.testing.assert_allclose([f_conf(int(s)) for s in steps], golden["lr_conf"], rtol=1e-12)
    np.testing.assert_allclose([f_cam(int(s)) for s in steps], golden["lr_cam"], rtol=1e-12)
    np.testing.assert_allclose([f_delay(int(s)) for s in steps], golden["lr_delay"], rtol=1e-12)
    np.testing.assert_allclose(losses.inverse_sigmoid(torch.tensor(golden["isig_in"])).numpy(), golden["isig_out"], rtol=1e-6)
    np.testing.assert_allclose(losses.rgb_to_sh(torch.tensor(golden["sh_rgb_in"])).numpy(), golden["sh_rgb2sh"], rtol=1e-6, atol=1e-7)
