"""Independent float64 restatement of DAS3R's optimisation loop around the dense autograd rasterizer.  TEST INFRASTRUCTURE ONLY
(imported by tests/ alone; nothing in the product may reference oracle/).

What it is for (BASELINE.json configs[2] / [4]): the datasets of the PSNR configs are
not available offline, so the stand-in is to run the SAME optimisation twice on the same tiny synthetic sequence — once with the
product (HIP rasterizer, fp32, das3r_amd.train.train_step) and once with this file, which shares no code with it:
  renderer      oracle.dense_oracle.rasterize_dense: float64, dense pixels x splats, gradients from torch.autograd
  pre-transform written out again below (pose -> camera frame, quaternion product, sigmoid * conf_static, exp):
                /root/reference/gaussian_renderer/__init__.py:83-97,107,126
  loss          mean(0.8 |I s - G s| + 0.2 (1 - SSIM_map(I s, G s))), psnr gate at 26 dB: /root/reference/train_gui.py:560-586,
                utils/loss_utils.py:39-66, utils/image_utils.py:17-19 (written out again below, float64)
  optimizers    torch.optim.Adam(lr=0, eps=1e-15) over float64 copies of the parameters, the reference's groups and learning
                rates, exponential schedules: /root/reference/scene/gaussian_model.py:228-323, arguments/__init__.py:73-90
  schedule      oneupSHdegree at iteration % 3000 == 0: train_gui.py:542-543
Runs on whatever device its tensors live on (plain torch ops).
"""
import math

import torch
import torch.nn.functional as F

from .dense_oracle import rasterize_dense


def _expon(lr_init, lr_final, max_steps):
    return lambda step: math.exp(math.log(lr_init) * (1 - min(max(step / max_steps, 0.0), 1.0)) + math.log(lr_final) * min(max(step / max_steps, 0.0), 1.0))


def _window(ch, like):
    g = torch.tensor([math.exp(-((x - 5) ** 2) / (2 * 1.5 ** 2)) for x in range(11)], dtype=like.dtype, device=like.device)
    g = g / g.sum()
    return (g[:, None] * g[None, :])[None, None].expand(ch, 1, 11, 11).contiguous()


def ssim_map(a, b):
    w = _window(3, a)
    conv = lambda x: F.conv2d(x[None], w, padding=5, groups=3)[0]
    mu_a, mu_b = conv(a), conv(b)
    va, vb, vab = conv(a * a) - mu_a * mu_a, conv(b * b) - mu_b * mu_b, conv(a * b) - mu_a * mu_b
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    return ((2 * mu_a * mu_b + c1) * (2 * vab + c2)) / ((mu_a * mu_a + mu_b * mu_b + c1) * (va + vb + c2))


def psnr_channels(a, b):
    mse = ((a - b) ** 2).reshape(3, -1).mean(1)
    return 20 * torch.log10(1.0 / torch.sqrt(mse))


class DenseTrainer:
    """params: dict of fp32 tensors xyz [P,3], f_dc [P,1,3], f_rest [P,15,3], opacity [P,1], scaling [P,3], rotation [P,4],
    conf_static [F,H,W], Q [F,4], T [F,3] + `mask` (bool [F*H*W], the confident pixels that became Gaussians).
    cameras: list of dicts gt [3,H,W], fovx, fovy, proj_T [4,4] (the transposed projection matrix), indexed by uid."""

    def __init__(self, params, cameras, iterations=4000, lambda_dssim=0.2, psnr_threshold=26.0, spatial_lr_scale=1.0, dtype=torch.float64):
        """dtype: float64 is the restatement the product is held against; float32 exists for ONE purpose — measuring how far two
        arithmetics of the same optimisation drift apart over a schedule (tools/schedule_psnr.py: the noise floor under the PSNR
        bounds of tests/test_gpu_trainstep.py)."""
        self.dtype = dtype
        f64 = lambda t: t.detach().to(dtype).clone().requires_grad_(True)
        self.p = {k: f64(v) for k, v in params.items() if k != "mask"}
        self.mask = params["mask"].clone()
        self.cams = cameras
        self.lam, self.gate = lambda_dssim, psnr_threshold
        self.active_deg = 0
        s = spatial_lr_scale
        self.opt = torch.optim.Adam([
            {"params": [self.p["xyz"]], "lr": 0.00016 * s, "name": "xyz"},
            {"params": [self.p["f_dc"]], "lr": 0.0025, "name": "f_dc"},
            {"params": [self.p["f_rest"]], "lr": 0.0025 / 20.0, "name": "f_rest"},
            {"params": [self.p["opacity"]], "lr": 0.05, "name": "opacity"},
            {"params": [self.p["scaling"]], "lr": 0.005, "name": "scaling"},
            {"params": [self.p["rotation"]], "lr": 0.001, "name": "rotation"},
            {"params": [self.p["conf_static"]], "lr": 3e-3, "name": "conf_static"}], lr=0.0, eps=1e-15)
        self.opt_cam = torch.optim.Adam([{"params": [self.p["Q"]], "lr": 0.00003, "name": "pose_Q"},
                                         {"params": [self.p["T"]], "lr": 0.00003, "name": "pose_T"}], lr=0.0, eps=1e-15)
        self.lr_xyz = _expon(0.00016 * s, 0.0000016 * s, 30000)
        self.lr_cam = _expon(0.00003, 0.000003, 1000)
        self.lr_conf = _expon(3e-3, 3e-4, iterations)

    # ---- render() restated: pose (qw,qx,qy,qz,tx,ty,tz) of frame uid -> inputs of the rasterizer, all float64
    def render(self, uid, bg, pose=None):
        p, cam = self.p, self.cams[uid]
        pose = torch.cat([p["Q"][uid], p["T"][uid]]) if pose is None else pose
        q = pose[:4] / pose[:4].norm()
        w, x, y, z = q
        R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                         2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                         2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]).reshape(3, 3)
        means3D = p["xyz"] @ R.t() + pose[4:]
        a, b = pose[:4], p["rotation"]                      # Hamilton product of the RAW pose quaternion with the raw rotations
        rot = torch.stack([a[0] * b[:, 0] - a[1] * b[:, 1] - a[2] * b[:, 2] - a[3] * b[:, 3],
                           a[0] * b[:, 1] + a[1] * b[:, 0] + a[2] * b[:, 3] - a[3] * b[:, 2],
                           a[0] * b[:, 2] - a[1] * b[:, 3] + a[2] * b[:, 0] + a[3] * b[:, 1],
                           a[0] * b[:, 3] + a[1] * b[:, 2] - a[2] * b[:, 1] + a[3] * b[:, 0]], 1)
        opac = torch.sigmoid(p["opacity"]) * p["conf_static"].reshape(-1, 1)[self.mask]
        shs = torch.cat([p["f_dc"], p["f_rest"]], 1)
        dev = p["xyz"].device
        H, W = cam["gt"].shape[1:]
        eye = torch.eye(4, dtype=self.dtype, device=dev)
        means2D = torch.zeros(p["xyz"].shape[0], 3, dtype=self.dtype, device=dev, requires_grad=True)
        color, radii, _ = rasterize_dense(means3D, means2D, opac, shs=shs, scales=torch.exp(p["scaling"]), rotations=rot,
                                          image_height=H, image_width=W, tanfovx=math.tan(cam["fovx"] * 0.5),
                                          tanfovy=math.tan(cam["fovy"] * 0.5), bg=bg, scale_modifier=1.0, viewmatrix=eye,
                                          projmatrix=cam["proj_T"].to(self.dtype), sh_degree=self.active_deg,
                                          campos=torch.zeros(3, dtype=self.dtype, device=dev), dtype=self.dtype)
        return color, means2D

    def loss_of(self, uid, bg):
        cam = self.cams[uid]
        image, means2D = self.render(uid, bg)
        static = self.p["conf_static"][uid]
        img, gt = image * static, cam["gt"].to(self.dtype) * static
        loss = ((1.0 - self.lam) * (img - gt).abs() + self.lam * (1.0 - ssim_map(img, gt))).mean()
        return loss, psnr_channels(img, gt).mean(), means2D

    def step(self, iteration, uid, bg):
        """One iteration of train_gui.py:530-589.  -> (loss, psnr_frame) as floats; gradients stay in .grad until the next step."""
        for g in self.opt_cam.param_groups:
            g["lr"] = self.lr_cam(iteration)
        for g in self.opt.param_groups:
            if g["name"] == "xyz":
                g["lr"] = self.lr_xyz(iteration)
            elif g["name"] == "conf_static":
                g["lr"] = self.lr_conf(iteration)
        if iteration % 3000 == 0 and self.active_deg < 3:
            self.active_deg += 1
        self.opt.zero_grad(set_to_none=True)
        self.opt_cam.zero_grad(set_to_none=True)
        loss, psnr_frame, means2D = self.loss_of(uid, bg)
        loss.backward()
        self.viewspace_grad = means2D.grad
        with torch.no_grad():
            self.opt.step()
            if float(psnr_frame) > self.gate:
                self.opt_cam.step()
        return float(loss), float(psnr_frame)

    @torch.no_grad()
    def heldout_psnr(self, gt, pose, cam_uid_for_intrinsics, bg, static_mask=None):
        """train_test_psnr.py:262-289 for one held-out view: clamp, optional (1 - gt_dynamic_mask), mean over channels."""
        image, _ = self.render(cam_uid_for_intrinsics, bg.to(self.dtype), pose=pose.to(self.dtype))
        img, g = image.clamp(0.0, 1.0), gt.to(self.dtype).clamp(0.0, 1.0)
        if static_mask is not None:
            img, g = img * static_mask, g * static_mask
        return float(psnr_channels(img, g).mean()), float((img - g).abs().mean())
