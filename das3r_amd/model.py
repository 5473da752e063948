"""SplatModel — the parameter container of the train-step harness: the repo's counterpart of the
parts of /root/reference/scene/gaussian_model.py the hot loop touches.

Reproduced: activations (exp / sigmoid / identity, :32-47); per-pixel initialisation create_from_cameras (:573-659):
every confident pixel of every frame becomes one Gaussian, xyz = unprojected depth, SH dc = RGB2SH(rgb), scales =
log(sqrt(clamp_min(distCUDA2(xyz), 1e-7))) repeated 3x, identity quaternions, opacity = inverse_sigmoid(1/num_frames),
conf_static = 1 - dyna_avg with shape (frames, H, W); the two Adam optimizers (lr=0, eps=1e-15) with the reference's
groups and LRs and the exponential schedules (:228-323; defaults /root/reference/arguments/__init__.py:73-90);
oneupSHdegree (:199-201); pose parameters Q/T as (frames,4)/(frames,3) tensors (:149-184); the held-out views' poses
test_Q/test_T with their own optimizer (:132-147,263-268) and the FoVx/FoVy "parameters" of the camera optimizer (:163-166,
253-254).  The last two are INERT in the reference and therefore here: render() turns FoV into Python floats with math.tan, so
no gradient ever reaches FoVx/FoVy (Adam skips parameters whose .grad is None and never creates state for them),
and the test-pose pass of train_test_psnr.py steps optimizer_cam — which holds Q/T, not test_Q/test_T — so optimizer_cam_test is
built and never stepped (C5).  They are kept so that optimizer groups, their order and their state match the reference's.
"""
from dataclasses import dataclass

import torch
from torch import nn

from .knn import distCUDA2
from .losses import expon_lr_func, inverse_sigmoid, rgb_to_sh


@dataclass
class OptimParams:   # /root/reference/arguments/__init__.py:73-90 (densification is disabled in DAS3R: SURVEY.md C1)
    iterations: int = 4000
    position_lr_init: float = 0.00016
    position_lr_final: float = 0.0000016
    position_lr_delay_mult: float = 0.01
    position_lr_max_steps: int = 30_000
    feature_lr: float = 0.0025
    opacity_lr: float = 0.05
    scaling_lr: float = 0.005
    rotation_lr: float = 0.001
    lambda_dssim: float = 0.2
    psnr_threshold: float = 26.0   # train_gui.py:584 camera-optimizer gate


def depth_to_points(K, cam2world, depth):
    """Unproject per-frame depth maps: K (F,3,3) with fx == fy, cam2world (F,4,4), depth (F,H,W) -> (F,H,W,3) world points
    (/root/reference/utils/pose_utils.py:572-583,672-682: x = d (u - cx)/f, y = d (v - cy)/f, z = d, then the pose)."""
    F, H, W = depth.shape
    v, u = torch.meshgrid(torch.arange(H, device=depth.device, dtype=depth.dtype),
                          torch.arange(W, device=depth.device, dtype=depth.dtype), indexing="ij")
    f = K[:, 0, 0][:, None, None]
    cx, cy = K[:, 0, 2][:, None, None], K[:, 1, 2][:, None, None]
    pts = torch.stack([depth * (u[None] - cx) / f, depth * (v[None] - cy) / f, depth], -1)
    return torch.einsum("bij,bhwj->bhwi", cam2world[:, :3, :3], pts) + cam2world[:, None, None, :3, 3]


class SplatModel:
    def __init__(self, sh_degree=3):
        self.max_sh_degree = sh_degree
        self.active_sh_degree = 0
        self.spatial_lr_scale = 1.0
        self.optimizer = self.optimizer_cam = self.optimizer_cam_test = None
        self.enable_test = False
        self.test_Q = self.test_T = None
        self.FoVx = self.FoVy = None

    # ---- activations
    @property
    def get_xyz(self):
        return self._xyz

    @property
    def get_scaling(self):
        return torch.exp(self._scaling)

    @property
    def get_rotation(self):
        return torch.nn.functional.normalize(self._rotation)

    @property
    def get_opacity(self):
        return torch.sigmoid(self._opacity)

    @property
    def get_features(self):
        return torch.cat((self._features_dc, self._features_rest), dim=1)

    def get_covariance(self, scaling_modifier=1):
        """scene/gaussian_model.py:195-196: Sigma from the WORLD-frame rotation parameter (render()'s compute_cov3D_python mode)."""
        from .render import covariance_from_scaling_rotation
        return covariance_from_scaling_rotation(self.get_scaling, scaling_modifier, self._rotation)

    def get_RT(self, idx):
        return torch.cat([self.Q[idx], self.T[idx]])

    def get_RT_test(self, idx):
        return torch.cat([self.test_Q[idx], self.test_T[idx]])

    def init_test_RT_seq(self, w2c_pose7):
        """Poses (qw,qx,qy,qz,tx,ty,tz) of the held-out views (gaussian_model.py:132-147); none -> enable_test stays False."""
        if w2c_pose7 is None or len(w2c_pose7) == 0:
            self.enable_test = False
            return
        self.enable_test = True
        self.test_Q = w2c_pose7[:, :4].clone().contiguous().requires_grad_(True)
        self.test_T = w2c_pose7[:, 4:].clone().contiguous().requires_grad_(True)

    def init_fov(self, fovx, fovy):
        """gaussian_model.py:163-166: the first training camera's FoV as 0-d tensors that require grad (and never get one)."""
        dev = self._xyz.device
        self.FoVx = torch.tensor(float(fovx), device=dev).requires_grad_(True)
        self.FoVy = torch.tensor(float(fovy), device=dev).requires_grad_(True)

    def oneupSHdegree(self):
        if self.active_sh_degree < self.max_sh_degree:
            self.active_sh_degree += 1
        if hasattr(self.optimizer, "set_active_sh_degree"):
            self.optimizer.set_active_sh_degree(self.active_sh_degree)

    # ---- initialisation from per-frame depth / confidence / dynamic maps
    def create_from_frames(self, images, depths, confs, dyna_avg, K, cam2world, w2c_pose7, spatial_lr_scale=1.0, conf_thre=1.0):
        """images (F,3,H,W) in [0,1]; depths/confs/dyna_avg (F,H,W); K (F,3,3); cam2world (F,4,4); w2c_pose7 (F,7)."""
        dev = images.device
        self.spatial_lr_scale = spatial_lr_scale
        F = images.shape[0]
        pts = depth_to_points(K.float(), cam2world.float(), depths.float()).reshape(-1, 3)
        col = images.permute(0, 2, 3, 1).reshape(-1, 3)
        self.aggregated_mask = confs.reshape(-1) > torch.tensor(conf_thre).log()
        pts = pts[self.aggregated_mask].contiguous()
        col = col[self.aggregated_mask]
        n = pts.shape[0]
        feats = torch.zeros(n, 3, (self.max_sh_degree + 1) ** 2, device=dev)
        feats[:, :3, 0] = rgb_to_sh(col)
        dist2 = torch.clamp_min(distCUDA2(pts), 0.0000001)
        scales = torch.log(torch.sqrt(dist2))[..., None].repeat(1, 3)
        rots = torch.zeros(n, 4, device=dev)
        rots[:, 0] = 1
        opac = inverse_sigmoid((1.0 / F) * torch.ones(n, 1, device=dev))
        self._xyz = nn.Parameter(pts.requires_grad_(True))
        self._features_dc = nn.Parameter(feats[:, :, 0:1].transpose(1, 2).contiguous())
        self._features_rest = nn.Parameter(feats[:, :, 1:].transpose(1, 2).contiguous())
        self._scaling = nn.Parameter(scales)
        self._rotation = nn.Parameter(rots)
        self._opacity = nn.Parameter(opac)
        self._conf_static = nn.Parameter((1 - dyna_avg.float()).contiguous())
        self.Q = nn.Parameter(w2c_pose7[:, :4].clone().contiguous())
        self.T = nn.Parameter(w2c_pose7[:, 4:].clone().contiguous())
        return self

    # ---- optimizers and schedules
    def training_setup(self, opt: OptimParams, fused=False):
        s = self.spatial_lr_scale
        groups = [
            {"params": [self._xyz], "lr": opt.position_lr_init * s, "name": "xyz"},
            {"params": [self._features_dc], "lr": opt.feature_lr, "name": "f_dc"},
            {"params": [self._features_rest], "lr": opt.feature_lr / 20.0, "name": "f_rest"},
            {"params": [self._opacity], "lr": opt.opacity_lr, "name": "opacity"},
            {"params": [self._scaling], "lr": opt.scaling_lr, "name": "scaling"},
            {"params": [self._rotation], "lr": opt.rotation_lr, "name": "rotation"},
            {"params": [self._conf_static], "lr": 3e-3, "name": "conf_static"},
        ]
        cam = [{"params": [self.Q], "lr": 0.00003, "name": "pose_Q"}, {"params": [self.T], "lr": 0.00003, "name": "pose_T"}]
        if self.FoVx is not None:   # gaussian_model.py:253-254 (inert: module docstring)
            cam += [{"params": [self.FoVx], "lr": 0.0001, "name": "fovX"}, {"params": [self.FoVy], "lr": 0.0001, "name": "fovY"}]
        cam_test = None
        if self.enable_test:        # gaussian_model.py:263-268 (built, never stepped: module docstring)
            cam_test = [{"params": [self.test_Q], "lr": 0.00003, "name": "test_pose_Q"},
                        {"params": [self.test_T], "lr": 0.00003, "name": "test_pose_T"}]
        if fused:   # opt-in: one HIP launch per step, SH coefficients swept only up to the active degree
            from .fused import FusedAdam
            groups[2]["sh_rest"] = True
            self.optimizer = FusedAdam(groups, lr=0.0, eps=1e-15)
            self.optimizer_cam = FusedAdam(cam[:2], lr=0.0, eps=1e-15)   # stepped with a device-side PSNR gate (train.py); the
            #                                                                inert FoV groups have nothing to fuse
        else:
            self.optimizer = torch.optim.Adam(groups, lr=0.0, eps=1e-15)
            self.optimizer_cam = torch.optim.Adam(cam, lr=0.0, eps=1e-15)
        if cam_test is not None:
            self.optimizer_cam_test = torch.optim.Adam(cam_test, lr=0.0, eps=1e-15)
        self._lr_xyz = expon_lr_func(opt.position_lr_init * s, opt.position_lr_final * s, lr_delay_mult=opt.position_lr_delay_mult,
                                     max_steps=opt.position_lr_max_steps)
        self._lr_cam = expon_lr_func(0.00003, 0.000003, lr_delay_mult=opt.position_lr_delay_mult, max_steps=1000)
        self._lr_conf = expon_lr_func(3e-3, 3e-4, lr_delay_mult=opt.position_lr_delay_mult, max_steps=opt.iterations)
        if hasattr(self.optimizer, "set_active_sh_degree"):
            self.optimizer.set_active_sh_degree(self.active_sh_degree)

    # ---- checkpoints (scene/gaussian_model.py:66-101 capture / restore; written by train_gui.py:626-628 as (capture(), iteration))
    def capture(self):
        """The reference's fourteen fields in the reference's order — (capture(), iteration) is what its restore() unpacks — so the
        file loads on either side.  max_radii2D / xyz_gradient_accum / denom are the densification statistics, which DAS3R never
        updates (the block is commented out: SURVEY.md C1): empty tensors, as the reference's own are before training_setup."""
        e = torch.empty(0, device=self._xyz.device)
        return (self.active_sh_degree, self._xyz, self._features_dc, self._features_rest, self._scaling, self._rotation, self._opacity,
                e, e, e, self.optimizer.state_dict(), self.spatial_lr_scale, self.Q, self.T)

    def capture_extras(self):
        """What the reference's capture() leaves out and a resumed job needs to continue EXACTLY (its checkpoints omit
        _conf_static, the camera optimizer and the held-out poses — a resumed reference run restarts those from their initial values)."""
        return dict(conf_static=self._conf_static, aggregated_mask=self.aggregated_mask, optimizer_cam=self.optimizer_cam.state_dict(),
                    test_Q=self.test_Q, test_T=self.test_T, FoVx=self.FoVx, FoVy=self.FoVy, max_sh_degree=self.max_sh_degree)

    def restore(self, model_args, opt, extras=None, fused=False, device=None):
        """Counterpart of GaussianModel.restore(model_args, training_args).  Also accepts a capture written by the reference (its
        optimizer state is torch.optim.Adam's, which FusedAdam reads).  extras: capture_extras() of the same moment, when there is one."""
        (self.active_sh_degree, xyz, f_dc, f_rest, scaling, rotation, opacity, _mr, _acc, _den, opt_dict, self.spatial_lr_scale, Q, T) = model_args
        dev = torch.device(device) if device is not None else xyz.device
        par = lambda t: nn.Parameter(t.detach().to(dev, torch.float32).clone().contiguous())
        self._xyz, self._features_dc, self._features_rest = par(xyz), par(f_dc), par(f_rest)
        self._scaling, self._rotation, self._opacity = par(scaling), par(rotation), par(opacity)
        self.Q, self.T = par(Q), par(T)
        if extras is not None:
            self._conf_static = par(extras["conf_static"])
            self.aggregated_mask = extras["aggregated_mask"].to(dev)
            self.max_sh_degree = extras.get("max_sh_degree", self.max_sh_degree)
            if extras.get("test_Q") is not None:
                self.enable_test = True
                self.test_Q = extras["test_Q"].detach().to(dev).clone().requires_grad_(True)
                self.test_T = extras["test_T"].detach().to(dev).clone().requires_grad_(True)
            if extras.get("FoVx") is not None:
                self.FoVx = extras["FoVx"].detach().to(dev).clone().requires_grad_(True)
                self.FoVy = extras["FoVy"].detach().to(dev).clone().requires_grad_(True)
        elif not hasattr(self, "_conf_static"):
            raise ValueError("SplatModel.restore: a reference checkpoint does not hold _conf_static / aggregated_mask (SURVEY.md C10): "
                             "build the model from its sequence first (create_from_frames), then restore into it")
        self.training_setup(opt, fused=fused)
        from .fused import remap_state_dict

        def load(optimizer, sd):   # (torch.optim.Adam insists on the same groups: a checkpoint of the other optimizer kind is re-keyed by group name)
            same = [len(g["params"]) for g in sd["param_groups"]] == [len(g["params"]) for g in optimizer.param_groups]
            sd = sd if same else remap_state_dict(sd, optimizer)
            if isinstance(optimizer, torch.optim.Optimizer):   # hyper-parameters torch needs in every group (a FusedAdam checkpoint holds lr / betas / eps)
                for g, tg in zip(sd["param_groups"], optimizer.param_groups):
                    for k, v in tg.items():
                        if k != "params":
                            g.setdefault(k, v)
            optimizer.load_state_dict(sd)

        load(self.optimizer, opt_dict)
        if extras is not None and extras.get("optimizer_cam") is not None:
            load(self.optimizer_cam, extras["optimizer_cam"])
        if hasattr(self.optimizer, "set_active_sh_degree"):
            self.optimizer.set_active_sh_degree(self.active_sh_degree)
        return self

    def update_learning_rate(self, iteration):
        for g in self.optimizer_cam.param_groups:
            if g["name"] in ("pose_Q", "pose_T", "test_pose_Q", "test_pose_T"):   # (the test names never occur in this optimizer:
                g["lr"] = self._lr_cam(iteration)                                  #  gaussian_model.py:302-315 looks for them all the same)
        for g in self.optimizer.param_groups:
            if g["name"] == "xyz":
                g["lr"] = self._lr_xyz(iteration)
            elif g["name"] == "conf_static":
                g["lr"] = self._lr_conf(iteration)
