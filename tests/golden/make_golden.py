#!/usr/bin/env python3
"""Generate golden vectors from the reference's importable pure-PyTorch helpers.

Run ONLY in the build container (needs /root/reference); the resulting
``ref_helpers.npz`` is committed and is the only thing that travels to the GPU box.
No reference source is copied: this script imports the reference's functions, feeds them
seeded random inputs and stores inputs + outputs.

What each vector pins:
  sh_*      utils/sh_utils.py:57-112   eval_sh / RGB2SH  -> SH basis + sign convention of the
                                       rasterizer's colour stage (kernel = max(eval_sh+0.5, 0))
  proj_*    utils/graphics_utils.py:80-100 getProjectionMatrix -> projmatrix layout
  w2v_*     utils/graphics_utils.py:47-58  getWorld2View2
  pose_*    utils/pose_utils.py:57-104 get_camera_from_tensor / quadmultiply -> the pre-transform
                                       gaussian_renderer/__init__.py:83-91 applies before the rasterizer
  loss_*    utils/loss_utils.py:17-66, utils/image_utils.py:17-19  l1 / ssim / psnr
  lr_*      utils/general_utils.py:29-62 get_expon_lr_func
  isig_*    utils/general_utils.py:18  inverse_sigmoid
"""
import os
import sys
import types
from unittest.mock import MagicMock

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_helpers.npz")


def _stub(names):
    for n in names:
        parts = n.split(".")
        for i in range(1, len(parts) + 1):
            k = ".".join(parts[:i])
            if k not in sys.modules:
                sys.modules[k] = MagicMock()


def main():
    assert os.path.isdir(REF), "reference checkout not present: run in the build container"
    _stub(["open3d", "plyfile", "simple_knn", "simple_knn._C", "diff_gaussian_rasterization",
           "evo", "evo.core", "evo.core.trajectory", "evo.tools", "evo.core.metrics",
           "evo.tools.plot", "evo.core.geometry", "evo.main_ape", "evo.main_rpe", "evo.tools.file_interface",
           "cv2", "matplotlib", "matplotlib.pyplot", "roma", "imageio", "icecream"])
    sys.path.insert(0, REF)
    from utils.sh_utils import eval_sh, RGB2SH, SH2RGB
    from utils.graphics_utils import getProjectionMatrix, getWorld2View2, focal2fov, fov2focal
    from utils.pose_utils import get_camera_from_tensor, quadmultiply, quad2rotation
    from utils.loss_utils import l1_loss, ssim
    from utils.image_utils import psnr
    from utils.general_utils import get_expon_lr_func, inverse_sigmoid

    g = torch.Generator().manual_seed(20250307)
    out = {}

    # (1) SH evaluation, degrees 0..3; sh laid out [N, C=3, 16] as eval_sh expects.
    N = 64
    sh = torch.randn(N, 3, 16, generator=g)
    dirs = torch.randn(N, 3, generator=g)
    dirs = dirs / dirs.norm(dim=1, keepdim=True)
    out["sh_coeffs"] = sh.numpy()
    out["sh_dirs"] = dirs.numpy()
    for d in range(4):
        out[f"sh_eval_deg{d}"] = eval_sh(d, sh, dirs).numpy()
    rgb = torch.rand(N, 3, generator=g)
    out["sh_rgb_in"] = rgb.numpy()
    out["sh_rgb2sh"] = RGB2SH(rgb).numpy()
    out["sh_sh2rgb"] = SH2RGB(RGB2SH(rgb)).numpy()

    # (2) projection matrices (transposed = the row-vector convention handed to the rasterizer,
    #     scene/cameras.py:91)
    fovs = np.array([[2 * np.arctan(0.8), 2 * np.arctan(0.45)],
                     [1.0, 0.7],
                     [focal2fov(600.0, 512), focal2fov(600.0, 208)]], dtype=np.float64)
    out["proj_fovs"] = fovs
    out["proj_T"] = np.stack([getProjectionMatrix(0.01, 100.0, float(fx), float(fy)).transpose(0, 1).numpy()
                              for fx, fy in fovs])
    out["proj_fov2focal"] = np.array([fov2focal(float(fovs[2, 0]), 512), fov2focal(float(fovs[2, 1]), 208)])

    # world->view from (R, t)
    q = torch.randn(4, 4, generator=g)
    R = quad2rotation(q).numpy().astype(np.float64)
    t = torch.randn(4, 3, generator=g).numpy().astype(np.float64)
    out["w2v_R"] = R
    out["w2v_t"] = t
    out["w2v_out"] = np.stack([getWorld2View2(R[i], t[i]) for i in range(4)])

    # (3) pose pre-transform
    poses = torch.randn(16, 7, generator=g)
    poses[:, :4] = poses[:, :4] / poses[:, :4].norm(dim=1, keepdim=True) * (0.9 + 0.2 * torch.rand(16, 1, generator=g))
    out["pose_in"] = poses.numpy()
    out["pose_w2c"] = np.stack([get_camera_from_tensor(poses[i]).numpy() for i in range(16)])
    gq = torch.randn(16, 32, 4, generator=g)
    out["pose_gq"] = gq.numpy()
    out["pose_quadmul"] = np.stack([quadmultiply(poses[i, :4], gq[i]).numpy() for i in range(16)])

    # (4) losses
    a = torch.rand(2, 3, 32, 48, generator=g)
    b = (a + 0.1 * torch.randn(2, 3, 32, 48, generator=g)).clamp(0, 1)
    out["loss_a"] = a.numpy()
    out["loss_b"] = b.numpy()
    out["loss_l1"] = np.array([l1_loss(a[i], b[i]).item() for i in range(2)])
    out["loss_l1_map"] = np.stack([l1_loss(a[i], b[i], reduce=False).numpy() for i in range(2)])
    out["loss_ssim"] = np.array([ssim(a[i], b[i]).item() for i in range(2)])
    out["loss_ssim_map"] = np.stack([ssim(a[i], b[i], size_average=False).numpy() for i in range(2)])
    out["loss_psnr"] = np.stack([psnr(a[i], b[i]).numpy() for i in range(2)])

    # (5) LR schedule + inverse sigmoid
    steps = np.array([0, 1, 10, 100, 500, 999, 1000, 2999, 3000, 4000, 30000, 40000])
    out["lr_steps"] = steps
    f_xyz = get_expon_lr_func(lr_init=1.6e-4 * 3.7, lr_final=1.6e-6 * 3.7, lr_delay_mult=0.01, max_steps=30000)
    f_conf = get_expon_lr_func(lr_init=3e-3, lr_final=3e-4, max_steps=4000)
    f_cam = get_expon_lr_func(lr_init=3e-5, lr_final=3e-6, max_steps=1000)
    f_delay = get_expon_lr_func(lr_init=1e-2, lr_final=1e-4, lr_delay_steps=500, lr_delay_mult=0.01, max_steps=4000)
    out["lr_xyz"] = np.array([f_xyz(int(s)) for s in steps])
    out["lr_conf"] = np.array([f_conf(int(s)) for s in steps])
    out["lr_cam"] = np.array([f_cam(int(s)) for s in steps])
    out["lr_delay"] = np.array([f_delay(int(s)) for s in steps])
    x = torch.rand(32, generator=g) * 0.98 + 0.01
    out["isig_in"] = x.numpy()
    out["isig_out"] = inverse_sigmoid(x).numpy()

    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
