#!/usr/bin/env python3
"""Golden vectors for the on-disk formats of SURVEY.md §8(f)-4, from the reference's own (importable) readers / writers.

Run ONLY in the build container (needs /root/reference); ``ref_formats.npz`` is committed and is what travels.
No reference source is copied: the reference functions are imported, fed seeded inputs, and inputs + outputs are stored
(text files as byte arrays).

  quat_*      scene/colmap_loader.py:43-66     qvec2rotmat / rotmat2qvec
  r2q_*       utils/pose_utils.py:117-181      rotation2quad (pose tensors of the trainable cameras)
  rtq_*       utils/rearrange.py:314-352       R_to_quaternion (images.txt writer)
  tum_*       utils/rearrange.py:251-273       tumpose_to_c2w (original_pose, incl. its quaternion-order quirk)
  cam_txt / img_txt + parsed_*   utils/rearrange.py:275-295 writers, scene/colmap_loader.py:156-178,244-272 readers
  d2p_*       utils/pose_utils.py:572-583      depth_to_pts3d (per-pixel Gaussian initialisation)
"""
import os
import sys
import tempfile
from unittest.mock import MagicMock

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_formats.npz")


def main():
    assert os.path.isdir(REF), "reference checkout not present: run in the build container"
    for n in ["open3d", "plyfile", "simple_knn", "simple_knn._C", "diff_gaussian_rasterization", "evo", "evo.core", "evo.core.trajectory",
              "evo.tools", "evo.core.metrics", "evo.tools.plot", "evo.core.geometry", "evo.main_ape", "evo.main_rpe",
              "evo.tools.file_interface", "cv2", "matplotlib", "matplotlib.pyplot", "roma", "imageio", "icecream", "vo_eval"]:
        parts = n.split(".")
        for i in range(1, len(parts) + 1):
            sys.modules.setdefault(".".join(parts[:i]), MagicMock())
    sys.path.insert(0, REF)
    from scene.colmap_loader import qvec2rotmat, rotmat2qvec, read_extrinsics_text, read_intrinsics_text
    from utils.pose_utils import rotation2quad
    from utils import rearrange
    rng = np.random.default_rng(20250307)
    out = {}
    q = rng.normal(size=(24, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    R = np.stack([qvec2rotmat(v) for v in q])
    out["quat_in"], out["quat_R"] = q, R
    out["quat_back"] = np.stack([rotmat2qvec(m) for m in R])
    out["r2q_out"] = rotation2quad(torch.from_numpy(R).float()).numpy()
    out["rtq_out"] = np.stack([rearrange.R_to_quaternion(m) for m in R])
    tum = np.concatenate([rng.normal(size=(12, 3)), q[:12]], axis=1)       # x y z qw qx qy qz (as the reference stacks them)
    out["tum_in"] = tum
    out["tum_c2w"] = np.stack([rearrange.tumpose_to_c2w(p) for p in tum])
    # writers -> text -> readers
    K = np.tile(np.array([[600.0, 0, 250.0], [0, 610.0, 104.0], [0, 0, 1]], dtype=np.float32), (5, 1, 1))
    K[:, 0, 0] += np.arange(5, dtype=np.float32)
    names = [f"frame_{i:04d}.png" for i in range(5)]
    with tempfile.TemporaryDirectory() as d:
        rearrange.save_colmap_cameras((512, 208), K, os.path.join(d, "cameras.txt"))
        rearrange.save_colmap_images(list(out["tum_c2w"][:5]), os.path.join(d, "images.txt"), names)
        cam_txt = open(os.path.join(d, "cameras.txt"), "rb").read()
        img_txt = open(os.path.join(d, "images.txt"), "rb").read()
        cams = read_intrinsics_text(os.path.join(d, "cameras.txt"))
        imgs = read_extrinsics_text(os.path.join(d, "images.txt"))
    out["fmt_K"] = K
    out["cam_txt"] = np.frombuffer(cam_txt, dtype=np.uint8)
    out["img_txt"] = np.frombuffer(img_txt, dtype=np.uint8)
    out["parsed_cam_wh"] = np.array([[cams[i].width, cams[i].height] for i in sorted(cams)])
    out["parsed_cam_params"] = np.stack([cams[i].params for i in sorted(cams)])
    out["parsed_img_qvec"] = np.stack([imgs[i].qvec for i in sorted(imgs)])
    out["parsed_img_tvec"] = np.stack([imgs[i].tvec for i in sorted(imgs)])
    out["parsed_img_camid"] = np.array([imgs[i].camera_id for i in sorted(imgs)])
    # depth -> world points
    from utils.pose_utils import depth_to_pts3d
    F, H, W = 3, 6, 9
    Kd = torch.tensor([[50.0, 0, W / 2], [0, 50.0, H / 2], [0, 0, 1]]).repeat(F, 1, 1)
    poses = torch.from_numpy(out["tum_c2w"][:F]).float()
    depth = torch.from_numpy(rng.uniform(1.0, 5.0, size=(F, H, W))).float()
    out["d2p_K"], out["d2p_pose"], out["d2p_depth"] = Kd.numpy(), poses.numpy(), depth.numpy()
    out["d2p_pts"] = depth_to_pts3d(Kd, poses, W, H, depth).numpy()
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
