"""Camera / pose maths around the rasterizer boundary (host side, plain PyTorch).

Counterparts of the reference helpers that DAS3R's render() composes with the rasterizer; each is pinned
by tests/golden/ref_helpers.npz (tests/test_golden_helpers.py):
  projection_matrix      /root/reference/utils/graphics_utils.py:80-100  getProjectionMatrix
  world2view             /root/reference/utils/graphics_utils.py:47-58   getWorld2View2
  quat_to_rotation       /root/reference/utils/pose_utils.py:40-55       quad2rotation (normalises)
  camera_from_tensor     /root/reference/utils/pose_utils.py:57-84       get_camera_from_tensor
  quat_multiply          /root/reference/utils/pose_utils.py:86-104      quadmultiply
"""
import math

import torch


def focal2fov(focal, pixels):
    return 2 * math.atan(pixels / (2 * focal))


def fov2focal(fov, pixels):
    return pixels / (2 * math.tan(fov / 2))


def projection_matrix(znear, zfar, fovX, fovY):
    """OpenGL-style perspective matrix (column-vector convention, z_sign=+1).  The rasterizer is handed its
    TRANSPOSE (scene/cameras.py:91)."""
    ty, tx = math.tan(fovY / 2), math.tan(fovX / 2)
    top, right = ty * znear, tx * znear
    bottom, left = -top, -right
    P = torch.zeros(4, 4)
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def world2view(R, t, translate=(0.0, 0.0, 0.0), scale=1.0):
    """R: camera-to-world rotation (3,3), t: world-to-camera translation (COLMAP convention) -> 4x4 float32."""
    R = torch.as_tensor(R, dtype=torch.float64)
    t = torch.as_tensor(t, dtype=torch.float64)
    Rt = torch.zeros(4, 4, dtype=torch.float64)
    Rt[:3, :3] = R.t()
    Rt[:3, 3] = t
    Rt[3, 3] = 1.0
    C2W = torch.linalg.inv(Rt)
    C2W[:3, 3] = (C2W[:3, 3] + torch.as_tensor(translate, dtype=torch.float64)) * scale
    return torch.linalg.inv(C2W).to(torch.float32)


def quat_to_rotation(q):
    """(N,4) (w,x,y,z) -> (N,3,3); normalises the quaternion first, like the reference's pose path."""
    q = q / q.norm(p=2, dim=1, keepdim=True)
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    rot = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                       2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                       2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1)
    return rot.reshape(-1, 3, 3)


def camera_from_tensor(pose):
    """pose (7,) = (qw,qx,qy,qz,tx,ty,tz) -> 4x4 world-to-camera (differentiable)."""
    if pose.dim() == 1:
        pose = pose.unsqueeze(0)
    quad, T = pose[:, :4], pose[:, 4:]
    R = quat_to_rotation(quad)[0]
    top = torch.cat([R, T.reshape(3, 1)], dim=1)
    bottom = torch.tensor([[0.0, 0.0, 0.0, 1.0]], dtype=pose.dtype, device=pose.device)
    return torch.cat([top, bottom], dim=0).float()


def quat_multiply(q1, q2):
    w1, x1, y1, z1 = q1.unbind(dim=-1)
    w2, x2, y2, z2 = q2.unbind(dim=-1)
    return torch.stack([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2,
                        w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                        w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
                        w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2], dim=-1)


def world_camera(R, t, fovx, fovy, image_height, image_width, device="cpu", znear=0.01, zfar=100.0, translate=(0.0, 0.0, 0.0), scale=1.0):
    """The geometric half of the reference's Camera (scene/cameras.py:84-100): world_view_transform = getWorld2View2(R, T)^T,
    projection_matrix = getProjectionMatrix(...)^T, full_proj_transform = their product, camera_center = row 3 of the inverse
    view transform — what the vanilla 3DGS renderer (das3r_render_3dgs) reads from its viewpoint camera."""
    from types import SimpleNamespace
    wvt = world2view(R, t, translate, scale).transpose(0, 1).to(device)
    proj = projection_matrix(znear, zfar, fovx, fovy).transpose(0, 1).to(device)
    cam = SimpleNamespace(FoVx=fovx, FoVy=fovy, image_height=int(image_height), image_width=int(image_width), znear=znear, zfar=zfar,
                          world_view_transform=wvt, projection_matrix=proj,
                          full_proj_transform=(wvt.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0), camera_center=wvt.inverse()[3, :3])
    cam.get_projection_matrix = lambda fx, fy: projection_matrix(znear, zfar, float(fx), float(fy)).transpose(0, 1).to(device)
    return cam
