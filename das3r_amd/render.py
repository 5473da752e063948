"""das3r_render — the repo's counterpart of DAS3R's render() (/root/reference/gaussian_renderer/__init__.py:23-149,
SURVEY.md §8 a1).  The reference file never travels to the GPU box, so the harness carries this restatement; an
unmodified DAS3R checkout keeps using its own render() on top of the drop-in `diff_gaussian_rasterization` package.

Reproduced exactly: dummy means2D leaf that receives dL/d(mean2D) (:41-50); tanfov = tan(FoV/2) (:53-54);
viewmatrix = I4, projmatrix = I4 @ P^T, campos = 0 (:57-61); the 12 settings (:62-78); Gaussians moved into the camera
frame in PyTorch — means3D = (rel_w2c @ [xyz,1]^T)^T[:, :3], rotations = quadmultiply(pose[:4], _rotation) (:83-93);
opacity = sigmoid(_opacity) * conf_static.reshape(-1,1)[aggregated_mask] (:95-97); scales = exp(_scaling) (:107);
shs = cat(f_dc, f_rest) (:126); returned dict keys (:144-149).
"""
import math

import torch

from .camera import camera_from_tensor, quat_multiply
from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer


def das3r_render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None, camera_pose=None,
                 filtering=None, use_conf=True, fused=False):
    """viewpoint_camera: .FoVx .FoVy .image_height .image_width .projection_matrix (4x4, already transposed);
    pc: splat model (das3r_amd.model.SplatModel or anything with the same attributes); pipe: .debug
    .compute_cov3D_python .convert_SHs_python; camera_pose: (7,) tensor (qw,qx,qy,qz,tx,ty,tz), may require grad."""
    xyz = pc.get_xyz
    device = xyz.device
    all_pass = filtering is None
    if filtering is None:
        filtering = torch.ones(xyz.shape[0], dtype=torch.bool, device=device)
    screenspace_points = torch.zeros_like(xyz[filtering], dtype=xyz.dtype, requires_grad=True, device=device) + 0
    try:
        screenspace_points.retain_grad()
    except Exception:  # noqa: BLE001
        pass

    tanfovx = math.tan(float(viewpoint_camera.FoVx) * 0.5)
    tanfovy = math.tan(float(viewpoint_camera.FoVy) * 0.5)
    w2c = torch.eye(4, device=device)
    projmatrix = w2c.unsqueeze(0).bmm(viewpoint_camera.projection_matrix.to(device).unsqueeze(0)).squeeze(0)
    camera_pos = w2c.inverse()[3, :3]
    raster_settings = GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height), image_width=int(viewpoint_camera.image_width), tanfovx=tanfovx,
        tanfovy=tanfovy, bg=bg_color, scale_modifier=scaling_modifier, viewmatrix=w2c, projmatrix=projmatrix,
        sh_degree=pc.active_sh_degree, campos=camera_pos, prefiltered=False, debug=bool(getattr(pipe, "debug", False)))
    rasterizer = GaussianRasterizer(raster_settings=raster_settings)

    if fused and override_color is None and not getattr(pipe, "compute_cov3D_python", False) and use_conf and all_pass:
        # opt-in: the pre-transform + activations below as ONE HIP kernel (+ one for its backward)
        from .fused import pretransform
        idx = getattr(pc, "_mask_index", None)
        if idx is None:
            idx = pc._mask_index = torch.nonzero(pc.aggregated_mask.reshape(-1), as_tuple=False).reshape(-1).contiguous()
        means3D, rotations, scales, opacity = pretransform(pc._xyz, pc._rotation, pc._scaling, pc._opacity, pc._conf_static, idx,
                                                           camera_pose)
        # while the active SH degree is 0 (iterations < 3000 of DAS3R's 4000) only the DC coefficient is read: hand the
        # rasterizer the [P, 1, 3] DC tensor itself (M = 1) instead of cat(f_dc, f_rest) — same image bit for bit, 12 instead
        # of 192 bytes of SH per splat each way, and f_rest gets no gradient at all (Adam skips it)
        shs = pc._features_dc if pc.active_sh_degree == 0 else pc.get_features
        rendered_image, radii = rasterizer(means3D=means3D, means2D=screenspace_points, shs=shs, colors_precomp=None,
                                           opacities=opacity, scales=scales, rotations=rotations, cov3D_precomp=None)
        return {"render": rendered_image, "viewspace_points": screenspace_points, "visibility_filter": radii > 0, "radii": radii}

    rel_w2c = camera_from_tensor(camera_pose)
    gaussians_xyz = pc._xyz.clone()[filtering]
    gaussians_rot = pc._rotation.clone()[filtering]
    xyz_ones = torch.ones(gaussians_xyz.shape[0], 1, device=device).float()
    xyz_homo = torch.cat((gaussians_xyz, xyz_ones), dim=1)
    means3D = (rel_w2c @ xyz_homo.T).T[:, :3]
    gaussians_rot_trans = quat_multiply(camera_pose[:4], gaussians_rot)
    means2D = screenspace_points

    opacity = pc.get_opacity[filtering]
    if use_conf:
        opacity = opacity * pc._conf_static.reshape(-1, 1)[pc.aggregated_mask]

    scales = rotations = cov3D_precomp = None
    if getattr(pipe, "compute_cov3D_python", False):
        cov3D_precomp = pc.get_covariance(scaling_modifier)
    else:
        scales = pc.get_scaling[filtering]
        rotations = gaussians_rot_trans

    shs = colors_precomp = None
    if override_color is None:
        shs = pc.get_features[filtering]
    else:
        colors_precomp = override_color

    rendered_image, radii = rasterizer(means3D=means3D, means2D=means2D, shs=shs, colors_precomp=colors_precomp,
                                       opacities=opacity, scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp)
    return {"render": rendered_image, "viewspace_points": screenspace_points, "visibility_filter": radii > 0, "radii": radii}
