"""das3r_render — the repo's counterpart of DAS3R's render() (/root/reference/gaussian_renderer/__init__.py:23-149,
SURVEY.md §8 a1).  The reference file never travels to the GPU box, so the harness carries this restatement; an
unmodified DAS3R checkout keeps using its own render() on top of the drop-in `diff_gaussian_rasterization` package.

Organised around what reaches the rasterizer — `rasterizer_inputs()` builds the settings and the eight tensor arguments, the
fused (one HIP kernel, §8f-1) and the PyTorch pre-transform being two producers of the same tuple — and pinned against the
reference itself: tests/golden/ref_render.npz holds what the reference's render() handed to a recording rasterizer for seeded
models in all three `pipe` modes, and tests/test_golden_render.py replays them through this file.

What the reference does, kept as is (quirks included):
  * the camera is the identity (viewmatrix = I4, projmatrix = I4 @ P^T, campos = 0, :57-61); the Gaussians are moved instead:
    means3D = (w2c(pose) [xyz 1]^T)^T[:, :3], rotations = quadmultiply(pose[:4], _rotation) (:83-93);
  * opacity = sigmoid(_opacity)[filtering] * conf_static.reshape(-1, 1)[aggregated_mask] (:95-97) — two different index sets,
    consistent only for the default filtering (C4);
  * pipe.compute_cov3D_python: cov3D_precomp = pc.get_covariance(scaling_modifier), built from the WORLD-frame _rotation
    (normalised) while the means are in the camera frame (:103-104);
  * pipe.convert_SHs_python: colours = clamp_min(eval_sh(active degree, features, normalise(get_xyz - camera_center)) + 0.5, 0)
    from the WORLD-frame positions and the camera's own centre, not filtered (:112-122);
  * a dummy means2D leaf (zeros + 0, retain_grad) that receives dL/d(mean2D) (:41-50); returned dict keys (:144-149).
"""
import math

import torch

from .camera import camera_from_tensor, projection_matrix, quat_multiply
from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer

_SH_C0 = 0.28209479177387814
_SH_C1 = 0.4886025119029199
_SH_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
_SH_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
          1.445305721320277, -0.5900435899266435)


def sh_to_rgb(degree, sh, dirs):
    """Real spherical harmonics up to `degree` (basis and signs of /root/reference/utils/sh_utils.py:57-112, pinned by
    tests/golden/ref_helpers.npz).  sh [..., C, (max_degree + 1)^2], dirs [..., 3] unit -> [..., C]."""
    x, y, z = dirs[..., 0:1], dirs[..., 1:2], dirs[..., 2:3]
    out = _SH_C0 * sh[..., 0]
    if degree >= 1:
        out = out - _SH_C1 * y * sh[..., 1] + _SH_C1 * z * sh[..., 2] - _SH_C1 * x * sh[..., 3]
    if degree >= 2:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        out = (out + _SH_C2[0] * xy * sh[..., 4] + _SH_C2[1] * yz * sh[..., 5] + _SH_C2[2] * (2.0 * zz - xx - yy) * sh[..., 6]
               + _SH_C2[3] * xz * sh[..., 7] + _SH_C2[4] * (xx - yy) * sh[..., 8])
    if degree >= 3:
        out = (out + _SH_C3[0] * y * (3 * xx - yy) * sh[..., 9] + _SH_C3[1] * xy * z * sh[..., 10]
               + _SH_C3[2] * y * (4 * zz - xx - yy) * sh[..., 11] + _SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[..., 12]
               + _SH_C3[4] * x * (4 * zz - xx - yy) * sh[..., 13] + _SH_C3[5] * z * (xx - yy) * sh[..., 14]
               + _SH_C3[6] * x * (xx - 3 * yy) * sh[..., 15])
    return out


def covariance_from_scaling_rotation(scaling, scaling_modifier, rotation):
    """pc.get_covariance of the reference (scene/gaussian_model.py:32-36 + utils/general_utils.py:62-110): Sigma = (R S)(R S)^T
    with R from the NORMALISED quaternion, as the 6 upper-triangular entries (xx, xy, xz, yy, yz, zz)."""
    q = rotation / rotation.norm(dim=1, keepdim=True)
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                     2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                     2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).reshape(-1, 3, 3)
    L = R * (scaling_modifier * scaling)[:, None, :]
    S = L @ L.transpose(1, 2)
    return torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], 1)


def _settings(cam, pc, pipe, bg_color, scaling_modifier, device, model_fov=False):
    ident = torch.eye(4, device=device)
    # render_no_soft (gaussian_renderer/__init__.py:308-319) takes the field of view from the MODEL (pc.FoVx / FoVy, the two inert
    # optimizer groups of scene/gaussian_model.py:165-166,253-254) and rebuilds the projection from it
    fovx, fovy = (pc.FoVx, pc.FoVy) if model_fov else (cam.FoVx, cam.FoVy)
    if model_fov and (fovx is None or fovy is None):
        raise RuntimeError("render_no_soft takes the field of view from the model: call SplatModel.init_fov(FoVx, FoVy) first "
                           "(build_from_sequence does)")
    if model_fov:   # rebuilt from the model's field of view; cameras that carry the reference's method keep theirs
        pm = (cam.get_projection_matrix(fovx, fovy) if hasattr(cam, "get_projection_matrix")
              else projection_matrix(0.01, 100.0, float(fovx), float(fovy)).transpose(0, 1))
    else:
        pm = cam.projection_matrix
    proj = ident.unsqueeze(0).bmm(pm.to(device).unsqueeze(0)).squeeze(0)
    return GaussianRasterizationSettings(
        image_height=int(cam.image_height), image_width=int(cam.image_width), tanfovx=math.tan(float(fovx) * 0.5),
        tanfovy=math.tan(float(fovy) * 0.5), bg=bg_color, scale_modifier=scaling_modifier, viewmatrix=ident, projmatrix=proj,
        sh_degree=pc.active_sh_degree, campos=ident.inverse()[3, :3], prefiltered=False, debug=bool(getattr(pipe, "debug", False)))


VARIANTS = ("render", "test", "no_soft", "confidence")


def rasterizer_inputs(cam, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None, camera_pose=None, filtering=None,
                      use_conf=True, fused=False, variant="render"):
    """-> (GaussianRasterizationSettings, kwargs of GaussianRasterizer.forward): everything render() computes in front of the
    rasterizer.  `variant` selects one of the reference's four renderers, which differ in three places only
    (gaussian_renderer/__init__.py): "render" (:23-149) opacity * conf_static[aggregated_mask]; "test" (render_test, :152-277)
    opacity * conf_static as stored; "no_soft" (render_no_soft, :279-408) no confidence factor, field of view and projection
    from the model's FoVx / FoVy; "confidence" (render_confidence, :410-510) opacity = 1 and the per-Gaussian confidence as a
    grey precomputed colour."""
    assert variant in VARIANTS, variant
    xyz = pc.get_xyz
    device = xyz.device
    everything = filtering is None
    if everything:
        filtering = torch.ones(xyz.shape[0], dtype=torch.bool, device=device)
    means2D = torch.zeros_like(xyz[filtering], dtype=xyz.dtype, requires_grad=True, device=device) + 0
    try:
        means2D.retain_grad()
    except Exception:  # noqa: BLE001
        pass
    settings = _settings(cam, pc, pipe, bg_color, scaling_modifier, device, model_fov=variant == "no_soft")
    cov_python = bool(getattr(pipe, "compute_cov3D_python", False))
    sh_python = bool(getattr(pipe, "convert_SHs_python", False))

    if fused and override_color is None and not cov_python and not sh_python and use_conf and everything and variant == "render":
        # opt-in: pose -> camera frame, quaternion product, exp / sigmoid * conf as ONE HIP kernel each way
        from .fused import pretransform
        idx = getattr(pc, "_mask_index", None)
        if idx is None:
            idx = pc._mask_index = torch.nonzero(pc.aggregated_mask.reshape(-1), as_tuple=False).reshape(-1).contiguous()
        means3D, rotations, scales, opacity = pretransform(pc._xyz, pc._rotation, pc._scaling, pc._opacity, pc._conf_static, idx, camera_pose)
        # while the active SH degree is 0 (iterations < 3000 of DAS3R's 4000) only the DC coefficient is read: the rasterizer gets
        # the [P, 1, 3] DC tensor itself (M = 1) instead of cat(f_dc, f_rest) — same image bit for bit, 12 instead of 192 bytes of
        # SH per splat each way; f_rest then has no gradient (FusedAdam counts its steps all the same: fused.py)
        # — only with the fused optimizer: torch.optim.Adam skips a parameter without gradient, so its step count (and bias
        # correction) for f_rest would restart at the degree bump, where the reference has counted 3000 zero-gradient steps
        optimizer = getattr(pc, "optimizer", None)
        fused_opt = (getattr(optimizer, "is_fused", False) and hasattr(optimizer, "handles_compact_sh")
                     and optimizer.handles_compact_sh(pc._features_rest))   # (f_rest must sit in an "sh_rest" group: ADVICE r3)
        if fused_opt and pc.active_sh_degree == 0:
            shs = pc._features_dc
        elif fused_opt and pc.active_sh_degree < pc.max_sh_degree and getattr(pc, "sh_prefix", True):
            # round 3: between degree 0 and the maximum (iterations 3000 .. of DAS3R's 4000 run at degree 1) the rasterizer gets the
            # coefficients of the ACTIVE degree only — [P, 4, 3] at degree 1 instead of [P, 16, 3]: same image, and the gradient of
            # the active prefix goes to FusedAdam as it is (fused._ShPrefix)
            from .fused import active_sh_prefix
            shs = active_sh_prefix(pc._features_dc, pc._features_rest, pc.active_sh_degree)
        else:
            shs = pc.get_features
        return settings, dict(means3D=means3D, means2D=means2D, shs=shs, colors_precomp=None, opacities=opacity, scales=scales,
                              rotations=rotations, cov3D_precomp=None)

    w2c = camera_from_tensor(camera_pose)
    pts = pc._xyz.clone()[filtering]
    homo = torch.cat((pts, torch.ones(pts.shape[0], 1, device=device).float()), dim=1)
    means3D = (w2c @ homo.T).T[:, :3]
    rot_cam = quat_multiply(camera_pose[:4], pc._rotation.clone()[filtering])
    opacity = pc.get_opacity[filtering]
    if variant == "confidence":
        opacity = torch.ones_like(opacity)
    elif variant == "test":
        opacity = opacity * pc._conf_static
    elif variant == "render" and use_conf:
        opacity = opacity * pc._conf_static.reshape(-1, 1)[pc.aggregated_mask]
    kw = dict(means3D=means3D, means2D=means2D, shs=None, colors_precomp=None, opacities=opacity, scales=None, rotations=None,
              cov3D_precomp=None)
    if cov_python:
        kw["cov3D_precomp"] = (pc.get_covariance(scaling_modifier) if hasattr(pc, "get_covariance")
                               else covariance_from_scaling_rotation(pc.get_scaling, scaling_modifier, pc._rotation))
    else:
        kw["scales"], kw["rotations"] = pc.get_scaling[filtering], rot_cam
    if variant == "confidence":
        kw["colors_precomp"] = pc._conf[filtering].unsqueeze(1).repeat(1, 3)
    elif override_color is not None:
        kw["colors_precomp"] = override_color
    elif sh_python:
        feats = pc.get_features
        per_channel = feats.transpose(1, 2).view(-1, 3, (pc.max_sh_degree + 1) ** 2)
        away = pc.get_xyz - cam.camera_center.to(device).repeat(feats.shape[0], 1)
        kw["colors_precomp"] = torch.clamp_min(sh_to_rgb(pc.active_sh_degree, per_channel, away / away.norm(dim=1, keepdim=True)) + 0.5, 0.0)
    else:
        kw["shs"] = pc.get_features[filtering]
    return settings, kw


def das3r_render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None, camera_pose=None,
                 filtering=None, use_conf=True, fused=False, variant="render"):
    """viewpoint_camera: .FoVx .FoVy .image_height .image_width .projection_matrix (4x4, already transposed) [.camera_center for
    pipe.convert_SHs_python]; pc: splat model (das3r_amd.model.SplatModel or anything with the same attributes); pipe: .debug
    .compute_cov3D_python .convert_SHs_python; camera_pose: (7,) tensor (qw,qx,qy,qz,tx,ty,tz), may require grad."""
    settings, kw = rasterizer_inputs(viewpoint_camera, pc, pipe, bg_color, scaling_modifier, override_color, camera_pose, filtering,
                                     use_conf, fused, variant)
    image, radii = GaussianRasterizer(raster_settings=settings)(**kw)
    if variant == "confidence":
        return image   # (render_confidence returns the image alone: gaussian_renderer/__init__.py:510)
    return {"render": image, "viewspace_points": kw["means2D"], "visibility_filter": radii > 0, "radii": radii}


def das3r_render_3dgs(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None):
    """The vanilla 3DGS renderer the reference keeps beside its own (gaussian_renderer/__init__3dgs.py:18-99): Gaussians in WORLD
    space, the camera in the settings (world_view_transform / full_proj_transform / camera_center), no pose tensor, no
    confidence factor, rotations normalised by the model's activation."""
    device = pc.get_xyz.device
    means2D = torch.zeros_like(pc.get_xyz, dtype=pc.get_xyz.dtype, requires_grad=True, device=device) + 0
    try:
        means2D.retain_grad()
    except Exception:  # noqa: BLE001
        pass
    settings = GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height), image_width=int(viewpoint_camera.image_width),
        tanfovx=math.tan(float(viewpoint_camera.FoVx) * 0.5), tanfovy=math.tan(float(viewpoint_camera.FoVy) * 0.5), bg=bg_color,
        scale_modifier=scaling_modifier, viewmatrix=viewpoint_camera.world_view_transform, projmatrix=viewpoint_camera.full_proj_transform,
        sh_degree=pc.active_sh_degree, campos=viewpoint_camera.camera_center, prefiltered=False, debug=bool(getattr(pipe, "debug", False)))
    kw = dict(means3D=pc.get_xyz, means2D=means2D, shs=None, colors_precomp=None, opacities=pc.get_opacity, scales=None, rotations=None,
              cov3D_precomp=None)
    if bool(getattr(pipe, "compute_cov3D_python", False)):
        kw["cov3D_precomp"] = pc.get_covariance(scaling_modifier)
    else:
        kw["scales"], kw["rotations"] = pc.get_scaling, pc.get_rotation
    if override_color is not None:
        kw["colors_precomp"] = override_color
    elif bool(getattr(pipe, "convert_SHs_python", False)):
        feats = pc.get_features
        per_channel = feats.transpose(1, 2).view(-1, 3, (pc.max_sh_degree + 1) ** 2)
        away = pc.get_xyz - viewpoint_camera.camera_center.repeat(feats.shape[0], 1)
        kw["colors_precomp"] = torch.clamp_min(sh_to_rgb(pc.active_sh_degree, per_channel, away / away.norm(dim=1, keepdim=True)) + 0.5, 0.0)
    else:
        kw["shs"] = pc.get_features
    image, radii = GaussianRasterizer(raster_settings=settings)(**kw)
    return {"render": image, "viewspace_points": means2D, "visibility_filter": radii > 0, "radii": radii}


def das3r_render_test(*a, **k):
    """gaussian_renderer/__init__.py:152 render_test"""
    return das3r_render(*a, variant="test", **k)


def das3r_render_no_soft(*a, **k):
    """gaussian_renderer/__init__.py:279 render_no_soft"""
    return das3r_render(*a, variant="no_soft", **k)


def das3r_render_confidence(*a, **k):
    """gaussian_renderer/__init__.py:410 render_confidence"""
    return das3r_render(*a, variant="confidence", **k)
