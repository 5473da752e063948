"""Dense float64 PyTorch-autograd restatement of the splat rasterizer.  TEST INFRASTRUCTURE ONLY.

Independent of raster_oracle.c: no hand-written backward — every gradient comes from torch.autograd on a
(pixels x gaussians) dense formulation of SURVEY.md Appendix A (upstream:cuda_rasterizer/forward.cu).
Feasible only for tiny scenes (P <= a few hundred, image <= ~64x64).  PARITY UNPINNED (see raster_oracle.c).

Upstream quirks reproduced on purpose so that autograd equals the upstream's analytic backward:
  * alpha = min(0.99, o*G) passes gradient straight through the clamp (A.7 "no zeroing").
  * the EWA clamp of t.x/t.z, t.y/t.z: the clamped coordinate is treated as a constant (A.8(3)).
  * hard tile-rect gate, power>0 / alpha<1/255 skips, T<1e-4 stop: non-differentiable masks.
  * means2D enters as pixel_xy += (W/2, H/2) * means2D[:, :2] so that means2D.grad has upstream's units.
"""
import math

import torch

C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
      1.445305721320277, -0.5900435899266435]


def _eval_sh(deg, sh, d):
    """sh: (P, M, 3), d: (P,3) unit.  Basis per /root/reference/utils/sh_utils.py:74-100."""
    x, y, z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    r = C0 * sh[:, 0]
    if deg > 0:
        r = r - C1 * y * sh[:, 1] + C1 * z * sh[:, 2] - C1 * x * sh[:, 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        r = (r + C2[0] * xy * sh[:, 4] + C2[1] * yz * sh[:, 5] + C2[2] * (2 * zz - xx - yy) * sh[:, 6]
             + C2[3] * xz * sh[:, 7] + C2[4] * (xx - yy) * sh[:, 8])
    if deg > 2:
        r = (r + C3[0] * y * (3 * xx - yy) * sh[:, 9] + C3[1] * xy * z * sh[:, 10]
             + C3[2] * y * (4 * zz - xx - yy) * sh[:, 11] + C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 12]
             + C3[4] * x * (4 * zz - xx - yy) * sh[:, 13] + C3[5] * z * (xx - yy) * sh[:, 14]
             + C3[6] * x * (xx - 3 * yy) * sh[:, 15])
    return r


def rasterize_dense(means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                    cov3D_precomp=None, *, image_height, image_width, tanfovx, tanfovy, bg, scale_modifier,
                    viewmatrix, projmatrix, sh_degree, campos, near=0.001, dtype=torch.float64):
    """Returns (color[3,H,W], radii[P], aux dict).  All tensor args may require grad."""
    H, W = int(image_height), int(image_width)
    dev = means3D.device   # runs wherever its inputs live (plain torch ops: on the GPU box a HIP device makes the 4000-step runs feasible)
    cast = lambda t: None if t is None else t.to(device=dev, dtype=dtype)
    means3D, means2D, opacities = cast(means3D), cast(means2D), cast(opacities)
    shs, colors_precomp, scales, rotations, cov3D_precomp = map(cast, (shs, colors_precomp, scales, rotations, cov3D_precomp))
    V = cast(torch.as_tensor(viewmatrix)).reshape(4, 4)
    PM = cast(torch.as_tensor(projmatrix)).reshape(4, 4)
    bg = cast(torch.as_tensor(bg)).reshape(3)
    campos = cast(torch.as_tensor(campos)).reshape(3)
    P = means3D.shape[0]
    if P == 0:
        return torch.zeros(3, H, W, dtype=dtype, device=dev), torch.zeros(0, dtype=torch.int32, device=dev), {}
    ones = torch.ones(P, 1, dtype=dtype, device=dev)
    ph = torch.cat([means3D, ones], 1)
    p_view = ph @ V[:, :3]          # row-vector convention
    p_hom = ph @ PM
    p_w = 1.0 / (p_hom[:, 3] + 1e-7)
    ndc = p_hom[:, :2] * p_w[:, None]
    vis = p_view[:, 2] > near

    if cov3D_precomp is None:
        q = rotations
        r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
        R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                         2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                         2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).reshape(P, 3, 3)
        Mm = R * (scale_modifier * scales)[:, None, :]
        Sigma = Mm @ Mm.transpose(1, 2)
    else:
        c = cov3D_precomp
        Sigma = torch.stack([c[:, 0], c[:, 1], c[:, 2], c[:, 1], c[:, 3], c[:, 4], c[:, 2], c[:, 4], c[:, 5]], 1).reshape(P, 3, 3)

    fx, fy = W / (2.0 * tanfovx), H / (2.0 * tanfovy)
    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    tz = torch.where(vis, p_view[:, 2], torch.ones_like(p_view[:, 2]))
    txtz, tytz = p_view[:, 0] / tz, p_view[:, 1] / tz
    cx = (txtz < -limx) | (txtz > limx)
    cy = (tytz < -limy) | (tytz > limy)
    tx = torch.where(cx, (txtz.clamp(-limx, limx) * tz).detach(), p_view[:, 0])
    ty = torch.where(cy, (tytz.clamp(-limy, limy) * tz).detach(), p_view[:, 1])
    zero = torch.zeros_like(tz)
    J = torch.stack([fx / tz, zero, -(fx * tx) / (tz * tz), zero, fy / tz, -(fy * ty) / (tz * tz)], 1).reshape(P, 2, 3)
    Rcw = V[:3, :3].t()             # column-convention world->camera rotation
    T = J @ Rcw
    cov2 = T @ Sigma @ T.transpose(1, 2)
    a = cov2[:, 0, 0] + 0.3
    b = cov2[:, 0, 1]
    c = cov2[:, 1, 1] + 0.3
    det = a * c - b * b
    vis = vis & (det != 0)
    det_s = torch.where(det != 0, det, torch.ones_like(det))
    conA, conB, conC = c / det_s, -b / det_s, a / det_s
    mid = 0.5 * (a + c)
    lam = mid + torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
    radius = torch.ceil(3.0 * torch.sqrt(lam)).detach()
    px = ((ndc[:, 0] + 1.0) * W - 1.0) * 0.5 + 0.5 * W * means2D[:, 0]
    py = ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5 + 0.5 * H * means2D[:, 1]
    gx, gy = (W + 15) // 16, (H + 15) // 16
    pxd, pyd = px.detach(), py.detach()
    rminx = torch.trunc((pxd - radius) / 16).clamp(0, gx)
    rmaxx = torch.trunc((pxd + radius + 15) / 16).clamp(0, gx)
    rminy = torch.trunc((pyd - radius) / 16).clamp(0, gy)
    rmaxy = torch.trunc((pyd + radius + 15) / 16).clamp(0, gy)
    area = (rmaxx - rminx) * (rmaxy - rminy)
    vis = vis & (area > 0)
    radii = torch.where(vis, radius, torch.zeros_like(radius)).to(torch.int32)

    if colors_precomp is None:
        d = means3D - campos[None]
        d = d / d.norm(dim=1, keepdim=True)
        rgb = torch.clamp_min(_eval_sh(sh_degree, shs, d) + 0.5, 0.0)
    else:
        rgb = colors_precomp

    # global order (depth asc, index asc) restricted per pixel by the tile gate == upstream's per-tile order
    depth = p_view[:, 2].detach().to(torch.float32).to(dtype)   # keys are fp32 depth bits upstream
    order = torch.sort(depth, stable=True).indices                 # stable: equal depths keep index order
    ys, xs = torch.meshgrid(torch.arange(H, dtype=dtype, device=dev), torch.arange(W, dtype=dtype, device=dev), indexing="ij")
    pix_x, pix_y = xs.reshape(-1, 1), ys.reshape(-1, 1)            # (Npix,1)
    tile_x, tile_y = torch.floor(pix_x / 16), torch.floor(pix_y / 16)
    o = order
    gate = (vis[o][None] & (tile_x >= rminx[o][None]) & (tile_x < rmaxx[o][None])
            & (tile_y >= rminy[o][None]) & (tile_y < rmaxy[o][None]))
    dx = px[o][None] - pix_x
    dy = py[o][None] - pix_y
    power = -0.5 * (conA[o][None] * dx * dx + conC[o][None] * dy * dy) - conB[o][None] * dx * dy
    G = torch.exp(torch.clamp(power, max=0.0))
    raw = opacities.reshape(-1)[o][None] * G
    alpha = raw + (torch.clamp(raw, max=0.99) - raw).detach()        # straight-through clamp
    live = gate & (power <= 0) & (alpha.detach() >= 1.0 / 255.0)
    alpha = torch.where(live, alpha, torch.zeros_like(alpha))
    one_m = 1.0 - alpha
    T_after = torch.cumprod(one_m, dim=1)
    T_before = torch.cat([torch.ones_like(T_after[:, :1]), T_after[:, :-1]], 1)
    stop = (live & (T_after.detach() < 1e-4)).to(torch.int64).cumsum(1) > 0   # includes the triggering splat
    keep = live & ~stop
    w = torch.where(keep, alpha * T_before, torch.zeros_like(alpha))
    C = w @ rgb[o]                                                  # (Npix,3)
    T_final = torch.where(keep, one_m, torch.ones_like(one_m)).prod(dim=1)
    out = (C + T_final[:, None] * bg[None]).t().reshape(3, H, W)
    n_contrib = torch.where(keep, torch.arange(1, P + 1, device=dev)[None].expand_as(keep), torch.zeros_like(keep, dtype=torch.long)).max(dim=1).values
    aux = dict(order=order, keep=keep, T_final=T_final.reshape(H, W), px=px, py=py, conic=torch.stack([conA, conB, conC], 1),
               rgb=rgb, depth=p_view[:, 2], vis=vis, n_pairs=int(keep.sum()), n_contrib_global=n_contrib.reshape(H, W))
    return out, radii, aux
