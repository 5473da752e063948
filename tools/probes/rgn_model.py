#!/usr/bin/env python3
"""Model of render_bwd_rgn.hip on the tile lists of a DAS3R-shaped training forward: pair slots per instance and the rounds' critical path
(slowest wave) for the ways a tile's 64 regions of 2x2 pixels can be dealt to 4 waves x 4 passes x 4 DPP rows, per round size MB.
    python tools/probes/rgn_model.py smooth,noise,consistent"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from types import SimpleNamespace
from das3r_amd import _lib
from das3r_amd.rasterizer import _forward_full
from das3r_amd.render import rasterizer_inputs
from das3r_amd.train import build_from_sequence, consistent_sequence, synthetic_sequence
from das3r_amd.model import OptimParams

def geometries():
    """name -> int array [4 waves, 4 passes, 4 rows] of region ids (8 ry + rx)"""
    g = {}
    a = np.zeros((4, 4, 4), int)
    for w in range(4):
        for B in range(4):
            for r in range(4):
                a[w, B, r] = 8 * (4 * (w >> 1) + B) + 4 * (w & 1) + r
    g["strips, wave = quadrant"] = a.copy()
    for w in range(4):
        for B in range(4):
            y8 = 2 * B + (w & 1); h = ((w - y8) & 3) >> 1
            for r in range(4):
                a[w, B, r] = 8 * y8 + 4 * h + r
    g["strips, interleaved"] = a.copy()
    for w in range(4):
        for B in range(4):
            by4 = B; bx4 = (w - 2 * B) & 3
            for r in range(4):
                a[w, B, r] = 8 * (2 * by4 + (r >> 1)) + 2 * bx4 + (r & 1)
    g["2x2 blocks of regions, interleaved"] = a.copy()
    for w in range(4):
        for B in range(4):
            for r in range(4):   # columns: a pass = a vertical strip of four regions
                x8 = 2 * B + (w & 1); h = ((w - x8) & 3) >> 1
                a[w, B, r] = 8 * (4 * h + r) + x8
    g["vertical strips, interleaved"] = a.copy()
    return g

for depth in sys.argv[1].split(","):
    if depth == "consistent":
        seq = consistent_sequence(frames=22, W=512, H=208, focal=600.0, n_splats=20000, seed=0)
        model, cams, _test = build_from_sequence(seq, heldout=True)
    else:
        seq = synthetic_sequence(frames=20, W=512, H=208, focal=600.0, n_splats=20000, seed=0, depth=depth)
        model, cams = build_from_sequence(seq)
    model.training_setup(OptimParams(iterations=4000), fused=True)
    pipe = SimpleNamespace(debug=False, compute_cov3D_python=False, convert_SHs_python=False)
    bg = torch.zeros(3, device="cuda")
    cam = cams[3]
    pose = model.get_RT(cam.uid) if hasattr(model, "get_RT") else None
    with torch.no_grad():
        rs, kw = rasterizer_inputs(cam, model, pipe, bg, camera_pose=pose, fused=True)
        e = torch.empty(0, device="cuda")
        I, color, radii, geom, binning, img, cap = _forward_full(rs, kw["means3D"], kw["shs"], e, kw["opacities"], kw["scales"], kw["rotations"], e, exact=True)
        torch.cuda.synchronize()
    W, H = 512, 208
    P = kw["means3D"].shape[0]
    L = _lib.layout(P, I, W, H)
    tx, ty = (W + 15) // 16, (H + 15) // 16
    nt = tx * ty
    rg = img[L["ranges"]:L["ranges"] + 8 * nt].view(torch.int32).reshape(nt, 2).long()
    nc = img[L["n_contrib"]:L["n_contrib"] + 4 * W * H].view(torch.int32).long().reshape(H, W)
    pl = binning[L["point_list"]:L["point_list"] + 4 * I].view(torch.int32).long()
    xyh = _lib.splat_field(geom, L, "xy", P)[pl]
    tile_of = torch.repeat_interleave(torch.arange(nt, device="cuda"), rg[:, 1] - rg[:, 0])
    pos = torch.arange(I, device="cuda") - rg[tile_of, 0]
    bx, by = (tile_of % tx).float() * 16, (tile_of // tx).float() * 16
    pad = torch.zeros(ty * 16, tx * 16, dtype=torch.long, device="cuda")
    pad[:H, :W] = nc
    rlast = pad.reshape(ty, 8, 2, tx, 8, 2).permute(0, 3, 1, 4, 2, 5).reshape(nt, 64, 4).max(2).values   # [tile, region]: last contributor
    tlast = rlast.max(1).values
    h2 = []
    for r in range(64):
        cx, cy = bx + (r % 8) * 2 + 0.5, by + (r // 8) * 2 + 0.5
        h2.append(((xyh[:, 0] - cx).abs() <= xyh[:, 2] + 0.5) & ((xyh[:, 1] - cy).abs() <= xyh[:, 3] + 0.5))
    h2 = torch.stack(h2, 1)
    if os.environ.get("OCT"):   # the octagon of render_common.h block_hit_oct at the size of a region: two diagonal slabs on top of the box
        co = _lib.splat_field(geom, L, "conic_opacity", P)[pl]
        ex, ey = (xyh[:, 2] - 0.02) / 1.0005, (xyh[:, 3] - 0.02) / 1.0005
        ex2, ey2 = ex * ex, ey * ey
        txy = -co[:, 1] * ex2 / co[:, 2]
        half, slack = 0.5 * (ex2 + ey2), 2e-6 * (ex2 + ey2) + 1e-3
        hd1 = torch.sqrt((half + txy).clamp_min(0) + slack) * 1.0005 + 0.05
        hd2 = torch.sqrt((half - txy).clamp_min(0) + slack) * 1.0005 + 0.05
        o2 = []
        for r in range(64):
            cx, cy = bx + (r % 8) * 2 + 0.5, by + (r // 8) * 2 + 0.5
            ddx, ddy = xyh[:, 0] - cx, xyh[:, 1] - cy
            o2.append(((ddx + ddy).abs() * 0.70710678 <= hd1 + 0.7081) & ((ddx - ddy).abs() * 0.70710678 <= hd2 + 0.7081))
        before = h2.float().sum(1).mean().item()
        h2 = h2 & torch.stack(o2, 1)
        print(f"   octagon: regions listed per entry {before:.2f} -> {h2.float().sum(1).mean().item():.2f}")
    h2 = h2 & (pos.unsqueeze(1) < rlast[tile_of])        # [I, 64]: listed and in front of the region's last contributor
    walked = pos < tlast[tile_of]                                        # entries the backward stages at all
    nwalk = int(walked.sum())
    print(f"== {depth}: I {I}, staged by the backward {nwalk}; regions listed per staged entry {h2[walked].float().sum(1).mean():.2f} (x 4 = pair slots at perfect fill)")
    # rounds run from the tile's last contributor backwards, bucket by bucket (1024): position within the bucket's replayed part, from its end
    bucket = pos // 1024
    bend = torch.minimum(tlast[tile_of], (bucket + 1) * 1024)             # end of the replayed part of the entry's bucket
    back = bend - 1 - pos                                                 # 0 = first staged
    for MB in (128,):
        rid = (tile_of * 64 + bucket) * 16 + back // MB                   # round id (<= 16 rounds per bucket at MB >= 64)
        rid = torch.where(walked, rid, torch.full_like(rid, nt * 64 * 16))
        per = torch.zeros(nt * 64 * 16 + 1, 64, dtype=torch.long, device="cuda").index_add_(0, rid, h2.long())[:-1]
        used = per.sum(1) > 0
        per = per[used]                                                   # [rounds, 64 regions]
        for name, a in geometries().items():
            lens = per[:, torch.from_numpy(a).cuda().reshape(-1)].reshape(-1, 4, 4, 4)      # [rounds, wave, pass, row]
            batches = (lens.max(3).values + 15) // 16                                           # [rounds, wave, pass]
            per_wave = batches.sum(2)                                                         # [rounds, wave]
            slots = 256 * batches.sum().item()
            crit = per_wave.max(1).values.sum().item()
            free = ((lens + 15) // 16).sum(3).float().div(4).sum().item()                      # rows walking on their own (no lockstep), in wave batches
            print(f"   MB {MB:3d}  {name:36s} slots / instance {slots / I:5.1f}   wave batches {batches.sum().item() / 1e3:7.0f} k   critical path (slowest wave per round) x 4 {4 * crit / 1e3:7.0f} k   rows on their own {free / 1e3:7.0f} k")
