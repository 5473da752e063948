#!/usr/bin/env python3
"""Per-block list shares of a bench workload's tiles (ds / dsc / c4): how unevenly a round's entries fall on the sixteen 4x4 blocks of a
tile, and what the compositing kernels' decompositions pay for it (the models of tools/probes/train_lists.py on make_workload scenes).
    python tools/probes/workload_blocks.py dsc,ds"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from das3r_amd import GaussianRasterizationSettings, _lib
from das3r_amd.rasterizer import _forward_full
from das3r_amd.synth import make_workload

for name in sys.argv[1].split(","):
    sc = make_workload(name).to("cuda")
    rs = GaussianRasterizationSettings(**sc.settings_kwargs())
    e = torch.empty(0, device="cuda")
    with torch.no_grad():
        I, color, radii, geom, binning, img, cap = _forward_full(rs, sc.means3D, sc.shs, e, sc.opacities, sc.scales, sc.rotations, e, exact=True)
    W, H, P = sc.W, sc.H, sc.P
    L = _lib.layout(P, I, W, H)
    tx, ty = (W + 15) // 16, (H + 15) // 16
    nt = tx * ty
    rg = img[L["ranges"]:L["ranges"] + 8 * nt].view(torch.int32).reshape(nt, 2).long()
    ln = (rg[:, 1] - rg[:, 0])
    nc = img[L["n_contrib"]:L["n_contrib"] + 4 * W * H].view(torch.int32).long().reshape(H, W)
    pl = binning[L["point_list"]:L["point_list"] + 4 * I].view(torch.int32).long()
    xyh = _lib.splat_field(geom, L, "xy", P)[pl]
    tile_of = torch.repeat_interleave(torch.arange(nt, device="cuda"), ln)
    pos = torch.arange(I, device="cuda") - rg[tile_of, 0]
    bx, by = (tile_of % tx).float() * 16, (tile_of // tx).float() * 16
    pad = torch.zeros(ty * 16, tx * 16, dtype=torch.long, device="cuda")
    pad[:H, :W] = nc
    blk_last = pad.reshape(ty, 4, 4, tx, 4, 4).permute(0, 3, 1, 4, 2, 5).reshape(nt, 16, 16).max(2).values    # [tile, block]: last contributor
    hits = []
    for r in range(16):
        cx, cy = bx + (r % 4) * 4 + 1.5, by + (r // 4) * 4 + 1.5
        hits.append(((xyh[:, 0] - cx).abs() <= xyh[:, 2] + 1.5) & ((xyh[:, 1] - cy).abs() <= xyh[:, 3] + 1.5))
    hits = torch.stack(hits, 1)
    live_blk = hits & (pos.unsqueeze(1) < blk_last[tile_of])          # listed and in front of the block's own last contributor
    print(f"== {name}: P {P} I {I} tiles {nt} mean list {ln.float().mean():.0f} max {ln.max().item()}; blocks hit per entry {hits.float().sum(1).mean():.2f}, "
          f"of them in front of the block's last contributor {live_blk.float().sum(1).mean():.2f}")
    wv = torch.tensor([[0, 1, 4, 5], [2, 3, 6, 7], [8, 9, 12, 13], [10, 11, 14, 15]], device="cuda")
    maxlen = int(ln.max().item())
    for MB in (128, 192):
        nbt = (maxlen + MB - 1) // MB + 1
        batch = tile_of * nbt + pos // MB
        per = torch.zeros(nt * nbt, 16, dtype=torch.long, device="cuda").index_add_(0, batch, live_blk.long()).float()
        pw = per[:, wv]
        normal = torch.ceil(pw.max(2).values / 16)
        wide = 1.25 * torch.ceil(pw / 64).sum(2)
        best = torch.minimum(normal, wide)
        ideal = torch.ceil(pw.sum(2) / 64)
        crit = lambda x: x.max(1).values.reshape(nt, nbt).sum(1)
        print(f"   backward, rounds of {MB}: batches per tile: lockstep {normal.sum().item() / nt:.0f}  with wide mode {best.sum().item() / nt:.0f}  perfect packing {ideal.sum().item() / nt:.0f}; "
              f"critical path mean / max tile: lockstep {crit(normal).mean():.0f} / {crit(normal).max():.0f}  wide {crit(best).mean():.0f} / {crit(best).max():.0f}; wide chosen {(wide < normal).float().mean().item():.2f}")
    for B in (512,):
        nbt = (maxlen + B - 1) // B + 1
        batch = tile_of * nbt + pos // B
        per = torch.zeros(nt * nbt, 16, dtype=torch.long, device="cuda").index_add_(0, batch, live_blk.long())
        steps = (per + 3) // 4
        tmax = steps.max(1).values.reshape(nt, nbt).sum(1).float()
        tmean = steps.float().mean(1).reshape(nt, nbt).sum(1)
        print(f"   forward, four lanes per pixel, batches of {B}: steps per tile, slowest block per batch: mean {tmax.mean():.0f} max {tmax.max():.0f};  mean block: mean {tmean.mean():.0f} max {tmean.max():.0f}")
