#!/usr/bin/env python3
"""Tile lists of the DAS3R-shaped training scene (tools/train_bench.py's model) per depth kind: list lengths, the pixels' Σalpha
and stops, entries per 4x4 block of a tile (the compositing kernels' row shares).   python tools/probes/train_lists.py smooth"""
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

for depth in sys.argv[1].split(","):
    if depth == "consistent":   # the self-consistent sequence a farm job runs on (all frames see ONE surface)
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
        _lib.pair_counters(True)
        I, color, radii, geom, binning, img, cap = _forward_full(rs, kw["means3D"], kw["shs"], e, kw["opacities"], kw["scales"], kw["rotations"], e, exact=True)
        torch.cuda.synchronize()
        pc = _lib.pair_counters(False)
    from das3r_amd import GaussianRasterizer
    leaves = {k: (v.detach().clone().requires_grad_(True) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in kw.items()}
    _lib.pair_counters(True)
    img2, _ = GaussianRasterizer(raster_settings=rs)(**leaves)
    img2.backward(torch.randn_like(img2))
    torch.cuda.synchronize()
    print(f"   forward + backward pair counters {_lib.pair_counters(False)}")
    W, H = 512, 208
    P = kw["means3D"].shape[0]
    L = _lib.layout(P, I, W, H)
    tx, ty = (W + 15) // 16, (H + 15) // 16
    nt = tx * ty
    rg = img[L["ranges"]:L["ranges"] + 8 * nt].view(torch.int32).reshape(nt, 2).long()
    ln = (rg[:, 1] - rg[:, 0]).cpu().numpy()
    nc = img[L["n_contrib"]:L["n_contrib"] + 4 * W * H].view(torch.int32).cpu().numpy()
    fT = img[L["final_T"]:L["final_T"] + 4 * W * H].view(torch.float32).cpu().numpy()
    print(f"== {depth}: P {P} I {I} tiles {nt} list mean {ln.mean():.0f} pcts 10/50/90/99/max {np.percentile(ln, [10, 50, 90, 99, 100])}")
    print(f"   radii>0 {(radii > 0).sum().item()}  n_contrib mean {nc.mean():.0f} max {nc.max()}  final_T mean {fT.mean():.4f} min {fT.min():.5f}  stopped pixels (T<1e-4-ish) {(fT < 2e-4).mean():.4f}")
    print(f"   pair counters {pc}")
    # entries per 4x4 block (the kernels' bounding-box test), per 256-entry batch of a tile's list: share of the busiest block
    pl = binning[L["point_list"]:L["point_list"] + 4 * I].view(torch.int32).long()
    xyh = _lib.splat_field(geom, L, "xy", P)[pl]
    tile_of = torch.repeat_interleave(torch.arange(nt, device="cuda"), rg[:, 1] - rg[:, 0])
    pos = torch.arange(I, device="cuda") - rg[tile_of, 0]
    bx, by = (tile_of % tx).float() * 16, (tile_of // tx).float() * 16
    hits = []
    for r in range(16):
        cx, cy = bx + (r % 4) * 4 + 1.5, by + (r // 4) * 4 + 1.5
        hits.append(((xyh[:, 0] - cx).abs() <= xyh[:, 2] + 1.5) & ((xyh[:, 1] - cy).abs() <= xyh[:, 3] + 1.5))
    hits = torch.stack(hits, 1).long()                                   # [I, 16]
    print(f"   blocks hit per entry: mean {hits.sum(1).float().mean():.2f}")
    batch = tile_of * 64 + pos // 256                                     # (lists < 16384)
    per = torch.zeros(nt * 64, 16, dtype=torch.long, device="cuda").index_add_(0, batch, hits)
    used = per.sum(1) > 0
    per = per[used].float()
    # a wave holds blocks {0,1,4,5}, {2,3,6,7}, {8,9,12,13}, {10,11,14,15}: it iterates for its longest row, the workgroup for its slowest wave
    wv = torch.tensor([[0, 1, 4, 5], [2, 3, 6, 7], [8, 9, 12, 13], [10, 11, 14, 15]], device="cuda")
    wmax = per[:, wv].max(2).values                                       # [batches, 4]
    print(f"   per 256-entry batch: mean row share {per.mean():.1f}  mean of the wave's longest row {wmax.mean():.1f}  mean of the workgroup's longest row {wmax.max(1).values.mean():.1f}")
    tot = torch.zeros(nt, 4, device="cuda").index_add_(0, torch.arange(nt * 64, device="cuda")[used] // 64, wmax)
    lock = tot.max(1).values.cpu().numpy()                                # per tile: Σ over batches of the workgroup's slowest wave ≈ its iterations (barrier per batch)
    print(f"   per tile Σ_batches max-wave iterations: mean {lock.mean():.0f} max {lock.max():.0f};   Σ_batches mean-row: {per.mean(1).sum().item() / nt:.0f}")
    for B in (256, 512, 1024):
        nbt = 16384 // B * 2
        batch = tile_of * nbt + pos // B
        per = torch.zeros(nt * nbt, 16, dtype=torch.long, device="cuda").index_add_(0, batch, hits)
        steps = (per + 3) // 4                                            # four-lanes kernel: steps of a block per batch
        tmax = steps.max(1).values.reshape(nt, nbt).sum(1).float()        # a barrier per batch: the workgroup's slowest block
        tmean = steps.float().mean(1).reshape(nt, nbt).sum(1)
        print(f"   four lanes per pixel, batches of {B}: steps per tile, slowest block per batch: mean {tmax.mean():.0f} max {tmax.max():.0f};  mean block: mean {tmean.mean():.0f} max {tmean.max():.0f}")
    # round 6 model: sixteen lanes per pixel — a wave = one 2x2 pixel region x 16 consecutive entries of the region's own list, a workgroup of
    # sixteen waves = one 8x8 quadrant (four workgroups per tile), batches of B tile entries
    h2 = []
    for r in range(64):
        cx, cy = bx + (r % 8) * 2 + 0.5, by + (r // 8) * 2 + 0.5
        h2.append(((xyh[:, 0] - cx).abs() <= xyh[:, 2] + 0.5) & ((xyh[:, 1] - cy).abs() <= xyh[:, 3] + 0.5))
    h2 = torch.stack(h2, 1)                                               # [I, 64]
    print(f"   2x2 regions hit per entry: mean {h2.float().sum(1).mean():.2f} (x 4 pixels = {4 * h2.float().sum(1).mean():.1f} pairs listed per entry; 4x4 blocks: {16 * hits.float().sum(1).mean():.1f})")
    quad_of = torch.tensor([(r % 8) // 4 + 2 * ((r // 8) // 4) for r in range(64)], device="cuda")
    for B in (256, 512):
        nbt = 32768 // B * 2
        batch = tile_of * nbt + pos // B
        per = torch.zeros(nt * nbt, 64, dtype=torch.long, device="cuda").index_add_(0, batch, h2.long())
        st16 = (per + 15) // 16                                           # steps of a region per batch
        for q in range(4):
            sq = st16[:, quad_of == q]                                    # [batches, 16 regions]
            slow = sq.max(1).values.reshape(nt, nbt).sum(1).float()
            mean = sq.float().mean(1).reshape(nt, nbt).sum(1)
            if q == 0:
                print(f"   16 lanes per pixel, batches of {B}, quadrant 0: steps per quadrant workgroup, slowest region per batch: mean {slow.mean():.0f} max {slow.max():.0f};  mean region: mean {mean.mean():.0f} max {mean.max():.0f}")
        print(f"      total region steps (x 64 lanes) {st16.sum().item()}  against four-lanes steps {((torch.zeros(nt * nbt, 16, dtype=torch.long, device='cuda').index_add_(0, batch, hits.long()) + 3) // 4).sum().item()}")
    # backward block walk (render_bwd_blk.hip): rounds of MB staged entries, a wave = four blocks in lockstep = ceil(longest of its four
    # lists / 16) batches of sixteen pixel steps.  "wide" (round 6 model): all four rows of a wave on ONE block at a time, 64 entries per
    # batch, at 1.25 x the cost of a batch (two more DPP steps per scan) — chosen per wave and round when cheaper.
    for MB in (128, 192):
        nbt = 16384 // MB * 2
        batch = tile_of * nbt + pos // MB
        per = torch.zeros(nt * nbt, 16, dtype=torch.long, device="cuda").index_add_(0, batch, hits).float()
        pw = per[:, wv]                                                    # [rounds, wave, row]
        normal = torch.ceil(pw.max(2).values / 16)
        wide = 1.25 * torch.ceil(pw / 64).sum(2)
        best = torch.minimum(normal, wide)
        ideal = torch.ceil(pw.sum(2) / 64)                                 # the four rows perfectly packed
        crit = lambda x: x.max(1).values.reshape(nt, nbt).sum(1)           # per tile: the slowest wave of every round (a barrier per round)
        print(f"   backward model, rounds of {MB}: batches issued per tile: lockstep {normal.sum().item() / nt:.0f}  with wide mode {best.sum().item() / nt:.0f}  "
              f"perfect packing {ideal.sum().item() / nt:.0f};  critical path (slowest wave per round) mean / max tile: lockstep {crit(normal).mean():.0f} / {crit(normal).max():.0f}  "
              f"wide {crit(best).mean():.0f} / {crit(best).max():.0f};  waves x rounds that would go wide: {(wide < normal).float().mean().item():.2f}")
    # run-ahead model: NB staging areas of B entries, a block starts batch i once it is staged; batch i is staged by the time every block has
    # finished batch i - NB + 1 (loaders write one batch ahead of their own walk); cost = steps (+ a fixed cull per batch)
    for B, NBs in ((512, (2,)), (256, (2, 3, 4, 6))):
        nbt = 16384 // B * 2
        batch = tile_of * nbt + pos // B
        per = torch.zeros(nt * nbt, 16, dtype=torch.long, device="cuda").index_add_(0, batch, hits)
        st = ((per + 3) // 4).reshape(nt, nbt, 16).float().cpu().numpy() + 3.0 * (B / 512)   # (cull: ~130 instructions per 512 entries = 3 steps)
        for NB in NBs:
            fin = np.zeros((nt, nbt + 1, 16))
            for i in range(nbt):
                if NB == 2:
                    start = np.broadcast_to(fin[:, i, :].max(1, keepdims=True), (nt, 16))          # a barrier per batch
                else:
                    ready = fin[:, max(i - NB + 2, 0), :].max(1, keepdims=True) if i - NB + 2 > 0 else 0.0
                    start = np.maximum(fin[:, i, :], ready)
                fin[:, i + 1, :] = start + st[:, i, :]
            mk = fin[:, nbt, :].max(1)
            print(f"   model B={B} NB={NB}: tile makespan mean {mk.mean():.0f} max {mk.max():.0f} steps")
    # backward: a block walks the entries in front of the TILE's last contributor; how many lie in front of its OWN last contributor?
    ncp = torch.from_numpy(nc.astype(np.int64)).cuda().reshape(H, W)
    pad = torch.zeros(ty * 16, tx * 16, dtype=torch.long, device="cuda")
    pad[:H, :W] = ncp
    blk = pad.reshape(ty, 4, 4, tx, 4, 4).permute(0, 3, 1, 4, 2, 5).reshape(nt, 16, 16).max(2).values    # [tile, block]: last contributor (1-based list position)
    tmax = blk.max(1).values
    walked_tile = (hits * (pos < tmax[tile_of]).unsqueeze(1)).sum().item()
    walked_blk = (hits * (pos.unsqueeze(1) < blk[tile_of])).sum().item()
    print(f"   backward (entry, block) pairs in front of the tile's last contributor {walked_tile}, in front of the block's own {walked_blk} ({walked_blk / max(walked_tile, 1):.2f})")
