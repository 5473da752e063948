#!/usr/bin/env python3
"""How well the wave-per-quadrant decomposition fits a scene: per (splat, tile) instance, how many 8x8 quadrants survive the
bounding-box cull, and what fraction of the 64 lanes of a surviving quadrant evaluate alpha >= 1/255 (ignoring the
T < 1e-4 early termination).   python tools/pair_stats.py --workload c2"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def _view(buf, off, dtype, count):
    return buf[off:off + count * torch.tensor([], dtype=dtype).element_size()].view(dtype)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c2")
    args = ap.parse_args()
    from das3r_amd import GaussianRasterizationSettings, _lib
    from das3r_amd.rasterizer import _forward_impl
    from das3r_amd.synth import make_workload
    dev = torch.device("cuda:0")
    sc = make_workload(args.workload).to(dev)
    rs = GaussianRasterizationSettings(**sc.settings_kwargs())
    e = torch.empty(0, device=dev)
    P = sc.P
    I, _, _, geom, binning, img = _forward_impl(rs, sc.means3D, sc.shs, e, sc.opacities, sc.scales, sc.rotations, e)
    L = _lib.layout(P, I, sc.W, sc.H)
    tx, ty = (sc.W + 15) // 16, (sc.H + 15) // 16
    tiles = tx * ty
    rg = _view(img, L["ranges"], torch.int32, 2 * tiles).reshape(tiles, 2).long()
    pl = _view(binning, L["point_list"], torch.int32, I).long()
    xyh = _lib.splat_field(geom, L, "xy", P)
    co = _lib.splat_field(geom, L, "conic_opacity", P)
    tile_of = torch.repeat_interleave(torch.arange(tiles, device=dev), rg[:, 1] - rg[:, 0])
    bx, by = (tile_of % tx).float() * 16, (tile_of // tx).float() * 16
    s = xyh[pl]
    c = co[pl]
    hits = torch.zeros(I, 4, dtype=torch.bool, device=dev)
    act = torch.zeros(I, 4, device=dev)
    lx = torch.arange(8, device=dev).float()
    for q in range(4):
        qx, qy = bx + 8 * (q & 1), by + 8 * (q >> 1)
        hit = ((s[:, 0] - (qx + 3.5)).abs() <= s[:, 2] + 3.5) & ((s[:, 1] - (qy + 3.5)).abs() <= s[:, 3] + 3.5)
        hits[:, q] = hit
        n_act = torch.zeros(I, device=dev)
        for chunk in torch.arange(I, device=dev).split(1 << 20):
            dx = s[chunk, 0, None, None] - (qx[chunk, None, None] + lx[None, None, :])
            dy = s[chunk, 1, None, None] - (qy[chunk, None, None] + lx[None, :, None])
            power = -0.5 * (c[chunk, 0, None, None] * dx * dx + c[chunk, 2, None, None] * dy * dy) - c[chunk, 1, None, None] * dx * dy
            alpha = torch.clamp(c[chunk, 3, None, None] * torch.exp(power), max=0.99)
            n_act[chunk] = ((power <= 0) & (alpha >= 1.0 / 255)).float().sum((1, 2))
        act[:, q] = n_act
    nq = hits.sum(1).float()
    surviving = int(hits.sum())
    print(f"{args.workload}: instances {I}  surviving (splat, quadrant) {surviving}  per instance {surviving / I:.2f} of 4")
    print("  instances by #quadrants hit 0..4:", [int((nq == k).sum()) for k in range(5)])
    a = act[hits]
    print(f"  active lanes per surviving quadrant: mean {float(a.mean()):.1f} / 64  (fraction {float(a.mean()) / 64:.2f});"
          f" quadrants with zero active lanes {float((a == 0).float().mean()):.2f}")
    # half-tile (8x16, two pixels per lane) alternative: a wave is busy if either of its two quadrants is hit
    for name, pairs in (("left/right halves (quadrants 0,2 | 1,3)", ((0, 2), (1, 3))), ("top/bottom halves (0,1 | 2,3)", ((0, 1), (2, 3)))):
        busy = sum(int((hits[:, a_] | hits[:, b_]).sum()) for a_, b_ in pairs)
        print(f"  {name}: busy (splat, half) {busy}  -> wave-iterations x{busy / surviving:.2f} of now, pixels per iteration x2")
    whole = int((nq > 0).sum())
    print(f"  whole tile (4 pixels per lane): busy {whole} -> wave-iterations x{whole / surviving:.2f} of now, pixels per iteration x4")


if __name__ == "__main__":
    main()


def row_private_estimate(workload):
    """Iterations per wave if every 16-lane row (4x4 pixel block) walked its own culled list: sum over 256-entry batches of
    the longest of the wave's four row lists, against one iteration per surviving (splat, quadrant) now."""
    from das3r_amd import GaussianRasterizationSettings, _lib
    from das3r_amd.rasterizer import _forward_impl
    from das3r_amd.synth import make_workload
    dev = torch.device("cuda:0")
    sc = make_workload(workload).to(dev)
    rs = GaussianRasterizationSettings(**sc.settings_kwargs())
    e = torch.empty(0, device=dev)
    P = sc.P
    I, _, _, geom, binning, img = _forward_impl(rs, sc.means3D, sc.shs, e, sc.opacities, sc.scales, sc.rotations, e)
    L = _lib.layout(P, I, sc.W, sc.H)
    tx, ty = (sc.W + 15) // 16, (sc.H + 15) // 16
    tiles = tx * ty
    rg = _view(img, L["ranges"], torch.int32, 2 * tiles).reshape(tiles, 2).long()
    pl = _view(binning, L["point_list"], torch.int32, I).long()
    xyh = _lib.splat_field(geom, L, "xy", P)
    lens = rg[:, 1] - rg[:, 0]
    tile_of = torch.repeat_interleave(torch.arange(tiles, device=dev), lens)
    pos = torch.arange(I, device=dev) - rg[tile_of, 0]
    batch = pos // 256
    nb = int(batch.max()) + 1
    bx, by = (tile_of % tx).float() * 16, (tile_of // tx).float() * 16
    s = xyh[pl]
    old_iters = new_iters = 0
    blk_hits_total = 0
    for q in range(4):
        qx, qy = bx + 8 * (q & 1), by + 8 * (q >> 1)
        qhit = ((s[:, 0] - (qx + 3.5)).abs() <= s[:, 2] + 3.5) & ((s[:, 1] - (qy + 3.5)).abs() <= s[:, 3] + 3.5)
        old_iters += int(qhit.sum())
        per_row = []
        for r in range(4):
            cx, cy = qx + 4 * (r & 1) + 1.5, qy + 4 * (r >> 1) + 1.5
            hit = ((s[:, 0] - cx).abs() <= s[:, 2] + 1.5) & ((s[:, 1] - cy).abs() <= s[:, 3] + 1.5)
            blk_hits_total += int(hit.sum())
            cnt = torch.zeros(tiles * nb, device=dev)
            cnt.index_add_(0, tile_of * nb + batch, hit.float())
            per_row.append(cnt)
        new_iters += int(torch.stack(per_row).max(0)[0].sum())
    print(f"{workload}: iterations now {old_iters}; row-private {new_iters} (x{new_iters / old_iters:.2f}); surviving (splat, 4x4 block) "
          f"{blk_hits_total} = {blk_hits_total / old_iters:.2f} per surviving quadrant; row occupancy {blk_hits_total / (4 * new_iters):.2f}")


def half_wave_estimate(workload):
    """Iterations per wave if each 32-lane half (8x4 pixels) of a quadrant's wave walked its own culled list."""
    from das3r_amd import GaussianRasterizationSettings, _lib
    from das3r_amd.rasterizer import _forward_impl
    from das3r_amd.synth import make_workload
    dev = torch.device("cuda:0")
    sc = make_workload(workload).to(dev)
    rs = GaussianRasterizationSettings(**sc.settings_kwargs())
    e = torch.empty(0, device=dev)
    P = sc.P
    I, _, _, geom, binning, img = _forward_impl(rs, sc.means3D, sc.shs, e, sc.opacities, sc.scales, sc.rotations, e)
    L = _lib.layout(P, I, sc.W, sc.H)
    tx, ty = (sc.W + 15) // 16, (sc.H + 15) // 16
    tiles = tx * ty
    rg = _view(img, L["ranges"], torch.int32, 2 * tiles).reshape(tiles, 2).long()
    pl = _view(binning, L["point_list"], torch.int32, I).long()
    xyh = _lib.splat_field(geom, L, "xy", P)
    lens = rg[:, 1] - rg[:, 0]
    tile_of = torch.repeat_interleave(torch.arange(tiles, device=dev), lens)
    pos = torch.arange(I, device=dev) - rg[tile_of, 0]
    batch = pos // 256
    nb = int(batch.max()) + 1
    bx, by = (tile_of % tx).float() * 16, (tile_of // tx).float() * 16
    s = xyh[pl]
    old_iters = new_iters = half_hits = 0
    for q in range(4):
        qx, qy = bx + 8 * (q & 1), by + 8 * (q >> 1)
        qhit = ((s[:, 0] - (qx + 3.5)).abs() <= s[:, 2] + 3.5) & ((s[:, 1] - (qy + 3.5)).abs() <= s[:, 3] + 3.5)
        old_iters += int(qhit.sum())
        per_half = []
        for h in range(2):
            cx, cy = qx + 3.5, qy + 4 * h + 1.5
            hit = ((s[:, 0] - cx).abs() <= s[:, 2] + 3.5) & ((s[:, 1] - cy).abs() <= s[:, 3] + 1.5)
            half_hits += int(hit.sum())
            cnt = torch.zeros(tiles * nb, device=dev)
            cnt.index_add_(0, tile_of * nb + batch, hit.float())
            per_half.append(cnt)
        new_iters += int(torch.stack(per_half).max(0)[0].sum())
    print(f"{workload}: iterations now {old_iters}; half-wave (8x4) {new_iters} (x{new_iters / old_iters:.2f}); surviving (splat, half) "
          f"{half_hits} = {half_hits / old_iters:.2f} per surviving quadrant; half occupancy {half_hits / (2 * new_iters):.2f}")


if __name__ == "__main__" and os.environ.get("HALFWAVE"):
    for w in os.environ["HALFWAVE"].split(","):
        half_wave_estimate(w)

if __name__ == "__main__" and os.environ.get("ROWPRIV"):
    for w in os.environ["ROWPRIV"].split(","):
        row_private_estimate(w)
