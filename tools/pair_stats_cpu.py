#!/usr/bin/env python3
"""CPU model of the compositing kernels' pair counts (no GPU needed): for a sample of tiles of a synthetic workload, how many
(pixel, splat) pairs each decomposition evaluates — wave per 8x8 quadrant (render_bwd_scan.hip today), 4x4 blocks with
row-private lists, 4x4 blocks walked one after the other — against the pairs that are live (alpha >= 1/255).
    python tools/pair_stats_cpu.py --workload c4 --every 4
Uses the CPU oracle only for the per-splat projection (xy, conic, opacity); test infrastructure, like tools/pair_stats.py."""
import argparse
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def project(sc):
    from oracle.c_oracle import RasterOracle
    kw = {k: (v.numpy() if hasattr(v, "numpy") else v) for k, v in sc.settings_kwargs().items()}
    o = RasterOracle(**kw)
    # a 16x16 image is enough for the per-splat part?  No: xy depends on W, H.  Run the real size; the render is the slow part.
    o.forward(sc.means3D.numpy(), sc.opacities.numpy(), shs=sc.shs.numpy(), scales=sc.scales.numpy(), rotations=sc.rotations.numpy())
    s = o.saved()
    o.free()
    return s


def min_q_rect(A, B, C, dxl, dxh, dyl, dyh):
    """min of A dx^2 + 2 B dx dy + C dy^2 over dx in [dxl, dxh], dy in [dyl, dyh] (render_scan.h rect_hit_tight)."""
    def qf(dx, dy):
        return A * dx * dx + 2 * B * dx * dy + C * dy * dy
    q0 = qf(dxl, np.clip(-B * dxl / C, dyl, dyh))
    q1 = qf(dxh, np.clip(-B * dxh / C, dyl, dyh))
    q2 = qf(np.clip(-B * dyl / A, dxl, dxh), dyl)
    q3 = qf(np.clip(-B * dyh / A, dxl, dxh), dyh)
    inside = (dxl <= 0) & (dxh >= 0) & (dyl <= 0) & (dyh >= 0)
    return np.where(inside, 0.0, np.minimum(np.minimum(q0, q1), np.minimum(q2, q3)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c4")
    ap.add_argument("--P", type=int, default=None)
    ap.add_argument("--every", type=int, default=4, help="sample tiles with tx % every == 0 and ty % every == 0")
    ap.add_argument("--mb", type=int, default=128)
    args = ap.parse_args()
    from das3r_amd.synth import make_workload
    sc = make_workload(args.workload, args.P)
    s = project(sc)
    W, H = sc.W, sc.H
    tx_n, ty_n = (W + 15) // 16, (H + 15) // 16
    xy, co, depth, radii_ok = s["xy"].astype(np.float64), s["conic_opacity"].astype(np.float64), s["depths"], s["tiles_touched"] > 0
    op = co[:, 3]
    vis = radii_ok & (255.0 * op > 1.0)
    idx = np.nonzero(vis)[0]
    x, y = xy[idx, 0], xy[idx, 1]
    A, B, C, o = co[idx, 0], co[idx, 1], co[idx, 2], op[idx]
    det = A * C - B * B
    sxx, syy = C / det, A / det   # Sigma' entries
    tau = 2 * np.log(255 * o)
    hx, hy = np.sqrt(tau * sxx) * 1.0005 + 0.02, np.sqrt(tau * syy) * 1.0005 + 0.02
    px0, px1 = np.ceil(x - hx), np.floor(x + hx)
    py0, py1 = np.ceil(y - hy), np.floor(y + hy)
    ok = (px1 >= px0) & (py1 >= py0) & (px1 >= 0) & (py1 >= 0) & (px0 < W) & (py0 < H)
    t0x, t1x = np.clip(px0, 0, None).astype(int) >> 4, np.minimum(np.clip(px1, 0, 1e6).astype(int) >> 4, tx_n - 1)
    t0y, t1y = np.clip(py0, 0, None).astype(int) >> 4, np.minimum(np.clip(py1, 0, 1e6).astype(int) >> 4, ty_n - 1)
    I_total = int((np.maximum(t1x - t0x + 1, 0) * np.maximum(t1y - t0y + 1, 0))[ok].sum())
    print(f"{args.workload}: visible {len(idx)}, instances (tight rect model) {I_total}, mean list {I_total / (tx_n * ty_n):.0f}")
    ev = args.every
    tot = dict(inst=0, quad_bbox=0, quad_exact=0, blk_bbox=0, blk_exact=0, live=0, rowpriv_batches=0, seq_batches=0, quad_batches=0,
               rowpriv_batches_bbox=0, half_exact=0, rowpriv_batches_mixed=0, rowpriv64=0, rowpriv256=0, quad256=0, blk_oct=0, rowpriv_oct=0)
    MB = args.mb
    for ty in range(0, ty_n, ev):
        for tx in range(0, tx_n, ev):
            m = ok & (t0x <= tx) & (t1x >= tx) & (t0y <= ty) & (t1y >= ty)
            k = np.nonzero(m)[0]
            if len(k) == 0:
                continue
            k = k[np.argsort(depth[idx[k]], kind="stable")]
            n = len(k)
            tot["inst"] += n
            X, Y, a, b, c, oo, hxk, hyk, tk = x[k], y[k], A[k], B[k], C[k], o[k], hx[k], hy[k], tau[k]
            # live pixels per instance per 4x4 block
            pxs = tx * 16 + np.arange(16)
            pys = ty * 16 + np.arange(16)
            dx = X[:, None, None] - pxs[None, None, :]
            dy = Y[:, None, None] - pys[None, :, None]
            power = -0.5 * (a[:, None, None] * dx * dx + c[:, None, None] * dy * dy) - b[:, None, None] * dx * dy
            alpha = np.minimum(0.99, oo[:, None, None] * np.exp(power))
            live = (power <= 0) & (alpha >= 1.0 / 255)
            live &= (pxs[None, None, :] < W) & (pys[None, :, None] < H)
            tot["live"] += int(live.sum())
            blk_b = np.zeros((n, 4, 4), bool)
            blk_e = np.zeros((n, 4, 4), bool)
            blk_o = np.zeros((n, 4, 4), bool)
            detk = a * c - b * b
            hd1 = np.sqrt(tk * (a + c - 2 * b) / (2 * detk)) * 1.0005 + 0.02   # extent along (1, 1) / sqrt 2
            hd2 = np.sqrt(tk * (a + c + 2 * b) / (2 * detk)) * 1.0005 + 0.02   # extent along (1, -1) / sqrt 2
            for byi in range(4):
                for bxi in range(4):
                    x0, y0 = tx * 16 + 4 * bxi, ty * 16 + 4 * byi
                    bb = (np.abs(X - (x0 + 1.5)) <= hxk + 1.5) & (np.abs(Y - (y0 + 1.5)) <= hyk + 1.5)
                    qmin = min_q_rect(a, b, c, X - (x0 + 3), X - x0, Y - (y0 + 3), Y - y0)
                    blk_b[:, byi, bxi] = bb
                    blk_e[:, byi, bxi] = bb & (qmin * 0.9999 - 1e-3 <= tk)
                    ddx, ddy = X - (x0 + 1.5), Y - (y0 + 1.5)
                    r2 = 3.0 / math.sqrt(2.0)   # half extent of the block's pixel-centre square along a diagonal: (1.5 + 1.5) / sqrt 2
                    blk_o[:, byi, bxi] = bb & (np.abs(ddx + ddy) / math.sqrt(2.0) <= hd1 + r2) & (np.abs(ddx - ddy) / math.sqrt(2.0) <= hd2 + r2)
            quad_b = np.zeros((n, 2, 2), bool)
            quad_e = np.zeros((n, 2, 2), bool)
            for qy in range(2):
                for qx in range(2):
                    x0, y0 = tx * 16 + 8 * qx, ty * 16 + 8 * qy
                    bb = (np.abs(X - (x0 + 3.5)) <= hxk + 3.5) & (np.abs(Y - (y0 + 3.5)) <= hyk + 3.5)
                    qmin = min_q_rect(a, b, c, X - (x0 + 7), X - x0, Y - (y0 + 7), Y - y0)
                    quad_b[:, qy, qx] = bb
                    quad_e[:, qy, qx] = bb & (qmin * 0.9999 - 1e-3 <= tk)
            tot["quad_bbox"] += int(quad_b.sum())
            tot["quad_exact"] += int(quad_e.sum())
            tot["blk_bbox"] += int(blk_b.sum())
            tot["blk_exact"] += int(blk_e.sum())
            tot["blk_oct"] += int((blk_o & quad_e.repeat(2, 1).repeat(2, 2)).sum())
            assert not (blk_e & ~blk_o).any(), "octagon test must be conservative"
            # 8x4 halves of a quadrant (two blocks side by side)
            half_e = blk_e.reshape(n, 4, 2, 2).any(3)
            tot["half_exact"] += int(half_e.sum())
            # rounds of MB entries; per wave (quadrant): batches of 16
            for r0 in range(0, n, MB):
                sl = slice(r0, min(n, r0 + MB))
                for qy in range(2):
                    for qx in range(2):
                        tot["quad_batches"] += math.ceil(int(quad_e[sl, qy, qx].sum()) / 16)
                        lens = [int(blk_e[sl, 2 * qy + j, 2 * qx + i].sum()) for j in range(2) for i in range(2)]
                        lens_b = [int(blk_b[sl, 2 * qy + j, 2 * qx + i].sum()) for j in range(2) for i in range(2)]
                        tot["rowpriv_batches"] += max(math.ceil(v / 16) for v in lens)
                        tot["rowpriv_batches_bbox"] += max(math.ceil(v / 16) for v in lens_b)
                        tot["seq_batches"] += sum(math.ceil(v / 16) for v in lens)
                        lens_m = [int((blk_b[sl, 2 * qy + j, 2 * qx + i] & quad_e[sl, qy, qx]).sum()) for j in range(2) for i in range(2)]
                        tot["rowpriv_batches_mixed"] += max(math.ceil(v / 16) for v in lens_m)
                        lens_o = [int((blk_o[sl, 2 * qy + j, 2 * qx + i] & quad_e[sl, qy, qx]).sum()) for j in range(2) for i in range(2)]
                        tot["rowpriv_oct"] += max(math.ceil(v / 16) for v in lens_o)
            for mb2, key in ((64, "rowpriv64"), (256, "rowpriv256")):
                for r0 in range(0, n, mb2):
                    sl = slice(r0, min(n, r0 + mb2))
                    for qy in range(2):
                        for qx in range(2):
                            lens = [int(blk_e[sl, 2 * qy + j, 2 * qx + i].sum()) for j in range(2) for i in range(2)]
                            tot[key] += max(math.ceil(v / 16) for v in lens)
                            if mb2 == 256:
                                tot["quad256"] += math.ceil(int(quad_e[sl, qy, qx].sum()) / 16)
    t = tot
    print(f"sampled instances {t['inst']}; live pairs {t['live']} = {t['live'] / t['inst']:.1f} per instance")
    print(f"quadrant hits per instance: bbox {t['quad_bbox'] / t['inst']:.2f}, exact {t['quad_exact'] / t['inst']:.2f}"
          f"  -> pairs {64 * t['quad_exact'] / t['inst']:.0f} per instance, live fraction {t['live'] / (64 * t['quad_exact']):.2f}")
    print(f"  with 16-splat batch padding: {1024 * t['quad_batches'] / t['inst']:.0f} pairs per instance, live {t['live'] / (1024 * t['quad_batches']):.2f}")
    print(f"4x4 block hits per instance: bbox {t['blk_bbox'] / t['inst']:.2f}, exact {t['blk_exact'] / t['inst']:.2f}"
          f"  -> pairs {16 * t['blk_exact'] / t['inst']:.0f} per instance, live fraction {t['live'] / (16 * t['blk_exact']):.2f}")
    print(f"8x4 halves exact: {t['half_exact'] / t['inst']:.2f} per instance -> pairs {32 * t['half_exact'] / t['inst']:.0f}")
    print(f"row-private (wave = 4 blocks in lockstep, exact): {1024 * t['rowpriv_batches'] / t['inst']:.0f} pairs per instance"
          f" (x{t['rowpriv_batches'] / t['quad_batches']:.2f} of today's batches), live {t['live'] / (1024 * t['rowpriv_batches']):.2f};"
          f" bbox only: x{t['rowpriv_batches_bbox'] / t['quad_batches']:.2f}")
    print(f"  quadrant-exact + block-bbox lists: x{t['rowpriv_batches_mixed'] / t['quad_batches']:.2f};  exact lists with 64-entry rounds: "
          f"x{t['rowpriv64'] / t['quad_batches']:.2f}, 256-entry rounds: x{t['rowpriv256'] / t['quad_batches']:.2f} (quadrant lists at 256: x{t['quad256'] / t['quad_batches']:.2f})")
    print(f"  quadrant-exact + block octagon (bbox + two diagonal extents): {t['blk_oct'] / t['inst']:.2f} block hits per instance, x{t['rowpriv_oct'] / t['quad_batches']:.2f}")
    print(f"blocks one after the other (4-step batches): {256 * t['seq_batches'] / t['inst']:.0f} pairs per instance"
          f" (x{256 * t['seq_batches'] / (1024 * t['quad_batches']):.2f}), live {t['live'] / (256 * t['seq_batches']):.2f}")


if __name__ == "__main__":
    main()
