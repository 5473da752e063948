#!/usr/bin/env python3
"""Timeline of the workgroups of the tile-partition passes (make EXPERIMENTS=1 builds only): when each took its ticket, had its
keys, had counted, had published and staged, finished its look-back, issued and completed its stores (100 MHz stamps).

    python tools/wg_trace.py --workload c4"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

NAMES = ["ticket", "keys here", "ranked", "published+staged", "look-back done", "stores issued", "stores done"]
PASSES, WGS, STAMPS = 8, 16384, 8


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c4")
    args = ap.parse_args()
    from das3r_amd import GaussianRasterizationSettings, _lib
    from das3r_amd.rasterizer import _forward_full
    from das3r_amd.synth import make_workload
    if not _lib.has_experiments():
        sys.exit("the trace needs the experiments build: make -C das3r_amd/csrc EXPERIMENTS=1")
    L = _lib.load()
    L.das3r_debug_wg_trace.restype = C.c_int
    L.das3r_debug_wg_trace.argtypes = [C.c_int, C.c_void_p]
    dev = torch.device("cuda:0")
    sc = make_workload(args.workload).to(dev)
    rs = GaussianRasterizationSettings(**sc.settings_kwargs())
    e = torch.empty(0, device=dev)

    def fwd():
        return _forward_full(rs, sc.means3D, sc.shs, e, sc.opacities, sc.scales, sc.rotations, e)
    for _ in range(5):
        fwd()
    torch.cuda.synchronize()
    from das3r_amd.rasterizer import _backward_impl
    _lib.check(L.das3r_debug_wg_trace(1, None), "trace on")
    I, color, radii, geom, binning, img, cap = fwd()
    _backward_impl(rs, I, sc.dL_dpix, sc.means3D, sc.shs, e, sc.opacities, sc.scales, sc.rotations, e, geom, binning, img, cap)
    torch.cuda.synchronize()
    buf = np.zeros(PASSES * WGS * STAMPS, dtype=np.uint64)
    _lib.check(L.das3r_debug_wg_trace(0, buf.ctypes.data_as(C.c_void_p)), "trace off")
    t = buf.reshape(PASSES, WGS, STAMPS).astype(np.int64)
    # scan + emission: region PASSES - 1, five stamps
    sw = t[PASSES - 1][t[PASSES - 1, :, 0] != 0]
    if len(sw):
        t0 = sw[:, 0].min()
        q = lambda a: " ".join(f"{v:7.1f}" for v in np.percentile(a, [0, 10, 50, 90, 100]))  # noqa: E731
        print(f"== scan + emission: {len(sw)} workgroups, first ticket -> last acknowledged {(sw[:, 4].max() - t0) / 100.0:.1f} us")
        print(f"   ticket time (us)                   min/10/50/90/max: {q((sw[:, 0] - t0) / 100.0)}")
        for k, name in ((1, "rectangles read + block total"), (2, "look-back"), (3, "offsets + instances written"), (4, "acknowledged")):
            print(f"   {name:34s} min/10/50/90/max: {q((sw[:, k] - sw[:, k - 1]) / 100.0)}")
        print(f"   lifetime                           min/10/50/90/max: {q((sw[:, 4] - sw[:, 0]) / 100.0)}")
    # compositing kernels: regions 4 (rows forward) and 5 (block-walk backward): start, stores issued, acknowledged, list length
    for region, name in ((4, "render forward (rows)"), (5, "render backward (blk)")):
        cw = t[region][t[region, :, 0] != 0]
        if not len(cw):
            continue
        t0 = cw[:, 0].min()
        q = lambda a: " ".join(f"{v:7.1f}" for v in np.percentile(a, [0, 10, 50, 90, 100]))  # noqa: E731
        start, end = (cw[:, 0] - t0) / 100.0, (cw[:, 2] - t0) / 100.0
        life = end - start
        span = end.max()
        # workgroups in flight over time
        ev = np.concatenate([np.stack([start, np.ones_like(start)], 1), np.stack([end, -np.ones_like(end)], 1)])
        ev = ev[np.argsort(ev[:, 0])]
        conc = np.cumsum(ev[:, 1])
        peak = conc.max()
        dt = np.diff(ev[:, 0], append=span)
        mean_conc = float((conc * dt).sum() / span)
        print(f"== {name}: {len(cw)} workgroups, first start -> last acknowledged {span:.1f} us; in flight: peak {int(peak)}, time-average {mean_conc:.0f}")
        print(f"   start time (us)                    min/10/50/90/max: {q(start)}")
        print(f"   lifetime (us)                      min/10/50/90/max: {q(life)}")
        print(f"   list length                        min/10/50/90/max: {q(cw[:, 3].astype(float))}")
        print(f"   lifetime vs list length: corr {np.corrcoef(life, cw[:, 3])[0, 1]:.2f};  us per 100 entries (median) {np.median(life / np.maximum(cw[:, 3], 1)) * 100:.1f}")
        print(f"   last start at {start.max():.1f} us; time with fewer than half the peak in flight: {float(dt[conc < peak / 2].sum()):.1f} us")
    # per-Gaussian backward: region PASSES - 2, eight stamps, by blockIdx
    pb = t[PASSES - 2][t[PASSES - 2, :, 0] != 0]
    if len(pb):
        t0 = pb[:, 0].min()
        q = lambda a: " ".join(f"{v:7.1f}" for v in np.percentile(a, [0, 10, 50, 90, 100]))  # noqa: E731
        print(f"== preprocess backward: {len(pb)} workgroups traced (of P / 256), first start -> last acknowledged {(pb[:, 7].max() - t0) / 100.0:.1f} us")
        print(f"   start time (us)                    min/10/50/90/max: {q((pb[:, 0] - t0) / 100.0)}")
        names = ["own words + run known", "first chunk of rows in LDS", "rows added", "SH rows in LDS", "arithmetic + small rows", "dL_dsh issued", "acknowledged"]
        for k, name in enumerate(names, start=1):
            d = (pb[:, k] - pb[:, k - 1]) / 100.0
            d = d[(pb[:, k] != 0) & (pb[:, k - 1] != 0)]
            if len(d):
                print(f"   {name:34s} min/10/50/90/max: {q(d)}")
        print(f"   lifetime                           min/10/50/90/max: {q((pb[:, 7] - pb[:, 0]) / 100.0)}")
    for p in range(PASSES - 2):
        used = t[p, :, 0] != 0
        n = int(used.sum())
        if n == 0:
            continue
        w = t[p, used]
        worked = w[:, 1] != 0            # (a workgroup past the end of the list returns before its first stamp-1)
        t0 = w[:, 0].min()
        us = lambda x: (x - t0) / 100.0   # noqa: E731
        span = us(w[worked][:, 6].max())
        print(f"== pass {p} (shift {8 * p}): {n} workgroups stamped, {int(worked.sum())} with work, first ticket -> last store done {span:.1f} us")
        ww = w[worked]
        order = np.argsort(ww[:, 0])
        q = lambda a: " ".join(f"{v:7.1f}" for v in np.percentile(a, [0, 10, 50, 90, 100]))  # noqa: E731
        print(f"   ticket time (us)                 min/10/50/90/max: {q(us(ww[:, 0]))}")
        for k in range(1, 7):
            print(f"   {NAMES[k]:20s} - {NAMES[k - 1]:18s} min/10/50/90/max: {q((ww[:, k] - ww[:, k - 1]) / 100.0)}")
        print(f"   lifetime (ticket -> stores done)  min/10/50/90/max: {q((ww[:, 6] - ww[:, 0]) / 100.0)}")
        print(f"   end time (us)                     min/10/50/90/max: {q(us(ww[:, 6]))}")
        # does ticket order follow block order?
        tick = np.nonzero(used)[0][worked]
        print(f"   corr(ticket, blockIdx) = {np.corrcoef(tick, ww[:, 7])[0, 1]:.3f}")


if __name__ == "__main__":
    main()
