#!/usr/bin/env python3
"""Diagnostic: structural checks of the binning output (depth order, tile partition, ranges) on a bench workload.
    python tools/sort_check.py --workload c4 --iters 3"""
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
    ap.add_argument("--workload", default="c4")
    ap.add_argument("--iters", type=int, default=3)
    args = ap.parse_args()
    from das3r_amd import GaussianRasterizationSettings, _lib
    from das3r_amd.rasterizer import _forward_impl
    from das3r_amd.synth import make_workload
    dev = torch.device("cuda:0")
    sc = make_workload(args.workload).to(dev)
    kw = sc.settings_kwargs(); kw["debug"] = True
    rs = GaussianRasterizationSettings(**kw)
    e = torch.empty(0, device=dev)
    P = sc.P
    tiles = ((sc.W + 15) // 16) * ((sc.H + 15) // 16)
    os.environ["DAS3R_SORT"] = "classic"
    _lib.reload_switches()
    I0, _, _, geom0, binning0, img0 = _forward_impl(rs, sc.means3D, sc.shs, e, sc.opacities, sc.scales, sc.rotations, e)
    L0 = _lib.layout(P, I0, sc.W, sc.H)
    ref_sidx = _view(geom0, L0["sorted_idx"], torch.int32, P).clone()
    ref_pl = _view(binning0, L0["point_list"], torch.int32, I0).clone()
    ref_rg = _view(img0, L0["ranges"], torch.int32, 2 * tiles).clone()
    del os.environ["DAS3R_SORT"]
    _lib.reload_switches()
    import time
    for it in range(args.iters):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        try:
            I, c0, r0, geom, binning, img = _forward_impl(rs, sc.means3D, sc.shs, e, sc.opacities, sc.scales, sc.rotations, e)
        except RuntimeError as ex:
            torch.cuda.synchronize()
            print(it, "FORWARD ERROR", ex, "after %.3f s" % (time.perf_counter() - t0))
            continue
        L = _lib.layout(P, I, sc.W, sc.H)
        key = _view(geom, L["depth_key"], torch.int32, P).long() & 0xFFFFFFFF
        sidx = _view(geom, L["sorted_idx"], torch.int32, P).long()
        perm_ok = bool(torch.equal(torch.sort(sidx)[0], torch.arange(P, device=dev))) if int(sidx.min()) >= 0 and int(sidx.max()) < P else False
        ks = key[sidx.clamp(0, P - 1)]
        depth_ok = bool(((ks[1:] > ks[:-1]) | ((ks[1:] == ks[:-1]) & (sidx[1:] > sidx[:-1]))).all())
        rg = _view(img, L["ranges"], torch.int32, 2 * tiles).reshape(tiles, 2).long()
        lens = rg[:, 1] - rg[:, 0]
        pl = _view(binning, L["point_list"], torch.int32, I).long()
        pl_ok = int(pl.min()) >= 0 and int(pl.max()) < P
        tt = _view(geom, L["tiles_touched"], torch.int32, P).long()
        cnt_ok = pl_ok and bool(torch.equal(torch.bincount(pl, minlength=P), tt))
        sidx32 = _view(geom, L["sorted_idx"], torch.int32, P)
        pl32 = _view(binning, L["point_list"], torch.int32, I)
        rg32 = _view(img, L["ranges"], torch.int32, 2 * tiles)
        print(it, "vs classic: sorted_idx diff", int((sidx32 != ref_sidx).sum()), "point_list diff",
              int((pl32 != ref_pl).sum()) if I == I0 else "I differs", "ranges diff", int((rg32 != ref_rg).sum()))
        print(it, "I", I, "tt.sum", int(tt.sum()), "perm_ok", perm_ok, "depth_sorted", depth_ok, "ranges.sum", int(lens.sum()),
              "lens.min", int(lens.min()), "point_list in range", pl_ok, "bincount==tiles_touched", cnt_ok)


if __name__ == "__main__":
    main()
