#!/usr/bin/env python3
"""Segmented binning path at full size against the global sort (round 4): the same forward with DAS3R_BINNING=radix and =seg,
lists / ranges / image / radii compared bit for bit.      python tools/seg_check.py [--workloads ds,c4]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workloads", default="ds,c4")
    a = ap.parse_args()
    from das3r_amd import GaussianRasterizationSettings, _lib
    from das3r_amd.rasterizer import _forward_impl
    from das3r_amd.synth import make_workload
    dev = torch.device("cuda:0")
    e = torch.empty(0, device=dev)
    for w in a.workloads.split(","):
        sc = make_workload(w).to(dev)
        rs = GaussianRasterizationSettings(**sc.settings_kwargs())
        out = {}
        for kind in ("radix", "seg", "seg"):
            os.environ["DAS3R_BINNING"] = kind
            _lib.reload_switches()
            I, color, radii, geom, binning, img = _forward_impl(rs, sc.means3D, sc.shs, e, sc.opacities, sc.scales, sc.rotations, e)
            torch.cuda.synchronize()
            L = _lib.layout(sc.P, I, sc.W, sc.H)
            pl = binning[L["point_list"]:L["point_list"] + 4 * I].view(torch.int32).clone()
            tiles = ((sc.W + 15) // 16) * ((sc.H + 15) // 16)
            rg = img[L["ranges"]:L["ranges"] + 8 * tiles].view(torch.int32).clone()
            npix = sc.W * sc.H
            ft = img[L["final_T"]:L["final_T"] + 4 * npix].view(torch.float32).clone()
            nc = img[L["n_contrib"]:L["n_contrib"] + 4 * npix].view(torch.int32).clone()
            if kind in out:
                kind = kind + "2"
            out[kind] = (I, pl, rg, color.clone(), radii.clone(), ft, nc)
        I0, pl0, rg0, c0, r0, ft0, nc0 = out["radix"]
        for k in ("seg", "seg2"):
            I1, pl1, rg1, c1, r1, ft1, nc1 = out[k]
            bad = int((pl0 != pl1).sum()) if I0 == I1 else -1
            print(f"{w} {k}: I {I0} {I1}  list mismatches {bad}  ranges equal {bool(torch.equal(rg0, rg1))}  image equal {bool(torch.equal(c0, c1))}  radii equal {bool(torch.equal(r0, r1))}"
                  f"  final_T equal {bool(torch.equal(ft0, ft1))}  n_contrib equal {bool(torch.equal(nc0, nc1))}")
            if bad > 0:
                idx = torch.nonzero(pl0 != pl1).reshape(-1)
                print("   first mismatches at", idx[:8].tolist(), "last", idx[-3:].tolist())
        print(_lib.load().das3r_last_error())


if __name__ == "__main__":
    main()
