#!/usr/bin/env python3
"""Repeat forward + backward of the small parity scenes many times and compare every run with the first: hunts run-to-run
differences (a race, an uninitialised read).  python tools/flaky_probe.py [--reps 100]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from tests import util
from das3r_amd import GaussianRasterizationSettings, GaussianRasterizer

ap = argparse.ArgumentParser(); ap.add_argument("--reps", type=int, default=100); ap.add_argument("--scenes", default=",".join(util.VARIANTS))
ap.add_argument("--alternate", action="store_true", help="interleave the scenes (deg1 / deg2 share P, W, H: the second one takes the speculative path with the first one's counts)")
args = ap.parse_args()
dev = torch.device("cuda:0")
bad = 0
if args.alternate:
    names = args.scenes.split(",")
    data = {n: (util.scene_variant(n), util.run_oracle(*util.scene_variant(n))[0]) for n in names}
    for rep in range(args.reps):
        for n in names:
            (sc, mode), ref_color = data[n]
            kw = {k: v.to(dev).clone().requires_grad_(True) for k, v in util.raster_inputs(sc, mode).items()}
            skw = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in util.settings_kwargs(sc, mode).items()}
            m2 = torch.zeros(sc.P, 3, device=dev, requires_grad=True)
            junk = torch.full((1 << 22,), float("nan"), device=dev).sin_()
            color, radii = GaussianRasterizer(GaussianRasterizationSettings(**skw))(means2D=m2, **kw)
            color.backward(sc.dL_dpix.to(dev))
            torch.cuda.synchronize()
            c = color.detach().cpu().numpy()
            err = np.abs(c - ref_color)
            ok = np.isfinite(c).all() and (err > 1e-4).mean() <= 1e-3 and err.max() <= 2.5e-2 and all(torch.isfinite(v.grad).all() for v in kw.values())
            if not ok:
                bad += 1
                print(f"{n} rep {rep}: max err {np.nanmax(err):.3e} nan={np.isnan(c).sum()} I={color.grad_fn.num_rendered}", flush=True)
    print("BAD RUNS:", bad)
    sys.exit(0)
for name in args.scenes.split(","):
    sc, mode = util.scene_variant(name)
    ref_color, ref_radii, ref_g, S = util.run_oracle(sc, mode)
    first = None
    for rep in range(args.reps):
        kw = {k: v.to(dev).clone().requires_grad_(True) for k, v in util.raster_inputs(sc, mode).items()}
        skw = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in util.settings_kwargs(sc, mode).items()}
        m2 = torch.zeros(sc.P, 3, device=dev, requires_grad=True)
        # dirty the LDS / caches between runs with an unrelated kernel
        junk = torch.randn(1 << 20, device=dev); junk = (junk * 3).sin_()
        color, radii = GaussianRasterizer(GaussianRasterizationSettings(**skw))(means2D=m2, **kw)
        color.backward(sc.dL_dpix.to(dev))
        torch.cuda.synchronize()
        c = color.detach().cpu().numpy()
        err = np.abs(c - ref_color)
        ok = np.isfinite(c).all() and (err > 1e-4).mean() <= 1e-3 and err.max() <= 2.5e-2
        if first is None:
            first = c
        same = np.array_equal(c, first)
        if not ok or not same:
            bad += 1
            ys, xs = np.nonzero(np.abs(c - first).max(0) > 0) if not same else ([], [])
            tiles = sorted({(int(y) // 16, int(x) // 16) for y, x in zip(ys, xs)})
            print(f"{name} rep {rep}: ok={ok} same_as_first={same} max err {np.nanmax(err):.3e} nan={np.isnan(c).sum()} differing tiles {tiles[:12]} ({len(tiles)}) I={color.grad_fn.num_rendered}", flush=True)
    print(f"{name}: done ({args.reps} reps)", flush=True)
print("BAD RUNS:", bad)
