#!/usr/bin/env python3
"""Tile-list length distribution of a workload (mean, percentiles, lists over 256 / 1024 entries: the switch-overs of the
compositing kernel's own sort).   python tools/list_lengths.py c2,c4"""
import sys, torch, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from das3r_amd import GaussianRasterizationSettings, _lib
from das3r_amd.rasterizer import _forward_full
from das3r_amd.synth import make_workload
dev = torch.device("cuda:0")
for w in sys.argv[1].split(","):
    sc = make_workload(w).to(dev)
    rs = GaussianRasterizationSettings(**sc.settings_kwargs())
    e = torch.empty(0, device=dev)
    I, color, radii, geom, binning, img, cap = _forward_full(rs, sc.means3D, sc.shs, e, sc.opacities, sc.scales, sc.rotations, e)
    torch.cuda.synchronize()
    L = _lib.layout(sc.P, I, sc.W, sc.H)
    nt = ((sc.W + 15) // 16) * ((sc.H + 15) // 16)
    rg = img[L["ranges"]:L["ranges"] + 8 * nt].view(torch.int32).cpu().numpy().reshape(nt, 2)
    ln = (rg[:, 1] - rg[:, 0]).astype(np.int64)
    nc = img[L["n_contrib"]:L["n_contrib"] + 4 * sc.W * sc.H].view(torch.int32).cpu().numpy()
    print(w, "I", I, "tiles", nt, "mean", ln.mean(), "pcts 50/90/99/99.9/max", np.percentile(ln, [50, 90, 99, 99.9, 100]), "n>256:", (ln > 256).sum(), "n>1024:", (ln > 1024).sum())
    print("  n_contrib mean/max", nc.mean(), nc.max())
