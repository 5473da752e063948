#!/usr/bin/env python3
"""(tile, depth bucket) segments of the segmented binning path with TWO partition passes (7 bucket bits at 512 x 208): how long they are on the
shapes DAS3R trains at, and what splitting a long segment by the top bits of its fractions inside LDS (segsort.hip, round 6) would leave for the
rank loop.   python tools/probes/seg_lengths.py dsc,ds,smooth,noise,consistent,davis"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["DAS3R_BINNING"] = "seg"
from types import SimpleNamespace
from das3r_amd import GaussianRasterizationSettings, _lib
from das3r_amd.rasterizer import _forward_full


def inputs(kind):
    dev = torch.device("cuda:0")
    e = torch.empty(0, device=dev)
    if kind in ("dsc", "ds"):
        from das3r_amd.synth import make_workload
        sc = make_workload(kind).to(dev)
        rs = GaussianRasterizationSettings(**sc.settings_kwargs())
        return rs, (sc.means3D, sc.shs, e, sc.opacities, sc.scales, sc.rotations, e), sc.W, sc.H
    from das3r_amd.render import rasterizer_inputs
    from das3r_amd.train import build_from_sequence, consistent_sequence, synthetic_sequence
    from das3r_amd.model import OptimParams
    W, H, frames = (512, 288, 45) if kind == "davis" else (512, 208, 22 if kind == "consistent" else 20)
    if kind in ("consistent", "davis"):
        seq = consistent_sequence(frames=frames, W=W, H=H, focal=600.0, n_splats=20000, seed=0)
        model, cams, _test = build_from_sequence(seq, heldout=True)
    else:
        seq = synthetic_sequence(frames=frames, W=W, H=H, focal=600.0, n_splats=20000, seed=0, depth=kind)
        model, cams = build_from_sequence(seq)
    model.training_setup(OptimParams(iterations=4000), fused=True)
    pipe = SimpleNamespace(debug=False, compute_cov3D_python=False, convert_SHs_python=False)
    cam = cams[3]
    with torch.no_grad():
        rs, kw = rasterizer_inputs(cam, model, pipe, torch.zeros(3, device=dev), camera_pose=model.get_RT(cam.uid), fused=True)
    return rs, (kw["means3D"], kw["shs"], e, kw["opacities"], kw["scales"], kw["rotations"], e), W, H


def pct(x, w=None):
    return np.percentile(x, [50, 90, 99, 100]).astype(np.int64).tolist()


for kind in (sys.argv[1] if len(sys.argv) > 1 else "dsc,ds,smooth,noise,consistent,davis").split(","):
    rs, args, W, H = inputs(kind)
    _lib.reload_switches()
    _lib.forget_shapes()
    with torch.no_grad():
        I, color, radii, geom, binning, img, cap = _forward_full(rs, *args, exact=True)
    torch.cuda.synchronize()
    P = args[0].shape[0]
    keys = binning[0:4 * I].view(torch.int32).cpu().numpy().view(np.uint32)
    seg = keys >> 16
    frac = keys & 0xFFFF
    assert (np.diff(seg.astype(np.int64)) >= 0).all(), "keys are not in partition order: is this the two-pass layout?"
    starts = np.flatnonzero(np.r_[True, seg[1:] != seg[:-1]])
    lens = np.diff(np.r_[starts, I])
    ntiles = ((W + 15) // 16) * ((H + 15) // 16)
    # entry-weighted: what the average ENTRY sees
    ew = np.repeat(lens, lens)
    print(f"== {kind}: P {P} I {I} tiles {ntiles} mean list {I // ntiles}  segments {len(lens)}  length 50/90/99/max {pct(lens)}  seen by an entry {pct(ew)}"
          f"  sum len^2 / I = {float((lens.astype(np.float64) ** 2).sum() / I):.0f}")
    for cap_l in (3072, 4096, 6144, 8192):
        print(f"   segments > {cap_l}: {(lens > cap_l).sum()}", end="")
    print()
    # the refinement: a segment of len >= 128 is cut by the top sb = floor(log2(len / 16)) bits of the 16 fraction bits
    sub_sq, direct_sq = 0.0, 0.0
    sub_max = 0
    longs = np.flatnonzero(lens >= 128)
    sample = longs if len(longs) <= 4000 else longs[np.random.default_rng(0).choice(len(longs), 4000, replace=False)]
    sub_lens = []
    for i in sample:
        s, n = starts[i], lens[i]
        sb = min(int(np.floor(np.log2(n / 16))), 8)
        c = np.bincount(frac[s:s + n] >> (16 - sb), minlength=1 << sb)
        sub_lens.append(c)
        sub_sq += float((c.astype(np.float64) ** 2).sum())
        direct_sq += float(n) ** 2
        sub_max = max(sub_max, int(c.max()))
    if len(sample):
        sl = np.concatenate(sub_lens)
        sw = np.repeat(sl, sl)
        print(f"   long segments (>= 128): {len(longs)} holding {lens[longs].sum() / I:.3f} of the entries; cut: sub-segments seen by an entry 50/90/99/max {pct(sw)}"
              f"  rank work {direct_sq / max(sub_sq, 1):.1f} x less  ties (equal fractions, share of entries of long segments) ", end="")
        ties = 0
        for i in sample[:400]:
            s, n = starts[i], lens[i]
            f = np.sort(frac[s:s + n])
            ties += int((np.r_[False, f[1:] == f[:-1]] | np.r_[f[1:] == f[:-1], False]).sum())
        print(f"{ties / max(int(lens[sample[:400]].sum()), 1):.4f}")
    # a bucket WINDOW per tile: the 23 bits of the global map below the tile id, re-cut per tile between the smallest
    # and the largest value a 6 % sample of the tile's entries holds — 128 buckets across the tile's own band instead of the scene's range
    g = (keys & 0x7FFFFF).astype(np.int64)
    tile = (keys >> 23).astype(np.int64)
    rng = np.random.default_rng(1)
    samp = rng.random(I) < 0.06
    lo = np.full(ntiles, 1 << 23, dtype=np.int64)
    hi = np.zeros(ntiles, dtype=np.int64)
    np.minimum.at(lo, tile[samp], g[samp])
    np.maximum.at(hi, tile[samp], g[samp])
    span = np.maximum(hi - lo + 1, 1)
    sh = np.maximum(np.ceil(np.log2(span / 128.0)), 0).astype(np.int64)
    b = np.clip((g - lo[tile]) >> sh[tile], 0, 127)
    wkey = tile * 128 + b
    order = np.argsort(wkey, kind="stable")
    ws = wkey[order]
    wstarts = np.flatnonzero(np.r_[True, ws[1:] != ws[:-1]])
    wlens = np.diff(np.r_[wstarts, I])
    wew = np.repeat(wlens, wlens)
    print(f"   per-tile windows (6 % sample): segments {len(wlens)}  length 50/90/99/max {pct(wlens)}  seen by an entry {pct(wew)}"
          f"  sum len^2 / I = {float((wlens.astype(np.float64) ** 2).sum() / I):.0f}  clamped below / above the sampled range {float((g < lo[tile]).mean()):.4f} / {float((g > hi[tile]).mean()):.4f}")
    print(_lib.load().das3r_last_error())
