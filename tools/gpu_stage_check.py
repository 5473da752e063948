#!/usr/bin/env python3
"""Stage-by-stage comparison of the HIP pipeline with the CPU oracle on the GPU box.  Diagnostic tool (the real
parity tests are tests/test_gpu_*.py): prints one line per intermediate buffer and never stops at the first mismatch.

    python tools/gpu_stage_check.py [--P 20000 --W 640 --H 360 --deg 3] [--reduce shfl]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def view(buf, off, dtype, count):
    itemsize = torch.tensor([], dtype=dtype).element_size()
    return buf[off:off + count * itemsize].view(dtype)


def report(name, a, b, tol=0.0):
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape:
        print(f"  {name:16s} SHAPE MISMATCH {a.shape} vs {b.shape}")
        return False
    if a.size == 0:
        print(f"  {name:16s} empty")
        return True
    if a.dtype.kind == "f":
        d = np.abs(a.astype(np.float64) - b.astype(np.float64))
        nbad = int((d > tol).sum())
        print(f"  {name:16s} max|d|={d.max():.3e} n(>{tol:g})={nbad}/{a.size} exact={bool((a == b).all())}")
        return nbad == 0
    nbad = int((a != b).sum())
    print(f"  {name:16s} mismatches={nbad}/{a.size}")
    return nbad == 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--P", type=int, default=20000)
    ap.add_argument("--W", type=int, default=640)
    ap.add_argument("--H", type=int, default=360)
    ap.add_argument("--focal", type=float, default=400.0)
    ap.add_argument("--deg", type=int, default=3)
    ap.add_argument("--seed", type=int, default=3)
    args = ap.parse_args()
    from das3r_amd import _lib, GaussianRasterizationSettings
    from das3r_amd.rasterizer import _forward_impl, _backward_impl
    from das3r_amd.synth import make_scene
    from oracle import c_oracle

    dev = torch.device("cuda:0")
    sc = make_scene(P=args.P, W=args.W, H=args.H, focal=args.focal, sh_degree=args.deg, seed=args.seed, bg=(0.3, 0.1, 0.2))
    scd = sc.to(dev)
    rs = GaussianRasterizationSettings(**scd.settings_kwargs())
    e = torch.empty(0, device=dev)
    print(f"scene P={sc.P} {sc.W}x{sc.H} deg={sc.sh_degree}")
    I, color, radii, geom, binning, img = _forward_impl(rs, scd.means3D, scd.shs, e, scd.opacities, scd.scales, scd.rotations, e)
    torch.cuda.synchronize()
    o = c_oracle.RasterOracle(**sc.settings_kwargs())
    rc, rr = o.forward(sc.means3D.numpy(), sc.opacities.numpy(), shs=sc.shs.numpy(), scales=sc.scales.numpy(), rotations=sc.rotations.numpy())
    S = o.saved()
    print(f"num_rendered hip={I} oracle={S['num_rendered']}")
    L = _lib.layout(sc.P, I, sc.W, sc.H)
    P, npix = sc.P, sc.W * sc.H
    vis = rr > 0
    ok = True
    ok &= report("radii", radii.cpu().numpy(), rr)
    ok &= report("tiles_touched", view(geom, L["tiles_touched"], torch.int32, P).cpu().numpy().astype(np.uint32), S["tiles_touched"])
    xy = _lib.splat_field(geom, L, "xy", P).cpu().numpy()[:, :2]
    ok &= report("xy", xy[vis], S["xy"][vis])
    co = _lib.splat_field(geom, L, "conic_opacity", P).cpu().numpy()
    ok &= report("conic_opacity", co[vis], S["conic_opacity"][vis])
    rgbd = _lib.splat_field(geom, L, "rgbd", P).cpu().numpy()
    ok &= report("rgb", rgbd[vis, :3], S["rgb"][vis])
    ok &= report("depth", rgbd[vis, 3], S["depths"][vis])
    cl = view(geom, L["clamped"], torch.uint8, P).cpu().numpy()
    clo = (S["clamped"][:, 0] | (S["clamped"][:, 1] << 1) | (S["clamped"][:, 2] << 2)).astype(np.uint8)
    ok &= report("clamped", cl[vis], clo[vis])
    # depth order
    sidx = view(geom, L["sorted_idx"], torch.int32, P).cpu().numpy()
    keys = np.where(vis, S["depths"].view(np.uint32), np.uint32(0xFFFFFFFF))
    ref_order = np.lexsort((np.arange(P), keys))
    ok &= report("sorted_idx", sidx, ref_order.astype(np.int32))
    if I == S["num_rendered"] and I > 0:
        pl = view(binning, L["point_list"], torch.int32, I).cpu().numpy()
        ok &= report("point_list", pl.astype(np.uint32), S["point_list"])
    tiles = S["ranges"].shape[0]
    rg = view(img, L["ranges"], torch.int32, 2 * tiles).cpu().numpy().reshape(tiles, 2)
    ok &= report("ranges", rg.astype(np.uint32), S["ranges"])
    ok &= report("n_contrib", view(img, L["n_contrib"], torch.int32, npix).cpu().numpy().reshape(sc.H, sc.W).astype(np.uint32), S["n_contrib"])
    ok &= report("final_T", view(img, L["final_T"], torch.float32, npix).cpu().numpy().reshape(sc.H, sc.W), S["final_T"], 1e-5)
    ok &= report("color", color.cpu().numpy(), rc, 1e-4)

    # backward, both reduction modes
    g_ref = o.backward(sc.dL_dpix.numpy())
    for mode in ("dpp", "shfl"):
        os.environ["DAS3R_BWD_REDUCE"] = mode
        _lib.reload_switches()
        out = _backward_impl(rs, I, scd.dL_dpix, scd.means3D, scd.shs, e, scd.opacities, scd.scales, scd.rotations, e, geom, binning, img)
        torch.cuda.synchronize()
        g_means2D, g_colors, g_opac, g_means3D, g_cov, g_sh, g_scales, g_rot = out
        print(f"backward (DAS3R_BWD_REDUCE={mode})")
        for name, t in [("means2D", g_means2D), ("opacities", g_opac), ("means3D", g_means3D), ("shs", g_sh), ("scales", g_scales),
                        ("rotations", g_rot)]:
            a, b = t.cpu().numpy().reshape(-1), g_ref[name].reshape(-1)
            scale = np.abs(b).max() + 1e-30
            d = np.abs(a - b) / scale
            print(f"  d{name:12s} max rel(to max)={d.max():.3e}  |ref|max={scale:.3e} nan={int(np.isnan(a).sum())}")
            ok &= bool(d.max() < 1e-3)
    print("ALL OK" if ok else "MISMATCHES PRESENT")


if __name__ == "__main__":
    main()
