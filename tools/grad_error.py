#!/usr/bin/env python3
"""Gradient error of the backward compositing kernels against the CPU oracle at the benchmarked sizes.

    python tools/grad_error.py --workload c4 --kinds default,dpp,scan128 --out profiles/r03_grad_error_c4.json

Per kernel kind (DAS3R_RENDER_BWD) and gradient tensor: the max-norm error relative to max|ref|, the 50 / 99 / 99.9 / 100th
percentile of the element-wise relative error |hip - ref| / |ref| over the elements with |ref| above 1e-4 max|ref| (elements
that are sums of cancelling terms are excluded: their relative error measures the summation order, not the kernel), and the
MEAN SIGNED relative error over the same elements — a one-sided rounding (e.g. operands truncated instead of rounded) shows
up there as a bias of the size of its step, a symmetric one averages out.  Test infrastructure: runs the oracle."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def stats(a, b):
    a, b = np.asarray(a, np.float64).reshape(-1), np.asarray(b, np.float64).reshape(-1)
    scale = float(np.abs(b).max())
    if scale == 0.0:
        return dict(max_ref=0.0, max_norm_rel=float(np.abs(a).max()))
    keep = np.abs(b) > 1e-4 * scale
    rel = (a[keep] - b[keep]) / np.abs(b[keep])
    ar = np.abs(rel)
    out = dict(max_ref=scale, max_norm_rel=float(np.abs(a - b).max() / scale), elements=int(keep.sum()),
               p50=float(np.percentile(ar, 50)), p99=float(np.percentile(ar, 99)), p999=float(np.percentile(ar, 99.9)),
               p100=float(ar.max()), mean_signed_rel=float((rel * np.sign(b[keep])).mean()),
               mean_abs_rel=float(ar.mean()))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c4")
    ap.add_argument("--P", type=int, default=None)
    ap.add_argument("--kinds", default="default,dpp,scan128")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    from das3r_amd import GaussianRasterizationSettings, GaussianRasterizer, _lib
    from das3r_amd.synth import make_workload
    from tests import util
    dev = torch.device("cuda:0")
    sc = make_workload(args.workload, args.P)
    scd = sc.to(dev)
    mode = dict(colors_precomp=False, cov3D_precomp=False, scale_modifier=1.0)
    ref_color, ref_radii, ref_g, S = util.run_oracle(sc, mode)
    res = dict(workload=args.workload, P=sc.P, W=sc.W, H=sc.H, sh_degree=sc.sh_degree, oracle_num_rendered=int(S["num_rendered"]), kinds={})
    for kind in args.kinds.split(","):
        if kind == "default":
            os.environ.pop("DAS3R_RENDER_BWD", None)
        else:
            os.environ["DAS3R_RENDER_BWD"] = kind
        _lib.reload_switches()
        leaves = {k: getattr(scd, k).clone().requires_grad_() for k in ("means3D", "opacities", "shs", "scales", "rotations")}
        m2 = torch.zeros(sc.P, 3, device=dev, requires_grad=True)
        color, radii = GaussianRasterizer(GaussianRasterizationSettings(**scd.settings_kwargs()))(means2D=m2, **leaves)
        color.backward(scd.dL_dpix)
        torch.cuda.synchronize()
        per = {}
        for k, t in list(leaves.items()) + [("means2D", m2)]:
            g = t.grad.cpu().numpy()
            r = ref_g[k]
            if k == "shs":   # only the active coefficients carry gradient
                n = (sc.sh_degree + 1) ** 2
                g, r = g[:, :n], r[:, :n]
            if k == "means2D":
                g, r = g[:, :2], r[:, :2]
            per[k] = stats(g, r)
        res["kinds"][kind] = dict(num_rendered=int(color.grad_fn.num_rendered),
                                  colour_max_abs_err=float(np.abs(color.detach().cpu().numpy() - ref_color).max()), grads=per)
        print(kind, json.dumps({k: (v["max_norm_rel"], v.get("p99"), v.get("mean_signed_rel")) for k, v in per.items()}), flush=True)
    os.environ.pop("DAS3R_RENDER_BWD", None)
    _lib.reload_switches()
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
