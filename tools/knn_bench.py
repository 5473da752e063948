#!/usr/bin/env python3
"""distCUDA2 (simple_knn._C) timing at P = 100 k / 1 M / 5 M, uniform and DAS3R-shaped point sets, with the CPU baselines beside
it: scipy.spatial.cKDTree (exact, all host cores) and — at 100 k only — the exhaustive fp32 oracle.  One JSON line per case.
    python tools/knn_bench.py [--sizes 100000,1000000,5000000] [--json profiles/r02_knn.json]
Per-stage split (HIP events around every launch): bounding box, Morton keys, radix sort of the keys (4 x hist / row scan / scatter),
gather into Morton order, 1024-point boxes, exact search.  The search stage is what bounds the call: per 256-point workgroup it
tests every box against the workgroup's current bound and scans the surviving boxes' points from LDS — vector-ALU work (distance =
3 sub + 3 fma, a three-way insertion per candidate), not HBM traffic: the 16 P algorithmic bytes are noise next to it."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def points(kind, P, seed=0):
    g = np.random.default_rng(seed)
    if kind == "uniform":
        return (g.random((P, 3), dtype=np.float32) * np.array([3.0, 2.0, 8.0], np.float32)).astype(np.float32)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from tests.test_gpu_knn import das3r_shaped_points
    frames = max(1, round(P / (208 * 512)))
    return das3r_shaped_points(frames=frames)[:P]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="100000,1000000,5000000")
    ap.add_argument("--json", default=None)
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()
    from das3r_amd import distCUDA2
    from scipy.spatial import cKDTree
    rows = []
    for kind in ("uniform", "das3r"):
        for P in (int(s) for s in args.sizes.split(",")):
            pts = points(kind, P)
            P = pts.shape[0]
            d = torch.from_numpy(pts).cuda()
            for _ in range(2):
                out = distCUDA2(d)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 5
            for _ in range(n):
                out = distCUDA2(d)
            torch.cuda.synchronize()
            t = (time.perf_counter() - t0) / n
            row = {"points": kind, "P": P, "gpu_ms": round(t * 1e3, 3), "Mpoints_per_s": round(P / t / 1e6, 1), "alg_bytes": 16 * P}
            from das3r_amd import _lib
            _lib.profile_enable(True)
            for _ in range(3):
                distCUDA2(d)
            torch.cuda.synchronize()
            rep = _lib.profile_report()
            _lib.profile_enable(False)
            stage_of = {"knn_aabb_partial_kernel": "bounding_box", "knn_aabb_final_kernel": "bounding_box", "knn_morton_kernel": "morton_keys",
                        "radix_hist_kernel": "radix_sort", "radix_rowscan_kernel": "radix_sort", "radix_scatter_kernel": "radix_sort",
                        "knn_gather_kernel": "gather", "knn_box_kernel": "boxes", "knn_search_kernel": "search"}
            stages = {}
            for name, (n_l, ms) in rep.items():
                st = stage_of.get(name, name)
                stages[st] = round(stages.get(st, 0.0) + ms / 3.0, 4)
            row["stage_ms"] = dict(sorted(stages.items(), key=lambda kv: -kv[1]))
            row["bound"] = "vector ALU of the search stage (box tests + candidate scans), not HBM"
            if not args.no_cpu:
                p64 = pts.astype(np.float64)
                t1 = time.perf_counter()
                dd, _ = cKDTree(p64).query(p64, k=4, workers=-1)
                row["ckdtree_ms"] = round((time.perf_counter() - t1) * 1e3, 1)
                row["ckdtree_cores"] = os.cpu_count()
                ref = (dd[:, 1:] ** 2).mean(1)
                row["max_rel_err_vs_ckdtree"] = float(np.max(np.abs(out.cpu().numpy() - ref) / np.maximum(ref, 1e-20)))
                if P <= 100_000:
                    from oracle import c_oracle
                    t2 = time.perf_counter()
                    ex = c_oracle.knn3_mean_dist2(pts)
                    row["exhaustive_oracle_ms"] = round((time.perf_counter() - t2) * 1e3, 1)
                    row["bit_exact_vs_oracle"] = bool(np.array_equal(out.cpu().numpy(), ex))
            rows.append(row)
            print(json.dumps(row), flush=True)
    if args.json:
        with open(args.json, "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
