#!/usr/bin/env python3
"""Where a tile's time goes inside the two compositing kernels (make EXPERIMENTS=1 builds only).

    python tools/phase_clocks.py --workloads c4,ds --steps 5
Wave 0 of every workgroup adds the shader clocks between its phase marks to per-phase words (common.h PHASE_MARK); the table is
the average per tile and the share of the workgroup's residency.  The marks cost a few s_memtime per phase: read the shares,
not the kernel time."""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

FWD = {6: "range, list words, depth keys", 7: "sort network", 0: "sorted lists back + rest of prologue", 1: "wait for slowest wave", 2: "stage the batch", 3: "row lists", 4: "walk", 5: "output"}
BWD = {8: "prologue", 14: "zero the tail rows", 9: "stage the round", 10: "row lists", 11: "batches", 12: "wait for slowest wave",
       13: "write-out"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workloads", default="c4")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    from das3r_amd import GaussianRasterizationSettings, _lib
    from das3r_amd.rasterizer import _backward_impl, _forward_full
    from das3r_amd.synth import make_workload
    if not _lib.has_experiments():
        sys.exit("phase clocks need the experiments build: make -C das3r_amd/csrc EXPERIMENTS=1")
    L = _lib.load()
    L.das3r_debug_phase_clocks.restype = C.c_int
    L.das3r_debug_phase_clocks.argtypes = [C.POINTER(C.c_uint64)]
    dev = torch.device("cuda:0")
    out = {}
    for w in args.workloads.split(","):
        sc = make_workload(w).to(dev)
        rs = GaussianRasterizationSettings(**sc.settings_kwargs())
        e = torch.empty(0, device=dev)

        def step():
            I, color, radii, geom, binning, img, cap = _forward_full(rs, sc.means3D, sc.shs, e, sc.opacities, sc.scales, sc.rotations, e)
            _backward_impl(rs, I, sc.dL_dpix, sc.means3D, sc.shs, e, sc.opacities, sc.scales, sc.rotations, e, geom, binning, img, cap)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        _lib.pair_counters(True)
        for _ in range(args.steps):
            step()
        words = (C.c_uint64 * 16)()
        _lib.check(L.das3r_debug_phase_clocks(words), "das3r_debug_phase_clocks")
        _lib.pair_counters(False)
        tiles = ((rs.image_width + 15) // 16) * ((rs.image_height + 15) // 16)
        rows = {}
        for title, names in (("forward (render_forward_rows_kernel)", FWD), ("backward (render_backward_blk_kernel)", BWD)):
            total = sum(int(words[k]) for k in names)
            print(f"== {w}: {title}: {total / args.steps / tiles:.0f} clocks per tile")
            for k, name in names.items():
                v = int(words[k])
                print(f"   {name:28s} {v / args.steps / tiles:10.0f} clocks  {100.0 * v / max(total, 1):5.1f} %")
                rows[name + (" (fwd)" if names is FWD else " (bwd)")] = v / args.steps / tiles
        out[w] = rows
    if args.json:
        with open(args.json, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
