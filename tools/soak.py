#!/usr/bin/env python3
"""Soak test of the forward+backward path: many back-to-back steps, any library error (incl. the deferred binning self-check)
is counted and printed in full.   python tools/soak.py --workloads c2,c4,ds --steps 3000"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workloads", default="c2,c4,ds")
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--fwd-only", action="store_true", help="forwards only (each one is still checked: rasterizer.check_forward)")
    ap.add_argument("--seconds", type=float, default=0.0, help="per workload: keep going for this long instead of --steps")
    ap.add_argument("--json", default=None, help="write the record (per workload: forwards, errors, image mismatches; library counters) here")
    ap.add_argument("--jitter", action="store_true", help="vary the scale modifier from step to step (num_rendered jumps: the speculative binning capacity overflows now and then) and check every image against an exact-capacity reference")
    args = ap.parse_args()
    import json
    from das3r_amd import GaussianRasterizationSettings, _lib
    from das3r_amd.rasterizer import _backward_impl, _forward_full, check_forward
    record = {"workloads": {}}
    from das3r_amd.synth import make_workload
    dev = torch.device("cuda:0")
    e = torch.empty(0, device=dev)
    for w in args.workloads.split(","):
        _lib.forget_shapes()   # (`ds` and `dsc` are one (P, W, H): what the library learnt on one must not choose the other's kernels — the image of a
        #  workload is compared with ITS first forward, and the compositing kernels differ in the last bit of T)
        sc = make_workload(w).to(dev)
        rs = GaussianRasterizationSettings(**sc.settings_kwargs())
        errs, counts, t0 = [], set(), time.perf_counter()
        ref_color, mismatches = None, 0
        if args.jitter:
            mods = [0.6, 0.8, 1.0, 1.3, 1.7]
            refs = {}
            for m in mods:
                r = _forward_full(rs._replace(scale_modifier=m), sc.means3D, sc.shs, e, sc.opacities, sc.scales, sc.rotations, e, exact=True)
                refs[m] = (r[0], r[1].clone())
            gen = torch.Generator().manual_seed(1)
            redone = 0
            for it in range(args.steps):
                m = mods[int(torch.randint(len(mods), (1,), generator=gen))]
                I, color, radii, geom, binning, img, cap = _forward_full(rs._replace(scale_modifier=m), sc.means3D, sc.shs, e, sc.opacities, sc.scales, sc.rotations, e)
                _backward_impl(rs._replace(scale_modifier=m), I, sc.dL_dpix, sc.means3D, sc.shs, e, sc.opacities, sc.scales, sc.rotations, e, geom, binning, img, cap)
                redone += cap == I
                if I != refs[m][0] or not torch.equal(color, refs[m][1]):
                    mismatches += 1
            torch.cuda.synchronize()
            print(f"{w} jitter: {args.steps} steps in {time.perf_counter() - t0:.1f} s, counts {sorted(v[0] for v in refs.values())}, "
                  f"exactly sized (first / redone) {redone}, mismatches {mismatches}")
            record["workloads"][w] = dict(jitter=True, steps=args.steps, seconds=round(time.perf_counter() - t0, 1), scale_modifiers=mods,
                                          num_rendered_values=sorted(v[0] for v in refs.values()), exactly_sized_or_redone=int(redone),
                                          image_mismatches=mismatches)
            continue
        it = 0
        while (time.perf_counter() - t0 < args.seconds) if args.seconds > 0 else (it < args.steps):
            try:
                I, color, radii, geom, binning, img, cap = _forward_full(rs, sc.means3D, sc.shs, e, sc.opacities, sc.scales, sc.rotations, e)
                counts.add(I)
                if ref_color is None:
                    ref_color = color.clone()
                elif it % 25 == 0 and not torch.equal(color, ref_color):   # the image is deterministic: any difference is a bug
                    mismatches += 1
                if args.fwd_only:
                    if it % 8 == 7:
                        check_forward(cap, dev)   # (earlier forwards' words are examined, without waiting, by the forwards themselves)
                else:
                    _backward_impl(rs, I, sc.dL_dpix, sc.means3D, sc.shs, e, sc.opacities, sc.scales, sc.rotations, e, geom, binning, img, cap)
            except RuntimeError as ex:
                errs.append((it, str(ex)))
            it += 1
        torch.cuda.synchronize()
        steps_done = it
        record["workloads"][w] = {"forwards": steps_done, "backwards": 0 if args.fwd_only else steps_done, "seconds": round(time.perf_counter() - t0, 1),
                                  "num_rendered_values": sorted(counts), "errors": len(errs), "image_mismatches": mismatches,
                                  "first_errors": [m for _, m in errs[:3]]}
        print(f"{w}: {steps_done} steps in {time.perf_counter() - t0:.1f} s, num_rendered values {sorted(counts)}, errors {len(errs)}, image mismatches {mismatches}")
        for it, msg in errs[:5]:
            print("   step", it, msg)
    record["library_counters"] = _lib.stats()
    print("library counters:", record["library_counters"])
    if args.json:
        os.makedirs(os.path.dirname(os.path.abspath(args.json)), exist_ok=True)
        with open(args.json, "w") as f:
            json.dump(record, f, indent=1)


if __name__ == "__main__":
    main()
