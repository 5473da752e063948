#!/usr/bin/env python3
"""Per-kernel timing of the hot path for perf experiments (HIP events through das3r_profile_*).

    python tools/gpu_perf.py --workloads c2,c4[,c4:400000] --steps 10 [--env DAS3R_ABLATE=1 ...]
Each --env K=V variant is timed in turn (the library reads its experiment switches on every launch)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workloads", default="c2,c4")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--env", action="append", default=[])
    ap.add_argument("--fwd-only", action="store_true")
    args = ap.parse_args()
    from das3r_amd import GaussianRasterizationSettings, _lib
    from das3r_amd.rasterizer import _backward_impl, _forward_full
    from das3r_amd.synth import make_workload
    dev = torch.device("cuda:0")
    variants = [""] + args.env
    for w in args.workloads.split(","):
        wname, _, wp = w.partition(":")          # "c4:400000" = the c4 shape with 400 k splats
        sc = make_workload(wname, int(wp) if wp else None).to(dev)
        rs = GaussianRasterizationSettings(**sc.settings_kwargs())
        e = torch.empty(0, device=dev)
        for var in variants:
            saved_env = {}
            for kv in filter(None, var.split(",")):
                k, v = kv.split("=")
                saved_env[k] = os.environ.get(k)
                os.environ[k] = v
            _lib.reload_switches()
            _lib.forget_shapes()   # (what the library learnt about the previous workload of the same size does not apply)

            def step():
                I, color, radii, geom, binning, img, cap = _forward_full(rs, sc.means3D, sc.shs, e, sc.opacities, sc.scales, sc.rotations, e)
                if not args.fwd_only:
                    _backward_impl(rs, I, sc.dL_dpix, sc.means3D, sc.shs, e, sc.opacities, sc.scales, sc.rotations, e, geom, binning, img, cap)
                return I
            for _ in range(3):
                I = step()
            torch.cuda.synchronize()
            t0 = torch.cuda.Event(enable_timing=True)
            t1 = torch.cuda.Event(enable_timing=True)
            t0.record()
            for _ in range(args.steps):
                step()
            t1.record()
            torch.cuda.synchronize()
            wall = t0.elapsed_time(t1) / args.steps
            _lib.profile_enable(True)
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            rep = _lib.profile_report()
            _lib.profile_enable(False)
            tot = sum(ms for _, ms in rep.values()) / args.steps
            print(f"== {w} [{var or 'default'}] P={sc.P} I={I} step(wall, uninstrumented)={wall:.4f} ms  sum(kernels)={tot:.4f} ms")
            for name, (n, ms) in sorted(rep.items(), key=lambda kv: -kv[1][1]):
                print(f"   {name:32s} {ms / args.steps:9.4f} ms/step  ({n // args.steps} launches)")
            _lib.pair_counters(True)
            step()
            torch.cuda.synchronize()
            pc = _lib.pair_counters(False)
            print(f"   pair slots per instance: forward {pc['fwd_pairs'] / max(I, 1):.1f}  backward {pc['bwd_pairs'] / max(I, 1):.1f}")
            for k, v in saved_env.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
            _lib.reload_switches()


if __name__ == "__main__":
    main()
