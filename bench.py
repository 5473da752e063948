#!/usr/bin/env python3
"""bench.py — throughput of the rasterizer hot path (forward + backward through the drop-in GaussianRasterizer API) on
synthetic random-splat scenes at 1080p, with the roofline of the dominant kernel and the CPU oracle timed beside it.

    python bench.py [--gpus N --steps K --warmup W] [--workload c2|c4|ds|c1]

A step = one forward + one backward of one scene (inputs resident in HBM).  With N > 1 (launched by torch.distributed.run,
one rank per GPU) every rank renders its OWN scene — independent sequences farmed across the node, no data-path
collective; RCCL is used only for the barrier and the max-over-ranks time — hence "scaling": "weak".
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

WORKLOAD_DESC = {
    "c1": "10k random splats, 256x256, SH degree 0, forward+backward (BASELINE.json configs[0] shape)",
    "c2": "100k random splats, 1920x1080, SH degree 3, forward+backward (BASELINE.json configs[1])",
    "c4": "1M random splats, 1920x1080, SH degree 3, forward+backward (BASELINE.json configs[3], no densification)",
    "ds": "5M random splats, 512x208, SH degree 0, forward+backward (shape of real DAS3R Sintel training)",
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--workload", default=os.environ.get("DAS3R_BENCH_WORKLOAD", "c2"), choices=sorted(WORKLOAD_DESC))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    from das3r_amd.hostpin import pin_to_ccx, unpin
    from das3r_amd.synth import WORKLOADS, make_scene
    cfg = dict(WORKLOADS[args.workload])
    cfg["seed"] = cfg["seed"] + 1000 * rank  # every rank = a different "sequence"
    sc_cpu = make_scene(**cfg)                # on the host, before the pin: torch's CPU thread pool keeps the full mask
    pinned = pin_to_ccx(local_rank)           # before the first HIP call: the runtime's threads inherit the mask
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)

    from das3r_amd import GaussianRasterizationSettings, GaussianRasterizer, _lib
    from das3r_amd.roofline import HBM_PEAK_GBS, algorithmic_bytes, group_kernel_times

    sc = sc_cpu.to(dev)
    rs = GaussianRasterizationSettings(**sc.settings_kwargs())
    rast = GaussianRasterizer(rs)
    leaves = {k: getattr(sc, k).clone().requires_grad_() for k in ("means3D", "opacities", "shs", "scales", "rotations")}
    means2D = torch.zeros(sc.P, 3, device=dev, requires_grad=True)
    dL = sc.dL_dpix
    all_leaves = list(leaves.values()) + [means2D]
    state = {}

    def step():
        for t in all_leaves:
            t.grad = None
        color, radii = rast(means3D=leaves["means3D"], means2D=means2D, opacities=leaves["opacities"], shs=leaves["shs"],
                            scales=leaves["scales"], rotations=leaves["rotations"])
        color.backward(dL)
        state["num_rendered"] = color.grad_fn.num_rendered if hasattr(color.grad_fn, "num_rendered") else None
        return color, radii

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    step()          # initialisation: first call loads the code objects, sizes torch's caching allocator and seeds the
    barrier()       # capacity cache of the sync-free forward (not a bench step)
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    msplats = world * sc.P / (elapsed / args.steps) / 1e6

    # ---- roofline of the dominant kernel: the same K steps again, every kernel launch bracketed by HIP events on its
    # launch stream (instrumented pass kept apart from the timed region so that `value` is not perturbed by the events)
    roofline, kernels_json, I = None, None, None
    if rank == 0:
        _lib.profile_enable(True)
        for _ in range(args.steps):
            color, _ = step()
        torch.cuda.synchronize()
        rep = _lib.profile_report()
        _lib.profile_enable(False)
        I = int(color.grad_fn.num_rendered)
        per_kernel, b_fwd, b_bwd = algorithmic_bytes(sc.P, sc.sh_degree, sc.shs.shape[1], I, sc.W, sc.H)
        grouped = group_kernel_times(rep)
        kernels_json = {}
        for name, (n, ms) in sorted(grouped.items(), key=lambda kv: -kv[1][1]):
            avg_ms = ms / args.steps  # per step (a 'binning' step = all its launches)
            ent = {"ms_per_step": round(avg_ms, 5), "launches_per_step": n / args.steps}
            if name in per_kernel:
                ent["alg_bytes"] = per_kernel[name]
                ent["GBps"] = round(per_kernel[name] / (avg_ms * 1e-3) / 1e9, 2)
            kernels_json[name] = ent
        # dominant KERNEL = the single kernel with the largest time per step (the 'binning' entry is a group of ~20 small
        # launches and is reported in `kernels`, not as a roofline kernel)
        dom = next(k for k in kernels_json if k in per_kernel and k != "binning")
        dom_ms = kernels_json[dom]["ms_per_step"]
        achieved = per_kernel[dom] / (dom_ms * 1e-3) / 1e9
        total_ms = sum(v["ms_per_step"] for v in kernels_json.values())
        traffic = None
        import glob
        pmc_files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_pmc_{args.workload}.json")))   # latest round
        pmc_path = pmc_files[-1] if pmc_files else ""
        if pmc_path:  # HBM bytes per launch from separate rocprofv3 --pmc passes (tools/pmc_summary.py)
            try:
                traffic = json.load(open(pmc_path)).get(dom, {}).get("hbm_bytes_per_launch")
            except Exception:  # noqa: BLE001
                traffic = None
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "alg_bytes_per_launch": per_kernel[dom],
                    "avg_launch_ms": round(dom_ms, 5),
                    "pipeline": {"alg_bytes_fwd_bwd": b_fwd + b_bwd, "kernel_ms_per_step": round(total_ms, 5),
                                 "GBps": round((b_fwd + b_bwd) / (total_ms * 1e-3) / 1e9, 2),
                                 "frac": round((b_fwd + b_bwd) / (total_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)},
                    "how": "HIP events around every launch on the launch stream, separate instrumented pass of the same K steps"}

    # ---- CPU baseline: the oracle (plain C + OpenMP restatement) on the host cores, same scene, rank 0, N=1 only
    cpu_baseline = None
    unpin(pinned)   # the CPU baseline below uses every host core
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import c_oracle
        o = c_oracle.RasterOracle(**sc_cpu.settings_kwargs())
        np_in = dict(shs=sc_cpu.shs.numpy(), scales=sc_cpu.scales.numpy(), rotations=sc_cpu.rotations.numpy())
        times = []
        t_budget = time.perf_counter()
        while len(times) < 3 and (time.perf_counter() - t_budget) < 20.0:
            t1 = time.perf_counter()
            o.forward(sc_cpu.means3D.numpy(), sc_cpu.opacities.numpy(), **np_in)
            o.backward(sc_cpu.dL_dpix.numpy())
            times.append(time.perf_counter() - t1)
        best = min(times)
        cpu_baseline = {"value": round(sc.P / best / 1e6, 4), "unit": "Msplats/s", "cores": c_oracle.max_threads(), "kind": "port",
                        "sample": f"whole {args.workload} scene ({sc.P} splats, {sc.W}x{sc.H}), fwd+bwd, best of {len(times)} runs, "
                                  f"{best * 1e3:.1f} ms per fwd+bwd",
                        "ms_per_step": round(best * 1e3, 2)}
        o.free()

    if rank == 0:
        out = {"metric": "render fwd+bwd Msplats/s at 1080p" if sc.W == 1920 else f"render fwd+bwd Msplats/s at {sc.W}x{sc.H}",
               "value": round(msplats, 3), "unit": "Msplats/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f32", "data": "synthetic",
               "config": {"workload": WORKLOAD_DESC[args.workload], "name": args.workload, "splats_per_gpu": sc.P,
                          "image": [sc.W, sc.H], "sh_degree": sc.sh_degree, "num_rendered": I,
                          "api": "GaussianRasterizer.forward + autograd backward (drop-in surface)",
                          "parallelism": f"{world} independent scenes, one per GPU",
                          "host_cpus": (f"pinned to CPUs {pinned[1][0]}-{pinned[1][-1]} (one core complex per rank)" if pinned
                                        else "not pinned")},
               "roofline": roofline, "cpu_baseline": cpu_baseline, "kernels": kernels_json}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
