#!/usr/bin/env python3
"""Sequences in flight per GPU: whole DAS3R jobs — model initialisation (distCUDA2), 4000 fused iterations with
the held-out pose passes, held-out PSNR report — K at a time on ONE MI355X (das3r_amd.farm.run_jobs: K host threads, each its own
stream and model), at the Sintel shape (22 frames of 512 x 208) and the DAVIS shape (50 frames of 512 x 288), on the self-consistent
synthetic sequences of das3r_amd.train.consistent_sequence (images, depth maps and poses from one scene + a moving object).

    python tools/jobs_per_gpu.py [--shapes sintel,davis] [--k 1,2,3] [--iterations 4000] --out profiles/r05_jobs_per_gpu.json

scenes_per_hour = K jobs / wall time of the K concurrent jobs (sequence generation excluded: it stands for the data on disk)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

SHAPES = {"sintel": dict(frames=22, W=512, H=208, focal=600.0, n_splats=20000),
          "davis": dict(frames=50, W=512, H=288, focal=614.4, n_splats=60000)}


def _proc_worker(k, K, name, iterations, barrier, q):
    """--mode process: one job per PROCESS (its own HIP context and interpreter: no GIL shared), all on GPU 0."""
    import os
    if K > 1:   # several PROCESSES on one GPU: the library cannot see its neighbours (it counts rendering threads per process), so the
        os.environ["DAS3R_TICKETS"] = "always"   # ticket-free short cut of the chained kernels is switched off by hand (api.hip grid_is_resident)
    import torch
    from das3r_amd.farm import run_sequence_job
    from das3r_amd.train import consistent_sequence
    dev = torch.device("cuda:0")
    run_sequence_job(0, 60, dev, fused=True, seq=consistent_sequence(frames=12, W=128, H=80, focal=150.0, n_splats=3000, seed=99, device="cuda:0"))
    seq = consistent_sequence(seed=k, device="cuda:0", **SHAPES[name])
    torch.cuda.synchronize()
    barrier.wait()
    t0 = time.perf_counter()
    rec = run_sequence_job(k, iterations, dev, fused=True, seq=seq)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    barrier.wait()
    q.put((k, t0, t1, rec, torch.cuda.max_memory_allocated()))


def process_mode(args):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    res = dict(iterations=args.iterations, mode="process", shapes={})
    for name in args.shapes.split(","):
        rows = {}
        for K in [int(k) for k in args.k.split(",")]:
            barrier, q = ctx.Barrier(K), ctx.Queue()
            procs = [ctx.Process(target=_proc_worker, args=(k, K, name, args.iterations, barrier, q)) for k in range(K)]
            for p in procs:
                p.start()
            got = sorted(q.get() for _ in range(K))
            for p in procs:
                p.join()
            wall = max(g[2] for g in got) - min(g[1] for g in got)   # (perf_counter is system-wide on Linux: CLOCK_MONOTONIC)
            rows[str(K)] = dict(wall_s=round(wall, 3), scenes_per_hour=round(K * 3600.0 / wall, 1), heldout_psnr=[round(g[3]["psnr"], 3) for g in got],
                                iters_per_s_per_job=[round(g[3]["iters_per_s"], 1) for g in got], peak_hbm_gb_per_process=[round(g[4] / 2 ** 30, 2) for g in got])
            print(name, "processes K =", K, rows[str(K)], flush=True)
        res["shapes"][name] = rows
    if args.out:
        with open(args.out, "w") as f:
            json.dump(res, f, indent=1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="thread", choices=("thread", "process"))
    ap.add_argument("--shapes", default="sintel,davis")
    ap.add_argument("--k", default="1,2,3")
    ap.add_argument("--iterations", type=int, default=4000)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    if args.mode == "process":
        return process_mode(args)
    from das3r_amd.farm import run_jobs, run_sequence_job
    from das3r_amd.train import consistent_sequence
    dev = torch.device("cuda:0")
    res = dict(iterations=args.iterations, what="K whole jobs in flight on one GPU (farm.run_jobs): init + fused iterations + held-out passes + report", shapes={})
    ks = [int(k) for k in args.k.split(",")]
    # one short job first: module load, allocator pools, the library's per-thread state — a farm pays them once per rank, not per sequence
    run_sequence_job(0, 60, dev, fused=True, seq=consistent_sequence(frames=12, W=128, H=80, focal=150.0, n_splats=3000, seed=99, device="cuda:0"))
    for name in args.shapes.split(","):
        cfg = SHAPES[name]
        seqs = [consistent_sequence(seed=s, device="cuda:0", **cfg) for s in range(max(ks))]
        torch.cuda.synchronize()
        rows = {}
        for K in ks:
            torch.cuda.empty_cache()
            torch.cuda.reset_peak_memory_stats()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            recs = run_jobs(range(K), lambda s: run_sequence_job(s, args.iterations, dev, fused=True, seq=seqs[s]), K, dev)
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
            rows[str(K)] = dict(wall_s=round(wall, 3), scenes_per_hour=round(K * 3600.0 / wall, 1), ok=[int(r["ok"]) for r in recs],
                                heldout_psnr=[round(r["psnr"], 3) for r in recs], iters_per_s_per_job=[round(r["iters_per_s"], 1) for r in recs],
                                n_splats=[int(r["n_splats"]) for r in recs], peak_hbm_gb=round(torch.cuda.max_memory_allocated() / 2 ** 30, 2))
            print(name, "K =", K, rows[str(K)], flush=True)
        for K in ks:
            rows[str(K)]["vs_k1"] = round(rows[str(K)]["scenes_per_hour"] / rows[str(ks[0])]["scenes_per_hour"], 3) if ks[0] == 1 else None
        res["shapes"][name] = dict(cfg, rows=rows)
        del seqs
    print(json.dumps(res, indent=1))
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
