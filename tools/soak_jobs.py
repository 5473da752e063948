#!/usr/bin/env python3
"""Soak of K jobs in flight on one GPU (farm.run_jobs): `--rounds` times, K whole (short) jobs at once on Sintel-shaped self-consistent
sequences; every job must finish, and — the direct iteration being bit-reproducible — every job must reproduce the held-out PSNR and L1
of its solo run EXACTLY, whichever neighbours it had.  Library counters (failed self-checks, rescued polls) are recorded.

    python tools/soak_jobs.py --rounds 12 --k 2 --iterations 1000 --json profiles/r05_soak_jobs_in_flight.json"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=12)
    ap.add_argument("--k", type=int, default=2)
    ap.add_argument("--iterations", type=int, default=1000)
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    from das3r_amd import _lib
    from das3r_amd.farm import run_jobs, run_sequence_job
    from das3r_amd.train import consistent_sequence
    dev = torch.device("cuda:0")
    shape = dict(frames=22, W=512, H=208, focal=600.0, n_splats=20000)
    nseq = args.k + 1
    seqs = [consistent_sequence(seed=s, device="cuda:0", **shape) for s in range(nseq)]
    solo = [run_sequence_job(s, args.iterations, dev, fused=True, seq=seqs[s]) for s in range(nseq)]
    assert all(r["ok"] == 1 for r in solo), solo
    before = _lib.stats()
    failed, differing, jobs, t0 = 0, 0, 0, time.perf_counter()
    for rnd in range(args.rounds):
        ids = [(rnd + j) % nseq for j in range(args.k)]   # neighbours change from round to round
        recs = run_jobs(ids, lambda s: run_sequence_job(s, args.iterations, dev, fused=True, seq=seqs[s]), args.k, dev)
        for s, r in zip(ids, recs):
            jobs += 1
            failed += r["ok"] != 1
            differing += r["ok"] == 1 and (r["psnr"] != solo[s]["psnr"] or r["l1"] != solo[s]["l1"])
    torch.cuda.synchronize()
    after = _lib.stats()
    rec = dict(k=args.k, rounds=args.rounds, iterations_per_job=args.iterations, jobs=jobs, failed_jobs=int(failed),
               jobs_differing_from_their_solo_run=int(differing), seconds=round(time.perf_counter() - t0, 1),
               solo_heldout_psnr=[round(r["psnr"], 4) for r in solo], n_splats=solo[0]["n_splats"],
               library_counters={k: after[k] - before[k] for k in after})
    print(json.dumps(rec, indent=1))
    if args.json:
        with open(args.json, "w") as f:
            json.dump(rec, f, indent=1)
    sys.exit(1 if failed or differing else 0)


if __name__ == "__main__":
    main()
