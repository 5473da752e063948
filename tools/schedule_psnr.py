"""The 4000-iteration stand-in schedule of tests/test_gpu_trainstep.py, characterised instead of asserted:
for one seed, every requested arithmetic runs the SAME optimisation in lockstep (same camera order) and the last-epoch training
PSNR and the held-out PSNR of each are recorded.

  dense64   oracle/dense_trainer.py in float64 — what the test holds the product against
  dense32   the same restatement in float32: (dense32 - dense64) is the NOISE FLOOR of the comparison, i.e. how far two
            correct arithmetics of this optimisation drift apart (Adam with eps = 1e-15 steps by lr * sign(g) wherever the
            gradient is rounding noise around zero, so the trajectories decorrelate)
  hip       the product with the reference's PyTorch glue (unfused)
  fused     the product with the fused pre-transform / loss / Adam kernels (SH prefix between degree 0 and the maximum)
  fused2    a second, identical fused run (run-to-run spread: the pose sums of the fused pre-transform meet in float atomics)
  fused_full  fused, but the rasterizer gets the full [P, 16, 3] SH tensor at degree 1 (no fused._ShPrefix)

    python tools/schedule_psnr.py --seed 5 --variants dense64,dense32,hip,fused,fused_full --out gpurun_out/sched/s5.json
    python tools/schedule_psnr.py --launch 1,2,3,4,5,6 --tag r04          # one process per seed, in parallel, then the summary
DAS3R_DETERMINISTIC=1 in the environment selects the fixed-order kernels for the HIP variants (recorded in the output)."""
import argparse
import copy
import json
import os
import random
import subprocess
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PIPE = SimpleNamespace(debug=False, compute_cov3D_python=False, convert_SHs_python=False)


def run_seed(seed, variants, iters, frames=12, W=32, H=24):
    import torch
    from das3r_amd.model import OptimParams
    from das3r_amd.train import build_from_sequence, psnr_report, synthetic_sequence, train_step
    from oracle.dense_trainer import DenseTrainer
    seq = synthetic_sequence(frames=frames, W=W, H=H, focal=0.9 * W, n_splats=1500, seed=seed)
    opt = OptimParams(iterations=iters)
    runs = {}
    for v in variants:
        model, cams, test = build_from_sequence(copy.deepcopy(seq), heldout=True)
        if v.startswith("dense"):
            params = dict(xyz=model._xyz, f_dc=model._features_dc, f_rest=model._features_rest, opacity=model._opacity,
                          scaling=model._scaling, rotation=model._rotation, conf_static=model._conf_static, Q=model.Q, T=model.T,
                          mask=model.aggregated_mask)
            cameras = [dict(gt=c.original_image, fovx=c.FoVx, fovy=c.FoVy, proj_T=c.projection_matrix) for c in cams]
            dt = torch.float64 if v == "dense64" else torch.float32
            runs[v] = dict(kind="dense", tr=DenseTrainer(params, cameras, iterations=iters, dtype=dt), dt=dt, test=test,
                           pose=model.get_RT_test(0).detach().clone(), tail=[])
        else:
            fused = v.startswith("fused")
            model.training_setup(opt, fused=fused)
            if v == "fused_full":
                model.sh_prefix = False
            runs[v] = dict(kind="hip", model=model, cams=cams, test=test, fused=fused, tail=[])
    n_cams = len(cams)
    bg = torch.zeros(3, device="cuda")
    rng, stack = random.Random(0), []
    t0 = time.time()
    first = {v: [] for v in variants}
    for it in range(1, iters + 1):
        if not stack:
            stack = list(range(n_cams))
        uid = stack.pop(rng.randint(0, len(stack) - 1))
        for v, r in runs.items():
            if r["kind"] == "dense":
                loss, ps = r["tr"].step(it, uid, bg.to(r["dt"]))
            else:
                loss, ps, _ = train_step(r["model"], r["cams"][uid], opt, it, PIPE, bg, fused=r["fused"])
            if it <= 10:
                first[v].append(float(loss))
            if it > iters - n_cams:
                r["tail"].append(float(ps))
    out = dict(seed=seed, iters=iters, deterministic=os.environ.get("DAS3R_DETERMINISTIC", "0"), seconds=round(time.time() - t0, 1), runs={})
    for v, r in runs.items():
        if r["kind"] == "dense":
            held, _ = r["tr"].heldout_psnr(r["test"][0].original_image, r["pose"], 0, bg.to(r["dt"]))
        else:
            held = psnr_report(r["model"], r["test"], test_poses=True)["psnr"]
        out["runs"][v] = dict(train_psnr_last_epoch=sum(r["tail"]) / len(r["tail"]), heldout_psnr=held, first_losses=first[v])
    return out


def summarise(files, tag):
    import statistics as st
    recs = [json.load(open(f)) for f in files]
    rows = []
    dense = {r["seed"]: r["runs"]["dense64"] for r in recs if "dense64" in r["runs"]}   # (a record without its own takes the seed's)
    for r in recs:
        base = dense.get(r["seed"])
        if base is None:
            continue
        for v, x in r["runs"].items():
            if v == "dense64":
                continue
            rows.append(dict(seed=r["seed"], deterministic=r["deterministic"], variant=v,
                             d_train=x["train_psnr_last_epoch"] - base["train_psnr_last_epoch"],
                             d_heldout=x["heldout_psnr"] - base["heldout_psnr"],
                             train=x["train_psnr_last_epoch"], heldout=x["heldout_psnr"],
                             dense64_train=base["train_psnr_last_epoch"], dense64_heldout=base["heldout_psnr"]))
    stats = {}
    for key in sorted({(x["variant"], x["deterministic"]) for x in rows}):
        sel = [x for x in rows if (x["variant"], x["deterministic"]) == key]
        for q in ("d_train", "d_heldout"):
            vals = [x[q] for x in sel]
            stats[f"{key[0]}|det={key[1]}|{q}"] = dict(n=len(vals), mean=st.mean(vals), sd=st.pstdev(vals) if len(vals) > 1 else None,
                                                       rms=(sum(v * v for v in vals) / len(vals)) ** 0.5, max_abs=max(abs(v) for v in vals))
    return dict(what="(variant - dense64) of the last-epoch training PSNR and the held-out PSNR after the 4000-iteration stand-in schedule, dB",
                tag=tag, stats=stats, rows=rows)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=5)
    ap.add_argument("--variants", default="dense64,dense32,hip,fused,fused_full")
    ap.add_argument("--iters", type=int, default=4000)
    ap.add_argument("--out", default=None)
    ap.add_argument("--launch", default=None, help="comma-separated seeds: one child process per seed, all at once")
    ap.add_argument("--tag", default="run")
    ap.add_argument("--summarise", nargs="*", default=None)
    a = ap.parse_args()
    if a.summarise is not None:
        print(json.dumps(summarise(a.summarise, a.tag), indent=1))
        return
    if a.launch:
        d = os.path.join(ROOT, "gpurun_out", "sched")
        os.makedirs(d, exist_ok=True)
        det = os.environ.get("DAS3R_DETERMINISTIC", "0")
        procs, files = [], []
        for s in [int(x) for x in a.launch.split(",")]:
            f = os.path.join(d, f"{a.tag}_det{det}_s{s}.json")
            files.append(f)
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--seed", str(s), "--variants", a.variants,
                                           "--iters", str(a.iters), "--out", f], stdout=open(f + ".log", "w"), stderr=subprocess.STDOUT))
        rcs = [p.wait() for p in procs]
        print("child exit codes", rcs)
        done = [f for f in files if os.path.exists(f)]
        with open(os.path.join(d, f"{a.tag}_det{det}_summary.json"), "w") as fh:
            json.dump(summarise(done, a.tag), fh, indent=1)
        print(open(os.path.join(d, f"{a.tag}_det{det}_summary.json")).read()[:3000])
        return
    out = run_seed(a.seed, a.variants.split(","), a.iters)
    s = json.dumps(out, indent=1)
    if a.out:
        open(a.out, "w").write(s)
    print(s)


if __name__ == "__main__":
    main()
