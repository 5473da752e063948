#!/usr/bin/env python3
"""bench.py — throughput of the rasterizer hot path (forward + backward through the drop-in GaussianRasterizer API) on
synthetic random-splat scenes, with the roofline of the dominant kernel and the CPU oracle timed beside it.

    python bench.py [--gpus N --steps K --warmup W] [--workload c4|c2|ds|dsc|c1|c4d] [--no-extras] [--no-cpu-baseline]

Default workload: c4 = BASELINE.json configs[3] (1 M splats at 1080p), the configuration north_star's roofline target is
stated on.  A step = one forward + one backward of one scene (inputs resident in HBM).  With N > 1 every rank renders its OWN
scene — independent sequences farmed across the node, no data-path collective; RCCL is used only for the
barrier and the max-over-ranks time — hence "scaling": "weak".  `--gpus N` without a torch.distributed environment starts
the N ranks itself (re-executes under `python -m torch.distributed.run --nproc-per-node N`); under torchrun it checks that
WORLD_SIZE == N.  Prints ONE JSON line on rank 0.

Extra keys beside the contract's: `roofline`, `cpu_baseline`, `kernels` (per-kernel time and algorithmic GB/s), `extras`
(N = 1 only, skipped with --no-extras: the other BASELINE shapes c2 / ds / c1, c4 with the densification stress, and
train-step ms of the DAS3R-shaped optimisation step, fused and unfused), `train_step_ms` + `scenes_per_hour` (every N: the
farm's unit of work is a 4000-iteration optimisation of one sequence, scripts/testing_psnr_davis.sh:35-59), `jobs_in_flight`
(N = 1: WHOLE jobs with K sequences in flight on the GPU — wall time, scenes per hour, held-out PSNR, peak HBM; Sintel shape with
K = 1, 2 by default, both shapes with K = 1, 2, 3 under DAS3R_BENCH_JOBS=full, nothing under DAS3R_BENCH_JOBS=0; when present,
`scenes_per_hour` is the measured figure), and per compositing kernel `live_pairs_per_instance` / `padded_work` / `valu_roofline`.
"""
import argparse
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOAD_DESC = {
    "c1": "10k random splats, 256x256, SH degree 0, forward+backward (BASELINE.json configs[0] shape)",
    "c2": "100k random splats, 1920x1080, SH degree 3, forward+backward (BASELINE.json configs[1])",
    "c4": "1M random splats, 1920x1080, SH degree 3, forward+backward (BASELINE.json configs[3], constant P)",
    "c4d": "1M -> 1.3M random splats (clone/split of the top-gradient 5 % every 100 steps), 1920x1080, SH degree 3, "
           "forward+backward (BASELINE.json configs[3] with its densification stress, SURVEY.md §8d)",
    "ds": "5M random splats, 512x208, SH degree 0, forward+backward (shape of real DAS3R Sintel training)",
    "dsc": "5M splats on a smooth depth relief (1 % noise), 512x208, SH degree 0, forward+backward (the DAS3R shape with the spatially "
           "coherent depth of a real scene: a tile sees a thin depth band)",
}
INIT_STEPS = 30
ITERS_PER_SCENE = 4000   # BASELINE.json configs[2] / [4]: one sequence = 4000 optimisation iterations


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--workload", default=os.environ.get("DAS3R_BENCH_WORKLOAD", "c4"), choices=sorted(WORKLOAD_DESC))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--extras-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 --pmc child passes (roofline.traffic then quotes profiles/)")
    ap.add_argument("--full-line", action="store_true", help="put the per-workload extras into the JSON line itself instead of stderr")
    ap.add_argument("--stub", action="store_true",
                    help="CPU dry run of the launch / barrier / reduction logic with a stand-in step (gloo; tests/test_bench_launch.py)")
    return ap.parse_args(argv)


def respawn_under_torchrun(args):
    """`python bench.py --gpus N` with no torch.distributed environment: become N ranks (one per GPU)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.execv(sys.executable, cmd)


class Ranks:
    """Rank bookkeeping, barrier and max-over-ranks (RCCL on GPUs, gloo in the stub run)."""

    def __init__(self, args):
        import torch
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if self.world != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={self.world}: launch with --nproc-per-node {args.gpus} "
                             f"(or without torchrun: bench.py starts the ranks itself)")
        self.stub = args.stub
        self.dist = None
        self.failed = None      # reason, once a step of this rank has raised
        self.last_local_s = 0.0
        if self.stub:
            self.dev = torch.device("cpu")
        else:
            if not torch.cuda.is_available():
                raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
            if self.local_rank >= torch.cuda.device_count():
                raise SystemExit(f"bench.py: rank {self.rank} has no GPU (local rank {self.local_rank}, {torch.cuda.device_count()} visible)")
            torch.cuda.set_device(self.local_rank)
            self.dev = torch.device("cuda", self.local_rank)
        if self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if self.stub:
                dist.init_process_group(backend="gloo")
            else:
                dist.init_process_group(backend="nccl", device_id=self.dev)
            self.dist = dist
            assert dist.get_world_size() == args.gpus

    def sync(self):
        import torch
        if self.dev.type == "cuda":
            torch.cuda.synchronize()

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()
        self.sync()

    def max_over_ranks(self, x):
        import torch
        if self.dist is None:
            return float(x)
        t = torch.tensor([x], device=self.dev, dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def timed(self, step, steps, warmup):
        """W untimed steps, then exactly K steps bracketed by barrier + synchronize on both sides; MAX over ranks.  -> seconds.
        A rank whose step raises keeps its appointments (barriers, reductions) so that the others' line still comes out; it is
        reported through `ranks_ok` (self.failed holds the reason) and its units do not count."""
        def guarded():
            if self.failed is None:
                try:
                    step()
                except Exception as ex:  # noqa: BLE001 - reported, not swallowed: ranks_ok / stderr
                    self.failed = repr(ex)
                    print(f"bench.py: rank {self.rank} failed: {self.failed}", file=sys.stderr, flush=True)
        for _ in range(warmup):
            guarded()
        self.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            guarded()
        self.sync()
        self.last_local_s = time.perf_counter() - t0   # this rank's own time, before the closing barrier
        self.barrier()
        return self.max_over_ranks(time.perf_counter() - t0)

    def gather_all(self, x):
        """-> [x of rank 0, x of rank 1, ...] on every rank"""
        import torch
        if self.dist is None:
            return [float(x)]
        t = torch.zeros(self.world, device=self.dev, dtype=torch.float64)
        t[self.rank] = float(x)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return [round(float(v), 4) for v in t.tolist()]

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()


class RasterJob:
    """One scene on one GPU: step() = forward + backward through the drop-in autograd surface."""

    def __init__(self, name, dev, seed_offset=0):
        import torch
        from das3r_amd import GaussianRasterizationSettings
        from das3r_amd.densify import GrowthSchedule
        from das3r_amd.synth import WORKLOADS, make_scene
        self.name = name
        cfg = dict(WORKLOADS["c4" if name == "c4d" else name])
        cfg["seed"] = cfg["seed"] + seed_offset   # every rank = a different "sequence"
        self.sc_cpu = make_scene(**cfg)           # on the host, before the pin: torch's CPU thread pool keeps the full mask
        self.dev = dev
        self.growth = GrowthSchedule(self.sc_cpu.P) if name == "c4d" else None
        self.gen = torch.Generator().manual_seed(1234 + seed_offset)
        self.events = []
        self._rs_cls = GaussianRasterizationSettings
        self.steps_done = 0

    def upload(self):
        import torch
        from das3r_amd import GaussianRasterizer
        from das3r_amd import _lib
        _lib.forget_shapes()   # a new scene: what the library learnt about another one of the same size (`ds` / `dsc`: one (P, W, H)) does not apply
        self.sc = self.sc_cpu.to(self.dev)
        self.rast = GaussianRasterizer(self._rs_cls(**self.sc.settings_kwargs()))
        self._set_leaves({k: getattr(self.sc, k) for k in ("means3D", "opacities", "shs", "scales", "rotations")})
        self.dL = self.sc.dL_dpix
        self.num_rendered = None
        self._torch = torch

    def _set_leaves(self, tensors):
        torch = __import__("torch")
        self.leaves = {k: v.detach().clone().requires_grad_() for k, v in tensors.items()}
        self.means2D = torch.zeros(self.leaves["means3D"].shape[0], 3, device=self.dev, requires_grad=True)
        self.all_leaves = list(self.leaves.values()) + [self.means2D]

    @property
    def P(self):
        return self.leaves["means3D"].shape[0]

    def step(self):
        L = self.leaves
        if self.growth is not None and self.growth.due(self.steps_done, self.P):   # uses the gradients of the previous step
            from das3r_amd.densify import grow_top_gradient
            with self._torch.no_grad():
                grown = grow_top_gradient({k: v.detach() for k, v in L.items()}, self.means2D.grad, self.growth.frac_for(self.P), self.gen)
            self._set_leaves(grown)
            L = self.leaves
            self.events.append((self.steps_done, self.P))
        for t in self.all_leaves:
            t.grad = None
        color, radii = self.rast(means3D=L["means3D"], means2D=self.means2D, opacities=L["opacities"], shs=L["shs"], scales=L["scales"],
                                 rotations=L["rotations"])
        color.backward(self.dL)
        self.num_rendered = int(color.grad_fn.num_rendered)
        self.steps_done += 1
        return color

    def forward_only(self):
        """The forward render alone (BASELINE.json's metric names "render fwd/bwd Msplats/s": this is the fwd figure): the same call as
        in step(), graph and saved buffers included, without the backward.  (Under torch.no_grad the rasterizer would wait for the
        forward's binning self-check on every call — what an evaluation render wants, not what a training forward does.)"""
        L = self.leaves
        color, radii = self.rast(means3D=L["means3D"], means2D=self.means2D, opacities=L["opacities"], shs=L["shs"], scales=L["scales"],
                                 rotations=L["rotations"])
        return color


def kernel_times(job, steps):
    """The same K steps again with every kernel launch bracketed by HIP events on its launch stream (das3r_profile_*), kept apart
    from the timed region so that `value` is not perturbed — and taken RIGHT BEHIND it, while the device is still at the clocks of
    the timed steps (r2: taken after the profiler's child passes, the compositing kernels came out 6 % slower than in the timed
    region they had just run in).  -> ({kernel: (launches, total ms)}, pairs per step {kernel: count})"""
    import torch
    from das3r_amd import _lib
    _lib.profile_enable(True)
    for _ in range(10):   # the instrumented pass warms up too (event pool)
        job.step()
    torch.cuda.synchronize()
    _lib.profile_report()
    for _ in range(steps):
        job.step()
    torch.cuda.synchronize()
    rep = _lib.profile_report()
    _lib.profile_enable(False)
    # (pixel, splat) pairs the two compositing kernels evaluate per step: counted by the kernels themselves in a pass of their own
    _lib.pair_counters(True)
    for _ in range(3):
        job.step()
    pc = _lib.pair_counters(False)
    pairs = {"render_forward_kernel": pc["fwd_pairs"] / 3.0, "render_backward_kernel": pc["bwd_pairs"] / 3.0}
    # the LIVE pairs of the same scene (the pairs that are blended: position < n_contrib, power <= 0, alpha >= 1/255), counted from a
    # forward's saved buffers by a plain walk (das3r_raster_count_live_pairs): evaluated / live = the padded-work ratio of a kernel
    try:
        from das3r_amd.rasterizer import _forward_full, count_live_pairs
        L = job.leaves
        e = torch.empty(0, device=job.dev)
        with torch.no_grad():
            I, _c, _r, geom, binning, img, cap = _forward_full(job.rast.raster_settings, L["means3D"].detach(), L["shs"].detach(), e, L["opacities"].detach(),
                                                               L["scales"].detach(), L["rotations"].detach(), e)
            live, below = count_live_pairs(job.rast.raster_settings, job.P, L["shs"].shape[1], I, geom, binning, img, cap)
        pairs["live_pairs"], pairs["pairs_below_last_contributor"] = live, below
    except Exception as ex:  # noqa: BLE001 - a measurement aid never takes the line down
        pairs["live_pairs_error"] = repr(ex)
    return rep, pairs


def kernel_table(job, steps, measured=None, timed=None):
    """-> (kernels dict, roofline dict) from kernel_times() (taken now unless `timed` holds them) and the measured HBM traffic."""
    import glob
    import torch
    from das3r_amd import _lib
    from das3r_amd.roofline import HBM_PEAK_GBS, algorithmic_bytes, group_kernel_times
    sc = job.sc
    rep, pairs = timed if timed is not None else kernel_times(job, steps)
    I = job.num_rendered
    per_kernel, b_fwd, b_bwd = algorithmic_bytes(job.P, sc.sh_degree, sc.shs.shape[1], I, sc.W, sc.H)
    kernels = {}
    for name, (n, ms) in sorted(group_kernel_times(rep).items(), key=lambda kv: -kv[1][1]):
        avg_ms = ms / steps   # per step (the 'binning' entry = all its launches)
        ent = {"ms_per_step": round(avg_ms, 5)}
        if name in per_kernel:
            ent["alg_bytes"] = per_kernel[name]
            ent["GBps"] = round(per_kernel[name] / (avg_ms * 1e-3) / 1e9, 1)
        if name in pairs and pairs[name] > 0:
            # "additionally report pairs/s so a shortfall can be attributed" — pairs the kernel EVALUATES (alpha computed)
            ent["pairs_per_step"] = int(pairs[name])
            ent["pairs_per_instance"] = round(pairs[name] / max(I, 1), 1)
            ent["Gpairs_per_s"] = round(pairs[name] / (avg_ms * 1e-3) / 1e9, 2)
            if pairs.get("live_pairs"):
                from das3r_amd.roofline import valu_roofline
                ent["live_pairs_per_instance"] = round(pairs["live_pairs"] / max(I, 1), 1)
                ent["padded_work"] = round(pairs[name] / pairs["live_pairs"], 3)   # pairs evaluated per pair that is blended
                ent["valu_roofline"] = valu_roofline(name, pairs["live_pairs"], avg_ms)
        kernels[name] = ent
    # dominant KERNEL = the single kernel with the largest time per step (the 'binning' entry is a group of small launches)
    dom = next(k for k in kernels if k in per_kernel and k != "binning")
    dom_ms = kernels[dom]["ms_per_step"]
    achieved = per_kernel[dom] / (dom_ms * 1e-3) / 1e9
    total_ms = sum(v["ms_per_step"] for v in kernels.values())
    # HBM bytes per launch: PMC counters need rocprofv3 passes of their own — measure_traffic() runs them as children of this
    # run when it can; otherwise the committed summary of the latest round is quoted, with its source
    traffic, traffic_source = None, None
    wname = "c4" if job.name == "c4d" else job.name
    if measured is not None and measured[0]:
        for k, ent in kernels.items():
            if k in measured[0]:
                # FETCH_SIZE correction per kernel (profiles/r03_fetch_calibration.json, tools/fetch_calib.sh): x2 for wide coalesced
                # streams (the per-Gaussian kernels, the radix passes), x1 for the compositing kernels, whose reads are 64-byte splat
                # records gathered per list entry — the counter is exact for those (known / raw = 1.02)
                gather = k in ("render_forward_kernel", "render_backward_kernel")
                hb = measured[0][k] - (0.5 * measured[3].get(k, 0.0) if (gather and len(measured) > 3) else 0.0)
                ent["hbm_bytes"] = int(hb)
                ent["fetch_factor"] = 1.0 if gather else 2.0
                if "alg_bytes" in ent:
                    ent["traffic_over_alg"] = round(hb / ent["alg_bytes"], 3)
            if len(measured) > 2 and measured[2] and k in measured[2]:
                # vector-ALU issue: wave instructions x 2 cycles (SIMD-32: a wave64 instruction occupies the ALU for two cycles at
                # best, MI355X_MICROARCH.md) over SIMDs x kernel cycles at the 2.4 GHz peak clock.  A lower bound on how busy the ALU
                # is: DPP-modified and compare instructions cost twice that, transcendentals four times (profiles/r03_valu_rate_probe.txt)
                ent["valu_insts"] = int(measured[2][k])
                ent["valu_issue_frac"] = round(measured[2][k] * 2.0 / (1024.0 * ent["ms_per_step"] * 1e-3 * 2.4e9), 3)
        traffic, traffic_source = kernels[dom].get("hbm_bytes"), measured[1]
    pmc = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_pmc_{wname}.json")))
    if traffic is None and pmc:
        try:
            ent = json.load(open(pmc[-1]))
            rec = (ent.get(dom) or (dom == "render_backward_kernel" and (ent.get("render_backward_blk_kernel") or ent.get("render_backward_scan_kernel")))
                   or (dom == "render_forward_kernel" and ent.get("render_forward_rows_kernel")) or {})
            traffic = rec.get("hbm_bytes_per_launch")
            if traffic is not None:
                traffic_source = (f"{os.path.relpath(pmc[-1], ROOT)} (separate rocprofv3 --pmc passes of the same workload, committed; "
                                  f"FETCH_SIZE doubled per MI355X_MICROARCH.md) — not collected by this run"
                                  + (f" ({measured[1]})" if measured is not None else ""))
        except Exception:  # noqa: BLE001
            traffic = None
    roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_source,
                "alg_bytes_per_launch": per_kernel[dom], "avg_launch_ms": round(dom_ms, 5),
                "pipeline": {"alg_bytes_fwd_bwd": b_fwd + b_bwd, "kernel_ms_per_step": round(total_ms, 5),
                             "GBps": round((b_fwd + b_bwd) / (total_ms * 1e-3) / 1e9, 2),
                             "frac": round((b_fwd + b_bwd) / (total_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)},
                "how": "HIP events around every launch, separate pass of the same K steps"}
    return kernels, roofline


def pmc_rows(files, counter):
    """rocprofv3 counter_collection CSVs -> (kernel group, kernel name, bytes) per launch of this library's kernels.  FETCH_SIZE /
    WRITE_SIZE are in KiB; FETCH_SIZE is doubled (gfx950: wide reads are under-counted by 2x, MI355X_MICROARCH.md)."""
    import csv
    from das3r_amd.roofline import ALIASES, BINNING_KERNELS
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("das3r::", "").split("<")[0]
            if not (k in BINNING_KERNELS or k in ALIASES or k.startswith("render_") or k.startswith("preprocess_")):
                continue
            b = float(r["Counter_Value"]) * (1.0 if counter == "SQ_INSTS_VALU" else 1024.0 * (2.0 if counter == "FETCH_SIZE" else 1.0))
            yield ("binning" if k in BINNING_KERNELS else ALIASES.get(k, k)), k, b


def measure_traffic(workload, timeout_s=90):
    """HBM bytes per launch of every kernel of `workload`, MEASURED BY THIS RUN: two child passes of rocprofv3 --pmc (FETCH_SIZE,
    then WRITE_SIZE — separate passes, kernel trace only, as /opt/skills/guides/MI355X_MICROARCH.md prescribes) over three steps
    of the same workload.  FETCH_SIZE / WRITE_SIZE come in KiB; on gfx950 FETCH_SIZE under-counts wide reads by 2x (same guide),
    so bytes = 2 * fetch + write.  -> ({kernel group: bytes per step}, note) or (None, why not)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    from das3r_amd.roofline import ALIASES, BINNING_KERNELS
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, "rocprofv3 not found"
    if any(k.startswith("ROCP") for k in os.environ):
        return None, "this process is itself under a profiler"
    env = dict(os.environ, TMPDIR="/tmp", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    steps = 3
    tot, raw = {}, {}
    work = tempfile.mkdtemp(prefix="das3r_pmc_", dir="/tmp")
    try:
        valu, fetch2 = {}, {}
        for counter in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU"):
            d = os.path.join(work, counter)
            cmd = [exe, "--pmc", counter, "--kernel-trace", "-d", d, "-o", "pmc", "--output-format", "csv", "--", sys.executable,
                   os.path.abspath(__file__), "--pmc-child", "--workload", workload, "--steps", str(steps)]
            p = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                p.wait(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                os.killpg(p.pid, 9)   # (the group this call started: the profiler and its child)
                p.wait()
                return None, f"rocprofv3 --pmc {counter} pass exceeded {timeout_s} s"
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if not files:
                return None, f"rocprofv3 --pmc {counter} pass wrote no counter file (rc {p.returncode})"
            for key, raw_name, b in pmc_rows(files, counter):
                if counter == "SQ_INSTS_VALU":
                    valu[key] = valu.get(key, 0.0) + b
                    continue
                if counter == "FETCH_SIZE":
                    fetch2[key] = fetch2.get(key, 0.0) + b
                tot[key] = tot.get(key, 0.0) + b
                if key == "binning":
                    raw[raw_name] = raw.get(raw_name, 0.0) + b
    finally:
        shutil.rmtree(work, ignore_errors=True)
    # the child ran one initialisation step + `steps` steps, every one of them counted
    print(json.dumps({"bench_binning_hbm_bytes_per_step": {k: int(v / (steps + 1)) for k, v in raw.items()}}), file=sys.stderr, flush=True)
    return ({k: v / (steps + 1) for k, v in tot.items()},
            f"this run: rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE (separate passes, {steps + 1} steps), bytes = f*FETCH + WRITE with f = 2 for the "
            f"streaming kernels (gfx950 undercounts wide coalesced reads by 2x) and f = 1 for the compositing kernels' 64-byte record "
            f"gathers (calibrated: profiles/r03_fetch_calibration.json)",
            {k: v / (steps + 1) for k, v in valu.items()}, {k: v / (steps + 1) for k, v in fetch2.items()})


def pmc_child_main(args):
    """What the rocprofv3 --pmc passes of measure_traffic() run: a few steps of the workload, nothing else."""
    import torch
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    job = RasterJob(args.workload, dev)
    job.upload()
    for _ in range(args.steps + 1):
        job.step()
    torch.cuda.synchronize()


def cpu_baseline_of(sc_cpu, name, budget_s=30.0):
    """The oracle (plain C + OpenMP restatement of the reference algorithm) on the host cores, same scene; rank 0, N = 1 only.
    SURVEY.md 8d: one warm-up, then the median of 5 runs, forward and forward + backward — as many of the 5 as fit in `budget_s`
    (a 5 M-splat scene takes ~11 s per run: the sample then says how many were taken)."""
    import statistics
    from oracle import c_oracle
    o = c_oracle.RasterOracle(**sc_cpu.settings_kwargs())
    np_in = dict(shs=sc_cpu.shs.numpy(), scales=sc_cpu.scales.numpy(), rotations=sc_cpu.rotations.numpy())
    m3, op, dl = sc_cpu.means3D.numpy(), sc_cpu.opacities.numpy(), sc_cpu.dL_dpix.numpy()
    t_start = time.perf_counter()
    o.forward(m3, op, **np_in)   # warm-up (page faults of the oracle's buffers, OpenMP pool)
    o.backward(dl)
    fb, fw = [], []
    while len(fb) < 5 and (not fb or (time.perf_counter() - t_start) < budget_s):
        t1 = time.perf_counter()
        o.forward(m3, op, **np_in)
        t2 = time.perf_counter()
        o.backward(dl)
        t3 = time.perf_counter()
        fw.append(t2 - t1)
        fb.append(t3 - t1)
    o.free()
    med, medf = statistics.median(fb), statistics.median(fw)
    return {"value": round(sc_cpu.P / med / 1e6, 4), "unit": "Msplats/s", "cores": c_oracle.max_threads(), "kind": "port",
            "sample": f"whole {name} scene ({sc_cpu.P} splats, {sc_cpu.W}x{sc_cpu.H}), fwd+bwd, median of {len(fb)} runs after 1 warm-up, "
                      f"{med * 1e3:.1f} ms per fwd+bwd", "ms_per_step": round(med * 1e3, 2),
            "fwd_only": {"value": round(sc_cpu.P / medf / 1e6, 4), "unit": "Msplats/s", "ms_per_step": round(medf * 1e3, 2)}}


def train_step_timer(dev, fused, frames=20, W=512, H=208, depth="noise", fused_loss=None):
    """-> (step callable, splats): the DAS3R-shaped optimisation step (render + masked L1/SSIM loss + backward + both Adam steps,
    train_gui.py:542-589) on a synthetic sequence with one Gaussian per pixel of every frame."""
    import torch
    from types import SimpleNamespace
    from das3r_amd.model import OptimParams
    from das3r_amd.train import build_from_sequence, synthetic_sequence, train_step
    seq = synthetic_sequence(frames=frames, W=W, H=H, focal=600.0, n_splats=20000, seed=0, device=str(dev), depth=depth)
    model, cams = build_from_sequence(seq)
    opt = OptimParams(iterations=ITERS_PER_SCENE)
    model.training_setup(opt, fused=fused)
    pipe = SimpleNamespace(debug=False, compute_cov3D_python=False, convert_SHs_python=False)
    bg = torch.zeros(3, device=dev)
    it = [0]

    def step():
        it[0] += 1
        train_step(model, cams[it[0] % len(cams)], opt, it[0], pipe, bg, fused=fused, fused_loss=fused_loss)
    return step, int(model.get_xyz.shape[0])


def eval_forward_figure(dev):
    """Eval-mode (torch.no_grad) forward of a trained model — what /root/reference/render.py:72-86 and every held-out report run: the
    Sintel-shaped model of a consistent sequence, rendered from its 20 training poses with render_test (das3r_amd/offline.py).  The
    no-grad path examines the binning self-check inside every call (there is no backward to do it): that wait is in the figure."""
    from das3r_amd import offline
    from das3r_amd.train import build_from_sequence, consistent_sequence
    seq = consistent_sequence(seed=0, device=str(dev), **JOB_SHAPES["sintel"])
    model, cams = build_from_sequence(seq)
    loaded = offline.SplatModel(3)   # the state load_ply leaves: per-Gaussian conf_static column, active degree = maximum
    for n in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation"):
        setattr(loaded, n, getattr(model, n).detach())
    loaded._conf_static = model._conf_static.detach().reshape(-1, 1)[model.aggregated_mask]
    loaded.active_sh_degree = 3
    views = offline.sequence_cameras(seq, dev)[:20]
    out = offline.forward_throughput(loaded, views, repeats=5)
    fz = offline.forward_throughput(loaded, views, repeats=5, fused=True)
    return {"ms_per_view": round(out["ms_per_view"], 4), "views_per_s": round(out["views_per_s"], 1), "Msplats_per_s": round(out["splats"] * out["views_per_s"] / 1e6, 1),
            "splats": out["splats"], "image": [512, 208], "sh_degree": 3,
            "fused": {"ms_per_view": round(fz["ms_per_view"], 4), "views_per_s": round(fz["views_per_s"], 1), "Msplats_per_s": round(fz["splats"] * fz["views_per_s"] / 1e6, 1),
                      "what": "offline.render_view_fused: the pose pre-transform inside the rasterizer's kernels (das3r_raster_in.pre) instead of the reference's PyTorch glue"},
            "what": "torch.no_grad render_test of a loaded model (render.py:72-86), 5 passes over 20 views; includes the per-call wait for the binning self-check"}


def extras_main(main_workload):
    """Child process of the default N = 1 run: the other BASELINE shapes and the unfused train step, one JSON object on stdout."""
    import torch
    from das3r_amd.hostpin import pin_to_ccx
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    rk = Ranks(parse_args(["--gpus", "1"]))
    pin_to_ccx(0)
    out = {}
    us_step, _ = train_step_timer(dev, fused=False)
    out["train_step_unfused_ms"] = round(rk.timed(us_step, 30, 5) / 30 * 1e3, 4)
    del us_step
    # what an UNMODIFIED DAS3R checkout gets with `import das3r_amd.integrate; das3r_amd.integrate.patch()`: fused
    # pre-transform + FusedAdam behind the reference's own loop, its loss in torch ops, its camera gate a host-side `if`
    pa_step, _ = train_step_timer(dev, fused=True, fused_loss="ssim")   # (round 6: patch() also answers the loop's `ssim` call)
    out["train_step_patched_ms"] = round(rk.timed(pa_step, 50, 5) / 50 * 1e3, 4)
    del pa_step
    try:
        out["eval_forward"] = eval_forward_figure(dev)
    except Exception as ex:  # noqa: BLE001
        out["eval_forward"] = {"error": repr(ex)}
    torch.cuda.empty_cache()
    fs_step, fs_splats = train_step_timer(dev, fused=True)
    for _ in range(INIT_STEPS):
        fs_step()
    out["train_step_fused_ms"] = round(rk.timed(fs_step, 100, 10) / 100 * 1e3, 4)
    out["train_step_splats"] = fs_splats
    del fs_step
    torch.cuda.empty_cache()
    # the same step on depth maps with the spatial coherence of a real predictor's (a smooth relief + 1 % noise instead of independent
    # noise per pixel: a tile then sees a thin depth band, DESIGN.md section 4 ledger (ao)) — what a real sequence costs
    co_step, _ = train_step_timer(dev, fused=True, depth="smooth")
    for _ in range(INIT_STEPS):
        co_step()
    out["train_step_fused_smooth_depth_ms"] = round(rk.timed(co_step, 100, 10) / 100 * 1e3, 4)
    del co_step
    torch.cuda.empty_cache()
    # the same step at the DAVIS size of the reference's held-out runs: 45 training frames of 512x288 = 6.6 M Gaussians
    # (scripts/testing_psnr_davis.sh:35-59; the whole 4000-iteration job at that size: profiles/r03_farm_davis_shape.json)
    ds_step, ds_splats = train_step_timer(dev, fused=True, frames=45, W=512, H=288)
    for _ in range(10):
        ds_step()
    out["train_step_davis"] = {"fused_ms": round(rk.timed(ds_step, 50, 5) / 50 * 1e3, 4), "splats": ds_splats, "frames": 45, "image": [512, 288]}
    del ds_step
    for w, k in (("c2", 300), ("ds", 50), ("dsc", 50), ("c1", 300), ("c4d", 700), ("c4", 200)):
        if w == main_workload:
            continue
        torch.cuda.empty_cache()
        j = RasterJob(w, dev)
        j.upload()
        j.step()
        t = rk.timed(j.step, k, 30) / k
        kt, rf = kernel_table(j, min(k, 50)) if w != "c4d" else (None, None)
        ent = {"workload": WORKLOAD_DESC[w], "ms_per_step": round(t * 1e3, 4), "Msplats_per_s": round(j.sc_cpu.P / t / 1e6, 2),
               "steps": k, "num_rendered": j.num_rendered}
        if rf:
            ent["roofline"] = {kk: rf[kk] for kk in ("kernel", "achieved", "frac", "avg_launch_ms", "pipeline")}
            ent["kernel_ms"] = {kk: v["ms_per_step"] for kk, v in kt.items()}
        if w == "c4d":
            ent["growth"] = {"P_final": j.P, "events": len(j.events), "every": 100, "fraction": 0.05}
        out[w] = ent
        del j
    # ---- whole jobs, K sequences in flight on this GPU (das3r_amd.farm.run_jobs): scenes per hour MEASURED from
    # jobs — model initialisation (distCUDA2), 4000 fused iterations with the held-out pose passes, held-out report — on the
    # self-consistent synthetic sequences (train.consistent_sequence: images, depth maps and poses from one scene + a moving object),
    # whose held-out static-region PSNR is printed beside the rate (the stand-in for configs[2] / [4]: DAS3R_BENCH_JOBS=0 skips it)
    # Default: the Sintel shape with K = 1 and 2 (+ 25 s: the line must stay a matter of a minute); DAS3R_BENCH_JOBS=full: both shapes,
    # K = 1, 2, 3 (+ 2.5 min — what tools/refresh_profiles.sh records in profiles/rNN_bench_c4.json)
    # "also run the new backward at C4 and report it in the line" — the 2x2-region walk (render_bwd_rgn.hip) forced onto
    # the 1 M-splat 1080p workload, whose Gaussians cover 45 pixels each: not its shape, and never chosen for it
    try:
        from das3r_amd import _lib
        os.environ["DAS3R_RENDER_BWD"] = "fine128"
        _lib.reload_switches()
        j = RasterJob("c4", dev)
        j.upload()
        j.step()
        kt, _rf = kernel_table(j, 30)
        out["c4_region_backward_ms"] = {"ms": kt["render_backward_kernel"]["ms_per_step"],
                                        "what": "render_backward_regions_kernel<128> forced at C4 (DAS3R_RENDER_BWD=fine128); the default there is the block walk"}
        del j
    except Exception as ex:  # noqa: BLE001
        out["c4_region_backward_ms"] = {"error": repr(ex)}
    finally:
        os.environ.pop("DAS3R_RENDER_BWD", None)
        try:
            _lib.reload_switches()
        except Exception:  # noqa: BLE001
            pass
    torch.cuda.empty_cache()
    mode = os.environ.get("DAS3R_BENCH_JOBS", "1")
    if mode != "0":
        try:
            if mode == "full":
                out["jobs_in_flight"] = jobs_in_flight(dev)
            else:   # (+ one DAVIS-shaped job, K = 1: its held-out PSNR belongs in the default line — VERDICT r5 item 7; + 20 s)
                out["jobs_in_flight"] = jobs_in_flight(dev, shapes=("sintel",), ks=(1, 2))
                out["jobs_in_flight"].update(jobs_in_flight(dev, shapes=("davis",), ks=(1,), warm=False))
        except Exception as ex:  # noqa: BLE001
            out["jobs_in_flight"] = {"error": repr(ex)}
    print("EXTRAS " + json.dumps(out), flush=True)


JOB_SHAPES = {"sintel": dict(frames=22, W=512, H=208, focal=600.0, n_splats=20000),     # 20 training frames: 2.13 M Gaussians
              "davis": dict(frames=50, W=512, H=288, focal=614.4, n_splats=60000)}      # 45 training frames: 6.64 M Gaussians


def jobs_in_flight(dev, shapes=("sintel", "davis"), ks=(1, 2, 3), iterations=ITERS_PER_SCENE, warm=True):
    """{shape: {K: {wall_s, scenes_per_hour, heldout_psnr, peak_hbm_gb}}} (tools/jobs_per_gpu.py is the same measurement as a tool)."""
    import torch
    from das3r_amd.farm import run_jobs, run_sequence_job
    from das3r_amd.train import consistent_sequence
    if warm:
        run_sequence_job(0, 60, dev, fused=True, seq=consistent_sequence(frames=12, W=128, H=80, focal=150.0, n_splats=3000, seed=99, device=str(dev)))
    res = {}
    for name in shapes:
        seqs = [consistent_sequence(seed=s, device=str(dev), **JOB_SHAPES[name]) for s in range(max(ks))]
        rows = {}
        for K in ks:
            torch.cuda.empty_cache()
            torch.cuda.reset_peak_memory_stats()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            recs = run_jobs(range(K), lambda s: run_sequence_job(s, iterations, dev, fused=True, seq=seqs[s]), K, dev)
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
            rows[str(K)] = {"wall_s": round(wall, 2), "scenes_per_hour": round(K * 3600.0 / wall, 1), "ok": int(all(r["ok"] for r in recs)),
                            "heldout_psnr": [round(r["psnr"], 2) if r["ok"] else None for r in recs],   # (never a NaN in the JSON line)
                            "peak_hbm_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}
        res[name] = rows
        del seqs
    return res


def ranked_jobs_in_flight(rk, K, job, what):
    """N > 1: every rank runs K whole jobs at once (farm.run_jobs: K host threads, each on its own stream and core),
    all ranks starting behind one barrier; the rate is N x K jobs over the SLOWEST rank's wall time (max over ranks, like `value`).
    job(s) runs sequence s of this rank.  -> dict on every rank (the gathers are collectives)."""
    from das3r_amd.farm import run_jobs
    rk.barrier()
    t0 = time.perf_counter()
    failed = None
    try:
        recs = run_jobs(range(K), job, K, rk.dev if rk.dev.type == "cuda" else None)
    except Exception as ex:  # noqa: BLE001 - one rank's failure is reported, not hung on
        recs, failed = [], repr(ex)
    rk.sync()
    local = time.perf_counter() - t0
    walls = rk.gather_all(local)
    oks = rk.gather_all(float(sum(1 for r in recs if (r.get("ok", 1) if isinstance(r, dict) else 1))))
    wall = max(walls)
    out = {"what": what, "jobs_per_gpu": K, "wall_s": round(wall, 3), "per_rank_wall_s": [round(w, 3) for w in walls], "jobs_ok": int(sum(oks)),
           "scenes_per_hour": round(sum(oks) * 3600.0 / wall, 1), "per_rank_scenes_per_hour": [round(o * 3600.0 / w, 1) for o, w in zip(oks, walls)]}
    psnr = [round(r["psnr"], 2) for r in recs if isinstance(r, dict) and r.get("ok") and "psnr" in r]
    if psnr:
        out["heldout_psnr_rank0"] = psnr
    if failed:
        out["error_this_rank"] = failed
    return out


def run_extras_child(main_workload):
    """The extras run in a process of their own, with the profiler's hooks stripped from its environment: a rocprofv3
    --kernel-trace --stats of `python bench.py` then holds the launches of the benchmarked workload only, and its per-kernel
    averages can be compared with this line's `roofline` (the kernels of the other shapes carry the same names)."""
    import subprocess
    env = {k: v for k, v in os.environ.items()
           if not (k.startswith("ROCP") or k.startswith("ROCPROF") or k in ("LD_PRELOAD", "HSA_TOOLS_LIB", "HSA_TOOLS_REPORT_LOAD_FAILURE"))}
    try:
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--extras-child", "--workload", main_workload], capture_output=True,
                           text=True, timeout=900, env=env, cwd=ROOT)
        for line in p.stdout.splitlines():
            if line.startswith("EXTRAS "):
                return json.loads(line[7:])
        return {"error": (p.stderr or p.stdout)[-500:]}
    except Exception as ex:  # noqa: BLE001 - the extras never take the bench line down
        return {"error": repr(ex)}


def main():
    args = parse_args()
    if args.extras_child:
        return extras_main(args.workload)
    if args.pmc_child:
        return pmc_child_main(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        respawn_under_torchrun(args)   # does not return
    import torch
    rk = Ranks(args)
    out = None
    if args.stub:   # the launch contract without a GPU: N gloo ranks, stand-in step, the same timing protocol
        x = torch.zeros(1024)
        fail_rank = int(os.environ.get("DAS3R_BENCH_STUB_FAIL_RANK", "-1"))   # tests: this rank's job dies half way

        def step():
            x.add_(1.0)
            if rk.rank == fail_rank and float(x[0]) > args.warmup + args.steps // 2:
                raise RuntimeError("stub job failure (DAS3R_BENCH_STUB_FAIL_RANK)")
        elapsed = rk.timed(step, args.steps, args.warmup)
        ranks_ok = [int(round(v)) for v in rk.gather_all(0.0 if rk.failed else 1.0)]
        per_rank_ms = rk.gather_all(rk.last_local_s / args.steps * 1e3)
        jobs = None
        if rk.world > 1:   # the K-jobs-in-flight leg of an N > 1 run, with stand-in jobs (rank r's take 0.05 (r + 1) s)
            jobs = ranked_jobs_in_flight(rk, 2, lambda s_: (time.sleep(0.05 * (rk.rank + 1)), {"ok": 1})[1], "stub jobs")
        if rk.rank == 0:
            out = {"metric": "stub steps/s (launch-logic dry run, no GPU work)", "value": round(sum(ranks_ok) * args.steps / elapsed, 3),
                   "unit": "steps/s", "n_gpus": rk.world, "steps": args.steps, "warmup": args.warmup,
                   "ms_per_step": round(elapsed / args.steps * 1e3, 6), "higher_is_better": True, "scaling": "weak",
                   "vs_baseline": None, "dtype": "f32", "data": "synthetic", "config": {"workload": "stub"},
                   "ranks_ok": ranks_ok, "per_rank_ms": per_rank_ms, "jobs_in_flight": jobs}
            print(json.dumps(out), flush=True)
        rk.close()
        return

    from das3r_amd.hostpin import pin_to_ccx, unpin
    job = RasterJob(args.workload, rk.dev, seed_offset=1000 * rk.rank)
    pinned = pin_to_ccx(rk.local_rank)            # before the first HIP call: the runtime's threads inherit the mask
    job.upload()
    # initialisation (not bench steps, reported as config.init_steps): the first call loads the code objects, sizes torch's caching
    # allocator and seeds the capacity cache of the sync-free forward; the rest bring the clocks up — with W = 5 alone the first
    # timed steps of a 20-step run still ran 3 % below the steady state (1.135 vs 1.099 ms per step at 300 / 30)
    for _ in range(INIT_STEPS if args.workload != "c4d" else 1):
        job.step()
    rk.barrier()
    elapsed = rk.timed(job.step, args.steps, args.warmup)
    ms_per_step = elapsed / args.steps * 1e3
    timed_kernels = kernel_times(job, args.steps) if rk.rank == 0 else None   # right behind the timed region (same clocks)
    per_rank_ms = rk.gather_all(rk.last_local_s / args.steps * 1e3) if rk.world > 1 else None   # a straggler shows here
    ranks_ok = [int(round(v)) for v in rk.gather_all(0.0 if rk.failed else 1.0)]
    # the forward render alone, same protocol (BASELINE.json metric: "render fwd/bwd Msplats/s")
    fwd_elapsed = rk.timed(job.forward_only, args.steps, min(args.warmup, 5)) if args.workload != "c4d" else None
    measured = measure_traffic(args.workload) if (rk.rank == 0 and rk.world == 1 and not args.no_pmc) else None
    P_report = job.sc_cpu.P                        # (c4d: P grows; the rate is quoted on the initial P, the growth is in config)
    msplats = sum(ranks_ok) * P_report / (elapsed / args.steps) / 1e6   # (a rank whose job failed processed nothing)

    kernels, roofline = (None, None)
    if rk.rank == 0:
        kernels, roofline = kernel_table(job, args.steps, measured, timed_kernels)

    # ---- train-step ms (fused §8f path) -> scenes/hour of the farm.  N = 1: in the extras child (its rasterizer launches carry the
    # same kernel names as the benchmarked workload's: kept out of this process, a rocprofv3 --stats of this command averages the
    # workload's launches only).  N > 1 (or --no-extras): on every rank, here, under the same timing protocol.
    extras, cpu_baseline, train = None, None, None
    if rk.rank == 0 and rk.world == 1 and not args.no_extras:
        extras = run_extras_child(args.workload)
        if extras and "train_step_fused_ms" in extras:
            train = {"fused": extras.pop("train_step_fused_ms"), "splats": extras.pop("train_step_splats", None), "frames": 20,
                     "image": [512, 208], "iters": 100}
            if "train_step_davis" in extras:
                train["davis_shape"] = extras.pop("train_step_davis")
            if "train_step_fused_smooth_depth_ms" in extras:
                train["fused_smooth_depth"] = extras.pop("train_step_fused_smooth_depth_ms")
            if "train_step_unfused_ms" in extras:
                train["unfused"] = extras.pop("train_step_unfused_ms")
            if "train_step_patched_ms" in extras:
                train["patched"] = extras.pop("train_step_patched_ms")   # das3r_amd.integrate.patch() on an unmodified train_gui.py
    if train is None:
        ts_step, ts_splats = train_step_timer(rk.dev, fused=True)
        ts_iters = max(20, min(100, args.steps))
        ts_ms = rk.timed(ts_step, ts_iters, 10) / ts_iters * 1e3
        train = {"fused": round(ts_ms, 4), "splats": ts_splats, "frames": 20, "image": [512, 208], "iters": ts_iters}
        del ts_step
    train["what"] = "render + masked L1/SSIM loss + backward + both Adam steps (train_gui.py:542-589), fused kernels"
    scenes_per_hour = rk.world * 3600e3 / (ITERS_PER_SCENE * train["fused"])
    sph_def = f"N x 3600 s / ({ITERS_PER_SCENE} it x train_step_ms.fused): derived from one step (no whole jobs were run)"
    jobs = extras.pop("jobs_in_flight", None) if extras else None
    if rk.world > 1 and os.environ.get("DAS3R_BENCH_JOBS", "1") != "0":
        # N > 1: whole Sintel-shaped jobs, two in flight per rank (the farm's default with --fused), every rank at once
        from das3r_amd.farm import run_sequence_job
        from das3r_amd.train import consistent_sequence
        run_sequence_job(0, 60, rk.dev, fused=True, seq=consistent_sequence(frames=12, W=128, H=80, focal=150.0, n_splats=3000, seed=99, device=str(rk.dev)))
        seqs = [consistent_sequence(seed=1000 * rk.rank + s_, device=str(rk.dev), **JOB_SHAPES["sintel"]) for s_ in range(2)]
        ranked = ranked_jobs_in_flight(rk, 2, lambda s_: run_sequence_job(s_, ITERS_PER_SCENE, rk.dev, fused=True, seq=seqs[s_]),
                                       "whole Sintel-shaped jobs (22 x 512x208, 4000 fused iterations, held-out report), 2 in flight per GPU, all ranks at once")
        del seqs
        if ranked["jobs_ok"] == 2 * rk.world:
            scenes_per_hour = ranked["scenes_per_hour"]
            sph_def = (f"measured: {2 * rk.world} whole Sintel-shaped jobs, 2 in flight per GPU on {rk.world} GPUs at once, over the slowest rank's wall time "
                       f"(derived from one step alone: {rk.world * 3600e3 / (ITERS_PER_SCENE * train['fused']):.1f})")
        jobs = {"ranked": ranked}
    if jobs and "sintel" in jobs:
        jobs["sintel"] = {k: v for k, v in jobs["sintel"].items() if v.get("ok")} or None   # (a job that failed finished early: not a rate)
    if jobs and jobs.get("sintel") and "1" in jobs["sintel"]:
        # measured from whole jobs (initialisation, 4000 iterations with the held-out passes, report), the best number of sequences in
        # flight per GPU; the figure derived from one step stays beside it
        best = max(jobs["sintel"], key=lambda k: jobs["sintel"][k]["scenes_per_hour"])
        train["heldout_psnr_4000_iters"] = {"sintel_shape": jobs["sintel"]["1"]["heldout_psnr"][0],
                                            "davis_shape": jobs.get("davis", {}).get("1", {}).get("heldout_psnr", [None])[0],
                                            "what": "static-region PSNR of the held-out views after a whole job on a self-consistent synthetic "
                                                    "sequence (train.consistent_sequence; protocol train_test_psnr.py:241-302)"}
        derived = scenes_per_hour
        scenes_per_hour = rk.world * jobs["sintel"][best]["scenes_per_hour"]
        sph_def = (f"N x measured whole Sintel-shaped jobs per hour with {best} sequences in flight per GPU (farm.run_jobs; K = 1: "
                   f"{jobs['sintel']['1']['scenes_per_hour']}); derived from one step alone: {derived:.1f}")
    unpin(pinned)   # the CPU baseline below uses every host core
    if rk.rank == 0 and rk.world == 1 and not args.no_cpu_baseline:
        cpu_baseline = cpu_baseline_of(job.sc_cpu, args.workload if args.workload != "c4d" else "c4")

    if rk.rank == 0:
        sc = job.sc
        out = {"metric": "render fwd+bwd Msplats/s at 1080p" if sc.W == 1920 else f"render fwd+bwd Msplats/s at {sc.W}x{sc.H}",
               "value": round(msplats, 3), "unit": "Msplats/s", "n_gpus": rk.world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f32", "data": "synthetic",
               "eval_forward": (extras.pop("eval_forward", None) if extras else None),
               "fwd_only": (None if fwd_elapsed is None else
                            {"value": round(rk.world * P_report / (fwd_elapsed / args.steps) / 1e6, 3), "unit": "Msplats/s",
                             "ms_per_step": round(fwd_elapsed / args.steps * 1e3, 4), "what": "forward render alone (training-mode call, no backward), same K / barrier protocol"}),
               "per_rank_ms": per_rank_ms, "ranks_ok": ranks_ok,
               "config": {"workload": WORKLOAD_DESC[args.workload], "name": args.workload, "splats_per_gpu": P_report,
                          "splats_per_gpu_final": job.P, "init_steps": INIT_STEPS if args.workload != "c4d" else 1, "image": [sc.W, sc.H], "sh_degree": sc.sh_degree, "num_rendered": job.num_rendered,
                          "api": "GaussianRasterizer.forward + autograd backward (drop-in surface)",
                          "parallelism": f"{rk.world} independent scenes, one per GPU",
                          "host_cpus": (f"pinned to CPUs {pinned[1][0]}-{pinned[1][-1]} (one core complex per rank)" if pinned
                                        else "not pinned")},
               "c4_region_backward_ms": (extras.pop("c4_region_backward_ms", None) if extras else None),
               "train_step_ms": train, "scenes_per_hour": round(scenes_per_hour, 2),
               "scenes_per_hour_def": sph_def, "jobs_in_flight": jobs,
               "roofline": roofline, "cpu_baseline": cpu_baseline, "kernels": kernels, "extras": extras}
        if extras and not args.full_line:
            # the line stays short enough for any log tail: per-workload detail goes to stderr (and to gpurun_out/ when it exists)
            detail = json.dumps({"bench_extras": extras})
            print(detail, file=sys.stderr, flush=True)
            if os.path.isdir(os.path.join(ROOT, "gpurun_out")):
                with open(os.path.join(ROOT, "gpurun_out", "bench_extras.json"), "w") as f:
                    f.write(detail + "\n")
            out["extras"] = {w: (e.get("ms_per_step") if isinstance(e, dict) else e) for w, e in extras.items()}
            out["extras_unit"] = "ms_per_step (detail: stderr bench_extras)"
        print(json.dumps(out), flush=True)
    rk.close()


if __name__ == "__main__":
    main()
