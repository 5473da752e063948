"""GPU tests of the Python callers the repo ships in place of the reference's: das3r_render(),
the train step and the held-out PSNR report, end to end on the HIP rasterizer + distCUDA2."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_das3r_render_matches_direct_rasterizer_call():
    """das3r_render's pre-transform (pose -> camera frame, quaternion product, opacity * conf) against doing the same
    algebra by hand and calling the rasterizer directly."""
    from types import SimpleNamespace
    from das3r_amd import GaussianRasterizationSettings, GaussianRasterizer
    from das3r_amd.camera import camera_from_tensor, quat_multiply
    from das3r_amd.render import das3r_render
    from das3r_amd.train import build_from_sequence, synthetic_sequence
    import math
    seq = synthetic_sequence(frames=3, W=96, H=64, focal=90.0, n_splats=2000, seed=1)
    model, cams = build_from_sequence(seq)
    pipe = SimpleNamespace(debug=False, compute_cov3D_python=False, convert_SHs_python=False)
    bg = torch.tensor([0.1, 0.2, 0.3], device="cuda")
    cam = cams[1]
    pose = model.get_RT(cam.uid)
    pkg = das3r_render(cam, model, pipe, bg, camera_pose=pose)
    assert set(pkg) == {"render", "viewspace_points", "visibility_filter", "radii"}
    w2c = camera_from_tensor(pose)
    xyz = model._xyz
    means3D = (w2c @ torch.cat([xyz, torch.ones_like(xyz[:, :1])], 1).T).T[:, :3]
    rot = quat_multiply(pose[:4], model._rotation)
    opac = torch.sigmoid(model._opacity) * model._conf_static.reshape(-1, 1)[model.aggregated_mask]
    eye = torch.eye(4, device="cuda")
    rs = GaussianRasterizationSettings(image_height=cam.image_height, image_width=cam.image_width, tanfovx=math.tan(cam.FoVx * 0.5),
                                       tanfovy=math.tan(cam.FoVy * 0.5), bg=bg, scale_modifier=1.0, viewmatrix=eye,
                                       projmatrix=eye @ cam.projection_matrix, sh_degree=0, campos=torch.zeros(3, device="cuda"),
                                       prefiltered=False, debug=False)
    ref, radii = GaussianRasterizer(rs)(means3D=means3D, means2D=torch.zeros_like(means3D), opacities=opac, shs=model.get_features,
                                        scales=model.get_scaling, rotations=rot)
    assert torch.equal(pkg["render"], ref) and torch.equal(pkg["radii"], radii)
    assert torch.equal(pkg["visibility_filter"], radii > 0)
    # the dummy screen-space tensor receives the 2D mean gradient, pose and conf_static get gradients through the pre-transform
    pkg["render"].sum().backward()
    assert pkg["viewspace_points"].grad is not None and pkg["viewspace_points"].grad.shape == (xyz.shape[0], 3)
    assert model.Q.grad is not None and model.T.grad is not None and model._conf_static.grad is not None
    assert float(pkg["viewspace_points"].grad[:, 2].abs().max()) == 0.0


def test_train_step_improves_psnr():
    from das3r_amd.model import OptimParams
    from das3r_amd.train import build_from_sequence, psnr_report, synthetic_sequence, train
    seq = synthetic_sequence(frames=4, W=96, H=64, focal=90.0, n_splats=3000, seed=2)
    model, cams = build_from_sequence(seq)
    n = model.get_xyz.shape[0]
    assert n == 4 * 96 * 64                       # every confident pixel of every frame is one Gaussian
    opt = OptimParams(iterations=60)
    model.training_setup(opt)
    before = psnr_report(model, cams)
    stats = train(model, cams, opt, 60, seed=0)
    after = psnr_report(model, cams)
    assert stats["iters_per_s"] > 0 and torch.isfinite(torch.tensor(stats["loss"]))
    assert after["psnr"] > before["psnr"] + 0.5, (before, after)
    # learning-rate schedule reached the optimizer (xyz group decays, conf_static follows its own schedule)
    lrs = {g["name"]: g["lr"] for g in model.optimizer.param_groups}
    assert lrs["xyz"] < opt.position_lr_init and 3e-4 < lrs["conf_static"] < 3e-3


@pytest.mark.parametrize("fused", [False, True])
def test_held_out_pose_pass_changes_nothing(fused):
    """train_test_psnr.py's pass over the held-out views (train.test_pose_pass): renders and back-propagates every test view and
    changes no parameter and no optimizer state — with the PyTorch glue and with the fused kernels (the form the farm runs:
    round 3, it used to take the unfused path and more than half of a DAVIS-shaped job)."""
    import random
    from types import SimpleNamespace
    from das3r_amd.model import OptimParams
    from das3r_amd.train import build_from_sequence, synthetic_sequence, test_pose_pass, train
    seq = synthetic_sequence(frames=12, W=64, H=48, focal=70.0, n_splats=2500, seed=7)
    model, cams, test_cams = build_from_sequence(seq, heldout=True)
    assert len(test_cams) >= 1 and model.enable_test
    opt = OptimParams(iterations=20)
    model.training_setup(opt, fused=fused)
    train(model, cams, opt, 5, seed=1, fused=fused)   # a few steps so that the optimizer has state
    def params():   # (SplatModel is not an nn.Module: its parameters are the optimizers' — Gaussians, conf_static, poses, FoV)
        out = {}
        for oi, o in enumerate((model.optimizer, model.optimizer_cam)):
            for g in o.param_groups:
                for k, p in enumerate(g["params"]):
                    out[f"{oi}:{g.get('name', '?')}:{k}"] = p
        return out
    before = {n: p.detach().clone() for n, p in params().items()}
    dev = model.get_xyz.device
    masks = {c.uid: (torch.rand(48, 64, device=dev) > 0.7) for c in test_cams}
    pipe = SimpleNamespace(debug=False, compute_cov3D_python=False, convert_SHs_python=False)
    for gt_masks in (None, masks):
        test_pose_pass(model, test_cams, gt_masks, opt, pipe, torch.zeros(3, device=dev), random.Random(0), fused=fused)
    for n, p in params().items():
        assert torch.equal(p.detach(), before[n]), n
        assert p.grad is None or float(p.grad.abs().max()) == 0.0, n


def test_ply_written_from_device_tensors_is_the_same_file(tmp_path):
    """save_gaussians_ply interleaves its table on the device when the tensors live there (one transfer instead of a strided host
    concatenate of 1.7 GB on a DAVIS-shaped job): the file is byte for byte the one written from host copies of the tensors."""
    from das3r_amd import io_formats as io
    g = torch.Generator().manual_seed(3)
    P = 5000
    t = dict(xyz=torch.randn(P, 3, generator=g), f_dc=torch.randn(P, 1, 3, generator=g), f_rest=torch.randn(P, 15, 3, generator=g),
             op=torch.randn(P, 1, generator=g) * 3, sc=torch.randn(P, 3, generator=g), rot=torch.randn(P, 4, generator=g),
             conf=torch.rand(P, 1, generator=g))
    dev = {k: v.cuda() for k, v in t.items()}
    for name, src in (("host.ply", t), ("device.ply", dev)):
        io.save_gaussians_ply(str(tmp_path / name), src["xyz"], src["f_dc"], src["f_rest"], src["op"], src["sc"], src["rot"], src["conf"])
    assert (tmp_path / "host.ply").read_bytes() == (tmp_path / "device.ply").read_bytes()


def test_farm_job_on_a_sequence_directory(tmp_path):
    """§8(f)-4 end to end: a preprocessed sequence directory on disk -> load_sequence -> per-pixel Gaussian model -> a few
    optimisation steps (fused kernels) -> the reference's output files, read back."""
    import numpy as np
    from PIL import Image
    from das3r_amd import io_formats as io
    from das3r_amd.farm import run_sequence_job, sequence_cost
    from das3r_amd.train import synthetic_sequence
    seq = synthetic_sequence(frames=4, W=64, H=48, focal=70.0, n_splats=2500, seed=5)
    d = tmp_path / "data" / "scene_a"
    io.write_sequence_dir(seq, str(d))
    F = seq["images"].shape[0]
    c2w = seq["cam2world"].cpu().numpy().astype(np.float64)
    for i in range(F):   # the TUM line of frame i reproduces its camera-to-world matrix
        _, xyz, quat = io.read_tum_trajectory(d / "pred_traj.txt")
        assert np.allclose(io.tumpose_to_c2w(np.concatenate([xyz[i], quat[i]])), c2w[i], atol=1e-6)
    assert sequence_cost(str(d)) == F * seq["W"] * seq["H"]
    out = tmp_path / "out" / "scene_a"
    rec = run_sequence_job(0, 12, torch.device("cuda:0"), seq_dir=str(d), out_dir=str(out), fused=True)
    # Gaussians come from the TRAINING frames only; a 4-frame sequence has no (idx + 5) % 10 == 0 view: its last frame is held out
    assert rec["ok"] == 1 and rec["n_splats"] == (F - 1) * seq["W"] * seq["H"] and np.isfinite(rec["psnr"])
    ply = io.load_gaussians_ply(out / "point_cloud" / "iteration_12" / "point_cloud.ply")
    assert ply["xyz"].shape == (rec["n_splats"], 3) and ply["features_rest"].shape == (rec["n_splats"], 15, 3)
    assert np.isfinite(ply["opacity"]).all() and np.isfinite(ply["scaling"]).all()
    poses = np.load(out / "pose" / "pose_12.npy")
    assert poses.shape == (F - 1, 4, 4) and np.allclose(poses[:, 3], [0, 0, 0, 1])   # the optimised TRAINING poses (train_gui.py:467-480)
    log = (out / "test_log.txt").read_text()
    assert log.startswith("[ITER 12] Evaluating test: L1 ") and " PSNR " in log


def test_farm_job_direct_and_autograd_iterations_agree_on_a_sequence_directory(tmp_path, monkeypatch):
    """Round 4: the whole job — a sequence that came through the on-disk formats, 60 fused iterations with the held-out passes, the
    report — with the direct iteration (das3r_amd/fast_step.py) and with the autograd form of the same kernels (DAS3R_FAST_STEP=0):
    same held-out PSNR and L1.  (The first direct iteration read the ground-truth image by plain pointer; from disk it was a
    [3, H, W] view of H x W x 3 memory, and a whole job trained towards a scrambled target with every unit test green.)"""
    from das3r_amd import io_formats as io
    from das3r_amd.farm import run_sequence_job
    from das3r_amd.train import synthetic_sequence
    seq = synthetic_sequence(frames=12, W=64, H=48, focal=70.0, n_splats=2500, seed=8)
    d = tmp_path / "data" / "scene_b"
    io.write_sequence_dir(seq, str(d))
    recs = []
    for fast in ("1", "0"):
        monkeypatch.setenv("DAS3R_FAST_STEP", fast)
        recs.append(run_sequence_job(0, 60, torch.device("cuda:0"), seq_dir=str(d), out_dir=str(tmp_path / ("out" + fast)), fused=True))
    a, b = recs
    assert a["ok"] == b["ok"] == 1
    assert abs(a["psnr"] - b["psnr"]) <= 1e-3 and abs(a["l1"] - b["l1"]) <= 1e-5 * abs(b["l1"]) + 1e-7, (a, b)


def test_consistent_sequence_is_self_consistent():
    """train.consistent_sequence (round 5): the depth map of a frame is the expected depth of the SAME cloud that rendered its image,
    so unprojecting every pixel at its depth with the frame's pose and re-projecting it into ANOTHER frame lands on that frame's
    surface: the depth the other frame's map holds at the landing pixel equals the point's depth there (static pixels, away from
    depth edges) — the property the round-1-4 stand-ins lacked (their depth maps were unrelated to the images)."""
    from das3r_amd.model import depth_to_points
    from das3r_amd.train import consistent_sequence
    seq = consistent_sequence(frames=6, W=256, H=104, focal=300.0, n_splats=12000, seed=3)
    F, H, W = seq["depths"].shape
    assert seq["dyna_avg"].sum() > 0 and len(seq["gt_dynamic_masks"]) == F and seq["gt_dynamic_masks"][2].any()
    pts = depth_to_points(seq["K"], seq["cam2world"], seq["depths"])           # [F, H, W, 3] world points
    a, b = 1, 4
    w2c = torch.linalg.inv(seq["cam2world"][b])
    p = pts[a].reshape(-1, 3) @ w2c[:3, :3].T + w2c[:3, 3]
    f = seq["focal"]
    u, v = (f * p[:, 0] / p[:, 2] + W / 2).round().long(), (f * p[:, 1] / p[:, 2] + H / 2).round().long()
    static = ~torch.from_numpy(seq["gt_dynamic_masks"][a]).reshape(-1).to(p.device)
    ok = static & (u >= 0) & (u < W) & (v >= 0) & (v < H)
    db = seq["depths"][b][v[ok], u[ok]]
    sb = ~torch.from_numpy(seq["gt_dynamic_masks"][b]).to(p.device)[v[ok], u[ok]]
    rel = ((db - p[ok, 2]).abs() / p[ok, 2])[sb]
    assert rel.numel() > 0.5 * H * W
    # (a depth map holds the blend of the CENTRE depths of the Gaussians over a pixel, not a ray-surface intersection: on the relief's
    #  slopes it is off by slope x a Gaussian's radius, ~1 % of the depth; the unrelated maps of depth="noise" / "smooth" are off by tens of %)
    print("reprojected depth, relative error: median", float(rel.median()), "below 3 %:", float((rel < 0.03).float().mean()), "below 10 %:", float((rel < 0.1).float().mean()))
    assert float(rel.median()) < 0.025 and float((rel < 0.1).float().mean()) > 0.9, (float(rel.median()), float((rel < 0.1).float().mean()))
    # ... and the colours agree: frame a's pixel colour is what frame b shows at the landing pixel (a textured static surface)
    ca = seq["images"][a].reshape(3, -1)[:, ok][:, sb]
    cb = seq["images"][b][:, v[ok], u[ok]][:, sb]
    assert float((ca - cb).abs().mean()) < 0.05


def test_two_jobs_in_flight_reproduce_their_solo_results():
    """VERDICT r4 item 3 (farm.run_jobs, `--jobs-per-gpu 2`): two independent sequences optimised at the same time on one GPU — two
    host threads, each with its own stream, model and library state — end EXACTLY where they end alone: every parameter tensor, the
    held-out PSNR and L1 bit for bit.  (Round 5: the direct iteration has no run-to-run freedom left — the 28 pose sums, which met in
    float atomics, are added in a fixed order: pretransform.hip — so "reproduce" can mean equality; a solo job run twice is the control.)
    Replaces the serial loop of /root/reference/scripts/testing_psnr_davis.sh:35-59."""
    from das3r_amd.farm import run_jobs, run_sequence_job
    from das3r_amd.train import consistent_sequence
    dev = torch.device("cuda:0")
    seqs = [consistent_sequence(frames=12, W=128, H=80, focal=150.0, n_splats=3000, seed=20 + s) for s in range(2)]
    iters = 150
    solo_keep, again_keep, duo_keep = {}, {}, {}
    solo = [run_sequence_job(s, iters, dev, fused=True, seq=seqs[s], keep=solo_keep) for s in range(2)]
    again = run_sequence_job(0, iters, dev, fused=True, seq=seqs[0], keep=again_keep)
    duo = run_jobs(range(2), lambda s: run_sequence_job(s, iters, dev, fused=True, seq=seqs[s], keep=duo_keep), 2, dev)
    torch.cuda.synchronize()
    names = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation", "_conf_static", "Q", "T")
    assert again["psnr"] == solo[0]["psnr"] and all(torch.equal(getattr(again_keep[0][0], n), getattr(solo_keep[0][0], n)) for n in names), \
        "a job run twice alone must reproduce itself bit for bit"
    for s in range(2):
        a, b = solo[s], duo[s]
        assert a["ok"] == b["ok"] == 1 and a["scene_id"] == b["scene_id"] == s and a["n_splats"] == b["n_splats"]
        assert a["psnr"] == b["psnr"] and a["l1"] == b["l1"], (a, b)
        ma, mb = solo_keep[s][0], duo_keep[s][0]
        for n in names:
            assert torch.equal(getattr(ma, n), getattr(mb, n)), (s, n, float((getattr(ma, n) - getattr(mb, n)).abs().max()))
    assert abs(solo[0]["psnr"] - solo[1]["psnr"]) > 1e-2, "the two sequences are meant to be different jobs"


def test_run_jobs_keeps_order_and_raises():
    from das3r_amd.farm import run_jobs
    dev = torch.device("cuda:0")
    seen = []

    def job(i):
        x = torch.full((1024,), float(i), device=dev)
        seen.append(torch.cuda.current_stream().cuda_stream)
        return int(x.sum().item()) // 1024

    assert run_jobs(range(7), job, 3, dev) == list(range(7))
    assert len(set(seen)) >= 2 and torch.cuda.default_stream().cuda_stream not in seen, "every worker thread has a stream of its own"

    def bad(i):
        if i == 2:
            raise ValueError("job 2")
        return i

    with pytest.raises(ValueError, match="job 2"):
        run_jobs(range(4), bad, 2, dev)


SINTEL_SHAPE = dict(frames=22, W=512, H=208, focal=600.0, n_splats=20000)
PSNR_BAR_SINTEL = 43.0     # dB, held-out static region after 4000 iterations; measured 43.91 (seed 0; 44.8 / 45.0 for seeds 1 / 2), the same in every run
PSNR_START_SINTEL = 36.0   # dB, the same report after 20 iterations stays BELOW this: the optimisation is what gets a job over the bar


def test_consistent_sintel_shaped_job_reaches_the_psnr_bar(monkeypatch):
    """VERDICT r4 item 4 — an ABSOLUTE bar for a whole job, the stand-in for BASELINE configs[2] (Sintel market_2, train 4000 iterations,
    published 29.03 dB on the real sequence): a Sintel-shaped self-consistent synthetic sequence (22 frames of 512 x 208, two held out,
    2.13 M Gaussians; images, depth maps and poses from ONE scene, a moving object under dyna_avg / ground-truth masks:
    train.consistent_sequence) through the job the farm runs — 4000 fused iterations with the held-out pose passes, the report of
    /root/reference/train_test_psnr.py:241-302 over the split of scene/dataset_readers.py:342-347 — must END above PSNR_BAR_SINTEL on
    the held-out static region, having STARTED below PSNR_START_SINTEL (measured: 26.1 dB after 20 iterations), with the direct iteration
    (fast_step.py) and the autograd form of the same kernels within 0.35 dB of each other.  Round 4's scrambled-ground-truth defect (a [3, H, W] view read as dense) cost
    0.6 - 2.2 dB with every unit test green: on the inconsistent stand-ins of rounds 1-4 (held-out PSNR 17 - 18 dB whatever one did)
    no test could have a bar; this one fails on it."""
    from das3r_amd.farm import run_sequence_job
    from das3r_amd.train import consistent_sequence
    dev = torch.device("cuda:0")
    seq = consistent_sequence(seed=0, **SINTEL_SHAPE)
    start = run_sequence_job(0, 20, dev, fused=True, seq=seq)
    direct = run_sequence_job(0, 4000, dev, fused=True, seq=seq)
    monkeypatch.setenv("DAS3R_FAST_STEP", "0")
    autograd = run_sequence_job(0, 4000, dev, fused=True, seq=seq)
    print("held-out static-region PSNR: after 20 iterations", start["psnr"], "direct", direct["psnr"], "autograd form", autograd["psnr"],
          "iterations/s", direct["iters_per_s"], autograd["iters_per_s"])
    assert start["ok"] == direct["ok"] == autograd["ok"] == 1 and direct["n_splats"] == 20 * 512 * 208
    assert start["psnr"] < PSNR_START_SINTEL, start
    assert direct["psnr"] >= PSNR_BAR_SINTEL and autograd["psnr"] >= PSNR_BAR_SINTEL, (direct, autograd)
    # (two arithmetics of the same 4000-iteration optimisation: the autograd form adds the pose gradient and the mask gradient with
    #  torch's own kernels, in another order; measured 0.03 - 0.21 dB apart over five pairs of runs)
    assert abs(direct["psnr"] - autograd["psnr"]) <= 0.35, (direct["psnr"], autograd["psnr"])


def test_consistent_job_three_forms_agree(monkeypatch):
    """The same bar at a size the reference's own plain-PyTorch iteration finishes in seconds (12 frames of 256 x 104, 1000 iterations):
    unfused (torch ops + torch.optim.Adam around the HIP rasterizer: what unmodified DAS3R runs on the drop-in), the autograd form of
    the fused kernels and the direct iteration end within 0.1 dB of each other (measured: 0.01) and above the bar measured for this size."""
    from das3r_amd.farm import run_sequence_job
    from das3r_amd.train import consistent_sequence
    dev = torch.device("cuda:0")
    seq = consistent_sequence(frames=12, W=256, H=104, focal=300.0, n_splats=8000, seed=4)
    direct = run_sequence_job(0, 1000, dev, fused=True, seq=seq)
    unfused = run_sequence_job(0, 1000, dev, fused=False, seq=seq)
    monkeypatch.setenv("DAS3R_FAST_STEP", "0")
    autograd = run_sequence_job(0, 1000, dev, fused=True, seq=seq)
    print("three forms:", direct["psnr"], autograd["psnr"], unfused["psnr"])
    assert direct["ok"] == autograd["ok"] == unfused["ok"] == 1
    for r in (direct, autograd, unfused):
        assert r["psnr"] >= 36.0, (direct, autograd, unfused)   # measured 36.49 - 36.54 for the three forms
    assert max(r["psnr"] for r in (direct, autograd, unfused)) - min(r["psnr"] for r in (direct, autograd, unfused)) <= 0.1


def test_many_short_jobs_in_flight_all_finish():
    """Three worker threads, 24 short jobs of different shapes: every job starts with forwards of a shape its thread has not seen (the
    exactly sized path, whose count comes through the preprocess kernel's arrival words) while the other threads keep the GPU busy.
    Round 5 found the arrival words' one-time fill on the null stream — which a worker's non-blocking stream does not wait for — landing
    after the first preprocess kernel had started counting in them: "the device never delivered the result of this forward to the host
    mailbox", once in a few dozen two-job runs.  Every job must finish (ok = 1) and a job's result must not depend on its neighbours."""
    from das3r_amd import _lib
    from das3r_amd.farm import run_jobs, run_sequence_job
    from das3r_amd.train import consistent_sequence
    dev = torch.device("cuda:0")
    shapes = [(6, 96, 64), (7, 128, 80), (6, 160, 96), (8, 112, 64)]
    seqs = [consistent_sequence(frames=f, W=w, H=h, focal=1.2 * w, n_splats=2500, seed=40 + i) for i, (f, w, h) in enumerate(shapes)]
    before = _lib.stats()
    solo = [run_sequence_job(i, 15, dev, fused=True, seq=seqs[i]) for i in range(len(seqs))]
    jobs = [i % len(seqs) for i in range(24)]
    recs = run_jobs(jobs, lambda i: run_sequence_job(i, 15, dev, fused=True, seq=seqs[i]), 3, dev)
    torch.cuda.synchronize()
    assert all(r["ok"] == 1 for r in recs), [r for r in recs if r["ok"] != 1]
    for i, r in zip(jobs, recs):
        assert r["psnr"] == solo[i]["psnr"] and r["l1"] == solo[i]["l1"], (i, r, solo[i])
    after = _lib.stats()
    assert after["failed_checks"] == before["failed_checks"]


def test_farm_cli_on_sequence_directories_with_two_jobs_in_flight(tmp_path):
    """The product as a user runs it for BASELINE configs[4]'s protocol (scripts/testing_psnr_davis.sh:35-59 + get_testing_psnr_davis.py:8-22):
    `python -m das3r_amd.farm --data <dir of preprocessed sequences> --out <dir> --gt-dynamic-mask <dir> --fused` — three self-consistent
    sequences written in the reference's on-disk formats (COLMAP text, TUM trajectory, per-frame npy maps, PNG images, Sintel-style
    ground-truth masks), two of them in flight at a time on one GPU (the default), the held-out static-region report, the LaTeX rows, and
    per sequence what the reference writes (test_log.txt, point_cloud.ply, pose npy).  The table equals what each sequence gives alone."""
    import os
    import subprocess
    import sys
    import numpy as np
    from PIL import Image
    from das3r_amd import io_formats as io
    from das3r_amd.farm import run_sequence_job
    from das3r_amd.train import consistent_sequence
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    data, out, masks = tmp_path / "data", tmp_path / "out", tmp_path / "gt"
    names = ["alley_1", "market_2", "temple_3"]
    for i, n in enumerate(names):
        seq = consistent_sequence(frames=12, W=128, H=80, focal=150.0, n_splats=3000, seed=60 + i)
        io.write_sequence_dir(seq, str(data / n))
        os.makedirs(masks / n)
        for f, m in enumerate(seq["gt_dynamic_masks"]):   # Sintel convention: frame_%04d.png, one-based, 0 / 255
            Image.fromarray((m * 255).astype(np.uint8)).save(masks / n / f"frame_{f + 1:04d}.png")
    env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    p = subprocess.run([sys.executable, "-m", "das3r_amd.farm", "--data", str(data), "--out", str(out), "--gt-dynamic-mask", str(masks),
                        "--iterations", "120", "--fused"], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert lines[-3] == "Scene & alley-1 & market-2 & temple-3& average" and lines[-2].startswith("PSNR & ")
    table = [float(v) for v in lines[-2].replace("PSNR &", "").split("&")]
    assert lines[-1].endswith("over 3/3 sequences")
    dev = torch.device("cuda:0")
    for i, n in enumerate(names):
        log = (out / n / "test_log.txt").read_text().strip().split("\\n")[-1]
        assert log.startswith("[ITER 120] Evaluating test: L1 ")
        assert (out / n / "point_cloud" / "iteration_120" / "point_cloud.ply").stat().st_size > 0 and (out / n / "pose" / "pose_120.npy").exists()
        solo = run_sequence_job(i, 120, dev, seq_dir=str(data / n), fused=True, gt_mask_dir=str(masks / n), dataset="sintel")
        assert solo["ok"] == 1 and abs(table[i] - solo["psnr"]) < 0.006, (n, table[i], solo["psnr"])   # (the table prints two decimals)
        assert abs(float(log.split()[-1]) - solo["psnr"]) < 1e-9
    assert abs(table[3] - sum(table[:3]) / 3) < 0.011


NOISY = dict(depth_noise=0.02, pose_noise=0.0065)   # the predictor's errors put back in: 2 % relative depth error per pixel, 6.5e-3 scene units of pose error
PSNR_BAR_NOISY = 27.8      # dB; measured 28.84 (direct form) and 28.83 (autograd form), seed 0; 28.02 - 28.03 for the three forms at the small size: the regime of the published numbers (29.03 dB Sintel market_2, 25.70 dB DAVIS mean)
PSNR_CEIL_NOISY = 31.0     # ... and well below the 43.9 dB of the noiseless sequence: the noise, not the optimiser, sets the level


def test_noisy_input_job_lands_in_the_published_regime(monkeypatch):
    """the 43 dB bar above feeds DAS3R perfect depth maps and perfect poses — the regime of the published numbers
    (29.03 dB on Sintel market_2, 25.70 dB DAVIS mean: /root/reference/index.html:284-286, assets/table2.png) is the one where the
    predictor's errors are in the input: poses the optimiser has to MOVE (the camera optimizer steps only on frames above 26 dB,
    /root/reference/train_gui.py:584-586 — at this level the gate opens on some iterations and stays shut on others), depth maps that put
    every Gaussian a little off its surface, conf_static learning under a loss that never goes to zero.  The same Sintel-shaped job with
    consistent_sequence(depth_noise=0.02, pose_noise=0.0065): ends in [PSNR_BAR_NOISY, PSNR_CEIL_NOISY], above where it started, the
    direct iteration and the autograd form of the same kernels within 0.5 dB of each other."""
    from das3r_amd.farm import run_sequence_job
    from das3r_amd.train import consistent_sequence
    dev = torch.device("cuda:0")
    seq = consistent_sequence(seed=0, **SINTEL_SHAPE, **NOISY)
    start = run_sequence_job(0, 20, dev, fused=True, seq=seq)
    direct = run_sequence_job(0, 4000, dev, fused=True, seq=seq)
    monkeypatch.setenv("DAS3R_FAST_STEP", "0")
    autograd = run_sequence_job(0, 4000, dev, fused=True, seq=seq)
    print("noisy inputs, held-out static-region PSNR: after 20 iterations", start["psnr"], "direct", direct["psnr"], "autograd form", autograd["psnr"])
    assert start["ok"] == direct["ok"] == autograd["ok"] == 1
    for r in (direct, autograd):
        assert PSNR_BAR_NOISY <= r["psnr"] <= PSNR_CEIL_NOISY, (start, direct, autograd)
        assert r["psnr"] > start["psnr"] + 1.5, (start, direct, autograd)
    assert abs(direct["psnr"] - autograd["psnr"]) <= 0.5, (direct["psnr"], autograd["psnr"])


def test_noisy_input_job_three_forms_agree(monkeypatch):
    """The noisy regime at the size the plain-PyTorch iteration finishes in seconds (12 frames of 256 x 104, 1000 iterations, 2 % depth
    and 8e-3 pose error): unfused (what unmodified DAS3R runs on the drop-in), the autograd form of the fused kernels and the direct
    iteration end within 0.3 dB of each other, in the same regime."""
    from das3r_amd.farm import run_sequence_job
    from das3r_amd.train import consistent_sequence
    dev = torch.device("cuda:0")
    seq = consistent_sequence(frames=12, W=256, H=104, focal=300.0, n_splats=8000, seed=4, depth_noise=0.02, pose_noise=0.008)
    direct = run_sequence_job(0, 1000, dev, fused=True, seq=seq)
    unfused = run_sequence_job(0, 1000, dev, fused=False, seq=seq)
    monkeypatch.setenv("DAS3R_FAST_STEP", "0")
    autograd = run_sequence_job(0, 1000, dev, fused=True, seq=seq)
    forms = (direct, autograd, unfused)
    print("noisy inputs, three forms:", [r["psnr"] for r in forms])
    assert all(r["ok"] == 1 for r in forms)
    assert all(26.0 <= r["psnr"] <= 33.0 for r in forms), forms
    assert max(r["psnr"] for r in forms) - min(r["psnr"] for r in forms) <= 0.3, forms
