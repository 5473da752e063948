"""Train-step harness and held-out PSNR report — the repo's counterparts of
/root/reference/train_gui.py:530-589 (one optimisation iteration) and /root/reference/train_test_psnr.py:241-302
(masked PSNR over held-out frames, test split = indices with (idx + 5) % 10 == 0, scene/dataset_readers.py:342-347).

The datasets DAS3R trains on (Sintel / DAVIS + predictor outputs) are not available offline, so `synthetic_sequence`
builds a small multi-frame scene with the same structure (per-frame image, depth, confidence, dynamic map, pose,
intrinsics) by rendering a random static splat cloud with the rasterizer itself.
"""
import math
import random
import time
from types import SimpleNamespace

import torch

from .camera import focal2fov, projection_matrix
from .losses import l1_loss, psnr, ssim
from .model import OptimParams, SplatModel
from .render import das3r_render


def make_camera(uid, image, focal, W, H, device, focal_y=None, camera_center=None):
    fovx, fovy = focal2fov(focal, W), focal2fov(focal if focal_y is None else focal_y, H)
    center = torch.zeros(3, device=device) if camera_center is None else camera_center.to(device)   # (only render()'s convert_SHs_python mode reads it)
    return SimpleNamespace(uid=uid, FoVx=fovx, FoVy=fovy, image_width=W, image_height=H, original_image=image, camera_center=center,
                           projection_matrix=projection_matrix(0.01, 100.0, fovx, fovy).transpose(0, 1).to(device))


def train_step(model: SplatModel, cam, opt: OptimParams, iteration, pipe, background, fused=False, fused_loss=None):
    """One iteration of the reference hot loop (train_gui.py:532-589).  Returns (loss, psnr_frame, render package).
    fused: the opt-in fused kernels of SURVEY.md section 8(f); with fused optimizers on both parameter sets and the default `pipe` the
    iteration runs as a straight sequence of C-ABI calls without autograd (das3r_amd/fast_step.py; `model.fast_step = False` keeps
    the autograd form of round 3 — same kernels, the reference's Python around them).
    fused_loss=False with fused=True: the fused pre-transform and FusedAdam behind render() / the optimizers, the loss in torch ops and the
    camera gate a host-side `if`, as the reference's loop has them; fused_loss="ssim": what das3r_amd.integrate.patch() gives an UNMODIFIED
    train_gui.py since round 6 — the same, with the loop's `ssim` call answered by the fused SSIM map (fused.ssim_map)."""
    ssim_fn = ssim
    if fused_loss == "ssim":
        from .fused import ssim_map
        ssim_fn, fused_loss = (lambda a, b, size_average=False: ssim_map(a, b)), False
    fused_loss = bool(fused) if fused_loss is None else bool(fused_loss)
    if fused and fused_loss:
        from . import fast_step
        if fast_step.available(model, pipe):
            return fast_step.train_step(model, cam, opt, iteration, pipe, background)
    model.update_learning_rate(iteration)
    if iteration % 3000 == 0:
        model.oneupSHdegree()
    pose = model.get_RT(cam.uid)
    pkg = das3r_render(cam, model, pipe, background, camera_pose=pose, fused=fused)
    image = pkg["render"]
    gt = cam.original_image
    static = model._conf_static[cam.uid]
    if fused and fused_loss:   # opt-in: the masked L1 + SSIM loss and the frame MSE as one HIP kernel each way
        from .fused import masked_photometric_loss
        loss, mse = masked_photometric_loss(image, gt, static, opt.lambda_dssim)
        psnr_frame = (20 * torch.log10(1.0 / torch.sqrt(mse))).mean()
        loss.backward(retain_graph=True)
        with torch.no_grad():
            model.optimizer.step()
            model.optimizer.zero_grad(set_to_none=True)
            if hasattr(model.optimizer_cam, "_gate_state"):   # FusedAdam: the 26 dB gate is evaluated on the device
                model.optimizer_cam.step(gate=psnr_frame, threshold=opt.psnr_threshold)
            elif psnr_frame > opt.psnr_threshold:
                model.optimizer_cam.step()
            model.optimizer_cam.zero_grad(set_to_none=True)
        return loss.detach(), psnr_frame.detach(), pkg
    image = image * static
    gt = gt * static
    Ll1 = l1_loss(image, gt, reduce=False)
    Lssim = ssim_fn(image, gt, size_average=False)
    psnr_frame = psnr(image, gt).mean()
    loss = ((1.0 - opt.lambda_dssim) * Ll1 + opt.lambda_dssim * (1.0 - Lssim)).mean()
    loss.backward(retain_graph=True)
    with torch.no_grad():
        model.optimizer.step()
        model.optimizer.zero_grad(set_to_none=True)
        if psnr_frame > opt.psnr_threshold:
            model.optimizer_cam.step()
        model.optimizer_cam.zero_grad(set_to_none=True)
    return loss.detach(), psnr_frame.detach(), pkg


def save_checkpoint(path, model, iteration, loop_state=None):
    """<model_path>/chkpnt<iteration>.pth = (model.capture(), iteration): the file train_gui.py:626-628 writes and the reference's
    restore() reads; beside it chkpnt<iteration>.das3r.pth = what the reference leaves out (SplatModel.capture_extras) and the state
    of the loop (the camera stack, the generator that draws from it, the running loss), so that a resumed job continues EXACTLY where
    the killed one was.  Both written to a temporary name first: a job killed while saving leaves the previous checkpoint intact."""
    import os
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    extras = dict(model=model.capture_extras(), loop=loop_state, iteration=iteration)
    for target, payload in ((path[:-4] + ".das3r.pth", extras), (path, (model.capture(), iteration))):   # (the main file last: it names a complete pair)
        torch.save(payload, target + ".tmp")
        os.replace(target + ".tmp", target)


def latest_checkpoint(model_dir):
    """-> (path, iteration) of the newest complete chkpnt<iteration>.pth pair under model_dir, or (None, 0)."""
    import os
    import re
    best = (None, 0)
    for f in (os.listdir(model_dir) if model_dir and os.path.isdir(model_dir) else []):
        m = re.fullmatch(r"chkpnt(\d+)\.pth", f)
        if m and int(m.group(1)) > best[1] and os.path.exists(os.path.join(model_dir, f[:-4] + ".das3r.pth")):
            best = (os.path.join(model_dir, f), int(m.group(1)))
    return best


def load_checkpoint(path, model, opt, fused=False, device=None):
    """Restore `model` from save_checkpoint's pair (or from a reference checkpoint alone, into a model built from its sequence).
    -> (iteration, loop_state or None)"""
    import os
    capture, iteration = torch.load(path, map_location=device, weights_only=False)
    extras_path = path[:-4] + ".das3r.pth"
    extras = torch.load(extras_path, map_location=device, weights_only=False) if os.path.exists(extras_path) else None
    if extras is not None and extras.get("iteration") != iteration:
        extras = None   # (a pair torn by a kill between the two renames: the extras belong to an older checkpoint of the same name — cannot happen with distinct iterations, kept as a guard)
    model.restore(capture, opt, extras=extras["model"] if extras else None, fused=fused, device=device)
    return int(iteration), (extras["loop"] if extras else None)


def train(model, cameras, opt: OptimParams, iterations, pipe=None, background=None, seed=0, log_every=0, fused=False,
          test_cameras=None, gt_dynamic_masks=None, on_progress=None, start_iteration=1, loop_state=None, checkpoint_every=0,
          checkpoint_dir=None):
    """Random camera without replacement per epoch (train_gui.py:546-555).  With test_cameras: train_test_psnr.py's loop, which
    walks the held-out views whenever the training stack has run empty (test_pose_pass).  Returns dict(loss, psnr, iters_per_s).
    checkpoint_every / checkpoint_dir: write chkpnt<iteration>.pth every so many iterations (train_gui.py:626-628 --checkpoint_iterations);
    start_iteration / loop_state: continue a job from load_checkpoint's result — the iterations that follow are the ones the
    uninterrupted job would have run (same cameras in the same order, same schedules, same optimizer moments)."""
    pipe = pipe or SimpleNamespace(debug=False, compute_cov3D_python=False, convert_SHs_python=False)
    dev = model.get_xyz.device
    background = background if background is not None else torch.zeros(3, device=dev)
    rng = random.Random(seed)
    stack, ema, last_psnr = [], torch.zeros((), device=dev), torch.zeros((), device=dev)
    if loop_state is not None:
        rng.setstate(loop_state["rng"])
        by_uid = {c.uid: c for c in cameras}
        stack = [by_uid[u] for u in loop_state["stack"]]
        ema, last_psnr = loop_state["ema"].to(dev), loop_state["last_psnr"].to(dev)
        if loop_state.get("library") is not None and dev.type == "cuda":
            from . import _lib
            with torch.cuda.device(dev):
                _lib.learning(tuple(loop_state["library"]))
    if dev.type == "cuda":
        torch.cuda.current_stream(dev).synchronize()   # (this job's stream only: other jobs may share the GPU — farm.run_jobs)
    t0 = time.perf_counter()
    for it in range(start_iteration, iterations + 1):
        if not stack:
            stack = list(cameras)
        cam = stack.pop(rng.randint(0, len(stack) - 1))
        loss, p, _ = train_step(model, cam, opt, it, pipe, background, fused=fused)
        if not stack and test_cameras and model.enable_test:
            test_pose_pass(model, test_cameras, gt_dynamic_masks, opt, pipe, background, rng, fused=fused)
        ema = torch.lerp(ema, loss, 0.4)      # 0.4 loss + 0.6 ema, one kernel; stays on the device: a float() here would stall the host every iteration
        last_psnr = p
        if on_progress is not None and it % 256 == 0:   # (farm.Rendezvous.tick: this rank's main thread is getting somewhere)
            on_progress()
        if log_every and it % log_every == 0:
            print(f"[ITER {it}] loss {float(ema):.5f} psnr_frame {float(last_psnr):.2f}")
        if checkpoint_every and checkpoint_dir and it % checkpoint_every == 0 and it < iterations:
            import os
            library = None
            if dev.type == "cuda":   # which forward kernel the shape has, and where its re-decision schedule stands (include/das3r_raster.h)
                from . import _lib
                with torch.cuda.device(dev):
                    library = _lib.learning()
            save_checkpoint(os.path.join(checkpoint_dir, f"chkpnt{it}.pth"), model, it,
                            dict(rng=rng.getstate(), stack=[c.uid for c in stack], ema=ema.detach().clone(), last_psnr=last_psnr.detach().clone(),
                                 library=library))
    if dev.type == "cuda":
        torch.cuda.current_stream(dev).synchronize()
    done = max(iterations - start_iteration + 1, 1)
    return dict(loss=float(ema), psnr=float(last_psnr), iters_per_s=done / (time.perf_counter() - t0))


def is_test_index(idx):
    """Held-out split of the reference (scene/dataset_readers.py:342-347)."""
    return (idx + 5) % 10 == 0


def resize_mask_nearest(mask, H, W):
    """gt_dynamic_mask as the reference's Camera stores it (scene/cameras.py:60-67): [h,w] bool/float -> float [3,H,W],
    nearest-neighbour resized to the render size."""
    m = mask.to(torch.float32)[None].repeat(3, 1, 1)[None]
    return torch.nn.functional.interpolate(m, size=(H, W), mode="nearest")[0]


def test_pose_pass(model, test_cams, gt_dynamic_masks, opt: OptimParams, pipe, background, rng, fused=False):
    """The pass over the held-out views train_test_psnr.py runs whenever the training stack runs empty (:109-147): every test
    view, in random order, is rendered with its test pose, the loss against the ground truth under (1 - gt_dynamic_mask) is
    back-propagated, the Gaussian optimizer's gradients are dropped WITHOUT a step, and optimizer_cam is stepped when the frame
    PSNR exceeds the gate.  optimizer_cam holds the TRAINING poses, whose gradients are None here, so no parameter changes
     — the pass costs time and nothing else; it is reproduced for the iterations/s of configs[4].
    fused: the same render + loss kernels as the training step (round 3: with the PyTorch glue this pass was more than half of a
    DAVIS-shaped job — five 6.6 M-Gaussian views at ~40 ms each per 45 iterations of 3.7 ms), and no host sync on the gate."""
    stack = list(test_cams)
    direct = False
    if fused:
        from . import fast_step
        direct = fast_step.available(model, pipe)
    while stack:
        cam = stack.pop(rng.randint(0, len(stack) - 1))
        if direct:   # (das3r_amd/fast_step.py: the same render + loss + backward as a straight sequence of C-ABI calls)
            m = gt_dynamic_masks.get(cam.uid) if gt_dynamic_masks else None
            H, W = cam.image_height, cam.image_width
            static_hw = (1 - resize_mask_nearest(m, H, W)[0]).contiguous() if m is not None else _ones_hw(model, H, W)
            fast_step.test_pose_step(model, cam, static_hw, opt, background)
            continue
        pkg = das3r_render(cam, model, pipe, background, camera_pose=model.get_RT_test(cam.uid), fused=fused)
        m = gt_dynamic_masks.get(cam.uid) if gt_dynamic_masks else None
        if fused:
            from .fused import masked_photometric_loss
            H, W = cam.image_height, cam.image_width
            static_hw = (1 - resize_mask_nearest(m, H, W)[0]) if m is not None else torch.ones(H, W, device=pkg["render"].device)
            loss, mse = masked_photometric_loss(pkg["render"], cam.original_image, static_hw, opt.lambda_dssim)
            psnr_frame = (20 * torch.log10(1.0 / torch.sqrt(mse))).mean()
        else:
            static = 1 - resize_mask_nearest(m, cam.image_height, cam.image_width) if m is not None else 1.0
            image, gt = pkg["render"] * static, cam.original_image * static
            psnr_frame = psnr(image, gt).mean()
            loss = ((1.0 - opt.lambda_dssim) * l1_loss(image, gt, reduce=False) + opt.lambda_dssim * (1.0 - ssim(image, gt, size_average=False))).mean()
        loss.backward(retain_graph=True)
        with torch.no_grad():
            model.optimizer.zero_grad(set_to_none=True)
            # (FusedAdam evaluates the gate on the device and owns no gradient here: nothing to do, and no host sync on psnr_frame)
            if not hasattr(model.optimizer_cam, "_gate_state") and psnr_frame > opt.psnr_threshold:
                model.optimizer_cam.step()   # every gradient it owns is None: a no-op, as in the reference
            model.optimizer_cam.zero_grad(set_to_none=True)
            if model.test_Q.grad is not None:   # (test_Q / test_T do get gradients; nothing ever consumes them)
                model.test_Q.grad = model.test_T.grad = None


def _ones_hw(model, H, W):
    t = getattr(model, "_ones_hw", None)
    if t is None or tuple(t.shape) != (H, W):
        t = model._ones_hw = torch.ones(H, W, device=model.get_xyz.device)
    return t


@torch.no_grad()
def psnr_report(model, cameras, dynamic_masks=None, pipe=None, background=None, test_poses=False, iteration=None, log_dir=None,
                name="test"):
    """Held-out report of train_test_psnr.py:241-302.  Per view: clamp the render to [0,1], mask render and ground truth with
    (1 - gt_dynamic_mask) — the mask nearest-resized to the render size (scene/cameras.py:60-67) —, L1 = mean |d|, PSNR = mean
    over channels of 20 log10(1 / sqrt(mse_c)) (utils/image_utils.py:17-19), both accumulated in float64.  ONLY views that have a
    mask count: a view without one is rendered and skipped (`lens` is incremented inside the mask branch, :262-289), so with
    `dynamic_masks` given the averages run over the masked views; `dynamic_masks=None` is the harness's own mode for sequences
    without ground-truth masks (every view counts, unmasked).  test_poses: use model.get_RT_test (the reference's 'test' config)
    instead of the training poses.  log_dir: append the reference's line to <log_dir>/<name>_log.txt (:299-300).
    -> dict(l1, psnr, views, skipped)"""
    pipe = pipe or SimpleNamespace(debug=False, compute_cov3D_python=False, convert_SHs_python=False)
    dev = model.get_xyz.device
    background = background if background is not None else torch.zeros(3, device=dev)
    l1_sum = torch.zeros((), dtype=torch.float64, device=dev)
    psnr_sum = torch.zeros((), dtype=torch.float64, device=dev)
    lens = skipped = 0
    for cam in cameras:
        pose = model.get_RT_test(cam.uid) if test_poses else model.get_RT(cam.uid)
        with torch.no_grad():   # evaluation: the rasterizer then examines the forward's self-check itself (no backward will)
            img = torch.clamp(das3r_render(cam, model, pipe, background, camera_pose=pose)["render"], 0.0, 1.0)
        gt = torch.clamp(cam.original_image, 0.0, 1.0)
        if dynamic_masks is not None:
            m = dynamic_masks.get(cam.uid) if hasattr(dynamic_masks, "get") else dynamic_masks[cam.uid]
            if m is None:
                skipped += 1
                continue
            static = 1 - resize_mask_nearest(m.to(dev), img.shape[1], img.shape[2])
            img, gt = img * static, gt * static
        l1_sum += l1_loss(img, gt).mean().double()
        psnr_sum += psnr(img, gt).mean().double()
        lens += 1
    l1_v = float(l1_sum) / lens if lens else float("nan")     # (the reference divides by zero here when no view has a mask)
    psnr_v = float(psnr_sum) / lens if lens else float("nan")
    if log_dir is not None and lens:
        import os
        os.makedirs(log_dir, exist_ok=True)
        with open(os.path.join(log_dir, f"{name}_log.txt"), "a") as f:
            f.write(f"[ITER {iteration}] Evaluating {name}: L1 {l1_v} PSNR {psnr_v}\n")
    return dict(l1=l1_v, psnr=psnr_v, views=lens, skipped=skipped)


def scrape_test_logs(root, exp, log_name="test_log.txt"):
    """/root/reference/scripts/get_testing_psnr_davis.py:8-17: {scene: last number of the last line of <root>/<scene>/<exp>/
    test_log.txt}, scenes in sorted order."""
    import os
    out = {}
    for scene in sorted(os.listdir(root)):
        path = os.path.join(root, scene, exp, log_name)
        if os.path.isdir(os.path.join(root, scene)) and os.path.exists(path):
            with open(path) as f:
                out[scene] = float(f.read().strip().split("\n")[-1].split()[-1])
    return out


def latex_rows(results):
    """The two rows get_testing_psnr_davis.py:19-22 prints for {scene: psnr}."""
    avg = sum(results.values()) / len(results) if results else 0
    head = "Scene & " + " & ".join(results.keys()).replace("_", "-") + "& average"
    row = "PSNR & " + " & ".join(f"{v:.2f}" for v in results.values()) + f" & {avg:.2f} "
    return head, row


def synthetic_sequence(frames=6, W=128, H=80, focal=110.0, n_splats=6000, seed=0, device="cuda", depth="noise"):
    """A static random splat cloud seen from `frames` slightly different poses -> per-frame images, depth proxies,
    confidences, dynamic maps, intrinsics and poses in the layout SplatModel.create_from_frames expects.
    depth: "noise" — 5 .. 5.5, independent per pixel (rounds 1-3: every tile sees the whole depth slab); "smooth" (round 4) — a
    smooth relief with 1 % of noise, i.e. the spatially coherent depth a predictor's depth maps have: a tile sees a thin band of the
    scene's depth range (ledger (ao)).  Both kinds of maps are UNRELATED to the cloud that made the images: such a
    sequence times a job, its held-out PSNR (17 - 18 dB) says nothing.  depth="rendered" (round 5) is the self-consistent one:
    consistent_sequence below."""
    if depth == "rendered":
        return consistent_sequence(frames=frames, W=W, H=H, focal=focal, n_splats=n_splats, seed=seed, device=device)
    from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    from .synth import make_scene
    dev = torch.device(device)
    sc = make_scene(P=n_splats, W=W, H=H, focal=focal, sh_degree=0, seed=100 + seed, s_px=(2.0, 6.0)).to(dev)
    g = torch.Generator().manual_seed(seed)
    images, poses7, c2w, depths = [], [], [], []
    for f in range(frames):
        t = torch.tensor([0.05 * (f - frames / 2), 0.02 * math.sin(f), 0.0]) + 0.005 * torch.randn(3, generator=g)
        view = torch.eye(4)
        view[:3, 3] = t                                    # world -> camera (pure translation)
        fovx, fovy = focal2fov(focal, W), focal2fov(focal, H)
        view_t = view.t().contiguous().to(dev)
        proj = (view_t @ projection_matrix(0.01, 100.0, fovx, fovy).t().to(dev)).contiguous()
        rs = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=math.tan(fovx / 2), tanfovy=math.tan(fovy / 2),
                                           bg=torch.zeros(3, device=dev), scale_modifier=1.0, viewmatrix=view_t, projmatrix=proj,
                                           sh_degree=0, campos=(-t).to(dev), prefiltered=False, debug=False)
        with torch.no_grad():
            img, _ = GaussianRasterizer(rs)(means3D=sc.means3D, means2D=torch.zeros_like(sc.means3D), opacities=sc.opacities,
                                            shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
        images.append(img.clamp(0, 1))
        poses7.append(torch.cat([torch.tensor([1.0, 0, 0, 0]), t]))
        c2w.append(torch.linalg.inv(view))
        if depth == "smooth":
            vv, uu = torch.meshgrid(torch.linspace(-1, 1, H), torch.linspace(-1, 1, W), indexing="ij")
            relief = 4.0 + 2.0 * torch.sin(3.0 * uu + 0.3 * f) * torch.cos(2.0 * vv) + 1.5 * torch.cos(5.0 * uu + 1.0)
            depths.append(relief * (1.0 + 0.01 * torch.randn(H, W, generator=g)))
        else:
            depths.append(torch.full((H, W), 5.0) + 0.5 * torch.rand(H, W, generator=g))
    images = torch.stack(images)
    K = torch.tensor([[focal, 0, W / 2], [0, focal, H / 2], [0, 0, 1.0]]).repeat(frames, 1, 1).to(dev)
    return dict(images=images, depths=torch.stack(depths).to(dev), confs=torch.full((frames, H, W), 2.0, device=dev),
                dyna_avg=torch.zeros(frames, H, W, device=dev), K=K, cam2world=torch.stack(c2w).to(dev),
                w2c_pose7=torch.stack(poses7).to(dev), focal=focal, W=W, H=H)


def consistent_sequence(frames=22, W=512, H=208, focal=600.0, n_splats=20000, seed=0, device="cuda", moving=True, depth_noise=0.0,
                        pose_noise=0.0):
    """A SELF-CONSISTENT synthetic sequence: images, depth maps and poses all come from ONE scene, so that the
    optimisation DAS3R runs on it has something to converge to and its held-out PSNR means something (a stand-in for BASELINE
    configs[2] / [4], whose datasets are not in the container; protocol /root/reference/train_test_psnr.py:241-302, split
    scene/dataset_readers.py:342-347).

      * the scene: `n_splats` opaque-ish Gaussians (opacity 0.9, 2 - 6 pixels) ON a smooth relief z(u, v) (das3r_amd.synth
        make_scene(coherent=...): the surface of a real scene), coloured at random — a textured surface;
      * per frame: the image rendered by the rasterizer from that frame's pose, and the DEPTH MAP RENDERED FROM THE SAME CLOUD: one
        more render with colors_precomp = (z_camera, 1, 0) gives sum(w z) and sum(w) = 1 - T_final per pixel, depth = their ratio
        (the expected depth of the blend; where less than 5 % of a pixel is covered, the relief itself) — what a perfect depth
        predictor would hand DAS3R.  depth_noise / pose_noise: relative depth error / pose translation error (the predictor's) on top;
      * a MOVING OBJECT (`moving`): a disc that crosses the frame, painted over the images in its own colour, nearer than the surface
        in the depth maps, with dyna_avg = 1 inside it (DAS3R's dynamic map: conf_static = 1 - dyna_avg keeps those pixels'
        Gaussians transparent and masks them in the loss) and returned per frame as `gt_dynamic_masks` (bool numpy arrays: the report
        then measures the static region only, as train_test_psnr.py does with --gt_dynamic_mask).
    -> the dict synthetic_sequence returns, plus gt_dynamic_masks."""
    from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    from .synth import make_scene
    dev = torch.device(device)
    sc = make_scene(P=n_splats, W=W, H=H, focal=focal, sh_degree=0, seed=100 + seed, s_px=(2.0, 6.0), opacity=0.9, coherent=0.002).to(dev)
    g = torch.Generator().manual_seed(seed)
    fovx, fovy = focal2fov(focal, W), focal2fov(focal, H)
    zero3 = torch.zeros(3, device=dev)
    vv, uu = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    disc_rgb = torch.tensor([0.9, 0.2, 0.1]).view(3, 1, 1)
    radius = 0.11 * H + 4.0
    images, poses7, c2w, depths, dyna, masks = [], [], [], [], [], []
    ur, vr = (uu + 0.5 - W / 2) / (W / 2), (vv + 0.5 - H / 2) / (H / 2)      # the relief of das3r_amd.synth.make_scene(coherent=...) at the pixel centres
    relief = 4.0 + 2.0 * torch.sin(3.0 * ur) * torch.cos(2.0 * vr) + 1.5 * torch.cos(5.0 * ur + 1.0)
    for f in range(frames):
        t = torch.tensor([0.05 * (f - frames / 2), 0.02 * math.sin(f), 0.0]) + 0.005 * torch.randn(3, generator=g)
        view = torch.eye(4)
        view[:3, 3] = t                                    # world -> camera (pure translation)
        view_t = view.t().contiguous().to(dev)
        proj = (view_t @ projection_matrix(0.01, 100.0, fovx, fovy).t().to(dev)).contiguous()
        rs = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=math.tan(fovx / 2), tanfovy=math.tan(fovy / 2),
                                           bg=zero3, scale_modifier=1.0, viewmatrix=view_t, projmatrix=proj,
                                           sh_degree=0, campos=(-t).to(dev), prefiltered=False, debug=False)
        m2 = torch.zeros_like(sc.means3D)
        with torch.no_grad():
            img, _ = GaussianRasterizer(rs)(means3D=sc.means3D, means2D=m2, opacities=sc.opacities, shs=sc.shs, scales=sc.scales,
                                            rotations=sc.rotations)
            zc = sc.means3D[:, 2] + float(t[2])            # camera-space depth of every Gaussian in this frame
            zcol = torch.stack([zc, torch.ones_like(zc), torch.zeros_like(zc)], 1).contiguous()
            zimg, _ = GaussianRasterizer(rs)(means3D=sc.means3D, means2D=m2, opacities=sc.opacities, colors_precomp=zcol, scales=sc.scales,
                                             rotations=sc.rotations)
        cover = zimg[1]
        # (where less than 5 % of a pixel is covered: the relief itself, not a constant — a constant would put thousands of Gaussians at
        #  EXACTLY one depth, a wall of ties no depth predictor produces and the worst case of the segmented binning)
        d = torch.where(cover > 0.05, zimg[0] / cover.clamp_min(1e-6), relief.to(dev) - float(t[2])).cpu()
        img = img.clamp(0, 1).cpu()
        mask = torch.zeros(H, W, dtype=torch.bool)
        if moving:   # the disc crosses the frame from left to right, bobbing
            cx = W * (0.12 + 0.76 * f / max(frames - 1, 1))
            cy = H * (0.5 + 0.22 * math.sin(0.7 * f))
            mask = (uu - cx) ** 2 + (vv - cy) ** 2 <= radius * radius
            img = torch.where(mask[None], disc_rgb.expand(3, H, W), img)
            d = torch.where(mask, 0.6 * d, d)
        if depth_noise:
            d = d * (1.0 + depth_noise * torch.randn(H, W, generator=g))
        tp = t + pose_noise * torch.randn(3, generator=g) if pose_noise else t
        vp = torch.eye(4)
        vp[:3, 3] = tp
        images.append(img)
        depths.append(d)
        dyna.append(mask.float())
        masks.append(mask.numpy())
        poses7.append(torch.cat([torch.tensor([1.0, 0, 0, 0]), tp]))
        c2w.append(torch.linalg.inv(vp))
    K = torch.tensor([[focal, 0, W / 2], [0, focal, H / 2], [0, 0, 1.0]]).repeat(frames, 1, 1).to(dev)
    return dict(images=torch.stack(images).to(dev), depths=torch.stack(depths).to(dev), confs=torch.full((frames, H, W), 2.0, device=dev),
                dyna_avg=torch.stack(dyna).to(dev), K=K, cam2world=torch.stack(c2w).to(dev), w2c_pose7=torch.stack(poses7).to(dev),
                focal=focal, W=W, H=H, gt_dynamic_masks=masks)


def split_sequence(seq):
    """Held-out split of the reference (scene/dataset_readers.py:336-347, eval mode): frames with (idx + 5) % 10 == 0 are test
    views, the others training views.  -> (train index list, test index list)"""
    F = seq["images"].shape[0]
    test = [i for i in range(F) if is_test_index(i)]
    return [i for i in range(F) if not is_test_index(i)], test


def build_from_sequence(seq, sh_degree=3, heldout=False):
    """-> (model, cameras) from every frame, or with heldout=True -> (model, train cameras, test cameras): Gaussians, training
    poses and conf_static come from the TRAINING frames only (the reference builds them from scene.train_cameras:
    scene/__init__.py:88-93), the held-out frames only contribute their poses (init_test_RT_seq) and their images as ground
    truth.  Camera uids index the model's per-frame tensors: 0..n_train-1 / 0..n_test-1."""
    dev = seq["images"].device
    F = seq["images"].shape[0]
    tr, te = split_sequence(seq) if heldout else (list(range(F)), [])
    if heldout and not te:   # short sequences (< 6 frames) have no held-out view under the reference's rule
        tr, te = list(range(F - 1)), [F - 1]
    sel = torch.tensor(tr, device=dev)
    model = SplatModel(sh_degree).create_from_frames(seq["images"][sel], seq["depths"][sel], seq["confs"][sel], seq["dyna_avg"][sel],
                                                     seq["K"][sel], seq["cam2world"][sel], seq["w2c_pose7"][sel])
    K = seq["K"]   # per-frame focals (cameras.txt: scene/dataset_readers.py:139-147); principal point at the image centre
    mk = lambda uid, i: make_camera(uid, seq["images"][i], float(K[i, 0, 0]), seq["W"], seq["H"], dev, focal_y=float(K[i, 1, 1]),
                                    camera_center=seq["cam2world"][i][:3, 3])
    cams = [mk(u, i) for u, i in enumerate(tr)]
    model.init_fov(cams[0].FoVx, cams[0].FoVy)
    if not heldout:
        return model, cams
    model.init_test_RT_seq(seq["w2c_pose7"][torch.tensor(te, device=dev)])
    test_cams = [mk(u, i) for u, i in enumerate(te)]
    for c, i in zip(test_cams, te):
        c.frame_index = i
    return model, cams, test_cams
