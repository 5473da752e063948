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


def make_camera(uid, image, focal, W, H, device, focal_y=None):
    fovx, fovy = focal2fov(focal, W), focal2fov(focal if focal_y is None else focal_y, H)
    return SimpleNamespace(uid=uid, FoVx=fovx, FoVy=fovy, image_width=W, image_height=H, original_image=image,
                           projection_matrix=projection_matrix(0.01, 100.0, fovx, fovy).transpose(0, 1).to(device))


def train_step(model: SplatModel, cam, opt: OptimParams, iteration, pipe, background, fused=False):
    """One iteration of the reference hot loop (train_gui.py:532-589).  Returns (loss, psnr_frame, render package)."""
    model.update_learning_rate(iteration)
    if iteration % 3000 == 0:
        model.oneupSHdegree()
    pose = model.get_RT(cam.uid)
    pkg = das3r_render(cam, model, pipe, background, camera_pose=pose, fused=fused)
    image = pkg["render"]
    gt = cam.original_image
    static = model._conf_static[cam.uid]
    if fused:   # opt-in: the masked L1 + SSIM loss and the frame MSE as one HIP kernel each way
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
    Lssim = ssim(image, gt, size_average=False)
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


def train(model, cameras, opt: OptimParams, iterations, pipe=None, background=None, seed=0, log_every=0, fused=False):
    """Random camera without replacement per epoch (train_gui.py:546-555).  Returns dict(loss, psnr, iters_per_s)."""
    pipe = pipe or SimpleNamespace(debug=False, compute_cov3D_python=False, convert_SHs_python=False)
    dev = model.get_xyz.device
    background = background if background is not None else torch.zeros(3, device=dev)
    rng = random.Random(seed)
    stack, ema, last_psnr = [], torch.zeros((), device=dev), torch.zeros((), device=dev)
    if dev.type == "cuda":
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(1, iterations + 1):
        if not stack:
            stack = list(cameras)
        cam = stack.pop(rng.randint(0, len(stack) - 1))
        loss, p, _ = train_step(model, cam, opt, it, pipe, background, fused=fused)
        ema = 0.4 * loss + 0.6 * ema          # stays on the device: a float() here would stall the host every iteration
        last_psnr = p
        if log_every and it % log_every == 0:
            print(f"[ITER {it}] loss {float(ema):.5f} psnr_frame {float(last_psnr):.2f}")
    if dev.type == "cuda":
        torch.cuda.synchronize()
    return dict(loss=float(ema), psnr=float(last_psnr), iters_per_s=iterations / (time.perf_counter() - t0))


def is_test_index(idx):
    """Held-out split of the reference (scene/dataset_readers.py:342-347)."""
    return (idx + 5) % 10 == 0


@torch.no_grad()
def psnr_report(model, cameras, dynamic_masks=None, pipe=None, background=None):
    """Mean masked PSNR / L1 over `cameras` (train_test_psnr.py:241-302): clamp the render to [0,1], mask both images with
    (1 - gt_dynamic_mask), per-channel psnr = 20 log10(1 / sqrt(mse)) then mean."""
    pipe = pipe or SimpleNamespace(debug=False, compute_cov3D_python=False, convert_SHs_python=False)
    dev = model.get_xyz.device
    background = background if background is not None else torch.zeros(3, device=dev)
    l1s, ps = [], []
    for cam in cameras:
        img = torch.clamp(das3r_render(cam, model, pipe, background, camera_pose=model.get_RT(cam.uid))["render"], 0.0, 1.0)
        gt = torch.clamp(cam.original_image, 0.0, 1.0)
        if dynamic_masks is not None:
            m = 1 - dynamic_masks[cam.uid].to(img.dtype)
            img, gt = img * m, gt * m
        l1s.append(float(l1_loss(img, gt)))
        ps.append(float(psnr(img, gt).mean()))
    n = max(len(ps), 1)
    return dict(l1=sum(l1s) / n, psnr=sum(ps) / n, views=len(ps))


def synthetic_sequence(frames=6, W=128, H=80, focal=110.0, n_splats=6000, seed=0, device="cuda"):
    """A static random splat cloud seen from `frames` slightly different poses -> per-frame images, depth proxies,
    confidences, dynamic maps, intrinsics and poses in the layout SplatModel.create_from_frames expects."""
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
        depths.append(torch.full((H, W), 5.0) + 0.5 * torch.rand(H, W, generator=g))
    images = torch.stack(images)
    K = torch.tensor([[focal, 0, W / 2], [0, focal, H / 2], [0, 0, 1.0]]).repeat(frames, 1, 1).to(dev)
    return dict(images=images, depths=torch.stack(depths).to(dev), confs=torch.full((frames, H, W), 2.0, device=dev),
                dyna_avg=torch.zeros(frames, H, W, device=dev), K=K, cam2world=torch.stack(c2w).to(dev),
                w2c_pose7=torch.stack(poses7).to(dev), focal=focal, W=W, H=H)


def build_from_sequence(seq, sh_degree=3):
    model = SplatModel(sh_degree).create_from_frames(seq["images"], seq["depths"], seq["confs"], seq["dyna_avg"], seq["K"],
                                                     seq["cam2world"], seq["w2c_pose7"])
    dev = seq["images"].device
    K = seq["K"]   # per-frame focals (cameras.txt: scene/dataset_readers.py:139-147); principal point at the image centre
    cams = [make_camera(i, seq["images"][i], float(K[i, 0, 0]), seq["W"], seq["H"], dev, focal_y=float(K[i, 1, 1]))
            for i in range(seq["images"].shape[0])]
    return model, cams
