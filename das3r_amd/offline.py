"""Offline render of a trained sequence — the counterpart of /root/reference/render.py:72-123 (`render_set` / `render_sets`):
load <model_path>/point_cloud/iteration_N/point_cloud.ply, render every training camera, write
<model_path>/interp/ours_N/renders/%05d.png, and turn <model_path>/pose/pose_N.npy into pose/pose_interpolated.npy
(render.py:32-52 `save_interpolate_pose`, minus the two matplotlib trajectory plots, which are figures, not data).

    python -m das3r_amd.offline --model-path OUT/market_2 --source-path DATA/market_2 [--iteration -1] [--optimised-poses]

What is rendered.  The reference calls `render()` on a model it has just loaded from the PLY — which reads `pc.aggregated_mask`, an
attribute only `create_from_cameras` sets: as written, its offline renderer stops with an AttributeError.  The renderer that works
on a loaded PLY is the one the reference's own evaluation uses, `render_test` (gaussian_renderer/__init__.py:152-277: opacity x the
per-Gaussian `conf_static` as stored — exactly the two columns `load_ply` reads back, scene/gaussian_model.py:371-418); that is what
runs here (das3r_amd.render variant="test"), under torch.no_grad like the reference.  Poses: the reference renders each view from
the pose its camera was LOADED with (sparse/0/images.txt: `view.world_view_transform`), not from the optimised pose it saved;
`optimised_poses=True` (CLI --optimised-poses) takes pose/pose_N.npy instead, which is what one wants to look at.

Eval-mode throughput: `forward_throughput` times the same no-grad forward (the path that waits for the binning self-check inside
every call: rasterizer.py) and is what bench.py reports as `eval_forward`."""
import argparse
import os
import re
import time
from types import SimpleNamespace

import numpy as np
import torch

from .model import SplatModel
from .render import das3r_render


def tensor_from_camera(RT, device="cuda"):
    """utils/pose_utils.py:183-215 get_tensor_from_camera: 4x4 world-to-camera -> (qw, qx, qy, qz, tx, ty, tz), the quaternion by
    rotation2quad (io_formats.matrix_to_quat_wxyz, pinned by the reference's own outputs in tests/golden)."""
    from .io_formats import matrix_to_quat_wxyz
    m = RT.detach().cpu().numpy() if torch.is_tensor(RT) else np.asarray(RT)
    return torch.from_numpy(np.concatenate([matrix_to_quat_wxyz(m[:3, :3]), m[:3, 3]]).astype(np.float32)).to(device)


def save_image(chw, path):
    """torchvision.utils.save_image for one image: x -> clamp(255 x + 0.5, 0, 255) -> uint8 -> PNG (render.py:84-86 writes with it)."""
    from PIL import Image
    arr = chw.detach().float().mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to("cpu", torch.uint8).numpy()
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    Image.fromarray(arr).save(path, format="PNG")
    return arr


def search_for_max_iteration(folder):
    """utils/system_utils.py searchForMaxIteration: the largest N among <folder>/iteration_N."""
    its = [int(m.group(1)) for m in (re.fullmatch(r"iteration_(\d+)", f) for f in os.listdir(folder)) if m]
    if not its:
        raise FileNotFoundError(f"no iteration_* under {folder}")
    return max(its)


def load_trained_model(model_path, iteration=-1, sh_degree=3, device="cuda"):
    """GaussianModel.load_ply (scene/gaussian_model.py:371-418) into a SplatModel: `opacity_ori` as the opacity parameter, the
    per-Gaussian `conf_static` column [P, 1], active degree = maximum degree.  -> (model, iteration)"""
    from .io_formats import load_gaussians_ply
    if iteration == -1:
        iteration = search_for_max_iteration(os.path.join(model_path, "point_cloud"))
    g = load_gaussians_ply(os.path.join(model_path, "point_cloud", f"iteration_{iteration}", "point_cloud.ply"), max_sh_degree=sh_degree)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device=device, dtype=torch.float32)
    m = SplatModel(sh_degree)
    m._xyz, m._features_dc, m._features_rest = t(g["xyz"]), t(g["features_dc"]), t(g["features_rest"])
    m._opacity, m._conf_static, m._scaling, m._rotation = t(g["opacity"]), t(g["conf_static"]), t(g["scaling"]), t(g["rotation"])
    m.active_sh_degree = sh_degree
    return m, iteration


def save_interpolate_pose(model_path, iteration):
    """render.py:32-52 with its interpolation commented out, as it is there: pose_N.npy ([N, 4, 4] world-to-camera) -> 4x4 matrices
    rebuilt from the rotation and translation blocks -> pose/pose_interpolated.npy.  -> the array, or None without a pose file."""
    src = os.path.join(model_path, "pose", f"pose_{iteration}.npy")
    if not os.path.exists(src):
        return None
    org = np.load(src)
    out = np.stack([np.block([[p[:3, :3], p[:3, 3:4]], [np.zeros((1, 3)), np.ones((1, 1))]]) for p in org], 0)
    np.save(os.path.join(model_path, "pose", "pose_interpolated.npy"), out)
    return out


def sequence_cameras(seq, device="cuda"):
    """Every frame of a sequence dict (io_formats.load_sequence / train.consistent_sequence) as a camera with the pose it was loaded
    with: what Scene(..., shuffle=False).getTrainCameras() is for render.py, which runs with args.eval = False (no held-out split)."""
    from .train import make_camera
    K = seq["K"]
    cams = []
    for i in range(seq["images"].shape[0]):
        c = make_camera(i, seq["images"][i].to(device), float(K[i, 0, 0]), seq["W"], seq["H"], device, focal_y=float(K[i, 1, 1]),
                        camera_center=seq["cam2world"][i][:3, 3])
        c.pose7 = seq["w2c_pose7"][i].to(device=device, dtype=torch.float32)
        cams.append(c)
    return cams


PIPE = SimpleNamespace(debug=False, compute_cov3D_python=False, convert_SHs_python=False)


class _EvalState:
    """What a model's fused no-grad renders share: the pose matrices' buffer and the packed [P, (D + 1)^2, 3] SH tensor the rasterizer reads
    (get_features is a torch.cat of 190 bytes per Gaussian — rebuilt only when a parameter was written since)."""

    def __init__(self, model):
        dev = model._xyz.device
        self.mats = torch.empty(28, device=dev)
        self.e = torch.empty(0, device=dev)
        self.shs, self.versions = None, None

    def packed_sh(self, model):
        v = (model._features_dc._version, model._features_rest._version, model._features_dc.data_ptr(), model._features_rest.data_ptr())
        if self.shs is None or v != self.versions:
            self.shs, self.versions = model.get_features.detach().contiguous(), v
        return self.shs


@torch.no_grad()
def render_view_fused(model, view, pose7, background, pipe=PIPE):
    """render_test of one view (gaussian_renderer/__init__.py:152-277) with the pose pre-transform INSIDE the rasterizer's per-Gaussian
    kernel (include/das3r_raster.h das3r_raster_in.pre, as the direct training iteration uses it): R xyz + t, quaternion product, exp,
    sigmoid x the per-Gaussian conf_static column — the dozen PyTorch kernels of the reference's glue (a boolean-mask gather of every
    tensor among them: 190 bytes of SH per Gaussian copied per view) are not launched and the camera-frame tensors never exist.
    Same arithmetic up to the rounding of the pre-transform (tests: within the parity bars of the glue form).  -> (image, radii)"""
    import ctypes as C
    from . import _lib
    from .rasterizer import _forward_full, _on_device, _stream, check_forward
    from .render import _settings
    st = model.__dict__.get("_das3r_eval")
    if st is None:
        st = model.__dict__["_das3r_eval"] = _EvalState(model)
    dev = model._xyz.device
    lib = _lib.load()
    pose7 = pose7.detach().to(device=dev, dtype=torch.float32).contiguous()
    conf = model._conf_static.detach().reshape(-1)
    if conf.shape[0] != model._xyz.shape[0]:
        raise RuntimeError("render_view_fused renders a LOADED model (one conf_static value per Gaussian: load_trained_model)")
    with _on_device(dev):
        _lib.check(lib.das3r_pose_matrices(C.c_void_p(pose7.data_ptr()), C.c_void_p(st.mats.data_ptr()), _stream(dev)), "das3r_pose_matrices")
    pre = _lib.PreTransform()
    xyz, rot, sc, op = model._xyz.detach(), model._rotation.detach(), model._scaling.detach(), model._opacity.detach()
    pre.xyz, pre.rot, pre.scaling, pre.opacity_raw = xyz.data_ptr(), rot.data_ptr(), sc.data_ptr(), op.data_ptr()
    pre.conf_flat, pre.mask_index = conf.data_ptr(), None
    pre.R, pre.t, pre.Lq = st.mats.data_ptr(), st.mats.data_ptr() + 36, st.mats.data_ptr() + 48
    rs = _settings(view, model, pipe, background, 1.0, dev)
    I, image, radii, geom, binning, img, cap = _forward_full(rs, xyz, st.packed_sh(model), st.e, op, sc, rot, st.e, pre=pre)
    check_forward(cap, dev)   # (no backward pass will examine this forward's binning self-check)
    return image, radii


@torch.no_grad()
def render_set(model_path, name, iteration, views, model, pipe=PIPE, background=None, poses=None, write=True, fused=False):
    """render.py:72-86.  views: cameras carrying .pose7 (qw, qx, qy, qz, tx, ty, tz world-to-camera); poses: optional [N, 4, 4]
    world-to-camera matrices that override them.  fused: render_view_fused instead of the reference's PyTorch glue in front of the
    rasterizer (opt-in, like every fused form).  -> list of the rendered [3, H, W] tensors (on the device)."""
    dev = model.get_xyz.device
    background = background if background is not None else torch.zeros(3, device=dev)
    render_path = os.path.join(model_path, name, f"ours_{iteration}", "renders")
    out = []
    for idx, view in enumerate(views):
        pose = view.pose7 if poses is None else tensor_from_camera(poses[idx], dev)
        img = (render_view_fused(model, view, pose, background, pipe)[0] if fused
               else das3r_render(view, model, pipe, background, camera_pose=pose, variant="test")["render"])
        out.append(img)
        if write:
            save_image(img, os.path.join(render_path, f"{idx:05d}.png"))
    return out


def render_sets(model_path, seq, iteration=-1, sh_degree=3, white_background=False, optimised_poses=False, device="cuda", write=True, fused=False):
    """render.py:89-123: load the trained model, write pose_interpolated.npy, render the "interp" set.  seq: the sequence the model was
    trained on (its cameras).  -> (iteration, list of rendered images)"""
    model, iteration = load_trained_model(model_path, iteration, sh_degree, device)
    inter = save_interpolate_pose(model_path, iteration)
    bg = torch.tensor([1.0, 1.0, 1.0] if white_background else [0.0, 0.0, 0.0], dtype=torch.float32, device=device)
    views = sequence_cameras(seq, device)
    poses = None
    if optimised_poses:
        if inter is None:
            raise FileNotFoundError(f"--optimised-poses: no pose/pose_{iteration}.npy under {model_path}")
        if len(inter) != len(views):   # (a job trained with the held-out split saved the training views' poses only)
            from .train import split_sequence
            tr, _ = split_sequence(seq)
            if len(inter) != len(tr):
                raise ValueError(f"pose_{iteration}.npy holds {len(inter)} poses, the sequence {len(views)} frames ({len(tr)} training frames)")
            views = [views[i] for i in tr]
        poses = inter
    return iteration, render_set(model_path, "interp", iteration, views, model, PIPE, bg, poses=poses, write=write, fused=fused)


@torch.no_grad()
def forward_throughput(model, views, repeats=3, background=None, fused=False):
    """Eval-mode (torch.no_grad) forward alone: views per second and ms per view over `repeats` passes of all views, after one
    warm-up pass.  In this mode the rasterizer examines every forward's binning self-check before it returns (there is no backward
    to do it), so the figure contains that wait — the path render.py and every held-out report take."""
    dev = model.get_xyz.device
    background = background if background is not None else torch.zeros(3, device=dev)
    run = ((lambda: [render_view_fused(model, v, v.pose7, background)[0] for v in views]) if fused
           else (lambda: [das3r_render(v, model, PIPE, background, camera_pose=v.pose7, variant="test")["render"] for v in views]))
    run()
    torch.cuda.current_stream(dev).synchronize()
    t0 = time.perf_counter()
    for _ in range(repeats):
        run()
    torch.cuda.current_stream(dev).synchronize()
    dt = (time.perf_counter() - t0) / (repeats * len(views))
    return dict(ms_per_view=dt * 1e3, views_per_s=1.0 / dt, views=len(views), splats=int(model.get_xyz.shape[0]))


def main(argv=None):
    ap = argparse.ArgumentParser(description="Testing script parameters (render.py)")
    ap.add_argument("--model-path", "-m", "--model_path", dest="model_path", required=True)
    ap.add_argument("--source-path", "-s", "--source_path", dest="source_path", required=True, help="the preprocessed sequence directory the model was trained on")
    ap.add_argument("--iteration", type=int, default=-1)
    ap.add_argument("--sh-degree", "--sh_degree", dest="sh_degree", type=int, default=3)
    ap.add_argument("--white-background", "--white_background", dest="white_background", action="store_true")
    ap.add_argument("--optimised-poses", action="store_true", help="render from pose/pose_N.npy instead of the poses the cameras were loaded with")
    ap.add_argument("--dataset", default="sintel", choices=("sintel", "davis"))
    ap.add_argument("--fused", action="store_true", help="the pose pre-transform inside the rasterizer's kernels instead of the reference's PyTorch glue (render_view_fused)")
    args = ap.parse_args(argv)
    from .io_formats import load_sequence
    print("Rendering " + args.model_path)
    seq = load_sequence(args.source_path, device="cuda", dataset=args.dataset)
    it, imgs = render_sets(args.model_path, seq, args.iteration, args.sh_degree, args.white_background, args.optimised_poses, fused=args.fused)
    print(f"wrote {len(imgs)} images to {os.path.join(args.model_path, 'interp', f'ours_{it}', 'renders')}")


if __name__ == "__main__":
    main()
