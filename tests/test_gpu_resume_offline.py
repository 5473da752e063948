"""Round 6 (VERDICT r5 missing #4 - #5): resume of a killed sequence job from chkpnt<iteration>.pth, and the offline renderer.
Counterparts: /root/reference/train_gui.py:626-628 + scene/gaussian_model.py:66-101 (checkpoint / restore), /root/reference/render.py:72-123."""
import os
import shutil

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SMALL = dict(frames=12, W=256, H=104, focal=300.0, n_splats=8000)
NAMES = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation", "_conf_static", "Q", "T")


@pytest.mark.parametrize("fused", [True, False])
def test_resumed_job_ends_where_the_uninterrupted_job_ends(tmp_path, fused):
    """A job of 90 iterations that checkpoints every 30 (its out directory then holds chkpnt30 and chkpnt60), and the same job "killed"
    after iteration 60 and resumed: a second process-like start (fresh model object, fresh optimizers) with resume=True finds chkpnt60,
    restores parameters, both optimizers' moments and step counts, the held-out poses, the camera stack and the generator that draws
    from it, runs iterations 61 .. 90 — and ends with the SAME parameters: bit for bit with the fused direct iteration (which is
    bit-reproducible, round 5), to fp32 noise with torch.optim.Adam around the autograd surface (its backward adds in varying order)."""
    from das3r_amd.farm import run_sequence_job
    from das3r_amd.train import consistent_sequence, latest_checkpoint
    dev = torch.device("cuda:0")
    seq = consistent_sequence(seed=5, **SMALL)
    full_dir, res_dir = str(tmp_path / "full"), str(tmp_path / "resumed")
    keep_full, keep_res = {}, {}
    full = run_sequence_job(3, 90, dev, fused=fused, seq=seq, out_dir=full_dir, checkpoint_every=30, keep=keep_full)
    assert full["ok"] == 1 and latest_checkpoint(full_dir)[1] == 60
    # the killed job: its directory as it was right after iteration 60 (no final PLY, no report)
    os.makedirs(res_dir)
    for f in ("chkpnt60.pth", "chkpnt60.das3r.pth"):
        shutil.copy(os.path.join(full_dir, f), os.path.join(res_dir, f))
    res = run_sequence_job(3, 90, dev, fused=fused, seq=seq, out_dir=res_dir, checkpoint_every=30, resume=True, keep=keep_res)
    assert res["ok"] == 1
    a, b = keep_full[3][0], keep_res[3][0]
    assert a is not b
    for n in NAMES:
        x, y = getattr(a, n).detach(), getattr(b, n).detach()
        if fused:
            assert torch.equal(x, y), f"{n}: a resumed job must end bit-identical to the uninterrupted one"
        else:
            assert torch.allclose(x, y, rtol=1e-4, atol=1e-6 * float(x.abs().max())), n
    if fused:
        assert res["psnr"] == full["psnr"]
    else:
        assert abs(res["psnr"] - full["psnr"]) < 0.02
    # moments and step counts came along (not re-started): compare one optimizer state
    sa, sb = a.optimizer.state_dict(), b.optimizer.state_dict()
    assert [int(v["step"]) for v in sa["state"].values()] == [int(v["step"]) for v in sb["state"].values()] and len(sa["state"]) > 0
    # a job that finished is not resumed into: nothing to do but the report
    again = run_sequence_job(3, 60, dev, fused=fused, seq=seq, out_dir=res_dir, resume=True)
    assert again["ok"] == 1   # (latest checkpoint is at 60 = iterations: the job starts over rather than "continue" past its end)


def test_checkpoint_loads_on_the_other_optimizer(tmp_path):
    """The file is the reference's: (capture(), iteration) with torch.optim.Adam's state_dict layout — a checkpoint written with the
    fused optimizer restores into the plain one and back, moments and step counts intact."""
    from das3r_amd.model import OptimParams
    from das3r_amd.train import build_from_sequence, consistent_sequence, load_checkpoint, save_checkpoint, train
    seq = consistent_sequence(seed=6, **SMALL)
    model, cams, test = build_from_sequence(seq, heldout=True)
    opt = OptimParams(iterations=40)
    model.training_setup(opt, fused=True)
    train(model, cams, opt, 20, fused=True, test_cameras=test)
    path = str(tmp_path / "chkpnt20.pth")
    save_checkpoint(path, model, 20, None)
    capture, it = torch.load(path, weights_only=False)
    assert it == 20 and len(capture) == 14, "the reference's restore() unpacks fourteen fields"
    plain, _, _ = build_from_sequence(seq, heldout=True)
    assert load_checkpoint(path, plain, opt, fused=False)[0] == 20
    assert isinstance(plain.optimizer, torch.optim.Adam)
    st_f, st_p = model.optimizer.state_dict()["state"], plain.optimizer.state_dict()["state"]
    assert set(st_f) == set(st_p)
    for k in st_f:
        assert int(st_f[k]["step"]) == int(st_p[k]["step"]) == 20
        assert torch.equal(st_f[k]["exp_avg"], st_p[k]["exp_avg"].to(st_f[k]["exp_avg"].device))
    assert torch.equal(plain._xyz, model._xyz) and torch.equal(plain._conf_static, model._conf_static)
    train(plain, cams, opt, 25, start_iteration=21, test_cameras=test)   # and it trains on


def test_offline_render_writes_what_a_direct_render_gives(tmp_path):
    """render.py:89-123: a finished job's directory (point_cloud/iteration_N/point_cloud.ply + pose/pose_N.npy) -> PNGs of every frame
    under interp/ours_N/renders, pose/pose_interpolated.npy; the PNG bytes are torchvision.utils.save_image's quantisation of the very
    image a direct render_test call on the in-memory model gives (the PLY round trip is exact: float32 columns)."""
    from PIL import Image
    from das3r_amd import offline
    from das3r_amd.farm import run_sequence_job
    from das3r_amd.render import das3r_render
    from das3r_amd.train import consistent_sequence
    dev = torch.device("cuda:0")
    seq = consistent_sequence(seed=7, **SMALL)
    out = str(tmp_path / "job")
    keep = {}
    assert run_sequence_job(0, 40, dev, fused=True, seq=seq, out_dir=out, keep=keep)["ok"] == 1
    it, imgs = offline.render_sets(out, seq, iteration=-1)
    assert it == 40 and len(imgs) == SMALL["frames"]
    files = sorted(os.listdir(os.path.join(out, "interp", "ours_40", "renders")))
    assert files == [f"{i:05d}.png" for i in range(SMALL["frames"])]
    inter = np.load(os.path.join(out, "pose", "pose_interpolated.npy"))
    assert inter.shape == (11, 4, 4) and np.array_equal(inter, np.load(os.path.join(out, "pose", "pose_40.npy")).astype(inter.dtype))
    # the in-memory model, rendered the way the loaded one is: per-Gaussian conf_static column, opacity as stored, full SH degree
    model = keep[0][0]
    views = offline.sequence_cameras(seq, dev)
    loaded, _ = offline.load_trained_model(out, 40)
    assert torch.equal(loaded._xyz, model._xyz.detach()) and torch.equal(loaded._opacity, model._opacity.detach())
    assert torch.equal(loaded._conf_static, model._conf_static.detach().reshape(-1, 1)[model.aggregated_mask])
    with torch.no_grad():
        direct = das3r_render(views[4], loaded, offline.PIPE, torch.zeros(3, device=dev), camera_pose=views[4].pose7, variant="test")["render"]
    assert torch.equal(direct, imgs[4])
    png = np.asarray(Image.open(os.path.join(out, "interp", "ours_40", "renders", "00004.png")))
    want = direct.mul(255).add(0.5).clamp(0, 255).permute(1, 2, 0).to(torch.uint8).cpu().numpy()
    assert png.shape == (SMALL["H"], SMALL["W"], 3) and np.array_equal(png, want)
    assert float(direct.mean()) > 0.02, "the render is not empty"
    # the optimised poses (the training views only: the job held frame 5 out)
    it2, imgs2 = offline.render_sets(out, seq, iteration=40, optimised_poses=True, write=False)
    assert len(imgs2) == 11 and all(tuple(i.shape) == (3, SMALL["H"], SMALL["W"]) for i in imgs2)
    tp = offline.forward_throughput(loaded, views[:4], repeats=1)
    assert tp["views"] == 4 and tp["ms_per_view"] > 0
    # round 6: the fused form of the same render (the pose pre-transform inside the rasterizer's kernels) — the glue form's image within
    # the parity bars, the same Gaussians visible (a radius may move by one where the pre-transform's rounding crosses a ceil)
    from tests import util
    fused_img, fused_radii = offline.render_view_fused(loaded, views[4], views[4].pose7, torch.zeros(3, device=dev))
    util.assert_color_close(fused_img.cpu().numpy(), direct.cpu().numpy(), "render_view_fused vs render_test")
    with torch.no_grad():
        glue_radii = das3r_render(views[4], loaded, offline.PIPE, torch.zeros(3, device=dev), camera_pose=views[4].pose7, variant="test")["radii"]
    assert float(((fused_radii > 0) != (glue_radii > 0)).float().mean()) < 1e-4 and float((fused_radii - glue_radii).abs().max()) <= 1
    it3, imgs3 = offline.render_sets(out, seq, iteration=40, write=False, fused=True)
    util.assert_color_close(imgs3[4].cpu().numpy(), direct.cpu().numpy(), "render_sets(fused=True)")
    assert offline.forward_throughput(loaded, views[:4], repeats=1, fused=True)["views"] == 4
