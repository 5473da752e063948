#!/usr/bin/env python3
"""Lockstep comparison of the two forms of the fused iteration (das3r_amd/fast_step.py vs the autograd form) on a sequence that went
through the on-disk formats, as the farm's jobs do: parameters after every step, then the held-out pass.
    python tools/fast_vs_autograd.py [--frames 22 --steps 60]"""
import argparse
import os
import random
import shutil
import sys
import tempfile
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

PIPE = SimpleNamespace(debug=False, compute_cov3D_python=False, convert_SHs_python=False)
NAMES = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation", "_conf_static", "Q", "T")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=22)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--memory", action="store_true", help="skip the trip through the on-disk formats")
    a = ap.parse_args()
    from das3r_amd import io_formats as io
    from das3r_amd.model import OptimParams
    from das3r_amd.train import build_from_sequence, psnr_report, synthetic_sequence, test_pose_pass, train_step
    seq = synthetic_sequence(frames=a.frames, W=512, H=208, focal=1.2 * 512, n_splats=60000, seed=11, device="cuda:0")
    work = tempfile.mkdtemp(prefix="das3r_fva_")
    try:
        if not a.memory:
            d = os.path.join(work, "seq")
            io.write_sequence_dir(seq, d)
            seq = io.load_sequence(d, device=torch.device("cuda:0"))
        models = []
        for direct in (True, False):
            import copy
            model, cams, test = build_from_sequence(copy.deepcopy({k: v for k, v in seq.items()}), heldout=True)
            opt = OptimParams(iterations=4000)
            model.training_setup(opt, fused=True)
            model.fast_step = direct
            models.append((model, cams, test, opt))
        for k in ("images", "depths", "confs"):
            print(k, seq[k].dtype, tuple(seq[k].shape), seq[k].is_contiguous())
        c0 = models[0][1][0]
        print("cam image", c0.original_image.dtype, c0.original_image.is_contiguous(), "proj", c0.projection_matrix.dtype, c0.projection_matrix.is_contiguous(),
              "FoV", c0.FoVx, c0.FoVy, "Q", models[0][0].Q.dtype, models[0][0].Q.is_contiguous(), "conf", models[0][0]._conf_static.dtype,
              "mask all", bool(models[0][0].aggregated_mask.all()))
        bg = torch.zeros(3, device="cuda")
        rng = random.Random(0)
        stack = []
        for it in range(1, a.steps + 1):
            if not stack:
                stack = list(range(len(models[0][1])))
            uid = stack.pop(rng.randint(0, len(stack) - 1))
            res = [train_step(m, c[uid], o, it, PIPE, bg, fused=True) for m, c, t, o in models]
            if not stack:
                for m, c, t, o in models:
                    test_pose_pass(m, t, None, o, PIPE, bg, random.Random(it), fused=True)
            worst = max((float((getattr(models[0][0], n) - getattr(models[1][0], n)).abs().max() / (getattr(models[1][0], n).abs().max() + 1e-12)), n) for n in NAMES)
            if it <= 5 or it % 10 == 0 or not stack:
                print(f"it {it} uid {uid} loss {float(res[0][0]):.6f} {float(res[1][0]):.6f} psnr {float(res[0][1]):.4f} {float(res[1][1]):.4f} worst rel param diff {worst[0]:.3e} ({worst[1]})")
        for m, c, t, o in models:
            print("heldout", psnr_report(m, t, test_poses=True)["psnr"])
    finally:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
