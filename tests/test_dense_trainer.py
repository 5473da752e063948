"""CPU smoke test of the float64 dense trainer (oracle/dense_trainer.py, test infrastructure): on a tiny random model one
iteration runs, every parameter group receives a gradient of the right shape, the schedule bumps the SH degree at 3000, and the
Adam step moves the parameters by at most their learning rate (first step: |delta| = lr wherever the gradient is non-zero).
Its parity role — the independent implementation the HIP train step is compared with — is exercised on the GPU box
(tests/test_gpu_trainstep.py)."""
import math

import torch


def _tiny(frames=2, H=16, W=16, seed=0):
    from das3r_amd.camera import projection_matrix
    g = torch.Generator().manual_seed(seed)
    P = frames * H * W
    params = dict(xyz=torch.randn(P, 3, generator=g) * 0.6 + torch.tensor([0.0, 0.0, 4.0]), f_dc=torch.randn(P, 1, 3, generator=g),
                  f_rest=torch.zeros(P, 15, 3), opacity=torch.full((P, 1), -1.0), scaling=torch.randn(P, 3, generator=g) * 0.3 - 3.0,
                  rotation=torch.tensor([[1.0, 0, 0, 0]]).repeat(P, 1), conf_static=torch.rand(frames, H, W, generator=g) * 0.5 + 0.5,
                  Q=torch.tensor([[1.0, 0, 0, 0]]).repeat(frames, 1), T=torch.zeros(frames, 3), mask=torch.ones(P, dtype=torch.bool))
    fov = 2 * math.atan(0.5)
    cams = [dict(gt=torch.rand(3, H, W, generator=g), fovx=fov, fovy=fov, proj_T=projection_matrix(0.01, 100.0, fov, fov).t().contiguous())
            for _ in range(frames)]
    return params, cams


def test_one_iteration_of_the_dense_trainer():
    from oracle.dense_trainer import DenseTrainer
    params, cams = _tiny()
    tr = DenseTrainer(params, cams, iterations=100)
    before = {k: v.detach().clone() for k, v in tr.p.items()}
    loss, ps = tr.step(1, 0, torch.zeros(3, dtype=torch.float64))
    assert math.isfinite(loss) and math.isfinite(ps) and 0 < loss < 2
    lrs = {"xyz": 0.00016, "f_dc": 0.0025, "opacity": 0.05, "scaling": 0.005, "rotation": 0.001, "conf_static": 3e-3}
    for k, lr in lrs.items():
        d = (tr.p[k].detach() - before[k]).abs()
        assert float(d.max()) <= lr * 1.0001 and float(d.max()) > 0.5 * lr, k     # first Adam step: lr * sign(grad)
    assert float((tr.p["f_rest"].detach() - before["f_rest"]).abs().max()) == 0.0   # degree 0: no gradient reaches the rest
    assert tr.viewspace_grad.shape == (params["xyz"].shape[0], 3) and float(tr.viewspace_grad[:, 2].abs().max()) == 0.0
    assert tr.active_deg == 0
    tr.step(3000, 1, torch.zeros(3, dtype=torch.float64))
    assert tr.active_deg == 1
