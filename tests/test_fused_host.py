"""Host logic of the opt-in fused path that needs no GPU: what FusedAdam and the direct iteration refuse, and the bookkeeping they
do before any kernel is launched (das3r_amd/fused.py, das3r_amd/fast_step.py).  The kernels themselves: tests/test_gpu_fused.py."""
from types import SimpleNamespace

import pytest
import torch

from das3r_amd.fused import FusedAdam


def _opt(*tensors, lr=1e-3):
    return FusedAdam([dict(params=[t], lr=lr, name=f"g{i}") for i, t in enumerate(tensors)], lr=0.0, eps=1e-15)


def test_adam_slots_refuses_what_it_cannot_step():
    """FusedAdam.adam_slots hands a kernel the Adam state of parameters whose step that kernel takes itself
    (das3r_pretransform_backward_adam): only device tensors of this optimizer that do not already carry a gradient."""
    a, b = torch.zeros(5, 3), torch.zeros(5, 4)
    opt = _opt(a)
    with pytest.raises(RuntimeError, match="not a parameter"):
        opt.adam_slots([b])
    with pytest.raises(RuntimeError, match="HIP device"):      # a CPU tensor: there is no CPU path
        opt.adam_slots([a])
    a.grad = torch.zeros_like(a)
    with pytest.raises(RuntimeError, match="already carries a gradient"):
        opt.adam_slots([a])
    assert opt.state == {}                                       # nothing was counted for a refused call


def test_step_passes_parameters_without_gradient_by_and_counts_nothing_for_them():
    """What lets a fused kernel take the step of some parameters: FusedAdam.step() treats a parameter without gradient as
    torch.optim.Adam does — no update, no step count (the count of an "sh_rest" group is the documented exception)."""
    a, rest = torch.zeros(4, 3), torch.zeros(4, 15, 3)
    opt = FusedAdam([dict(params=[a], lr=1e-3), dict(params=[rest], lr=1e-3, sh_rest=True)], lr=0.0, eps=1e-15)
    opt.set_active_sh_degree(0)
    opt.step()                                                   # no gradients anywhere: no launch, hence no device needed
    assert a not in opt.state and opt.state[rest]["step"] == 1
    assert opt.handles_compact_sh(rest) and not opt.handles_compact_sh(a)


def test_direct_iteration_is_only_taken_with_fused_optimizers_and_the_default_pipe(monkeypatch):
    from das3r_amd import fast_step
    pipe = SimpleNamespace(debug=False, compute_cov3D_python=False, convert_SHs_python=False)
    plain = SimpleNamespace(optimizer=torch.optim.Adam([torch.zeros(1, requires_grad=True)]), optimizer_cam=None)
    assert not fast_step.available(plain, pipe)                  # torch optimizers: the autograd form
    fused = SimpleNamespace(optimizer=_opt(torch.zeros(1)), optimizer_cam=_opt(torch.zeros(1)), fast_step=False)
    assert not fast_step.available(fused, pipe)                  # switched off on the model
    monkeypatch.setenv("DAS3R_FAST_STEP", "0")
    fused.fast_step = True
    assert not fast_step.available(fused, pipe)                  # switched off in the environment


def test_fast_step_settings_are_keyed_on_values_not_on_the_camera_id():
    """ADVICE r4 (fast_step.py _settings): the cache was keyed on id(cam) without keeping the camera alive — CPython hands a freed
    object's id to the next allocation, so a camera made later with another size / FoV got the stale settings.  Now: values + the two
    tensors the settings are built from (kept alive by the entry)."""
    from types import SimpleNamespace

    from das3r_amd import fast_step
    st = SimpleNamespace(dev=torch.device("cpu"), settings={})
    model = SimpleNamespace(active_sh_degree=0)
    bg = torch.zeros(3)
    mk = lambda W, H, f: SimpleNamespace(image_height=H, image_width=W, FoVx=f, FoVy=f, projection_matrix=torch.eye(4) * f)
    a, b = mk(64, 32, 0.5), mk(64, 32, 0.5)
    ra, rb = fast_step._settings(st, a, model, bg), fast_step._settings(st, b, model, bg)
    assert fast_step._settings(st, a, model, bg) is ra and fast_step._settings(st, b, model, bg) is rb and ra is not rb
    for _ in range(64):   # cameras made per iteration: whatever id they get, the settings are theirs
        del a
        a = mk(128, 48, 0.7)
        rs = fast_step._settings(st, a, model, bg)
        assert (rs.image_width, rs.image_height) == (128, 48) and abs(rs.tanfovx - __import__("math").tan(0.35)) < 1e-12
        del a
        a = mk(64, 32, 0.5)
        rs = fast_step._settings(st, a, model, bg)
        assert (rs.image_width, rs.image_height) == (64, 32)
    a.projection_matrix.mul_(2.0)   # written in place: rebuilt
    r2 = fast_step._settings(st, a, model, bg)
    assert r2 is not rs and torch.equal(r2.projmatrix, a.projection_matrix)
    model.active_sh_degree = 1
    assert fast_step._settings(st, a, model, bg).sh_degree == 1
