"""Host-side pieces of the round-6 additions (no GPU): image quantisation, pose tensor <-> matrix, checkpoint discovery, and the
FusedAdam state_dict in torch.optim.Adam's layout."""
import os

import numpy as np
import torch


def test_save_image_quantises_like_torchvision(tmp_path):
    from PIL import Image
    from das3r_amd.offline import save_image
    x = torch.tensor([-0.2, 0.0, 0.001, 0.00196, 0.5, 0.998, 0.999, 1.0, 1.7]).repeat(3, 2, 1)   # [3, 2, 9]
    arr = save_image(x, str(tmp_path / "a" / "x.png"))
    assert arr.dtype == np.uint8 and arr.shape == (2, 9, 3)
    assert arr[0, :, 0].tolist() == [0, 0, 0, 0, 128, 254, 255, 255, 255]      # floor(255 x + 0.5), clamped
    assert np.array_equal(np.asarray(Image.open(tmp_path / "a" / "x.png")), arr)


def test_tensor_from_camera_inverts_camera_from_tensor():
    from das3r_amd.camera import camera_from_tensor
    from das3r_amd.offline import tensor_from_camera
    g = torch.Generator().manual_seed(3)
    for _ in range(20):
        q = torch.nn.functional.normalize(torch.randn(4, generator=g), dim=0)
        pose = torch.cat([q, torch.randn(3, generator=g)])
        back = tensor_from_camera(camera_from_tensor(pose), "cpu")
        if back[0] * pose[0] < 0:
            back = torch.cat([-back[:4], back[4:]])
        assert torch.allclose(back, pose, atol=2e-6)


def test_latest_checkpoint_wants_the_complete_pair(tmp_path):
    from das3r_amd.train import latest_checkpoint
    assert latest_checkpoint(str(tmp_path)) == (None, 0) and latest_checkpoint(None) == (None, 0)
    for n in ("chkpnt30.pth", "chkpnt30.das3r.pth", "chkpnt60.pth", "chkpnt60.das3r.pth", "chkpnt90.pth", "chkpnt7.pth.tmp", "other.pth"):
        (tmp_path / n).write_bytes(b"")
    assert latest_checkpoint(str(tmp_path)) == (os.path.join(str(tmp_path), "chkpnt60.pth"), 60)   # 90 lacks its extras: torn by a kill


def test_search_for_max_iteration(tmp_path):
    from das3r_amd.offline import search_for_max_iteration
    for n in ("iteration_7", "iteration_4000", "iteration_300", "junk"):
        os.makedirs(tmp_path / n)
    assert search_for_max_iteration(str(tmp_path)) == 4000


def test_fused_adam_state_dict_is_torch_adams():
    """No step is taken here (the kernels need a GPU): the state is planted by hand, written out, and read by torch.optim.Adam and back."""
    from das3r_amd.fused import FusedAdam
    a, b = torch.nn.Parameter(torch.randn(5, 3)), torch.nn.Parameter(torch.randn(7))
    fa = FusedAdam([{"params": [a], "lr": 1e-3, "name": "xyz"}, {"params": [b], "lr": 2e-3, "name": "opacity"}], lr=0.0, eps=1e-15)
    fa.state[a] = dict(step=12, exp_avg=torch.randn(5, 3), exp_avg_sq=torch.rand(5, 3))
    fa.set_active_sh_degree(1)
    sd = fa.state_dict()
    assert set(sd["state"]) == {0} and sd["param_groups"][1]["params"] == [1] and sd["param_groups"][0]["name"] == "xyz"
    ta = torch.optim.Adam([{"params": [a], "lr": 1e-3, "name": "xyz"}, {"params": [b], "lr": 2e-3, "name": "opacity"}], lr=0.0, eps=1e-15)
    ta.load_state_dict(sd)
    assert int(ta.state[a]["step"]) == 12 and torch.equal(ta.state[a]["exp_avg"], fa.state[a]["exp_avg"]) and b not in ta.state
    fb = FusedAdam([{"params": [a], "lr": 0.0, "name": "xyz"}, {"params": [b], "lr": 0.0, "name": "opacity"}], lr=0.0, eps=1e-15)
    fb.load_state_dict(ta.state_dict())
    assert fb.state[a]["step"] == 12 and torch.equal(fb.state[a]["exp_avg_sq"], fa.state[a]["exp_avg_sq"]) and fb.param_groups[0]["lr"] == 1e-3
    fb.load_state_dict(sd)
    assert fb.active_sh_degree == 1
