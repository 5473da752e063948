"""Densification stress for BASELINE.json configs[3] ("1M-splat synthetic scene, 1080p, densification on").

The reference never densifies: both training loops have the block commented out and `densify_and_prune` no longer calls
clone / split (/root/reference/train_gui.py:612-623, scene/gaussian_model.py:556-557), so P is constant
in real DAS3R training.  "Densification on" is therefore a synthetic stress of the rasterizer's buffer regrowth (SURVEY.md
§8d): every `every` steps the 5 % of the splats with the largest view-space gradient grow the scene — the smaller half of them
is cloned, the larger half split in two (what densify_and_clone / densify_and_split did, gaussian_model.py:511-551) — until P
has grown from P0 to 1.3 P0.  Each event changes P, so the library's shape cache resets, the binning buffer is sized
exactly again and the speculative capacity starts over (api.hip).

Works on the tensors the rasterizer takes (activated scales, raw quaternions); harness code, not part of the drop-in surface.
"""
import torch


def _rotation_matrices(q):
    q = q / q.norm(dim=1, keepdim=True)
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).reshape(-1, 3, 3)


def grow_top_gradient(tensors, viewspace_grad, frac=0.05, generator=None):
    """tensors: dict means3D [P,3], scales [P,3], rotations [P,4], opacities [P,1], shs [P,M,3] (same device).
    viewspace_grad: means2D.grad [P,3].  Returns a new dict with P' = P + round(frac * P) splats: of the `frac` splats with the
    largest |grad[:, :2]|, those whose largest scale is at most the selection's median are cloned (copy appended), the others
    are replaced by two samples of their own Gaussian with scales / 1.6."""
    P = tensors["means3D"].shape[0]
    k = max(1, int(round(frac * P)))
    score = viewspace_grad[:, :2].norm(dim=1)
    sel = torch.topk(score, k).indices
    size = tensors["scales"][sel].max(dim=1).values
    small = size <= size.median()
    clone_idx, split_idx = sel[small], sel[~small]
    out = {}
    keep = torch.ones(P, dtype=torch.bool, device=sel.device)
    keep[split_idx] = False
    ns = split_idx.numel()
    stds = tensors["scales"][split_idx].repeat(2, 1)
    noise = torch.randn(stds.shape, generator=generator, device="cpu").to(stds.device) * stds
    R = _rotation_matrices(tensors["rotations"][split_idx]).repeat(2, 1, 1)
    new_xyz = torch.bmm(R, noise.unsqueeze(-1)).squeeze(-1) + tensors["means3D"][split_idx].repeat(2, 1)
    for name, t in tensors.items():
        parts = [t[keep], t[clone_idx]]
        if name == "means3D":
            parts.append(new_xyz)
        elif name == "scales":
            parts.append(stds / 1.6)
        else:
            parts.append(t[split_idx].repeat(2, *([1] * (t.dim() - 1))))
        out[name] = torch.cat(parts, 0).contiguous()
    assert out["means3D"].shape[0] == P + k and ns + clone_idx.numel() == k
    return out


class GrowthSchedule:
    """P0 -> target (1.3 P0) in +5 % steps, one every `every` steps."""

    def __init__(self, P0, every=100, frac=0.05, target=1.3):
        self.P0, self.every, self.frac, self.limit = int(P0), int(every), float(frac), int(round(target * P0))

    def due(self, step, P):
        return step > 0 and step % self.every == 0 and P < self.limit

    def frac_for(self, P):
        return min(self.frac, (self.limit - P) / P)
