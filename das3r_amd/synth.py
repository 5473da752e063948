"""Deterministic synthetic random-splat scenes (SURVEY.md §8d / BASELINE.md §2) used by bench.py and the tests.

All tensors are generated on the CPU with a seeded torch.Generator and uploaded by the caller, so the same
seed gives bit-identical inputs here and on the GPU box.
"""
import math
from dataclasses import dataclass, field

import torch

from .camera import projection_matrix

WORKLOADS = {
    # name: (P, W, H, focal, sh_degree, seed, s_px range, opacity mode)
    "c1": dict(P=10_000, W=256, H=256, focal=284.0, sh_degree=0, seed=1),
    "c2": dict(P=100_000, W=1920, H=1080, focal=1200.0, sh_degree=3, seed=2),
    "c4": dict(P=1_000_000, W=1920, H=1080, focal=1200.0, sh_degree=3, seed=4),
    "ds": dict(P=5_000_000, W=512, H=208, focal=600.0, sh_degree=0, seed=5, s_px=(0.3, 1.5), opacity=0.02),
    # ds with SPATIALLY COHERENT depth — what a real DAS3R scene looks like to a tile: every pixel of every frame is a Gaussian on
    # the scene's surfaces, so a 16x16 tile sees a thin depth band (here a smooth relief z(u, v) with 1 % of noise), not the whole
    # depth range as the random-depth `ds` does.  Same shape and count; the worst case for depth buckets that are global (round 4)
    "dsc": dict(P=5_000_000, W=512, H=208, focal=600.0, sh_degree=0, seed=5, s_px=(0.3, 1.5), opacity=0.02, coherent=0.01),
    # c4 with the splats crowded towards the top of the frame (density ~ 1 / sqrt(height)): what an image with a busy band does to a
    # tile order that hands every XCD one contiguous band (tools/gpu_perf.py --workloads c4s; not a bench line)
    "c4s": dict(P=1_000_000, W=1920, H=1080, focal=1200.0, sh_degree=3, seed=4, y_skew=2.0),
}


@dataclass
class Scene:
    W: int
    H: int
    tanfovx: float
    tanfovy: float
    sh_degree: int
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    campos: torch.Tensor
    bg: torch.Tensor
    means3D: torch.Tensor
    scales: torch.Tensor
    rotations: torch.Tensor
    opacities: torch.Tensor
    shs: torch.Tensor
    dL_dpix: torch.Tensor
    extra: dict = field(default_factory=dict)

    @property
    def P(self):
        return self.means3D.shape[0]

    def settings_kwargs(self):
        return dict(image_height=self.H, image_width=self.W, tanfovx=self.tanfovx, tanfovy=self.tanfovy, bg=self.bg,
                    scale_modifier=1.0, viewmatrix=self.viewmatrix, projmatrix=self.projmatrix,
                    sh_degree=self.sh_degree, campos=self.campos, prefiltered=False, debug=False)

    def to(self, device):
        kw = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in self.__dict__.items()}
        return Scene(**kw)


def _logu(g, n, lo, hi):
    return torch.exp(torch.rand(n, generator=g) * (math.log(hi) - math.log(lo)) + math.log(lo))


def make_scene(P, W, H, focal, sh_degree, seed, s_px=(0.5, 4.0), opacity=None, max_sh_degree=3, bg=(0.0, 0.0, 0.0), y_skew=1.0, coherent=None):
    """Camera at the origin looking down +z (DAS3R convention: viewmatrix = I, campos = 0,
    projmatrix = I @ P^T — /root/reference/gaussian_renderer/__init__.py:57-61)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    tanfovx, tanfovy = W / (2.0 * focal), H / (2.0 * focal)
    fovx, fovy = 2 * math.atan(tanfovx), 2 * math.atan(tanfovy)
    view = torch.eye(4)
    proj = view @ projection_matrix(0.01, 100.0, fovx, fovy).transpose(0, 1)
    z = torch.rand(P, generator=g) * 9.0 + 1.0
    u, v = torch.rand(P, generator=g) * 2.2 - 1.1, torch.rand(P, generator=g) ** y_skew * 2.2 - 1.1   # (y_skew > 1: crowded towards the top)
    if coherent is not None:   # depth = a smooth relief over the image + relative noise `coherent`
        z = (4.0 + 2.0 * torch.sin(3.0 * u) * torch.cos(2.0 * v) + 1.5 * torch.cos(5.0 * u + 1.0)) * (1.0 + coherent * torch.randn(P, generator=g))
        z = z.clamp_min(0.5)
    x = z * tanfovx * u
    y = z * tanfovy * v
    means3D = torch.stack([x, y, z], 1).contiguous()
    spx = _logu(g, P, *s_px)
    aniso = torch.stack([_logu(g, P, 0.5, 2.0) for _ in range(3)], 1)
    scales = (spx[:, None] * aniso * z[:, None] / focal).contiguous()
    q = torch.randn(P, 4, generator=g)
    q = q / q.norm(dim=1, keepdim=True) * (0.9 + 0.2 * torch.rand(P, 1, generator=g))
    if opacity is None:
        op = torch.sigmoid(torch.randn(P, 1, generator=g) * 2.0)
    else:
        op = torch.full((P, 1), float(opacity))
    M = (max_sh_degree + 1) ** 2
    shs = torch.randn(P, M, 3, generator=g) * 0.1
    shs[:, 0, :] = torch.rand(P, 3, generator=g) * 3.54 - 1.77
    dL = torch.randn(3, H, W, generator=g) / float(W * H)
    return Scene(W=W, H=H, tanfovx=tanfovx, tanfovy=tanfovy, sh_degree=sh_degree, viewmatrix=view, projmatrix=proj.contiguous(),
                 campos=torch.zeros(3), bg=torch.tensor(bg, dtype=torch.float32), means3D=means3D, scales=scales,
                 rotations=q.contiguous(), opacities=op.contiguous(), shs=shs.contiguous(), dL_dpix=dL.contiguous())


def make_workload(name, P=None):
    cfg = dict(WORKLOADS[name])
    if P is not None:
        cfg["P"] = int(P)
    return make_scene(**cfg)
