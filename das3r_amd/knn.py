"""distCUDA2 — host-side mirror of `simple_knn._C.distCUDA2` (reference call sites
/root/reference/scene/gaussian_model.py:213,641): float32 (P,3) device tensor -> float32 (P,) mean squared distance to
the 3 nearest other points.  Hand-written HIP behind the C-ABI (include/das3r_raster.h das3r_knn3_mean_dist2)."""
import ctypes as C

import torch

from . import _lib


def distCUDA2(points):
    lib = _lib.load()
    if not isinstance(points, torch.Tensor) or points.dim() != 2 or points.shape[1] != 3:
        raise RuntimeError("points must have dimensions (num_points, 3)")
    if points.device.type != "cuda":
        raise RuntimeError("distCUDA2: points must live on a HIP device (torch device 'cuda'); there is no CPU path")
    pts = points.detach().contiguous().float()
    P = pts.shape[0]
    out = torch.empty(P, dtype=torch.float32, device=pts.device)
    if P == 0:
        return out
    ws = torch.empty(int(lib.das3r_knn3_workspace_bytes(P)), dtype=torch.uint8, device=pts.device)
    with torch.cuda.device(pts.device):
        rc = lib.das3r_knn3_mean_dist2(P, C.c_void_p(pts.data_ptr()), C.c_void_p(out.data_ptr()), C.c_void_p(ws.data_ptr()),
                                       C.c_void_p(torch.cuda.current_stream(pts.device).cuda_stream))
    _lib.check(rc, "das3r_knn3_mean_dist2")
    return out
