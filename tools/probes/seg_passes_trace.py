"""Per forward of a workload: partition passes, segment sort time, whether the global depth sort ran (which binning path each of the first forwards of a shape takes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from das3r_amd import GaussianRasterizationSettings, _lib
from das3r_amd.rasterizer import _forward_full, _backward_impl
from das3r_amd.synth import make_workload
name = sys.argv[1] if len(sys.argv) > 1 else "dsc"
dev = torch.device("cuda:0")
sc = make_workload(name).to(dev)
rs = GaussianRasterizationSettings(**sc.settings_kwargs())
e = torch.empty(0, device=dev)
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 8):
    _lib.profile_report(); _lib.profile_enable(True)
    I, color, radii, geom, binning, img, cap = _forward_full(rs, sc.means3D, sc.shs, e, sc.opacities, sc.scales, sc.rotations, e)
    torch.cuda.synchronize(); _lib.profile_enable(False)
    k = _lib.profile_report()
    print(it, "passes", k.get("onesweep_pass_kernel", (0, 0))[0], "segsort ms", round(k.get("segment_sort_kernel", (0, 0.0))[1], 4), "scan_emit", round(k.get("scan_emit_kernel", (0, 0.0))[1], 4),
          "onesweep ms", round(k.get("onesweep_pass_kernel", (0, 0.0))[1], 4), "preprocess", round(k.get("preprocess_kernel", (0, 0.0))[1], 4), "radix" if "depth_hist_kernel" in k else "")
