"""(applies with r06_segment_long_kernel.patch: DAS3R_SEG_LONG exists only there)  A-B of the segmented binning path's answers to long
segments on the DAS3R-shaped train step: one more partition pass (round 4, shipped) against the long-segment kernel with two passes
(round 6, measured slower, not kept: docs/ledger.md (bn), profiles/r06_seg_long_kernel_ab.txt).   python docs/patches/r06_seg_levels_ab.py [smooth,noise]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
dev = torch.device('cuda:0'); torch.cuda.set_device(0)
rk = bench.Ranks(bench.parse_args(['--gpus', '1']))
from das3r_amd import _lib
BIN = ("onesweep_pass_kernel", "scan_emit_kernel", "segment_sort_kernel", "segment_long_kernel", "tile_ranges_kernel", "depth_hist_kernel")
for depth in (sys.argv[1] if len(sys.argv) > 1 else "smooth,noise").split(","):
    for name, env in (("learnt", {}), ("two passes + long kernel", {"DAS3R_BINNING": "seg", "DAS3R_SEG_LONG": "1"}), ("three passes", {"DAS3R_BINNING": "seg3", "DAS3R_SEG_LONG": "0"}),
                      ("two passes, short kernel alone", {"DAS3R_BINNING": "seg", "DAS3R_SEG_LONG": "0"})):
        for k in ("DAS3R_BINNING", "DAS3R_SEG_LONG"):
            os.environ.pop(k, None)
        os.environ.update(env)
        _lib.reload_switches(); _lib.forget_shapes()
        step, n = bench.train_step_timer(dev, fused=True, depth=depth)
        for _ in range(30): step()
        t = rk.timed(step, 100, 10) / 100 * 1e3
        _lib.profile_enable(True)
        for _ in range(20): step()
        torch.cuda.synchronize(); rep = _lib.profile_report(); _lib.profile_enable(False)
        b = {k: (rep[k][0] // 20, round(rep[k][1] / 20, 4)) for k in BIN if k in rep}
        print(f"{depth:8s} {name:32s} train step {t:.4f} ms  binning {sum(v[1] for v in b.values()):.4f} ms  {b}", flush=True)
