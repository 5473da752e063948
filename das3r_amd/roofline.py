"""Algorithmic (compulsory) HBM bytes of the rasterizer hot path — the figures bench.py's `roofline` object uses.

Model = SURVEY.md §8(d): every input read once, every API-visible output written once, the instance list written once
and read once (12 B/instance each way), splat attributes fetched once per instance (36 B), sort passes / scratch
re-reads / atomics excluded.  With P splats, D active SH degree, M stored SH coeffs, I = num_rendered, Npix = W*H:

    B_fwd = P*(44 + 12(D+1)^2) + 4P + 24 I + 36 I + 12 Npix
    B_bwd = 20 Npix + 40 I + P*(44 + 12(D+1)^2) + 56 P + 12 M P

Per-kernel shares (they sum to B_fwd + B_bwd):
    preprocess_kernel            P*(44 + 12(D+1)^2) + 4P          inputs + radii
    binning (sort/scan/emit/...) 12 I                             instance list written once
    render_forward_kernel        12 I + 36 I + 12 Npix            list read + per-instance splat fetch + image
    render_backward_kernel       20 Npix + 40 I                   dL_dpix, final_T, n_contrib + list + splat fetch
    preprocess_backward_kernel   P*(44 + 12(D+1)^2) + 56 P + 12 M P   inputs again + per-splat grads + dense dL_dsh
"""

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md (≈6.3 TB/s achievable)

BINNING_KERNELS = ("radix_hist_kernel", "radix_rowscan_kernel", "radix_scatter_kernel", "depth_hist_kernel", "onesweep_pass_kernel",
                   "scan_emit_kernel", "emit_kernel", "tile_ranges_kernel", "segment_sort_kernel")


def algorithmic_bytes(P, D, M, I, W, H):
    npix = W * H
    inp = P * (44 + 12 * (D + 1) ** 2)
    per_kernel = {
        "preprocess_kernel": inp + 4 * P,
        "binning": 12 * I,
        "render_forward_kernel": 48 * I + 12 * npix,
        "render_backward_kernel": 20 * npix + 40 * I,
        "preprocess_backward_kernel": inp + 56 * P + 12 * M * P,
    }
    b_fwd = inp + 4 * P + 60 * I + 12 * npix
    b_bwd = 20 * npix + 40 * I + inp + 56 * P + 12 * M * P
    assert sum(per_kernel.values()) == b_fwd + b_bwd
    return per_kernel, b_fwd, b_bwd


ALIASES = {"render_forward_rows_kernel": "render_forward_kernel",     # several implementations of each compositing stage
           "render_forward_lanes_kernel": "render_forward_kernel",
           "render_backward_mfma_kernel": "render_backward_kernel",    # (render_rows.hip, render_bwd_mfma.hip,
           "render_backward_scan_kernel": "render_backward_kernel",    #  render_bwd_scan.hip, render_bwd_blk.hip)
           "render_backward_blk_kernel": "render_backward_kernel",
           "render_forward_regions_kernel": "render_forward_kernel",   # round 6: 2x2 regions (render_regions.hip, render_bwd_rgn.hip)
           "render_forward_slices_kernel": "render_forward_kernel",
           "render_backward_regions_kernel": "render_backward_kernel"}


def group_kernel_times(report):
    """{kernel: (launches, total_ms)} -> same with the binning kernels folded into one 'binning' entry."""
    out = {}
    for name, (n, ms) in report.items():
        key = "binning" if name in BINNING_KERNELS else ALIASES.get(name, name)
        c, t = out.get(key, (0, 0.0))
        out[key] = (c + n, t + ms)
    return out


# ---- the VALU roofline of the compositing kernels -------------------------------------------------------------
# The compositing kernels are bound by vector-ALU issue, not by HBM: what bounds them from below is the walk of
# the LIVE (pixel, splat) pairs — the pairs upstream blends (position < n_contrib, power <= 0, alpha >= 1/255), counted by
# das3r_raster_count_live_pairs — at the instruction count of the kernels' own pair loops, 64 pairs per wave instruction, every
# instruction at the full rate of 2 cycles per wave64 instruction per SIMD (MI355X_MICROARCH.md: SIMD-32 lanes; half-rate DPP / compare
# and quarter-rate transcendental instructions make the real walk slower than this bound):
#     t_valu = live_pairs / 64 * WALK_VALU_PER_64_PAIRS * 2 cycles / (1024 SIMDs * 2.4 GHz)
# Instruction counts of the shipped pair loops (llvm-objdump of the walk bodies): forward 34 VALU + 1 v_exp per 64
# pairs (render_rows.hip), backward 48 VALU + 2 transcendentals (render_blk.h block_row).
WALK_VALU_PER_64_PAIRS = {"render_forward_kernel": 35, "render_backward_kernel": 50}
SIMDS, CLOCK_HZ, CYCLES_PER_VALU = 1024, 2.4e9, 2.0


def valu_roofline(kernel, live_pairs, ms):
    """-> dict(ideal_valu_insts, ideal_ms, frac): the walk of the live pairs alone against the kernel's measured time."""
    n = WALK_VALU_PER_64_PAIRS[kernel]
    insts = live_pairs / 64.0 * n
    ideal_ms = insts * CYCLES_PER_VALU / (SIMDS * CLOCK_HZ) * 1e3
    return {"ideal_valu_insts": int(insts), "ideal_ms": round(ideal_ms, 5), "frac": round(ideal_ms / ms, 4) if ms > 0 else None,
            "def": f"live_pairs / 64 x {n} wave instructions x 2 cycles / (1024 SIMDs x 2.4 GHz) / measured ms"}
