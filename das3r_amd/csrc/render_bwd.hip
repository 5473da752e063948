// render_bwd.hip — K7: back-to-front replay of the per-pixel compositing, producing per-INSTANCE partial sums of
// dL/d(colour, mean2D, conic, opacity).  Replaces upstream:cuda_rasterizer/backward.cu renderCUDA, whose
// 9 float atomicAdds per (pixel, splat) pair are the classic 3DGS training hot spot.
//
// MI355X design (decomposition in render_common.h): each wave64 owns an 8x8 quadrant and walks only the splats that can
// touch it (ballot-culled sub-list).  The 64 pixels of the wave are reduced on the cross-lane network — 8 of the 9 sums
// with the transposed reduction (permlane32/16 swap + DPP, 18 ops), the ninth with a 6-step DPP chain — and the four
// waves of the tile meet in LDS (one ds_add_f32 per wave and splat, 9 lanes -> 9 addresses).
// NO global atomics: device-scope float atomics measured ~0.5 ms of a 1.2 ms kernel at 1M splats (they leave the XCD's
// L2).  Instead the tile writes its 9 sums per list entry to partial[emission slot][9] (the binning keeps every list entry's
// emission slot in list order) — a splat's instances own consecutive slots, so the per-Gaussian backward kernel
// (preprocess_bwd.hip) reads its few rows contiguously.  The only run-to-run variation left is the order of the four
// per-wave LDS adds of a (tile, splat) sum.
#include "render_common.h"

namespace das3r {

constexpr int NACC = 9;  // dcolor[3], dmean2D[2], dconic[3], dopacity

template <bool USE_DPP>
__global__ void __launch_bounds__(256) render_backward_kernel(
    const uint2 *__restrict__ ranges, const uint32_t *__restrict__ point_list, int W, int H, int tiles_x, int ntiles_strip /*render_common.h pack_tiles*/,
    const float4 *__restrict__ xyh, const float4 *__restrict__ conic_opacity, const float4 *__restrict__ rgbd,
    const float *__restrict__ bg, const float *__restrict__ final_T, const uint32_t *__restrict__ n_contrib,
    const float *__restrict__ dL_dpix, const uint32_t *__restrict__ slot_list, float *__restrict__ partial /*[I,9]*/, int ablate,
    uint32_t last_g, uint32_t cap /*bounds of the list contents: render_common.h safe_range*/,
    unsigned long long *__restrict__ pairs /*common.h pair_counters(): null unless bench.py counts*/) {
    __shared__ StagedSplat stage[TILE_PIX];
    __shared__ uint32_t s_slot[TILE_PIX];   // emission slot of every staged entry = its row of `partial`
    __shared__ float acc[TILE_PIX * NACC];
    __shared__ uint32_t s_max[4];

    const int ntiles = packed_ntiles(ntiles_strip);
    const int tile = xcd_tile(blockIdx.x, ntiles_strip, tiles_x);
    if (tile < 0) return;
    const int tid = threadIdx.x, lane = __lane_id(), wave = tid >> 6;
    const int bx = tile % tiles_x, by = tile / tiles_x;
    int px, py;
    quadrant_pixel(bx, by, wave, lane, px, py);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const float qcx = (float)(bx * TILE_X + ((wave & 1) << 3)) + 3.5f, qcy = (float)(by * TILE_Y + ((wave >> 1) << 3)) + 3.5f;
    const uint2 range = safe_range(ranges[tile], cap);
    const size_t pix = (size_t)py * W + px, plane = (size_t)H * W;

    const float T_final = inside ? final_T[pix] : 0.f;
    const uint32_t last_contributor = inside ? n_contrib[pix] : 0u;
    float dLp0 = 0.f, dLp1 = 0.f, dLp2 = 0.f;
    if (inside) {
        dLp0 = dL_dpix[pix];
        dLp1 = dL_dpix[plane + pix];
        dLp2 = dL_dpix[2 * plane + pix];
    }
    const float bg_dot_dpixel = bg[0] * dLp0 + bg[1] * dLp1 + bg[2] * dLp2;
    const float ddelx_dx = 0.5f * W, ddely_dy = 0.5f * H;

    // no pixel of this tile blended anything past list position max_contrib: start the replay there
    uint32_t mx = last_contributor;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, o, 64));
    if (lane == 0) s_max[wave] = mx;
    __syncthreads();
    const uint32_t max_contrib = min(max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3])), range.y - range.x);
    const int rounds = ((int)max_contrib + TILE_PIX - 1) / TILE_PIX;

    // list entries beyond max_contrib receive no gradient from this tile: their partial rows are zero
    {
        const uint32_t len = range.y - range.x;
        const uint32_t ntail = (len - max_contrib) * NACC;
        for (uint32_t f = tid; f < ntail; f += TILE_PIX) {
            const uint32_t t = f / NACC, q = f - t * NACC;
            partial[(size_t)min(slot_list[range.x + max_contrib + t], cap - 1u) * NACC + q] = 0.f;
        }
    }

    ReplayState st = {T_final, 0.f};
    const float tfbg = T_final * bg_dot_dpixel;
    // which accumulator this lane feeds after the cross-lane reduction
    // DPP path: lanes 0, 8, .., 56 hold the wave totals of sums 0..7, lane 15 of every row its ROW's total of the ninth — the four
    // rows meet in the LDS add like the four waves do (two broadcast steps, six instructions, less on the vector ALU)
    const bool writer = USE_DPP ? (((lane & 7) == 0) || ((lane & 15) == 15)) : (lane < NACC);
    const int widx = USE_DPP ? ((lane & 15) == 15 ? 8 : (lane >> 3)) : lane;
    const unsigned widx4 = (unsigned)widx * 4u;

    int visits = 0;   // (wave-uniform) (splat, quadrant) visits = 64 pairs each
    for (int i = 0; i < rounds; i++) {
        const int done_before = i * TILE_PIX;
        const int n = min(TILE_PIX, (int)max_contrib - done_before);
        // stage the batch in reverse list order; entry j holds list position (max_contrib - 1 - done_before - j)
        if (tid < n) {
            const uint32_t pos = range.x + max_contrib - 1 - done_before - tid;
            const uint32_t g = min(point_list[pos], last_g);
            s_slot[tid] = min(slot_list[pos], cap - 1u);
            stage[tid].xyh = xyh[(size_t)g * SPLAT_REC];           // one 64-byte record: a single cache line per splat
            stage[tid].co = conic_opacity[(size_t)g * SPLAT_REC];
            stage[tid].rgbd = rgbd[(size_t)g * SPLAT_REC];
        }
#pragma unroll
        for (int k = 0; k < NACC; k++) acc[k * TILE_PIX + tid] = 0.f;   // acc[j*9 + q], zeroed with unit-stride stores
        __syncthreads();

        uint64_t masks[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int s = k * 64 + lane;
            masks[k] = __ballot(s < n && quadrant_hit(stage[s].xyh, qcx, qcy));
            visits += __popcll(masks[k]);
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            uint64_t m = masks[k];
            while (m != 0ull) {
                const int j = k * 64 + __builtin_ctzll(m);
                m &= m - 1ull;
                const uint32_t position = max_contrib - 1 - done_before - j;  // 0-based list position
                const float4 p = stage[j].xyh;
                const float4 co = stage[j].co;
                float dx, dy, G, alpha;
                const bool active = pair_alpha(p.x, p.y, co, pxf, pyf, dx, dy, G, alpha) & (position < last_contributor);
                if (__ballot(active) == 0ull) continue;  // wave-uniform skip
                const float4 c = lds_read4(&stage[j].rgbd);   // (b128, not b96: half the LDS cycles)
                float v[NACC];
                replay_pair_moments(active, alpha, G, dx, dy, c, dLp0, dLp1, dLp2, tfbg, st, v);   // geometry sums as raw moments
                float out;
                if (ablate & 4) {  // experiment: no cross-lane traffic, heavy arithmetic kept alive
                    out = v[0] + v[1] + v[2] + v[3] + v[4] + v[5] + v[6] + v[7] + v[8];
                } else if (USE_DPP) {
                    const float r8 = wave_reduce8_transposed(v, lane);   // lane l: total of v[l >> 3]
                    const float r1 = row_sum_to_lane15(v[8]);            // lane 15 of every row: that row's total of v[8]
                    out = (lane & 15) == 15 ? r1 : r8;
                } else {
                    // reference reduction (ds_bpermute butterflies): every lane gets every total
                    float tot[NACC];
#pragma unroll
                    for (int q = 0; q < NACC; q++) tot[q] = wave_sum_shfl(v[q]);
                    out = 0.f;
#pragma unroll
                    for (int q = 0; q < NACC; q++) out = lane == q ? tot[q] : out;
                }
                // (address: one full-rate 24-bit multiply-add; the compiler's choice for j * 9 + widx is a quarter-rate v_mad_u64_u32)
                float *const slot = reinterpret_cast<float *>(reinterpret_cast<char *>(acc) + __umul24((unsigned)j, NACC * 4u) + widx4);
                if (ablate & 8) { if (writer) *slot = out; }   // timing experiment: plain store instead of the atomic
                else if (writer) atomicAdd(slot, out);  // divergent addresses: one ds_add_f32 for the whole wave
            }
        }
        __syncthreads();
        if (tid < n) moments_to_sums(&acc[tid * NACC], stage[tid].co, ddelx_dx, ddely_dy);   // once per (tile, splat), not per pair
        __syncthreads();
        // every staged entry's 9 sums go to the row of its emission slot: a splat's rows are then contiguous for the
        // per-Gaussian backward kernel (36-byte row stores, 9 lanes each)
        if (!(ablate & 1)) {
            for (int f = tid; f < n * NACC; f += TILE_PIX) {
                const int j = f / NACC, q = f - j * NACC;
                partial[(size_t)s_slot[j] * NACC + q] = acc[f];
            }
        }
        __syncthreads();
    }
    if (pairs != nullptr && lane == 0 && visits > 0) {
        atomicAdd(pairs + 1, (unsigned long long)visits * 64ull);
        atomicAdd(pairs + 3, (unsigned long long)visits);
    }
}

int launch_render_backward(const das3r_raster_args *a, const float *dL_dpix, char *geom, char *binning, char *img, const Layout &L,
                           float *partial, hipStream_t s, bool *quad_rows, int64_t num_rendered, uint32_t fwd_flags) {
    *quad_rows = false;
    // Which decomposition (DAS3R_RENDER_BWD=dpp | mfma | scan<N> | scana<N> | stream forces one; measurements: DESIGN.md §4):
    //   dpp     pixel per lane, cross-lane reduction on the vector ALU (this file): lists of a few hundred entries per tile
    //   fine    every DPP row of a wave on a 2x2 region's list, four pixel steps per batch (render_bwd_rgn.hip): long, spatially coherent lists
    //   blk     every DPP row of a wave on its own 4x4 block: lanes = 16 splats of the block's culled list, time = its 16 pixels,
    //           recurrences as DPP row scans, sums in fp32 registers (render_bwd_blk.hip): everything but short lists
    //   scan    lanes = 4 pixels x 16 splats of the QUADRANT's list, sums as split-bf16 products on the matrix cores
    //           (render_bwd_scan.hip): round 2's kernel for long lists, kept as the reference for blk
    //   stream  the same arithmetic, every wave streaming the tile's list on its own (render_bwd_stream.hip; experimental)
    //   mfma    pixel per lane + LDS-transposed slab -> fp32 matrix cores (render_bwd_mfma.hip; superseded by scan)
    const Switches &sw = switches();
    int kind = sw.render_bwd;
    int mb = sw.render_bwd_mb ? sw.render_bwd_mb : 256;
    if (kind == 0) {
        // measured (render backward, ms; tools/gpu_perf.py): 100 k splats at 1080p, mean list 32: dpp 0.071 / blk64 0.068 / scan128 0.129;
        // 1 M splats, mean 320: dpp 0.532 / scan128 0.476 / blk128p1 0.381; DAS3R shape (13 800, bucket-parallel): dpp 1.89 / scan128 0.78 /
        // blk192 0.497.  Round 2's crossover between dpp and the quadrant walk was a mean list of 192; the block walk culls per 4x4
        // block and is ahead from ~100 entries per tile on (tools/gpu_perf.py --workloads c4:<P>: DESIGN.md section 5).
        const int64_t mean_list = num_rendered / std::max(L.ntiles, 1);   // (the count, not the capacity: the same scene takes the same kernel however its buffer was sized)
        const bool long_lists = mean_list >= 96;
        kind = (sw.bwd_reduce_set || sw.ablate_set || !long_lists) ? 1 : 6;
        mb = mean_list >= 1024 ? 192 : 128;
        // round 6: long lists that the forward found spatially coherent or skewed (das3r_raster_saved.flags bit 0: the depth maps of a real
        // sequence) take the 2x2-region walk — self-consistent Sintel-shaped job: backward 0.548 -> 0.414 ms, dsc 0.865 -> 0.590; random depths
        // stay on the block walk (ds 0.50 against 0.55, noise-depth train step 0.344 against 0.369)
        if (kind == 6 && mean_list >= 1024 && L.ntiles <= 1024 && (fwd_flags & 1u) && !sw.ablate_set) {
            kind = 7;
            mb = 128;
        }
        if (sw.deterministic && kind == 1) {   // short lists too on the block walk: every sum has a fixed order (rows 0..3 of a wave, waves 0..3)
            kind = 6;
            mb = 64;
        }
    }
#ifdef DAS3R_EXPERIMENTS
    if (kind == 5) {
        *quad_rows = true;
        return launch_render_backward_stream(a, dL_dpix, geom, binning, img, L, partial, s);
    }
    if (kind == 2) return launch_render_backward_mfma(a, dL_dpix, geom, binning, img, L, partial, s);
#else
    if (kind == 5 || kind == 2) {
        set_error("DAS3R_RENDER_BWD=%s: this library was built without the superseded kernels (make EXPERIMENTS=1)", kind == 5 ? "stream" : "mfma");
        return DAS3R_ERR_INVALID_ARG;
    }
#endif
    if (kind == 3 || kind == 6 || kind == 7) {
        // long lists are replayed bucket by bucket in parallel workgroups (checkpoints from the forward: common.h BUCKET); slices =
        // buckets of an average tile, so that a tile's workgroups take about one bucket each
        int slices = sw.bwd_buckets;
        if (slices < 0) slices = (int)std::min<int64_t>(32, std::max<int64_t>(1, num_rendered / ((int64_t)BUCKET * std::max(L.ntiles, 1))));
        if (slices > 1 && mb > 256) slices = 1;
        // round 6: skewed lists (das3r_raster_saved.flags bits 8 - 15: buckets of the longest list the forward last measured) — a workgroup per
        // bucket of the LONGEST tile; the workgroups of shorter tiles that have no bucket leave before they load anything
        const int hint = (int)((fwd_flags >> 8) & 0xFFu);
        if (sw.bwd_buckets < 0 && kind == 7 && slices > 1 && hint > slices) slices = std::min(hint, 64);
        if (kind == 6) return launch_render_backward_blk(a, dL_dpix, geom, binning, img, L, partial, mb, slices, s);
        if (kind == 7) return launch_render_backward_regions(a, dL_dpix, geom, binning, img, L, partial, mb, slices, s);
        return launch_render_backward_scan(a, dL_dpix, geom, binning, img, L, partial, mb, slices, s);
    }
    const bool use_dpp = !sw.bwd_reduce_shfl;   // "shfl" selects the ds_bpermute reference reduction (diagnostics)
    const int ablate = sw.ablate;               // perf experiments only: bit0 = no partial stores, bit2 = no cross-lane reduction
#define ARGS                                                                                                                 \
    (const uint2 *)(img + L.pub.ranges), (const uint32_t *)(binning + L.pub.point_list), a->image_width, a->image_height,    \
        L.tiles_x, pack_tiles(L), (const float4 *)(geom + L.pub.xy), (const float4 *)(geom + L.pub.conic_opacity),                \
        (const float4 *)(geom + L.pub.rgbd), a->bg, (const float *)(img + L.pub.final_T),                                    \
        (const uint32_t *)(img + L.pub.n_contrib), dL_dpix, (const uint32_t *)(binning + L.b_slot), partial, ablate, \
        (uint32_t)(a->P - 1), (uint32_t)L.capacity, pair_counters()
    const int pad_lds = sw.bwd_pad_lds;   // occupancy experiments
    if (use_dpp) DAS3R_LAUNCH((render_backward_kernel<true>), dim3(xcd_grid(L)), dim3(TILE_PIX), pad_lds, s, ARGS);
    else DAS3R_LAUNCH((render_backward_kernel<false>), dim3(xcd_grid(L)), dim3(TILE_PIX), 0, s, ARGS);
#undef ARGS
    KERNEL_CHECK(s, a->debug, "render_backward");
    return DAS3R_OK;
}

}  // namespace das3r
