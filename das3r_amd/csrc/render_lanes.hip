// render_lanes.hip — forward compositing (K6) for FEW TILES WITH LONG LISTS: four lanes per pixel.  Replaces
// upstream:cuda_rasterizer/forward.cu renderCUDA like render_fwd.hip / render_rows.hip; same inputs and outputs.
//
// Why (round 4, DESIGN.md section 4 ledger (aq)).  The DAS3R training shape — one Gaussian per pixel of twenty 512 x 208 frames — has
// 416 tiles whose lists hold 6 000 - 20 000 entries.  With one pixel per lane (render_rows.hip) that is 1 664 waves for the chip's
// 1 024 SIMDs: 1.6 waves per SIMD, every one a dependent chain of ~50 instructions per list position which nobody else covers —
// 4.3 ns per instruction where a saturated SIMD takes 0.96 (profiles/r03_valu_rate_probe.txt) — the wave iterating for the
// longest of its four rows, and the kernel as long as the tile with the longest list (twice the mean on smooth depth maps).
// The list order cannot be cut (T is a running product), but forty of the fifty instructions of a position — fetching the
// entry, the exponent, alpha and its two tests — do not depend on the pixel's state.  Here a wave owns ONE 4x4 block and the four
// lanes of a quad own one pixel: a step takes FOUR consecutive entries of the block's list, one per lane of the quad;
//   * every lane evaluates alpha of its own entry (the same arithmetic: render_common.h);
//   * T in front of entry k is the pixel's T times (1 - alpha_j) of the quad's lanes j < k, multiplied IN LIST ORDER — three
//     quad-broadcast DPP instructions, the lanes j >= k multiplying by max(1 - alpha_j, 1) = 1, which is exact — so every entry
//     sees the very T the one-pixel-per-lane kernels compute: final_T, n_contrib and every stop are theirs bit for bit;
//   * a stop (T would fall below 1e-4) inside a step is rare — once in a pixel's life — and is detected for the whole wave with one
//     compare; the step is then finished with the masks the stop implies;
//   * each lane adds its own entries' colour into its own partial sums; the four are added when a checkpoint or the image is
//     written: the colour is the same sum in a different order (last-bit differences; the tests hold it to 2e-6).
// 16 waves per tile: 6 656 waves, 6.5 per SIMD, none waiting for a longer sibling row.  Work per pixel and entry: 52 issue slots
// per step of 16 pixels x 4 entries against 200 per trip of 64 pixels x 4 positions — the same.
#include "render_quad.h"

namespace das3r {

__global__ void __launch_bounds__(LN_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) render_forward_lanes_kernel(const uint2 *__restrict__ ranges, const uint32_t *__restrict__ point_list, int W, int H,
                                                                          int tiles_x, int ntiles_strip /*render_common.h pack_tiles*/,
                                                                          const float4 *__restrict__ xyh, const float4 *__restrict__ conic_opacity,
                                                                          const float4 *__restrict__ rgbd, const float *__restrict__ bg,
                                                                          float *__restrict__ final_T, uint32_t *__restrict__ n_contrib,
                                                                          float *__restrict__ out_color, const LocalBin lb,
                                                                          unsigned long long *__restrict__ pairs /*common.h pair_counters()*/) {
    __shared__ StagedSplat stage_all[2 * LN_BATCH];
    __shared__ uint16_t lists[16][LN_LIST];     // [wave][position]: staged index
    __shared__ uint32_t s_done[2][16];
    const int tile = xcd_tile(blockIdx.x, ntiles_strip, tiles_x);
    if (tile < 0) return;
    const int tid = threadIdx.x, lane = __lane_id(), wave = __builtin_amdgcn_readfirstlane(tid >> 6), k = lane & 3, pix = lane >> 2;
    const int bx = tile % tiles_x, by = tile / tiles_x;
    const int px = bx * TILE_X + ((wave & 3) << 2) + (pix & 3), py = by * TILE_Y + ((wave >> 2) << 2) + (pix >> 2);
    const bool inside = px < W && py < H;
    const float bcx = (float)(bx * TILE_X + ((wave & 3) << 2)) + 1.5f, bcy = (float)(by * TILE_Y + ((wave >> 2) << 2)) + 1.5f;
    const uint2 range = safe_range(ranges[tile], lb.cap);
    const uint32_t n = range.y - range.x;
    const int rounds = (int)((n + LN_BATCH - 1) / LN_BATCH);
    QuadLane q;
    q.T = 1.0f; q.live = inside ? 1.f : 0.f; q.C0 = q.C1 = q.C2 = 0.f;
    q.pxf = (float)px; q.pyf = (float)py; q.k = k; q.kf = (float)k;
    q.mk0 = k > 0 ? 0.f : 1.f; q.mk1 = k > 1 ? 0.f : 1.f; q.mk2 = k > 2 ? 0.f : 1.f;
    uint32_t last_contributor = 0;                                              // (this lane's entries; the quad's maximum is the pixel's)
    const int nb = ckpt_buckets(range);
    const int cpix = ((py - by * TILE_Y) << 4) + (px - bx * TILE_X);
    int next_slot = 0;
    int steps = 0;
    for (int i = tid; i < 16 * LN_LIST; i += LN_THREADS) (&lists[0][0])[i] = 0;   // (a stale list word must name a staged entry)

    // Staging (threads 0 .. LN_BATCH - 1: waves 0 .. 7): batch i + 1 goes into the other area DURING batch i — its records are fetched behind
    // the barrier of batch i (nobody reads that area any more: everybody has finished batch i - 1), arrive while the wave builds its list
    // and are written in front of its walk, so that the walk does not carry them in registers; the list words of batch i + 2 are in
    // flight meanwhile.
    uint32_t g_ahead = 0u;
    const bool loader = tid < LN_BATCH;   // (uniform per wave)
    if (loader) {
        StagedSplat rec = null_splat();
        if ((uint32_t)tid < n) {
            const uint32_t g = min(point_list[range.x + tid], lb.last_g);
            rec.xyh = xyh[(size_t)g * SPLAT_REC];
            rec.co = conic_opacity[(size_t)g * SPLAT_REC];
            rec.rgbd = rgbd[(size_t)g * SPLAT_REC];
        }
        if ((uint32_t)(LN_BATCH + tid) < n) g_ahead = point_list[range.x + LN_BATCH + tid];
        stage_all[tid] = rec;
    }
    for (int i = 0; i < rounds; i++) {
        StagedSplat *const stage = stage_all + (i & 1) * LN_BATCH;
        const uint32_t first = (uint32_t)i * LN_BATCH;
        const bool wave_done = __ballot(q.live != 0.f) == 0ull;
        if (lane == 0) s_done[i & 1][wave] = wave_done ? 1u : 0u;
        lds_barrier();   // (the loads just issued stay in flight: render_common.h)
        {   // every pixel of the tile has stopped?  (flags of this batch: rewritten two batches on, behind the next barrier)
            const uint32_t d = s_done[i & 1][lane & 15];
            if (__ballot(d != 0u) == ~0ull) break;
        }
        if (nb > 1 && i > 0 && first % BUCKET == 0) {   // the state in front of list position `first`
            const float q0 = quad_sum(q.C0), q1 = quad_sum(q.C1), q2 = quad_sum(q.C2);
            if (k == 0) ckpt_slot(lb.ckpt, range, tile, next_slot)[cpix] = make_float4(q.T, q0, q1, q2);
            next_slot++;
        }
        StagedSplat rec = null_splat();
        const uint32_t progress = range.x + first + LN_BATCH + tid;   // my entry of batch i + 1
        if (loader && i + 1 < rounds) {
            if (progress < range.y) {
                const uint32_t g = min(g_ahead, lb.last_g);
                rec.xyh = xyh[(size_t)g * SPLAT_REC];
                rec.co = conic_opacity[(size_t)g * SPLAT_REC];
                rec.rgbd = rgbd[(size_t)g * SPLAT_REC];
            }
            if (progress + LN_BATCH < range.y) g_ahead = point_list[progress + LN_BATCH];
        }
        // ---- this block's list of the batch ------------------------------------------------------------------------------------
        uint16_t *const mine = lists[wave];
        int len = 0;
        if (!wave_done) {
            const int nstaged = (int)min(n - first, (uint32_t)LN_BATCH);
#pragma unroll
            for (int c = 0; c < LN_BATCH / 64; c++) {
                const int s = c * 64 + lane;
                const float4 p = stage[s].xyh;   // (entries past the list hold extents no block can meet)
                const bool hit = s < nstaged && fabsf(p.x - bcx) <= p.z + 1.5f && fabsf(p.y - bcy) <= p.w + 1.5f;
                const uint64_t m = __ballot(hit);
                const int at = len + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                if (hit) mine[at] = (uint16_t)s;
                len += __popcll(m);
            }
        }
        if (loader && i + 1 < rounds) stage_all[((i + 1) & 1) * LN_BATCH + tid] = rec;
        if (wave_done) continue;   // (uniform; the wave has staged its share and meets the barriers)
        // ---- the walk: four entries per step, one per lane of a quad -----------------------------------------------------------
        const float lastf = lanes_walk(stage, mine, len, q, steps);
        if (lastf >= 0.0f) last_contributor = first + (uint32_t)lastf + 1u;
    }
    const float q0 = quad_sum(q.C0), q1 = quad_sum(q.C1), q2 = quad_sum(q.C2);
    const uint32_t last = quad_max(last_contributor);
    if (k == 0)
        for (; nb > 1 && next_slot < nb; next_slot++) ckpt_slot(lb.ckpt, range, tile, next_slot)[cpix] = make_float4(q.T, q0, q1, q2);   // (final values)
    if (inside && k == 0) {
        const size_t at = (size_t)py * W + px, plane = (size_t)H * W;
        final_T[at] = q.T;
        n_contrib[at] = last;
        out_color[at] = q0 + q.T * bg[0];
        out_color[plane + at] = q1 + q.T * bg[1];
        out_color[2 * plane + at] = q2 + q.T * bg[2];
    }
    if (pairs != nullptr && lane == 0 && steps > 0) {
        atomicAdd(pairs, (unsigned long long)steps * 64ull);
        atomicAdd(pairs + 2, (unsigned long long)steps);
    }
}

// Few tiles (fewer waves than the SIMDs can interleave with one pixel per lane), lists from the global sort: see the head of the file.
bool use_quad_lanes(const Layout &L, const LocalBin &lb) {
    if (lb.point_list != nullptr) return false;   // (lists in local depth order: render_rows.hip sorts them itself)
    const int forced = switches().render_fwd;
    if (forced) return forced >= 3;
    return L.ntiles <= 1024 && L.capacity >= (int64_t)1024 * L.ntiles;
}

int launch_render_forward_lanes(const das3r_raster_args *a, float *out_color, char *geom, char *binning, char *img, const Layout &L, const LocalBin &lb,
                                hipStream_t s) {
#define ARGS                                                                                                                                   \
    (const uint2 *)(img + L.pub.ranges), (const uint32_t *)(binning + L.pub.point_list), a->image_width, a->image_height, L.tiles_x, pack_tiles(L), \
        (const float4 *)(geom + L.pub.xy), (const float4 *)(geom + L.pub.conic_opacity), (const float4 *)(geom + L.pub.rgbd), a->bg,                \
        (float *)(img + L.pub.final_T), (uint32_t *)(img + L.pub.n_contrib), out_color, lb, pair_counters()
    DAS3R_LAUNCH(render_forward_lanes_kernel, dim3(xcd_grid(L)), dim3(LN_THREADS), 0, s, ARGS);
#undef ARGS
    KERNEL_CHECK(s, a->debug, "render_forward_lanes");
    return DAS3R_OK;
}

}  // namespace das3r
