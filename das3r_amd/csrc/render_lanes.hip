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
#include "render_common.h"

namespace das3r {

constexpr int LN_THREADS = 1024;   // sixteen waves: one per 4x4 block of the tile
#ifndef LN_BATCH_N
#define LN_BATCH_N 512
#endif
constexpr int LN_BATCH = LN_BATCH_N;      // entries staged per batch (two staging areas: one barrier per batch)
#ifndef LN_UNROLL
#define LN_UNROLL 2   // (1, 2 or 4)
#endif
constexpr int LN_LIST = LN_BATCH + 4 * LN_UNROLL + 4;

template <int CTRL>
__device__ __forceinline__ float quad_perm(const float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
// f_j = max(v of the quad's lane j, m_j), j = 0, 1, 2: one DPP instruction each.  (fmaxf(update_dpp(..), m) costs three: the compiler
// canonicalises the moved bits first.  Inline assembly is invisible to the hazard recogniser: a VALU write of v needs two wait
// states in front of a DPP read — the s_nop; the three reads sit in one block so that it covers them all.)
__device__ __forceinline__ void quad_factors(const float v, const float m0, const float m1, const float m2, float &f0, float &f1, float &f2) {
    asm("s_nop 1\n\t"
        "v_max_f32_dpp %0, %3, %4 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_max_f32_dpp %1, %3, %5 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_max_f32_dpp %2, %3, %6 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf bound_ctrl:1"
        : "=&v"(f0), "=&v"(f1), "=&v"(f2)
        : "v"(v), "v"(m0), "v"(m1), "v"(m2));
}
// min / max of numbers known not to be signalling NaNs (fminf / fmaxf canonicalise operands whose origin the compiler cannot see)
__device__ __forceinline__ float min_raw(const float a, const float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float max_raw(const float a, const float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float quad_sum(const float v) {
    const float a = v + quad_perm<0xB1>(v);   // [1,0,3,2]
    return a + quad_perm<0x4E>(a);            // [2,3,0,1]
}
__device__ __forceinline__ float quad_min(const float v) {
    const float a = fminf(v, quad_perm<0xB1>(v));
    return fminf(a, quad_perm<0x4E>(a));
}
__device__ __forceinline__ uint32_t quad_max(const uint32_t v) {
    const uint32_t a = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true));
    return max(a, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a, 0x4E, 0xf, 0xf, true));
}

// Per-pixel state of a quad's lanes and the lane's constants.  T and live are the same in the four lanes of a quad; C: this lane's entries.
struct QuadLane {
    float T, live, C0, C1, C2;
    float pxf, pyf, kf, mk0, mk1, mk2;   // pixel centre; lane of the quad; max(1 - alpha_j, mk[j]): the factor of the quad's lane j in front of MY entry — 1 - alpha_j for j < k, 1 for j >= k
    int k;
};

// The walk of one block's list `mine[0 .. len)` (staged indices) of a staged batch: four entries per step, one per lane of a quad.
// -> staged index of this lane's last contributing entry as a float, -1 = none.
__device__ __forceinline__ float lanes_walk(const StagedSplat *__restrict__ stage, const uint16_t *__restrict__ mine, const int len, QuadLane &q, int &steps) {
    float lastf = -1.0f;
    const float lenf = (float)len - q.kf;   // (my position of step t exists where lenf - t >= 1)
    // alpha of staged entry j for my pixel (0 where it is invisible or the list has no such position), its colour
    auto entry_alpha = [&](const int j, const float rem, float4 &c) -> float {
        const float4 p = stage[j].xyh;
        const float4 co = stage[j].co;
        c = lds_read4(&stage[j].rgbd);
        const float dx = p.x - q.pxf, dy = p.y - q.pyf;
        const float qq = __fmaf_rn(__fmul_rn(co.x, dx), dx, __fmul_rn(__fmul_rn(co.z, dy), dy));
        const float power = __fmaf_rn(-0.5f, qq, -__fmul_rn(__fmul_rn(co.y, dx), dy));   // (pair_alpha's arithmetic)
        const float a1 = fminf(fminf(0.99f, __fmul_rn(co.w, __expf(power))), rem);
        return alpha_if_visible(a1, power);   // (a1 is not positive where the list has no position)
    };
    // the quad's four entries into the pixel, in list order
    auto blend_step = [&](const float av, const float4 c, const float jf) {
        const float a = av * q.live;
        const float om = 1.0f - a;
        // T in front of my entry: the pixel's T times the factors of the lanes in front of me, in list order
        float f0, f1, f2;
        quad_factors(om, q.mk0, q.mk1, q.mk2, f0, f1, f2);
        const float x = __fmul_rn(__fmul_rn(__fmul_rn(q.T, f0), f1), f2);
        const float tn = __fmul_rn(x, om);   // the reference's test_T of my entry
        // A pixel of this wave stops inside the step (rare: once in a pixel's life): test_T falls along the quad, the entries in front
        // of the first failure are taken as they are (their T does not involve the failing entry), the failing one and those behind
        // it are not.  Only the three values below differ; the common path overwrites nothing it has to keep.
        float s = 1.0f, t_next = quad_perm<0xFF>(tn), l_next = q.live;
        if (__builtin_expect(__ballot(tn < 0.0001f) != 0ull, 0)) {
            s = tn < 0.0001f ? 0.f : 1.f;
            t_next = quad_min(s != 0.f ? tn : q.T);   // T behind the last entry taken (the pixel's T where none is)
            l_next = q.live * quad_min(s);
        }
        const float w = a * s, wT = w * x;
        q.C0 = __fmaf_rn(c.x, wT, q.C0);
        q.C1 = __fmaf_rn(c.y, wT, q.C1);
        q.C2 = __fmaf_rn(c.z, wT, q.C2);
        lastf = max_raw(lastf, min_raw(jf, __fmaf_rn(w, 1e30f, -1.0f)));
        q.T = t_next;
        q.live = l_next;
    };
    // LN_UNROLL steps per trip: the entries' fetches and exponents are independent of the pixel's state and overlap; the blends follow in order
    for (int t = 0; t < len; t += 4 * LN_UNROLL) {
        if ((t & 63) == 0 && __ballot(q.live != 0.f) == 0ull) break;
        steps += min(LN_UNROLL, (len - t + 3) >> 2);
        int j[LN_UNROLL];
        float4 c[LN_UNROLL];
        float av[LN_UNROLL];
        const float rem = lenf - (float)t;
#pragma unroll
        for (int u = 0; u < LN_UNROLL; u++) j[u] = (int)mine[t + 4 * u + q.k];
#pragma unroll
        for (int u = 0; u < LN_UNROLL; u++) av[u] = entry_alpha(j[u], rem - (float)(4 * u), c[u]);
#pragma unroll
        for (int u = 0; u < LN_UNROLL; u++) blend_step(av[u], c[u], (float)j[u]);
    }
    return lastf;
}

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
    if (forced) return forced == 3;
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
