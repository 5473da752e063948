// render_regions.hip — forward compositing (K6) for few tiles with long lists: SIXTEEN lanes per pixel, a wave per 2x2 pixel region, four
// workgroups per tile.  Replaces upstream:cuda_rasterizer/forward.cu renderCUDA like the other forward kernels; same inputs and outputs.
//
// Why (round 6; docs/ledger.md (bd)).  render_lanes.hip runs one workgroup of sixteen waves per tile: a wave = a 4x4 block,
// a quad of lanes = a pixel, four list entries per step.  On the data a job really trains on — the self-consistent sequence: every frame
// sees ONE surface, the dense part of the image holds lists of 30 000 entries where the mean is 5 600 — that kernel does 58 us of
// arithmetic in 550 us: it lasts as long as the chain of its longest tile's slowest block (3 517 steps of ~50 dependent-issue
// instructions, tools/probes/train_lists.py), with most of the chip idle.  Two things shorten that chain, and both also cut the work:
//   * sixteen lanes per pixel: a 16-lane DPP row = one pixel, a step takes SIXTEEN consecutive entries of the pixel group's list; T in
//     front of entry k = the pixel's T times the product of (1 - alpha_j) of the lanes before it — an exclusive row scan, four
//     v_mul_f32_dpp steps for the inclusive product + one shift — so the per-pixel recurrence advances sixteen entries per ~55 issue slots
//     instead of four per ~50;
//   * the unit of culling is the 2x2 region the wave owns: Gaussians of one or two pixels (the DAS3R shape) are listed for 5.9 regions
//     = 23 (pixel, entry) pairs instead of 2.5 blocks = 41 (self-consistent sequence; 50 against 70 on the shifted reliefs).
// A tile's 64 regions are four workgroups of sixteen waves, one per 8x8 quadrant, each staging the tile's list for itself (the same
// 64-byte records, requested four times: they are in the XCD's L2 after the first — the four workgroups of a tile are dealt to one XCD,
// back to back).  Model on the lists of a real forward: the longest workgroup chain falls 5.9 x (3 517 -> 593 steps), the walk's total
// steps by 23 - 37 %.
// What differs from the four-lanes kernel: T in front of an entry is T_pixel x (tree-order product of the lanes before it), not the
// running product — a relative 1e-7; every stop is still decided entry by entry in list order (test_T = T x inclusive product, monotone
// along the row), so n_contrib differs only where a pixel's T comes within that rounding of 1e-4 (tests: within util.FLIP_FRACTION of the
// four-lanes kernel, and within the parity bars of the oracle); the colour is the same sum in another order.
#include "render_quad.h"

namespace das3r {

constexpr int RG_THREADS = 1024;          // sixteen waves: the sixteen 2x2 regions of an 8x8 quadrant
constexpr int RG_BATCH = 512;             // tile entries staged per batch (two areas, as in render_lanes.hip)
constexpr int RG_UNROLL = 2;              // steps per trip of the walk
constexpr int RG_LIST = RG_BATCH + 16 * RG_UNROLL + 16;

// inclusive product along every 16-lane row, tree order (row_shr 1, 2, 4, 8; a lane without a source lane keeps its value)
__device__ __forceinline__ float row_scan_mul(float x) {
    asm("s_nop 1\n\t"
        "v_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_mul_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_mul_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_mul_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(x));
    return x;
}
// two rows of independent values scanned side by side: the second chain fills the first one's wait states
__device__ __forceinline__ void row_scan_mul_x2(float &x, float &y) {
    asm("s_nop 1\n\t"
        "v_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_mul_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_mul_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_mul_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(x), "+v"(y));
}
// the value of the lane before me in the row; lane 0 of a row gets `first`
__device__ __forceinline__ float row_shift1(const float x, const float first) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(first), __float_as_int(x), 0x111 /*row_shr:1*/, 0xf, 0xf, false));
}
template <int K>
__device__ __forceinline__ float row_bcast(const float x) {   // lane K of every row, to the whole row
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x150 + K, 0xf, 0xf, true));
}
__device__ __forceinline__ float row_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128 /*row_ror:8*/, 0xf, 0xf, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124 /*row_ror:4*/, 0xf, 0xf, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122 /*row_ror:2*/, 0xf, 0xf, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121 /*row_ror:1*/, 0xf, 0xf, true));
    return v;
}
__device__ __forceinline__ float row_min(float v) {
    v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, true)));
    v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, true)));
    v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122, 0xf, 0xf, true)));
    v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xf, 0xf, true)));
    return v;
}
__device__ __forceinline__ uint32_t row_max_u32(uint32_t v) {
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, true));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xf, 0xf, true));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x122, 0xf, 0xf, true));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x121, 0xf, 0xf, true));
    return v;
}

// (region_mask: render_common.h — shared with the backward of render_bwd_rgn.hip)
// state of a row's pixel: T and live are the same in its sixteen lanes; C and the last contributor are per-lane partial results
struct RowLane {
    float T, live, C0, C1, C2, pxf, pyf, ef;   // ef: my slot in a step, 0 .. 15
};

// The walk of the region's list `mine[0 .. len)` (staged indices) of a staged batch: sixteen entries per step, one per lane of a row.
// -> staged index of this lane's last contributing entry as a float, -1 = none.
__device__ __forceinline__ float regions_walk(const StagedSplat *__restrict__ stage, const uint16_t *__restrict__ mine, const int len, const int e,
                                              RowLane &q, int &steps) {
    float lastf = -1.0f;
    const float lenf = (float)len - q.ef;   // (my position of a step that starts at t exists where lenf - t >= 1)
    for (int t = 0; t < len; t += 16 * RG_UNROLL) {
        if ((t & 63) == 0 && __ballot(q.live != 0.f) == 0ull) break;
        steps += min(RG_UNROLL, (len - t + 15) >> 4);
        int j[RG_UNROLL];
        float4 c[RG_UNROLL];
        float av[RG_UNROLL];
#pragma unroll
        for (int u = 0; u < RG_UNROLL; u++) j[u] = (int)mine[t + 16 * u + e];
#pragma unroll
        for (int u = 0; u < RG_UNROLL; u++) {   // (render_quad.h lanes_walk entry_alpha: the same arithmetic, bit for bit)
            const float4 p = stage[j[u]].xyh;
            const float4 co = stage[j[u]].co;
            c[u] = lds_read4(&stage[j[u]].rgbd);
            const float dx = p.x - q.pxf, dy = p.y - q.pyf;
            const float qq = __fmaf_rn(__fmul_rn(co.x, dx), dx, __fmul_rn(__fmul_rn(co.z, dy), dy));
            const float power = __fmaf_rn(-0.5f, qq, -__fmul_rn(__fmul_rn(co.y, dx), dy));
            const float a1 = fminf(fminf(0.99f, __fmul_rn(co.w, __expf(power))), lenf - (float)(t + 16 * u));
            av[u] = alpha_if_visible(a1, power);   // (a1 is not positive where the list has no such position)
        }
        // the two steps' products are scanned side by side (the second chain fills the first one's DPP wait states), both with the pixel's
        // `live` of the trip's start; a stop inside the first step — once in a pixel's life — makes the second step's scan stale for that
        // pixel: redone on the spot
        static_assert(RG_UNROLL == 2, "the walk scans two steps at a time");
        float a0 = av[0] * q.live, a1 = av[1] * q.live;
        float incl0 = 1.0f - a0, incl1 = 1.0f - a1;
        row_scan_mul_x2(incl0, incl1);   // prod (1 - alpha) of the row's lanes 0 .. mine, tree order
#define RG_STEP(U, A, INCL)                                                                                                        \
        {                                                                                                                           \
            const float excl = row_shift1(INCL, 1.0f);       /* lanes 0 .. mine - 1 */                                              \
            const float x = __fmul_rn(q.T, excl);            /* T in front of my entry */                                           \
            const float tn = __fmul_rn(q.T, INCL);           /* the reference's test_T of my entry (falls along the row) */         \
            float s = 1.0f, t_next = row_bcast<15>(tn), l_next = q.live;                                                            \
            if (__builtin_expect(__ballot(tn < 0.0001f) != 0ull, 0)) {   /* a pixel of this wave stops inside the step */          \
                s = tn < 0.0001f ? 0.f : 1.f;                            /* (test_T only falls: the entries behind the first failure fail too) */ \
                t_next = row_min(s != 0.f ? tn : q.T);                   /* T behind the last entry taken (the pixel's T where none is) */ \
                l_next = q.live * row_min(s);                                                                                       \
                stopped = true;                                                                                                     \
            }                                                                                                                       \
            const float w = A * s, wT = w * x;                                                                                      \
            q.C0 = __fmaf_rn(c[U].x, wT, q.C0);                                                                                     \
            q.C1 = __fmaf_rn(c[U].y, wT, q.C1);                                                                                     \
            q.C2 = __fmaf_rn(c[U].z, wT, q.C2);                                                                                     \
            lastf = max_raw(lastf, min_raw((float)j[U], __fmaf_rn(w, 1e30f, -1.0f)));                                               \
            q.T = t_next;                                                                                                           \
            q.live = l_next;                                                                                                        \
        }
        bool stopped = false;
        RG_STEP(0, a0, incl0)
        if (__builtin_expect(stopped, 0)) {   // (uniform) the second step with the pixels' new `live`
            a1 = av[1] * q.live;
            incl1 = row_scan_mul(1.0f - a1);
        }
        RG_STEP(1, a1, incl1)
#undef RG_STEP
    }
    return lastf;
}

__global__ void __launch_bounds__(RG_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) render_forward_regions_kernel(
    const uint2 *__restrict__ ranges, const uint32_t *__restrict__ point_list, int W, int H, int tiles_x, int ntiles_strip /*render_common.h pack_tiles*/,
    const float4 *__restrict__ xyh, const float4 *__restrict__ conic_opacity, const float4 *__restrict__ rgbd, const float *__restrict__ bg,
    float *__restrict__ final_T, uint32_t *__restrict__ n_contrib, float *__restrict__ out_color, const LocalBin lb,
    unsigned long long *__restrict__ pairs /*common.h pair_counters()*/) {
    __shared__ StagedSplat stage_all[2 * RG_BATCH];
    __shared__ uint16_t lists[16][RG_LIST];     // [wave][position]: staged index
    __shared__ uint32_t s_done[2][16];
    __shared__ uint16_t s_mask[2][RG_BATCH];   // per staged entry: the regions of this quadrant it can reach (bit = wave), written with the record
    // workgroup -> (tile, quadrant): the four quadrants of a tile are consecutive workgroups of ONE XCD (the dispatcher deals workgroup b to
    // XCD b % 8): b = (4 j' + quadrant) * 8 + xcd with tile slot 8 j' + xcd of render_common.h's XCD-aware tile order
    const int xcd = blockIdx.x & 7, jq = blockIdx.x >> 3, quad = jq & 3;
    int tile;
    if (lb.tile_order != nullptr) {   // (uniform) longest lists first: slot 8 j' + xcd of the order tile_lpt_kernel left (below)
        const int slot = ((jq >> 2) << 3) | xcd;
        if (slot >= packed_ntiles(ntiles_strip)) return;
        tile = (int)min(lb.tile_order[slot], (uint32_t)(packed_ntiles(ntiles_strip) - 1));
    } else {
        tile = xcd_tile(((jq >> 2) << 3) | xcd, ntiles_strip, tiles_x);
        if (tile < 0) return;
    }
    const int tid = threadIdx.x, lane = __lane_id(), wave = __builtin_amdgcn_readfirstlane(tid >> 6), e = lane & 15, row = lane >> 4;
    const int bx = tile % tiles_x, by = tile / tiles_x;
    const int rx0 = bx * TILE_X + ((quad & 1) << 3) + ((wave & 3) << 1), ry0 = by * TILE_Y + ((quad >> 1) << 3) + ((wave >> 2) << 1);   // the wave's region
    const int px = rx0 + (row & 1), py = ry0 + (row >> 1);
    const bool inside = px < W && py < H;
    const float qcx0 = (float)(bx * TILE_X + ((quad & 1) << 3)) + 0.5f, qcy0 = (float)(by * TILE_Y + ((quad >> 1) << 3)) + 0.5f;   // centre of the quadrant's region (0, 0)
    const uint2 range = safe_range(ranges[tile], lb.cap);
    const uint32_t n = range.y - range.x;
    const int rounds = (int)((n + RG_BATCH - 1) / RG_BATCH);
    RowLane q;
    q.T = 1.0f; q.live = inside ? 1.f : 0.f; q.C0 = q.C1 = q.C2 = 0.f;
    q.pxf = (float)px; q.pyf = (float)py; q.ef = (float)e;
    uint32_t last_contributor = 0;                                              // (this lane's entries; the row's maximum is the pixel's)
    const int nb = ckpt_buckets(range);
    const int cpix = ((py - by * TILE_Y) << 4) + (px - bx * TILE_X);
    int next_slot = 0;
    int steps = 0;
    for (int i = tid; i < 16 * RG_LIST; i += RG_THREADS) (&lists[0][0])[i] = 0;   // (a stale list word must name a staged entry)

    // staging as in render_lanes.hip: batch i + 1 is fetched behind the barrier of batch i and written in front of the wave's walk
    uint32_t g_ahead = 0u;
    const bool loader = tid < RG_BATCH;   // (uniform per wave)
    if (loader) {
        StagedSplat rec = null_splat();
        if ((uint32_t)tid < n) {
            const uint32_t g = min(point_list[range.x + tid], lb.last_g);
            rec.xyh = xyh[(size_t)g * SPLAT_REC];
            rec.co = conic_opacity[(size_t)g * SPLAT_REC];
            rec.rgbd = rgbd[(size_t)g * SPLAT_REC];
        }
        if ((uint32_t)(RG_BATCH + tid) < n) g_ahead = point_list[range.x + RG_BATCH + tid];
        stage_all[tid] = rec;
        s_mask[0][tid] = (uint16_t)region_mask(rec.xyh, qcx0, qcy0);
    }
    for (int i = 0; i < rounds; i++) {
        StagedSplat *const stage = stage_all + (i & 1) * RG_BATCH;
        const uint32_t first = (uint32_t)i * RG_BATCH;
        const bool wave_done = __ballot(q.live != 0.f) == 0ull;
        if (lane == 0) s_done[i & 1][wave] = wave_done ? 1u : 0u;
        lds_barrier();   // (the loads just issued stay in flight: render_common.h)
        {   // every pixel of the quadrant has stopped?
            const uint32_t d = s_done[i & 1][lane & 15];
            if (__ballot(d != 0u) == ~0ull) break;
        }
        if (nb > 1 && i > 0 && first % BUCKET == 0) {   // the state in front of list position `first`
            const float q0 = row_sum(q.C0), q1 = row_sum(q.C1), q2 = row_sum(q.C2);
            if (e == 0) ckpt_slot(lb.ckpt, range, tile, next_slot)[cpix] = make_float4(q.T, q0, q1, q2);
            next_slot++;
        }
        StagedSplat rec = null_splat();
        const uint32_t progress = range.x + first + RG_BATCH + tid;   // my entry of batch i + 1
        if (loader && i + 1 < rounds) {
            if (progress < range.y) {
                const uint32_t g = min(g_ahead, lb.last_g);
                rec.xyh = xyh[(size_t)g * SPLAT_REC];
                rec.co = conic_opacity[(size_t)g * SPLAT_REC];
                rec.rgbd = rgbd[(size_t)g * SPLAT_REC];
            }
            if (progress + RG_BATCH < range.y) g_ahead = point_list[progress + RG_BATCH];
        }
        // ---- this region's list of the batch -----------------------------------------------------------------------------------
        uint16_t *const mine = lists[wave];
        int len = 0;
        if (!wave_done) {
            const uint16_t *const masks = s_mask[i & 1];
#pragma unroll
            for (int c = 0; c < RG_BATCH / 64; c++) {
                const int s = c * 64 + lane;
                const bool hit = (((uint32_t)masks[s] >> wave) & 1u) != 0u;   // (entries past the list hold extents no region can meet: mask 0)
                const uint64_t m = __ballot(hit);
                const int at = len + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                if (hit) mine[at] = (uint16_t)s;
                len += __popcll(m);
            }
        }
        if (loader && i + 1 < rounds) {
            stage_all[((i + 1) & 1) * RG_BATCH + tid] = rec;
            s_mask[(i + 1) & 1][tid] = (uint16_t)region_mask(rec.xyh, qcx0, qcy0);
        }
        if (wave_done) continue;   // (uniform; the wave has staged its share and meets the barriers)
        const float lastf = regions_walk(stage, mine, len, e, q, steps);
        if (lastf >= 0.0f) last_contributor = first + (uint32_t)lastf + 1u;
    }
    const float q0 = row_sum(q.C0), q1 = row_sum(q.C1), q2 = row_sum(q.C2);
    const uint32_t last = row_max_u32(last_contributor);
    if (e == 0)
        for (; nb > 1 && next_slot < nb; next_slot++) ckpt_slot(lb.ckpt, range, tile, next_slot)[cpix] = make_float4(q.T, q0, q1, q2);   // (final values)
    if (inside && e == 0) {
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

// Two numbers about a forward's tile lists, delivered to the host (pinned mailbox: {longest, tag, crowding}) — what the choice between the
// compositing kernels for long lists is made from (api.hip): the LONGEST list (one workgroup per tile lasts as long as its longest tile), and
// how CROWDED a stretch of a list is on the screen: of 64 consecutive entries from the middle of every list of >= 256 entries, how many have
// their centre in the most popular 8x8 quadrant of the tile (mean over the tiles, in 1/64).  A tile's list is in depth order; on random depths
// the four quadrants share a stretch evenly (the fullest holds ~20 of 64), on the depth maps of a real sequence a depth slab is a band or a
// corner of the tile — the kernels with a wave per quadrant then leave three waves waiting for one at every barrier.  Integer sums in a fixed
// order: the same lists give the same numbers.  One workgroup; the lists are complete (this runs behind the last binning kernel on the stream).
__global__ void __launch_bounds__(256) list_skew_kernel(const uint2 *__restrict__ ranges, int ntiles, int tiles_x, uint32_t cap, const uint32_t *__restrict__ point_list,
                                                        const float4 *__restrict__ xyh, uint32_t last_g, uint32_t *__restrict__ out, uint32_t tag) {
    __shared__ uint32_t s_max[4], s_crowd[4], s_cnt[4];
    uint32_t m = 0, crowd = 0, cnt = 0;
    for (int t = threadIdx.x; t < ntiles; t += 256) {
        const uint2 r = safe_range(ranges[t], cap);
        const uint32_t n = r.y - r.x;
        m = max(m, n);
        if (n >= 256u) {
            const uint32_t at = r.x + n / 2u;
            const float cx = (float)((t % tiles_x) * TILE_X + 8), cy = (float)((t / tiles_x) * TILE_Y + 8);
            uint32_t q[4] = {0u, 0u, 0u, 0u};
            for (int k = 0; k < 64; k++) {
                const float4 b = xyh[(size_t)min(point_list[at + k], last_g) * SPLAT_REC];
                const bool right = b.x >= cx, low = b.y >= cy;
                q[0] += (!right && !low) ? 1u : 0u;
                q[1] += (right && !low) ? 1u : 0u;
                q[2] += (!right && low) ? 1u : 0u;
                q[3] += (right && low) ? 1u : 0u;
            }
            crowd += max(max(q[0], q[1]), max(q[2], q[3]));
            cnt += 1u;
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
        crowd += (uint32_t)__shfl_xor((int)crowd, o, 64);
        cnt += (uint32_t)__shfl_xor((int)cnt, o, 64);
    }
    if ((threadIdx.x & 63) == 0) { s_max[threadIdx.x >> 6] = m; s_crowd[threadIdx.x >> 6] = crowd; s_cnt[threadIdx.x >> 6] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t c = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3], d = s_crowd[0] + s_crowd[1] + s_crowd[2] + s_crowd[3];
        __hip_atomic_store(out, max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3])), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(out + 2, c ? (d * 16u) / c : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // (entries of 64 in the fullest quadrant, x 16)
        __hip_atomic_store(out + 1, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);   // (the values first)
    }
}

// The tiles in the order the region forward starts them: longest list first (ties: lower tile first — the same lists give the same order).
// One workgroup of 1024 threads ranks up to 1024 tiles by counting the keys above its own (broadcast LDS reads); it runs behind the last
// binning kernel of every forward that takes the region kernel (5 us), because the lists change with every view.
__global__ void __launch_bounds__(1024) tile_lpt_kernel(const uint2 *__restrict__ ranges, int ntiles, uint32_t cap, uint32_t *__restrict__ order) {
    __shared__ __attribute__((aligned(16))) uint32_t key[1024];
    const int t = threadIdx.x;
    uint32_t k = 0u;
    if (t < ntiles) {
        const uint2 r = safe_range(ranges[t], cap);
        k = (min(r.y - r.x, 0x3FFFFFu) << 10) | (uint32_t)(1023 - t);   // (distinct; larger = earlier)
    }
    key[t] = k;
    __syncthreads();
    if (t >= ntiles) return;
    uint32_t rank = 0u;
    for (int j = 0; j < ntiles; j += 4) {
        const uint4 v = *reinterpret_cast<const uint4 *>(&key[j]);   // (entries past ntiles hold 0: never above a live key)
        rank += (v.x > k ? 1u : 0u) + (v.y > k ? 1u : 0u) + (v.z > k ? 1u : 0u) + (v.w > k ? 1u : 0u);
    }
    order[rank] = (uint32_t)t;
}

int launch_tile_lpt(char *img, const Layout &L, uint32_t cap, bool debug, hipStream_t s) {
    DAS3R_LAUNCH(tile_lpt_kernel, dim3(1), dim3(1024), 0, s, (const uint2 *)(img + L.pub.ranges), L.ntiles, cap, (uint32_t *)(img + L.i_order));
    KERNEL_CHECK(s, debug, "tile_lpt");
    return DAS3R_OK;
}

int launch_list_skew(const char *img, const char *binning, const char *geom, const Layout &L, uint32_t cap, uint32_t last_g, uint32_t *mailbox_words, uint32_t tag,
                     bool debug, hipStream_t s) {
    DAS3R_LAUNCH(list_skew_kernel, dim3(1), dim3(256), 0, s, (const uint2 *)(img + L.pub.ranges), L.ntiles, L.tiles_x, cap, (const uint32_t *)(binning + L.pub.point_list),
                 (const float4 *)(geom + L.pub.xy), last_g, mailbox_words, tag);
    KERNEL_CHECK(s, debug, "list_skew");
    return DAS3R_OK;
}

int launch_render_forward_regions(const das3r_raster_args *a, float *out_color, char *geom, char *binning, char *img, const Layout &L, const LocalBin &lb,
                                  hipStream_t s) {
#define ARGS                                                                                                                                   \
    (const uint2 *)(img + L.pub.ranges), (const uint32_t *)(binning + L.pub.point_list), a->image_width, a->image_height, L.tiles_x, pack_tiles(L), \
        (const float4 *)(geom + L.pub.xy), (const float4 *)(geom + L.pub.conic_opacity), (const float4 *)(geom + L.pub.rgbd), a->bg,                \
        (float *)(img + L.pub.final_T), (uint32_t *)(img + L.pub.n_contrib), out_color, lb, pair_counters()
    DAS3R_LAUNCH(render_forward_regions_kernel, dim3(lb.tile_order ? 32 * div_up(L.ntiles, 8) : 4 * xcd_grid(L)), dim3(RG_THREADS), 0, s, ARGS);
#undef ARGS
    KERNEL_CHECK(s, a->debug, "render_forward_regions");
    return DAS3R_OK;
}

}  // namespace das3r
