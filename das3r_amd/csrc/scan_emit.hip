// scan_emit.hip — exclusive scan of tiles_touched in depth-rank order (instance offsets, num_rendered) as ONE kernel, and,
// fused behind it, the emission of the (tile id, emission slot) instances.  Replaces upstream's cub::DeviceScan::InclusiveSum
// + duplicateWithKeys (cuda_rasterizer/rasterizer_impl.cu) and this repo's earlier tt_blocksum -> tt_scan ->
// emit chain: at 100 k splats those three launches were 45 us of a 300 us step, almost all of it launch/drain latency.
//
// Chained scan: a workgroup takes a ticket (arrival order), sums its 4096 ranks, publishes the sum as a granule and gets
// its exclusive prefix from the granules of the earlier workgroups — two levels (own group + totals of earlier groups), 64
// lanes of one wave polling 64 words at a time, so the wait is ~3 round trips for any grid size.  Granules, tickets and the
// histograms are zeroed earlier in the same forward (preprocess_kernel / the binning memset).
#include "granule.h"
#include "segkey.h"
#include "splat_math.h"

namespace das3r {

constexpr int SCAN_GROUP_LOG2 = 6;     // group size 64 = one wave-wide poll

// sum of `count` (<= 64 per round) published granules starting at g[0]; executed by ONE wave, result in every lane
__device__ __forceinline__ uint32_t wave_sum_published(const u64 *g, const int count, uint32_t *err) {
    const int lane = __lane_id();
    uint32_t sum = 0;
    unsigned spins = 0;
    for (int p = 0; p < count; p += 64) {
        const bool mine = p + lane < count;
        u64 x = TAG_AGG;
        while (true) {
            if (mine) x = granule_poll(g + p + lane, spins);
            if (__all((x & TAG_MASK) != 0)) break;   // wave-uniform exit: every lane's word is published
            if (spins == SOFT_SPINS && lane == 0) atomicOr(err, ERR_HARD_POLL);
            if (++spins > SPIN_LIMIT) {
                if (lane == 0) atomicOr(err, ERR_TIMEOUT);
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        sum += mine ? (uint32_t)(x & 0xFFFFFFFFull) : 0u;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) sum += (uint32_t)__shfl_xor((int)sum, o, 64);
    return sum;
}

// SCAN_ITEMS = ranks per thread; a workgroup owns 256 * SCAN_ITEMS consecutive depth ranks
template <bool EMIT, int SCAN_ITEMS>
__global__ void __launch_bounds__(256) scan_emit_kernel(
    int P, const uint32_t *__restrict__ sorted_idx, const uint32_t *__restrict__ tiles_touched, uint32_t *__restrict__ offsets,
    uint32_t *__restrict__ off_by_gid, uint32_t *__restrict__ count, u64 *__restrict__ status, u64 *__restrict__ group_status,
    uint32_t *__restrict__ ticket, uint32_t *__restrict__ err, uint32_t *__restrict__ host_out /*pinned host mailbox*/, uint32_t tag,
    // emission (EMIT only)
    int tiles_x, int tiles_y, const float4 *__restrict__ xyh, const int32_t *__restrict__ radii, uint32_t *__restrict__ tile_keys,
    uint32_t *__restrict__ gids, uint32_t cap, uint32_t *__restrict__ ghist /*[passes][256]*/, int tbits /*bits of the partition key*/, int tight_rect,
    const uint32_t *__restrict__ rect32 /*packed binned rectangles by splat (common.h g_rect) or null*/,
    // segmented path (segkey.h; index order only): partition key = tile id << dbits | depth bucket of the splat
    int dbits, int fbits /*fraction bits below the bucket in the key: Layout.kshift*/, const uint32_t *__restrict__ depth_keys /*[P] by splat*/,
    const uint32_t *__restrict__ dhist /*[256] depth histogram of this forward*/
#ifdef DAS3R_EXPERIMENTS
    , unsigned long long *__restrict__ wg_trace_ptr /*common.h SCAN_STAMP*/
#endif
    ) {
    __shared__ uint32_t ws[4];
    __shared__ uint32_t s_block, s_carry;
    __shared__ uint32_t h[EMIT ? 4 : 1][RADIX_SIZE];
    // Emission through LDS (round 3): the instances of 256 consecutive ranks are one contiguous run of the output; written per lane
    // they are 4-byte stores at a stride of the splats' tile counts (r02 PMC: 96 MB of HBM traffic for 21 MB of instances).  The
    // run is assembled in LDS and leaves as unit-stride stores; a run longer than EMIT_LDS instances (huge splats) is written
    // directly as before.
    constexpr int EMIT_LDS = 3072;
    __shared__ uint32_t e_key[EMIT ? EMIT_LDS : 1], e_gid[EMIT ? EMIT_LDS : 1];
    const int tid = threadIdx.x, lane = __lane_id(), wave = tid >> 6;
    // order of the chained scan: a ticket (arrival order) in general; the workgroup index when the whole grid is resident at
    // once (ticket == null: nobody can wait for a workgroup that has not started) — the returned atomic is ~2 us of every
    // workgroup's critical path, a tenth of this kernel at 100 k splats
    if (tid == 0) s_block = ticket ? atomicAdd(ticket, 1u) : blockIdx.x;
    if (EMIT) {
#pragma unroll
        for (int q = 0; q < 4; q++) h[EMIT ? q : 0][tid] = 0;
    }
    __syncthreads();
    const uint32_t b = s_block;
    SCAN_STAMP(0)   // ticket
    const int base = (int)b * 256 * SCAN_ITEMS;
    // depth buckets (segkey.h): every workgroup turns the forward's 256-bin depth histogram into the same monotone map
    __shared__ float s_cnt[EMIT ? DBINS : 1], s_cdf[EMIT ? DBINS : 1];
    float dscale = 0.f;
    if (EMIT && dbits > 0) {   // (uniform)
        const uint32_t c = dhist[tid];
        uint32_t tot, tot2;
        block_exclusive_scan_256(c, ws, &tot);
        int sh = 0;
        while ((tot >> sh) >= (1u << 24)) sh++;   // counts below 2^24: exact as floats
        const uint32_t cs = c >> sh;
        const uint32_t ex = block_exclusive_scan_256(cs, ws, &tot2);
        s_cnt[EMIT ? tid : 0] = (float)cs;
        s_cdf[EMIT ? tid : 0] = (float)ex;
        dscale = tot2 ? (float)(1u << (dbits + fbits)) / (float)tot2 : 0.f;   // (bucket + fraction: segkey.h)
        __syncthreads();
    }

    uint32_t g[SCAN_ITEMS], v[SCAN_ITEMS], rc[SCAN_ITEMS], dkv[SCAN_ITEMS], sum = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        const int r = base + k * 256 + tid;
        g[k] = r < P ? (sorted_idx ? sorted_idx[r] : (uint32_t)r) : 0u;   // sorted_idx == null: index order (local depth order)
        dkv[k] = (EMIT && dbits > 0 && r < P) ? depth_keys[g[k]] : 0u;    // (segmented path: requested with everything else, up front)
    }
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        const int r = base + k * 256 + tid;
        // In depth order everything the scan and the emission need of a splat is its binned rectangle: ONE random 4-byte read
        // of the packed array (a 32-byte sector) instead of the first and the last 16 bytes of its 64-byte record (r02: the
        // 5 M-splat DAS3R-shaped scene spent 0.31 ms here).  Without the array (more than 255 tiles per axis): tiles_touched
        // from the record.  In index order: the plain array, coalesced.
        rc[k] = 0u;
        if (r < P && rect32) {   // (in index order too: 4 coalesced bytes per splat instead of its 64-byte record for the emission)
            rc[k] = rect32[g[k]];
            v[k] = (((rc[k] >> 8) & 255u) - (rc[k] & 255u)) * ((rc[k] >> 24) - ((rc[k] >> 16) & 255u));
        } else {
            v[k] = r < P ? (sorted_idx ? __float_as_uint(xyh[(size_t)g[k] * SPLAT_REC + 3].y) : tiles_touched[r]) : 0u;
        }
        sum += v[k];
    }
    uint32_t total;
    block_exclusive_scan_256(sum, ws, &total);
    SCAN_STAMP(1)   // rectangles read, workgroup total known

    if (wave == 0) {   // one wave publishes and looks back for the whole workgroup
        const uint32_t grp = b >> SCAN_GROUP_LOG2, r = b & ((1u << SCAN_GROUP_LOG2) - 1u);
        if (lane == 0) granule_store(status + b, TAG_AGG | total);
        const uint32_t in_group = wave_sum_published(status + (b - r), (int)r, err);
        if (r == (1u << SCAN_GROUP_LOG2) - 1u && lane == 0) granule_store(group_status + grp, TAG_AGG | (u64)(in_group + total));
        const uint32_t before = wave_sum_published(group_status, (int)grp, err);
        if (lane == 0) s_carry = in_group + before;
    }
    __syncthreads();
    uint32_t carry = s_carry;
    SCAN_STAMP(2)   // look-back done

#pragma unroll 1
    for (int k = 0; k < SCAN_ITEMS; k++) {
        const int r = base + k * 256 + tid;
        uint32_t tot;
        const uint32_t ex = block_exclusive_scan_256(v[k], ws, &tot);
        const bool via_lds = EMIT && tot <= (uint32_t)EMIT_LDS;   // (uniform)
        if (r < P) {
            uint32_t o = carry + ex;
            offsets[r] = o;
            off_by_gid[g[k]] = o;   // first emission slot of splat g (its instances are emitted contiguously)
            if (EMIT && v[k] != 0u) {
                int rminx, rminy, rmaxx, rmaxy;
                if (rect32) {
                    rminx = (int)(rc[k] & 255u), rmaxx = (int)((rc[k] >> 8) & 255u), rminy = (int)((rc[k] >> 16) & 255u), rmaxy = (int)(rc[k] >> 24);
                } else {
                    const float4 p = xyh[(size_t)g[k] * SPLAT_REC];
                    const int radius = __float_as_int(xyh[(size_t)g[k] * SPLAT_REC + 3].x);
                    binned_rect(p, radius, tiles_x, tiles_y, tight_rect != 0, rminx, rminy, rmaxx, rmaxy);
                }
                uint32_t l = ex;   // place in the workgroup's run
                // segmented path: key = tile id | bucket | 16 bits of fraction (kshift = 16: the partition's digits start above the fraction)
                const int kshift = (EMIT && dbits > 0) ? fbits : 0;
                const uint32_t bucket = (EMIT && dbits > 0) ? depth_bucket(dkv[k], s_cnt, s_cdf, dscale, 1u << (dbits + fbits)) : 0u;
                for (int y = rminy; y < rmaxy; y++)
                    for (int x = rminx; x < rmaxx; x++) {
                        const uint32_t t = ((uint32_t)(y * tiles_x + x) << (dbits + kshift)) | bucket;
                        if (via_lds) {
                            e_key[EMIT ? l : 0] = t;
                            e_gid[EMIT ? l : 0] = g[k];
                        } else if (o < cap) {   // cap < num_rendered only when the capacity hint was too small (the binning is then redone)
                            tile_keys[o] = t;
                            gids[o] = g[k];   // gid_of[emission slot]
                            for (int q = 0, sh = 0, dw = tile_digit_width(tbits); sh < tbits; q++, sh += dw) {
                                const int bits = (tbits - sh) < dw ? (tbits - sh) : dw;
                                atomicAdd(&h[EMIT ? q : 0][(t >> (sh + kshift)) & ((1u << bits) - 1u)], 1u);
                            }
                        }
                        o++;
                        l++;
                    }
            }
        }
        if (via_lds) {
            __syncthreads();
            const int kshift2 = (EMIT && dbits > 0) ? fbits : 0;
            for (uint32_t l = tid; l < tot; l += 256u) {   // unit stride: tile ids and splat ids of the run [carry, carry + tot)
                const uint32_t o = carry + l;
                if (o < cap) {
                    const uint32_t t = e_key[EMIT ? l : 0];
                    tile_keys[o] = t;
                    gids[o] = e_gid[EMIT ? l : 0];
                    for (int q = 0, sh = 0, dw = tile_digit_width(tbits); sh < tbits; q++, sh += dw) {
                        const int bits = (tbits - sh) < dw ? (tbits - sh) : dw;
                        atomicAdd(&h[EMIT ? q : 0][(t >> (sh + kshift2)) & ((1u << bits) - 1u)], 1u);
                    }
                }
            }
            __syncthreads();   // the staging arrays are free for the next 256 ranks
        }
        carry += tot;
    }
    SCAN_STAMP(3)   // offsets and instances written (stores issued)
    if (b == gridDim.x - 1 && tid == 0) {
        const uint32_t flags = *err;   // look-back timeout flags of the depth sort / this scan travel with the count
        count[0] = carry;
        count[1] = flags;
        // the host polls a pinned mailbox instead of waiting for a copy + event: count and flags first, then the tag of this
        // call with system-scope release semantics
        if (host_out) {
            host_out[0] = carry;
            host_out[1] = flags;
            __hip_atomic_store(host_out + 2, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if (EMIT) {
        __syncthreads();
        for (int q = 0, sh = 0, dw = tile_digit_width(tbits); sh < tbits; q++, sh += dw) {
            const uint32_t c = h[EMIT ? q : 0][tid];
            if (c) atomicAdd(&ghist[q * RADIX_SIZE + tid], c);
        }
    }
#ifdef DAS3R_EXPERIMENTS
    if (wg_trace_ptr != nullptr) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        SCAN_STAMP(4)   // everything acknowledged
    }
#endif
}

// (Round 3, tried and dropped: BLOCKED ranks — thread t owns SCAN_ITEMS consecutive ranks, one block-wide scan, the workgroup's whole
//  run assembled in one LDS window — instead of a scan + emission + write-out per stripe of 256 ranks: 39 -> 35 us at 1 M splats
//  (tools/wg_trace.py: emission 20.7 -> 16.5 us of a workgroup's 31), but 157 -> 177 us on the 5 M-splat DAS3R shape.  Counting the
//  digit histograms with ballot matches + wave-private counters instead of LDS integer atomics: 35 -> 44 us.)
// ranks per thread: as many as keeps >= 256 workgroups in flight (the emission loop is the long pole, it wants parallelism;
// the chain wants few workgroups), between 1 and 8.  (r3, scan + emission, ms: 1 M splats 2 / 4 / 8 / 16 ranks per thread 0.067 /
// 0.046 / 0.039 / 0.056; 2 M 0.091 (4) / 0.080 (8) / 0.112 (16); 5 M in depth order 0.171 (4) / 0.157 (8) / 0.196 (16): sixteen
// leaves one workgroup per CU at these sizes — nobody covers its trips to memory.)
static inline int scan_items(int P) {
    if (switches().scan_items) return switches().scan_items;   // DAS3R_SCAN_ITEMS (A-B runs)
    int it = 1;
    while (it < 8 && (int64_t)P >= (int64_t)256 * 256 * (it * 2)) it *= 2;
    return it;
}
static inline int scan_blocks(int P) { return div_up(P > 0 ? P : 1, 256 * scan_items(P)); }

size_t scan_status_bytes(int P) {
    const int nblocks = scan_blocks(P);
    return (size_t)(nblocks + div_up(nblocks, 1 << SCAN_GROUP_LOG2)) * sizeof(u64);
}

#ifdef DAS3R_EXPERIMENTS
#define SCAN_TRACE_ARG , wg_trace()
#else
#define SCAN_TRACE_ARG
#endif
#define RECT32 ((L.tiles_x <= 255 && L.tiles_y <= 255) ? (const uint32_t *)(geom + L.g_rect) : (const uint32_t *)nullptr)
#define SCAN_COMMON                                                                                                           \
    P, order, (const uint32_t *)(geom + L.pub.tiles_touched),                          \
        (uint32_t *)(geom + L.pub.offsets), (uint32_t *)(geom + L.g_off_by_gid), (uint32_t *)(geom + L.g_count),             \
        (u64 *)(geom + L.g_scan_status), (u64 *)(geom + L.g_scan_status) + nblocks,                                    \
        grid_is_resident(nblocks) ? (uint32_t *)nullptr : (uint32_t *)(geom + L.g_ticket) + 16,   \
        (uint32_t *)(geom + L.g_ticket) + 8, host_out, tag

int launch_scan(int P, char *geom, const Layout &L, uint32_t *host_out, uint32_t tag, bool debug, hipStream_t s, bool index_order) {
    const int nblocks = scan_blocks(P);
    const uint32_t *order = index_order ? nullptr : (const uint32_t *)(geom + L.pub.sorted_idx);
#define GO(IT)                                                                                                               \
    DAS3R_LAUNCH((scan_emit_kernel<false, IT>), dim3(nblocks), dim3(256), 0, s, SCAN_COMMON, 0, 0,                            \
                 (const float4 *)(geom + L.pub.xy),                                                                           \
                 (const int32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, 0u, (uint32_t *)nullptr, 0, 0, RECT32, 0, 0,           \
                 (const uint32_t *)nullptr, (const uint32_t *)nullptr SCAN_TRACE_ARG)
    switch (scan_items(P)) { case 1: GO(1); break; case 2: GO(2); break; case 4: GO(4); break; case 8: GO(8); break; default: GO(16); }
#undef GO
    KERNEL_CHECK(s, debug, "scan");
    return DAS3R_OK;
}

int launch_scan_emit(int P, int64_t cap, const int32_t *radii, char *geom, char *binning, const Layout &L, uint32_t *host_out, uint32_t tag,
                     bool debug, hipStream_t s, bool index_order) {
    const int nblocks = scan_blocks(P);
    const uint32_t *order = index_order ? nullptr : (const uint32_t *)(geom + L.pub.sorted_idx);
#define GO(IT)                                                                                                               \
    DAS3R_LAUNCH((scan_emit_kernel<true, IT>), dim3(nblocks), dim3(256), 0, s, SCAN_COMMON, L.tiles_x, L.tiles_y,              \
                 (const float4 *)(geom + L.pub.xy), radii, (uint32_t *)(binning + L.b_keyA), (uint32_t *)(binning + L.b_gid_of), \
                 (uint32_t)cap, (uint32_t *)(binning + L.b_ghist), L.kbits, use_tight_rect() ? 1 : 0, RECT32, index_order ? L.dbits : 0, L.kshift, \
                 (const uint32_t *)(geom + L.pub.depth_key), L.dhist_ptr ? L.dhist_ptr : (const uint32_t *)(geom + L.g_dhist) SCAN_TRACE_ARG)
    switch (scan_items(P)) { case 1: GO(1); break; case 2: GO(2); break; case 4: GO(4); break; case 8: GO(8); break; default: GO(16); }
#undef GO
    KERNEL_CHECK(s, debug, "scan_emit");
    return DAS3R_OK;
}

}  // namespace das3r
