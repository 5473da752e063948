// segsort.hip — the per-tile sort of the segmented binning path (round 4): every (tile, depth bucket) segment of
// the partitioned instance list is put into exact (depth bits, index) order inside LDS; the keys never leave the CU.
// Replaces, for scenes with long tile lists, the global depth sort of the P splats in front of the tile partition
// (depth_hist + 4 onesweep passes; upstream: the upper 32 bits of cub::DeviceRadixSort's 64-bit keys).
// See segkey.h for how the segments come about.
//
// A workgroup owns the segments that START inside its window of SEG_CH list positions and holds them — the window plus up to
// SEG_OVER positions of overhang — in LDS.  Segment boundaries come from the partition keys (flags + one block-wide running
// maximum).  The key's low 16 bits are the FRACTION of the depth map below the bucket (segkey.h): monotone in the depth, so an
// entry's rank inside its segment is the number of (fraction << 12 | position) words below its own — positions follow the index
// order, the partition passes being stable — counted with broadcast LDS reads (neighbouring lanes sit in the same segment); only
// where two entries of a segment share a fraction (about one segment in ten on the benchmark) are their exact depth bits
// fetched and the tie group re-ranked by (depth bits, position).  The (splat id, emission slot) pairs are stored at their ranks,
// in place: a segment is read and written by its owner only.
// A segment that does not end inside the LDS span (a wall parallel to the image plane: thousands of equal depths in one tile) is
// sorted by the same workgroup with a bitonic network over global memory — slow, exact — and the host is told through the mailbox
// word the local depth order uses, so that the next forwards of this shape take the global sort for a while (api.hip).
#include "granule.h"
#include "segkey.h"

namespace das3r {

constexpr int SEG_CH = 2048;                  // list positions whose segment starts a workgroup owns
constexpr int SEG_OVER = 1024;                // overhang: an owned segment may reach this far past the window
constexpr int SEG_CAP = SEG_CH + SEG_OVER;    // entries in LDS
constexpr int SEG_T = 512;                    // threads per workgroup: the kernel is a chain of short phases between barriers and trips to
                                              // memory, and LDS lets only four workgroups share a CU — eight waves each hide more of it than four
                                              // (0.115 -> see DESIGN.md on the 5 M-splat DAS3R shape)
constexpr int SEG_PAIRS = SEG_CAP / (2 * SEG_T);      // pairs of neighbouring entries per thread
constexpr int SEG_BLK = (SEG_CAP + 1 + SEG_T - 1) / SEG_T;   // flag positions per thread (blocked), SEG_CAP + 1 of them
constexpr uint32_t SEG_NONE = 0xFFFFu;

// exclusive running maximum across the SEG_T threads of (v + 1) style values (0 = none)
__device__ __forceinline__ uint32_t block_exclusive_max(const uint32_t v, uint32_t *ws /*[SEG_T / 64]*/) {
    const int lane = __lane_id(), wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t nb = __shfl_up(incl, o, 64);
        if (lane >= o) incl = max(incl, nb);
    }
    uint32_t ex = __shfl_up(incl, 1, 64);
    if (lane == 0) ex = 0u;
    __syncthreads();
    if (lane == 63) ws[wave] = incl;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < SEG_T / 64; w++)
        if (w < wave) ex = max(ex, ws[w]);
    return ex;
}

__device__ __forceinline__ void count_below(uint32_t &rank, const uint32_t kj, const uint32_t ki) {
    // rank += [kj < ki]: the borrow of kj - ki, added with carry — two full-rate instructions (v_cmp + v_cndmask issue at half rate)
    uint32_t tmp;
    asm("v_sub_co_u32 %1, vcc, %2, %3\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(rank), "=&v"(tmp) : "v"(kj), "v"(ki) : "vcc");
}

constexpr uint32_t SEG_WANT_MORE = 384;   // ranking is quadratic in a segment's length: a workgroup whose owned segments average more than this
                                          // (weighted by their length: sum len^2 > SEG_CH * this) asks the host for more bucket bits
                                          // (mailbox word host_flag + 1) — one long segment among short ones does not
__global__ void __launch_bounds__(SEG_T) segment_sort_kernel(uint32_t cap, const uint32_t *__restrict__ n_ptr, const uint32_t *__restrict__ kk /*partition keys, final order*/,
                                                             int fbits /*fraction bits below (tile, bucket) in a key*/,
                                                           uint32_t *__restrict__ pl, uint32_t *__restrict__ sl, const uint32_t *__restrict__ depth_key /*[P] by splat*/,
                                                           uint32_t last_g, uint32_t *__restrict__ dk /*u32[cap] scratch (the dead key buffer)*/,
                                                           uint32_t *__restrict__ host_flag, uint32_t flag_value,
                                                           // round 6: the tile ranges and the forward's self-check word, which tile_ranges_kernel used to produce in a
                                                           // launch of its own behind this one, from the keys this kernel holds anyway
                                                           int dbits, uint2 *__restrict__ ranges, const uint32_t *__restrict__ err, uint32_t *__restrict__ host_late,
                                                           uint32_t tag, uint32_t inject) {
    const uint32_t n = n_ptr ? min(*n_ptr, cap) : cap;
    const uint32_t w0 = blockIdx.x * (uint32_t)SEG_CH;
    if (blockIdx.x == 0 && threadIdx.x == 0 && host_late) {   // last binning kernel: hand the self-check word of this forward to the host mailbox (every
        host_late[0] = *err | inject;                         // kernel that raises bits in it has finished; this one raises none)
        __hip_atomic_store(host_late + 1, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (w0 >= n) return;
    __shared__ uint32_t s_a[SEG_CAP + 2];       // the partition keys of positions w0 - 1 .. w0 + SEG_CAP; then the sort words by position; then tie depths by rank
    __shared__ uint32_t s_sorted[SEG_CAP];      // the sort words at their ranks
    __shared__ uint16_t s_start[SEG_CAP + 1], s_end[SEG_CAP + 1];
    __shared__ uint32_t ws[SEG_T / 64];
    __shared__ uint32_t s_long;                 // start (relative) of the owned segment that leaves the LDS span, or SEG_NONE
    __shared__ uint16_t s_defer[SEG_CAP / 2];   // deferred second entries of straddling pairs: position, then rank (the owner keeps its list slot)
    __shared__ uint32_t s_ndefer, s_work;
    const int tid = threadIdx.x;
    if (tid == 0) s_ndefer = s_work = 0u;
    for (int r = tid; r < SEG_CAP + 2; r += SEG_T) {
        const long long idx = (long long)w0 - 1 + r;
        s_a[r] = (idx >= 0 && idx < (long long)n) ? kk[idx] : 0u;   // s_a[r] = kk[w0 - 1 + r]
    }
    for (int r = tid; r < SEG_CAP + 1; r += SEG_T) s_end[r] = (uint16_t)SEG_NONE;
    if (tid == 0) s_long = SEG_NONE;
    // the (splat id, emission slot) pairs of every position of the span are requested NOW, whoever turns out to own them: they arrive
    // while the boundaries are found (a workgroup's life is a chain of trips to memory and barriers, and only four fit a CU)
    uint32_t g[SEG_PAIRS][2], slot[SEG_PAIRS][2];
#pragma unroll
    for (int u = 0; u < SEG_PAIRS; u++) {
        const int p0 = 2 * tid + 2 * SEG_T * u;   // (8-byte loads: w0 and p0 are even; an odd n leaves the last word unread)
        uint2 gv = make_uint2(0u, 0u), sv = make_uint2(0u, 0u);
        if (w0 + (uint32_t)p0 + 1u < n) {
            gv = *reinterpret_cast<const uint2 *>(pl + w0 + p0);
            sv = *reinterpret_cast<const uint2 *>(sl + w0 + p0);
        } else if (w0 + (uint32_t)p0 < n) {
            gv.x = pl[w0 + p0];
            sv.x = sl[w0 + p0];
        }
        g[u][0] = min(gv.x, last_g); g[u][1] = min(gv.y, last_g);
        slot[u][0] = sv.x; slot[u][1] = sv.y;
    }
    __syncthreads();
    // tile ranges (identifyTileRanges): list position w0 + p of the window starts a tile iff its tile id differs from the one before it
    for (int p = tid; p < SEG_CH; p += SEG_T) {
        const uint32_t idx = w0 + (uint32_t)p;
        if (idx < n) {
            const uint32_t t = s_a[p + 1] >> (fbits + dbits);
            if (idx == 0u) ranges[t].x = 0u;
            else {
                const uint32_t prev = s_a[p] >> (fbits + dbits);
                if (prev != t) {
                    ranges[prev].y = idx;
                    ranges[t].x = idx;
                }
            }
            if (idx == n - 1u) ranges[t].y = n;
        }
    }
    // position p (0 .. SEG_CAP) starts a segment iff its (tile, bucket) — the key above the 16 fraction bits — differs from the one
    // before it; the position just behind the list counts as a start (it ends the last segment), nothing beyond it does
    auto flag = [&](const int p) -> bool {
        const uint32_t idx = w0 + (uint32_t)p;
        if (idx > n) return false;
        if (idx == n || idx == 0u) return true;
        return (s_a[p] >> fbits) != (s_a[p + 1] >> fbits);
    };
    uint32_t last = 0u;   // (last start of this thread's positions) + 1, 0 = none
#pragma unroll
    for (int u = 0; u < SEG_BLK; u++) {
        const int p = tid * SEG_BLK + u;
        if (p <= SEG_CAP && flag(p)) last = (uint32_t)p + 1u;
    }
    uint32_t running = block_exclusive_max(last, ws);
#pragma unroll
    for (int u = 0; u < SEG_BLK; u++) {
        const int p = tid * SEG_BLK + u;
        if (p <= SEG_CAP) {
            if (flag(p)) {
                if (running) s_end[running - 1u] = (uint16_t)p;   // the segment before this one ends here
                running = (uint32_t)p + 1u;
            }
            s_start[p] = running ? (uint16_t)(running - 1u) : (uint16_t)SEG_NONE;
        }
    }
    // the sort words (fraction << 12 | position), in place of the keys: read, barrier, write
    uint32_t word[SEG_CAP / SEG_T];
#pragma unroll
    for (int u = 0; u < SEG_CAP / SEG_T; u++) {
        const int p = tid + SEG_T * u;
        word[u] = ((s_a[p + 1] & ((1u << fbits) - 1u)) << 12) | (uint32_t)p;
    }
    __syncthreads();   // (flags and keys read by everybody)
#pragma unroll
    for (int u = 0; u < SEG_CAP / SEG_T; u++) s_a[tid + SEG_T * u] = word[u];
    // owned in LDS: a true start inside the window whose end is known.  Thread t holds the PAIRS of neighbouring positions
    // 2 t + 512 u, + 1 (u < SEG_PAIRS): neighbours nearly always share their segment, and then every word of it is read from LDS once
    // for both of them
    uint32_t dest[SEG_PAIRS][2];
    bool own[SEG_PAIRS][2];
#pragma unroll
    for (int u = 0; u < SEG_PAIRS; u++)
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const int p = 2 * tid + 2 * SEG_T * u + e;
            const uint32_t s = s_start[p];
            const bool cand = w0 + (uint32_t)p < n && s != SEG_NONE && s < (uint32_t)SEG_CH;
            own[u][e] = cand && s_end[s] != SEG_NONE;
            if (cand && s_end[s] == SEG_NONE && (uint32_t)p == s) s_long = s;   // (one writer at most)
            if (cand && (uint32_t)p == s && s_end[s] != SEG_NONE) {
                const uint32_t len = (uint32_t)s_end[s] - s;
                if (len > 64u) atomicAdd(&s_work, len * len);
            }
            dest[u][e] = 0u;
        }
    __syncthreads();   // the sort words are in place, s_long is published, every load of pl / sl has arrived (the stores below overwrite them)
    if (tid == 0 && s_work > (uint32_t)SEG_CH * SEG_WANT_MORE)   // (still sorted here, exactly — but by rank, and that is quadratic)
        __hip_atomic_store(host_flag + 1, flag_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // A pair that straddles a segment boundary (one pair in ~50, i.e. most waves hold one) would send its whole wave through a second,
    // nearly empty pass over a segment per straddler: its second entry is DEFERRED instead — noted in LDS, and ranked afterwards by
    // a compacted pass, one deferred entry per lane (r4: 0.22 -> see DESIGN.md on the 5 M-splat DAS3R shape).
    uint32_t didx[SEG_PAIRS];
#pragma unroll
    for (int u = 0; u < SEG_PAIRS; u++) {   // (unrolled: own / g / slot / dest are registers)
        const int p0 = 2 * tid + 2 * SEG_T * u;
        didx[u] = 0xFFFFFFFFu;
        if (!(own[u][0] || own[u][1])) continue;
        const bool both = own[u][0] && own[u][1];
        const uint32_t s0 = own[u][0] ? s_start[p0] : s_start[p0 + 1];
        const bool straddle = both && s_start[p0 + 1] != s0;
        if (straddle) {
            didx[u] = atomicAdd(&s_ndefer, 1u);
            s_defer[didx[u]] = (uint16_t)(p0 + 1);
        }
        const uint32_t t = s_end[s0];
        const uint32_t k0 = s_a[own[u][0] ? p0 : p0 + 1], k1 = s_a[(own[u][1] && !straddle) ? p0 + 1 : (own[u][0] ? p0 : p0 + 1)];
        uint32_t r0 = 0, r1 = 0, j = s0;
#if defined(SEG_ABL) && (SEG_ABL & 1)   // (timing experiment: no rank loop; results are wrong)
        j = t; r0 = (own[u][0] ? p0 : p0 + 1) - s0; r1 = r0 + 1;
#endif
        for (; j + 8 <= t; j += 8) {   // eight independent LDS reads in flight
            uint32_t a[8];
#pragma unroll
            for (int q = 0; q < 8; q++) a[q] = s_a[j + q];
#pragma unroll
            for (int q = 0; q < 8; q++) { count_below(r0, a[q], k0); count_below(r1, a[q], k1); }
        }
        for (; j < t; j++) {
            const uint32_t a = s_a[j];
            count_below(r0, a, k0); count_below(r1, a, k1);
        }
        dest[u][0] = s0 + r0;
        dest[u][1] = s0 + r1;   // (a deferred entry's comes from the pass below)
        if (own[u][0]) s_sorted[dest[u][0]] = s_a[p0];
        if (own[u][1] && !straddle) s_sorted[dest[u][1]] = s_a[p0 + 1];
    }
    __syncthreads();
    {   // the deferred entries, one per lane
        const uint32_t nd = s_ndefer;
        for (uint32_t i = tid; i < nd; i += (uint32_t)SEG_T) {
            const uint32_t p = s_defer[i], s1 = s_start[p], t = s_end[s1], k = s_a[p];
            uint32_t r = 0, j = s1;
            for (; j + 8 <= t; j += 8) {
                uint32_t a[8];
#pragma unroll
                for (int q = 0; q < 8; q++) a[q] = s_a[j + q];
#pragma unroll
                for (int q = 0; q < 8; q++) count_below(r, a[q], k);
            }
            for (; j < t; j++) count_below(r, s_a[j], k);
            s_sorted[s1 + r] = k;
            s_defer[i] = (uint16_t)(s1 + r);   // (read back by the entry's owner below)
        }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < SEG_PAIRS; u++)
        if (didx[u] != 0xFFFFFFFFu) dest[u][1] = s_defer[didx[u]];
    // ---- ties: two entries of a segment with the same fraction need their exact depth bits ----
    bool tied[SEG_PAIRS][2];
    int any = 0;
#pragma unroll
    for (int u = 0; u < SEG_PAIRS; u++)
#pragma unroll
        for (int e = 0; e < 2; e++) {
            tied[u][e] = false;
            if (own[u][e]) {
                const int p = 2 * tid + 2 * SEG_T * u + e;
                const uint32_t s = s_start[p], t = s_end[s], at = dest[u][e], f = s_sorted[at] >> 12;
                tied[u][e] = (at > s && (s_sorted[at - 1u] >> 12) == f) || (at + 1u < t && (s_sorted[at + 1u] >> 12) == f);
                any |= tied[u][e] ? 1 : 0;
            }
        }
    if (__syncthreads_or(any)) {   // (uniform; rare)
#pragma unroll
        for (int u = 0; u < SEG_PAIRS; u++)
#pragma unroll
            for (int e = 0; e < 2; e++)
                if (tied[u][e]) s_a[dest[u][e]] = depth_key[g[u][e]];   // (the sort words by position are dead: the array takes the depths, by rank)
        __syncthreads();
#pragma unroll
        for (int u = 0; u < SEG_PAIRS; u++)
#pragma unroll
            for (int e = 0; e < 2; e++)
                if (tied[u][e]) {
                    const int p = 2 * tid + 2 * SEG_T * u + e;
                    const uint32_t s = s_start[p], t = s_end[s], at = dest[u][e], f = s_sorted[at] >> 12, mine = s_a[at];
                    uint32_t lo = at, hi = at + 1u;
                    while (lo > s && (s_sorted[lo - 1u] >> 12) == f) lo--;
                    while (hi < t && (s_sorted[hi] >> 12) == f) hi++;
                    uint32_t r = 0;   // the group [lo, hi) in (depth bits, position) order
                    for (uint32_t k = lo; k < hi; k++) {
                        const uint32_t dkk = s_a[k], pk = s_sorted[k] & 0xFFFu;
                        r += (dkk < mine || (dkk == mine && pk < (uint32_t)p)) ? 1u : 0u;
                    }
                    dest[u][e] = lo + r;
                }
    }
#pragma unroll
    for (int u = 0; u < SEG_PAIRS; u++)
#pragma unroll
        for (int e = 0; e < 2; e++)
            if (own[u][e]) {
#if defined(SEG_ABL) && (SEG_ABL & 2)   // (timing experiment: stores only if something impossible holds)
                if (g[u][e] != 0xFFFFFFFEu) continue;
#endif
                pl[w0 + dest[u][e]] = g[u][e];
                sl[w0 + dest[u][e]] = slot[u][e];
            }
    // ---- a segment that leaves the LDS span (rare; uniform branch) ----
    const uint32_t sl0 = s_long;
    if (sl0 == SEG_NONE) return;
    const uint32_t S = w0 + sl0;
    const uint32_t key = kk[S] >> fbits;
    __shared__ uint32_t s_T;
    if (tid == 0) {
        s_T = n;
        __hip_atomic_store(host_flag, flag_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();
    for (uint32_t base = w0 + (uint32_t)SEG_CAP; base < n; base += (uint32_t)SEG_T) {   // its end: the first position with another key
        const uint32_t idx = base + tid;
        if (idx < n && (kk[idx] >> fbits) != key) atomicMin(&s_T, idx);
        __syncthreads();
        if (s_T != n) break;   // (uniform: read behind the barrier)
        __syncthreads();
    }
    __syncthreads();
    const uint32_t T = s_T;
    const int m = (int)(T - S);
    uint32_t *dkk = dk + S, *plk = pl + S, *slk = sl + S;
    for (int i = tid; i < m; i += SEG_T) dkk[i] = depth_key[min(plk[i], last_g)];
    __syncthreads();
    for (int k = 2; (k >> 1) < m; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            const int flip = (j == (k >> 1)) ? k - 1 : j;
            for (int i = tid; i < m; i += SEG_T) {
                const int q = i ^ flip;
                if (q > i && q < m) {
                    const uint32_t da = dkk[i], db = dkk[q], ga = plk[i], gb = plk[q];
                    if (da > db || (da == db && ga > gb)) {   // (within a segment the index order is the splat-id order)
                        dkk[i] = db; dkk[q] = da;
                        plk[i] = gb; plk[q] = ga;
                        const uint32_t t = slk[i];
                        slk[i] = slk[q];
                        slk[q] = t;
                    }
                }
            }
            __syncthreads();   // same workgroup, same CU: its write-through L1 keeps the exchanged words coherent
        }
}

int launch_segment_sort(int64_t cap, const uint32_t *n_ptr, const uint32_t *keys_final, int fbits, uint32_t *point_list, uint32_t *slot_list,
                        const uint32_t *depth_key, uint32_t last_g, uint32_t *scratch_keys, uint32_t *host_flag, uint32_t flag_value, bool debug,
                        hipStream_t s, int dbits, uint2 *ranges, const uint32_t *err, uint32_t *host_late, uint32_t tag, uint32_t inject) {
    if (cap <= 0) return DAS3R_OK;
    DAS3R_LAUNCH(segment_sort_kernel, dim3(div_up(cap, SEG_CH)), dim3(SEG_T), 0, s, (uint32_t)cap, n_ptr, keys_final, fbits, point_list, slot_list, depth_key, last_g,
                 scratch_keys, host_flag, flag_value, dbits, ranges, err, host_late, tag, inject);
    KERNEL_CHECK(s, debug, "segment_sort");
    return DAS3R_OK;
}

}  // namespace das3r
