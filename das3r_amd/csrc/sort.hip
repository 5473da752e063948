// sort.hip — binning: turns the per-Gaussian tile rectangles into per-tile, depth-ordered splat lists.
// Replaces upstream:cuda_rasterizer/rasterizer_impl.cu's  InclusiveSum -> duplicateWithKeys ->
// cub::DeviceRadixSort::SortPairs(64-bit) -> identifyTileRanges   with an MI355X-first pipeline:
//
//   1. stable LSD radix sort of the P depth keys (32-bit, 4 x 8-bit passes)       -> depth rank -> gaussian id
//   2. exclusive scan of tiles_touched in depth-rank order                         -> instance offsets, num_rendered
//   3. emit (tile id, gaussian id) per touched tile, in depth-rank order
//   4. stable radix PARTITION of the I instances by tile id (ceil(log2(tiles)) bits: 2 passes at 1080p)
//   5. tile ranges from the partitioned tile ids
//
// Emitting in depth order and partitioning stably by tile yields exactly upstream's (tile, depth, index) order
// while moving each instance 2x instead of 6x (45-bit global sort).  All integer work, bit-exact by construction.
//
// Radix pass = hist -> rowscan -> scatter.  The unit of work is a WAVE: each 64-lane wave owns a contiguous
// chunk of 64*ipl keys, keeps its 256 digit counters in LDS and ranks keys with wavefront ballots
// (match-any over the digit bits + popcount prefix) — no block barriers in the ranking loop.
#include "granule.h"
#include "segkey.h"
#include <algorithm>
#include "splat_math.h"

namespace das3r {

// ---------------------------------------------------------------- radix pass
// (hist -> rowscan -> scatter: what distCUDA2's Morton sort uses (knn.hip); the binning's own passes are sort_onesweep.hip's —
//  its "classic" variant on these kernels is an EXPERIMENTS=1 build option)
// IPL = keys per lane (compile time): the whole chunk is fetched into registers before any of it is ranked, so the
// HBM/L2 latency is paid once per wave instead of once per 64 keys.
template <int IPL>
__global__ void __launch_bounds__(256) radix_hist_kernel(const uint32_t *__restrict__ keys, uint32_t cap, const uint32_t *__restrict__ n_ptr,
                                                         int shift, uint32_t mask, int nchunks, uint32_t *__restrict__ hist) {
    const uint32_t n = n_ptr ? min(*n_ptr, cap) : cap;  // element count lives in device memory when the host did not wait for it
    __shared__ uint32_t cnt[SORT_WAVES_PER_BLOCK][RADIX_SIZE];
    const int lane = __lane_id(), wave = threadIdx.x >> 6;
    const int chunk = blockIdx.x * SORT_WAVES_PER_BLOCK + wave;
    const uint32_t base = (uint32_t)chunk * 64u * (uint32_t)IPL;
    uint32_t k[IPL];
#pragma unroll
    for (int s = 0; s < IPL; s++) {
        const uint32_t i = base + s * 64 + lane;
        k[s] = (chunk < nchunks && i < n) ? keys[i] : 0u;
    }
#pragma unroll
    for (int d = lane; d < RADIX_SIZE; d += 64) cnt[wave][d] = 0;
    __builtin_amdgcn_wave_barrier();
    if (chunk >= nchunks) return;
#pragma unroll
    for (int s = 0; s < IPL; s++) {
        const uint32_t i = base + s * 64 + lane;
        if (i < n) atomicAdd(&cnt[wave][(k[s] >> shift) & mask], 1u);
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int d = lane; d < RADIX_SIZE; d += 64) hist[(size_t)d * nchunks + chunk] = cnt[wave][d];
}

// one block per digit: exclusive scan of that digit's per-chunk counts (in place) + digit total
__global__ void __launch_bounds__(256) radix_rowscan_kernel(uint32_t *__restrict__ hist, int nchunks, uint32_t *__restrict__ totals) {
    __shared__ uint32_t ws[4];
    uint32_t *row = hist + (size_t)blockIdx.x * nchunks;
    uint32_t carry = 0;
    for (int base = 0; base < nchunks; base += 256) {
        const int i = base + threadIdx.x;
        const uint32_t v = i < nchunks ? row[i] : 0;
        uint32_t tot;
        const uint32_t ex = block_exclusive_scan_256(v, ws, &tot);
        if (i < nchunks) row[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = carry;
}

// keys_out may be null (last pass of a sort whose keys are not needed); vals_in null => identity (index).
// gather_src / inv_out: see launch_binning (final pass of the tile partition).
template <int IPL>
__global__ void __launch_bounds__(256) radix_scatter_kernel(const uint32_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in,
                                                            uint32_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out,
                                                            uint32_t cap, const uint32_t *__restrict__ n_ptr, int shift, int bits,
                                                            int nchunks, const uint32_t *__restrict__ hist,
                                                            const uint32_t *__restrict__ totals, const uint32_t *__restrict__ gather_src,
                                                            uint32_t *__restrict__ inv_out) {
    const uint32_t n = n_ptr ? min(*n_ptr, cap) : cap;
    __shared__ uint32_t off_s[SORT_WAVES_PER_BLOCK][RADIX_SIZE];
    const int lane = __lane_id(), wave = threadIdx.x >> 6;
    const int chunk = blockIdx.x * SORT_WAVES_PER_BLOCK + wave;
    if (chunk >= nchunks) return;  // no block-level barriers below
    volatile uint32_t *off = off_s[wave];
    const uint32_t mask = (1u << bits) - 1u;
    const uint32_t base = (uint32_t)chunk * 64u * (uint32_t)IPL;

    // fetch the whole chunk (keys, payloads and, on the final partition pass, the gathered splat ids) up front
    uint32_t k[IPL], v[IPL], out[IPL];
#pragma unroll
    for (int s = 0; s < IPL; s++) {
        const uint32_t i = base + s * 64 + lane;
        k[s] = i < n ? keys_in[i] : 0u;
        v[s] = (vals_in && i < n) ? vals_in[i] : i;
    }
#pragma unroll
    for (int s = 0; s < IPL; s++) {
        const uint32_t i = base + s * 64 + lane;
        out[s] = (gather_src && i < n) ? gather_src[v[s]] : v[s];
    }
    // digit base = exclusive scan of the 256 digit totals (4 consecutive digits per lane)
    {
        uint32_t t[4], h[4], sum = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            t[q] = totals[4 * lane + q];
            h[q] = hist[(size_t)(4 * lane + q) * nchunks + chunk];
            sum += t[q];
        }
        uint32_t ex = wave_incl_scan_u32(sum) - sum;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            off[4 * lane + q] = ex + h[q];
            ex += t[q];
        }
    }
    __builtin_amdgcn_wave_barrier();

#pragma unroll
    for (int s = 0; s < IPL; s++) {
        const uint32_t i = base + s * 64 + lane;
        const bool valid = i < n;
        const uint32_t digit = (k[s] >> shift) & mask;
        // match-any: lanes holding the same digit
        uint64_t peers = __ballot(valid);
        for (int b = 0; b < bits; b++) {
            const bool bit = (digit >> b) & 1u;
            const uint64_t m = __ballot(valid && bit);
            peers &= bit ? m : ~m;
        }
        const uint64_t lt = (1ull << lane) - 1ull;
        const uint32_t rank = (uint32_t)__popcll(peers & lt);
        const uint32_t count = (uint32_t)__popcll(peers);
        uint32_t o = 0;
        if (valid) o = off[digit];
        __builtin_amdgcn_wave_barrier();
        if (valid && rank == 0) off[digit] = o + count;
        __builtin_amdgcn_wave_barrier();
        if (valid) {
            const uint32_t dst = o + rank;
            if (keys_out) keys_out[dst] = k[s];
            // final pass of the tile partition: the payload is the emission slot e; store the splat id it stands for
            // and keep e itself in list order (slot list: where the backward pass puts this instance's partial sums)
            vals_out[dst] = out[s];
            if (inv_out) inv_out[dst] = v[s];   // slot list
        }
    }
}

static int radix_pass(const uint32_t *kin, const uint32_t *vin, uint32_t *kout, uint32_t *vout, int64_t n, int shift, int bits,
                      uint32_t *hist, uint32_t *totals, bool debug, hipStream_t s, const uint32_t *gather_src = nullptr,
                      uint32_t *inv_out = nullptr, const uint32_t *n_ptr = nullptr) {
    const int ipl = sort_items_per_lane(n), nchunks = sort_num_chunks(n);
    const int nblocks = div_up(nchunks, SORT_WAVES_PER_BLOCK);
    const uint32_t mask = (1u << bits) - 1u;
#define HIST(IPL) DAS3R_LAUNCH((radix_hist_kernel<IPL>), dim3(nblocks), dim3(256), 0, s, kin, (uint32_t)n, n_ptr, shift, mask, nchunks, hist)
#define SCAT(IPL)                                                                                                             \
    DAS3R_LAUNCH((radix_scatter_kernel<IPL>), dim3(nblocks), dim3(256), 0, s, kin, vin, kout, vout, (uint32_t)n, n_ptr, shift, bits, \
                 nchunks, hist, totals, gather_src, inv_out)
    if (ipl == 4) HIST(4); else if (ipl == 8) HIST(8); else HIST(16);
    KERNEL_CHECK(s, debug, "radix_hist");
    DAS3R_LAUNCH(radix_rowscan_kernel, dim3(RADIX_SIZE), dim3(256), 0, s, hist, nchunks, totals);
    KERNEL_CHECK(s, debug, "radix_rowscan");
    if (ipl == 4) SCAT(4); else if (ipl == 8) SCAT(8); else SCAT(16);
    KERNEL_CHECK(s, debug, "radix_scatter");
#undef HIST
#undef SCAT
    return DAS3R_OK;
}

// Full stable sort of (key, index) pairs on key bits [0,total_bits): 8 bits per pass, ping-pong A <-> B, identity
// values on the first pass, keys not written on the last.  keys_in must be keyA.  *vals_final = sorted indices.
int radix_sort_u32_pairs(const uint32_t *keys_in, uint32_t *keyA, uint32_t *keyB, uint32_t *valA, uint32_t *valB, int64_t n,
                         int total_bits, uint32_t *hist, uint32_t *totals, uint32_t **vals_final, hipStream_t s) {
    (void)keys_in;
    const int passes = (total_bits + 7) / 8;
    uint32_t *kin = keyA, *kout = keyB, *vin = nullptr, *vout = valB;
    int shift = 0;
    for (int p = 0; p < passes; p++) {
        const int bits = (total_bits - shift) < 8 ? (total_bits - shift) : 8;
        const bool last = p == passes - 1;
        int rc = radix_pass(kin, vin, last ? nullptr : kout, vout, n, shift, bits, hist, totals, false, s);
        if (rc) return rc;
        shift += bits;
        uint32_t *t = kin; kin = kout; kout = t;
        vin = vout;
        vout = (vout == valB) ? valA : valB;
    }
    *vals_final = vin;
    return DAS3R_OK;
}

#ifdef DAS3R_EXPERIMENTS   // the "classic" binning (DAS3R_SORT=classic): separate emission kernel + three-kernel partition passes
// ---------------------------------------------------------------- instance emission (depth-rank order)
__global__ void __launch_bounds__(256) emit_kernel(int P, int tiles_x, int tiles_y, const uint32_t *__restrict__ sorted_idx,
                                                   const uint32_t *__restrict__ tiles_touched, const uint32_t *__restrict__ offsets,
                                                   const float4 *__restrict__ xyh, const int32_t *__restrict__ radii,
                                                   uint32_t *__restrict__ tile_keys, uint32_t *__restrict__ gids, uint32_t cap,
                                                   uint32_t *__restrict__ ghist /*[passes][256] or null*/, int tbits, int tight_rect) {
    // grid-stride over depth ranks; the digit histograms of the tile ids (needed by the single-pass partition) are
    // accumulated in LDS and flushed with one global atomic per non-empty bin and workgroup
    __shared__ uint32_t h[4][RADIX_SIZE];
    if (ghist) {
#pragma unroll
        for (int q = 0; q < 4; q++) h[q][threadIdx.x] = 0;
        __syncthreads();
    }
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < P; r += gridDim.x * blockDim.x) {
        const uint32_t g = sorted_idx[r];
        if (tiles_touched[g] == 0) continue;
        uint32_t o = offsets[r];
        const float4 p = xyh[(size_t)g * SPLAT_REC];
        int rminx, rminy, rmaxx, rmaxy;
        binned_rect(p, radii[g], tiles_x, tiles_y, tight_rect != 0, rminx, rminy, rmaxx, rmaxy);
        for (int y = rminy; y < rmaxy; y++)
            for (int x = rminx; x < rmaxx; x++) {
                if (o < cap) {  // cap < num_rendered only when a capacity hint was too small (the forward is then re-run)
                    const uint32_t t = (uint32_t)(y * tiles_x + x);
                    tile_keys[o] = t;
                    gids[o] = g;  // gid_of[emission slot]
                    if (ghist) {
                        for (int q = 0, sh = 0, dw = tile_digit_width(tbits); sh < tbits; q++, sh += dw) {
                            const int bits = (tbits - sh) < dw ? (tbits - sh) : dw;
                            atomicAdd(&h[q][(t >> sh) & ((1u << bits) - 1u)], 1u);
                        }
                    }
                }
                o++;
            }
    }
    if (ghist) {
        __syncthreads();
        for (int q = 0, sh = 0, dw = tile_digit_width(tbits); sh < tbits; q++, sh += dw) {
            const uint32_t c = h[q][threadIdx.x];
            if (c) atomicAdd(&ghist[q * RADIX_SIZE + threadIdx.x], c);
        }
    }
}

#endif   // DAS3R_EXPERIMENTS

// Four consecutive list positions per thread (round 6: one per thread, each with a load of its own and one of its neighbour's key, was
// 16 700 workgroups at the DAS3R shape: 10.5 us; the keys are 16-byte aligned, whichever ping-pong buffer they end in).
__global__ void __launch_bounds__(256) tile_ranges_kernel(uint32_t cap, const uint32_t *__restrict__ n_ptr,
                                                          const uint32_t *__restrict__ tile_keys, uint2 *__restrict__ ranges,
                                                          uint32_t *__restrict__ err, uint32_t *__restrict__ host_late, uint32_t tag,
                                                          uint32_t rearm_words /*fused emission: err[0 .. rearm_words) back to zero*/,
                                                          uint32_t inject /*das3r_debug_inject_fault: bits forced into the word (tests)*/,
                                                          int dbits /*segmented path: the keys are tile id << dbits | depth bucket*/) {
    const uint32_t I = n_ptr ? min(*n_ptr, cap) : cap;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && host_late) {   // last binning kernel: hand the self-check word of this forward to the host mailbox
        host_late[0] = *err | inject;
        __hip_atomic_store(host_late + 1, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (i < rearm_words) err[i] = 0u;   // (thread 0 has read err[0] just above) the ring slot is zero at rest again
    const uint32_t i0 = 4u * i;
    if (i0 >= I) return;
    uint32_t k[4];
    if (i0 + 4u <= I) {
        const uint4 v = *reinterpret_cast<const uint4 *>(tile_keys + i0);
        k[0] = v.x; k[1] = v.y; k[2] = v.z; k[3] = v.w;
    } else {
#pragma unroll
        for (int q = 0; q < 4; q++) k[q] = i0 + q < I ? tile_keys[i0 + q] : 0u;
    }
    uint32_t prev = i0 ? tile_keys[i0 - 1u] >> dbits : 0u;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint32_t at = i0 + (uint32_t)q;
        if (at < I) {
            const uint32_t t = k[q] >> dbits;
            if (at == 0u) ranges[t].x = 0u;
            else if (prev != t) {
                ranges[prev].y = at;
                ranges[t].x = at;
            }
            if (at == I - 1u) ranges[t].y = I;
            prev = t;
        }
    }
}

// ---------------------------------------------------------------- host side
int launch_depth_sort(int P, char *geom, const Layout &L, int part, char *binning_ctrl, size_t binning_ctrl_bytes, bool debug, hipStream_t s) {
    uint32_t *keyA = (uint32_t *)(geom + L.g_keyA), *keyB = (uint32_t *)(geom + L.g_keyB);
    uint32_t *valA = (uint32_t *)(geom + L.g_valA), *valB = (uint32_t *)(geom + L.g_valB);
    uint32_t *hist = (uint32_t *)(geom + L.g_hist), *totals = (uint32_t *)(geom + L.g_totals);
    if (P == 0) return DAS3R_OK;
    // 4 passes: A -> B -> A -> B -> A ; final ranks land in valA (== pub.sorted_idx)
    int rc;
    if (use_onesweep()) return launch_onesweep_depth_sort(P, geom, L, part, (uint32_t *)binning_ctrl, (uint32_t)(binning_ctrl_bytes / 4), debug, s);
#ifdef DAS3R_EXPERIMENTS
    if (part == 1) return DAS3R_OK;   // classic path: everything ran in part 0 (its binning does not use control words)
    if ((rc = radix_pass(keyA, nullptr, keyB, valB, P, 0, 8, hist, totals, debug, s))) return rc;
    if ((rc = radix_pass(keyB, valB, keyA, valA, P, 8, 8, hist, totals, debug, s))) return rc;
    if ((rc = radix_pass(keyA, valA, keyB, valB, P, 16, 8, hist, totals, debug, s))) return rc;
    if ((rc = radix_pass(keyB, valB, nullptr, valA, P, 24, 8, hist, totals, debug, s))) return rc;
#else
    (void)keyA; (void)keyB; (void)valA; (void)valB; (void)hist; (void)totals; (void)rc;
#endif
    return DAS3R_OK;
}

// Hinted path, first half of the binning: zero the control words, then scan + emit in one kernel.  The caller copies the
// count back right behind it and then calls launch_binning(..., fused_scan = true) for the partition.
int launch_binning_scan_emit(int P, int64_t I, const int32_t *radii, char *geom, char *binning, const Layout &L, bool ctrl_zeroed,
                             uint32_t *host_out, uint32_t tag, bool debug, hipStream_t s, bool index_order) {
    if (I == 0 || P == 0) return DAS3R_OK;
    if (!ctrl_zeroed) HIP_TRY(hipMemsetAsync(binning + L.b_ghist, 0, L.b_ctrl_bytes, s));
    return launch_scan_emit(P, I, radii, geom, binning, L, host_out, tag, debug, s, index_order);
}

// I = capacity of the binning buffer; the true instance count is read by the kernels from geom + L.g_count
int launch_binning(int P, int64_t I, int W, int H, const int32_t *radii, char *geom, char *binning, char *img, const Layout &L,
                   bool fused_scan, uint32_t *host_late, uint32_t tag, bool debug, hipStream_t s, uint32_t **dead_keys, uint32_t *emit_slot,
                   uint32_t *seg_host_flag, uint32_t seg_flag_value) {
    const uint32_t *n_ptr = (const uint32_t *)(geom + L.g_count);
    (void)W; (void)H;
    uint2 *ranges = (uint2 *)(img + L.pub.ranges);  // zeroed by preprocess_kernel
    if (I == 0 || P == 0) return DAS3R_OK;
    uint32_t *keyA = (uint32_t *)(binning + L.b_keyA), *keyB = (uint32_t *)(binning + L.b_keyB);
    uint32_t *valA = (uint32_t *)(binning + L.b_valA), *valB = (uint32_t *)(binning + L.b_valB);
    uint32_t *hist = (uint32_t *)(binning + L.b_hist), *totals = (uint32_t *)(binning + L.b_totals);
    uint32_t *gid_of = (uint32_t *)(binning + L.b_gid_of), *inv = (uint32_t *)(binning + L.b_slot);
    const bool onesweep = use_onesweep();
    (void)onesweep;   // the binning control words were zeroed by the last depth pass (launch_depth_sort part 1)
#ifdef DAS3R_EXPERIMENTS
    if (!fused_scan) {   // (hinted path: launch_binning_scan_emit has already emitted the instances)
    const int emit_blocks = div_up(P, 256) < 1024 ? div_up(P, 256) : 1024;
    DAS3R_LAUNCH(emit_kernel, dim3(emit_blocks), dim3(256), 0, s, P, L.tiles_x, L.tiles_y,
                 (const uint32_t *)(geom + L.pub.sorted_idx), (const uint32_t *)(geom + L.pub.tiles_touched),
                 (const uint32_t *)(geom + L.pub.offsets), (const float4 *)(geom + L.pub.xy), radii, keyA, gid_of, (uint32_t)I,
                 onesweep ? (uint32_t *)(binning + L.b_ghist) : (uint32_t *)nullptr, L.tbits, use_tight_rect() ? 1 : 0);
    KERNEL_CHECK(s, debug, "emit");
    }
#else
    (void)radii; (void)fused_scan; (void)gid_of;
#endif
    if (onesweep) {
        uint32_t *kfinal = nullptr;
        int rc1 = launch_onesweep_partition(I, geom, binning, L, &kfinal, debug, s, emit_slot ? emit_slot + 64 : nullptr, emit_slot);
        if (rc1) return rc1;
        if (dead_keys) *dead_keys = kfinal;   // the tile keys are dead once tile_ranges_kernel has run
        if (L.dbits > 0) {   // segmented path: exact depth order inside every (tile, bucket) segment, in place (segsort.hip)
            uint32_t *other = kfinal == keyA ? keyB : keyA;   // the previous pass's keys: dead, scratch for a segment too long for LDS
            // (round 6: it also writes the tile ranges and hands over the self-check word — no tile_ranges launch behind it; the segmented path
            //  never runs with the fused emission, whose ring slot that kernel re-arms)
            return launch_segment_sort(I, n_ptr, kfinal, L.kshift, (uint32_t *)(binning + L.pub.point_list), (uint32_t *)(binning + L.b_slot),
                                       (const uint32_t *)(geom + L.pub.depth_key), (uint32_t)(P - 1), other, seg_host_flag, seg_flag_value, debug, s,
                                       L.dbits, ranges, (const uint32_t *)(geom + L.g_ticket) + 8, host_late, tag, (uint32_t)switches().inject_fault);
        }
        const int rearm = emit_slot ? EMIT_SLOT_WORDS : 0;
        DAS3R_LAUNCH(tile_ranges_kernel, dim3(div_up(std::max<int64_t>(div_up(I, 4), rearm), 256)), dim3(256), 0, s, (uint32_t)I, n_ptr, kfinal, ranges,
                     emit_slot ? emit_slot : (uint32_t *)(geom + L.g_ticket) + 8, host_late, tag, (uint32_t)rearm, (uint32_t)switches().inject_fault, L.dbits + L.kshift);
        KERNEL_CHECK(s, debug, "tile_ranges");
        return DAS3R_OK;
    }
#ifdef DAS3R_EXPERIMENTS
    // stable partition by tile id: tile_passes passes of <= 8 bits; ping-pong A -> B (-> A)
    // payload = emission slot e (identity on the first pass); the last pass turns it into the splat id and records inv[e]
    uint32_t *kin = keyA, *vin = nullptr, *kout = keyB, *vout = valB;
    int shift = 0, rc;
    for (int p = 0; p < L.tile_passes; p++) {
        const int dw = tile_digit_width(L.tbits), bits = (L.tbits - shift) < dw ? (L.tbits - shift) : dw;
        const bool last = p == L.tile_passes - 1;
        if ((rc = radix_pass(kin, vin, kout, vout, I, shift, bits, hist, totals, debug, s, last ? gid_of : nullptr, last ? inv : nullptr, n_ptr)))
            return rc;
        shift += bits;
        uint32_t *t = kin; kin = kout; kout = t;
        vin = vout;
        vout = (vout == valB) ? valA : valB;
    }
    // kin/vin now hold the partitioned (tile id, gaussian id) lists; vin == binning + pub.point_list by construction
    if (dead_keys) *dead_keys = kin;
    DAS3R_LAUNCH(tile_ranges_kernel, dim3(div_up(div_up(I, 4), 256)), dim3(256), 0, s, (uint32_t)I, n_ptr, kin, ranges,
                 (uint32_t *)(geom + L.g_ticket) + 8, host_late, tag, 0u, (uint32_t)switches().inject_fault, 0);
    KERNEL_CHECK(s, debug, "tile_ranges");
#else
    (void)keyA; (void)keyB; (void)valA; (void)valB; (void)hist; (void)totals; (void)inv;
#endif
    return DAS3R_OK;
}

}  // namespace das3r
