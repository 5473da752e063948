// sort_onesweep.hip — single-pass-per-digit stable radix passes (chained scan with decoupled look-back) used by the
// rasterizer's binning: ONE kernel per 8-bit digit instead of histogram + row scan + scatter, and no second read of the
// keys.  At 100 k splats the binning was launch/latency bound (22 launches of ~4.5 us each); this path needs 11.
//
// Each 256-lane workgroup takes a ticket (arrival order => every predecessor is already running: no dependence on
// dispatch order), ranks its 256*IPL keys exactly like the classic scatter kernel (wave-private LDS counters, ballot
// match-any), publishes its per-digit counts and finds the number of equal-digit keys in all earlier workgroups by
// walking back over their published words.
//
// Inter-workgroup hand-off follows the MI355X rule for XCD-private L2s (cdna_hip_programming.md §6 G16, recipe R2): the
// payload IS the flag — one naturally aligned 8-byte {tag, value} word per (workgroup, digit), written with ONE relaxed
// agent-scope atomic store (sc1, write-through) and polled with relaxed agent-scope atomic loads; no fences, no plain
// accesses to shared words.  Status words, tickets and global histograms are zeroed by an earlier kernel of the same
// forward (preprocess / a memset), never by a previous call.  Spins are bounded: on timeout an error word is raised and
// the forward reports DAS3R_ERR_HIP instead of hanging the GPU.
#include "common.h"

namespace das3r {

typedef unsigned long long u64;
constexpr u64 TAG_AGG = 1ull << 62, TAG_PREFIX = 2ull << 62, TAG_MASK = 3ull << 62;
constexpr unsigned SPIN_LIMIT = 1u << 22;

__device__ __forceinline__ void granule_store(u64 *p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u64 granule_load(const u64 *p) {
    return __hip_atomic_load(const_cast<u64 *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) {
    const int lane = __lane_id();
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t n = __shfl_up(v, o, 64);
        if (lane >= o) v += n;
    }
    return v;
}

// global histograms of all four 8-bit digits of the depth keys (read once, LDS-privatised, few global atomics)
__global__ void __launch_bounds__(256) depth_hist_kernel(const uint32_t *__restrict__ keys, uint32_t n, uint32_t *__restrict__ ghist) {
    __shared__ uint32_t h[4][RADIX_SIZE];
#pragma unroll
    for (int q = 0; q < 4; q++) h[q][threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const uint32_t k = keys[i];
        atomicAdd(&h[0][k & 255u], 1u);
        atomicAdd(&h[1][(k >> 8) & 255u], 1u);
        atomicAdd(&h[2][(k >> 16) & 255u], 1u);
        atomicAdd(&h[3][k >> 24], 1u);
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint32_t c = h[q][threadIdx.x];
        if (c) atomicAdd(&ghist[q * RADIX_SIZE + threadIdx.x], c);
    }
}

// One digit of a stable LSD sort / partition.  keys_out may be null; vals_in null => payload = index.
// gather_src / inv_out: final pass of the tile partition (payload = emission slot e): store gather_src[e] and inv[e] = dst.
template <int IPL>
__global__ void __launch_bounds__(256) onesweep_pass_kernel(const uint32_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in,
                                                            uint32_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out,
                                                            uint32_t cap, const uint32_t *__restrict__ n_ptr, int shift, int bits,
                                                            const uint32_t *__restrict__ ghist /*[256] this digit*/,
                                                            u64 *__restrict__ status /*[nblocks][256]*/, uint32_t *__restrict__ ticket,
                                                            const uint32_t *__restrict__ gather_src, uint32_t *__restrict__ inv_out,
                                                            uint32_t *__restrict__ err) {
    __shared__ uint32_t cnt[4][RADIX_SIZE];  // per-wave digit counts, then per-wave running offsets
    __shared__ uint32_t ws[4];
    __shared__ uint32_t s_block;
    const int tid = threadIdx.x, lane = __lane_id(), wave = tid >> 6;
    const uint32_t n = n_ptr ? min(*n_ptr, cap) : cap;
    if (tid == 0) s_block = atomicAdd(ticket, 1u);
#pragma unroll
    for (int w = 0; w < 4; w++) cnt[w][tid] = 0;
    __syncthreads();
    const uint32_t b = s_block;
    const uint32_t base = (b * 4u + (uint32_t)wave) * 64u * (uint32_t)IPL;
    const uint32_t mask = (1u << bits) - 1u;

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
        if (i < n) atomicAdd(&cnt[wave][(k[s] >> shift) & mask], 1u);
    }
    __syncthreads();

    // thread d owns digit d: publish this workgroup's count, look back over the earlier workgroups
    {
        const uint32_t c0 = cnt[0][tid], c1 = cnt[1][tid], c2 = cnt[2][tid], c3 = cnt[3][tid];
        const uint32_t total = c0 + c1 + c2 + c3;
        u64 *mine = status + (size_t)b * RADIX_SIZE + tid;
        uint32_t excl = 0;
        if (b == 0) {
            granule_store(mine, TAG_PREFIX | total);
        } else {
            granule_store(mine, TAG_AGG | total);
            // windowed look-back: LB predecessors are fetched with independent loads per step (one L2 round trip per
            // window instead of one per predecessor), then consumed in order
            constexpr int LB = 8;
            unsigned spins = 0;
            int p = (int)b - 1;
            bool found = false;
            while (p >= 0 && !found) {
                u64 x[LB];
#pragma unroll
                for (int j = 0; j < LB; j++) x[j] = (p - j >= 0) ? granule_load(status + (size_t)(p - j) * RADIX_SIZE + tid) : TAG_PREFIX;
                int used = 0;
#pragma unroll
                for (int j = 0; j < LB; j++) {
                    if (found || used < j) continue;  // stop consuming behind an unpublished word / after a prefix
                    const u64 tag = x[j] & TAG_MASK;
                    if (tag == 0) continue;           // not published yet: retry from here
                    excl += (uint32_t)(x[j] & 0xFFFFFFFFull);
                    used = j + 1;
                    found = tag == TAG_PREFIX;
                }
                p -= used;
                if (used == 0) {
                    if (++spins > SPIN_LIMIT) {  // predecessor never published: give up loudly instead of hanging
                        atomicOr(err, 1u);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            granule_store(mine, TAG_PREFIX | (u64)(excl + total));
        }
        // first output slot of digit d = (keys with smaller digits) + (same digit in earlier workgroups)
        const uint32_t g = ghist[tid];
        const uint32_t incl = wave_incl_scan_u32(g);
        if (lane == 63) ws[wave] = incl;
        __syncthreads();
        uint32_t wbase = 0;
#pragma unroll
        for (int w = 0; w < 4; w++)
            if (w < wave) wbase += ws[w];
        const uint32_t digit_base = wbase + incl - g;
        const uint32_t start = digit_base + excl;
        cnt[0][tid] = start;
        cnt[1][tid] = start + c0;
        cnt[2][tid] = start + c0 + c1;
        cnt[3][tid] = start + c0 + c1 + c2;
    }
    __syncthreads();

    volatile uint32_t *off = cnt[wave];
#pragma unroll
    for (int s = 0; s < IPL; s++) {
        const uint32_t i = base + s * 64 + lane;
        const bool valid = i < n;
        const uint32_t digit = (k[s] >> shift) & mask;
        uint64_t peers = __ballot(valid);
        for (int bb = 0; bb < bits; bb++) {
            const bool bit = (digit >> bb) & 1u;
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
            vals_out[dst] = out[s];
            if (inv_out) inv_out[v[s]] = dst;
        }
    }
}

static int onesweep_pass(const uint32_t *kin, const uint32_t *vin, uint32_t *kout, uint32_t *vout, int64_t cap, const uint32_t *n_ptr,
                         int shift, int bits, const uint32_t *ghist, u64 *status, uint32_t *ticket, const uint32_t *gather_src,
                         uint32_t *inv_out, uint32_t *err, bool debug, hipStream_t s) {
    const int ipl = sort_items_per_lane(cap);
    const int nblocks = div_up(cap, (int64_t)256 * ipl);
#define PASS(IPL)                                                                                                              \
    DAS3R_LAUNCH((onesweep_pass_kernel<IPL>), dim3(nblocks), dim3(256), 0, s, kin, vin, kout, vout, (uint32_t)cap, n_ptr, shift, \
                 bits, ghist, status, ticket, gather_src, inv_out, err)
    if (ipl == 4) PASS(4); else if (ipl == 8) PASS(8); else PASS(16);
#undef PASS
    KERNEL_CHECK(s, debug, "onesweep_pass");
    return DAS3R_OK;
}

size_t onesweep_status_bytes(int64_t n, int passes) {
    if (n <= 0) return 256;
    const int nblocks = div_up(n, (int64_t)256 * sort_items_per_lane(n));
    return (size_t)passes * (size_t)nblocks * RADIX_SIZE * sizeof(u64);
}

// Depth sort of the P splats: ctrl = [ghist 4x256 u32][tickets 4 u32 (+pad)][status 4 passes], zeroed by preprocess_kernel.
int launch_onesweep_depth_sort(int P, char *geom, const Layout &L, bool debug, hipStream_t s) {
    uint32_t *keyA = (uint32_t *)(geom + L.g_keyA), *keyB = (uint32_t *)(geom + L.g_keyB);
    uint32_t *valA = (uint32_t *)(geom + L.g_valA), *valB = (uint32_t *)(geom + L.g_valB);
    uint32_t *ghist = (uint32_t *)(geom + L.g_ghist), *ticket = (uint32_t *)(geom + L.g_ticket);
    uint32_t *err = (uint32_t *)(geom + L.g_ticket) + 8;  // inside the zeroed control region
    u64 *status = (u64 *)(geom + L.g_status);
    const size_t per_pass = onesweep_status_bytes(P, 1) / sizeof(u64);
    const int hist_blocks = div_up(P, 256) < 512 ? div_up(P, 256) : 512;
    DAS3R_LAUNCH(depth_hist_kernel, dim3(hist_blocks), dim3(256), 0, s, keyA, (uint32_t)P, ghist);
    KERNEL_CHECK(s, debug, "depth_hist");
    int rc;
    // A -> B -> A -> B -> A ; final ranks land in valA (== pub.sorted_idx)
    if ((rc = onesweep_pass(keyA, nullptr, keyB, valB, P, nullptr, 0, 8, ghist + 0, status + 0 * per_pass, ticket + 0, nullptr, nullptr, err, debug, s))) return rc;
    if ((rc = onesweep_pass(keyB, valB, keyA, valA, P, nullptr, 8, 8, ghist + 256, status + 1 * per_pass, ticket + 1, nullptr, nullptr, err, debug, s))) return rc;
    if ((rc = onesweep_pass(keyA, valA, keyB, valB, P, nullptr, 16, 8, ghist + 512, status + 2 * per_pass, ticket + 2, nullptr, nullptr, err, debug, s))) return rc;
    if ((rc = onesweep_pass(keyB, valB, nullptr, valA, P, nullptr, 24, 8, ghist + 768, status + 3 * per_pass, ticket + 3, nullptr, nullptr, err, debug, s))) return rc;
    return DAS3R_OK;
}

// Stable partition of the emitted instances by tile id.  ctrl (binning buffer) = [ghist 2x256][tickets][status], zeroed by
// a memset before emit; emit_kernel accumulated the two digit histograms.
int launch_onesweep_partition(int64_t cap, char *geom, char *binning, const Layout &L, uint32_t **keys_final, bool debug, hipStream_t s) {
    const uint32_t *n_ptr = (const uint32_t *)(geom + L.g_count);
    uint32_t *err = (uint32_t *)(geom + L.g_ticket) + 8;  // inside the zeroed control region
    uint32_t *keyA = (uint32_t *)(binning + L.b_keyA), *keyB = (uint32_t *)(binning + L.b_keyB);
    uint32_t *valA = (uint32_t *)(binning + L.b_valA), *valB = (uint32_t *)(binning + L.b_valB);
    uint32_t *gid_of = (uint32_t *)(binning + L.b_gid_of), *inv = (uint32_t *)(binning + L.b_inv);
    uint32_t *ghist = (uint32_t *)(binning + L.b_ghist), *ticket = (uint32_t *)(binning + L.b_ticket);
    u64 *status = (u64 *)(binning + L.b_status);
    const size_t per_pass = onesweep_status_bytes(cap, 1) / sizeof(u64);
    uint32_t *kin = keyA, *vin = nullptr, *kout = keyB, *vout = valB;
    int shift = 0, rc;
    for (int p = 0; p < L.tile_passes; p++) {
        const int bits = (L.tbits - shift) < 8 ? (L.tbits - shift) : 8;
        const bool last = p == L.tile_passes - 1;
        if ((rc = onesweep_pass(kin, vin, kout, vout, cap, n_ptr, shift, bits, ghist + p * 256, status + p * per_pass, ticket + p,
                                last ? gid_of : nullptr, last ? inv : nullptr, err, debug, s)))
            return rc;
        shift += bits;
        uint32_t *t = kin; kin = kout; kout = t;
        vin = vout;
        vout = (vout == valB) ? valA : valB;
    }
    *keys_final = kin;
    return DAS3R_OK;
}

}  // namespace das3r
