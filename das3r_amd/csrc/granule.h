// granule.h — inter-workgroup hand-off words and small scan helpers shared by the binning kernels.
// Hand-off = one naturally aligned 8-byte {tag, value} word written by ONE relaxed agent-scope (sc1, write-through) store
// and polled with relaxed agent-scope loads: the payload is the flag (cdna_hip_programming.md §6 G16, recipe R2).
#pragma once
#include "common.h"

namespace das3r {

typedef unsigned long long u64;
constexpr u64 TAG_AGG = 1ull << 62, TAG_MASK = 3ull << 62;
constexpr unsigned SPIN_LIMIT = 1u << 22;
constexpr uint32_t ERR_TIMEOUT = 1u, ERR_RANGE = 2u, ERR_COUNTS = 8u;   // bits of the error word

__device__ __forceinline__ void granule_store(u64 *p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u64 granule_load(const u64 *p) {
    return __hip_atomic_load(const_cast<u64 *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Poll number `spins` of a hand-off word.  Once in ~1e5 forwards a poller kept reading the unpublished value of a word that had
// long been published, until the 4 s spin limit (soak runs, a rocprofv3 run): a cached copy that agent-scope loads kept
// hitting is the only explanation left.  After SOFT_SPINS fruitless polls (~0.1 ms) the word is therefore read with an atomic
// read-modify-write (OR 0), which is performed where the stores land; ERR_HARD_POLL in the error word records that it was needed
// (informational: the host does not treat it as a failure).
constexpr unsigned SOFT_SPINS = 96;
constexpr uint32_t ERR_HARD_POLL = 16u;
__device__ __forceinline__ u64 granule_poll(const u64 *p, const unsigned spins) {
    if (spins < SOFT_SPINS) return granule_load(p);
    return __hip_atomic_fetch_or(const_cast<u64 *>(p), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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

// exclusive scan across a 256-thread block; returns the exclusive prefix of `v`, *total = block sum
__device__ __forceinline__ uint32_t block_exclusive_scan_256(uint32_t v, uint32_t *lds_wave_sums /*[4]*/, uint32_t *total) {
    const int lane = __lane_id(), wave = threadIdx.x >> 6;
    const uint32_t incl = wave_incl_scan_u32(v);
    __syncthreads();  // protect lds_wave_sums reuse across calls
    if (lane == 63) lds_wave_sums[wave] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const uint32_t s = lds_wave_sums[w];
        if (w < wave) base += s;
        tot += s;
    }
    *total = tot;
    return base + incl - v;
}

}  // namespace das3r
