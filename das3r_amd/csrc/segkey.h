// segkey.h — depth buckets of the segmented binning path (round 4; segsort.hip, scan_emit.hip, preprocess.hip).
//
// Long tile lists (the DAS3R shape: 416 tiles x ~14 k entries) used to take a global depth sort of the P splats (4 radix passes +
// a histogram) in front of the tile partition.  The segmented path sorts nothing globally: the partition key of an instance
// becomes (tile id << dbits | depth bucket), the two partition passes the tile ids need anyway carry the bucket bits for free
// (9 + 7 = 16 bits), and every (tile, bucket) SEGMENT — a few dozen to a few hundred entries, in index order because the passes
// are stable — is put into exact (depth bits, index) order inside LDS by segment_sort_kernel.  The lists come out bit-identical to
// the global sort's (upstream's 64-bit key order: SURVEY.md A.6).
//
// The bucket of a splat is a MONOTONE function of its depth bits (a smaller depth never lands in a later bucket — the only
// property correctness needs), chosen so that the buckets are about equally full whatever the scene's depth distribution is
// (two walls and a sky are three spikes on any fixed scale): a 256-bin histogram of the depth bits' top 13 bits (sign, exponent,
// four mantissa bits: sixteen bins per octave over [2^-8, 2^8), clamped outside), weighted by tiles_touched, is accumulated by the
// preprocess kernel; its running sum, interpolated linearly inside a bin with the next 19 mantissa bits, is the splat's position
// in [0, total), scaled to [0, 2^(dbits + 16)): bucket and sixteen bits of fraction.  Counts are shifted down to stay below 2^24, so every float below is an exact integer
// until the fma, and fma / multiply / truncation are monotone: see tests/test_segkey_model.py for the numpy restatement.
#pragma once
#include <stdint.h>

namespace das3r {

constexpr int DBINS = 256;
constexpr int DBIN_SHIFT = 19;              // depth bits >> 19: sign + exponent + 4 mantissa bits
constexpr int DBIN0 = (127 - 8) << 4;       // bin 0 starts at depth 2^-8

#ifdef __HIPCC__
__device__ __forceinline__ uint32_t depth_bin(const uint32_t bits) {
    const int raw = (int)(bits >> DBIN_SHIFT) - DBIN0;
    return (uint32_t)(raw < 0 ? 0 : (raw > DBINS - 1 ? DBINS - 1 : raw));
}
// cnt / cdf: the workgroup's LDS copies of the (shifted) bin counts and their exclusive running sums, as floats (exact integers
// below 2^24); scale = nb / total (0 when nothing is visible).  nb = 2^(dbits + fbits), fbits = min(16, 32 - key bits): the result is the
// bucket AND the next fbits bits of the same monotone map below it — the partition passes sort on the bucket bits only, the fraction
// rides along in the key's low bits for free and orders a segment without a trip to the depth keys wherever it has no ties
// (segsort.hip; the first version gathered the depth of every instance: 4 bytes out of a 64-byte line each, 0.20 - 0.28 ms of a
// 2.1 ms step on the 5 M-splat DAS3R shape).
__device__ __forceinline__ uint32_t depth_bucket(const uint32_t bits, const float *cnt, const float *cdf, const float scale, const uint32_t nb) {
    const int raw = (int)(bits >> DBIN_SHIFT) - DBIN0;
    const uint32_t bin = (uint32_t)(raw < 0 ? 0 : (raw > DBINS - 1 ? DBINS - 1 : raw));
    const float frac = raw < 0 ? 0.0f : (raw > DBINS - 1 ? 1.0f : (float)(bits & ((1u << DBIN_SHIFT) - 1u)) * (1.0f / (float)(1u << DBIN_SHIFT)));
    const float pos = __fmaf_rn(cnt[bin], frac, cdf[bin]);
    const uint32_t b = (uint32_t)__fmul_rn(pos, scale);
    return b < nb ? b : nb - 1u;
}
#endif

}  // namespace das3r
