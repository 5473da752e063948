// render_quad.h — four lanes per pixel: the quad helpers and the exact walk of one block's list, shared by render_lanes.hip (a wave per
// 4x4 block for the whole tile) and render_slices.hip (round 6: a block's list of a batch cut into chunks that any wave takes).
// See render_lanes.hip for the decomposition.
#pragma once
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

}  // namespace das3r
