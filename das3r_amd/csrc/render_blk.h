// render_blk.h — the pair loop of render_bwd_blk.hip (one image row of a 4x4 block, four pixel steps interleaved) and its DPP
// helpers, in a header so that tools/probes/blk_loop_probe.hip can time exactly this code in isolation.
#pragma once
#include "render_scan.h"

namespace das3r {

// lane K of every 16-lane row, to all lanes of the row (DPP row_newbcast: gfx90a and later; folded into VOP2 consumers)
template <int K>
__device__ __forceinline__ float bc(const float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x150 + K, 0xf, 0xf, true));
}
// acc += x[lane K of the row] * v as ONE v_fmac_f32_dpp.  (The compiler folds a broadcast into its consumer only when that is the
// broadcast's single use; dL/dpixel is used twice per step — in c . dL/dpix and here — and came out as v_mov_b32_dpp + fmac.)
// x is a per-pixel constant: never written inside the walk, so the DPP read needs no wait states.
template <int K>
__device__ __forceinline__ void fmac_bc(float &acc, const float x, const float v) {
    asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(v), "i"(K));
}
// Pixel lane K of every row takes the row's totals (lane 15) of `t15` and `r15`; the other lanes keep theirs.  One scalar move for
// the lane mask + two v_cndmask_b32_dpp.  `order`: a value computed AFTER t15 / r15 in program order (>= 2 VALU instructions later):
// a DPP read needs two wait states behind the VALU write of its source, and the assembler does not add them inside inline asm.
template <int K>
__device__ __forceinline__ void state_to_pixel_lane(float &stT, float &stR, const float t15, const float r15, const float order) {
    constexpr unsigned long long keep = ~(0x0001000100010001ull << K);   // vcc = 1: keep the old value
    asm("s_mov_b64 vcc, %5\n\t"
        "v_cndmask_b32_dpp %0, %2, %0, vcc row_newbcast:15 row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %1, %3, %1, vcc row_newbcast:15 row_mask:0xf bank_mask:0xf"
        : "+v"(stT), "+v"(stR)
        : "v"(t15), "v"(r15), "v"(order), "s"(keep)
        : "vcc");
}

// per-pixel registers of a pixel lane (lane s of row r owns pixel s of block r: x = s & 3, y = s >> 2)
struct PixelRegs {
    float pxf, pyf;            // PIX == 0: the lane's pixel centre; PIX > 0: the corner pixel of the lane's block (same on the 16 lanes of a row)
    float d0, d1, d2;          // dL/dpixel
    float T, R;                // replay state; R carries T_final * (bg . dL/dpixel) with it (only their sum is ever used)
    float lastrel;             // n_contrib relative to the round's staged window, as a float in [0, MB]
};
struct SplatRegs {
    float x, y;                // centre
    float A, B, C, o;          // conic, opacity (0 on lanes without an entry)
    float c0, c1, c2;          // colour
    float posrel;              // list position relative to the round's window (MB - 1 - j)
};
struct Sums {
    float C0, C1, C2, M0, Mu, Mv, Muu, Muv, Mvv;
};
// PIX > 0: the per-pixel values come out of LDS instead of the pixel lanes' registers.  Measured (tools/probes/valu_rate_probe.hip,
// saturated SIMD): a DPP-modified VALU instruction costs 1.76 ns of the SIMD against 0.96 ns for a plain one — folding a broadcast
// into its consumer is not free, it turns a full-rate instruction into a half-rate one.  Two broadcast ds_read per step (four
// addresses per instruction) keep those ~11 instructions per step at full rate.
//   PIX == 1: constants {d0, d1, d2, lastrel} from LDS (one ds_read_b128 per step), state (T, R) in the pixel lanes' registers
//   PIX == 2: state {T, R} from LDS too (ds_read_b64; lane 15 of every row writes the new state back with an exec-masked ds_write2_b32)
constexpr int PIX_CST_ROW = 16 * 16 + 16;   // bytes per block row of constants (+16: the four rows of a wave on different banks)
constexpr int PIX_ST_ROW = 16 * 8 + 8;      // bytes per block row of state

// The pair's two tests (power <= 0, alpha >= 1/255) and what hangs on them: alpha and G where both pass, 0 elsewhere.
// DAS3R_BLK_ARITH: by arithmetic, as the forward kernels take them (render_common.h alpha_if_visible: same decisions bit for bit) —
// v_cmp / v_cndmask issue at half rate on this chip; G of a dropped pair may be anything, inf included (a degenerate conic):
// min(G, alpha * 1e38) is G wherever alpha >= 1/255 (then G <= 1) and 0 where alpha is 0, NaN and inf included (v_min drops a NaN).
#ifdef DAS3R_BLK_ARITH
#define BLK_SELECT(a1, power, G, am_out, Gm_out)                                                                              \
    am_out = alpha_if_visible(a1, power);                                                                                     \
    Gm_out = fminf(G, __fmul_rn(am_out, 1e38f));
#else
#define BLK_SELECT(a1, power, G, am_out, Gm_out)                                                                              \
    {                                                                                                                         \
        const bool active = (!(power > 0.0f)) & (a1 >= (1.0f / 255.0f));                                                      \
        am_out = active ? a1 : 0.f;                                                                                           \
        Gm_out = active ? G : 0.f;                                                                                            \
    }
#endif

// One image row KY of the block: its four pixel steps K = 4 KY .. 4 KY + 3, interleaved.  pair_alpha's arithmetic, bit for bit
// (render_common.h), so that every pair takes the decision the forward kernel took.
// ABL (tools/probes/blk_loop_probe.hip only; results are wrong): 1 = without the two row scans, 2 = without the state hand-off
// MUT: exp(power) (1 + 1e-4) — das3r_debug_mutate(1), the biased kernel of tests/test_gpu_fullsize.py's mutation test
// cst / st: LDS byte addresses of the lane's block row of constants / state (PIX > 0 / PIX == 2)
template <int KY, int PIX = 0, int ABL = 0, bool MUT = false>
__device__ __forceinline__ void block_row(const SplatRegs &sp, PixelRegs &px, Sums &acc, const char *cst = nullptr, char *st = nullptr) {
    const float dy = PIX ? sp.y - (px.pyf + (float)KY) : sp.y - bc<4 * KY>(px.pyf);
    const float cyy = __fmul_rn(__fmul_rn(sp.C, dy), dy);
    float am[4], Gm[4], rinv[4], Pinc[4], T[4], cd[4], w[4], wc[4], Sinc[4], Rinc[4], g[4];
    float4 pc[4];      // PIX > 0: {d0, d1, d2, lastrel} of the step's pixel
    float2 pst[4];     // PIX == 2: {T, R}
#define ALPHA_STEP(U)                                                                                                         \
    {                                                                                                                         \
        constexpr int K = 4 * KY + U;                                                                                         \
        if constexpr (PIX > 0) pc[U] = *reinterpret_cast<const float4 *>(cst + K * 16);                                       \
        if constexpr (PIX == 2) pst[U] = *reinterpret_cast<const float2 *>(st + K * 8);                                       \
        const float dx = PIX ? sp.x - (px.pxf + (float)U) : sp.x - bc<K>(px.pxf);                                             \
        const float q = __fmaf_rn(__fmul_rn(sp.A, dx), dx, cyy);                                                              \
        const float power = __fmaf_rn(-0.5f, q, -__fmul_rn(__fmul_rn(sp.B, dx), dy));                                         \
        const float G = MUT ? __expf(power) * 1.0001f : __expf(power);   /* MUT: das3r_debug_mutate (the mutation the parity tests must catch) */ \
        /* position < n_contrib  <=>  lastrel - posrel >= 1, else <= 0 (small integers): a third operand of the alpha clamp —  */ \
        /* where the pair takes part the minimum is min(0.99, o G) as in pair_alpha, elsewhere it fails the 1/255 test         */ \
        const float lastrel = PIX ? pc[U].w : bc<K>(px.lastrel);                                                              \
        const float a1 = fminf(fminf(0.99f, __fmul_rn(sp.o, G)), lastrel - sp.posrel);                                        \
        BLK_SELECT(a1, power, G, am[U], Gm[U])                                                                                \
        rinv[U] = __builtin_amdgcn_rcpf(1.f - am[U]);                                                                         \
        Pinc[U] = rinv[U];                                                                                                    \
    }
    ALPHA_STEP(0) ALPHA_STEP(1) ALPHA_STEP(2) ALPHA_STEP(3)
#undef ALPHA_STEP
    if constexpr (ABL != 1) row_scan_mul_x4(Pinc[0], Pinc[1], Pinc[2], Pinc[3]);   // lane s: product of 1 / (1 - alpha) over splats 0..s of the batch
#define W_STEP(U)                                                                                                             \
    {                                                                                                                         \
        constexpr int K = 4 * KY + U;                                                                                         \
        T[U] = (PIX == 2 ? pst[U].x : bc<K>(px.T)) * Pinc[U];   /* transmittance in front of splat s at pixel K */            \
        w[U] = am[U] * T[U];                                                                                                  \
        if constexpr (PIX > 0) {                                                                                              \
            cd[U] = sp.c0 * pc[U].x + sp.c1 * pc[U].y + sp.c2 * pc[U].z;                                                      \
        } else {                                                                                                              \
            cd[U] = sp.c0 * bc<K>(px.d0);                                                                                     \
            fmac_bc<K>(cd[U], px.d1, sp.c1);                                                                                  \
            fmac_bc<K>(cd[U], px.d2, sp.c2);                                                                                  \
        }                                                                                                                     \
        wc[U] = cd[U] * w[U];                                                                                                 \
        Sinc[U] = wc[U];                                                                                                      \
    }
    W_STEP(0) W_STEP(1) W_STEP(2) W_STEP(3)
#undef W_STEP
    if constexpr (ABL != 1) row_scan_add_x4(Sinc[0], Sinc[1], Sinc[2], Sinc[3]);   // lane s: sum of w (c . dL/dpix) over splats 0..s of the batch
#define G_STEP(U)                                                                                                             \
    {                                                                                                                         \
        constexpr int K = 4 * KY + U;                                                                                         \
        Rinc[U] = (PIX == 2 ? pst[U].y : bc<K>(px.R)) + Sinc[U];   /* lane 15: the pixel's R (+ tfbg) behind the next batch */ \
        const float Rex = Rinc[U] - wc[U];          /* R (+ tfbg) behind splat s */                                           \
        const float dL_dalpha = T[U] * cd[U] - Rex * rinv[U];                                                                 \
        g[U] = Gm[U] * dL_dalpha;                                                                                             \
        if constexpr (PIX > 0) {                                                                                              \
            acc.C0 += w[U] * pc[U].x;                                                                                         \
            acc.C1 += w[U] * pc[U].y;                                                                                         \
            acc.C2 += w[U] * pc[U].z;                                                                                         \
        } else {                                                                                                              \
            fmac_bc<K>(acc.C0, px.d0, w[U]);                                                                                  \
            fmac_bc<K>(acc.C1, px.d1, w[U]);                                                                                  \
            fmac_bc<K>(acc.C2, px.d2, w[U]);                                                                                  \
        }                                                                                                                     \
        acc.M0 += g[U];                                                                                                       \
        if (U > 0) acc.Mu += (float)U * g[U];                                                                                 \
        if (KY > 0) acc.Mv += (float)KY * g[U];                                                                               \
        if (U > 0) acc.Muu += (float)(U * U) * g[U];                                                                          \
        if (U > 0 && KY > 0) acc.Muv += (float)(U * KY) * g[U];                                                               \
        if (KY > 0) acc.Mvv += (float)(KY * KY) * g[U];                                                                       \
    }
    G_STEP(0) G_STEP(1) G_STEP(2) G_STEP(3)
#undef G_STEP
    if constexpr (ABL == 2) { px.T += T[0] + T[1] + T[2] + T[3]; px.R += Rinc[0] + Rinc[1] + Rinc[2] + Rinc[3]; return; }
    if constexpr (PIX == 2) {
        const unsigned long long lanes15 = 0x8000800080008000ull;
        const uint32_t sa = (uint32_t)(uintptr_t)st;   // (the low 32 bits of a generic pointer into LDS are its LDS address)
        store2_lane15<2 * (4 * KY + 0)>(sa, T[0], Rinc[0], lanes15);
        store2_lane15<2 * (4 * KY + 1)>(sa, T[1], Rinc[1], lanes15);
        store2_lane15<2 * (4 * KY + 2)>(sa, T[2], Rinc[2], lanes15);
        store2_lane15<2 * (4 * KY + 3)>(sa, T[3], Rinc[3], lanes15);
    } else {
        state_to_pixel_lane<4 * KY + 0>(px.T, px.R, T[0], Rinc[0], acc.M0);
        state_to_pixel_lane<4 * KY + 1>(px.T, px.R, T[1], Rinc[1], acc.M0);
        state_to_pixel_lane<4 * KY + 2>(px.T, px.R, T[2], Rinc[2], acc.M0);
        state_to_pixel_lane<4 * KY + 3>(px.T, px.R, T[3], Rinc[3], acc.M0);
    }
}

}  // namespace das3r
