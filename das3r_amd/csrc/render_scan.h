// render_scan.h — helpers shared by the backward compositing kernels whose waves are laid out as 4 pixels x 16 splats
// (render_bwd_scan.hip: workgroup-synchronous rounds; render_bwd_stream.hip: independent waves).
#pragma once
#include "render_common.h"

namespace das3r {


constexpr int NACC = 9;   // C0, C1, C2, M0, Mu, Mv, Muu, Muv, Mvv
typedef float v4f __attribute__((ext_vector_type(4)));

// One row of partial[] (nine floats, 4-byte aligned: the row stride is 36 bytes) leaves as 16 + 16 + 4 bytes instead of nine scattered words:
// the rows of a tile's entries lie anywhere in the buffer (slots are handed out in splat or depth order), every lane's store is its own
// request, and a third as many of them go through the memory pipeline.  (global_store_dwordx4 needs dword alignment only.)
typedef float v4f_u __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ void store_partial_row(float *__restrict__ rowp, const float r0, const float r1, const float r2, const float r3, const float r4,
                                                  const float r5, const float r6, const float r7, const float r8) {
    *reinterpret_cast<v4f_u *>(rowp) = v4f_u{r0, r1, r2, r3};
    *reinterpret_cast<v4f_u *>(rowp + 4) = v4f_u{r4, r5, r6, r7};
    rowp[8] = r8;
}

struct PixRow {   // one pixel of a wave's quadrant (32 B, two ds_read_b128)
    float dLp0, dLp1, dLp2, tfbg;   // dL/dpixel, T_final * (bg . dL/dpixel)
    float T, R;                     // replay state (render_common.h: ReplayState)
    uint32_t last;                  // n_contrib: list positions >= last take no part
    float zero;                     // (A operand of the lanes that carry no colour row)
};

// inclusive product / sum over the lanes of each 16-lane DPP row (Hillis-Steele, row_shr 1, 2, 4, 8)
// product: a lane without a source lane must keep its value, which the update_dpp builtin only offers as mov + mov_dpp + mul;
// v_mul_f32_dpp with bound_ctrl off leaves exactly those lanes unwritten.  Four independent chains per statement: a DPP read of
// a register needs two wait states after the VALU write, here the three other chains' instructions.
__device__ __forceinline__ void row_scan_mul_x4(float &x0, float &x1, float &x2, float &x3) {
    asm("s_nop 1\n\t"
        "v_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %2, %2, %2 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %3, %3, %3 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %2, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %3, %3, %3 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %2, %2, %2 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %3, %3, %3 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0"
        : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
}
// sum: the same network (a lane without a source lane keeps its value — what an inclusive scan wants).  Written out like the
// product because the compiler serialises the four chains of a group and pads every step with s_nop.
__device__ __forceinline__ void row_scan_add_x4(float &x0, float &x1, float &x2, float &x3) {
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %3, %3, %3 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %3, %3, %3 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %3, %3, %3 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0"
        : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
}

// Tight wave-level cull: can the splat reach alpha >= 1/255 on any point of the rectangle spanned by the pixel centres
// [x0, x0 + 7] x [y0, y0 + 7]?  The bounding-box test (render_common.h: quadrant_hit) passes every splat whose axis-aligned extent
// overlaps the quadrant — for elongated, rotated splats more than the ellipse itself does.  With d = centre - pixel,
// q(d) = A dx^2 + 2 B dx dy + C dy^2 is convex; alpha >= 1/255  <=>  q <= 2 ln(255 o) =: tau.  min of q over the rectangle: 0 if the
// centre is inside, else the smallest of the four edge minima (one coordinate fixed, the other = clamp of its stationary point).
// tau is taken from the opacity exactly as preprocess.hip takes it for the extents (2 ln(255 o) 1.0005 + 1e-3: its own margin) —
// NOT recovered from the stored extent through Sigma_xx = C / (A C - B^2): for long thin splats A C - B^2 cancels in fp32 (6.6 %
// error of tau at a 1000 px major sigma), and a pair the forward blended with alpha just above 1/255 could lose its
// gradient.
__device__ __forceinline__ float splat_tau(const float opacity) { return 2.0f * __logf(255.0f * opacity) * 1.0005f + 1e-3f; }
__device__ __forceinline__ bool rect_hit_tight_tau(const float4 xyh, const float4 co, const float tau, const float x0, const float y0) {
    const float dxl = xyh.x - (x0 + 7.0f), dxh = xyh.x - x0, dyl = xyh.y - (y0 + 7.0f), dyh = xyh.y - y0;
    const float A = co.x, B = co.y, C = co.z;
    const float rA = __builtin_amdgcn_rcpf(A), rC = __builtin_amdgcn_rcpf(C);
    auto qf = [&](const float dx, const float dy) { return A * dx * dx + 2.0f * B * dx * dy + C * dy * dy; };
    const float q0 = qf(dxl, __builtin_amdgcn_fmed3f(-B * dxl * rC, dyl, dyh));
    const float q1 = qf(dxh, __builtin_amdgcn_fmed3f(-B * dxh * rC, dyl, dyh));
    const float q2 = qf(__builtin_amdgcn_fmed3f(-B * dyl * rA, dxl, dxh), dyl);
    const float q3 = qf(__builtin_amdgcn_fmed3f(-B * dyh * rA, dxl, dxh), dyh);
    const bool inside = (dxl <= 0.f) & (dxh >= 0.f) & (dyl <= 0.f) & (dyh >= 0.f);
    const float qmin = fminf(fminf(q0, q1), fminf(q2, q3));
    return (xyh.z > 0.f) & (inside | (qmin * 0.9999f - 1e-3f <= tau));
}
__device__ __forceinline__ bool rect_hit_tight(const float4 xyh, const float4 co, const float x0, const float y0) {
    return rect_hit_tight_tau(xyh, co, splat_tau(co.w), x0, y0);
}

typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
struct U4 { uint32_t x, y, z, w; };
__device__ __forceinline__ v8bf as_bf8(const U4 u) { return __builtin_bit_cast(v8bf, u); }
// two fp32 -> their upper halves (bf16 by truncation) side by side: a in the low half, b in the high half
__device__ __forceinline__ uint32_t pack_hi(const float a, const float b) {
    return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
}
// Lane 15 of every DPP row stores two floats at LDS byte address `addr` + 4 * DW0 (the pixel's new state) — without the
// s_and_saveexec / branch-out-of-line / s_or pair a divergent `if` costs per step.  Only valid where every lane is active.
template <int DW0>
__device__ __forceinline__ void store2_lane15(const uint32_t addr, const float a, const float b, const unsigned long long lanes15) {
    asm volatile("s_mov_b64 exec, %3\n\tds_write2_b32 %0, %1, %2 offset0:%4 offset1:%5\n\ts_mov_b64 exec, -1"
                 :: "v"(addr), "v"(a), "v"(b), "s"(lanes15), "i"(DW0), "i"(DW0 + 1) : "memory");
}
__device__ __forceinline__ float lo_part(const float x) { return x - __uint_as_float(__float_as_uint(x) & 0xFFFF0000u); }   // exact

}  // namespace das3r
