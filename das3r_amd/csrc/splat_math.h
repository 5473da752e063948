// splat_math.h — per-Gaussian device maths shared by the forward and backward preprocess kernels.
// Spec: SURVEY.md Appendix A.1-A.5 / A.8 (upstream:cuda_rasterizer/forward.cu, backward.cu, auxiliary.h).
// Files including this header are compiled with -ffp-contract=off (see Makefile) so every expression below is
// evaluated as written in IEEE fp32.
#pragma once
#include <hip/hip_runtime.h>

namespace das3r {

__device__ constexpr float SH_C0 = 0.28209479177387814f;
__device__ constexpr float SH_C1 = 0.4886025119029199f;
__device__ constexpr float SH_C2_0 = 1.0925484305920792f, SH_C2_1 = -1.0925484305920792f, SH_C2_2 = 0.31539156525252005f,
                           SH_C2_3 = -1.0925484305920792f, SH_C2_4 = 0.5462742152960396f;
__device__ constexpr float SH_C3_0 = -0.5900435899266435f, SH_C3_1 = 2.890611442640554f, SH_C3_2 = -0.4570457994644658f,
                           SH_C3_3 = 0.3731763325901154f, SH_C3_4 = -0.4570457994644658f, SH_C3_5 = 1.445305721320277f,
                           SH_C3_6 = -0.5900435899266435f;

// p_row @ M[:, :3] and p_row @ M for a row-vector-layout 4x4 (element [r][c] at 4r+c)
__device__ __forceinline__ float3 xform43(const float3 p, const float *m) {
    return make_float3(m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
                       m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]);
}
__device__ __forceinline__ float4 xform44(const float3 p, const float *m) {
    return make_float4(m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
                       m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14], m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15]);
}

// rotation matrix of the quaternion (r,x,y,z) AS GIVEN (no normalisation — DAS3R feeds raw parameters,
// /root/reference/gaussian_renderer/__init__.py:91,108)
__device__ __forceinline__ void quat_to_R(const float4 q, float R[3][3]) {
    const float r = q.x, x = q.y, y = q.z, z = q.w;
    R[0][0] = 1.f - 2.f * (y * y + z * z); R[0][1] = 2.f * (x * y - r * z);       R[0][2] = 2.f * (x * z + r * y);
    R[1][0] = 2.f * (x * y + r * z);       R[1][1] = 1.f - 2.f * (x * x + z * z); R[1][2] = 2.f * (y * z - r * x);
    R[2][0] = 2.f * (x * z - r * y);       R[2][1] = 2.f * (y * z + r * x);       R[2][2] = 1.f - 2.f * (x * x + y * y);
}

// Sigma = (R S)(R S)^T, upper triangle (00,01,02,11,12,22)
__device__ __forceinline__ void cov3d_from_scale_rot(const float3 scale, const float mod, const float4 q, float *cov) {
    float R[3][3];
    quat_to_R(q, R);
    const float s[3] = {mod * scale.x, mod * scale.y, mod * scale.z};
    float Mm[3][3];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int k = 0; k < 3; k++) Mm[i][k] = R[i][k] * s[k];
#define SIG(i, j) (Mm[i][0] * Mm[j][0] + Mm[i][1] * Mm[j][1] + Mm[i][2] * Mm[j][2])
    cov[0] = SIG(0, 0); cov[1] = SIG(0, 1); cov[2] = SIG(0, 2); cov[3] = SIG(1, 1); cov[4] = SIG(1, 2); cov[5] = SIG(2, 2);
#undef SIG
}

// EWA: clamp the view-space point to 1.3x the frustum, T = J * Rcw (2x3) with Rcw[i][j] = V[4j+i].
__device__ __forceinline__ void ewa_T(const float3 p_view, const float *V, const float focal_x, const float focal_y,
                                      const float tanfovx, const float tanfovy, float T[2][3], float3 &t, bool &clampx, bool &clampy) {
    t = p_view;
    const float limx = 1.3f * tanfovx, limy = 1.3f * tanfovy;
    const float txtz = t.x / t.z, tytz = t.y / t.z;
    clampx = (txtz < -limx || txtz > limx);
    clampy = (tytz < -limy || tytz > limy);
    t.x = fminf(limx, fmaxf(-limx, txtz)) * t.z;
    t.y = fminf(limy, fmaxf(-limy, tytz)) * t.z;
    const float J00 = focal_x / t.z, J02 = -(focal_x * t.x) / (t.z * t.z);
    const float J11 = focal_y / t.z, J12 = -(focal_y * t.y) / (t.z * t.z);
#pragma unroll
    for (int j = 0; j < 3; j++) {
        T[0][j] = J00 * V[4 * j + 0] + 0.f * V[4 * j + 1] + J02 * V[4 * j + 2];
        T[1][j] = 0.f * V[4 * j + 0] + J11 * V[4 * j + 1] + J12 * V[4 * j + 2];
    }
}

// cov2D = T Sigma T^T + 0.3 I  ->  (a, b, c)
__device__ __forceinline__ void cov2d_from_T(const float T[2][3], const float *c3, float &a, float &b, float &c) {
    const float S[3][3] = {{c3[0], c3[1], c3[2]}, {c3[1], c3[3], c3[4]}, {c3[2], c3[4], c3[5]}};
    float TS[2][3];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) TS[i][j] = T[i][0] * S[0][j] + T[i][1] * S[1][j] + T[i][2] * S[2][j];
    a = TS[0][0] * T[0][0] + TS[0][1] * T[0][1] + TS[0][2] * T[0][2] + 0.3f;
    b = TS[0][0] * T[1][0] + TS[0][1] * T[1][1] + TS[0][2] * T[1][2];
    c = TS[1][0] * T[1][0] + TS[1][1] * T[1][1] + TS[1][2] * T[1][2] + 0.3f;
}

// tile rectangle touched by a 2D splat of integer radius r (C truncation toward zero, clamped to the grid)
__device__ __forceinline__ void tile_rect(const float px, const float py, const int r, const int tiles_x, const int tiles_y,
                                          int &rminx, int &rminy, int &rmaxx, int &rmaxy) {
    rminx = min(tiles_x, max(0, (int)((px - r) / TILE_X)));
    rminy = min(tiles_y, max(0, (int)((py - r) / TILE_Y)));
    rmaxx = min(tiles_x, max(0, (int)((px + r + TILE_X - 1) / TILE_X)));
    rmaxy = min(tiles_y, max(0, (int)((py + r + TILE_Y - 1) / TILE_Y)));
}

// Tile rectangle actually binned: the upstream 3-sigma square clipped to the axis-aligned box (half extents hx, hy, already
// padded) outside which alpha = opacity*exp(power) < 1/255 on every pixel centre.  Tiles dropped by the clip can only hold
// pairs upstream would `continue` on, so image and gradients are unchanged; only list lengths / num_rendered shrink.
// tight == false reproduces upstream's lists exactly (used by the bit-exact binning tests).
__device__ __forceinline__ void binned_rect(const float4 xyh, const int r, const int tiles_x, const int tiles_y, const bool tight,
                                            int &rminx, int &rminy, int &rmaxx, int &rmaxy) {
    tile_rect(xyh.x, xyh.y, r, tiles_x, tiles_y, rminx, rminy, rmaxx, rmaxy);
    if (tight) {
        if (xyh.z < 0.f) {  // opacity <= 1/255: never visible
            rmaxx = rminx;
            return;
        }
        // pixel centres are integers: the splat can only matter on the centres ceil(x - hx) .. floor(x + hx) (and likewise in y);
        // an interval that contains no integer touches nothing.  The tile of pixel p is p >> 4.
        const float px0 = ceilf(xyh.x - xyh.z), px1 = floorf(xyh.x + xyh.z);
        const float py0 = ceilf(xyh.y - xyh.w), py1 = floorf(xyh.y + xyh.w);
        if (px1 < px0 || py1 < py0 || px1 < 0.f || py1 < 0.f) {
            rmaxx = rminx;
            return;
        }
        const int tx0 = (int)fmaxf(px0, 0.f) >> 4, ty0 = (int)fmaxf(py0, 0.f) >> 4;
        const int tx1 = ((int)fminf(px1, 1.0e6f) >> 4) + 1, ty1 = ((int)fminf(py1, 1.0e6f) >> 4) + 1;
        rminx = max(rminx, tx0);
        rminy = max(rminy, ty0);
        rmaxx = max(rminx, min(rmaxx, tx1));
        rmaxy = max(rminy, min(rmaxy, ty1));
    }
}

// Load the first n3 = 3*(D+1)^2 floats of a (M,3) SH row into registers.  Rows are 16-byte aligned when
// 3*M is a multiple of 4 (M = 4 or 16): read float4; otherwise scalar.
__device__ __forceinline__ void load_sh_row(const float *__restrict__ row, const int D, const bool vec_ok, float sh[48]) {
    const int n3 = 3 * (D + 1) * (D + 1);
    if (vec_ok) {
        const float4 *r4 = reinterpret_cast<const float4 *>(row);
#pragma unroll
        for (int i = 0; i < 12; i++) {
            if (4 * i < n3) {
                const float4 v = r4[i];
                sh[4 * i] = v.x; sh[4 * i + 1] = v.y; sh[4 * i + 2] = v.z; sh[4 * i + 3] = v.w;
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < 48; i++)
            if (i < n3) sh[i] = row[i];
    }
}

// SH -> RGB (+0.5, clamp at 0).  Basis as /root/reference/utils/sh_utils.py:74-100.  clamp_bits: bit c = channel c clamped.
__device__ __forceinline__ float3 sh_eval(const int deg, const float sh[48], const float x, const float y, const float z) {
    float out[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
#define SHC(k) sh[(k) * 3 + c]
        float r = SH_C0 * SHC(0);
        if (deg > 0) {
            r = r - SH_C1 * y * SHC(1) + SH_C1 * z * SHC(2) - SH_C1 * x * SHC(3);
            if (deg > 1) {
                const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                r = r + SH_C2_0 * xy * SHC(4) + SH_C2_1 * yz * SHC(5) + SH_C2_2 * (2.f * zz - xx - yy) * SHC(6) +
                    SH_C2_3 * xz * SHC(7) + SH_C2_4 * (xx - yy) * SHC(8);
                if (deg > 2) {
                    r = r + SH_C3_0 * y * (3.f * xx - yy) * SHC(9) + SH_C3_1 * xy * z * SHC(10) +
                        SH_C3_2 * y * (4.f * zz - xx - yy) * SHC(11) + SH_C3_3 * z * (2.f * zz - 3.f * xx - 3.f * yy) * SHC(12) +
                        SH_C3_4 * x * (4.f * zz - xx - yy) * SHC(13) + SH_C3_5 * z * (xx - yy) * SHC(14) +
                        SH_C3_6 * x * (xx - 3.f * yy) * SHC(15);
                }
            }
        }
#undef SHC
        out[c] = r + 0.5f;
    }
    return make_float3(out[0], out[1], out[2]);
}

// float4 reads are legal when the row is 16-byte aligned and the padded read stays inside the row
__device__ __forceinline__ bool sh_vec_ok(const float *row, const int D, const int M) {
    const int n3 = 3 * (D + 1) * (D + 1);
    return (((uintptr_t)row & 15) == 0) && (((n3 + 3) & ~3) <= 3 * M);
}

__device__ __forceinline__ float3 sh_regs_to_rgb(const int D, const float sh[48], const float3 p, const float *campos,
                                                 uint8_t &clamp_bits) {
    const float dx = p.x - campos[0], dy = p.y - campos[1], dz = p.z - campos[2];
    const float inv = 1.f / sqrtf(dx * dx + dy * dy + dz * dz);
    const float3 r = sh_eval(D, sh, dx * inv, dy * inv, dz * inv);
    clamp_bits = (uint8_t)((r.x < 0.f ? 1 : 0) | (r.y < 0.f ? 2 : 0) | (r.z < 0.f ? 4 : 0));
    return make_float3(fmaxf(r.x, 0.f), fmaxf(r.y, 0.f), fmaxf(r.z, 0.f));
}

__device__ __forceinline__ float3 sh_to_rgb(const int D, const int M, const float *__restrict__ row, const float3 p,
                                            const float *campos, uint8_t &clamp_bits) {
    float sh[48];
    load_sh_row(row, D, sh_vec_ok(row, D, M), sh);
    const float dx = p.x - campos[0], dy = p.y - campos[1], dz = p.z - campos[2];
    const float inv = 1.f / sqrtf(dx * dx + dy * dy + dz * dz);
    const float3 r = sh_eval(D, sh, dx * inv, dy * inv, dz * inv);
    clamp_bits = (uint8_t)((r.x < 0.f ? 1 : 0) | (r.y < 0.f ? 2 : 0) | (r.z < 0.f ? 4 : 0));
    return make_float3(fmaxf(r.x, 0.f), fmaxf(r.y, 0.f), fmaxf(r.z, 0.f));
}

}  // namespace das3r
