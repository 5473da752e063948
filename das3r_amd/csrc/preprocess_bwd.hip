// preprocess_bwd.hip — K8+K9 fused: per-Gaussian chain rule from (dL/dmean2D, dL/dconic, dL/dcolour) to
// (dL/dmean3D, dL/dscale, dL/drotation, dL/dSH) or dL/dcov3D.  Replaces upstream:cuda_rasterizer/backward.cu
// computeCov2DCUDA + preprocessCUDA (+ computeColorFromSH / computeCov3D backward) — SURVEY.md A.8.
//
// HBM-bound streaming map.  Fusion removes the dL_dcov3D round trip (24 B/G written then re-read upstream) and the
// pre-zeroing of every output: each lane writes ALL of its Gaussian's gradient rows (zeros for culled Gaussians and
// for SH coefficients above the active degree), so the dense (P,M,3) dL_dsh tensor is touched exactly once.
#include "common.h"
#include "splat_math.h"

namespace das3r {

template <bool HAS_SH, bool HAS_COV>
__global__ void __launch_bounds__(256) preprocess_backward_kernel(
    int P, int D, int M, const float *__restrict__ means3D, const float *__restrict__ scales, float scale_modifier,
    const float *__restrict__ rotations, const float *__restrict__ shs, const float *__restrict__ cov3D_precomp,
    const float *__restrict__ viewmatrix, const float *__restrict__ projmatrix, const float *__restrict__ campos, int W, int H,
    float tanfovx, float tanfovy, const uint32_t *__restrict__ tiles_touched, const uint8_t *__restrict__ clamped,
    const float *__restrict__ partial /*[I,9]*/, const uint32_t *__restrict__ inv /*[I]*/, const uint32_t *__restrict__ off_by_gid,
    float *__restrict__ dL_dmeans2D /*[P,3] out*/, float *__restrict__ dL_dopacity /*[P] out*/,
    float *__restrict__ dL_dcolors_precomp /*[P,3] out, precomp mode*/, float *__restrict__ dL_dmeans3D,
    float *__restrict__ dL_dscales, float *__restrict__ dL_drot, float *__restrict__ dL_dsh, float *__restrict__ dL_dcov3D) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= P) return;
    const uint32_t ntiles_g = tiles_touched[idx];
    const bool visible = ntiles_g > 0;

    // gather this splat's per-instance sums (one row per touched tile) through the inverse permutation of the binning
    float acc[9];
#pragma unroll
    for (int q = 0; q < 9; q++) acc[q] = 0.f;
    if (visible) {
        const uint32_t e0 = off_by_gid[idx];
        for (uint32_t k = 0; k < ntiles_g; k++) {
            const float *row = partial + (size_t)inv[e0 + k] * 9;
#pragma unroll
            for (int q = 0; q < 9; q++) acc[q] += row[q];
        }
    }
    dL_dmeans2D[3 * (size_t)idx] = acc[3];
    dL_dmeans2D[3 * (size_t)idx + 1] = acc[4];
    dL_dmeans2D[3 * (size_t)idx + 2] = 0.f;
    dL_dopacity[idx] = acc[8];
    if (!HAS_SH) {
        dL_dcolors_precomp[3 * (size_t)idx] = acc[0];
        dL_dcolors_precomp[3 * (size_t)idx + 1] = acc[1];
        dL_dcolors_precomp[3 * (size_t)idx + 2] = acc[2];
    }

    float dmean[3] = {0.f, 0.f, 0.f};
    float dcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float dscale[3] = {0.f, 0.f, 0.f};
    float drot[4] = {0.f, 0.f, 0.f, 0.f};
    float dsh[48];
#pragma unroll
    for (int i = 0; i < 48; i++) dsh[i] = 0.f;

    if (visible) {
        float V[16], PM[16];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            V[i] = viewmatrix[i];
            PM[i] = projmatrix[i];
        }
        const float focal_x = W / (2.0f * tanfovx), focal_y = H / (2.0f * tanfovy);
        const float3 mean = make_float3(means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]);
        float3 sc = make_float3(0.f, 0.f, 0.f);
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
        float c3[6];
        if (HAS_COV) {
#pragma unroll
            for (int i = 0; i < 6; i++) c3[i] = cov3D_precomp[6 * idx + i];
        } else {
            sc = make_float3(scales[3 * idx], scales[3 * idx + 1], scales[3 * idx + 2]);
            q = reinterpret_cast<const float4 *>(rotations)[idx];
            cov3d_from_scale_rot(sc, scale_modifier, q, c3);
        }
        // ---- conic -> cov2D -> (cov3D, view-space mean)   [computeCov2DCUDA]
        {
            const float3 p_view = xform43(mean, V);
            float T[2][3];
            float3 t;
            bool cx, cy;
            ewa_T(p_view, V, focal_x, focal_y, tanfovx, tanfovy, T, t, cx, cy);
            const float x_grad_mul = cx ? 0.f : 1.f, y_grad_mul = cy ? 0.f : 1.f;
            const float S[3][3] = {{c3[0], c3[1], c3[2]}, {c3[1], c3[3], c3[4]}, {c3[2], c3[4], c3[5]}};
            float TS0[3], TS1[3];
#pragma unroll
            for (int j = 0; j < 3; j++) {
                TS0[j] = T[0][0] * S[j][0] + T[0][1] * S[j][1] + T[0][2] * S[j][2];
                TS1[j] = T[1][0] * S[j][0] + T[1][1] * S[j][1] + T[1][2] * S[j][2];
            }
            const float ca = TS0[0] * T[0][0] + TS0[1] * T[0][1] + TS0[2] * T[0][2] + 0.3f;
            const float cb = TS0[0] * T[1][0] + TS0[1] * T[1][1] + TS0[2] * T[1][2];
            const float cc = TS1[0] * T[1][0] + TS1[1] * T[1][1] + TS1[2] * T[1][2] + 0.3f;
            const float gA = acc[5], gB = acc[6], gC = acc[7];
            const float denom = ca * cc - cb * cb;
            float dL_da = 0.f, dL_db = 0.f, dL_dc = 0.f;
            const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
            if (denom2inv != 0.f) {
                dL_da = denom2inv * (-cc * cc * gA + 2 * cb * cc * gB + (denom - ca * cc) * gC);
                dL_dc = denom2inv * (-ca * ca * gC + 2 * ca * cb * gB + (denom - ca * cc) * gA);
                dL_db = denom2inv * 2 * (cb * cc * gA - (denom + 2 * cb * cb) * gB + ca * cb * gC);
                dcov[0] = (T[0][0] * T[0][0] * dL_da + T[0][0] * T[1][0] * dL_db + T[1][0] * T[1][0] * dL_dc);
                dcov[3] = (T[0][1] * T[0][1] * dL_da + T[0][1] * T[1][1] * dL_db + T[1][1] * T[1][1] * dL_dc);
                dcov[5] = (T[0][2] * T[0][2] * dL_da + T[0][2] * T[1][2] * dL_db + T[1][2] * T[1][2] * dL_dc);
                dcov[1] = 2 * T[0][0] * T[0][1] * dL_da + (T[0][0] * T[1][1] + T[0][1] * T[1][0]) * dL_db + 2 * T[1][0] * T[1][1] * dL_dc;
                dcov[2] = 2 * T[0][0] * T[0][2] * dL_da + (T[0][0] * T[1][2] + T[0][2] * T[1][0]) * dL_db + 2 * T[1][0] * T[1][2] * dL_dc;
                dcov[4] = 2 * T[0][2] * T[0][1] * dL_da + (T[0][1] * T[1][2] + T[0][2] * T[1][1]) * dL_db + 2 * T[1][1] * T[1][2] * dL_dc;
            }
            float dT0[3], dT1[3];
#pragma unroll
            for (int j = 0; j < 3; j++) {
                dT0[j] = 2 * TS0[j] * dL_da + TS1[j] * dL_db;
                dT1[j] = 2 * TS1[j] * dL_dc + TS0[j] * dL_db;
            }
            const float dJ00 = V[0] * dT0[0] + V[4] * dT0[1] + V[8] * dT0[2];
            const float dJ02 = V[2] * dT0[0] + V[6] * dT0[1] + V[10] * dT0[2];
            const float dJ11 = V[1] * dT1[0] + V[5] * dT1[1] + V[9] * dT1[2];
            const float dJ12 = V[2] * dT1[0] + V[6] * dT1[1] + V[10] * dT1[2];
            const float tz = 1.f / t.z, tz2 = tz * tz, tz3 = tz2 * tz;
            const float dtx = x_grad_mul * -focal_x * tz2 * dJ02;
            const float dty = y_grad_mul * -focal_y * tz2 * dJ12;
            const float dtz = -focal_x * tz2 * dJ00 - focal_y * tz2 * dJ11 + (2 * focal_x * t.x) * tz3 * dJ02 + (2 * focal_y * t.y) * tz3 * dJ12;
            dmean[0] = V[0] * dtx + V[1] * dty + V[2] * dtz;
            dmean[1] = V[4] * dtx + V[5] * dty + V[6] * dtz;
            dmean[2] = V[8] * dtx + V[9] * dty + V[10] * dtz;
        }
        // ---- screen-space mean -> 3D mean through the perspective projection   [preprocessCUDA bwd]
        {
            const float4 m_hom = xform44(mean, PM);
            const float m_w = 1.0f / (m_hom.w + 0.0000001f);
            const float mul1 = (PM[0] * mean.x + PM[4] * mean.y + PM[8] * mean.z + PM[12]) * m_w * m_w;
            const float mul2 = (PM[1] * mean.x + PM[5] * mean.y + PM[9] * mean.z + PM[13]) * m_w * m_w;
            const float gx = acc[3], gy = acc[4];
            dmean[0] += (PM[0] * m_w - PM[3] * mul1) * gx + (PM[1] * m_w - PM[3] * mul2) * gy;
            dmean[1] += (PM[4] * m_w - PM[7] * mul1) * gx + (PM[5] * m_w - PM[7] * mul2) * gy;
            dmean[2] += (PM[8] * m_w - PM[11] * mul1) * gx + (PM[9] * m_w - PM[11] * mul2) * gy;
        }
        // ---- colour -> SH coefficients and view direction   [computeColorFromSH bwd]
        if (HAS_SH) {
            const float *row = shs + (size_t)idx * M * 3;
            float sh[48];
            load_sh_row(row, D, sh_vec_ok(row, D, M), sh);
            const float dox = mean.x - campos[0], doy = mean.y - campos[1], doz = mean.z - campos[2];
            const float inv = 1.f / sqrtf(dox * dox + doy * doy + doz * doz);
            const float x = dox * inv, y = doy * inv, z = doz * inv;
            const uint8_t cl = clamped[idx];
            float g[3];
#pragma unroll
            for (int c = 0; c < 3; c++) g[c] = ((cl >> c) & 1) ? 0.f : acc[c];
            float ddir[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 3; c++) {
#define SHC(k) sh[(k) * 3 + c]
#define DSH(k) dsh[(k) * 3 + c]
                float dRdx = 0.f, dRdy = 0.f, dRdz = 0.f;
                DSH(0) = SH_C0 * g[c];
                if (D > 0) {
                    DSH(1) = -SH_C1 * y * g[c];
                    DSH(2) = SH_C1 * z * g[c];
                    DSH(3) = -SH_C1 * x * g[c];
                    dRdx = -SH_C1 * SHC(3);
                    dRdy = -SH_C1 * SHC(1);
                    dRdz = SH_C1 * SHC(2);
                    if (D > 1) {
                        const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                        DSH(4) = SH_C2_0 * xy * g[c];
                        DSH(5) = SH_C2_1 * yz * g[c];
                        DSH(6) = SH_C2_2 * (2.f * zz - xx - yy) * g[c];
                        DSH(7) = SH_C2_3 * xz * g[c];
                        DSH(8) = SH_C2_4 * (xx - yy) * g[c];
                        dRdx += SH_C2_0 * y * SHC(4) + SH_C2_2 * 2.f * -x * SHC(6) + SH_C2_3 * z * SHC(7) + SH_C2_4 * 2.f * x * SHC(8);
                        dRdy += SH_C2_0 * x * SHC(4) + SH_C2_1 * z * SHC(5) + SH_C2_2 * 2.f * -y * SHC(6) + SH_C2_4 * 2.f * -y * SHC(8);
                        dRdz += SH_C2_1 * y * SHC(5) + SH_C2_2 * 2.f * 2.f * z * SHC(6) + SH_C2_3 * x * SHC(7);
                        if (D > 2) {
                            DSH(9) = SH_C3_0 * y * (3.f * xx - yy) * g[c];
                            DSH(10) = SH_C3_1 * xy * z * g[c];
                            DSH(11) = SH_C3_2 * y * (4.f * zz - xx - yy) * g[c];
                            DSH(12) = SH_C3_3 * z * (2.f * zz - 3.f * xx - 3.f * yy) * g[c];
                            DSH(13) = SH_C3_4 * x * (4.f * zz - xx - yy) * g[c];
                            DSH(14) = SH_C3_5 * z * (xx - yy) * g[c];
                            DSH(15) = SH_C3_6 * x * (xx - 3.f * yy) * g[c];
                            dRdx += SH_C3_0 * SHC(9) * 3.f * 2.f * xy + SH_C3_1 * SHC(10) * yz + SH_C3_2 * SHC(11) * -2.f * xy +
                                    SH_C3_3 * SHC(12) * -3.f * 2.f * xz + SH_C3_4 * SHC(13) * (-3.f * xx + 4.f * zz - yy) +
                                    SH_C3_5 * SHC(14) * 2.f * xz + SH_C3_6 * SHC(15) * 3.f * (xx - yy);
                            dRdy += SH_C3_0 * SHC(9) * 3.f * (xx - yy) + SH_C3_1 * SHC(10) * xz +
                                    SH_C3_2 * SHC(11) * (-3.f * yy + 4.f * zz - xx) + SH_C3_3 * SHC(12) * -3.f * 2.f * yz +
                                    SH_C3_4 * SHC(13) * -2.f * xy + SH_C3_5 * SHC(14) * -2.f * yz + SH_C3_6 * SHC(15) * -3.f * 2.f * xy;
                            dRdz += SH_C3_1 * SHC(10) * xy + SH_C3_2 * SHC(11) * 4.f * 2.f * yz +
                                    SH_C3_3 * SHC(12) * 3.f * (2.f * zz - xx - yy) + SH_C3_4 * SHC(13) * 4.f * 2.f * xz +
                                    SH_C3_5 * SHC(14) * (xx - yy);
                        }
                    }
                }
#undef SHC
#undef DSH
                ddir[0] += dRdx * g[c];
                ddir[1] += dRdy * g[c];
                ddir[2] += dRdz * g[c];
            }
            // gradient through dir = v / |v|
            const float sum2 = dox * dox + doy * doy + doz * doz;
            const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
            dmean[0] += ((+sum2 - dox * dox) * ddir[0] - doy * dox * ddir[1] - doz * dox * ddir[2]) * invsum32;
            dmean[1] += (-dox * doy * ddir[0] + (sum2 - doy * doy) * ddir[1] - doz * doy * ddir[2]) * invsum32;
            dmean[2] += (-dox * doz * ddir[0] - doy * doz * ddir[1] + (sum2 - doz * doz) * ddir[2]) * invsum32;
        }
        // ---- cov3D -> scale, quaternion   [computeCov3D bwd]; Sigma = M M^T, M = R diag(mod * s)
        if (!HAS_COV) {
            float R[3][3];
            quat_to_R(q, R);
            const float sv[3] = {scale_modifier * sc.x, scale_modifier * sc.y, scale_modifier * sc.z};
            const float Gs[3][3] = {{dcov[0], 0.5f * dcov[1], 0.5f * dcov[2]}, {0.5f * dcov[1], dcov[3], 0.5f * dcov[4]},
                                    {0.5f * dcov[2], 0.5f * dcov[4], dcov[5]}};
            float Mm[3][3], dM[3][3], dR[3][3];
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int k = 0; k < 3; k++) Mm[i][k] = R[i][k] * sv[k];
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int k = 0; k < 3; k++) dM[i][k] = 2.f * (Gs[i][0] * Mm[0][k] + Gs[i][1] * Mm[1][k] + Gs[i][2] * Mm[2][k]);
#pragma unroll
            for (int k = 0; k < 3; k++) dscale[k] = R[0][k] * dM[0][k] + R[1][k] * dM[1][k] + R[2][k] * dM[2][k];
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int k = 0; k < 3; k++) dR[i][k] = dM[i][k] * sv[k];
            const float r = q.x, x = q.y, y = q.z, z = q.w;
            drot[0] = 2 * z * (dR[1][0] - dR[0][1]) + 2 * y * (dR[0][2] - dR[2][0]) + 2 * x * (dR[2][1] - dR[1][2]);
            drot[1] = 2 * y * (dR[0][1] + dR[1][0]) + 2 * z * (dR[0][2] + dR[2][0]) + 2 * r * (dR[2][1] - dR[1][2]) - 4 * x * (dR[2][2] + dR[1][1]);
            drot[2] = 2 * x * (dR[0][1] + dR[1][0]) + 2 * r * (dR[0][2] - dR[2][0]) + 2 * z * (dR[2][1] + dR[1][2]) - 4 * y * (dR[2][2] + dR[0][0]);
            drot[3] = 2 * r * (dR[1][0] - dR[0][1]) + 2 * x * (dR[0][2] + dR[2][0]) + 2 * y * (dR[2][1] + dR[1][2]) - 4 * z * (dR[1][1] + dR[0][0]);
        }
    }

    dL_dmeans3D[3 * (size_t)idx] = dmean[0];
    dL_dmeans3D[3 * (size_t)idx + 1] = dmean[1];
    dL_dmeans3D[3 * (size_t)idx + 2] = dmean[2];
    if (HAS_COV) {
#pragma unroll
        for (int i = 0; i < 6; i++) dL_dcov3D[6 * (size_t)idx + i] = dcov[i];
    } else {
#pragma unroll
        for (int k = 0; k < 3; k++) dL_dscales[3 * (size_t)idx + k] = dscale[k];
        reinterpret_cast<float4 *>(dL_drot)[idx] = make_float4(drot[0], drot[1], drot[2], drot[3]);
    }
    if (HAS_SH) {
        float *row = dL_dsh + (size_t)idx * M * 3;
        if ((((uintptr_t)row & 15) == 0) && ((3 * M) % 4 == 0) && 3 * M <= 48) {
#pragma unroll
            for (int i = 0; i < 12; i++)
                if (4 * i < 3 * M) reinterpret_cast<float4 *>(row)[i] = make_float4(dsh[4 * i], dsh[4 * i + 1], dsh[4 * i + 2], dsh[4 * i + 3]);
        } else {
#pragma unroll
            for (int i = 0; i < 48; i++)
                if (i < 3 * M) row[i] = dsh[i];
        }
    }
}

int launch_preprocess_backward(const das3r_raster_args *a, const das3r_raster_in *in, char *geom, char *binning, const Layout &L,
                               const das3r_raster_grads *g, const float *partial, hipStream_t s) {
    const int P = a->P;
    if (P == 0) return DAS3R_OK;
    dim3 grid(div_up(P, 256)), block(256);
    const bool has_sh = in->shs != nullptr, has_cov = in->cov3D_precomp != nullptr;
    const uint32_t *inv = binning ? (const uint32_t *)(binning + L.b_inv) : nullptr;
#define ARGS                                                                                                                 \
    P, a->sh_degree, a->M, in->means3D, in->scales, a->scale_modifier, in->rotations, in->shs, in->cov3D_precomp,            \
        a->viewmatrix, a->projmatrix, a->campos, a->image_width, a->image_height, a->tanfovx, a->tanfovy,                    \
        (const uint32_t *)(geom + L.pub.tiles_touched), (const uint8_t *)(geom + L.pub.clamped), partial, inv,               \
        (const uint32_t *)(geom + L.g_off_by_gid), g->dL_dmeans2D, g->dL_dopacities, g->dL_dcolors_precomp, g->dL_dmeans3D,  \
        g->dL_dscales, g->dL_drotations, g->dL_dshs, g->dL_dcov3D
    if (has_sh && !has_cov) DAS3R_LAUNCH((preprocess_backward_kernel<true, false>), grid, block, 0, s, ARGS);
    else if (has_sh && has_cov) DAS3R_LAUNCH((preprocess_backward_kernel<true, true>), grid, block, 0, s, ARGS);
    else if (!has_sh && !has_cov) DAS3R_LAUNCH((preprocess_backward_kernel<false, false>), grid, block, 0, s, ARGS);
    else DAS3R_LAUNCH((preprocess_backward_kernel<false, true>), grid, block, 0, s, ARGS);
#undef ARGS
    KERNEL_CHECK(s, a->debug, "preprocess_backward");
    return DAS3R_OK;
}

}  // namespace das3r
