/*
 * oracle/raster_oracle.c — CPU restatement of the differentiable Gaussian-splat tile rasterizer
 * that DAS3R calls through gaussian_renderer/__init__.py:131-140 (reference call site).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (das3r_amd/, diff_gaussian_rasterization/,
 * simple_knn/) may import, link or execute this file.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, as the checker / the reported CPU baseline.
 *
 * PARITY UNPINNED.  The algorithm lives in the un-vendored git submodule
 *   submodules/diff-gaussian-rasterization  (/root/reference/.gitmodules:4-6; directory empty, no pinned
 *   SHA recoverable; the 12-field settings / 2-tuple return at gaussian_renderer/__init__.py:62-78,131
 *   and the README.md:41-44 near-plane patch identify the pre-Oct-2024 graphdeco-inria line).
 * The reference holds no test, golden vector or fixture for this path, and the CUDA sources cannot be
 * built here.  This file therefore restates the PUBLISHED algorithm of that dependency
 * (upstream:cuda_rasterizer/{forward,backward,rasterizer_impl}.cu, auxiliary.h, config.h — behavioural
 * spec in SURVEY.md Appendix A) and is anchored by
 *   - the reference's own call sites (tensor layouts, conventions: SURVEY.md §8b),
 *   - golden vectors captured from the reference's importable helpers (tests/golden/ref_helpers.npz:
 *     utils/sh_utils.py:57-112 eval_sh pins the SH basis used in sh_to_rgb below;
 *     utils/graphics_utils.py:80-100 pins the projection-matrix layout),
 *   - an independent float64 PyTorch-autograd restatement (oracle/dense_oracle.py) and central finite
 *     differences (tests/test_oracle.py).
 *
 * All arithmetic fp32, same operation order as the upstream kernels where it matters for the discrete
 * decisions (radius ceil, tile rect, alpha<1/255, T<1e-4).  Build: see oracle/Makefile
 * (-ffp-contract=off so the CPU result does not depend on FMA availability).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define BLOCK_X 16 /* upstream:cuda_rasterizer/config.h */
#define BLOCK_Y 16
#define NEAR_Z 0.001f /* /root/reference/README.md:41-44 (upstream default 0.2f, auxiliary.h) */

/* SH constants: /root/reference/utils/sh_utils.py:26-43 == upstream:auxiliary.h SH_C0..SH_C3 */
static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                               -1.0925484305920792f, 0.5462742152960396f};
static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                               -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};

typedef struct {
    int P, D, M;        /* #gaussians, active SH degree, stored SH coeffs per channel ((max_deg+1)^2) */
    int W, H;
    float tanfovx, tanfovy;
    float scale_modifier;
    float bg[3];
    float viewmatrix[16]; /* row-vector convention: p_row @ V  (scene/cameras.py:90-93 stores transposes) */
    float projmatrix[16];
    float campos[3];
    int prefiltered;
} OracleArgs;

typedef struct {
    OracleArgs a;
    /* inputs are borrowed (caller keeps them alive between forward and backward) */
    const float *means3D, *shs, *colors_precomp, *opacities, *scales, *rotations, *cov3D_precomp;
    /* geometry state (upstream GeometryState) */
    float *depths, *xy, *conic_opacity, *rgb, *cov3D;
    uint8_t *clamped;
    int *radii;
    uint32_t *tiles_touched, *point_offsets;
    /* binning state */
    int64_t num_rendered;
    uint32_t *point_list;
    uint64_t *keys;
    uint32_t *ranges; /* 2 per tile */
    /* image state */
    float *final_T;
    uint32_t *n_contrib;
    int tiles_x, tiles_y;
} OracleState;

/* upstream:auxiliary.h transformPoint4x3 / transformPoint4x4 */
static inline void xf43(const float *p, const float *m, float *o) {
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
}
static inline void xf44(const float *p, const float *m, float *o) {
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
    o[3] = m[3] * p[0] + m[7] * p[1] + m[11] * p[2] + m[15];
}

/* upstream:forward.cu computeCov3D — Sigma = R diag((mod*s)^2) R^T, quaternion (r,x,y,z) used as given
 * (NOT normalised; same R formula as /root/reference/utils/general_utils.py:90-98 minus :79-81). */
static void quat_to_R(const float *q, float R[3][3]) {
    float r = q[0], x = q[1], y = q[2], z = q[3];
    R[0][0] = 1.f - 2.f * (y * y + z * z); R[0][1] = 2.f * (x * y - r * z);       R[0][2] = 2.f * (x * z + r * y);
    R[1][0] = 2.f * (x * y + r * z);       R[1][1] = 1.f - 2.f * (x * x + z * z); R[1][2] = 2.f * (y * z - r * x);
    R[2][0] = 2.f * (x * z - r * y);       R[2][1] = 2.f * (y * z + r * x);       R[2][2] = 1.f - 2.f * (x * x + y * y);
}
static void compute_cov3D(const float *scale, float mod, const float *q, float *cov) {
    float R[3][3];
    quat_to_R(q, R);
    float s[3] = {mod * scale[0], mod * scale[1], mod * scale[2]};
    float Mm[3][3]; /* M = R * diag(s) */
    for (int i = 0; i < 3; i++) for (int k = 0; k < 3; k++) Mm[i][k] = R[i][k] * s[k];
    float S[3][3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
        float acc = 0.f;
        for (int k = 0; k < 3; k++) acc += Mm[i][k] * Mm[j][k];
        S[i][j] = acc;
    }
    cov[0] = S[0][0]; cov[1] = S[0][1]; cov[2] = S[0][2]; cov[3] = S[1][1]; cov[4] = S[1][2]; cov[5] = S[2][2];
}

/* J (2x3) and T = J * Rcw (2x3), Rcw[i][j] = V[4j+i]  (upstream:forward.cu computeCov2D, glm column-major) */
static void cov2d_T(const float *mean, const OracleArgs *a, float focal_x, float focal_y, float T[2][3], float t[3],
                    int *clampx, int *clampy) {
    xf43(mean, a->viewmatrix, t);
    const float limx = 1.3f * a->tanfovx, limy = 1.3f * a->tanfovy;
    const float txtz = t[0] / t[2], tytz = t[1] / t[2];
    *clampx = (txtz < -limx || txtz > limx);
    *clampy = (tytz < -limy || tytz > limy);
    t[0] = fminf(limx, fmaxf(-limx, txtz)) * t[2];
    t[1] = fminf(limy, fmaxf(-limy, tytz)) * t[2];
    float J[2][3] = {{focal_x / t[2], 0.f, -(focal_x * t[0]) / (t[2] * t[2])},
                     {0.f, focal_y / t[2], -(focal_y * t[1]) / (t[2] * t[2])}};
    const float *V = a->viewmatrix;
    for (int i = 0; i < 2; i++) for (int j = 0; j < 3; j++)
        T[i][j] = J[i][0] * V[4 * j + 0] + J[i][1] * V[4 * j + 1] + J[i][2] * V[4 * j + 2];
}
static void compute_cov2D(const float *mean, const OracleArgs *a, float fx, float fy, const float *c3, float *cov) {
    float T[2][3], t[3];
    int cx, cy;
    cov2d_T(mean, a, fx, fy, T, t, &cx, &cy);
    float S[3][3] = {{c3[0], c3[1], c3[2]}, {c3[1], c3[3], c3[4]}, {c3[2], c3[4], c3[5]}};
    float TS[2][3];
    for (int i = 0; i < 2; i++) for (int j = 0; j < 3; j++) TS[i][j] = T[i][0] * S[0][j] + T[i][1] * S[1][j] + T[i][2] * S[2][j];
    cov[0] = TS[0][0] * T[0][0] + TS[0][1] * T[0][1] + TS[0][2] * T[0][2] + 0.3f;
    cov[1] = TS[0][0] * T[1][0] + TS[0][1] * T[1][1] + TS[0][2] * T[1][2];
    cov[2] = TS[1][0] * T[1][0] + TS[1][1] * T[1][1] + TS[1][2] * T[1][2] + 0.3f;
}

/* upstream:forward.cu computeColorFromSH; basis identical to /root/reference/utils/sh_utils.py:74-100
 * (pinned by tests/golden/ref_helpers.npz sh_eval_deg*).  sh layout (M,3): coefficient-major, then RGB
 * (/root/reference/scene/gaussian_model.py:186-190). */
static void sh_to_rgb(int deg, const float *sh, const float *pos, const float *campos, float *rgb, uint8_t *clamped) {
    float dir[3] = {pos[0] - campos[0], pos[1] - campos[1], pos[2] - campos[2]};
    float inv = 1.f / sqrtf(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
    float x = dir[0] * inv, y = dir[1] * inv, z = dir[2] * inv;
    for (int c = 0; c < 3; c++) {
#define SHC(k) sh[(k) * 3 + c]
        float r = SH_C0 * SHC(0);
        if (deg > 0) {
            r = r - SH_C1 * y * SHC(1) + SH_C1 * z * SHC(2) - SH_C1 * x * SHC(3);
            if (deg > 1) {
                float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                r = r + SH_C2[0] * xy * SHC(4) + SH_C2[1] * yz * SHC(5) + SH_C2[2] * (2.f * zz - xx - yy) * SHC(6) +
                    SH_C2[3] * xz * SHC(7) + SH_C2[4] * (xx - yy) * SHC(8);
                if (deg > 2) {
                    r = r + SH_C3[0] * y * (3.f * xx - yy) * SHC(9) + SH_C3[1] * xy * z * SHC(10) +
                        SH_C3[2] * y * (4.f * zz - xx - yy) * SHC(11) + SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy) * SHC(12) +
                        SH_C3[4] * x * (4.f * zz - xx - yy) * SHC(13) + SH_C3[5] * z * (xx - yy) * SHC(14) +
                        SH_C3[6] * x * (xx - 3.f * yy) * SHC(15);
                }
            }
        }
#undef SHC
        r += 0.5f;
        clamped[c] = (r < 0.f);
        rgb[c] = fmaxf(r, 0.f);
    }
}

/* stable LSD radix sort of (key,value) on bits [0,end_bit) — stands in for cub::DeviceRadixSort::SortPairs
 * (upstream:rasterizer_impl.cu forward): any stable sort yields the same (tile, depth, index) order. */
static void radix_sort_pairs(uint64_t *keys, uint32_t *vals, int64_t n, int end_bit) {
    if (n <= 1) return;
    uint64_t *k2 = (uint64_t *)malloc(sizeof(uint64_t) * n);
    uint32_t *v2 = (uint32_t *)malloc(sizeof(uint32_t) * n);
    uint64_t *ka = keys, *kb = k2;
    uint32_t *va = vals, *vb = v2;
    for (int shift = 0; shift < end_bit; shift += 11) {
        int64_t hist[2049];
        memset(hist, 0, sizeof(hist));
        for (int64_t i = 0; i < n; i++) hist[((ka[i] >> shift) & 2047) + 1]++;
        for (int d = 0; d < 2048; d++) hist[d + 1] += hist[d];
        for (int64_t i = 0; i < n; i++) {
            int64_t dst = hist[(ka[i] >> shift) & 2047]++;
            kb[dst] = ka[i];
            vb[dst] = va[i];
        }
        uint64_t *tk = ka; ka = kb; kb = tk;
        uint32_t *tv = va; va = vb; vb = tv;
    }
    if (ka != keys) {
        memcpy(keys, ka, sizeof(uint64_t) * n);
        memcpy(vals, va, sizeof(uint32_t) * n);
    }
    free(k2);
    free(v2);
}

void oracle_raster_free(OracleState *s) {
    if (!s) return;
    free(s->depths); free(s->xy); free(s->conic_opacity); free(s->rgb); free(s->cov3D); free(s->clamped);
    free(s->radii); free(s->tiles_touched); free(s->point_offsets); free(s->point_list); free(s->keys);
    free(s->ranges); free(s->final_T); free(s->n_contrib);
    free(s);
}

/*
 * Forward.  upstream:rasterizer_impl.cu CudaRasterizer::Rasterizer::forward:
 *   preprocess -> inclusive scan of tiles_touched -> duplicateWithKeys -> sort -> identifyTileRanges -> render.
 * Returns an opaque state (saved for backward / inspection) or NULL on bad arguments.
 * out_color: [3,H,W] planar; out_radii: [P].
 */
OracleState *oracle_raster_forward(const OracleArgs *args, const float *means3D, const float *shs,
                                   const float *colors_precomp, const float *opacities, const float *scales,
                                   const float *rotations, const float *cov3D_precomp, float *out_color, int *out_radii) {
    OracleState *s = (OracleState *)calloc(1, sizeof(OracleState));
    s->a = *args;
    const OracleArgs *a = &s->a;
    const int P = a->P, W = a->W, H = a->H;
    s->means3D = means3D; s->shs = shs; s->colors_precomp = colors_precomp; s->opacities = opacities;
    s->scales = scales; s->rotations = rotations; s->cov3D_precomp = cov3D_precomp;
    const int tx = (W + BLOCK_X - 1) / BLOCK_X, ty = (H + BLOCK_Y - 1) / BLOCK_Y;
    s->tiles_x = tx; s->tiles_y = ty;
    const size_t Pn = P > 0 ? (size_t)P : 1;
    s->depths = (float *)calloc(Pn, sizeof(float));
    s->xy = (float *)calloc(Pn * 2, sizeof(float));
    s->conic_opacity = (float *)calloc(Pn * 4, sizeof(float));
    s->rgb = (float *)calloc(Pn * 3, sizeof(float));
    s->cov3D = (float *)calloc(Pn * 6, sizeof(float));
    s->clamped = (uint8_t *)calloc(Pn * 3, 1);
    s->radii = (int *)calloc(Pn, sizeof(int));
    s->tiles_touched = (uint32_t *)calloc(Pn, sizeof(uint32_t));
    s->point_offsets = (uint32_t *)calloc(Pn, sizeof(uint32_t));
    s->ranges = (uint32_t *)calloc((size_t)tx * ty * 2, sizeof(uint32_t));
    s->final_T = (float *)calloc((size_t)W * H, sizeof(float));
    s->n_contrib = (uint32_t *)calloc((size_t)W * H, sizeof(uint32_t));
    const float focal_x = W / (2.0f * a->tanfovx), focal_y = H / (2.0f * a->tanfovy);

    /* ---- K1 preprocess (upstream:forward.cu preprocessCUDA) ---- */
#pragma omp parallel for schedule(static)
    for (int idx = 0; idx < P; idx++) {
        s->radii[idx] = 0;
        s->tiles_touched[idx] = 0;
        const float *p = means3D + 3 * idx;
        float p_view[3];
        xf43(p, a->viewmatrix, p_view);
        if (p_view[2] <= NEAR_Z) continue; /* in_frustum (auxiliary.h) with the DAS3R threshold */
        float p_hom[4];
        xf44(p, a->projmatrix, p_hom);
        float p_w = 1.0f / (p_hom[3] + 0.0000001f);
        float p_proj[2] = {p_hom[0] * p_w, p_hom[1] * p_w};
        const float *c3;
        if (cov3D_precomp) {
            c3 = cov3D_precomp + 6 * idx;
        } else {
            compute_cov3D(scales + 3 * idx, a->scale_modifier, rotations + 4 * idx, s->cov3D + 6 * idx);
            c3 = s->cov3D + 6 * idx;
        }
        float cov[3];
        compute_cov2D(p, a, focal_x, focal_y, c3, cov);
        float det = cov[0] * cov[2] - cov[1] * cov[1];
        if (det == 0.0f) continue;
        float det_inv = 1.f / det;
        float conic[3] = {cov[2] * det_inv, -cov[1] * det_inv, cov[0] * det_inv};
        float mid = 0.5f * (cov[0] + cov[2]);
        float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
        float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
        float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
        float px = ((p_proj[0] + 1.0f) * W - 1.0f) * 0.5f; /* ndc2Pix */
        float py = ((p_proj[1] + 1.0f) * H - 1.0f) * 0.5f;
        int max_radius = (int)my_radius;
        /* getRect (auxiliary.h): C truncation toward zero, clamp to grid */
        int rminx = (int)((px - max_radius) / BLOCK_X), rminy = (int)((py - max_radius) / BLOCK_Y);
        int rmaxx = (int)((px + max_radius + BLOCK_X - 1) / BLOCK_X), rmaxy = (int)((py + max_radius + BLOCK_Y - 1) / BLOCK_Y);
        rminx = rminx < 0 ? 0 : (rminx > tx ? tx : rminx);
        rminy = rminy < 0 ? 0 : (rminy > ty ? ty : rminy);
        rmaxx = rmaxx < 0 ? 0 : (rmaxx > tx ? tx : rmaxx);
        rmaxy = rmaxy < 0 ? 0 : (rmaxy > ty ? ty : rmaxy);
        if ((rmaxx - rminx) * (rmaxy - rminy) == 0) continue;
        if (!colors_precomp) sh_to_rgb(a->D, shs + (size_t)idx * a->M * 3, p, a->campos, s->rgb + 3 * idx, s->clamped + 3 * idx);
        s->depths[idx] = p_view[2];
        s->radii[idx] = max_radius;
        s->xy[2 * idx] = px;
        s->xy[2 * idx + 1] = py;
        s->conic_opacity[4 * idx + 0] = conic[0];
        s->conic_opacity[4 * idx + 1] = conic[1];
        s->conic_opacity[4 * idx + 2] = conic[2];
        s->conic_opacity[4 * idx + 3] = opacities[idx];
        s->tiles_touched[idx] = (uint32_t)((rmaxy - rminy) * (rmaxx - rminx));
    }

    /* ---- K2 inclusive scan ---- */
    uint64_t run = 0;
    for (int i = 0; i < P; i++) {
        run += s->tiles_touched[i];
        s->point_offsets[i] = (uint32_t)run;
    }
    s->num_rendered = (int64_t)run;
    const int64_t I = s->num_rendered;
    s->keys = (uint64_t *)malloc(sizeof(uint64_t) * (I > 0 ? I : 1));
    s->point_list = (uint32_t *)malloc(sizeof(uint32_t) * (I > 0 ? I : 1));

    /* ---- K3 duplicateWithKeys: key = tile_id << 32 | float_bits(depth), value = gaussian index ---- */
#pragma omp parallel for schedule(static)
    for (int idx = 0; idx < P; idx++) {
        if (s->radii[idx] <= 0) continue;
        uint32_t off = idx == 0 ? 0 : s->point_offsets[idx - 1];
        float px = s->xy[2 * idx], py = s->xy[2 * idx + 1];
        int r = s->radii[idx];
        int rminx = (int)((px - r) / BLOCK_X), rminy = (int)((py - r) / BLOCK_Y);
        int rmaxx = (int)((px + r + BLOCK_X - 1) / BLOCK_X), rmaxy = (int)((py + r + BLOCK_Y - 1) / BLOCK_Y);
        rminx = rminx < 0 ? 0 : (rminx > tx ? tx : rminx);
        rminy = rminy < 0 ? 0 : (rminy > ty ? ty : rminy);
        rmaxx = rmaxx < 0 ? 0 : (rmaxx > tx ? tx : rmaxx);
        rmaxy = rmaxy < 0 ? 0 : (rmaxy > ty ? ty : rmaxy);
        uint32_t dbits;
        memcpy(&dbits, &s->depths[idx], 4);
        for (int y = rminy; y < rmaxy; y++)
            for (int x = rminx; x < rmaxx; x++) {
                uint64_t key = (uint64_t)(y * tx + x);
                key = (key << 32) | dbits;
                s->keys[off] = key;
                s->point_list[off] = (uint32_t)idx;
                off++;
            }
    }

    /* ---- K4 sort, K5 identifyTileRanges ---- */
    int bit = 0;
    {
        uint32_t n = (uint32_t)(tx * ty);
        while (n > 0) { bit++; n >>= 1; } /* getHigherMsb-equivalent upper bound */
    }
    radix_sort_pairs(s->keys, s->point_list, I, 32 + bit);
    for (int64_t i = 0; i < I; i++) {
        uint32_t tile = (uint32_t)(s->keys[i] >> 32);
        if (i == 0) s->ranges[2 * tile] = 0;
        else {
            uint32_t prev = (uint32_t)(s->keys[i - 1] >> 32);
            if (tile != prev) {
                s->ranges[2 * prev + 1] = (uint32_t)i;
                s->ranges[2 * tile] = (uint32_t)i;
            }
        }
        if (i == I - 1) s->ranges[2 * tile + 1] = (uint32_t)I;
    }

    /* ---- K6 render (upstream:forward.cu renderCUDA), one pixel at a time ---- */
    const float *feat = colors_precomp ? colors_precomp : s->rgb;
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
    for (int by = 0; by < ty; by++)
        for (int bx = 0; bx < tx; bx++) {
            const uint32_t r0 = s->ranges[2 * (by * tx + bx)], r1 = s->ranges[2 * (by * tx + bx) + 1];
            for (int ly = 0; ly < BLOCK_Y; ly++)
                for (int lx = 0; lx < BLOCK_X; lx++) {
                    int pxi = bx * BLOCK_X + lx, pyi = by * BLOCK_Y + ly;
                    if (pxi >= W || pyi >= H) continue;
                    float pixf[2] = {(float)pxi, (float)pyi};
                    float T = 1.0f, C[3] = {0, 0, 0};
                    uint32_t contributor = 0, last_contributor = 0;
                    for (uint32_t k = r0; k < r1; k++) {
                        contributor++;
                        uint32_t g = s->point_list[k];
                        float dx = s->xy[2 * g] - pixf[0], dy = s->xy[2 * g + 1] - pixf[1];
                        const float *co = s->conic_opacity + 4 * g;
                        float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                        if (power > 0.0f) continue;
                        float alpha = fminf(0.99f, co[3] * expf(power));
                        if (alpha < 1.0f / 255.0f) continue;
                        float test_T = T * (1 - alpha);
                        if (test_T < 0.0001f) break; /* done: this splat is NOT blended */
                        for (int ch = 0; ch < 3; ch++) C[ch] += feat[3 * g + ch] * alpha * T;
                        T = test_T;
                        last_contributor = contributor;
                    }
                    size_t pix = (size_t)pyi * W + pxi;
                    s->final_T[pix] = T;
                    s->n_contrib[pix] = last_contributor;
                    for (int ch = 0; ch < 3; ch++) out_color[(size_t)ch * H * W + pix] = C[ch] + T * a->bg[ch];
                }
        }
    for (int i = 0; i < P; i++) out_radii[i] = s->radii[i];
    return s;
}

static inline void atomic_addf(float *p, float v) {
#pragma omp atomic
    *p += v;
}

/*
 * Backward.  upstream:rasterizer_impl.cu Rasterizer::backward: BACKWARD::render (K7) then
 * BACKWARD::preprocess = computeCov2DCUDA (K8) + preprocessCUDA (K9).  All grad buffers are
 * (re)zeroed here, like the torch::zeros in upstream:rasterize_points.cu RasterizeGaussiansBackwardCUDA.
 *   dL_dmean2D [P,3] (z = 0; x,y already multiplied by W/2, H/2 — SURVEY.md A.7)
 *   dL_dconic [P,4] (x,y,_,w)   dL_dopacity [P]   dL_dcolor [P,3]   dL_dcov3D [P,6]
 *   dL_dmean3D [P,3]   dL_dsh [P,M,3]   dL_dscale [P,3]   dL_drot [P,4]
 */
void oracle_raster_backward(OracleState *s, const float *dL_dpix, float *dL_dmean2D, float *dL_dconic, float *dL_dopacity,
                            float *dL_dcolor, float *dL_dmean3D, float *dL_dcov3D, float *dL_dsh, float *dL_dscale,
                            float *dL_drot) {
    const OracleArgs *a = &s->a;
    const int P = a->P, W = a->W, H = a->H, tx = s->tiles_x, ty = s->tiles_y;
    memset(dL_dmean2D, 0, sizeof(float) * 3 * P);
    memset(dL_dconic, 0, sizeof(float) * 4 * P);
    memset(dL_dopacity, 0, sizeof(float) * P);
    memset(dL_dcolor, 0, sizeof(float) * 3 * P);
    memset(dL_dmean3D, 0, sizeof(float) * 3 * P);
    memset(dL_dcov3D, 0, sizeof(float) * 6 * P);
    memset(dL_dsh, 0, sizeof(float) * 3 * (size_t)a->M * P);
    memset(dL_dscale, 0, sizeof(float) * 3 * P);
    memset(dL_drot, 0, sizeof(float) * 4 * P);
    const float *feat = s->colors_precomp ? s->colors_precomp : s->rgb;
    const float ddelx_dx = 0.5f * W, ddely_dy = 0.5f * H;

    /* ---- K7 (upstream:backward.cu renderCUDA): back-to-front replay per pixel ---- */
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
    for (int by = 0; by < ty; by++)
        for (int bx = 0; bx < tx; bx++) {
            const uint32_t r0 = s->ranges[2 * (by * tx + bx)], r1 = s->ranges[2 * (by * tx + bx) + 1];
            for (int ly = 0; ly < BLOCK_Y; ly++)
                for (int lx = 0; lx < BLOCK_X; lx++) {
                    int pxi = bx * BLOCK_X + lx, pyi = by * BLOCK_Y + ly;
                    if (pxi >= W || pyi >= H) continue;
                    size_t pix = (size_t)pyi * W + pxi;
                    float pixf[2] = {(float)pxi, (float)pyi};
                    const float T_final = s->final_T[pix];
                    float T = T_final;
                    uint32_t contributor = r1 - r0;
                    const uint32_t last_contributor = s->n_contrib[pix];
                    float accum_rec[3] = {0, 0, 0}, dL_dpixel[3], last_color[3] = {0, 0, 0};
                    for (int ch = 0; ch < 3; ch++) dL_dpixel[ch] = dL_dpix[(size_t)ch * H * W + pix];
                    float last_alpha = 0.f;
                    for (int64_t k = (int64_t)r1 - 1; k >= (int64_t)r0; k--) {
                        contributor--;
                        if (contributor >= last_contributor) continue;
                        uint32_t g = s->point_list[k];
                        float dx = s->xy[2 * g] - pixf[0], dy = s->xy[2 * g + 1] - pixf[1];
                        const float *co = s->conic_opacity + 4 * g;
                        float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                        if (power > 0.0f) continue;
                        float G = expf(power);
                        float alpha = fminf(0.99f, co[3] * G);
                        if (alpha < 1.0f / 255.0f) continue;
                        T = T / (1.f - alpha);
                        float dchannel_dcolor = alpha * T;
                        float dL_dalpha = 0.0f;
                        for (int ch = 0; ch < 3; ch++) {
                            float c = feat[3 * g + ch];
                            accum_rec[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
                            last_color[ch] = c;
                            float dL_dchannel = dL_dpixel[ch];
                            dL_dalpha += (c - accum_rec[ch]) * dL_dchannel;
                            atomic_addf(&dL_dcolor[3 * g + ch], dchannel_dcolor * dL_dchannel);
                        }
                        dL_dalpha *= T;
                        last_alpha = alpha;
                        float bg_dot_dpixel = 0;
                        for (int i = 0; i < 3; i++) bg_dot_dpixel += a->bg[i] * dL_dpixel[i];
                        dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot_dpixel;
                        float dL_dG = co[3] * dL_dalpha;
                        float gdx = G * dx, gdy = G * dy;
                        float dG_ddelx = -gdx * co[0] - gdy * co[1];
                        float dG_ddely = -gdy * co[2] - gdx * co[1];
                        atomic_addf(&dL_dmean2D[3 * g + 0], dL_dG * dG_ddelx * ddelx_dx);
                        atomic_addf(&dL_dmean2D[3 * g + 1], dL_dG * dG_ddely * ddely_dy);
                        atomic_addf(&dL_dconic[4 * g + 0], -0.5f * gdx * dx * dL_dG);
                        atomic_addf(&dL_dconic[4 * g + 1], -0.5f * gdx * dy * dL_dG);
                        atomic_addf(&dL_dconic[4 * g + 3], -0.5f * gdy * dy * dL_dG);
                        atomic_addf(&dL_dopacity[g], G * dL_dalpha);
                    }
                }
        }

    const float focal_x = W / (2.0f * a->tanfovx), focal_y = H / (2.0f * a->tanfovy);
    const float *V = a->viewmatrix, *PM = a->projmatrix;
#pragma omp parallel for schedule(static)
    for (int idx = 0; idx < P; idx++) {
        if (!(s->radii[idx] > 0)) continue;
        const float *mean = s->means3D + 3 * idx;
        float dmean[3] = {0, 0, 0};
        /* ---- K8 (upstream:backward.cu computeCov2DCUDA) ---- */
        {
            const float *c3 = s->cov3D_precomp ? s->cov3D_precomp + 6 * idx : s->cov3D + 6 * idx;
            float T[2][3], t[3];
            int cx, cy;
            cov2d_T(mean, a, focal_x, focal_y, T, t, &cx, &cy);
            const float x_grad_mul = cx ? 0.f : 1.f, y_grad_mul = cy ? 0.f : 1.f;
            float S[3][3] = {{c3[0], c3[1], c3[2]}, {c3[1], c3[3], c3[4]}, {c3[2], c3[4], c3[5]}};
            float TS0[3], TS1[3];
            for (int j = 0; j < 3; j++) {
                TS0[j] = T[0][0] * S[j][0] + T[0][1] * S[j][1] + T[0][2] * S[j][2];
                TS1[j] = T[1][0] * S[j][0] + T[1][1] * S[j][1] + T[1][2] * S[j][2];
            }
            float ca = TS0[0] * T[0][0] + TS0[1] * T[0][1] + TS0[2] * T[0][2] + 0.3f;
            float cb = TS0[0] * T[1][0] + TS0[1] * T[1][1] + TS0[2] * T[1][2];
            float cc = TS1[0] * T[1][0] + TS1[1] * T[1][1] + TS1[2] * T[1][2] + 0.3f;
            float gA = dL_dconic[4 * idx], gB = dL_dconic[4 * idx + 1], gC = dL_dconic[4 * idx + 3];
            float denom = ca * cc - cb * cb;
            float dL_da = 0, dL_db = 0, dL_dc = 0;
            float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
            if (denom2inv != 0) {
                dL_da = denom2inv * (-cc * cc * gA + 2 * cb * cc * gB + (denom - ca * cc) * gC);
                dL_dc = denom2inv * (-ca * ca * gC + 2 * ca * cb * gB + (denom - ca * cc) * gA);
                dL_db = denom2inv * 2 * (cb * cc * gA - (denom + 2 * cb * cb) * gB + ca * cb * gC);
                float *o = dL_dcov3D + 6 * idx;
                o[0] = (T[0][0] * T[0][0] * dL_da + T[0][0] * T[1][0] * dL_db + T[1][0] * T[1][0] * dL_dc);
                o[3] = (T[0][1] * T[0][1] * dL_da + T[0][1] * T[1][1] * dL_db + T[1][1] * T[1][1] * dL_dc);
                o[5] = (T[0][2] * T[0][2] * dL_da + T[0][2] * T[1][2] * dL_db + T[1][2] * T[1][2] * dL_dc);
                o[1] = 2 * T[0][0] * T[0][1] * dL_da + (T[0][0] * T[1][1] + T[0][1] * T[1][0]) * dL_db + 2 * T[1][0] * T[1][1] * dL_dc;
                o[2] = 2 * T[0][0] * T[0][2] * dL_da + (T[0][0] * T[1][2] + T[0][2] * T[1][0]) * dL_db + 2 * T[1][0] * T[1][2] * dL_dc;
                o[4] = 2 * T[0][2] * T[0][1] * dL_da + (T[0][1] * T[1][2] + T[0][2] * T[1][1]) * dL_db + 2 * T[1][1] * T[1][2] * dL_dc;
            }
            /* dL/dT = 2 * (T Sigma) * dL_d{a,c} + (other row of T Sigma) * dL_db */
            float dT0[3], dT1[3];
            for (int j = 0; j < 3; j++) {
                dT0[j] = 2 * TS0[j] * dL_da + TS1[j] * dL_db;
                dT1[j] = 2 * TS1[j] * dL_dc + TS0[j] * dL_db;
            }
            /* dL/dJ = dL/dT * Rcw^T, Rcw[i][j] = V[4j+i] */
            float dJ00 = V[0] * dT0[0] + V[4] * dT0[1] + V[8] * dT0[2];
            float dJ02 = V[2] * dT0[0] + V[6] * dT0[1] + V[10] * dT0[2];
            float dJ11 = V[1] * dT1[0] + V[5] * dT1[1] + V[9] * dT1[2];
            float dJ12 = V[2] * dT1[0] + V[6] * dT1[1] + V[10] * dT1[2];
            float tz = 1.f / t[2], tz2 = tz * tz, tz3 = tz2 * tz;
            float dtx = x_grad_mul * -focal_x * tz2 * dJ02;
            float dty = y_grad_mul * -focal_y * tz2 * dJ12;
            float dtz = -focal_x * tz2 * dJ00 - focal_y * tz2 * dJ11 + (2 * focal_x * t[0]) * tz3 * dJ02 + (2 * focal_y * t[1]) * tz3 * dJ12;
            /* transformVec4x3Transpose */
            dmean[0] = V[0] * dtx + V[1] * dty + V[2] * dtz;
            dmean[1] = V[4] * dtx + V[5] * dty + V[6] * dtz;
            dmean[2] = V[8] * dtx + V[9] * dty + V[10] * dtz;
        }
        /* ---- K9 (upstream:backward.cu preprocessCUDA): projection part ---- */
        {
            float m_hom[4];
            xf44(mean, PM, m_hom);
            float m_w = 1.0f / (m_hom[3] + 0.0000001f);
            float mul1 = (PM[0] * mean[0] + PM[4] * mean[1] + PM[8] * mean[2] + PM[12]) * m_w * m_w;
            float mul2 = (PM[1] * mean[0] + PM[5] * mean[1] + PM[9] * mean[2] + PM[13]) * m_w * m_w;
            float gx = dL_dmean2D[3 * idx], gy = dL_dmean2D[3 * idx + 1];
            dmean[0] += (PM[0] * m_w - PM[3] * mul1) * gx + (PM[1] * m_w - PM[3] * mul2) * gy;
            dmean[1] += (PM[4] * m_w - PM[7] * mul1) * gx + (PM[5] * m_w - PM[7] * mul2) * gy;
            dmean[2] += (PM[8] * m_w - PM[11] * mul1) * gx + (PM[9] * m_w - PM[11] * mul2) * gy;
        }
        /* ---- SH backward (upstream:backward.cu computeColorFromSH) ---- */
        if (s->shs) {
            const int deg = a->D;
            const float *sh = s->shs + (size_t)idx * a->M * 3;
            float *dsh = dL_dsh + (size_t)idx * a->M * 3;
            float dir_orig[3] = {mean[0] - a->campos[0], mean[1] - a->campos[1], mean[2] - a->campos[2]};
            float inv = 1.f / sqrtf(dir_orig[0] * dir_orig[0] + dir_orig[1] * dir_orig[1] + dir_orig[2] * dir_orig[2]);
            float x = dir_orig[0] * inv, y = dir_orig[1] * inv, z = dir_orig[2] * inv;
            float g[3];
            for (int c = 0; c < 3; c++) g[c] = s->clamped[3 * idx + c] ? 0.f : dL_dcolor[3 * idx + c];
            float ddir[3] = {0, 0, 0};
            for (int c = 0; c < 3; c++) {
#define SHC(k) sh[(k) * 3 + c]
#define DSH(k) dsh[(k) * 3 + c]
                float dRdx = 0, dRdy = 0, dRdz = 0;
                DSH(0) = SH_C0 * g[c];
                if (deg > 0) {
                    DSH(1) = -SH_C1 * y * g[c];
                    DSH(2) = SH_C1 * z * g[c];
                    DSH(3) = -SH_C1 * x * g[c];
                    dRdx = -SH_C1 * SHC(3);
                    dRdy = -SH_C1 * SHC(1);
                    dRdz = SH_C1 * SHC(2);
                    if (deg > 1) {
                        float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                        DSH(4) = SH_C2[0] * xy * g[c];
                        DSH(5) = SH_C2[1] * yz * g[c];
                        DSH(6) = SH_C2[2] * (2.f * zz - xx - yy) * g[c];
                        DSH(7) = SH_C2[3] * xz * g[c];
                        DSH(8) = SH_C2[4] * (xx - yy) * g[c];
                        dRdx += SH_C2[0] * y * SHC(4) + SH_C2[2] * 2.f * -x * SHC(6) + SH_C2[3] * z * SHC(7) + SH_C2[4] * 2.f * x * SHC(8);
                        dRdy += SH_C2[0] * x * SHC(4) + SH_C2[1] * z * SHC(5) + SH_C2[2] * 2.f * -y * SHC(6) + SH_C2[4] * 2.f * -y * SHC(8);
                        dRdz += SH_C2[1] * y * SHC(5) + SH_C2[2] * 2.f * 2.f * z * SHC(6) + SH_C2[3] * x * SHC(7);
                        if (deg > 2) {
                            DSH(9) = SH_C3[0] * y * (3.f * xx - yy) * g[c];
                            DSH(10) = SH_C3[1] * xy * z * g[c];
                            DSH(11) = SH_C3[2] * y * (4.f * zz - xx - yy) * g[c];
                            DSH(12) = SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy) * g[c];
                            DSH(13) = SH_C3[4] * x * (4.f * zz - xx - yy) * g[c];
                            DSH(14) = SH_C3[5] * z * (xx - yy) * g[c];
                            DSH(15) = SH_C3[6] * x * (xx - 3.f * yy) * g[c];
                            dRdx += SH_C3[0] * SHC(9) * 3.f * 2.f * xy + SH_C3[1] * SHC(10) * yz + SH_C3[2] * SHC(11) * -2.f * xy +
                                    SH_C3[3] * SHC(12) * -3.f * 2.f * xz + SH_C3[4] * SHC(13) * (-3.f * xx + 4.f * zz - yy) +
                                    SH_C3[5] * SHC(14) * 2.f * xz + SH_C3[6] * SHC(15) * 3.f * (xx - yy);
                            dRdy += SH_C3[0] * SHC(9) * 3.f * (xx - yy) + SH_C3[1] * SHC(10) * xz + SH_C3[2] * SHC(11) * (-3.f * yy + 4.f * zz - xx) +
                                    SH_C3[3] * SHC(12) * -3.f * 2.f * yz + SH_C3[4] * SHC(13) * -2.f * xy + SH_C3[5] * SHC(14) * -2.f * yz +
                                    SH_C3[6] * SHC(15) * -3.f * 2.f * xy;
                            dRdz += SH_C3[1] * SHC(10) * xy + SH_C3[2] * SHC(11) * 4.f * 2.f * yz + SH_C3[3] * SHC(12) * 3.f * (2.f * zz - xx - yy) +
                                    SH_C3[4] * SHC(13) * 4.f * 2.f * xz + SH_C3[5] * SHC(14) * (xx - yy);
                        }
                    }
                }
#undef SHC
#undef DSH
                ddir[0] += dRdx * g[c];
                ddir[1] += dRdy * g[c];
                ddir[2] += dRdz * g[c];
            }
            /* dnormvdv (auxiliary.h) */
            float sum2 = dir_orig[0] * dir_orig[0] + dir_orig[1] * dir_orig[1] + dir_orig[2] * dir_orig[2];
            float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
            const float *v = dir_orig;
            dmean[0] += ((+sum2 - v[0] * v[0]) * ddir[0] - v[1] * v[0] * ddir[1] - v[2] * v[0] * ddir[2]) * invsum32;
            dmean[1] += (-v[0] * v[1] * ddir[0] + (sum2 - v[1] * v[1]) * ddir[1] - v[2] * v[1] * ddir[2]) * invsum32;
            dmean[2] += (-v[0] * v[2] * ddir[0] - v[1] * v[2] * ddir[1] + (sum2 - v[2] * v[2]) * ddir[2]) * invsum32;
        }
        dL_dmean3D[3 * idx] = dmean[0];
        dL_dmean3D[3 * idx + 1] = dmean[1];
        dL_dmean3D[3 * idx + 2] = dmean[2];
        /* ---- cov3D backward (upstream:backward.cu computeCov3D): Sigma = M M^T, M = R diag(mod*s) ---- */
        if (s->scales) {
            const float *q = s->rotations + 4 * idx, *sc = s->scales + 3 * idx;
            float R[3][3];
            quat_to_R(q, R);
            float sv[3] = {a->scale_modifier * sc[0], a->scale_modifier * sc[1], a->scale_modifier * sc[2]};
            const float *gc = dL_dcov3D + 6 * idx;
            float Gs[3][3] = {{gc[0], 0.5f * gc[1], 0.5f * gc[2]}, {0.5f * gc[1], gc[3], 0.5f * gc[4]}, {0.5f * gc[2], 0.5f * gc[4], gc[5]}};
            float Mm[3][3], dM[3][3];
            for (int i = 0; i < 3; i++) for (int k = 0; k < 3; k++) Mm[i][k] = R[i][k] * sv[k];
            for (int i = 0; i < 3; i++) for (int k = 0; k < 3; k++)
                dM[i][k] = 2.f * (Gs[i][0] * Mm[0][k] + Gs[i][1] * Mm[1][k] + Gs[i][2] * Mm[2][k]);
            /* NB upstream omits the scale_modifier factor in dL/dscale */
            for (int k = 0; k < 3; k++) dL_dscale[3 * idx + k] = R[0][k] * dM[0][k] + R[1][k] * dM[1][k] + R[2][k] * dM[2][k];
            float dR[3][3];
            for (int i = 0; i < 3; i++) for (int k = 0; k < 3; k++) dR[i][k] = dM[i][k] * sv[k];
            float r = q[0], x = q[1], y = q[2], z = q[3];
            /* derivative of the UNNORMALISED quaternion->R map */
            dL_drot[4 * idx + 0] = 2 * z * (dR[1][0] - dR[0][1]) + 2 * y * (dR[0][2] - dR[2][0]) + 2 * x * (dR[2][1] - dR[1][2]);
            dL_drot[4 * idx + 1] = 2 * y * (dR[0][1] + dR[1][0]) + 2 * z * (dR[0][2] + dR[2][0]) + 2 * r * (dR[2][1] - dR[1][2]) - 4 * x * (dR[2][2] + dR[1][1]);
            dL_drot[4 * idx + 2] = 2 * x * (dR[0][1] + dR[1][0]) + 2 * r * (dR[0][2] - dR[2][0]) + 2 * z * (dR[2][1] + dR[1][2]) - 4 * y * (dR[2][2] + dR[0][0]);
            dL_drot[4 * idx + 3] = 2 * r * (dR[1][0] - dR[0][1]) + 2 * x * (dR[0][2] + dR[2][0]) + 2 * y * (dR[2][1] + dR[1][2]) - 4 * z * (dR[1][1] + dR[0][0]);
        }
    }
}

/* upstream:forward.cu / rasterizer_impl.cu markVisible -> checkFrustum */
void oracle_mark_visible(int P, const float *means3D, const float *viewmatrix, const float *projmatrix, uint8_t *present) {
    (void)projmatrix;
    for (int i = 0; i < P; i++) {
        float pv[3];
        xf43(means3D + 3 * i, viewmatrix, pv);
        present[i] = pv[2] > NEAR_Z;
    }
}

/* accessors for stage-by-stage parity checks */
int64_t oracle_num_rendered(const OracleState *s) { return s->num_rendered; }
const float *oracle_depths(const OracleState *s) { return s->depths; }
const float *oracle_xy(const OracleState *s) { return s->xy; }
const float *oracle_conic_opacity(const OracleState *s) { return s->conic_opacity; }
const float *oracle_rgb(const OracleState *s) { return s->rgb; }
const float *oracle_cov3D(const OracleState *s) { return s->cov3D; }
const uint8_t *oracle_clamped(const OracleState *s) { return s->clamped; }
const uint32_t *oracle_tiles_touched(const OracleState *s) { return s->tiles_touched; }
const uint32_t *oracle_point_list(const OracleState *s) { return s->point_list; }
const uint32_t *oracle_ranges(const OracleState *s) { return s->ranges; }
const float *oracle_final_T(const OracleState *s) { return s->final_T; }
const uint32_t *oracle_n_contrib(const OracleState *s) { return s->n_contrib; }
int oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
