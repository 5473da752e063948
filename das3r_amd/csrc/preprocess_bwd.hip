// preprocess_bwd.hip — K8+K9 fused: per-Gaussian chain rule from the screen-space gradients (dL/dmean2D, dL/dconic,
// dL/dcolour, dL/dopacity) to (dL/dmean3D, dL/dscale, dL/drotation, dL/dSH) or dL/dcov3D.  Replaces
// upstream:cuda_rasterizer/backward.cu computeCov2DCUDA + preprocessCUDA (+ computeColorFromSH / computeCov3D backward)
// — SURVEY.md A.8.
//
// HBM-bound streaming map, lane per splat:
//   * the screen-space gradients are GATHERED: the render backward left one row of 9 sums per (tile, splat) instance in
//     partial[]; a splat's instances own the consecutive rows off_by_gid[g] .. + tiles_touched[g] (no atomics, no memsets);
//   * every output row is written in full (zeros for culled splats and for SH coefficients above the active degree), so
//     no output needs pre-zeroing and the dense (P,M,3) dL_dsh tensor is touched exactly once;
//   * M == 16: the workgroup's 256 dL_dsh rows (192 B each, contiguous) are assembled in LDS and stored with fully
//     coalesced float4 stores; at degree >= 2 the SH input rows come in the same way (see preprocess.hip).  The SH
//     backward is streamed (basis values -> LDS row) instead of holding two 48-float arrays in registers.
#include <string.h>

#include <algorithm>
#include <vector>

#include "common.h"
#include "splat_math.h"
#include "pretransform_math.h"
#include "pretransform_chain.h"

namespace das3r {

// basis functions of the real SH up to degree 3 evaluated at unit direction (x,y,z): rgb_c = sum_k b[k] * sh[k][c]
__device__ __forceinline__ void sh_basis(const int D, const float x, const float y, const float z, float b[16]) {
    b[0] = SH_C0;
    if (D > 0) {
        b[1] = -SH_C1 * y;
        b[2] = SH_C1 * z;
        b[3] = -SH_C1 * x;
        if (D > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            b[4] = SH_C2_0 * xy;
            b[5] = SH_C2_1 * yz;
            b[6] = SH_C2_2 * (2.f * zz - xx - yy);
            b[7] = SH_C2_3 * xz;
            b[8] = SH_C2_4 * (xx - yy);
            if (D > 2) {
                b[9] = SH_C3_0 * y * (3.f * xx - yy);
                b[10] = SH_C3_1 * xy * z;
                b[11] = SH_C3_2 * y * (4.f * zz - xx - yy);
                b[12] = SH_C3_3 * z * (2.f * zz - 3.f * xx - 3.f * yy);
                b[13] = SH_C3_4 * x * (4.f * zz - xx - yy);
                b[14] = SH_C3_5 * z * (xx - yy);
                b[15] = SH_C3_6 * x * (xx - 3.f * yy);
            }
        }
    }
}

// d(rgb_c)/d(dir) for one channel; sh = that channel's coefficients accessed as SHC(k)
#define SH_DDIR(SHC, dRdx, dRdy, dRdz)                                                                                          \
    do {                                                                                                                        \
        dRdx = 0.f; dRdy = 0.f; dRdz = 0.f;                                                                                     \
        if (D > 0) {                                                                                                            \
            dRdx = -SH_C1 * SHC(3);                                                                                             \
            dRdy = -SH_C1 * SHC(1);                                                                                             \
            dRdz = SH_C1 * SHC(2);                                                                                              \
            if (D > 1) {                                                                                                        \
                dRdx += SH_C2_0 * y * SHC(4) + SH_C2_2 * 2.f * -x * SHC(6) + SH_C2_3 * z * SHC(7) + SH_C2_4 * 2.f * x * SHC(8); \
                dRdy += SH_C2_0 * x * SHC(4) + SH_C2_1 * z * SHC(5) + SH_C2_2 * 2.f * -y * SHC(6) + SH_C2_4 * 2.f * -y * SHC(8); \
                dRdz += SH_C2_1 * y * SHC(5) + SH_C2_2 * 2.f * 2.f * z * SHC(6) + SH_C2_3 * x * SHC(7);                       \
                if (D > 2) {                                                                                                    \
                    dRdx += SH_C3_0 * SHC(9) * 3.f * 2.f * xy + SH_C3_1 * SHC(10) * yz + SH_C3_2 * SHC(11) * -2.f * xy +       \
                            SH_C3_3 * SHC(12) * -3.f * 2.f * xz + SH_C3_4 * SHC(13) * (-3.f * xx + 4.f * zz - yy) +            \
                            SH_C3_5 * SHC(14) * 2.f * xz + SH_C3_6 * SHC(15) * 3.f * (xx - yy);                                \
                    dRdy += SH_C3_0 * SHC(9) * 3.f * (xx - yy) + SH_C3_1 * SHC(10) * xz +                                      \
                            SH_C3_2 * SHC(11) * (-3.f * yy + 4.f * zz - xx) + SH_C3_3 * SHC(12) * -3.f * 2.f * yz +            \
                            SH_C3_4 * SHC(13) * -2.f * xy + SH_C3_5 * SHC(14) * -2.f * yz + SH_C3_6 * SHC(15) * -3.f * 2.f * xy; \
                    dRdz += SH_C3_1 * SHC(10) * xy + SH_C3_2 * SHC(11) * 4.f * 2.f * yz +                                      \
                            SH_C3_3 * SHC(12) * 3.f * (2.f * zz - xx - yy) + SH_C3_4 * SHC(13) * 4.f * 2.f * xz +              \
                            SH_C3_5 * SHC(14) * (xx - yy);                                                                      \
                }                                                                                                               \
            }                                                                                                                   \
        }                                                                                                                       \
    } while (0)

// A wave's 64 rows of K floats (12- or 24-byte AoS gradient rows), stored with unit-stride dword stores: every lane parks its
// row in the wave's private LDS scratch and the wave writes the 64 K floats back out in order.  Written lane by lane these rows
// are K stores of 4 bytes at a K-word stride; on the 5 M-splat DAS3R-shaped scene the per-Gaussian backward wrote 1.24 GB for
// 0.34 GB of gradients that way (r02 PMC WRITE_SIZE): the partial sectors leave the L2 before their neighbours arrive.
// All 64 lanes must call it; rows_valid = rows of this wave that exist (0..64).
template <int K>
__device__ __forceinline__ void wave_store_rows(float *__restrict__ dst /*row 0 of the wave*/, const int rows_valid, const float (&v)[K],
                                                float *scratch /*[64 * K], wave-private*/, const int lane) {
#pragma unroll
    for (int q = 0; q < K; q++) scratch[lane * K + q] = v[q];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");   // (the wave reads its own words back: LDS is in order per wave)
    const int n = rows_valid * K;
#pragma unroll
    for (int q = 0; q < K; q++) {
        const int e = q * 64 + lane;
        if (e < n) dst[e] = scratch[e];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");   // before the scratch is reused
}

// DEG0: the active degree is 0 (DAS3R's own setting, arguments.py sh_degree): no SH row is read and the view-direction terms
// vanish at compile time, which halves the register count of the unstaged variant (6 waves per SIMD instead of 3)
// CHAIN (round 6, das3r_raster_grads.chain): the kernel does not stop at dL/d(camera-frame means, scales, rotations, opacities) — it goes on
// through the pose pre-transform (pretransform_chain.h: the arithmetic of pretransform_backward_kernel<1>, bit for bit): chain rule to the raw
// parameters, their Adam step, dL/d(confidence), the camera's 28 sums in a fixed order.  The four gradient tensors are neither written nor
// read back (88 B per Gaussian) and the raw parameters are read once for the forward's inputs and the chain rule (40 B): the pair
// preprocess_backward + pretransform_backward<1> 0.221 -> see docs/ledger.md (bi).  Needs the raw parameters (`pre`) and an unstaged SH layout.
struct ChainArgs {
    GeometryAdam A;
    float *g_conf_flat, *g_small, *det_partials;
    uint32_t *arrived;
};
template <bool HAS_SH, bool HAS_COV, bool STAGE_IN, bool STAGE_OUT, bool DEG0 = false, bool CHAIN = false>
__global__ void __launch_bounds__(256, DEG0 ? (CHAIN ? 4 : 6) : 3) preprocess_backward_kernel(
    int P, int D_in, int M, const float *__restrict__ means3D, const float *__restrict__ scales, float scale_modifier,
    const float *__restrict__ rotations, const float *__restrict__ shs, const float *__restrict__ cov3D_precomp,
    const float *__restrict__ viewmatrix, const float *__restrict__ projmatrix, const float *__restrict__ campos, int W, int H,
    float tanfovx, float tanfovy, const uint32_t *__restrict__ tiles_touched, const uint8_t *__restrict__ clamped,
    const float *__restrict__ partial /*[I,9], row = emission slot — or, with row_exists, [I][4][12]: a row per (instance, quadrant)*/,
    const uint8_t *__restrict__ row_exists /*[I][4] or null*/, const uint32_t *__restrict__ off_by_gid,
    float *__restrict__ dL_dmeans2D /*[P,3] out*/, float *__restrict__ dL_dopacity /*[P] out*/,
    float *__restrict__ dL_dcolors_precomp /*[P,3] out, precomp mode*/, float *__restrict__ dL_dmeans3D,
    float *__restrict__ dL_dscales, float *__restrict__ dL_drot, float *__restrict__ dL_dsh, float *__restrict__ dL_dcov3D,
    const PreXform pre /*xyz != null (das3r_raster_in.pre): the raw parameters + the pose, as in preprocess.hip*/,
    const ChainArgs ch /*CHAIN only*/
#ifdef DAS3R_EXPERIMENTS
    , unsigned long long *__restrict__ trace /*common.h BLK_STAMP (tools/wg_trace.py), region 6*/
#endif
    ) {
#ifndef DAS3R_EXPERIMENTS
    constexpr unsigned long long *trace = nullptr;
#endif
    BLK_STAMP(trace, 6, 0)
    // one 208-byte (13 x float4) LDS row per lane: SH coefficients in (STAGE_IN), dL_dsh out (STAGE_OUT)
    __shared__ float4 sh_lds[(STAGE_IN || STAGE_OUT) ? 256 * 13 : 1];
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int D = DEG0 ? 0 : D_in;
    // Everything a lane needs that does not depend on another load is requested FIRST, ahead of the SH staging loads and their
    // barrier (one trip to memory instead of three in a row: tiles_touched -> off_by_gid -> partial rows used to start behind it)
    const int ic = idx < P ? idx : P - 1;
    const uint32_t ntiles_g = idx < P ? tiles_touched[ic] : 0u;
    const uint32_t e0_in = off_by_gid[ic];
    float3 mean_in, sc_in = make_float3(0.f, 0.f, 0.f);
    float4 q_in = make_float4(0.f, 0.f, 0.f, 0.f);
    float c3_in[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float raw_x = 0.f, raw_y = 0.f, raw_z = 0.f, raw_s0 = 0.f, raw_s1 = 0.f, raw_s2 = 0.f, raw_o = 0.f, raw_c = 0.f;   // CHAIN: kept for the chain rule at the end
    float4 raw_q = make_float4(0.f, 0.f, 0.f, 0.f);
    long long raw_ci = 0;
    ChainMoments moments;
    if (!HAS_COV && (CHAIN || pre.xyz != nullptr)) {   // (uniform) the forward's inputs again, from the raw parameters (pretransform_math.h: the same bits)
        PoseRegs pose;
        load_pose(pre.Rm, pre.tv, pre.Lq, pose);
        const float rx = pre.xyz[3 * ic], ry = pre.xyz[3 * ic + 1], rz = pre.xyz[3 * ic + 2];
        const float4 rq = reinterpret_cast<const float4 *>(pre.rot)[ic];
        const float r0 = pre.scaling[3 * ic], r1 = pre.scaling[3 * ic + 1], r2 = pre.scaling[3 * ic + 2];
        if constexpr (CHAIN) {
            raw_x = rx; raw_y = ry; raw_z = rz; raw_q = rq; raw_s0 = r0; raw_s1 = r1; raw_s2 = r2;
            raw_o = pre.opacity_raw[ic];
            raw_ci = pre.mask_index ? (long long)pre.mask_index[ic] : (long long)ic;
            raw_c = pre.conf_flat[raw_ci];
            chain_load_moments(ch.A, (size_t)ic, moments);
        }
        mean_in = pre_mean(pose, rx, ry, rz);
        q_in = pre_rot(pose, rq);
        sc_in = make_float3(pre_scale(r0), pre_scale(r1), pre_scale(r2));
    } else {
        mean_in = make_float3(means3D[3 * ic], means3D[3 * ic + 1], means3D[3 * ic + 2]);
        if (HAS_COV) {
#pragma unroll
            for (int i = 0; i < 6; i++) c3_in[i] = cov3D_precomp[6 * ic + i];
        } else {
            sc_in = make_float3(scales[3 * ic], scales[3 * ic + 1], scales[3 * ic + 2]);
            q_in = reinterpret_cast<const float4 *>(rotations)[ic];
        }
    }
    const size_t blk4 = (size_t)blockIdx.x * 256 * 12;                    // first float4 of this workgroup's rows
    const size_t limit4 = (size_t)P * 12 > blk4 ? (size_t)P * 12 - blk4 : 0;  // float4s this workgroup owns
    // The staged variants gather the per-instance rows THROUGH LDS: the rows of 256 consecutive splats are one contiguous run of
    // partial[] (rows are indexed by emission slot and slots are handed out in splat order), so the workgroup streams that run in
    // with coalesced loads, 1024 rows at a time, and each lane then adds its own rows out of LDS in the same order as before
    // (bit-identical sums).  A lane used to walk its rows with one trip to L2 per row: the longest walk of 64 lanes set the pace.
    constexpr bool LDS_GATHER = STAGE_OUT;
    constexpr int CHUNK_ROWS = 1020;                                      // 36 KB of the 52 KB region: 2304 float4 = 9 per thread with the 3 + 3 floats of alignment slack
    float acc[9];
#pragma unroll
    for (int q = 0; q < 9; q++) acc[q] = 0.f;
    float4 sh_in[STAGE_IN ? 12 : 1];
    // The SH rows of the workgroup (48 KB): requested with a clamped index, i.e. all twelve loads of a thread in flight at once, but
    // only BEHIND the loads of the gather (loads return in order: ahead of them they delayed the rows the kernel is waiting for —
    // r3: 0.129 -> 0.165 ms; after the gather was done their latency was exposed: 0.136 -> 0.150 ms).  Parked in registers while the
    // LDS region serves the gather.
    auto request_sh = [&]() {
        if (STAGE_IN) {
            const float4 *src = reinterpret_cast<const float4 *>(shs) + blk4;
#pragma unroll
            for (int i = 0; i < 12; i++) {
                const size_t f = (size_t)(i * 256 + threadIdx.x);
                sh_in[i] = limit4 ? src[f < limit4 ? f : limit4 - 1] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    bool gathered = false;
    uint32_t run0 = 0u, run1 = 0u;
    if (LDS_GATHER && !row_exists) {
        // Is it one run?  Splat-order emission (local depth order) makes it so; the globally sorted path hands slots out in depth
        // order and culled splats have no slot at all.  The workgroup decides for itself: the rows of its visible splats are
        // disjoint intervals, so they tile [min first, max end) exactly when their lengths add up to max end - min first.
        __shared__ uint32_t s_run[4][3];
        uint32_t lo = ntiles_g ? e0_in : 0xFFFFFFFFu, hi = ntiles_g ? e0_in + ntiles_g : 0u, cnt = ntiles_g;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            lo = min(lo, (uint32_t)__shfl_xor((int)lo, o, 64));
            hi = max(hi, (uint32_t)__shfl_xor((int)hi, o, 64));
            cnt += (uint32_t)__shfl_xor((int)cnt, o, 64);
        }
        if ((threadIdx.x & 63) == 0) {
            s_run[threadIdx.x >> 6][0] = lo;
            s_run[threadIdx.x >> 6][1] = hi;
            s_run[threadIdx.x >> 6][2] = cnt;
        }
        __syncthreads();
        run0 = min(min(s_run[0][0], s_run[1][0]), min(s_run[2][0], s_run[3][0]));
        run1 = max(max(s_run[0][1], s_run[1][1]), max(s_run[2][1], s_run[3][1]));
        const uint32_t rows = s_run[0][2] + s_run[1][2] + s_run[2][2] + s_run[3][2];
        gathered = rows == 0u || rows == run1 - run0;   // (workgroup-uniform)
        BLK_STAMP(trace, 6, 1)   // the splats' own words are here, run known
    }
    if (gathered) {
        // Round 3: the run comes in as float4 words, ALL of a thread's loads of a chunk in flight at once (it used to be 4-byte loads,
        // four at a time: six trips to memory in a row for the ~670 rows of a workgroup at 1 M splats — and this kernel runs three
        // waves per SIMD, nobody covers a trip).  A run starts at a multiple of 36 bytes: the words are taken from the 16-byte
        // boundary below it (head = 0..3 floats that are not ours, skipped when the rows are read back) to the one above its end
        // (the scratch buffer carries 16 bytes of slack: das3r_raster_backward_scratch_bytes).
        float *const buf = reinterpret_cast<float *>(sh_lds);
        float4 *const buf4 = sh_lds;
        const uint32_t mine0 = e0_in, mine1 = e0_in + ntiles_g;
        const bool aligned_base = (reinterpret_cast<uintptr_t>(partial) & 15u) == 0;   // (uniform; torch's allocations are)
        bool sh_requested = false;
        for (uint32_t c0 = run0; c0 < run1; c0 += CHUNK_ROWS) {
            const uint32_t nrows = min((uint32_t)CHUNK_ROWS, run1 - c0);
            const float *src = partial + (size_t)c0 * 9;
            const int n9 = (int)nrows * 9;
            int head = 0;
            if (aligned_base) {
                head = (int)((reinterpret_cast<uintptr_t>(src) & 15u) >> 2);
                const float4 *src4 = reinterpret_cast<const float4 *>(src - head);
                const int n4 = (head + n9 + 3) >> 2;   // <= 2304 = 9 x 256 (CHUNK_ROWS)
                float4 w[9];
#pragma unroll
                for (int i = 0; i < 9; i++) {
                    const int f = i * 256 + (int)threadIdx.x;
                    w[i] = src4[f < n4 ? f : n4 - 1];
                }
                if (!sh_requested) request_sh();
#pragma unroll
                for (int i = 0; i < 9; i++) {
                    const int f = i * 256 + (int)threadIdx.x;
                    if (f < n4) buf4[f] = w[i];
                }
            } else {
                if (!sh_requested) request_sh();
#pragma unroll 4
                for (int f = threadIdx.x; f < n9; f += 256) buf[f] = src[f];
            }
            sh_requested = true;
            __syncthreads();
            BLK_STAMP(trace, 6, 2)   // the chunk's rows are in LDS
            const uint32_t k0 = max(mine0, c0), k1 = min(mine1, c0 + nrows);
            for (uint32_t k = k0; k < k1; k++) {
                const float *row = buf + head + (k - c0) * 9;   // (stride 9 words: consecutive rows fall in different banks)
#pragma unroll
                for (int q = 0; q < 9; q++) acc[q] += row[q];
            }
            __syncthreads();
        }
        if (!sh_requested) request_sh();   // (an empty run)
    } else {
        request_sh();
    }
    BLK_STAMP(trace, 6, 3)   // rows added
    if (STAGE_IN) {
#pragma unroll
        for (int i = 0; i < 12; i++) {
            const int f = i * 256 + threadIdx.x;
            if ((size_t)f < limit4) sh_lds[(f / 12) * 13 + (f % 12)] = sh_in[i];
        }
        __syncthreads();
    }
    BLK_STAMP(trace, 6, 4)   // SH rows in LDS

    // unstaged variants: AoS gradient rows leave through a wave-private LDS scratch (wave_store_rows); the staged ones (M == 16)
    // keep their direct stores — their LDS is spoken for and their 12-byte rows are a small share next to the 192-byte SH rows
    constexpr bool XPOSE = !STAGE_OUT;
    __shared__ float xp_lds[XPOSE ? 4 * 64 * 6 : 1];
    float *const xp = xp_lds + (XPOSE ? (threadIdx.x >> 6) * 64 * 6 : 0);
    const int lane = threadIdx.x & 63;
    const int wave_first = blockIdx.x * 256 + (int)(threadIdx.x & ~63u);
    const int rows_valid = min(max(P - wave_first, 0), 64);
    float o_m2d[3] = {0.f, 0.f, 0.f}, o_m3d[3] = {0.f, 0.f, 0.f}, o_sc[3] = {0.f, 0.f, 0.f}, o_cov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float o_col[3] = {0.f, 0.f, 0.f}, o_sh0[3] = {0.f, 0.f, 0.f};
    static_assert(!CHAIN || (XPOSE && HAS_SH && !HAS_COV), "the chained backward: unstaged SH layouts, scales + rotations");
    float o_rot[4] = {0.f, 0.f, 0.f, 0.f}, o_op = 0.f;   // CHAIN: dL/d(camera-frame rotation, opacity) stay in registers like o_m3d / o_sc

    if (idx < P) {
        const bool visible = ntiles_g > 0;

        // add this splat's per-instance sums (one row per touched tile; the render backward wrote them at the emission slots)
        if (visible && !gathered) {
            const uint32_t e0 = e0_in;
#ifdef DAS3R_EXPERIMENTS
            if (row_exists) {   // render_bwd_stream.hip: up to four 48-byte rows per instance, one per quadrant of the tile that met the splat
                for (uint32_t k = 0; k < ntiles_g; k++) {
                    const uint32_t have = *reinterpret_cast<const uint32_t *>(row_exists + (size_t)(e0 + k) * 4);
#pragma unroll
                    for (int w = 0; w < 4; w++) {
                        if ((have >> (8 * w)) & 0xffu) {
                            const float4 *row = reinterpret_cast<const float4 *>(partial + ((size_t)(e0 + k) * 4 + w) * 12);
                            const float4 r0 = row[0], r1 = row[1];
                            const float r2 = reinterpret_cast<const float *>(row)[8];
                            acc[0] += r0.x; acc[1] += r0.y; acc[2] += r0.z; acc[3] += r0.w;
                            acc[4] += r1.x; acc[5] += r1.y; acc[6] += r1.z; acc[7] += r1.w;
                            acc[8] += r2;
                        }
                    }
                }
            } else
#endif
            {
                for (uint32_t k = 0; k < ntiles_g; k++) {
                    const float *row = partial + (size_t)(e0 + k) * 9;   // rows are indexed by emission slot: contiguous per splat
#pragma unroll
                    for (int q = 0; q < 9; q++) acc[q] += row[q];
                }
            }
        }
        if constexpr (CHAIN) o_op = acc[8];
        else dL_dopacity[idx] = acc[8];
        if (XPOSE) {
            o_m2d[0] = acc[3]; o_m2d[1] = acc[4];
            o_col[0] = acc[0]; o_col[1] = acc[1]; o_col[2] = acc[2];
        } else {
            dL_dmeans2D[3 * (size_t)idx] = acc[3];
            dL_dmeans2D[3 * (size_t)idx + 1] = acc[4];
            dL_dmeans2D[3 * (size_t)idx + 2] = 0.f;
            if (!HAS_SH) {
                dL_dcolors_precomp[3 * (size_t)idx] = acc[0];
                dL_dcolors_precomp[3 * (size_t)idx + 1] = acc[1];
                dL_dcolors_precomp[3 * (size_t)idx + 2] = acc[2];
            }
        }

        float dmean[3] = {0.f, 0.f, 0.f};
        float dcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        float dscale[3] = {0.f, 0.f, 0.f};
        float drot[4] = {0.f, 0.f, 0.f, 0.f};
        float basis[16];          // SH basis at this splat's view direction (dL_dsh[k][c] = basis[k] * g[c])
        float g[3] = {0.f, 0.f, 0.f};

        if (visible) {
            float V[16], PM[16];
#pragma unroll
            for (int i = 0; i < 16; i++) {
                V[i] = viewmatrix[i];
                PM[i] = projmatrix[i];
            }
            const float focal_x = W / (2.0f * tanfovx), focal_y = H / (2.0f * tanfovy);
            const float3 mean = mean_in;
            const float3 sc = sc_in;
            const float4 q = q_in;
            float c3[6];
            if (HAS_COV) {
#pragma unroll
                for (int i = 0; i < 6; i++) c3[i] = c3_in[i];
            } else {
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
            // ---- colour -> view direction (needs the SH coefficients) and SH basis   [computeColorFromSH bwd]
            if (HAS_SH) {
                const float dox = mean.x - campos[0], doy = mean.y - campos[1], doz = mean.z - campos[2];
                const float inv_len = 1.f / sqrtf(dox * dox + doy * doy + doz * doz);
                const float x = dox * inv_len, y = doy * inv_len, z = doz * inv_len;
                const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                const uint8_t cl = clamped[idx];
#pragma unroll
                for (int c = 0; c < 3; c++) g[c] = ((cl >> c) & 1) ? 0.f : acc[c];
                sh_basis(D, x, y, z, basis);
                float ddir[3] = {0.f, 0.f, 0.f};
                if (D > 0) {
                    if (STAGE_IN) {
                        // the lane's 48 coefficients come out of LDS as twelve 16-byte reads (rows are 13 quad-words apart: no
                        // bank conflict); single-word reads of the same rows collide four ways (52-word stride; r02 PMC:
                        // 6.8e6 conflict cycles per launch at 1 M splats)
                        float shl[48];
#pragma unroll
                        for (int i = 0; i < 12; i++) {
                            const float4 v = sh_lds[threadIdx.x * 13 + i];
                            shl[4 * i] = v.x; shl[4 * i + 1] = v.y; shl[4 * i + 2] = v.z; shl[4 * i + 3] = v.w;
                        }
#pragma unroll
                        for (int c = 0; c < 3; c++) {
                            float dRdx, dRdy, dRdz;
#define SHC_LDS(k) shl[(k) * 3 + c]
                            SH_DDIR(SHC_LDS, dRdx, dRdy, dRdz);
#undef SHC_LDS
                            ddir[0] += dRdx * g[c];
                            ddir[1] += dRdy * g[c];
                            ddir[2] += dRdz * g[c];
                        }
                    } else {
                        const float *row = shs + (size_t)idx * M * 3;
                        float sh[48];
                        load_sh_row(row, D, sh_vec_ok(row, D, M), sh);
#pragma unroll
                        for (int c = 0; c < 3; c++) {
                            float dRdx, dRdy, dRdz;
#define SHC_REG(k) sh[(k) * 3 + c]
                            SH_DDIR(SHC_REG, dRdx, dRdy, dRdz);
#undef SHC_REG
                            ddir[0] += dRdx * g[c];
                            ddir[1] += dRdy * g[c];
                            ddir[2] += dRdz * g[c];
                        }
                    }
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

        if (XPOSE) {
#pragma unroll
            for (int k = 0; k < 3; k++) o_m3d[k] = dmean[k], o_sc[k] = dscale[k];
#pragma unroll
            for (int i = 0; i < 6; i++) o_cov[i] = dcov[i];
        } else {
            dL_dmeans3D[3 * (size_t)idx] = dmean[0];
            dL_dmeans3D[3 * (size_t)idx + 1] = dmean[1];
            dL_dmeans3D[3 * (size_t)idx + 2] = dmean[2];
            if (HAS_COV) {
#pragma unroll
                for (int i = 0; i < 6; i++) dL_dcov3D[6 * (size_t)idx + i] = dcov[i];
            } else {
#pragma unroll
                for (int k = 0; k < 3; k++) dL_dscales[3 * (size_t)idx + k] = dscale[k];
            }
        }
        if constexpr (CHAIN) {
#pragma unroll
            for (int k = 0; k < 4; k++) o_rot[k] = drot[k];
        } else if (!HAS_COV) reinterpret_cast<float4 *>(dL_drot)[idx] = make_float4(drot[0], drot[1], drot[2], drot[3]);   // (16-byte rows: unit stride as they are)
        // ---- dL_dsh row: basis[k] * g[c] for the active coefficients, zero above; zero row for culled splats
        if (HAS_SH) {
            const int nk = visible ? (D + 1) * (D + 1) : 0;
            if (STAGE_OUT) {   // own LDS row (the SH inputs staged there are dead by now): 12 x ds_write_b128
#pragma unroll
                for (int i = 0; i < 12; i++) {
                    float o[4];
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const int f = 4 * i + e, k = f / 3, c = f - 3 * k;   // compile-time after unrolling
                        o[e] = k < nk ? basis[k] * g[c] : 0.f;
                    }
                    sh_lds[threadIdx.x * 13 + i] = make_float4(o[0], o[1], o[2], o[3]);
                }
            } else if (XPOSE && M == 1) {   // (uniform) one coefficient per splat: a 12-byte row like the others
#pragma unroll
                for (int c = 0; c < 3; c++) o_sh0[c] = nk > 0 ? basis[0] * g[c] : 0.f;
            } else {
                float *row = dL_dsh + (size_t)idx * M * 3;
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    if (k < M) {
                        row[3 * k] = k < nk ? basis[k] * g[0] : 0.f;
                        row[3 * k + 1] = k < nk ? basis[k] * g[1] : 0.f;
                        row[3 * k + 2] = k < nk ? basis[k] * g[2] : 0.f;
                    }
                }
            }
        }
    }
    if (XPOSE) {   // every lane of the wave takes part (rows_valid guards the stores)
        wave_store_rows<3>(dL_dmeans2D + 3 * (size_t)wave_first, rows_valid, o_m2d, xp, lane);
        if (!CHAIN) wave_store_rows<3>(dL_dmeans3D + 3 * (size_t)wave_first, rows_valid, o_m3d, xp, lane);
        if (HAS_COV) wave_store_rows<6>(dL_dcov3D + 6 * (size_t)wave_first, rows_valid, o_cov, xp, lane);
        else if (!CHAIN) wave_store_rows<3>(dL_dscales + 3 * (size_t)wave_first, rows_valid, o_sc, xp, lane);
        if (!HAS_SH) wave_store_rows<3>(dL_dcolors_precomp + 3 * (size_t)wave_first, rows_valid, o_col, xp, lane);
        else if (M == 1) wave_store_rows<3>(dL_dsh + 3 * (size_t)wave_first, rows_valid, o_sh0, xp, lane);
    }
    BLK_STAMP(trace, 6, 5)   // per-Gaussian arithmetic done, small rows stored
    if (STAGE_OUT) {
        __syncthreads();
        float4 *dst = reinterpret_cast<float4 *>(dL_dsh) + blk4;
#pragma unroll
        for (int i = 0; i < 12; i++) {
            const int f = i * 256 + threadIdx.x;
            if ((size_t)f < limit4) dst[f] = sh_lds[(f / 12) * 13 + (f % 12)];
        }
    }
#ifdef DAS3R_EXPERIMENTS
    if (trace != nullptr) {
        BLK_STAMP(trace, 6, 6)   // dL_dsh rows issued
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        BLK_STAMP(trace, 6, 7)   // acknowledged
    }
#endif
    if constexpr (CHAIN) {   // ---- on through the pose pre-transform: pretransform_chain.h, the work of pretransform_backward_kernel<1> ----
        __shared__ float red[4][28];
        float Rp[9], Lp[16], pacc[28];
#pragma unroll
        for (int i = 0; i < 9; i++) Rp[i] = pre.Rm[i];
#pragma unroll
        for (int i = 0; i < 16; i++) Lp[i] = pre.Lq[i];
#pragma unroll
        for (int i = 0; i < 28; i++) pacc[i] = 0.f;
        if (idx < P) {   // (a culled Gaussian: zero gradients — its Adam step still decays its moments, as in the two-kernel form)
            ChainIn in;
            in.x = raw_x; in.y = raw_y; in.z = raw_z; in.q = raw_q;
            in.sc[0] = raw_s0; in.sc[1] = raw_s1; in.sc[2] = raw_s2; in.o = raw_o; in.c = raw_c;
            in.gx = o_m3d[0]; in.gy = o_m3d[1]; in.gz = o_m3d[2];
            in.gq = make_float4(o_rot[0], o_rot[1], o_rot[2], o_rot[3]);
            in.gs[0] = o_sc[0]; in.gs[1] = o_sc[1]; in.gs[2] = o_sc[2]; in.go = o_op;
            chain_pose_acc(in, pacc);
            ChainOut o;
            chain_grads(Rp, Lp, in, o);
            ch.g_conf_flat[raw_ci] = o.gconf;   // mask positions are unique: plain store
            chain_adam(ch.A, (size_t)idx, in, o, moments);
        }
        pose_sums_finish(pacc, red, ch.det_partials, ch.arrived, ch.g_small);
    }
}

// Scratch of the chained form's fixed-order pose sums: one row of 28 floats per workgroup + the arrival word, per (host thread, device,
// stream) like pretransform.hip's; grown when a larger model comes along.  Zeroed when allocated (the kernel re-arms the word).
static float *chain_scratch(hipStream_t s, size_t blocks, uint32_t **arrived) {
    struct Slot { int dev; hipStream_t stream; float *buf; size_t rows; };
    static thread_local std::vector<Slot> slots;
    *arrived = nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    Slot *hit = nullptr;
    for (auto &e : slots)
        if (e.dev == dev && e.stream == s) hit = &e;
    if (!hit || hit->rows < blocks) {
        auto complain = [] {
            static bool said = false;
            if (!__atomic_exchange_n(&said, true, __ATOMIC_RELAXED))
                fprintf(stderr, "[das3r] no scratch for the fixed-order pose sums: this thread's pose gradients are summed with float atomics (not bit-reproducible)\n");
        };
        const size_t rows = std::max(blocks, (size_t)4096) * 2;
        float *buf = nullptr;
        if (hit) { (void)hipStreamSynchronize(s); (void)hipFree(hit->buf); hit->buf = nullptr; hit->rows = 0; }
        if (hipMalloc((void **)&buf, (rows * 28 + 4) * sizeof(float)) != hipSuccess) { complain(); return nullptr; }
        if (hipMemsetAsync(buf, 0, (rows * 28 + 4) * sizeof(float), s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) { (void)hipFree(buf); complain(); return nullptr; }
        if (hit) { hit->buf = buf; hit->rows = rows; }
        else { slots.push_back({dev, s, buf, rows}); hit = &slots.back(); }
    }
    if (!hit->buf) return nullptr;
    *arrived = reinterpret_cast<uint32_t *>(hit->buf + hit->rows * 28);
    return hit->buf;
}

int launch_preprocess_backward(const das3r_raster_args *a, const das3r_raster_in *in, char *geom, char *binning, const Layout &L,
                               const das3r_raster_grads *g, const float *partial, hipStream_t s, bool quad_rows) {
    const int P = a->P;
    if (P == 0) return DAS3R_OK;
    dim3 grid(div_up(P, 256)), block(256);
    const bool has_sh = in->shs != nullptr, has_cov = in->cov3D_precomp != nullptr;
    const bool nostage = switches().no_sh_stage;
    const size_t cap_rows = L.capacity > 0 ? (size_t)L.capacity : 1;
    const uint8_t *exists = quad_rows ? reinterpret_cast<const uint8_t *>(partial) + align_up(cap_rows * 4 * 12 * sizeof(float)) : nullptr;
    const bool stage_out = has_sh && a->M == 16 && ((uintptr_t)g->dL_dshs & 15) == 0 && !nostage;
    const bool stage_in = stage_out && a->sh_degree >= 2 && ((uintptr_t)in->shs & 15) == 0;
#ifdef DAS3R_EXPERIMENTS
#define PB_TRACE_ARG , wg_trace()
#else
#define PB_TRACE_ARG
#endif
    const PreXform pre = pre_xform(in);
#define ARGS                                                                                                                 \
    P, a->sh_degree, a->M, in->means3D, in->scales, a->scale_modifier, in->rotations, in->shs, in->cov3D_precomp,            \
        a->viewmatrix, a->projmatrix, a->campos, a->image_width, a->image_height, a->tanfovx, a->tanfovy,                    \
        (const uint32_t *)(geom + L.pub.tiles_touched), (const uint8_t *)(geom + L.pub.clamped), partial, exists,        \
        (const uint32_t *)(geom + L.g_off_by_gid), g->dL_dmeans2D, g->dL_dopacities, g->dL_dcolors_precomp, g->dL_dmeans3D,  \
        g->dL_dscales, g->dL_drotations, g->dL_dshs, g->dL_dcov3D, pre, ch PB_TRACE_ARG
    // ---- the chained form (das3r_raster_grads.chain): unstaged SH rows, raw parameters ----
    ChainArgs ch;
    memset(&ch, 0, sizeof(ch));
    if (g->chain) {
        const das3r_chain *c = g->chain;
        if (!pre.xyz || !pre.opacity_raw || !pre.conf_flat || !has_sh || has_cov || stage_out || quad_rows || !c->slots || !c->g_conf_flat || !c->g_small) {
            set_error("das3r_raster_backward: grads->chain needs in->pre, SH coefficients in an unstaged layout (M < 16 or the active degree below 2 ...), "
                      "scales + rotations, and g_conf_flat / g_small / slots");
            return DAS3R_ERR_INVALID_ARG;
        }
        for (int k = 0; k < 4; k++) {
            if (!c->slots[k].param || !c->slots[k].exp_avg || !c->slots[k].exp_avg_sq || !(c->slots[k].bc2_sqrt > 0.f)) {
                set_error("das3r_raster_backward: grads->chain: bad slot %d", k);
                return DAS3R_ERR_INVALID_ARG;
            }
            ch.A.p[k] = c->slots[k].param; ch.A.m[k] = c->slots[k].exp_avg; ch.A.v[k] = c->slots[k].exp_avg_sq;
            ch.A.step_size[k] = c->slots[k].step_size; ch.A.bc2_sqrt[k] = c->slots[k].bc2_sqrt;
        }
        if (ch.A.p[0] != pre.xyz || ch.A.p[1] != pre.rot || ch.A.p[2] != pre.scaling || ch.A.p[3] != pre.opacity_raw) {
            set_error("das3r_raster_backward: grads->chain: the slots must be the tensors of in->pre (xyz, rot, scaling, opacity_raw)");
            return DAS3R_ERR_INVALID_ARG;
        }
        ch.A.beta1 = c->beta1; ch.A.beta2 = c->beta2; ch.A.eps = c->eps;
        ch.g_conf_flat = c->g_conf_flat; ch.g_small = c->g_small;
        ch.det_partials = chain_scratch(s, (size_t)grid.x, &ch.arrived);   // (null: float atomics, said once)
    }
#define LAUNCH(SH, COV, SI, SO) DAS3R_LAUNCH((preprocess_backward_kernel<SH, COV, SI, SO>), grid, block, 0, s, ARGS)
#define LAUNCH0(SH, COV) DAS3R_LAUNCH((preprocess_backward_kernel<SH, COV, false, false, true>), grid, block, 0, s, ARGS)
    const bool deg0 = a->sh_degree == 0 && !stage_out;
    if (g->chain) {
        if (deg0) DAS3R_LAUNCH((preprocess_backward_kernel<true, false, false, false, true, true>), grid, block, 0, s, ARGS);
        else DAS3R_LAUNCH((preprocess_backward_kernel<true, false, false, false, false, true>), grid, block, 0, s, ARGS);
    } else if (has_sh && !has_cov) {
        if (stage_in) LAUNCH(true, false, true, true);
        else if (stage_out) LAUNCH(true, false, false, true);
        else if (deg0) LAUNCH0(true, false);
        else LAUNCH(true, false, false, false);
    } else if (has_sh && has_cov) {
        if (stage_in) LAUNCH(true, true, true, true);
        else if (stage_out) LAUNCH(true, true, false, true);
        else if (deg0) LAUNCH0(true, true);
        else LAUNCH(true, true, false, false);
    } else if (!has_sh && !has_cov) {
        LAUNCH(false, false, false, false);
    } else {
        LAUNCH(false, true, false, false);
    }
#undef LAUNCH
#undef LAUNCH0
#undef ARGS
    KERNEL_CHECK(s, a->debug, "preprocess_backward");
    return DAS3R_OK;
}

}  // namespace das3r
