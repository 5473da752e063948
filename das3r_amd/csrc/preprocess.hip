// preprocess.hip — K1: per-Gaussian cull, projection, 3D->2D covariance (EWA), conic, 3-sigma extent, tile
// rect, SH -> RGB.  Replaces upstream:cuda_rasterizer/forward.cu preprocessCUDA for the call
// at /root/reference/gaussian_renderer/__init__.py:131-140.
//
// Pure streaming map, HBM-bound: 44 + 12(D+1)^2 input bytes and 53 output bytes per Gaussian.  One lane per
// Gaussian, 256-lane blocks; the (P,M,3) SH rows are 16-byte aligned (M = 16 -> 192 B) and are read as float4;
// outputs are SoA so every store instruction of a wave writes one contiguous run.
// Compiled with -ffp-contract=off: all arithmetic is plain IEEE fp32 (+,-,*,/,sqrt correctly rounded), which
// makes radii / tile rects / conics reproducible bit-for-bit by the CPU oracle.
#include "common.h"
#include "granule.h"
#include "segkey.h"
#include "splat_math.h"
#include "pretransform_math.h"

namespace das3r {

template <bool HAS_SH, bool HAS_COV, bool STAGE>
__global__ void __launch_bounds__(256) preprocess_kernel(
    int P, int D, int M, const float *__restrict__ means3D, const float *__restrict__ scales, float scale_modifier,
    const float *__restrict__ rotations, const float *__restrict__ opacities, const float *__restrict__ shs,
    const float *__restrict__ cov3D_precomp, const float *__restrict__ colors_precomp, const float *__restrict__ viewmatrix,
    const float *__restrict__ projmatrix, const float *__restrict__ campos, int W, int H, float tanfovx, float tanfovy,
    int tiles_x, int tiles_y, int32_t *__restrict__ radii, uint32_t *__restrict__ depth_key, float4 *__restrict__ xyh,
    float4 *__restrict__ conic_opacity, float4 *__restrict__ rgbd, uint8_t *__restrict__ clamped,
    uint32_t *__restrict__ tiles_touched, uint32_t *__restrict__ rect32, int tight_rect /*bit 0: opacity-aware clipped rectangle; bit 1: prefiltered*/, uint32_t *__restrict__ zero_a, uint32_t zero_a_words, uint32_t *__restrict__ zero_b,
    uint32_t zero_b_words, uint32_t *__restrict__ zero_c, uint32_t zero_c_words, unsigned long long *__restrict__ arrive,
    uint32_t *__restrict__ host_out, uint32_t tag, const EmitArgs em,
    uint32_t *__restrict__ dhist /*segmented binning path: 256-bin depth histogram of this forward, zeroed by the caller (segkey.h); else null*/,
    uint32_t dhist_mask /*a pseudo-random 1 / (mask + 1) of the workgroups contribute (`sampled` below): a sample is all the bucket map needs*/,
    uint32_t *__restrict__ dhist_next /*round 6: the library's OTHER histogram slot, zeroed here for the next forward of this stream (api.hip dhist_slots); else null*/,
    const PreXform pre /*xyz != null (round 6, das3r_raster_in.pre): the raw parameters + the pose; means3D / scales / rotations / opacities are not read*/) {
    const int gidx = blockIdx.x * blockDim.x + threadIdx.x;
    // Which workgroups sample the depth histogram: a full-avalanche hash of the index (round 5).  "Every (mask + 1)-th workgroup" is a
    // biased sample of a DAS3R model — its Gaussians are the pixels of its frames in row-major order, 256 of them are half an image row, and
    // every 16th workgroup of the Sintel shape is the LEFT half of every 8th row: the histogram never saw the depths of the right half of the
    // scene, whose instances then piled up in the end buckets — segments of thousands, the segmented path backing off to the global sort on
    // every self-consistent sequence (tools/probes/job_binning_trace.py).
    bool sampled = false;   // (uniform)
    if (dhist != nullptr) {
        uint32_t h = blockIdx.x * 0x9E3779B1u;
        h ^= h >> 15; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
        sampled = (h & dhist_mask) == 0u;
    }
    // Every per-Gaussian input is requested FIRST, ahead of the SH staging loads and their barrier: one trip to memory per
    // workgroup instead of two back to back (the kernel spent 77 % of its wave cycles parked on s_waitcnt at 1 M splats).
    const bool live = gidx < P;   // lanes past the end stay alive (workgroup-wide reduction below): they redo the last splat and store nothing
    const int idx = live ? gidx : P - 1;
    float3 p, s_in = make_float3(0.f, 0.f, 0.f);
    float op;
    float4 q_in = make_float4(0.f, 0.f, 0.f, 0.f);
    float c3_in[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (!HAS_COV && pre.xyz != nullptr) {   // (uniform) the pose pre-transform on the way in: pretransform_math.h, the bits of pretransform_forward_kernel
        PoseRegs pose;
        load_pose(pre.Rm, pre.tv, pre.Lq, pose);
        const float rx = pre.xyz[3 * idx], ry = pre.xyz[3 * idx + 1], rz = pre.xyz[3 * idx + 2];
        const float4 rq = reinterpret_cast<const float4 *>(pre.rot)[idx];
        const float r0 = pre.scaling[3 * idx], r1 = pre.scaling[3 * idx + 1], r2 = pre.scaling[3 * idx + 2];
        const float ro = pre.opacity_raw[idx];
        const float rc = pre.conf_flat[pre.mask_index ? pre.mask_index[idx] : (int64_t)idx];
        p = pre_mean(pose, rx, ry, rz);
        q_in = pre_rot(pose, rq);
        s_in = make_float3(pre_scale(r0), pre_scale(r1), pre_scale(r2));
        op = pre_opacity(ro, rc);
    } else {
        p = make_float3(means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]);
        op = opacities[idx];
        if (HAS_COV) {
#pragma unroll
            for (int i = 0; i < 6; i++) c3_in[i] = cov3D_precomp[6 * idx + i];
        } else {
            s_in = make_float3(scales[3 * idx], scales[3 * idx + 1], scales[3 * idx + 2]);
            q_in = reinterpret_cast<const float4 *>(rotations)[idx];
        }
    }
    __shared__ uint32_t s_tiles;   // this workgroup's sum of tiles_touched (num_rendered is their grand total)
    if (threadIdx.x == 0) s_tiles = 0;
    if (!STAGE) __syncthreads();
    // this kernel runs before every consumer of the radix control words (geom side) and of the tile ranges: zero them here
    // instead of spending two memset launches
    for (uint32_t i = gidx; i < zero_a_words; i += gridDim.x * blockDim.x) zero_a[i] = 0u;
    for (uint32_t i = gidx; i < zero_b_words; i += gridDim.x * blockDim.x) zero_b[i] = 0u;
    for (uint32_t i = gidx; i < zero_c_words; i += gridDim.x * blockDim.x) zero_c[i] = 0u;   // binning control words (hinted path)
    if (dhist_next != nullptr && gidx < DBINS) dhist_next[gidx] = 0u;
    // STAGE (M == 16, degree >= 2): the workgroup's 256 SH rows (192 B each, contiguous) are fetched with fully coalesced
    // float4 loads — every 128-B line exactly once — and re-read per lane from LDS.  Per-lane strided row loads re-fetch
    // lines evicted from L1/L2 between the 12 loads of a row: measured 2.5x the algorithmic HBM traffic at 1M splats.
    // Rows are padded to 13 float4 (208 B) so that the per-lane ds_read_b128 of a 16-lane group hit 16 distinct bank slots.
    __shared__ float4 sh_lds[STAGE ? 256 * 13 : 256 * 5];   // (the outgoing records and the fused emission's digit histograms reuse it)
    if (STAGE) {
        const float4 *src = reinterpret_cast<const float4 *>(shs) + (size_t)blockIdx.x * 256 * 12;
        const size_t limit = (size_t)P * 12 - (size_t)blockIdx.x * 256 * 12;  // float4s available from src (>= 12: the workgroup has a splat)
        // All twelve loads are issued back to back, into registers, with a CLAMPED index instead of a guard: guarded
        // (`if (f < limit) lds[..] = src[f]`) every load sat in its own basic block behind s_waitcnt vmcnt(0) — twelve trips to memory
        // one after the other, 43 of the kernel's 78 us at 1 M splats (r3 ablation: no SH loads 34 us).
        float4 v[12];
#pragma unroll
        for (int i = 0; i < 12; i++) {
            const size_t f = (size_t)(i * 256 + threadIdx.x);
            v[i] = src[f < limit ? f : limit - 1];
        }
#pragma unroll
        for (int i = 0; i < 12; i++) {
            const int f = i * 256 + threadIdx.x;
            sh_lds[(f / 12) * 13 + (f % 12)] = v[i];   // (rows past the end hold copies of the last float4: never read)
        }
        __syncthreads();
    }
    float V[16], PM[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        V[i] = viewmatrix[i];   // uniform -> scalar loads
        PM[i] = projmatrix[i];
    }
    const float focal_x = W / (2.0f * tanfovx), focal_y = H / (2.0f * tanfovy);

    int radius_out = 0;
    uint32_t key_out = 0xFFFFFFFFu, tiles_out = 0;
    uint8_t clamp_out = 0;
    float4 xy_out = make_float4(0.f, 0.f, -1e30f, -1e30f);
    float4 co_out = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 rgbd_out = make_float4(0.f, 0.f, 0.f, 0.f);
    int bx0 = 0, by0 = 0, bx1 = 0, by1 = 0;   // binned tile rectangle (fused emission)

    const float3 p_view = xform43(p, V);
    if (p_view.z > NEAR_PLANE) {
        const float4 p_hom = xform44(p, PM);
        const float p_w = 1.0f / (p_hom.w + 0.0000001f);
        const float ndc_x = p_hom.x * p_w, ndc_y = p_hom.y * p_w;

        float c3[6];
        if (HAS_COV) {
#pragma unroll
            for (int i = 0; i < 6; i++) c3[i] = c3_in[i];
        } else {
            cov3d_from_scale_rot(s_in, scale_modifier, q_in, c3);
        }
        float T[2][3];
        float3 t;
        bool cx, cy;
        ewa_T(p_view, V, focal_x, focal_y, tanfovx, tanfovy, T, t, cx, cy);
        float a, b, c;
        cov2d_from_T(T, c3, a, b, c);
        const float det = a * c - b * b;
        if (det != 0.0f) {
            const float det_inv = 1.f / det;
            const float conA = c * det_inv, conB = -b * det_inv, conC = a * det_inv;
            const float mid = 0.5f * (a + c);
            const float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
            const float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
            const float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
            const float px = ((ndc_x + 1.0f) * W - 1.0f) * 0.5f;
            const float py = ((ndc_y + 1.0f) * H - 1.0f) * 0.5f;
            const int r = (int)my_radius;
            int rminx, rminy, rmaxx, rmaxy;
            tile_rect(px, py, r, tiles_x, tiles_y, rminx, rminy, rmaxx, rmaxy);
            const int area = (rmaxx - rminx) * (rmaxy - rminy);
            if (area != 0) {
                float3 col;
                if (HAS_SH && STAGE) {
                    float sh[48];
                    const int n4 = (3 * (D + 1) * (D + 1) + 3) >> 2;
#pragma unroll
                    for (int i = 0; i < 12; i++) {
                        if (i < n4) {
                            const float4 v = sh_lds[threadIdx.x * 13 + i];
                            sh[4 * i] = v.x; sh[4 * i + 1] = v.y; sh[4 * i + 2] = v.z; sh[4 * i + 3] = v.w;
                        }
                    }
                    col = sh_regs_to_rgb(D, sh, p, campos, clamp_out);
                } else if (HAS_SH) {
                    col = sh_to_rgb(D, M, shs + (size_t)idx * M * 3, p, campos, clamp_out);
                } else {
                    col = make_float3(colors_precomp[3 * idx], colors_precomp[3 * idx + 1], colors_precomp[3 * idx + 2]);
                }
                radius_out = r;
                key_out = __float_as_uint(p_view.z);
                // half extents of the axis-aligned box outside which alpha = opacity * exp(power) cannot reach 1/255
                // (ellipse d^T Sigma'^-1 d <= 2 ln(255 o)); generous safety margin, used only for wave-level culling
                float hx = -1e30f, hy = -1e30f;
                if (255.0f * op > 1.0f) {
                    const float tau2 = 2.0f * __logf(255.0f * op) * 1.0005f + 1e-3f;
                    hx = sqrtf(tau2 * a) * 1.0005f + 0.02f;
                    hy = sqrtf(tau2 * c) * 1.0005f + 0.02f;
                }
                xy_out = make_float4(px, py, hx, hy);
                // instances are binned over the clipped rectangle (radii keeps upstream's value)
                binned_rect(xy_out, r, tiles_x, tiles_y, (tight_rect & 1) != 0, bx0, by0, bx1, by1);
                tiles_out = (uint32_t)((bx1 - bx0) * (by1 - by0));
                co_out = make_float4(conA, conB, conC, op);
                rgbd_out = make_float4(col.x, col.y, col.z, p_view.z);
            }
        }
    } else if ((tight_rect & 2) && live) {
        // prefiltered = true is the caller's promise that no point is culled here; upstream:auxiliary.h in_frustum prints "Point is
        // filtered although prefiltered is set" and traps.  Here the forward fails with that message: the tag of this call in word 12
        // of the host mailbox, examined by das3r_raster_forward once the count has arrived (api.hip)
        __hip_atomic_store(host_out + 12, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __threadfence_system();
    }
    if (live) {
    radii[idx] = radius_out;
    depth_key[idx] = key_out;
    tiles_touched[idx] = tiles_out;
    // the binned rectangle, packed (tile counts <= 255 per axis: 4080 pixels; above that the scan reads the records instead)
    rect32[idx] = tiles_out ? ((uint32_t)bx0 | ((uint32_t)bx1 << 8) | ((uint32_t)by0 << 16) | ((uint32_t)by1 << 24)) : 0u;
    clamped[idx] = clamp_out;
    }
    // The 64-byte records (xyh | conic + opacity | rgb + depth | radius, tiles_touched) leave through LDS: written per lane they
    // are four 16-byte stores at a 64-byte stride (256 separate write requests per wave); the workgroup's 256 records are one
    // contiguous 16 KB run, stored here as four fully coalesced float4 sweeps.  The fourth float4 serves the scan/emit kernel,
    // which walks the splats in depth order and would otherwise pay three random cache lines per splat.
    {
        __syncthreads();   // every lane is done with its SH row: the staging area is free
        float4 *rec = sh_lds;   // [256][5]: records 80 bytes apart, so that the per-lane b128 writes spread over all banks
        const int t = threadIdx.x, slot = t * 5;
        rec[slot] = xy_out;
        rec[slot + 1] = co_out;
        rec[slot + 2] = rgbd_out;
        rec[slot + 3] = make_float4(__int_as_float(radius_out), __uint_as_float(tiles_out), 0.f, 0.f);
        __syncthreads();
        float4 *dst = xyh + (size_t)blockIdx.x * 256 * SPLAT_REC;
        const size_t lim = ((size_t)P - (size_t)blockIdx.x * 256) * SPLAT_REC;   // float4s of this workgroup's live records
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int f = i * 256 + t;
            if ((size_t)f < lim) dst[f] = rec[f + (f >> 2)];
        }
        if (em.status != nullptr || sampled) __syncthreads();   // (the fused emission / the depth histogram reuse the area once more)
    }
    // Segmented binning path: this workgroup's share of the forward's depth histogram (weights = tiles_touched: instances, not
    // splats), counted in LDS — integer ds_add is cheap, unlike the float one — and handed on with one atomic per NON-EMPTY bin:
    // 256 consecutive splats of a real sequence sit in a handful of bins.  Only every (mask + 1)-th workgroup takes part: the
    // histogram steers how evenly the buckets fill, nothing else (any histogram gives a monotone map), and a few hundred
    // workgroups' worth of splats is sample enough — with all 19 531 workgroups of the 5 M-splat benchmark (random depths: ~100
    // non-empty bins each) the 2 M global atomics on 256 addresses cost this kernel 0.16 ms (0.246 -> 0.407).
    if (sampled) {   // (uniform)
        uint32_t *lh = reinterpret_cast<uint32_t *>(sh_lds);
        lh[threadIdx.x] = 0u;
        __syncthreads();
        if (live && tiles_out) atomicAdd(&lh[depth_bin(key_out)], tiles_out);
        __syncthreads();
        const uint32_t c = lh[threadIdx.x];
        if (c) atomicAdd(&dhist[threadIdx.x], c);
    }

    // num_rendered = sum of tiles_touched does not depend on the depth order: deliver it to the host NOW, five kernels before the
    // scan that needs it on the device, so that the host can size the binning buffer exactly without ever waiting for the sort.
    // One 64-bit atomic per workgroup carries (workgroups done << 40 | tiles); the last arriver owns the total, re-arms the
    // counters for their next use and writes {count, tag} to the pinned mailbox (see api.hip).
    // Fused emission (speculative local-order path, the whole grid resident: common.h EmitArgs): the index-order scan of
    // tiles_touched and the (tile id, splat) instances of scan_emit.hip, without its launch, its drain and its second trip to the
    // records — this workgroup's splats are still in registers.
    if (em.status != nullptr) {   // (uniform)
        __shared__ uint32_t ws[4], s_carry;
        uint32_t(*h)[RADIX_SIZE] = reinterpret_cast<uint32_t(*)[RADIX_SIZE]>(sh_lds);   // [4][256]: the SH rows are dead (4 KB more
                                                                                        // LDS would cost a third of the occupancy)
        const int tid = threadIdx.x, lane = __lane_id(), wave = tid >> 6;
        const uint32_t b = blockIdx.x;
        const uint32_t mine = live ? tiles_out : 0u;
        uint32_t total;
        const uint32_t ex = block_exclusive_scan_256(mine, ws, &total);   // (its barriers: every lane is done with its SH row)
#pragma unroll
        for (int q = 0; q < 4; q++) h[q][tid] = 0u;                      // (published by the barrier behind the look-back)
        if (wave == 0) {   // one wave publishes and looks back for the whole workgroup: own group + totals of the earlier groups
            constexpr uint32_t GS = 64u;
            const uint32_t grp = b / GS, r = b % GS, nblocks = gridDim.x;
            const u64 t64 = (u64)em.tag << 32;
            if (lane == 0) granule_store(em.status + b, t64 | total);
            uint32_t in_group = 0, before = 0;
            unsigned spins = 0;
            {
                const bool has = (uint32_t)lane < r;
                u64 x = t64;
                while (true) {
                    if (has) x = granule_poll(em.status + (b - r) + lane, spins);
                    if (__all((x >> 32) == (u64)em.tag)) break;
                    if (spins == SOFT_SPINS && lane == 0) atomicOr(em.err, ERR_HARD_POLL);
                    if (++spins > SPIN_LIMIT) { if (lane == 0) atomicOr(em.err, ERR_TIMEOUT); break; }
                    __builtin_amdgcn_s_sleep(2);
                }
                in_group = has ? (uint32_t)x : 0u;
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) in_group += (uint32_t)__shfl_xor((int)in_group, o, 64);
            }
            if (r == GS - 1u && lane == 0) granule_store(em.status + nblocks + grp, t64 | (u64)(in_group + total));
            for (uint32_t p0 = 0; p0 < grp; p0 += 64u) {
                const bool has = p0 + (uint32_t)lane < grp;
                u64 x = t64;
                while (true) {
                    if (has) x = granule_poll(em.status + nblocks + p0 + lane, spins);
                    if (__all((x >> 32) == (u64)em.tag)) break;
                    if (spins == SOFT_SPINS && lane == 0) atomicOr(em.err, ERR_HARD_POLL);
                    if (++spins > SPIN_LIMIT) { if (lane == 0) atomicOr(em.err, ERR_TIMEOUT); break; }
                    __builtin_amdgcn_s_sleep(2);
                }
                uint32_t v = has ? (uint32_t)x : 0u;
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o, 64);
                before += v;
            }
            if (lane == 0) s_carry = in_group + before;
        }
        __syncthreads();
        uint32_t o = s_carry + ex;
        if (live) {
            em.off_by_gid[idx] = o;   // first emission slot of the splat: its instances (and their rows of `partial`) are consecutive
            for (int y = by0; y < by1; y++)
                for (int x = bx0; x < bx1; x++) {
                    if (o < em.cap) {   // (a scene that outgrew the speculative capacity is redone with the exact size)
                        const uint32_t t = (uint32_t)(y * tiles_x + x);
                        em.tile_keys[o] = t;
                        em.gids[o] = (uint32_t)idx;
                        for (int q = 0, sh = 0, dw = tile_digit_width(em.tbits); sh < em.tbits; q++, sh += dw) {
                            const int bits = (em.tbits - sh) < dw ? (em.tbits - sh) : dw;
                            atomicAdd(&h[q][(t >> sh) & ((1u << bits) - 1u)], 1u);
                        }
                    }
                    o++;
                }
        }
        if (b == gridDim.x - 1 && tid == 0) {   // the last workgroup owns the grand total
            const uint32_t count = s_carry + total, flags = *em.err;
            em.count[0] = count;
            em.count[1] = flags;
            em.count_out[0] = count;
            em.count_out[1] = flags;
            __hip_atomic_store(em.count_out + 2, em.tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        __syncthreads();
        for (int q = 0, sh = 0, dw = tile_digit_width(em.tbits); sh < em.tbits; q++, sh += dw) {
            const uint32_t c = h[q][tid];
            if (c) atomicAdd(&em.ghist[q * RADIX_SIZE + tid], c);
        }
        return;
    }
    // (arrive == null: the caller does not need the count early — speculative capacity, api.hip — and lets the scan deliver it)
    if (arrive == nullptr) return;
    uint32_t wsum = live ? tiles_out : 0u;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) wsum += (uint32_t)__shfl_xor((int)wsum, o, 64);
    if (__lane_id() == 0 && wsum) atomicAdd(&s_tiles, wsum);   // one LDS atomic per wave
    __syncthreads();
    if (threadIdx.x == 0) {
        // two levels, so that thousands of workgroups do not serialise on one address: 64 sub-counters (workgroup index mod
        // 64), whose last arrivers forward their sub-totals to the top word arrive[0]
        constexpr unsigned long long LOW = (1ull << 40) - 1ull;
        const unsigned r = blockIdx.x & 63u;
        const unsigned long long expected = (gridDim.x - r + 63u) / 64u;
        unsigned long long mine = (1ull << 40) | (unsigned long long)s_tiles;
        unsigned long long *sub = arrive + 8 + 8 * r;   // one 64-byte line per counter
        unsigned long long old = atomicAdd(sub, mine);
        if ((old >> 40) + 1ull == expected) {
            __hip_atomic_store(sub, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm
            mine = (1ull << 40) | ((old + mine) & LOW);
            old = atomicAdd(arrive, mine);
            const unsigned long long subs = gridDim.x < 64u ? gridDim.x : 64u;
            if ((old >> 40) + 1ull == subs) {
                const unsigned long long total = (old + mine) & LOW;
                __hip_atomic_store(arrive, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // re-arm
                host_out[0] = (uint32_t)(total > 0xFFFFFFFFull ? 0xFFFFFFFFull : total);
                __hip_atomic_store(host_out + 2, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

__global__ void __launch_bounds__(256) mark_visible_kernel(int P, const float *__restrict__ means3D,
                                                           const float *__restrict__ viewmatrix, uint8_t *__restrict__ present) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= P) return;
    float V[16];
#pragma unroll
    for (int i = 0; i < 16; i++) V[i] = viewmatrix[i];
    const float3 p = make_float3(means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]);
    present[idx] = xform43(p, V).z > NEAR_PLANE ? 1 : 0;
}

int launch_preprocess(const das3r_raster_args *a, const das3r_raster_in *in, int32_t *radii, char *geom, char *img, char *binning_ctrl,
                      size_t binning_ctrl_bytes, const Layout &L, unsigned long long *arrive, uint32_t *host_out, uint32_t tag,
                      hipStream_t s, const EmitArgs *emit, uint32_t *dhist, uint32_t *dhist_next) {
    const int P = a->P;
    if (P == 0) return DAS3R_OK;
    const EmitArgs em = emit ? *emit : EmitArgs{nullptr, 0u, nullptr, nullptr, nullptr, nullptr, nullptr, 0u, 0, nullptr, nullptr};
    dim3 grid(div_up(P, 256)), block(256);
    uint32_t dhist_mask = 0u;   // sample the depth histogram from <= 512 workgroups spread pseudo-randomly over the grid
    while ((grid.x >> __builtin_popcount(dhist_mask)) > 512u) dhist_mask = (dhist_mask << 1) | 1u;
    const bool has_sh = in->shs != nullptr, has_cov = in->cov3D_precomp != nullptr;
    const PreXform pre = pre_xform(in);
#define ARGS                                                                                                              \
    P, a->sh_degree, a->M, in->means3D, in->scales, a->scale_modifier, in->rotations, in->opacities, in->shs,             \
        in->cov3D_precomp, in->colors_precomp, a->viewmatrix, a->projmatrix, a->campos, a->image_width, a->image_height,  \
        a->tanfovx, a->tanfovy, L.tiles_x, L.tiles_y, radii, (uint32_t *)(geom + L.g_keyA), (float4 *)(geom + L.pub.xy),  \
        (float4 *)(geom + L.pub.conic_opacity), (float4 *)(geom + L.pub.rgbd), (uint8_t *)(geom + L.pub.clamped),         \
        (uint32_t *)(geom + L.pub.tiles_touched), (uint32_t *)(geom + L.g_rect), (use_tight_rect() ? 1 : 0) | (a->prefiltered ? 2 : 0), (uint32_t *)(geom + L.g_ghist), (uint32_t)(L.g_ctrl_bytes / 4),                 \
        (uint32_t *)(img + L.pub.ranges), (uint32_t)(2 * L.ntiles), (uint32_t *)binning_ctrl, (uint32_t)(binning_ctrl_bytes / 4), arrive, host_out, tag, em, dhist, dhist_mask, dhist_next, pre
    const bool stage = has_sh && a->M == 16 && a->sh_degree >= 2 && ((uintptr_t)in->shs & 15) == 0 && !switches().no_sh_stage;
    if (has_sh && !has_cov && stage) DAS3R_LAUNCH((preprocess_kernel<true, false, true>), grid, block, 0, s, ARGS);
    else if (has_sh && has_cov && stage) DAS3R_LAUNCH((preprocess_kernel<true, true, true>), grid, block, 0, s, ARGS);
    else if (has_sh && !has_cov) DAS3R_LAUNCH((preprocess_kernel<true, false, false>), grid, block, 0, s, ARGS);
    else if (has_sh && has_cov) DAS3R_LAUNCH((preprocess_kernel<true, true, false>), grid, block, 0, s, ARGS);
    else if (!has_sh && !has_cov) DAS3R_LAUNCH((preprocess_kernel<false, false, false>), grid, block, 0, s, ARGS);
    else DAS3R_LAUNCH((preprocess_kernel<false, true, false>), grid, block, 0, s, ARGS);
#undef ARGS
    KERNEL_CHECK(s, a->debug, "preprocess");
    return DAS3R_OK;
}

int launch_mark_visible(int P, const float *means3D, const float *viewmatrix, uint8_t *present, hipStream_t s) {
    if (P == 0) return DAS3R_OK;
    DAS3R_LAUNCH(mark_visible_kernel, dim3(div_up(P, 256)), dim3(256), 0, s, P, means3D, viewmatrix, present);
    KERNEL_CHECK(s, false, "mark_visible");
    return DAS3R_OK;
}

}  // namespace das3r
