// pretransform_chain.h — the BACKWARD of the pose pre-transform for one Gaussian (chain rule, the Adam step of the four geometry tensors,
// the camera's 28 sums), as ONE piece of arithmetic shared by pretransform_backward_kernel (pretransform.hip: reads the camera-frame
// gradients the rasterizer's backward wrote) and by preprocess_backward_kernel's CHAIN variant (preprocess_bwd.hip, round 6: takes them
// straight out of its registers — dL/d(camera-frame means, rotations, scales, opacities) never reach HBM).  Every operation is spelled out:
// the two translation units are compiled with different contraction flags and must move every parameter by the same bits.
// Replaces the PyTorch glue of /root/reference/gaussian_renderer/__init__.py:83-97,107 (backward) and the optimizer step of
// /root/reference/scene/gaussian_model.py:236-261 for xyz / rotation / scaling / opacity.
#pragma once
#include "adam_math.h"
#include "pretransform_math.h"

namespace das3r {

// The four per-Gaussian parameter tensors the pre-transform reads, as Adam sees them: parameter, first and second moment,
// step_size = lr / (1 - beta1^t) and sqrt(1 - beta2^t) of the group each belongs to.
struct GeometryAdam {
    float *p[4], *m[4], *v[4];   // xyz [P,3], rotation [P,4], scaling [P,3], opacity [P,1]
    float step_size[4], bc2_sqrt[4];
    float beta1, beta2, eps;
};

constexpr int POSE_MAX_BLOCKS = 2048;   // largest grid of pretransform.hip's backward kernels (rows of their fixed-order pose-sum scratch)

#ifdef __HIPCC__
struct ChainIn {    // one Gaussian: raw parameters, its confidence, the gradients of its camera-frame quantities
    float x, y, z;
    float4 q;
    float sc[3], o, c;
    float gx, gy, gz;
    float4 gq;
    float gs[3], go;
};
struct ChainOut {   // dL/d(xyz, rotation, scaling, opacity logit), dL/d(confidence)
    float rx[3];
    float4 rq;
    float rs[3], ro, gconf;
};

// a b as a rounded fp32 number, whatever consumes it: a gradient is the same number whether it is written to memory or handed to the Adam
// step in a register, where the compiler would otherwise contract the product into the step's first subtraction (__fmul_rn is a plain
// multiplication to this compiler, contraction included)
__device__ __forceinline__ float rounded_product(const float a, const float b) {
    float r = a * b;
    asm volatile("" : "+v"(r));
    return r;
}
__device__ __forceinline__ float chain_dot3t(const float a, const float b, const float c, const float x, const float y, const float z) {
    return __fmaf_rn(c, z, __fmaf_rn(b, y, __fmul_rn(a, x)));
}
__device__ __forceinline__ float chain_dot4t(const float a, const float b, const float c, const float d, const float4 g) {
    return __fmaf_rn(d, g.w, __fmaf_rn(c, g.z, __fmaf_rn(b, g.y, __fmul_rn(a, g.x))));
}
// R [9], L [16]: row-major pose matrices.  (geometry = false: the caller only wants the camera's sums — MODE 2 of pretransform.hip)
__device__ __forceinline__ void chain_grads(const float *R, const float *L, const ChainIn &in, ChainOut &o) {
    o.rx[0] = chain_dot3t(R[0], R[3], R[6], in.gx, in.gy, in.gz);   // R^T g
    o.rx[1] = chain_dot3t(R[1], R[4], R[7], in.gx, in.gy, in.gz);
    o.rx[2] = chain_dot3t(R[2], R[5], R[8], in.gx, in.gy, in.gz);
    o.rq = make_float4(chain_dot4t(L[0], L[4], L[8], L[12], in.gq), chain_dot4t(L[1], L[5], L[9], L[13], in.gq),     // Lq^T g
                       chain_dot4t(L[2], L[6], L[10], L[14], in.gq), chain_dot4t(L[3], L[7], L[11], L[15], in.gq));
#pragma unroll
    for (int k = 0; k < 3; k++) o.rs[k] = rounded_product(in.gs[k], expf(in.sc[k]));
    const float s = __fdiv_rn(1.0f, __fadd_rn(1.0f, expf(-in.o)));
    o.ro = rounded_product(rounded_product(rounded_product(in.go, in.c), s), __fsub_rn(1.0f, s));
    o.gconf = rounded_product(in.go, s);
}
// acc[28] += this Gaussian's share of dL/dR (9, row-major), dL/dt (3), dL/dLq (16, row-major)
__device__ __forceinline__ void chain_pose_acc(const ChainIn &in, float *acc) {
    acc[0] = __fmaf_rn(in.gx, in.x, acc[0]); acc[1] = __fmaf_rn(in.gx, in.y, acc[1]); acc[2] = __fmaf_rn(in.gx, in.z, acc[2]);
    acc[3] = __fmaf_rn(in.gy, in.x, acc[3]); acc[4] = __fmaf_rn(in.gy, in.y, acc[4]); acc[5] = __fmaf_rn(in.gy, in.z, acc[5]);
    acc[6] = __fmaf_rn(in.gz, in.x, acc[6]); acc[7] = __fmaf_rn(in.gz, in.y, acc[7]); acc[8] = __fmaf_rn(in.gz, in.z, acc[8]);
    acc[9] = __fadd_rn(acc[9], in.gx); acc[10] = __fadd_rn(acc[10], in.gy); acc[11] = __fadd_rn(acc[11], in.gz);
    const float gv[4] = {in.gq.x, in.gq.y, in.gq.z, in.gq.w}, qv[4] = {in.q.x, in.q.y, in.q.z, in.q.w};
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) acc[12 + 4 * a + b] = __fmaf_rn(gv[a], qv[b], acc[12 + 4 * a + b]);
}
// the moments of Gaussian i's rows, requested ahead of their use (preprocess_bwd.hip CHAIN asks for them with its other inputs: a lane's whole
// backward lies between the request and the step — read at the step they were one more exposed trip to memory at the end of every wave)
struct ChainMoments {
    float m0[3], v0[3], m2[3], v2[3], m3, v3;
    float4 m1, v1;
};
__device__ __forceinline__ void chain_load_moments(const GeometryAdam &A, const size_t i, ChainMoments &mo) {
#pragma unroll
    for (int k = 0; k < 3; k++) {
        mo.m0[k] = A.m[0][3 * i + k]; mo.v0[k] = A.v[0][3 * i + k];
        mo.m2[k] = A.m[2][3 * i + k]; mo.v2[k] = A.v[2][3 * i + k];
    }
    mo.m1 = reinterpret_cast<const float4 *>(A.m[1])[i];
    mo.v1 = reinterpret_cast<const float4 *>(A.v[1])[i];
    mo.m3 = A.m[3][i];
    mo.v3 = A.v[3][i];
}
// the Adam step of the four tensors' rows of Gaussian i (adam_math.h), parameters read from `in`, moments from `mo`
__device__ __forceinline__ void chain_adam(const GeometryAdam &A, const size_t i, const ChainIn &in, const ChainOut &o, const ChainMoments &mo) {
    float pv[3] = {in.x, in.y, in.z};
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float m = mo.m0[k], v = mo.v0[k];
        adam_update(pv[k], m, v, o.rx[k], A.beta1, A.beta2, A.eps, A.step_size[0], A.bc2_sqrt[0]);
        A.p[0][3 * i + k] = pv[k];
        A.m[0][3 * i + k] = m;
        A.v[0][3 * i + k] = v;
    }
    float4 qm = mo.m1, qvv = mo.v1, qp = in.q;
    adam_update(qp.x, qm.x, qvv.x, o.rq.x, A.beta1, A.beta2, A.eps, A.step_size[1], A.bc2_sqrt[1]);
    adam_update(qp.y, qm.y, qvv.y, o.rq.y, A.beta1, A.beta2, A.eps, A.step_size[1], A.bc2_sqrt[1]);
    adam_update(qp.z, qm.z, qvv.z, o.rq.z, A.beta1, A.beta2, A.eps, A.step_size[1], A.bc2_sqrt[1]);
    adam_update(qp.w, qm.w, qvv.w, o.rq.w, A.beta1, A.beta2, A.eps, A.step_size[1], A.bc2_sqrt[1]);
    reinterpret_cast<float4 *>(A.p[1])[i] = qp;
    reinterpret_cast<float4 *>(A.m[1])[i] = qm;
    reinterpret_cast<float4 *>(A.v[1])[i] = qvv;
    float sc[3] = {in.sc[0], in.sc[1], in.sc[2]};
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float m = mo.m2[k], v = mo.v2[k];
        adam_update(sc[k], m, v, o.rs[k], A.beta1, A.beta2, A.eps, A.step_size[2], A.bc2_sqrt[2]);
        A.p[2][3 * i + k] = sc[k];
        A.m[2][3 * i + k] = m;
        A.v[2][3 * i + k] = v;
    }
    float op = in.o, m = mo.m3, v = mo.v3;
    adam_update(op, m, v, o.ro, A.beta1, A.beta2, A.eps, A.step_size[3], A.bc2_sqrt[3]);
    A.p[3][i] = op;
    A.m[3][i] = m;
    A.v[3][i] = v;
}
__device__ __forceinline__ void chain_adam(const GeometryAdam &A, const size_t i, const ChainIn &in, const ChainOut &o) {
    ChainMoments mo;
    chain_load_moments(A, i, mo);
    chain_adam(A, i, in, o, mo);
}

// The 28 sums over all Gaussians of a launch, from every lane's acc[28]: wave reduction on the DPP network, the workgroup's four waves through
// LDS (red), then — det_partials == nullptr — one float atomic per workgroup and sum, or (round 5) in a FIXED order: every workgroup stores
// its 28 partial sums in row blockIdx.x of det_partials, the last one to arrive (one integer atomic per workgroup on *arrived) adds the rows
// in index order — thread t takes rows t, t + 256, ..., the 256 threads meet in the same fixed tree — and adds the totals to g_small.
// Bit-identical from run to run.  Workgroups of 256 threads; every thread of the workgroup must call it.
__device__ __forceinline__ void pose_sums_finish(const float *acc, float (*red)[28], float *__restrict__ det_partials, uint32_t *__restrict__ arrived,
                                                 float *__restrict__ g_small) {
    const int lane = __lane_id(), wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 28; i++) {
        const float r = wave_sum_to_lane63(acc[i]);
        if (lane == 63) red[wave][i] = r;
    }
    __syncthreads();
    if (det_partials == nullptr) {
        if (threadIdx.x < 28) {
            const float r = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
            if (r != 0.f) unsafeAtomicAdd(&g_small[threadIdx.x], r);
        }
        return;
    }
    __shared__ uint32_t s_last;
    if (threadIdx.x < 28)
        __hip_atomic_store(det_partials + (size_t)blockIdx.x * 28 + threadIdx.x,
                           red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // (no __threadfence here: an agent-scope release fence writes the XCD's whole L2 back — these kernels have just written > 100 MB of
    //  parameters and moments, and 2048 workgroups doing that took the kernel from 0.15 to 0.60 ms.  The 28 words are agent-scope atomic
    //  stores — written through to where the other XCDs see them — and are complete when the store counter says so; the barrier then
    //  orders them in front of thread 0's arrival, itself a relaxed agent-scope atomic)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__) && !defined(__gfx90a__)
#error "the pose-sum hand-off counts its stores in vmcnt (gfx9: one counter for loads and stores); gfx10+ counts stores in vscnt — use a release-ordered arrival atomic there"
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) s_last = (__hip_atomic_fetch_add(arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u) ? 1u : 0u;
    __syncthreads();
    if (!s_last) return;   // (uniform)
    // (every wave of the last workgroup acquires at agent scope — the other workgroups' rows came through other XCDs' L2s — and then reads
    //  its rows with plain 16-byte loads, all in flight at once: read one atomic word at a time the 224 loads of a thread were 0.45 ms)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    float part[28];
#pragma unroll
    for (int i = 0; i < 28; i++) part[i] = 0.f;
    const float4 *const rows4 = reinterpret_cast<const float4 *>(det_partials);   // a row = 28 floats = 7 float4
    for (uint32_t b0 = threadIdx.x; b0 < gridDim.x; b0 += 1024u) {
        float4 v[4][7];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t b = b0 + 256u * u;
            const uint32_t bc = b < gridDim.x ? b : gridDim.x - 1u;   // (clamped, not guarded: the loads stay in one block, all in flight)
#pragma unroll
            for (int q = 0; q < 7; q++) v[u][q] = rows4[(size_t)bc * 7 + q];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const bool have = (b0 + 256u * u) < gridDim.x;   // (a select, not a factor: a row may hold an inf)
#pragma unroll
            for (int q = 0; q < 7; q++) {
                part[4 * q] += have ? v[u][q].x : 0.f; part[4 * q + 1] += have ? v[u][q].y : 0.f;
                part[4 * q + 2] += have ? v[u][q].z : 0.f; part[4 * q + 3] += have ? v[u][q].w : 0.f;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 28; i++) {   // the same fixed tree as above: DPP within the wave, the four waves through LDS
        const float r = wave_sum_to_lane63(part[i]);
        if (lane == 63) red[wave][i] = r;
    }
    __syncthreads();
    if (threadIdx.x < 28)   // (g_small accumulates: zero at rest, das3r_pose_chain_qt re-arms it)
        g_small[threadIdx.x] += red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    if (threadIdx.x == 0) __hip_atomic_store(arrived, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-armed for the next launch
}
#endif

}  // namespace das3r
