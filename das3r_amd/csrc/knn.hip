// knn.hip — distCUDA2: mean squared distance to the 3 nearest other points (exact).  Replaces
// upstream:simple-knn/simple_knn.cu (SimpleKNN::knn) for /root/reference/scene/gaussian_model.py:213,641.
//
// Pipeline: AABB -> 30-bit Morton codes -> stable radix sort (shared with the rasterizer's binning) -> gather points
// into Morton order -> AABB of every run of 1024 sorted points ("box") -> exact 3-NN search with box pruning.
// MI355X design for the search: a 256-lane workgroup owns 256 CONSECUTIVE Morton-sorted points (spatially compact), so
// box pruning is first done once per workgroup against the workgroup's own AABB (uniform branch), and a surviving
// box's 1024 points are staged through LDS with coalesced float4 loads and then read at wave-uniform addresses
// (LDS broadcast) by all lanes.  Compiled with -ffp-contract=off: squared distances are evaluated exactly as
// dx*dx + dy*dy + dz*dz, so the output is bit-identical to an exhaustive fp32 scan.
#include <float.h>

#include "common.h"

namespace das3r {

int radix_sort_u32_pairs(const uint32_t *keys_in, uint32_t *keyA, uint32_t *keyB, uint32_t *valA, uint32_t *valB, int64_t n,
                         int total_bits, uint32_t *hist, uint32_t *totals, uint32_t **vals_final, hipStream_t s);

constexpr int KNN_BOX = 1024;

struct KnnLayout {
    size_t aabb_partial, aabb, keyA, keyB, valA, valB, hist, totals, spts, boxes, bytes;
    int nblocks_red, nboxes;
};
static void knn_layout(int P, KnnLayout *L) {
    size_t o = 0;
    auto take = [&](size_t b) { size_t r = o; o += align_up(b); return r; };
    L->nblocks_red = div_up(P, 1024);
    L->nboxes = div_up(P, KNN_BOX);
    L->aabb_partial = take(sizeof(float) * 6 * (size_t)L->nblocks_red);
    L->aabb = take(sizeof(float) * 8);
    L->keyA = take(4 * (size_t)P);
    L->keyB = take(4 * (size_t)P);
    L->valA = take(4 * (size_t)P);
    L->valB = take(4 * (size_t)P);
    L->hist = take(4 * (size_t)RADIX_SIZE * (size_t)sort_num_chunks(P));
    L->totals = take(4 * RADIX_SIZE);
    L->spts = take(16 * (size_t)P);
    L->boxes = take(sizeof(float) * 8 * (size_t)L->nboxes);
    L->bytes = o;
}

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// block-wide min/max of 3 coordinates; result in out[0..5] (min xyz, max xyz) by thread 0
__device__ __forceinline__ void block_minmax3(float3 mn, float3 mx, float *lds /*[4][6]*/, float out[6]) {
    const int lane = __lane_id(), wave = threadIdx.x >> 6;
    float v[6] = {wave_min(mn.x), wave_min(mn.y), wave_min(mn.z), wave_max(mx.x), wave_max(mx.y), wave_max(mx.z)};
    __syncthreads();
    if (lane == 0)
        for (int k = 0; k < 6; k++) lds[wave * 6 + k] = v[k];
    __syncthreads();
    const int nw = blockDim.x >> 6;
    for (int k = 0; k < 3; k++) {
        float a = lds[k], b = lds[3 + k];
        for (int w = 1; w < nw; w++) {
            a = fminf(a, lds[w * 6 + k]);
            b = fmaxf(b, lds[w * 6 + 3 + k]);
        }
        out[k] = a;
        out[3 + k] = b;
    }
}

__global__ void __launch_bounds__(256) knn_aabb_partial_kernel(int P, const float *__restrict__ pts, float *__restrict__ partial) {
    __shared__ float lds[4 * 6];
    float3 mn = make_float3(FLT_MAX, FLT_MAX, FLT_MAX), mx = make_float3(-FLT_MAX, -FLT_MAX, -FLT_MAX);
    for (int k = 0; k < 4; k++) {
        const int i = blockIdx.x * 1024 + k * 256 + threadIdx.x;
        if (i < P) {
            const float x = pts[3 * (size_t)i], y = pts[3 * (size_t)i + 1], z = pts[3 * (size_t)i + 2];
            mn = make_float3(fminf(mn.x, x), fminf(mn.y, y), fminf(mn.z, z));
            mx = make_float3(fmaxf(mx.x, x), fmaxf(mx.y, y), fmaxf(mx.z, z));
        }
    }
    float out[6];
    block_minmax3(mn, mx, lds, out);
    if (threadIdx.x == 0)
        for (int k = 0; k < 6; k++) partial[blockIdx.x * 6 + k] = out[k];
}

__global__ void __launch_bounds__(256) knn_aabb_final_kernel(int n, const float *__restrict__ partial, float *__restrict__ aabb) {
    __shared__ float lds[4 * 6];
    float3 mn = make_float3(FLT_MAX, FLT_MAX, FLT_MAX), mx = make_float3(-FLT_MAX, -FLT_MAX, -FLT_MAX);
    for (int i = threadIdx.x; i < n; i += 256) {
        mn = make_float3(fminf(mn.x, partial[6 * i]), fminf(mn.y, partial[6 * i + 1]), fminf(mn.z, partial[6 * i + 2]));
        mx = make_float3(fmaxf(mx.x, partial[6 * i + 3]), fmaxf(mx.y, partial[6 * i + 4]), fmaxf(mx.z, partial[6 * i + 5]));
    }
    float out[6];
    block_minmax3(mn, mx, lds, out);
    if (threadIdx.x == 0)
        for (int k = 0; k < 6; k++) aabb[k] = out[k];
}

__device__ __forceinline__ uint32_t spread10(uint32_t x) {  // 10 bits -> every third bit
    x = (x | (x << 16)) & 0x030000FF;
    x = (x | (x << 8)) & 0x0300F00F;
    x = (x | (x << 4)) & 0x030C30C3;
    x = (x | (x << 2)) & 0x09249249;
    return x;
}

__global__ void __launch_bounds__(256) knn_morton_kernel(int P, const float *__restrict__ pts, const float *__restrict__ aabb,
                                                         uint32_t *__restrict__ codes) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    float c[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float lo = aabb[k], hi = aabb[3 + k];
        const float ext = hi - lo;
        float u = ext > 0.f ? (pts[3 * (size_t)i + k] - lo) / ext : 0.f;
        u = fminf(fmaxf(u, 0.f), 1.f);
        c[k] = u * 1023.f;
    }
    codes[i] = spread10((uint32_t)c[0]) | (spread10((uint32_t)c[1]) << 1) | (spread10((uint32_t)c[2]) << 2);
}

__global__ void __launch_bounds__(256) knn_gather_kernel(int P, const float *__restrict__ pts, const uint32_t *__restrict__ sorted_idx,
                                                         float4 *__restrict__ spts) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const uint32_t g = sorted_idx[i];
    spts[i] = make_float4(pts[3 * (size_t)g], pts[3 * (size_t)g + 1], pts[3 * (size_t)g + 2], 0.f);
}

__global__ void __launch_bounds__(256) knn_box_kernel(int P, const float4 *__restrict__ spts, float *__restrict__ boxes) {
    __shared__ float lds[4 * 6];
    float3 mn = make_float3(FLT_MAX, FLT_MAX, FLT_MAX), mx = make_float3(-FLT_MAX, -FLT_MAX, -FLT_MAX);
    for (int k = 0; k < KNN_BOX / 256; k++) {
        const int i = blockIdx.x * KNN_BOX + k * 256 + threadIdx.x;
        if (i < P) {
            const float4 p = spts[i];
            mn = make_float3(fminf(mn.x, p.x), fminf(mn.y, p.y), fminf(mn.z, p.z));
            mx = make_float3(fmaxf(mx.x, p.x), fmaxf(mx.y, p.y), fmaxf(mx.z, p.z));
        }
    }
    float out[6];
    block_minmax3(mn, mx, lds, out);
    if (threadIdx.x == 0)
        for (int k = 0; k < 6; k++) boxes[blockIdx.x * 8 + k] = out[k];
}

__device__ __forceinline__ void update_kbest3(float d, float &b0, float &b1, float &b2) {
    if (b0 > d) { const float t = b0; b0 = d; d = t; }
    if (b1 > d) { const float t = b1; b1 = d; d = t; }
    if (b2 > d) { b2 = d; }
}
__device__ __forceinline__ float dist2(const float4 a, const float4 b) {
    const float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
    return dx * dx + dy * dy + dz * dz;
}
// squared distance from p to an AABB (0 inside); never larger than dist2(p, q) for any q inside the box
__device__ __forceinline__ float dist2_box_point(const float *bmin, const float *bmax, const float4 p) {
    float dx = 0.f, dy = 0.f, dz = 0.f;
    if (p.x < bmin[0] || p.x > bmax[0]) dx = fminf(fabsf(p.x - bmin[0]), fabsf(p.x - bmax[0]));
    if (p.y < bmin[1] || p.y > bmax[1]) dy = fminf(fabsf(p.y - bmin[1]), fabsf(p.y - bmax[1]));
    if (p.z < bmin[2] || p.z > bmax[2]) dz = fminf(fabsf(p.z - bmin[2]), fabsf(p.z - bmax[2]));
    return dx * dx + dy * dy + dz * dz;
}
__device__ __forceinline__ float dist2_box_box(const float *amin, const float *amax, const float *bmin, const float *bmax) {
    float g[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        g[k] = 0.f;
        if (amax[k] < bmin[k]) g[k] = bmin[k] - amax[k];
        else if (bmax[k] < amin[k]) g[k] = amin[k] - bmax[k];
    }
    return g[0] * g[0] + g[1] * g[1] + g[2] * g[2];
}

__global__ void __launch_bounds__(256) knn_search_kernel(int P, int nboxes, const float4 *__restrict__ spts,
                                                         const uint32_t *__restrict__ sorted_idx, const float *__restrict__ boxes,
                                                         float *__restrict__ out) {
    __shared__ float4 tile[KNN_BOX];
    __shared__ float lds[4 * 6];
    __shared__ float s_rej[4];
    const int pos = blockIdx.x * 256 + threadIdx.x;
    const bool valid = pos < P;
    const float4 p = valid ? spts[pos] : make_float4(0.f, 0.f, 0.f, 0.f);

    // bound from the +-3 neighbours in Morton order (upstream boxMeanDist prologue)
    float b0 = FLT_MAX, b1 = FLT_MAX, b2 = FLT_MAX;
    if (valid) {
        const int lo = max(0, pos - 3), hi = min(P - 1, pos + 3);
        for (int i = lo; i <= hi; i++)
            if (i != pos) update_kbest3(dist2(p, spts[i]), b0, b1, b2);
    }
    const float reject = b2;
    b0 = b1 = b2 = FLT_MAX;

    // workgroup AABB and workgroup-wide reject radius
    float bb[6];
    {
        const float3 mn = valid ? make_float3(p.x, p.y, p.z) : make_float3(FLT_MAX, FLT_MAX, FLT_MAX);
        const float3 mx = valid ? make_float3(p.x, p.y, p.z) : make_float3(-FLT_MAX, -FLT_MAX, -FLT_MAX);
        block_minmax3(mn, mx, lds, bb);
    }
    float wrej = wave_max(valid ? reject : 0.f);
    if (__lane_id() == 0) s_rej[threadIdx.x >> 6] = wrej;
    __syncthreads();
    const float block_reject = fmaxf(fmaxf(s_rej[0], s_rej[1]), fmaxf(s_rej[2], s_rej[3]));

    for (int b = 0; b < nboxes; b++) {
        const float *bmin = boxes + 8 * (size_t)b, *bmax = bmin + 3;
        if (dist2_box_box(bb, bb + 3, bmin, bmax) > block_reject) continue;  // uniform
        bool want = false;
        if (valid) {
            const float d = dist2_box_point(bmin, bmax, p);
            want = !(d > reject || d > b2);
        }
        if (!__syncthreads_or(want)) continue;
        const int start = b * KNN_BOX, cnt = min(KNN_BOX, P - start);
        for (int i = threadIdx.x; i < cnt; i += 256) tile[i] = spts[start + i];
        __syncthreads();
        if (want) {
            const int self = pos - start;  // index of this point inside the tile, if it is there
            for (int i = 0; i < cnt; i++) {
                if (i == self) continue;
                update_kbest3(dist2(p, tile[i]), b0, b1, b2);
            }
        }
        __syncthreads();
    }
    if (valid) out[sorted_idx[pos]] = (b0 + b1 + b2) / 3.0f;
}

}  // namespace das3r

extern "C" size_t das3r_knn3_workspace_bytes(int32_t P) {
    if (P <= 0) return 256;
    das3r::KnnLayout L;
    das3r::knn_layout(P, &L);
    return L.bytes;
}

extern "C" int das3r_knn3_mean_dist2(int32_t P, const float *points, float *out, char *ws, das3r_stream_t stream) {
    using namespace das3r;
    hipStream_t s = (hipStream_t)stream;
    if (P < 0 || (P > 0 && (!points || !out || !ws))) {
        set_error("das3r_knn3_mean_dist2: invalid argument");
        return DAS3R_ERR_INVALID_ARG;
    }
    if (P == 0) return DAS3R_OK;
    KnnLayout L;
    knn_layout(P, &L);
    float *partial = (float *)(ws + L.aabb_partial), *aabb = (float *)(ws + L.aabb);
    DAS3R_LAUNCH(knn_aabb_partial_kernel, dim3(L.nblocks_red), dim3(256), 0, s, P, points, partial);
    KERNEL_CHECK(s, false, "knn_aabb_partial");
    DAS3R_LAUNCH(knn_aabb_final_kernel, dim3(1), dim3(256), 0, s, L.nblocks_red, partial, aabb);
    KERNEL_CHECK(s, false, "knn_aabb_final");
    uint32_t *keyA = (uint32_t *)(ws + L.keyA), *keyB = (uint32_t *)(ws + L.keyB);
    uint32_t *valA = (uint32_t *)(ws + L.valA), *valB = (uint32_t *)(ws + L.valB);
    DAS3R_LAUNCH(knn_morton_kernel, dim3(div_up(P, 256)), dim3(256), 0, s, P, points, aabb, keyA);
    KERNEL_CHECK(s, false, "knn_morton");
    uint32_t *sorted_idx = nullptr;
    int rc = radix_sort_u32_pairs(keyA, keyA, keyB, valA, valB, P, 30, (uint32_t *)(ws + L.hist), (uint32_t *)(ws + L.totals),
                                  &sorted_idx, s);
    if (rc) return rc;
    float4 *spts = (float4 *)(ws + L.spts);
    DAS3R_LAUNCH(knn_gather_kernel, dim3(div_up(P, 256)), dim3(256), 0, s, P, points, sorted_idx, spts);
    KERNEL_CHECK(s, false, "knn_gather");
    float *boxes = (float *)(ws + L.boxes);
    DAS3R_LAUNCH(knn_box_kernel, dim3(L.nboxes), dim3(256), 0, s, P, spts, boxes);
    KERNEL_CHECK(s, false, "knn_box");
    DAS3R_LAUNCH(knn_search_kernel, dim3(div_up(P, 256)), dim3(256), 0, s, P, L.nboxes, spts, sorted_idx, boxes, out);
    KERNEL_CHECK(s, false, "knn_search");
    return DAS3R_OK;
}
