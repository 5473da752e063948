// render_common.h — per-(pixel, splat) arithmetic shared by the forward (K6) and backward (K7) compositing
// kernels, so both evaluate bit-identical alpha for the same pair.
#pragma once
#include "common.h"

namespace das3r {

struct StagedSplat {  // one LDS-staged entry of a tile's splat list
    float2 xy;
    float4 co;   // conic A,B,C + opacity
    float4 rgbd; // r,g,b,(depth)
};

// XCD-aware block -> tile map: the dispatcher places block b on XCD b % 8 (MI355X_MICROARCH.md); give every XCD
// one contiguous band of tiles so that neighbouring tiles (which share splats) hit the same 4 MiB L2.
__device__ __forceinline__ int xcd_tile(const int block, const int ntiles) {
    const int per = (ntiles + 7) >> 3;
    const int t = (block & 7) * per + (block >> 3);
    return t < ntiles ? t : -1;
}
static inline int xcd_grid(int ntiles) { return ((ntiles + 7) / 8) * 8; }

// Gaussian falloff of splat at pixel (px,py).  Returns false when the pair is skipped
// (power > 0 or alpha < 1/255).  Explicit fma placement => same rounding in every kernel that inlines this.
__device__ __forceinline__ bool pair_alpha(const float2 xy, const float4 co, const float px, const float py, float &dx, float &dy,
                                           float &G, float &alpha) {
    dx = xy.x - px;
    dy = xy.y - py;
    const float q = __fmaf_rn(__fmul_rn(co.x, dx), dx, __fmul_rn(__fmul_rn(co.z, dy), dy));
    const float power = __fmaf_rn(-0.5f, q, -__fmul_rn(__fmul_rn(co.y, dx), dy));
    if (power > 0.0f) return false;
    G = __expf(power);
    alpha = fminf(0.99f, __fmul_rn(co.w, G));
    return alpha >= (1.0f / 255.0f);
}

}  // namespace das3r
