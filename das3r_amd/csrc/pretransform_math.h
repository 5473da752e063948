// pretransform_math.h — the pose pre-transform of a Gaussian (/root/reference/gaussian_renderer/__init__.py:83-97,107)
// as ONE piece of arithmetic, shared by pretransform_forward_kernel (pretransform.hip: writes the camera-frame tensors) and by the
// per-Gaussian kernels of the rasterizer when the caller hands them the raw parameters instead (das3r_raster_in.pre, round 6: the
// camera-frame means / rotations / scales / opacities then never reach HBM — preprocess.hip, preprocess_bwd.hip).  Every operation is spelled
// out (explicit fused multiply-adds, rounded products): the translation units that include this are compiled with different contraction
// flags, and the three kernels must produce the same bits.
#pragma once
#include <stdint.h>

namespace das3r {

struct PreXform {   // device-side view of include/das3r_raster.h das3r_pretransform (xyz == nullptr: not in use)
    const float *xyz, *rot, *scaling, *opacity_raw, *conf_flat;
    const int64_t *mask_index;
    const float *Rm, *tv, *Lq;
};

#ifdef __HIPCC__
struct PoseRegs { float R[9], t[3], L[16]; };
__device__ __forceinline__ void load_pose(const float *__restrict__ Rm, const float *__restrict__ tv, const float *__restrict__ Lq, PoseRegs &p) {
#pragma unroll
    for (int i = 0; i < 9; i++) p.R[i] = Rm[i];
#pragma unroll
    for (int i = 0; i < 3; i++) p.t[i] = tv[i];
#pragma unroll
    for (int i = 0; i < 16; i++) p.L[i] = Lq[i];
}
__device__ __forceinline__ float pre_dot3(const float a, const float b, const float c, const float x, const float y, const float z, const float t) {
    return __fadd_rn(__fmaf_rn(c, z, __fmaf_rn(b, y, __fmul_rn(a, x))), t);
}
__device__ __forceinline__ float pre_dot4(const float a, const float b, const float c, const float d, const float4 q) {
    return __fmaf_rn(d, q.w, __fmaf_rn(c, q.z, __fmaf_rn(b, q.y, __fmul_rn(a, q.x))));
}
__device__ __forceinline__ float3 pre_mean(const PoseRegs &p, const float x, const float y, const float z) {   // R xyz + t
    return make_float3(pre_dot3(p.R[0], p.R[1], p.R[2], x, y, z, p.t[0]), pre_dot3(p.R[3], p.R[4], p.R[5], x, y, z, p.t[1]),
                       pre_dot3(p.R[6], p.R[7], p.R[8], x, y, z, p.t[2]));
}
__device__ __forceinline__ float4 pre_rot(const PoseRegs &p, const float4 q) {   // Lq rot (quaternion product with the pose: linear in rot)
    return make_float4(pre_dot4(p.L[0], p.L[1], p.L[2], p.L[3], q), pre_dot4(p.L[4], p.L[5], p.L[6], p.L[7], q),
                       pre_dot4(p.L[8], p.L[9], p.L[10], p.L[11], q), pre_dot4(p.L[12], p.L[13], p.L[14], p.L[15], q));
}
__device__ __forceinline__ float pre_scale(const float s) { return expf(s); }
__device__ __forceinline__ float pre_opacity(const float raw, const float conf) {   // sigmoid(raw) conf
    const float s = __fdiv_rn(1.0f, __fadd_rn(1.0f, expf(-raw)));
    return __fmul_rn(s, conf);
}
#endif

}  // namespace das3r
