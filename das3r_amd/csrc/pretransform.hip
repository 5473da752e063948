// pretransform.hip — SURVEY.md §8(f)-1: the per-Gaussian pre-transform + activations that DAS3R's render() performs in
// ~15 PyTorch kernels before every rasterizer call (/root/reference/gaussian_renderer/__init__.py:83-97,107):
//     means3D   = R xyz + t                      (rel_w2c = get_camera_from_tensor(pose), :83-90)
//     rotations = Lq rot                         (quadmultiply(pose[:4], _rotation) is linear in rot: 4x4 matrix Lq, :91)
//     scales    = exp(_scaling)                  (:107, scene/gaussian_model.py:32)
//     opacities = sigmoid(_opacity) * conf[idx]  (:95-97; idx = positions of aggregated_mask)
// fused into one streaming pass, with a hand-written backward.  The 3x3 / 3 / 4x4 camera matrices stay tiny PyTorch
// tensors (built from the 7-vector pose by autograd-tracked torch code), so the kernel only returns dL/dR, dL/dt, dL/dLq
// (28 sums over all splats: DPP wave reduction -> LDS -> one global atomic per workgroup and component).
// HBM-bound: 44 B in + 44 B out per splat forward, 88 B in + 44 B out backward.  Opt-in (das3r_render(fused=True)); the
// default path of an unmodified DAS3R checkout is untouched.
#include "common.h"

namespace das3r {

__global__ void __launch_bounds__(256) pretransform_forward_kernel(int P, const float *__restrict__ xyz, const float *__restrict__ rot,
                                                                  const float *__restrict__ scaling, const float *__restrict__ opacity_raw,
                                                                  const float *__restrict__ conf_flat, const int64_t *__restrict__ mask_index,
                                                                  const float *__restrict__ Rm, const float *__restrict__ tv,
                                                                  const float *__restrict__ Lq, float *__restrict__ means3D,
                                                                  float *__restrict__ rotations, float *__restrict__ scales,
                                                                  float *__restrict__ opacities) {
    float R[9], t[3], L[16];
#pragma unroll
    for (int i = 0; i < 9; i++) R[i] = Rm[i];
#pragma unroll
    for (int i = 0; i < 3; i++) t[i] = tv[i];
#pragma unroll
    for (int i = 0; i < 16; i++) L[i] = Lq[i];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < P; i += gridDim.x * blockDim.x) {
        const float x = xyz[3 * (size_t)i], y = xyz[3 * (size_t)i + 1], z = xyz[3 * (size_t)i + 2];
        means3D[3 * (size_t)i] = R[0] * x + R[1] * y + R[2] * z + t[0];
        means3D[3 * (size_t)i + 1] = R[3] * x + R[4] * y + R[5] * z + t[1];
        means3D[3 * (size_t)i + 2] = R[6] * x + R[7] * y + R[8] * z + t[2];
        const float4 q = reinterpret_cast<const float4 *>(rot)[i];
        reinterpret_cast<float4 *>(rotations)[i] =
            make_float4(L[0] * q.x + L[1] * q.y + L[2] * q.z + L[3] * q.w, L[4] * q.x + L[5] * q.y + L[6] * q.z + L[7] * q.w,
                        L[8] * q.x + L[9] * q.y + L[10] * q.z + L[11] * q.w, L[12] * q.x + L[13] * q.y + L[14] * q.z + L[15] * q.w);
#pragma unroll
        for (int k = 0; k < 3; k++) scales[3 * (size_t)i + k] = expf(scaling[3 * (size_t)i + k]);
        const float s = 1.0f / (1.0f + expf(-opacity_raw[i]));
        const float c = conf_flat[mask_index ? mask_index[i] : (int64_t)i];
        opacities[i] = s * c;
    }
}

__global__ void __launch_bounds__(256) pretransform_backward_kernel(
    int P, const float *__restrict__ xyz, const float *__restrict__ rot, const float *__restrict__ scaling,
    const float *__restrict__ opacity_raw, const float *__restrict__ conf_flat, const int64_t *__restrict__ mask_index,
    const float *__restrict__ Rm, const float *__restrict__ Lq, const float *__restrict__ g_means3D, const float *__restrict__ g_rot,
    const float *__restrict__ g_scales, const float *__restrict__ g_opac, float *__restrict__ g_xyz, float *__restrict__ g_rotation,
    float *__restrict__ g_scaling, float *__restrict__ g_opacity_raw, float *__restrict__ g_conf_flat, float *__restrict__ g_small) {
    __shared__ float red[4][28];
    float R[9], L[16];
#pragma unroll
    for (int i = 0; i < 9; i++) R[i] = Rm[i];
#pragma unroll
    for (int i = 0; i < 16; i++) L[i] = Lq[i];
    float acc[28];   // dR (9, row-major), dt (3), dLq (16, row-major)
#pragma unroll
    for (int i = 0; i < 28; i++) acc[i] = 0.f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < P; i += gridDim.x * blockDim.x) {
        const float x = xyz[3 * (size_t)i], y = xyz[3 * (size_t)i + 1], z = xyz[3 * (size_t)i + 2];
        const float gx = g_means3D[3 * (size_t)i], gy = g_means3D[3 * (size_t)i + 1], gz = g_means3D[3 * (size_t)i + 2];
        g_xyz[3 * (size_t)i] = R[0] * gx + R[3] * gy + R[6] * gz;       // R^T g
        g_xyz[3 * (size_t)i + 1] = R[1] * gx + R[4] * gy + R[7] * gz;
        g_xyz[3 * (size_t)i + 2] = R[2] * gx + R[5] * gy + R[8] * gz;
        acc[0] += gx * x; acc[1] += gx * y; acc[2] += gx * z;
        acc[3] += gy * x; acc[4] += gy * y; acc[5] += gy * z;
        acc[6] += gz * x; acc[7] += gz * y; acc[8] += gz * z;
        acc[9] += gx; acc[10] += gy; acc[11] += gz;
        const float4 q = reinterpret_cast<const float4 *>(rot)[i];
        const float4 gq = reinterpret_cast<const float4 *>(g_rot)[i];
        reinterpret_cast<float4 *>(g_rotation)[i] =
            make_float4(L[0] * gq.x + L[4] * gq.y + L[8] * gq.z + L[12] * gq.w, L[1] * gq.x + L[5] * gq.y + L[9] * gq.z + L[13] * gq.w,
                        L[2] * gq.x + L[6] * gq.y + L[10] * gq.z + L[14] * gq.w, L[3] * gq.x + L[7] * gq.y + L[11] * gq.z + L[15] * gq.w);
        const float gv[4] = {gq.x, gq.y, gq.z, gq.w}, qv[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int b = 0; b < 4; b++) acc[12 + 4 * a + b] += gv[a] * qv[b];
#pragma unroll
        for (int k = 0; k < 3; k++) g_scaling[3 * (size_t)i + k] = g_scales[3 * (size_t)i + k] * expf(scaling[3 * (size_t)i + k]);
        const float s = 1.0f / (1.0f + expf(-opacity_raw[i]));
        const int64_t ci = mask_index ? mask_index[i] : (int64_t)i;
        const float c = conf_flat[ci], go = g_opac[i];
        g_opacity_raw[i] = go * c * s * (1.0f - s);
        g_conf_flat[ci] = go * s;   // mask positions are unique: plain store into the pre-zeroed buffer
    }
    // 28 sums over all splats: wave reduction on the DPP network, 4 waves through LDS, one atomic per workgroup
    const int lane = __lane_id(), wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 28; i++) {
        const float r = wave_sum_to_lane63(acc[i]);
        if (lane == 63) red[wave][i] = r;
    }
    __syncthreads();
    if (threadIdx.x < 28) {
        const float r = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        if (r != 0.f) unsafeAtomicAdd(&g_small[threadIdx.x], r);
    }
}

}  // namespace das3r

using namespace das3r;

extern "C" int das3r_pretransform_forward(int32_t P, const float *xyz, const float *rot, const float *scaling, const float *opacity_raw,
                                          const float *conf_flat, const int64_t *mask_index, const float *R, const float *t,
                                          const float *Lq, float *means3D, float *rotations, float *scales, float *opacities,
                                          das3r_stream_t stream) {
    if (P < 0 || (P > 0 && (!xyz || !rot || !scaling || !opacity_raw || !conf_flat || !R || !t || !Lq || !means3D || !rotations ||
                            !scales || !opacities))) {
        set_error("das3r_pretransform_forward: invalid argument");
        return DAS3R_ERR_INVALID_ARG;
    }
    if (P == 0) return DAS3R_OK;
    hipStream_t s = (hipStream_t)stream;
    const int blocks = div_up(P, 256) < 4096 ? div_up(P, 256) : 4096;
    DAS3R_LAUNCH(pretransform_forward_kernel, dim3(blocks), dim3(256), 0, s, P, xyz, rot, scaling, opacity_raw, conf_flat, mask_index, R,
                 t, Lq, means3D, rotations, scales, opacities);
    KERNEL_CHECK(s, false, "pretransform_forward");
    return DAS3R_OK;
}

extern "C" int das3r_pretransform_backward(int32_t P, const float *xyz, const float *rot, const float *scaling, const float *opacity_raw,
                                           const float *conf_flat, const int64_t *mask_index, const float *R, const float *Lq,
                                           const float *g_means3D, const float *g_rot, const float *g_scales, const float *g_opac,
                                           float *g_xyz, float *g_rotation, float *g_scaling, float *g_opacity_raw, float *g_conf_flat,
                                           float *g_small, das3r_stream_t stream) {
    if (P < 0 || (P > 0 && (!xyz || !rot || !scaling || !opacity_raw || !conf_flat || !R || !Lq || !g_means3D || !g_rot || !g_scales ||
                            !g_opac || !g_xyz || !g_rotation || !g_scaling || !g_opacity_raw || !g_conf_flat || !g_small))) {
        set_error("das3r_pretransform_backward: invalid argument");
        return DAS3R_ERR_INVALID_ARG;
    }
    if (P == 0) return DAS3R_OK;
    hipStream_t s = (hipStream_t)stream;
    const int blocks = div_up(P, 256) < 1024 ? div_up(P, 256) : 1024;
    DAS3R_LAUNCH(pretransform_backward_kernel, dim3(blocks), dim3(256), 0, s, P, xyz, rot, scaling, opacity_raw, conf_flat, mask_index, R,
                 Lq, g_means3D, g_rot, g_scales, g_opac, g_xyz, g_rotation, g_scaling, g_opacity_raw, g_conf_flat, g_small);
    KERNEL_CHECK(s, false, "pretransform_backward");
    return DAS3R_OK;
}
