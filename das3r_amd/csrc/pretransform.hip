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
#include <vector>

#include "common.h"
#include "adam_math.h"
#include "pretransform_math.h"
#include "pretransform_chain.h"

namespace das3r {

__global__ void __launch_bounds__(256) pretransform_forward_kernel(int P, const float *__restrict__ xyz, const float *__restrict__ rot,
                                                                  const float *__restrict__ scaling, const float *__restrict__ opacity_raw,
                                                                  const float *__restrict__ conf_flat, const int64_t *__restrict__ mask_index,
                                                                  const float *__restrict__ Rm, const float *__restrict__ tv,
                                                                  const float *__restrict__ Lq, float *__restrict__ means3D,
                                                                  float *__restrict__ rotations, float *__restrict__ scales,
                                                                  float *__restrict__ opacities) {
    PoseRegs pose;   // (pretransform_math.h: the arithmetic the rasterizer's own kernels repeat when they take the raw parameters)
    load_pose(Rm, tv, Lq, pose);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < P; i += gridDim.x * blockDim.x) {
        const float3 m = pre_mean(pose, xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2]);
        means3D[3 * (size_t)i] = m.x;
        means3D[3 * (size_t)i + 1] = m.y;
        means3D[3 * (size_t)i + 2] = m.z;
        reinterpret_cast<float4 *>(rotations)[i] = pre_rot(pose, reinterpret_cast<const float4 *>(rot)[i]);
#pragma unroll
        for (int k = 0; k < 3; k++) scales[3 * (size_t)i + k] = pre_scale(scaling[3 * (size_t)i + k]);
        opacities[i] = pre_opacity(opacity_raw[i], conf_flat[mask_index ? mask_index[i] : (int64_t)i]);
    }
}

// (GeometryAdam, the per-Gaussian chain rule + Adam step and the fixed-order pose sums: pretransform_chain.h — shared with the rasterizer's
//  backward, which can take this whole kernel's work on itself: preprocess_bwd.hip CHAIN)

// MODE 0: the backward as a producer of gradients (g_xyz, g_rotation, g_scaling, g_opacity_raw, g_conf_flat, the 28 pose sums).
// MODE 1 (round 4): the same gradients never leave the registers — the Adam step of the four tensors is taken on the spot (adam_math.h:
//   the arithmetic of adam_kernel): the gradients are not written (44 B per Gaussian) and not read back by the optimizer (44 B), the
//   parameters are read once for both; g_conf_flat and the pose sums leave as before (the confidence map and the camera have their own steps).
// MODE 2: the pose sums alone (the held-out-pose pass drops every other gradient: das3r_amd/fast_step.py test_pose_step).
template <int MODE>
__global__ void __launch_bounds__(256) pretransform_backward_kernel(
    int P, const float *xyz, const float *rot, const float *scaling, const float *opacity_raw /*(MODE 1: A.p[0 .. 3], written below — no __restrict__)*/,
    const float *__restrict__ conf_flat, const int64_t *__restrict__ mask_index, const float *__restrict__ Rm, const float *__restrict__ Lq,
    const float *__restrict__ g_means3D, const float *__restrict__ g_rot, const float *__restrict__ g_scales, const float *__restrict__ g_opac,
    float *__restrict__ g_xyz, float *__restrict__ g_rotation, float *__restrict__ g_scaling, float *__restrict__ g_opacity_raw,
    float *__restrict__ g_conf_flat, float *__restrict__ g_small, const GeometryAdam A,
    float *__restrict__ det_partials /*[POSE_MAX_BLOCKS][28] + one arrival word behind them; null: one float atomic per workgroup and sum*/) {
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
        ChainIn in;
        in.x = xyz[3 * (size_t)i]; in.y = xyz[3 * (size_t)i + 1]; in.z = xyz[3 * (size_t)i + 2];
        in.gx = g_means3D[3 * (size_t)i]; in.gy = g_means3D[3 * (size_t)i + 1]; in.gz = g_means3D[3 * (size_t)i + 2];
        in.q = reinterpret_cast<const float4 *>(rot)[i];
        in.gq = reinterpret_cast<const float4 *>(g_rot)[i];
        chain_pose_acc(in, acc);
        if constexpr (MODE == 2) continue;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            in.sc[k] = scaling[3 * (size_t)i + k];
            in.gs[k] = g_scales[3 * (size_t)i + k];
        }
        in.o = opacity_raw[i];
        const int64_t ci = mask_index ? mask_index[i] : (int64_t)i;
        in.c = conf_flat[ci];
        in.go = g_opac[i];
        ChainOut o;
        chain_grads(R, L, in, o);
        g_conf_flat[ci] = o.gconf;   // mask positions are unique: plain store into the pre-zeroed buffer
        if constexpr (MODE == 0) {
            g_xyz[3 * (size_t)i] = o.rx[0];
            g_xyz[3 * (size_t)i + 1] = o.rx[1];
            g_xyz[3 * (size_t)i + 2] = o.rx[2];
            reinterpret_cast<float4 *>(g_rotation)[i] = o.rq;
#pragma unroll
            for (int k = 0; k < 3; k++) g_scaling[3 * (size_t)i + k] = o.rs[k];
            g_opacity_raw[i] = o.ro;
        } else {
            chain_adam(A, (size_t)i, in, o);
        }
    }
    // 28 sums over all splats, in a fixed order when the caller has scratch for it (pretransform_chain.h)
    pose_sums_finish(acc, red, det_partials, det_partials ? reinterpret_cast<uint32_t *>(det_partials + (size_t)POSE_MAX_BLOCKS * 28) : nullptr, g_small);
}

// pose (qw,qx,qy,qz,tx,ty,tz) -> mats[28] = R (9, row-major; rotation of the NORMALISED quaternion, like get_camera_from_tensor),
// t (3), Lq (16, row-major; left-multiplication matrix of the RAW quaternion, like quadmultiply).  One lane.
__global__ void pose_matrices_kernel(const float *__restrict__ pose, const float *__restrict__ trans, float *__restrict__ mats) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float w = pose[0], x = pose[1], y = pose[2], z = pose[3];
    const float inv = 1.0f / sqrtf(w * w + x * x + y * y + z * z);
    const float r = w * inv, i = x * inv, j = y * inv, k = z * inv;
    const float R[9] = {1 - 2 * (j * j + k * k), 2 * (i * j - r * k), 2 * (i * k + r * j), 2 * (i * j + r * k), 1 - 2 * (i * i + k * k),
                        2 * (j * k - r * i), 2 * (i * k - r * j), 2 * (j * k + r * i), 1 - 2 * (i * i + j * j)};
    const float L[16] = {w, -x, -y, -z, x, w, -z, y, y, z, w, -x, z, -y, x, w};
    for (int a = 0; a < 9; a++) mats[a] = R[a];
    for (int a = 0; a < 3; a++) mats[9 + a] = trans[a];
    for (int a = 0; a < 16; a++) mats[12 + a] = L[a];
}

// chain rule of pose_matrices: g[28] = dL/d(R, t, Lq) -> g_pose[7].  One lane.
// g_t: where dL/dt goes (g_pose + 4 for the 7-vector form); rearm: zero g[0 .. 28) afterwards (the sums of the next backward are
// accumulated into it with atomics: das3r_pose_chain_qt keeps the caller's buffer zero at rest)
__global__ void pose_chain_kernel(const float *__restrict__ pose, float *__restrict__ g, float *__restrict__ g_pose, float *__restrict__ g_t, int rearm,
                                  float *__restrict__ zero_q = nullptr, float *__restrict__ zero_t = nullptr) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    // (das3r_pose_chain_qt_rearm: the rows another view left in the dense pose gradients, zeroed BEFORE this view's are written — they may be the same)
    if (zero_q != nullptr)
        for (int a = 0; a < 4; a++) zero_q[a] = 0.f;
    if (zero_t != nullptr)
        for (int a = 0; a < 3; a++) zero_t[a] = 0.f;
    const float w = pose[0], x = pose[1], y = pose[2], z = pose[3];
    const float n = sqrtf(w * w + x * x + y * y + z * z), inv = 1.0f / n;
    const float r = w * inv, i = x * inv, j = y * inv, k = z * inv;
    const float *G = g;          // dL/dR, row-major
    // dL/d(normalised quaternion): entry-wise derivative of R(r, i, j, k)
    const float dr = -2 * k * G[1] + 2 * j * G[2] + 2 * k * G[3] - 2 * i * G[5] - 2 * j * G[6] + 2 * i * G[7];
    const float di = 2 * j * G[1] + 2 * k * G[2] + 2 * j * G[3] - 4 * i * G[4] - 2 * r * G[5] + 2 * k * G[6] + 2 * r * G[7] - 4 * i * G[8];
    const float dj = -4 * j * G[0] + 2 * i * G[1] + 2 * r * G[2] + 2 * i * G[3] + 2 * k * G[5] - 2 * r * G[6] + 2 * k * G[7] - 4 * j * G[8];
    const float dk = -4 * k * G[0] - 2 * r * G[1] + 2 * i * G[2] + 2 * r * G[3] - 4 * k * G[4] + 2 * j * G[5] + 2 * i * G[6] + 2 * j * G[7];
    // through the normalisation q / |q|
    const float dot = r * dr + i * di + j * dj + k * dk;
    float gq[4] = {(dr - r * dot) * inv, (di - i * dot) * inv, (dj - j * dot) * inv, (dk - k * dot) * inv};
    const float *H = g + 12;     // dL/dLq, row-major; Lq is linear in the raw quaternion
    gq[0] += H[0] + H[5] + H[10] + H[15];
    gq[1] += -H[1] + H[4] - H[11] + H[14];
    gq[2] += -H[2] + H[7] + H[8] - H[13];
    gq[3] += -H[3] - H[6] + H[9] + H[12];
    for (int a = 0; a < 4; a++) g_pose[a] = gq[a];
    for (int a = 0; a < 3; a++) g_t[a] = g[9 + a];
    if (rearm)
        for (int a = 0; a < 28; a++) g[a] = 0.f;
}

// Scratch of the fixed-order pose sums: [MAX_BLOCKS][28] floats + the arrival word, one per (host thread, device, stream) — launches on
// one stream are ordered, launches of different threads / streams each get their own.  Zeroed once (the kernel re-arms the word).
static float *pose_sum_scratch(hipStream_t s) {
    struct Slot { int dev; hipStream_t stream; float *buf; };
    static thread_local std::vector<Slot> slots;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    for (auto &e : slots)
        if (e.dev == dev && e.stream == s) return e.buf;
    float *buf = nullptr;
    const size_t bytes = ((size_t)POSE_MAX_BLOCKS * 28 + 4) * sizeof(float);
    // (a failure here sends the launch down the float-atomic path — correct sums, not bit-reproducible ones: said once, not silently)
    auto complain = [] {
        static bool said = false;
        if (!__atomic_exchange_n(&said, true, __ATOMIC_RELAXED))
            fprintf(stderr, "[das3r] no scratch for the fixed-order pose sums: this thread's pose gradients are summed with float atomics (not bit-reproducible)\n");
    };
    if (hipMalloc((void **)&buf, bytes) != hipSuccess) { complain(); return nullptr; }
    // the one-time fill is COMPLETE before anybody counts in it: the slot is keyed by stream, but a caller may hand the same scratch's
    // stream over to another thread's work order later; one synchronise per (thread, device, stream) lifetime costs nothing
    if (hipMemsetAsync(buf, 0, bytes, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) { (void)hipFree(buf); complain(); return nullptr; }
    slots.push_back({dev, s, buf});
    return buf;
}

}  // namespace das3r

using namespace das3r;

extern "C" int das3r_pose_matrices(const float *pose, float *mats, das3r_stream_t stream) {
    if (!pose || !mats) { set_error("das3r_pose_matrices: invalid argument"); return DAS3R_ERR_INVALID_ARG; }
    hipStream_t s = (hipStream_t)stream;
    DAS3R_LAUNCH(pose_matrices_kernel, dim3(1), dim3(64), 0, s, pose, pose + 4, mats);
    KERNEL_CHECK(s, false, "pose_matrices");
    return DAS3R_OK;
}

extern "C" int das3r_pose_matrices_qt(const float *q, const float *t, float *mats, das3r_stream_t stream) {
    if (!q || !t || !mats) { set_error("das3r_pose_matrices_qt: invalid argument"); return DAS3R_ERR_INVALID_ARG; }
    hipStream_t s = (hipStream_t)stream;
    DAS3R_LAUNCH(pose_matrices_kernel, dim3(1), dim3(64), 0, s, q, t, mats);
    KERNEL_CHECK(s, false, "pose_matrices");
    return DAS3R_OK;
}

extern "C" int das3r_pose_chain_qt(const float *q, float *g_mats, float *g_q, float *g_t, das3r_stream_t stream) {
    if (!q || !g_mats || !g_q || !g_t) { set_error("das3r_pose_chain_qt: invalid argument"); return DAS3R_ERR_INVALID_ARG; }
    hipStream_t s = (hipStream_t)stream;
    DAS3R_LAUNCH(pose_chain_kernel, dim3(1), dim3(64), 0, s, q, g_mats, g_q, g_t, 1);
    KERNEL_CHECK(s, false, "pose_chain");
    return DAS3R_OK;
}

extern "C" int das3r_pose_chain_qt_rearm(const float *q, float *g_mats, float *g_q, float *g_t, float *zero_q, float *zero_t, das3r_stream_t stream) {
    if (!q || !g_mats || !g_q || !g_t) { set_error("das3r_pose_chain_qt_rearm: invalid argument"); return DAS3R_ERR_INVALID_ARG; }
    hipStream_t s = (hipStream_t)stream;
    DAS3R_LAUNCH(pose_chain_kernel, dim3(1), dim3(64), 0, s, q, g_mats, g_q, g_t, 1, zero_q, zero_t);
    KERNEL_CHECK(s, false, "pose_chain");
    return DAS3R_OK;
}

extern "C" int das3r_pose_chain(const float *pose, const float *g_mats, float *g_pose, das3r_stream_t stream) {
    if (!pose || !g_mats || !g_pose) { set_error("das3r_pose_chain: invalid argument"); return DAS3R_ERR_INVALID_ARG; }
    hipStream_t s = (hipStream_t)stream;
    DAS3R_LAUNCH(pose_chain_kernel, dim3(1), dim3(64), 0, s, pose, const_cast<float *>(g_mats), g_pose, g_pose + 4, 0);
    KERNEL_CHECK(s, false, "pose_chain");
    return DAS3R_OK;
}

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
    DAS3R_LAUNCH((pretransform_backward_kernel<0>), dim3(blocks), dim3(256), 0, s, P, xyz, rot, scaling, opacity_raw, conf_flat, mask_index, R,
                 Lq, g_means3D, g_rot, g_scales, g_opac, g_xyz, g_rotation, g_scaling, g_opacity_raw, g_conf_flat, g_small, GeometryAdam{}, pose_sum_scratch(s));
    KERNEL_CHECK(s, false, "pretransform_backward");
    return DAS3R_OK;
}

// The backward of the pre-transform with the Adam step of the four tensors it differentiates taken in the same pass (ABI 11; see
// pretransform_backward_kernel MODE 1).  xyz / rot / scaling / opacity_raw ARE slots[0 .. 3].param and are updated in place.
extern "C" int das3r_pretransform_backward_adam(int32_t P, const float *conf_flat, const int64_t *mask_index, const float *R, const float *Lq,
                                                const float *g_means3D, const float *g_rot, const float *g_scales, const float *g_opac,
                                                float *g_conf_flat, float *g_small, const das3r_adam_slot *slots, float beta1, float beta2, float eps,
                                                das3r_stream_t stream) {
    if (P < 0 || !slots || (P > 0 && (!conf_flat || !R || !Lq || !g_means3D || !g_rot || !g_scales || !g_opac || !g_conf_flat || !g_small))) {
        set_error("das3r_pretransform_backward_adam: invalid argument");
        return DAS3R_ERR_INVALID_ARG;
    }
    GeometryAdam A;
    for (int k = 0; k < 4; k++) {
        if (P > 0 && (!slots[k].param || !slots[k].exp_avg || !slots[k].exp_avg_sq || !(slots[k].bc2_sqrt > 0.f))) {
            set_error("das3r_pretransform_backward_adam: bad slot %d", k);
            return DAS3R_ERR_INVALID_ARG;
        }
        A.p[k] = slots[k].param; A.m[k] = slots[k].exp_avg; A.v[k] = slots[k].exp_avg_sq;
        A.step_size[k] = slots[k].step_size; A.bc2_sqrt[k] = slots[k].bc2_sqrt;
    }
    A.beta1 = beta1; A.beta2 = beta2; A.eps = eps;
    if (P == 0) return DAS3R_OK;
    hipStream_t s = (hipStream_t)stream;
    const int blocks = div_up(P, 256) < 2048 ? div_up(P, 256) : 2048;
    DAS3R_LAUNCH((pretransform_backward_kernel<1>), dim3(blocks), dim3(256), 0, s, P, (const float *)A.p[0], (const float *)A.p[1], (const float *)A.p[2],
                 (const float *)A.p[3], conf_flat, mask_index, R, Lq, g_means3D, g_rot, g_scales, g_opac, (float *)nullptr, (float *)nullptr,
                 (float *)nullptr, (float *)nullptr, g_conf_flat, g_small, A, pose_sum_scratch(s));
    KERNEL_CHECK(s, false, "pretransform_backward_adam");
    return DAS3R_OK;
}

// The 28 pose sums alone (MODE 2): what a pass that only differentiates the camera needs.
extern "C" int das3r_pretransform_pose_sums(int32_t P, const float *xyz, const float *rot, const float *R, const float *Lq, const float *g_means3D,
                                            const float *g_rot, float *g_small, das3r_stream_t stream) {
    if (P < 0 || (P > 0 && (!xyz || !rot || !R || !Lq || !g_means3D || !g_rot || !g_small))) {
        set_error("das3r_pretransform_pose_sums: invalid argument");
        return DAS3R_ERR_INVALID_ARG;
    }
    if (P == 0) return DAS3R_OK;
    hipStream_t s = (hipStream_t)stream;
    const int blocks = div_up(P, 256) < 1024 ? div_up(P, 256) : 1024;
    DAS3R_LAUNCH((pretransform_backward_kernel<2>), dim3(blocks), dim3(256), 0, s, P, xyz, rot, (const float *)nullptr, (const float *)nullptr,
                 (const float *)nullptr, (const int64_t *)nullptr, R, Lq, g_means3D, g_rot, (const float *)nullptr, (const float *)nullptr,
                 (float *)nullptr, (float *)nullptr, (float *)nullptr, (float *)nullptr, (float *)nullptr, g_small, GeometryAdam{}, pose_sum_scratch(s));
    KERNEL_CHECK(s, false, "pretransform_pose_sums");
    return DAS3R_OK;
}
