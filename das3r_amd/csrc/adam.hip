// adam.hip — SURVEY.md §8(f)-2: one multi-tensor Adam step for the seven per-Gaussian parameter groups DAS3R optimises
// (/root/reference/scene/gaussian_model.py:236-261: torch.optim.Adam(lr=0, eps=1e-15), no weight decay, no amsgrad).
// torch makes ~7 passes over every array (59 floats per Gaussian); here each element's (p, g, m, v) is read once and
// (p, m, v) written once, all groups in ONE launch, and the update is DEGREE-AWARE: a tensor laid out as rows
// (f_rest: [P, 15, 3]) only touches the first `active_len` floats of each row.  Coefficients above the active SH degree
// have g = m = v = 0, for which Adam's update is exactly 0, so skipping them is exact and removes 45 of the 59 floats
// per Gaussian from the sweep while the active degree is 0 (iterations 1..2999 of a 4000-iteration DAS3R run).
// Same arithmetic as torch's single-tensor path: m += (g - m)(1 - b1); v = v b2 + (1 - b2) g g;
// p -= step_size * m / (sqrt(v) / sqrt(bc2) + eps) with step_size = lr / bc1 computed on the host in double.
#include <string.h>

#include "common.h"
#include "adam_math.h"

namespace das3r {

constexpr int ADAM_MAX_TENSORS = 16;
constexpr int ADAM_CHUNK = 256 * 8;

struct AdamTable {
    float *p[ADAM_MAX_TENSORS];
    const float *g[ADAM_MAX_TENSORS];
    float *m[ADAM_MAX_TENSORS];
    float *v[ADAM_MAX_TENSORS];
    float *mirror[ADAM_MAX_TENSORS];
    int mirror_row_len[ADAM_MAX_TENSORS];
    int flat[ADAM_MAX_TENSORS];                // no row structure to resolve: offsets = element index
    long long n_active[ADAM_MAX_TENSORS];     // rows * active_len
    int row_len[ADAM_MAX_TENSORS], active_len[ADAM_MAX_TENSORS], grad_row_len[ADAM_MAX_TENSORS], state_row_len[ADAM_MAX_TENSORS];
    int first_chunk[ADAM_MAX_TENSORS + 1];    // prefix of chunk counts
    float step_size[ADAM_MAX_TENSORS], bc2_sqrt[ADAM_MAX_TENSORS], step_size_tail[ADAM_MAX_TENSORS];
    int head_len[ADAM_MAX_TENSORS];
    int n;
};

// GATED: the step only happens if gate[0] > threshold, decided by every workgroup for itself (round 6: a one-thread kernel in front used to
// decide it — 4.8 us of every iteration for a comparison); the bias corrections then come from the device-side step count state[0] + 1 and
// T.step_size holds the plain learning rate.  state[1] counts the workgroups that have READ state[0]: the last one to arrive writes the new
// count and leaves state[1] zero for the next launch.
template <bool GATED>
__global__ void __launch_bounds__(256) adam_kernel(const AdamTable T, float beta1, float beta2, float eps, int32_t *__restrict__ state,
                                                   const float *__restrict__ gate, float threshold) {
    float gate_bc1 = 1.f, gate_bc2_sqrt = 1.f;
    if (GATED) {
        const bool go = gate[0] > threshold;
        const float t = (float)(state[0] + 1);
        __syncthreads();   // (every thread has read the count)
        if (threadIdx.x == 0) {
            const int32_t before = __hip_atomic_fetch_add(&state[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (before == (int32_t)gridDim.x - 1) {
                if (go) state[0] = (int32_t)t;
                __hip_atomic_store(&state[1], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (!go) return;
        gate_bc1 = 1.f - powf(beta1, t);
        gate_bc2_sqrt = sqrtf(1.f - powf(beta2, t));
    }
    int t = 0;
#pragma unroll 1
    while (t + 1 < T.n && (int)blockIdx.x >= T.first_chunk[t + 1]) t++;
    const long long base = (long long)(blockIdx.x - T.first_chunk[t]) * ADAM_CHUNK;
    float *__restrict__ p = T.p[t];
    const float *__restrict__ g = T.g[t];
    float *__restrict__ m = T.m[t];
    float *__restrict__ v = T.v[t];
    const long long n = T.n_active[t];
    float *__restrict__ mirror = T.mirror[t];
    const int mirror_row_len = T.mirror_row_len[t];
    const int row_len = T.row_len[t], active_len = T.active_len[t], grad_row_len = T.grad_row_len[t], state_row_len = T.state_row_len[t];
    const float step_size = GATED ? T.step_size[t] / gate_bc1 : T.step_size[t], bc2_sqrt = GATED ? gate_bc2_sqrt : T.bc2_sqrt[t];
    const float step_size_tail = GATED ? T.step_size_tail[t] / gate_bc1 : T.step_size_tail[t];
    const int head_len = T.head_len[t];
    const bool flat = T.flat[t] != 0;
#pragma unroll
    for (int k = 0; k < ADAM_CHUNK / 256; k++) {
        const long long e = base + k * 256 + threadIdx.x;
        if (e < n) {
            // (row, column) of element e of the active prefix.  A flat tensor — one row, or every stride equal to the active length — needs
            // neither (uniform branch); a row-form tensor has fewer than 2^32 active elements (checked on the host): one 32-bit division
            // instead of the 64-bit division and remainder this used to cost every element of every tensor
            long long off = e, goff = e, moff = e;
            int col = 0;
            uint32_t row = 0u;
            if (!flat) {
                row = (uint32_t)e / (uint32_t)active_len;
                col = (int)((uint32_t)e - row * (uint32_t)active_len);
                off = (long long)row * row_len + col;
                goff = (long long)row * grad_row_len + col;      // (a compact gradient: its own row stride)
                moff = (long long)row * state_row_len + col;     // (compact moments: theirs)
            }
            const float gr = g[goff];
            float pp = p[off], mm = m[moff], vv = v[moff];
            adam_update(pp, mm, vv, gr, beta1, beta2, eps, (flat || col < head_len) ? step_size : step_size_tail, bc2_sqrt);
            p[off] = pp;
            m[moff] = mm;
            v[moff] = vv;
            if (mirror != nullptr) mirror[(long long)row * mirror_row_len + col] = pp;   // (uniform per tensor; never with a flat one)
        }
    }
}

}  // namespace das3r

using namespace das3r;

// tensors: host array of n entries.  rows * row_len = numel; only the first active_len floats of every row are updated
// (active_len == row_len: the whole tensor).  step_size = lr / (1 - beta1^t), bc2_sqrt = sqrt(1 - beta2^t).
static int adam_launch(int32_t n, const das3r_adam_tensor *tensors, float beta1, float beta2, float eps, const float *gate, float threshold,
                       int32_t *state, das3r_stream_t stream) {
    if (n < 0 || n > ADAM_MAX_TENSORS || (n > 0 && !tensors)) {
        set_error("das3r_adam_step: between 0 and %d tensors per call", ADAM_MAX_TENSORS);
        return DAS3R_ERR_INVALID_ARG;
    }
    AdamTable T;
    memset(&T, 0, sizeof(T));
    int chunks = 0, k = 0;
    for (int i = 0; i < n; i++) {
        const das3r_adam_tensor &a = tensors[i];
        const int grl = a.grad_row_len > 0 ? a.grad_row_len : a.row_len;
        const int srl = a.state_row_len > 0 ? a.state_row_len : a.row_len;
        if (a.rows < 0 || a.row_len <= 0 || a.active_len < 0 || a.active_len > a.row_len || a.active_len > grl || a.active_len > srl || !a.param || !a.grad ||
            ((!a.exp_avg || !a.exp_avg_sq) && (long long)a.rows * a.active_len > 0) || (a.mirror && a.mirror_row_len < a.active_len)) {
            set_error("das3r_adam_step: bad tensor %d", i);
            return DAS3R_ERR_INVALID_ARG;
        }
        const long long na = (long long)a.rows * a.active_len;
        if (na == 0) continue;   // nothing active in this tensor (e.g. f_rest while the SH degree is 0)
        T.p[k] = a.param; T.g[k] = a.grad; T.m[k] = a.exp_avg; T.v[k] = a.exp_avg_sq;
        T.n_active[k] = na; T.row_len[k] = a.row_len; T.active_len[k] = a.active_len; T.grad_row_len[k] = grl; T.state_row_len[k] = srl;
        T.mirror[k] = a.mirror; T.mirror_row_len[k] = a.mirror_row_len;
        const bool split_rates = a.head_len > 0 && a.head_len < a.active_len;
        T.flat[k] = (!a.mirror && !split_rates && a.row_len == a.active_len && grl == a.row_len && srl == a.row_len) ? 1 : 0;
        if (!T.flat[k] && na >= (1ll << 32)) { set_error("das3r_adam_step: tensor %d: a row-form tensor with 2^32 or more active elements", i); return DAS3R_ERR_INVALID_ARG; }
        T.step_size[k] = a.step_size; T.bc2_sqrt[k] = a.bc2_sqrt;
        const bool split = a.head_len > 0 && a.head_len < a.active_len;   // otherwise one rate for the whole row
        T.head_len[k] = split ? a.head_len : a.row_len;
        T.step_size_tail[k] = split ? a.step_size_tail : a.step_size;
        T.first_chunk[k] = chunks;
        chunks += (int)((na + ADAM_CHUNK - 1) / ADAM_CHUNK);
        k++;
    }
    T.first_chunk[k] = chunks;
    T.n = k;
    if (chunks == 0) return DAS3R_OK;
    hipStream_t s = (hipStream_t)stream;
    if (gate) {
        DAS3R_LAUNCH((adam_kernel<true>), dim3(chunks), dim3(256), 0, s, T, beta1, beta2, eps, state, gate, threshold);
    } else {
        DAS3R_LAUNCH((adam_kernel<false>), dim3(chunks), dim3(256), 0, s, T, beta1, beta2, eps, (int32_t *)nullptr, (const float *)nullptr, 0.f);
    }
    KERNEL_CHECK(s, false, "adam");
    return DAS3R_OK;
}

extern "C" int das3r_adam_step(int32_t n, const das3r_adam_tensor *tensors, float beta1, float beta2, float eps, das3r_stream_t stream) {
    return adam_launch(n, tensors, beta1, beta2, eps, nullptr, 0.f, nullptr, stream);
}

// Conditional step decided ON THE DEVICE: the update happens iff gate[0] > threshold (DAS3R steps its camera optimizer only
// when the frame's PSNR exceeds 26 dB, train_gui.py:584-586 — a host-side `if` on a device scalar stalls the host every
// iteration).  state[0] = number of steps taken so far (device int32, zero-initialised by the caller, incremented here),
// state[1] = scratch flag; tensors[i].step_size / step_size_tail hold the plain learning rates, bc2_sqrt is ignored.
extern "C" int das3r_adam_step_gated(int32_t n, const das3r_adam_tensor *tensors, float beta1, float beta2, float eps, const float *gate,
                                     float threshold, int32_t *state, das3r_stream_t stream) {
    if (!gate || !state) { set_error("das3r_adam_step_gated: gate and state are required"); return DAS3R_ERR_INVALID_ARG; }
    return adam_launch(n, tensors, beta1, beta2, eps, gate, threshold, state, stream);
}
