// photometric.hip — SURVEY.md §8(f)-3: DAS3R's per-iteration photometric loss as two kernels (forward, backward) instead of
// the ~60 PyTorch launches (6 depthwise 11x11 convolutions each way + elementwise chains) the reference spends on it:
//     image = render * static, gt' = gt * static                                    (/root/reference/train_gui.py:560-566)
//     loss  = mean over (c, y, x) of (1 - lambda) |image - gt'| + lambda (1 - SSIM_map(image, gt'))          (:567-571)
//     mse_c = mean over (y, x) of (image - gt')^2                         (psnr_frame, utils/image_utils.py:17-19)
// SSIM_map: 11x11 Gaussian window (sigma 1.5), zero padding, per channel (/root/reference/utils/loss_utils.py:39-66).
//
// One workgroup = one 16x16 pixel tile.  The Gaussian window is separable: the (16+10)^2 halo tile of each input goes to LDS
// once, a horizontal pass leaves 26 x 16 partial rows in LDS, the vertical pass finishes in registers.  The forward keeps, per
// pixel and channel, the four derivative maps the backward needs (dm/dmu1, dm/dmu2, dm/dE[a^2] = dm/dE[b^2], dm/dE[ab]); since
// dL/dSSIM_map is the same constant for every pixel, the backward is the same separable convolution applied to those maps.
// Reductions are deterministic: every tile writes its partial sums, the host side adds the few hundred rows.
#include "common.h"

namespace das3r {

constexpr int PT = 16, PR = 5, PH = PT + 2 * PR;   // tile, window radius, halo tile edge (26)
constexpr float SSIM_C1 = 0.01f * 0.01f, SSIM_C2 = 0.03f * 0.03f;

struct GaussWin {
    float w[2 * PR + 1];
};

// horizontal pass over a PH x PH tile in LDS -> PH x PT, then vertical pass for this thread's pixel
template <int NQ>
__device__ __forceinline__ void separable(const float (*halo)[PH][PH + 1], float (*tmp)[PH][PT + 1], const GaussWin &g, const int tid,
                                          float out[NQ]) {
    for (int e = tid; e < PH * PT; e += PT * PT) {
        const int r = e / PT, c = e % PT;
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k <= 2 * PR; k++) s = fmaf(g.w[k], halo[q][r][c + k], s);
            tmp[q][r][c] = s;
        }
    }
    __syncthreads();
    const int ty = tid / PT, tx = tid % PT;
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k <= 2 * PR; k++) s = fmaf(g.w[k], tmp[q][ty + k][tx], s);
        out[q] = s;
    }
}

__global__ void __launch_bounds__(256) photometric_forward_kernel(int H, int W, const float *__restrict__ render, const float *__restrict__ gt,
                                                                  const float *__restrict__ stat, float lambda, GaussWin g,
                                                                  float *__restrict__ partials /*[blocks][8]*/,
                                                                  float *__restrict__ dmaps /*[4][3][H][W]*/,
                                                                  float *__restrict__ ssim_map /*[3][H][W] or null (das3r_ssim_map_forward)*/) {
    // (round 6) the three channels side by side: fifteen quantities go through ONE separable pass — three barriers per tile instead of nine,
    // and fifteen independent dot products per thread between them instead of five
    __shared__ float halo[15][PH][PH + 1];
    __shared__ float tmp[15][PH][PT + 1];
    __shared__ float red[4][8];
    const int tid = threadIdx.x, ty = tid / PT, tx = tid % PT;
    const int x0 = blockIdx.x * PT, y0 = blockIdx.y * PT;
    const int x = x0 + tx, y = y0 + ty;
    const bool inside = x < W && y < H;
    const size_t plane = (size_t)H * W;
    float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};   // l1, 1 - ssim, squared error per channel
    // Round 6: the halo tiles of ALL THREE channels are requested before anything is used (21 loads per thread in flight, indices clamped
    // instead of guarded: a guarded load is a basic block of its own with a wait behind it) — the kernel used to make nine exposed trips
    // to memory, one per channel and sweep of the halo, for an image that fits the L2 fifty times
    constexpr int SWEEPS = (PH * PH + PT * PT - 1) / (PT * PT);
#pragma unroll
    for (int u = 0; u < SWEEPS; u++) {
        const int e = tid + u * PT * PT, r = e / PH, col = e % PH;
        const int yy = y0 + r - PR, xx = x0 + col - PR;
        const bool in = e < PH * PH && yy >= 0 && yy < H && xx >= 0 && xx < W;
        const size_t p = (size_t)min(max(yy, 0), H - 1) * W + (size_t)min(max(xx, 0), W - 1);
        const float sm = in ? (stat != nullptr ? stat[p] : 1.f) : 0.f;
        float va[3], vb[3];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float rv = render[c * plane + p], gv = gt[c * plane + p];
            va[c] = in ? rv * sm : 0.f;   // (outside the image: the zero padding of the reference's convolution)
            vb[c] = in ? gv * sm : 0.f;
        }
        if (e < PH * PH) {
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float a = va[c], b = vb[c];
                halo[5 * c + 0][r][col] = a;
                halo[5 * c + 1][r][col] = b;
                halo[5 * c + 2][r][col] = a * a;
                halo[5 * c + 3][r][col] = b * b;
                halo[5 * c + 4][r][col] = a * b;
            }
        }
    }
    __syncthreads();
    float o[15];
    separable<15>(halo, tmp, g, tid, o);
#pragma unroll
    for (int c = 0; c < 3; c++) {
        if (inside) {
            const float mu1 = o[5 * c + 0], mu2 = o[5 * c + 1], e1 = o[5 * c + 2], e2 = o[5 * c + 3], e12 = o[5 * c + 4];
            const float mu1s = mu1 * mu1, mu2s = mu2 * mu2, mu12 = mu1 * mu2;
            const float s1 = e1 - mu1s, s2 = e2 - mu2s, s12 = e12 - mu12;
            const float A = 2.f * mu12 + SSIM_C1, B = 2.f * s12 + SSIM_C2, C = mu1s + mu2s + SSIM_C1, D = s1 + s2 + SSIM_C2;
            const float rC = 1.f / C, rD = 1.f / D, rCD = rC * rD;
            const float m = A * B * rCD;
            const float a = halo[5 * c + 0][ty + PR][tx + PR], b = halo[5 * c + 1][ty + PR][tx + PR];
            const float d = a - b;
            acc[0] += fabsf(d);
            acc[1] += 1.f - m;
            acc[2 + c] = d * d;
            const size_t pix = (size_t)y * W + x;
            const float common = 2.f * (B - A) * rCD;                      // d m / d mu1 = mu2 * common - m * 2 mu1 (1/C - 1/D)
            dmaps[(0 * 3 + c) * plane + pix] = mu2 * common - 2.f * m * mu1 * (rC - rD);
            dmaps[(1 * 3 + c) * plane + pix] = mu1 * common - 2.f * m * mu2 * (rC - rD);
            dmaps[(2 * 3 + c) * plane + pix] = -m * rD;                    // d m / d E[a^2] = d m / d E[b^2]
            dmaps[(3 * 3 + c) * plane + pix] = 2.f * A * rCD;              // d m / d E[ab]
            if (ssim_map != nullptr) ssim_map[c * plane + pix] = m;
        }
    }
    // deterministic tile sums: DPP-free plain LDS tree is plenty here (5 values, once per tile)
    const int lane = __lane_id(), wave = tid >> 6;
#pragma unroll
    for (int q = 0; q < 5; q++) {
        float v = acc[q];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) red[wave][q] = v;
    }
    __syncthreads();
    if (tid < 5) partials[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
}

// the few hundred rows of tile sums -> {loss, mse_r, mse_g, mse_b, psnr_frame} (what the host side of round 3 did with ~12 tiny
// PyTorch kernels per iteration); one workgroup, rows added in a fixed order: deterministic
__device__ __forceinline__ void finish_sums(const int nblocks, const float *__restrict__ partials, const float npix, const float lambda,
                                            float *__restrict__ out /*[8]*/, float (*red)[5] /*[4][5] LDS*/) {
    const int tid = threadIdx.x, lane = __lane_id(), wave = tid >> 6;
    float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    for (int b = tid; b < nblocks; b += 256)
#pragma unroll
        for (int q = 0; q < 5; q++) acc[q] += partials[(size_t)b * 8 + q];
#pragma unroll
    for (int q = 0; q < 5; q++) {
        float v = acc[q];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) red[wave][q] = v;
    }
    __syncthreads();
    if (tid == 0) {
        float s[5];
#pragma unroll
        for (int q = 0; q < 5; q++) s[q] = red[0][q] + red[1][q] + red[2][q] + red[3][q];
        out[0] = ((1.f - lambda) * s[0] + lambda * s[1]) / (3.f * npix);
        float psnr = 0.f;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float mse = s[2 + c] / npix;
            out[1 + c] = mse;
            psnr += 20.f * log10f(1.f / sqrtf(mse));   // utils/image_utils.py:17-19
        }
        out[4] = psnr / 3.f;
        out[5] = out[6] = out[7] = 0.f;
    }
}
__global__ void __launch_bounds__(256) photometric_finish_kernel(int nblocks, const float *__restrict__ partials, float npix, float lambda,
                                                                 float *__restrict__ out /*[8]*/) {
    __shared__ float red[4][5];
    finish_sums(nblocks, partials, npix, lambda, out, red);
}

__global__ void __launch_bounds__(256) photometric_backward_kernel(int H, int W, const float *__restrict__ render, const float *__restrict__ gt,
                                                                   const float *__restrict__ stat, float lambda, GaussWin g,
                                                                   const float *__restrict__ dmaps, const float *__restrict__ grad_loss,
                                                                   float *__restrict__ d_render, float *__restrict__ d_static,
                                                                   const float *__restrict__ partials, int nblocks, float *__restrict__ out8,
                                                                   // das3r_ssim_map_backward: gmap = dL/d(SSIM map) [3][H][W] (then stat may be null, lambda and
                                                                   // grad_loss are not read, and d_b [3][H][W] receives dL/d(second image)); else both null
                                                                   const float *__restrict__ gmap, float *__restrict__ d_b) {
    __shared__ float halo[12][PH][PH + 1];   // (round 6: the four derivative maps of the three channels through one separable pass)
    __shared__ float tmp[12][PH][PT + 1];
    const int tid = threadIdx.x, ty = tid / PT, tx = tid % PT;
    // round 6 (das3r_photometric_backward_finish): the loss and the frame's PSNR — the forward's few hundred rows of tile sums added in a
    // fixed order — are the first workgroup's side duty instead of a launch of their own between the two kernels (4.8 us of every step)
    if (out8 != nullptr && blockIdx.x == 0 && blockIdx.y == 0) {   // (uniform)
        __shared__ float red[4][5];
        finish_sums(nblocks, partials, (float)H * (float)W, lambda, out8, red);
    }
    const int x0 = blockIdx.x * PT, y0 = blockIdx.y * PT;
    const int x = x0 + tx, y = y0 + ty;
    const bool inside = x < W && y < H;
    const size_t plane = (size_t)H * W, pix = (size_t)y * W + x;
    const float scale = gmap != nullptr ? 1.f : grad_loss[0] / (3.f * (float)plane);       // d loss / d (per-element term)
    const float s = inside ? (stat != nullptr ? stat[pix] : 1.f) : 0.f;
    float ds = 0.f;
    // (as in the forward: everything is requested up front, indices clamped instead of guarded)
    constexpr int SWEEPS = (PH * PH + PT * PT - 1) / (PT * PT);
#pragma unroll
    for (int u = 0; u < SWEEPS; u++) {
        const int e = tid + u * PT * PT, r = e / PH, col = e % PH;
        const int yy = y0 + r - PR, xx = x0 + col - PR;
        const bool in = e < PH * PH && yy >= 0 && yy < H && xx >= 0 && xx < W;
        const size_t p = (size_t)min(max(yy, 0), H - 1) * W + (size_t)min(max(xx, 0), W - 1);
        float vm[12];
#pragma unroll
        for (int c = 0; c < 3; c++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                // (with an upstream gradient PER PIXEL the derivative maps are weighted by it before the window sums: dL/da(p) =
                //  sum over the window centres q of G(q) w(p - q) dm(q) — the same separable convolution, of G x dm)
                const float v = dmaps[(q * 3 + c) * plane + p] * (gmap != nullptr ? gmap[c * plane + p] : 1.f);
                vm[4 * c + q] = in ? v : 0.f;
            }
        if (e < PH * PH) {
#pragma unroll
            for (int k = 0; k < 12; k++) halo[k][r][col] = vm[k];
        }
    }
    const size_t pixc = (size_t)min(y, H - 1) * W + (size_t)min(x, W - 1);
    float Rv[3], Gv[3];
#pragma unroll
    for (int c = 0; c < 3; c++) Rv[c] = render[c * plane + pixc], Gv[c] = gt[c * plane + pixc];
    __syncthreads();
    float o[12];
    separable<12>(halo, tmp, g, tid, o);
#pragma unroll
    for (int c = 0; c < 3; c++) {
        if (inside) {
            const float R = Rv[c], G = Gv[c];
            const float a = R * s, b = G * s;
            const float sgn = a > b ? 1.f : (a < b ? -1.f : 0.f);
            const float l1 = (1.f - lambda) * sgn;
            if (gmap != nullptr) {   // (uniform) the SSIM map's own backward: both images, no mask, no L1 term
                d_render[c * plane + pix] = o[4 * c + 0] + 2.f * a * o[4 * c + 2] + b * o[4 * c + 3];
                d_b[c * plane + pix] = o[4 * c + 1] + 2.f * b * o[4 * c + 2] + a * o[4 * c + 3];
                continue;
            }
            const float da = scale * (l1 - lambda * (o[4 * c + 0] + 2.f * a * o[4 * c + 2] + b * o[4 * c + 3]));
            const float db = scale * (-l1 - lambda * (o[4 * c + 1] + 2.f * b * o[4 * c + 2] + a * o[4 * c + 3]));
            d_render[c * plane + pix] = da * s;
            ds += da * R + db * G;
        }
    }
    if (inside && d_static != nullptr) d_static[pix] = ds;
}

static GaussWin make_window() {
    GaussWin g;
    double v[2 * PR + 1], sum = 0.0;
    for (int i = 0; i <= 2 * PR; i++) {
        v[i] = exp(-(double)((i - PR) * (i - PR)) / (2.0 * 1.5 * 1.5));
        sum += v[i];
    }
    for (int i = 0; i <= 2 * PR; i++) g.w[i] = (float)(v[i] / sum);
    return g;
}

}  // namespace das3r

using namespace das3r;

extern "C" int64_t das3r_photometric_blocks(int32_t H, int32_t W) {
    if (H <= 0 || W <= 0) return 0;
    return (int64_t)div_up(W, PT) * div_up(H, PT);
}

extern "C" int das3r_photometric_forward(int32_t H, int32_t W, const float *render, const float *gt, const float *static_mask, float lambda,
                                         float *partials, float *dmaps, das3r_stream_t stream) {
    if (H <= 0 || W <= 0 || !render || !gt || !static_mask || !partials || !dmaps) {
        set_error("das3r_photometric_forward: invalid argument");
        return DAS3R_ERR_INVALID_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    DAS3R_LAUNCH(photometric_forward_kernel, dim3(div_up(W, PT), div_up(H, PT)), dim3(PT * PT), 0, s, H, W, render, gt, static_mask, lambda,
                 make_window(), partials, dmaps, (float *)nullptr);
    KERNEL_CHECK(s, false, "photometric_forward");
    return DAS3R_OK;
}

extern "C" int das3r_photometric_finish(int32_t H, int32_t W, const float *partials, float lambda, float *out8, das3r_stream_t stream) {
    if (H <= 0 || W <= 0 || !partials || !out8) { set_error("das3r_photometric_finish: invalid argument"); return DAS3R_ERR_INVALID_ARG; }
    hipStream_t s = (hipStream_t)stream;
    DAS3R_LAUNCH(photometric_finish_kernel, dim3(1), dim3(256), 0, s, (int)das3r_photometric_blocks(H, W), partials, (float)H * (float)W, lambda, out8);
    KERNEL_CHECK(s, false, "photometric_finish");
    return DAS3R_OK;
}

extern "C" int das3r_photometric_backward(int32_t H, int32_t W, const float *render, const float *gt, const float *static_mask, float lambda,
                                          const float *dmaps, const float *grad_loss, float *d_render, float *d_static,
                                          das3r_stream_t stream) {
    if (H <= 0 || W <= 0 || !render || !gt || !static_mask || !dmaps || !grad_loss || !d_render || !d_static) {
        set_error("das3r_photometric_backward: invalid argument");
        return DAS3R_ERR_INVALID_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    DAS3R_LAUNCH(photometric_backward_kernel, dim3(div_up(W, PT), div_up(H, PT)), dim3(PT * PT), 0, s, H, W, render, gt, static_mask,
                 lambda, make_window(), dmaps, grad_loss, d_render, d_static, (const float *)nullptr, 0, (float *)nullptr, (const float *)nullptr, (float *)nullptr);
    KERNEL_CHECK(s, false, "photometric_backward");
    return DAS3R_OK;
}

extern "C" int das3r_photometric_backward_finish(int32_t H, int32_t W, const float *render, const float *gt, const float *static_mask, float lambda,
                                                 const float *dmaps, const float *grad_loss, float *d_render, float *d_static,
                                                 const float *partials, float *out8, das3r_stream_t stream) {
    if (H <= 0 || W <= 0 || !render || !gt || !static_mask || !dmaps || !grad_loss || !d_render || !d_static || !partials || !out8) {
        set_error("das3r_photometric_backward_finish: invalid argument");
        return DAS3R_ERR_INVALID_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    DAS3R_LAUNCH(photometric_backward_kernel, dim3(div_up(W, PT), div_up(H, PT)), dim3(PT * PT), 0, s, H, W, render, gt, static_mask,
                 lambda, make_window(), dmaps, grad_loss, d_render, d_static, partials, (int)das3r_photometric_blocks(H, W), out8, (const float *)nullptr, (float *)nullptr);
    KERNEL_CHECK(s, false, "photometric_backward");
    return DAS3R_OK;
}

// ABI 15 — the SSIM MAP of two images (utils/loss_utils.py:39-66 with size_average = False: what train_gui.py:568 combines with its L1 map)
// and its backward for an arbitrary upstream gradient, on the photometric kernels: two launches each way instead of the reference's
// twelve depthwise convolutions and their elementwise chains (0.75 ms of MIOpen kernels per iteration at 512 x 208).
extern "C" int das3r_ssim_map_forward(int32_t H, int32_t W, const float *img1, const float *img2, float *ssim_map, float *dmaps, float *partials,
                                      das3r_stream_t stream) {
    if (H <= 0 || W <= 0 || !img1 || !img2 || !ssim_map || !dmaps || !partials) { set_error("das3r_ssim_map_forward: invalid argument"); return DAS3R_ERR_INVALID_ARG; }
    hipStream_t s = (hipStream_t)stream;
    DAS3R_LAUNCH(photometric_forward_kernel, dim3(div_up(W, PT), div_up(H, PT)), dim3(PT * PT), 0, s, H, W, img1, img2, (const float *)nullptr, 1.0f,
                 make_window(), partials, dmaps, ssim_map);
    KERNEL_CHECK(s, false, "ssim_map_forward");
    return DAS3R_OK;
}
extern "C" int das3r_ssim_map_backward(int32_t H, int32_t W, const float *img1, const float *img2, const float *dmaps, const float *grad_map,
                                       float *d_img1, float *d_img2, das3r_stream_t stream) {
    if (H <= 0 || W <= 0 || !img1 || !img2 || !dmaps || !grad_map || !d_img1 || !d_img2) { set_error("das3r_ssim_map_backward: invalid argument"); return DAS3R_ERR_INVALID_ARG; }
    hipStream_t s = (hipStream_t)stream;
    DAS3R_LAUNCH(photometric_backward_kernel, dim3(div_up(W, PT), div_up(H, PT)), dim3(PT * PT), 0, s, H, W, img1, img2, (const float *)nullptr, 1.0f,
                 make_window(), dmaps, (const float *)nullptr, d_img1, (float *)nullptr, (const float *)nullptr, 0, (float *)nullptr, grad_map, d_img2);
    KERNEL_CHECK(s, false, "ssim_map_backward");
    return DAS3R_OK;
}
