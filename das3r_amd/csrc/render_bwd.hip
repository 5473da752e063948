// render_bwd.hip — K7: back-to-front replay of the per-pixel compositing, accumulating
// dL/d(colour, mean2D, conic, opacity) per splat.  Replaces upstream:cuda_rasterizer/backward.cu renderCUDA
// (SURVEY.md A.7), whose 9 float atomicAdds per (pixel, splat) pair are the classic 3DGS training hot spot.
//
// MI355X design: the 64 pixels of a wavefront are pre-reduced on the DPP network (wave_sum_to_lane63: 6 VALU ops per
// value, no LDS traffic), the 4 waves of the tile are combined with one LDS float-add each, and only ONE global
// atomic per (tile, splat, component) leaves the CU — 256x fewer device atomics than one per pair.  Splats that no
// lane of a wave touches are skipped with a single wave vote.
#include "render_common.h"

namespace das3r {

constexpr int NACC = 9;  // dcolor[3], dmean2D[2], dconic[3], dopacity

template <bool USE_DPP>
__global__ void __launch_bounds__(256) render_backward_kernel(
    const uint2 *__restrict__ ranges, const uint32_t *__restrict__ point_list, int W, int H, int tiles_x, int ntiles,
    const float2 *__restrict__ xy, const float4 *__restrict__ conic_opacity, const float4 *__restrict__ rgbd,
    const float *__restrict__ bg, const float *__restrict__ final_T, const uint32_t *__restrict__ n_contrib,
    const float *__restrict__ dL_dpix, float *__restrict__ dL_dmean2D /*[P,3]*/, float *__restrict__ dL_dconic /*[P,*] stride cs*/,
    int conic_stride, float *__restrict__ dL_dopacity /*[P]*/, float *__restrict__ dL_dcolor /*[P,*] stride ls*/, int color_stride, int ablate) {
    __shared__ StagedSplat stage[TILE_PIX];
    __shared__ uint32_t stage_id[TILE_PIX];
    __shared__ float acc[TILE_PIX][NACC];
    __shared__ uint32_t s_max[4];

    const int tile = xcd_tile(blockIdx.x, ntiles);
    if (tile < 0) return;
    const int tid = threadIdx.x, lane = __lane_id(), wave = tid >> 6;
    const int bx = tile % tiles_x, by = tile / tiles_x;
    const int px = bx * TILE_X + (tid & 15), py = by * TILE_Y + (tid >> 4);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const uint2 range = ranges[tile];
    const size_t pix = (size_t)py * W + px, plane = (size_t)H * W;

    const float T_final = inside ? final_T[pix] : 0.f;
    const uint32_t last_contributor = inside ? n_contrib[pix] : 0u;
    float dLp0 = 0.f, dLp1 = 0.f, dLp2 = 0.f;
    if (inside) {
        dLp0 = dL_dpix[pix];
        dLp1 = dL_dpix[plane + pix];
        dLp2 = dL_dpix[2 * plane + pix];
    }
    const float bg_dot_dpixel = bg[0] * dLp0 + bg[1] * dLp1 + bg[2] * dLp2;
    const float ddelx_dx = 0.5f * W, ddely_dy = 0.5f * H;

    // no pixel of this tile blended anything past list position max_contrib: start the replay there
    uint32_t m = last_contributor;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
    if (lane == 0) s_max[wave] = m;
    __syncthreads();
    const uint32_t max_contrib = max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3]));
    const int rounds = ((int)max_contrib + TILE_PIX - 1) / TILE_PIX;

    float T = T_final;
    float accum0 = 0.f, accum1 = 0.f, accum2 = 0.f, last_alpha = 0.f, lc0 = 0.f, lc1 = 0.f, lc2 = 0.f;

    for (int i = 0; i < rounds; i++) {
        const int done_before = i * TILE_PIX;
        const int n = min(TILE_PIX, (int)max_contrib - done_before);
        // stage the batch in reverse list order; entry j holds list position (max_contrib - 1 - done_before - j)
        if (tid < n) {
            const uint32_t g = point_list[range.x + max_contrib - 1 - done_before - tid];
            stage_id[tid] = g;
            stage[tid].xy = xy[g];
            stage[tid].co = conic_opacity[g];
            stage[tid].rgbd = rgbd[g];
        }
#pragma unroll
        for (int k = 0; k < NACC; k++) acc[tid][k] = 0.f;
        __syncthreads();

        for (int j = 0; j < n; j++) {
            const uint32_t position = max_contrib - 1 - done_before - j;  // 0-based list position
            float dx, dy, G, alpha;
            const bool active = (position < last_contributor) && pair_alpha(stage[j].xy, stage[j].co, pxf, pyf, dx, dy, G, alpha);
            if (__ballot(active) == 0ull) continue;  // wave-uniform skip
            float v[NACC];
#pragma unroll
            for (int k = 0; k < NACC; k++) v[k] = 0.f;
            if (active) {
                const float4 co = stage[j].co;
                const float4 c = stage[j].rgbd;
                T = T / (1.f - alpha);
                const float dchannel_dcolor = alpha * T;
                float dL_dalpha = 0.f;
                accum0 = last_alpha * lc0 + (1.f - last_alpha) * accum0;
                lc0 = c.x;
                dL_dalpha += (c.x - accum0) * dLp0;
                accum1 = last_alpha * lc1 + (1.f - last_alpha) * accum1;
                lc1 = c.y;
                dL_dalpha += (c.y - accum1) * dLp1;
                accum2 = last_alpha * lc2 + (1.f - last_alpha) * accum2;
                lc2 = c.z;
                dL_dalpha += (c.z - accum2) * dLp2;
                v[0] = dchannel_dcolor * dLp0;
                v[1] = dchannel_dcolor * dLp1;
                v[2] = dchannel_dcolor * dLp2;
                dL_dalpha *= T;
                last_alpha = alpha;
                dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot_dpixel;
                const float dL_dG = co.w * dL_dalpha;
                const float gdx = G * dx, gdy = G * dy;
                const float dG_ddelx = -gdx * co.x - gdy * co.y;
                const float dG_ddely = -gdy * co.z - gdx * co.y;
                v[3] = dL_dG * dG_ddelx * ddelx_dx;
                v[4] = dL_dG * dG_ddely * ddely_dy;
                v[5] = -0.5f * gdx * dx * dL_dG;
                v[6] = -0.5f * gdx * dy * dL_dG;
                v[7] = -0.5f * gdy * dy * dL_dG;
                v[8] = G * dL_dalpha;
            }
            if (ablate & 2) {
            } else if (USE_DPP) {
#pragma unroll
                for (int k = 0; k < NACC; k++) v[k] = wave_sum_to_lane63(v[k]);
                if (lane == 63) {
#pragma unroll
                    for (int k = 0; k < NACC; k++) atomicAdd(&acc[j][k], v[k]);
                }
            } else {
#pragma unroll
                for (int k = 0; k < NACC; k++) v[k] = wave_sum_shfl(v[k]);
                if (lane == 0) {
#pragma unroll
                    for (int k = 0; k < NACC; k++) atomicAdd(&acc[j][k], v[k]);
                }
            }
        }
        __syncthreads();
        // one global atomic per (tile, splat, component)
        if (tid < n) {
            const uint32_t g = stage_id[tid];
            float a[NACC];
            bool any = false;
#pragma unroll
            for (int k = 0; k < NACC; k++) {
                a[k] = acc[tid][k];
                any |= (a[k] != 0.f);
            }
            if (any && !(ablate & 1)) {
                unsafeAtomicAdd(&dL_dcolor[(size_t)g * color_stride + 0], a[0]);
                unsafeAtomicAdd(&dL_dcolor[(size_t)g * color_stride + 1], a[1]);
                unsafeAtomicAdd(&dL_dcolor[(size_t)g * color_stride + 2], a[2]);
                unsafeAtomicAdd(&dL_dmean2D[(size_t)g * 3 + 0], a[3]);
                unsafeAtomicAdd(&dL_dmean2D[(size_t)g * 3 + 1], a[4]);
                unsafeAtomicAdd(&dL_dconic[(size_t)g * conic_stride + 0], a[5]);
                unsafeAtomicAdd(&dL_dconic[(size_t)g * conic_stride + 1], a[6]);
                unsafeAtomicAdd(&dL_dconic[(size_t)g * conic_stride + 2], a[7]);
                unsafeAtomicAdd(&dL_dopacity[g], a[8]);
            }
        }
        __syncthreads();
    }
}

int launch_render_backward(const das3r_raster_args *a, const float *dL_dpix, char *geom, char *binning, char *img, const Layout &L,
                           float *dL_dmean2D, float *dL_dconic, float *dL_dopacity, float *dL_dcolor, int color_stride,
                           hipStream_t s) {
    const char *e = getenv("DAS3R_BWD_REDUCE");  // "shfl" selects the ds_bpermute reference reduction (diagnostics)
    const bool use_dpp = !(e && e[0] == 's');
    const char *ea = getenv("DAS3R_ABLATE");  // perf experiments only: bit0 = no global atomics, bit1 = no wave reduction
    const int ablate = ea ? atoi(ea) : 0;
#define ARGS                                                                                                                 \
    (const uint2 *)(img + L.pub.ranges), (const uint32_t *)(binning + L.pub.point_list), a->image_width, a->image_height,    \
        L.tiles_x, L.ntiles, (const float2 *)(geom + L.pub.xy), (const float4 *)(geom + L.pub.conic_opacity),                \
        (const float4 *)(geom + L.pub.rgbd), a->bg, (const float *)(img + L.pub.final_T),                                    \
        (const uint32_t *)(img + L.pub.n_contrib), dL_dpix, dL_dmean2D, dL_dconic, 8, dL_dopacity, dL_dcolor,               \
        color_stride, ablate
    if (use_dpp) DAS3R_LAUNCH((render_backward_kernel<true>), dim3(xcd_grid(L.ntiles)), dim3(TILE_PIX), 0, s, ARGS);
    else DAS3R_LAUNCH((render_backward_kernel<false>), dim3(xcd_grid(L.ntiles)), dim3(TILE_PIX), 0, s, ARGS);
#undef ARGS
    KERNEL_CHECK(s, a->debug, "render_backward");
    return DAS3R_OK;
}

}  // namespace das3r
