// pair_count.hip — measurement aid (bench.py): the LIVE (pixel, splat) pairs of a forward — the pairs that were
// blended: list position < n_contrib of the pixel, power <= 0, alpha >= 1/255 — counted from the forward's saved
// buffers by a plain pixel-per-lane walk with pair_alpha's arithmetic (render_common.h: the same decisions, bit for bit, as every
// compositing kernel).  What the compositing kernels EVALUATE per instance (das3r_pair_counters) over this number is their padded-work
// ratio; the walk a kernel would need for the live pairs alone prices its VALU roofline.  Not on any timed path.
#include "render_common.h"

namespace das3r {

__global__ void __launch_bounds__(256) count_live_pairs_kernel(const uint2 *__restrict__ ranges, const uint32_t *__restrict__ point_list, int W, int H,
                                                               int tiles_x, int ntiles, const float4 *__restrict__ xyh,
                                                               const float4 *__restrict__ conic_opacity, const uint32_t *__restrict__ n_contrib,
                                                               uint32_t last_g, uint32_t cap, unsigned long long *__restrict__ out /*[2]: live pairs, listed (pixel, entry) positions below n_contrib*/) {
    const int tile = blockIdx.x;
    if (tile >= ntiles) return;
    __shared__ float4 s_xy[256], s_co[256];
    const int tid = threadIdx.x;
    const int bx = tile % tiles_x, by = tile / tiles_x;
    const int px = bx * TILE_X + (tid & 15), py = by * TILE_Y + (tid >> 4);
    const bool inside = px < W && py < H;
    const uint32_t mine = inside ? n_contrib[(size_t)py * W + px] : 0u;
    const uint2 range = safe_range(ranges[tile], cap);
    const uint32_t len = range.y - range.x;
    unsigned long long live = 0ull;
    for (uint32_t base = 0; base < len; base += 256u) {
        __syncthreads();
        if (base + tid < len) {
            const uint32_t g = min(point_list[range.x + base + tid], last_g);
            s_xy[tid] = xyh[(size_t)g * SPLAT_REC];
            s_co[tid] = conic_opacity[(size_t)g * SPLAT_REC];
        }
        __syncthreads();
        const uint32_t n = min(256u, len - base);
        for (uint32_t j = 0; j < n && base + j < mine; j++) {
            float dx, dy, G, alpha;
            if (pair_alpha(s_xy[j].x, s_xy[j].y, s_co[j], (float)px, (float)py, dx, dy, G, alpha)) live++;
        }
    }
    unsigned long long below = (unsigned long long)min(mine, len);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        live += __shfl_xor(live, o, 64);
        below += __shfl_xor(below, o, 64);
    }
    if (__lane_id() == 0) {
        atomicAdd(out, live);
        atomicAdd(out + 1, below);
    }
}

int launch_count_live_pairs(const das3r_raster_args *a, char *geom, char *binning, char *img, const Layout &L, unsigned long long *out, hipStream_t s) {
    DAS3R_LAUNCH(count_live_pairs_kernel, dim3(L.ntiles), dim3(256), 0, s, (const uint2 *)(img + L.pub.ranges),
                 (const uint32_t *)(binning + L.pub.point_list), a->image_width, a->image_height, L.tiles_x, L.ntiles, (const float4 *)(geom + L.pub.xy),
                 (const float4 *)(geom + L.pub.conic_opacity), (const uint32_t *)(img + L.pub.n_contrib), (uint32_t)(a->P - 1), (uint32_t)L.capacity, out);
    KERNEL_CHECK(s, a->debug, "count_live_pairs");
    return DAS3R_OK;
}

}  // namespace das3r
