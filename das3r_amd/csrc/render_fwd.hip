// render_fwd.hip — K6: per-pixel front-to-back alpha compositing of a tile's depth-ordered splat list.
// Replaces upstream:cuda_rasterizer/forward.cu renderCUDA.  Decomposition: see render_common.h
// (tile per workgroup, 8x8 quadrant per wave64, LDS-staged batches, ballot-culled per-wave sub-lists).
#include "render_common.h"

namespace das3r {

__global__ void __launch_bounds__(256) render_forward_kernel(const uint2 *__restrict__ ranges, const uint32_t *__restrict__ point_list,
                                                             int W, int H, int tiles_x, int ntiles_strip /*render_common.h pack_tiles*/, const float4 *__restrict__ xyh,
                                                             const float4 *__restrict__ conic_opacity, const float4 *__restrict__ rgbd,
                                                             const float *__restrict__ bg, float *__restrict__ final_T,
                                                             uint32_t *__restrict__ n_contrib, float *__restrict__ out_color, const LocalBin lb,
                                                             unsigned long long *__restrict__ pairs /*common.h pair_counters()*/) {
    __shared__ StagedSplat stage[TILE_PIX];
    __shared__ uint32_t s_gid[LOCAL_MAX];   // local depth order: the tile's sorted list
    const int ntiles = packed_ntiles(ntiles_strip);
    const int tile = xcd_tile(blockIdx.x, ntiles_strip, tiles_x);
    if (tile < 0) return;
    const int tid = threadIdx.x, lane = __lane_id(), wave = tid >> 6;
    const int bx = tile % tiles_x, by = tile / tiles_x;
    int px, py;
    quadrant_pixel(bx, by, wave, lane, px, py);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const float qcx = (float)(bx * TILE_X + ((wave & 1) << 3)) + 3.5f, qcy = (float)(by * TILE_Y + ((wave >> 1) << 3)) + 3.5f;
    const uint2 range = safe_range(ranges[tile], lb.cap);
    const bool sorted_here = lb.point_list != nullptr && (int)(range.y - range.x) <= LOCAL_MAX;   // (uniform)
    // local depth order: sort this tile's list first (a list of one batch is staged by the sort itself)
    const bool prestaged = lb.point_list != nullptr && local_order_tile(lb, range, xyh, conic_opacity, rgbd, stage, s_gid, threadIdx.x);
    int toDo = (int)(range.y - range.x);
    const int rounds = (toDo + TILE_PIX - 1) / TILE_PIX;

    // Per-lane state and decisions are kept in VECTOR registers and taken with compare + select pairs (the compare's mask is
    // consumed at once): a boolean carried across instructions lives in a scalar register pair, every && / || on it is an
    // instruction of the CU's single scalar ALU, and this kernel issued 550 of those per wave and tile against 700 vector
    // instructions spread over four SIMDs — it was bound by the scalar unit (1.8e7 SALU instructions per launch at 100 k splats =
    // 29 us of 45).  `live` (1 / 0) replaces the `done` flag; the transmittance itself says when a pixel stops:
    // T never drops below 1e-4 while a lane is live, so test_T < 1e-4 can only come from a pair that contributes.
    // Round 3: not even selects — blend_pair (render_common.h) takes the decisions by arithmetic.
    PixelBlend pb = {1.0f, 0.f, 0.f, 0.f, inside ? 1.f : 0.f, -1.0f};
    float &live = pb.live, &T = pb.T, &C0 = pb.C0, &C1 = pb.C1, &C2 = pb.C2;
    uint32_t last_contributor = 0;

    const int nb = ckpt_buckets(range);                                 // (> 1: a long list, checkpointed for the bucket-parallel backward)
    const int cpix = ((py - by * TILE_Y) << 4) + (px - bx * TILE_X);      // pixel's place in a checkpoint slot
    int next_slot = 0;
    int visits = 0;   // (wave-uniform) (splat, quadrant) visits = 64 pairs each
    for (int i = 0; i < rounds; i++, toDo -= TILE_PIX) {
        if (__syncthreads_count(live == 0.f) == TILE_PIX) break;
        if (nb > 1 && i > 0 && (i * TILE_PIX) % BUCKET == 0) ckpt_slot(lb.ckpt, range, tile, next_slot++)[cpix] = make_float4(T, C0, C1, C2);
        const uint32_t progress = range.x + i * TILE_PIX + tid;
        if (progress < range.y && !prestaged) {
            const uint32_t g = min(sorted_here ? s_gid[i * TILE_PIX + tid] : (lb.point_list ? lb.point_list[progress] : point_list[progress]), lb.last_g);
            stage[tid].xyh = xyh[(size_t)g * SPLAT_REC];           // one 64-byte record: a single cache line per splat
            stage[tid].co = conic_opacity[(size_t)g * SPLAT_REC];
            stage[tid].rgbd = rgbd[(size_t)g * SPLAT_REC];
        }
        __syncthreads();
        const int n = toDo < TILE_PIX ? toDo : TILE_PIX;
        blend_batch_begin(pb);
        // wave-level cull of the batch against this wave's quadrant: 4 splats per lane, one ballot each
        uint64_t masks[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int s = k * 64 + lane;
            masks[k] = __ballot(s < n && quadrant_hit(stage[s].xyh, qcx, qcy));
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            uint64_t m = masks[k];
            if (m != 0ull && __ballot(live != 0.f) == 0ull) break;   // every pixel of the quadrant has stopped (checked per 64 entries)
            visits += __popcll(m);
            while (m != 0ull) {
                const int j = k * 64 + __builtin_ctzll(m);
                m &= m - 1ull;
                const float4 p = stage[j].xyh;
                const float4 co = stage[j].co;
                const float4 c = lds_read4(&stage[j].rgbd);   // (b128, not b96: half the LDS cycles)
                const float dx = p.x - pxf, dy = p.y - pyf;
                const float q = __fmaf_rn(__fmul_rn(co.x, dx), dx, __fmul_rn(__fmul_rn(co.z, dy), dy));
                const float power = __fmaf_rn(-0.5f, q, -__fmul_rn(__fmul_rn(co.y, dx), dy));   // (pair_alpha's arithmetic, its
                const float a1 = fminf(0.99f, __fmul_rn(co.w, __expf(power)));                    // two tests by arithmetic)
                blend_pair(pb, alpha_if_visible(a1, power), c, (float)j);
            }
        }
        last_contributor = blend_batch_end(pb, last_contributor, (uint32_t)(i * TILE_PIX));
    }
    for (; nb > 1 && next_slot < nb; next_slot++) ckpt_slot(lb.ckpt, range, tile, next_slot)[cpix] = make_float4(T, C0, C1, C2);   // (early exit: nothing changes any more)
    if (inside) {
        const size_t pix = (size_t)py * W + px, plane = (size_t)H * W;
        final_T[pix] = T;
        n_contrib[pix] = last_contributor;
        out_color[pix] = C0 + T * bg[0];
        out_color[plane + pix] = C1 + T * bg[1];
        out_color[2 * plane + pix] = C2 + T * bg[2];
    }
    if (pairs != nullptr && lane == 0 && visits > 0) {
        atomicAdd(pairs, (unsigned long long)visits * 64ull);
        atomicAdd(pairs + 2, (unsigned long long)visits);
    }
}

int launch_render_forward(const das3r_raster_args *a, const float *colors_precomp, float *out_color, char *geom, char *binning,
                          char *img, const Layout &L, const LocalBin &lb, hipStream_t s) {
    (void)colors_precomp;  // precomputed colours were copied into rgbd by the preprocess kernel
    if (use_quad_lanes(L, lb))
    {   // one workgroup per tile (lanes), or four (regions) where the tile lists are skewed: the host has been told by the forwards before this one
        const int f = switches().render_fwd;
        if (f == 4) return launch_render_forward_slices(a, out_color, geom, binning, img, L, lb, s);
        if (f == 5 || (f == 0 && lb.prefer_regions)) return launch_render_forward_regions(a, out_color, geom, binning, img, L, lb, s);
        return launch_render_forward_lanes(a, out_color, geom, binning, img, L, lb, s);
    }
    if (use_row_private(L.capacity, L.ntiles)) return launch_render_forward_rows(a, out_color, geom, binning, img, L, lb, s);
    const int pad_lds = switches().fwd_pad_lds;   // occupancy experiments
    DAS3R_LAUNCH(render_forward_kernel, dim3(xcd_grid(L)), dim3(TILE_PIX), pad_lds, s, (const uint2 *)(img + L.pub.ranges),
                 (const uint32_t *)(binning + L.pub.point_list), a->image_width, a->image_height, L.tiles_x, pack_tiles(L),
                 (const float4 *)(geom + L.pub.xy), (const float4 *)(geom + L.pub.conic_opacity),
                 (const float4 *)(geom + L.pub.rgbd), a->bg, (float *)(img + L.pub.final_T), (uint32_t *)(img + L.pub.n_contrib),
                 out_color, lb, pair_counters());
    KERNEL_CHECK(s, a->debug, "render_forward");
    return DAS3R_OK;
}

}  // namespace das3r
