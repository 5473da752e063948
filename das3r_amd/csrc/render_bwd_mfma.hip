// render_bwd_mfma.hip — K7 with the 64-pixel reductions on the matrix cores (chosen for very long tile lists: render_bwd.hip).
//
// The wave-per-quadrant backward (render_bwd.hip) spends ~30 % of its instructions reducing nine values over the 64 lanes of
// a wave for every (splat, quadrant) pair, and another ~15 on per-lane products that only exist to be reduced.  All nine sums
// are FIXED-weight linear functionals of two per-pixel scalars of the pair:
//     w   = alpha_i T_i                       (d pixel colour / d splat colour)
//     gdl = G dL/dalpha
//   dL/dcolour_c               = sum_p w_p   dLp_c[p]                   (dLp = dL/dpixel: constant per pixel for the whole tile)
//   dL/dopacity, mean2D, conic = combinations of the MOMENTS  sum_p gdl_p {1, u_p, v_p, u_p^2, u_p v_p, v_p^2}
// with (u, v) = pixel position relative to the tile centre (dx = X - u with X = splat x relative to the tile centre, so e.g.
// sum gdl dx^2 = X^2 M0 - 2 X Mu + Muu; the conversion happens once per (splat, tile) when the batch is written out).
// That is a [9 x 64] x [64 x columns] matrix product: v_mfma_f32_16x16x4_f32 (an exact fp32 FMA chain), 16 instructions per
// chunk of 8 pairs — rows = the nine weight vectors of a pixel, columns = (w of pair 0..7 | gdl of pair 0..7); the products of
// w with moment rows and of gdl with colour rows are computed and ignored.  Per chunk the lanes evaluate eight alphas back to
// back (independent: instruction-level parallelism instead of occupancy), run the eight (T, R) recurrences in order and drop
// their two scalars per pair into a wave-private LDS slab laid out so that four ds_read_b128 give a lane its sixteen B
// operands (MFMA sums over its K dimension = DPP row x instruction index, never over the lanes of a row: the pixel-per-lane
// values have to be transposed, and LDS is the cheap way).
//
// Measured (MI355X): VALU instructions per pair 78 -> 48, but the slab costs LDS (44 KB per workgroup, 3 workgroups per CU
// instead of 7) and the kernel becomes latency-bound: 0.60 -> 0.82 ms at 1 M splats / 1080p (a few hundred list entries per
// tile), 2.06 -> 1.67 ms on the 5 M-splat DAS3R-shaped scene (14 k entries per tile).  Hence the length-based switch.
#include "render_common.h"

namespace das3r {

constexpr int NACC = 9;     // C0, C1, C2, M0, Mu, Mv, Muu, Muv, Mvv
constexpr int CHUNK = 8;    // pairs per MFMA chunk (16 columns: w | gdl)
constexpr int MB = 256;     // splats per staged batch (smaller than the other kernels': LDS is what limits the waves per SIMD here)
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) render_backward_mfma_kernel(
    const uint2 *__restrict__ ranges, const uint32_t *__restrict__ point_list, int W, int H, int tiles_x, int ntiles_strip /*render_common.h pack_tiles*/,
    const float4 *__restrict__ xyh, const float4 *__restrict__ conic_opacity, const float4 *__restrict__ rgbd,
    const float *__restrict__ bg, const float *__restrict__ final_T, const uint32_t *__restrict__ n_contrib,
    const float *__restrict__ dL_dpix, const uint32_t *__restrict__ slot_list, float *__restrict__ partial /*[I,9]*/,
    uint32_t last_g, uint32_t cap /*bounds of the list contents: render_common.h safe_range*/) {
    __shared__ StagedSplat stage[MB];
    __shared__ uint32_t s_slot[MB];
    __shared__ float acc[MB * NACC];
    // [wave][DPP row k][column n][t]: value of pixel 16k + t in column n.  Columns are 20 floats apart (16-byte aligned, and the
    // sixteen lanes of a ds_read_b128 spread over all banks), rows 328 (their 16-lane stores land 8 banks apart)
    constexpr int COLS = 20, ROWS = 16 * COLS + 8;
    __shared__ __attribute__((aligned(16))) float slab[4][4 * ROWS];
    __shared__ uint32_t chunk_j[4][CHUNK];    // staged index of every pair of the wave's current chunk
    __shared__ uint32_t s_max[4];

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
    const float tcx = (float)(bx * TILE_X) + 7.5f, tcy = (float)(by * TILE_Y) + 7.5f;   // tile centre
    const uint2 range = safe_range(ranges[tile], cap);
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

    // ---- A operands: lane (i + 16k) of instruction t supplies weight i of pixel 16k + t (constant for the whole tile) ----
    // rows 0..2 = dL/dpixel of that pixel (fetched from its lane through LDS), rows 3..8 = moment weights of its position
    float A[16];
    {
        float(*dl)[4] = reinterpret_cast<float(*)[4]>(&slab[wave][0]);   // scratch: [pixel][channel]
        dl[lane][0] = dLp0;
        dl[lane][1] = dLp1;
        dl[lane][2] = dLp2;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        const int i = lane & 15, k = lane >> 4;
#pragma unroll
        for (int t = 0; t < 16; t++) {
            const int p = 16 * k + t;                       // pixel (= lane) whose weights this register holds
            const float u = (float)(bx * TILE_X + ((wave & 1) << 3) + (p & 7)) - tcx;
            const float v = (float)(by * TILE_Y + ((wave >> 1) << 3) + (p >> 3)) - tcy;
            float a = 0.f;
            if (i < 3) a = dl[p][i];
            else if (i == 3) a = 1.f;
            else if (i == 4) a = u;
            else if (i == 5) a = v;
            else if (i == 6) a = u * u;
            else if (i == 7) a = u * v;
            else if (i == 8) a = v * v;
            A[t] = a;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    }

    uint32_t mx = last_contributor;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, o, 64));
    if (lane == 0) s_max[wave] = mx;
    __syncthreads();
    const uint32_t max_contrib = min(max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3])), range.y - range.x);
    const int rounds = ((int)max_contrib + MB - 1) / MB;
    {   // list entries beyond max_contrib receive no gradient from this tile: their partial rows are zero
        const uint32_t len = range.y - range.x;
        const uint32_t ntail = (len - max_contrib) * NACC;
        for (uint32_t f = tid; f < ntail; f += TILE_PIX) {
            const uint32_t t = f / NACC, q = f - t * NACC;
            partial[(size_t)min(slot_list[range.x + max_contrib + t], cap - 1u) * NACC + q] = 0.f;
        }
    }

    ReplayState st = {T_final, 0.f};
    const float tfbg = T_final * bg_dot_dpixel;
    // where this lane's two scalars of a pair go: slab[wave][lane / 16][column][lane % 16]
    float *my_col0 = &slab[wave][(lane >> 4) * ROWS + (lane & 15)];   // + column * COLS
    // which accumulators this lane feeds after the MFMAs: it holds D[4 * (lane / 16) + r][n = lane % 16]
    const int col = lane & 15, kq = lane >> 4;
    const bool wcol = col < CHUNK;   // columns 0..7: w of pair n (colour rows), 8..15: gdl of pair n - 8 (moment rows)

    auto flush = [&](const int m /*pairs in the chunk*/) {
        for (int n = m; n < CHUNK; n++) {   // unused columns of a short chunk
            my_col0[n * COLS] = 0.f;
            my_col0[(CHUNK + n) * COLS] = 0.f;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        const v4f *src = reinterpret_cast<const v4f *>(&slab[wave][kq * ROWS + col * COLS]);   // 16 consecutive floats: B operands t = 0..15
        const v4f b0 = src[0], b1 = src[1], b2 = src[2], b3 = src[3];
        const uint32_t j = chunk_j[wave][col & (CHUNK - 1)];
        v4f d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
        d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[0], b0[0], d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[1], b0[1], d1, 0, 0, 0);
        d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[2], b0[2], d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[3], b0[3], d1, 0, 0, 0);
        d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[4], b1[0], d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[5], b1[1], d1, 0, 0, 0);
        d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[6], b1[2], d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[7], b1[3], d1, 0, 0, 0);
        d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[8], b2[0], d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[9], b2[1], d1, 0, 0, 0);
        d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[10], b2[2], d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[11], b2[3], d1, 0, 0, 0);
        d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[12], b3[0], d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[13], b3[1], d1, 0, 0, 0);
        d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[14], b3[2], d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[15], b3[3], d1, 0, 0, 0);
        const v4f d = d0 + d1;
        const bool real = (col & (CHUNK - 1)) < m;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int q = 4 * kq + r;   // row of D = which of the nine sums
            const bool mine = real && (wcol ? q < 3 : (q >= 3 && q < NACC));
            if (mine) atomicAdd(&acc[j * NACC + q], d[r]);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");   // the slab is rewritten by the next chunk
    };

    for (int i = 0; i < rounds; i++) {
        const int done_before = i * MB;
        const int n = min(MB, (int)max_contrib - done_before);
        // stage the batch in reverse list order; entry j holds list position (max_contrib - 1 - done_before - j)
        if (tid < n) {
            const uint32_t pos = range.x + max_contrib - 1 - done_before - tid;
            const uint32_t g = min(point_list[pos], last_g);
            s_slot[tid] = min(slot_list[pos], cap - 1u);
            stage[tid].xyh = xyh[(size_t)g * SPLAT_REC];
            stage[tid].co = conic_opacity[(size_t)g * SPLAT_REC];
            stage[tid].rgbd = rgbd[(size_t)g * SPLAT_REC];
        }
        for (int f = tid; f < MB * NACC; f += TILE_PIX) acc[f] = 0.f;
        __syncthreads();

        uint64_t masks[MB / 64];
#pragma unroll
        for (int k = 0; k < MB / 64; k++) {
            const int s = k * 64 + lane;
            masks[k] = __ballot(s < n && quadrant_hit(stage[s].xyh, qcx, qcy));
        }
        // A whole chunk per trip: collect eight surviving splats, evaluate their alphas back to back (independent: the
        // instruction-level parallelism hides the latency a single pair's chain would expose at four waves per SIMD), then run
        // the eight (T, R) recurrences in order and multiply.
        int kcur = 0;
        uint64_t mk = masks[0];
        auto advance = [&]() -> int {   // next surviving staged index, or -1
            while (mk == 0ull) {
                if (++kcur >= MB / 64) return -1;
                mk = kcur == 1 ? masks[1] : (kcur == 2 ? masks[(MB / 64) > 2 ? 2 : 1] : masks[(MB / 64) > 3 ? 3 : 1]);
            }
            const int j = kcur * 64 + __builtin_ctzll(mk);
            mk &= mk - 1ull;
            return j;
        };
        while (true) {
            int js[CHUNK];
#pragma unroll
            for (int u = 0; u < CHUNK; u++) js[u] = advance();
            if (js[0] < 0) break;
            float al[CHUNK], Gs[CHUNK], cds[CHUNK];
            bool act[CHUNK];
#pragma unroll
            for (int u = 0; u < CHUNK; u++) {
                const int jc = js[u] < 0 ? 0 : js[u];
                const float4 p = stage[jc].xyh, co = stage[jc].co, c = stage[jc].rgbd;
                float dx, dy;
                act[u] = pair_alpha(p.x, p.y, co, pxf, pyf, dx, dy, Gs[u], al[u]) &
                         ((uint32_t)(max_contrib - 1 - done_before - jc) < last_contributor) & (js[u] >= 0);
                cds[u] = c.x * dLp0 + c.y * dLp1 + c.z * dLp2;
                if (lane == u) chunk_j[wave][u] = (uint32_t)jc;
            }
            int m_valid = 0;
#pragma unroll
            for (int u = 0; u < CHUNK; u++) {
                // (T, R) recurrence of render_common.h: replay_pair without its products
                const float am = act[u] ? al[u] : 0.f, Gm = act[u] ? Gs[u] : 0.f;
                const float rinv = __builtin_amdgcn_rcpf(1.f - am);
                st.T = st.T * rinv;
                const float w = am * st.T;
                const float dL_dalpha = st.T * cds[u] - (st.R + tfbg) * rinv;
                st.R = st.R + cds[u] * w;
                my_col0[u * COLS] = w;
                my_col0[(CHUNK + u) * COLS] = Gm * dL_dalpha;
                m_valid += js[u] >= 0 ? 1 : 0;
            }
            flush(m_valid);
        }
        __syncthreads();
        // moments -> the nine per-instance sums, one staged splat per thread; rows go to the emission slot
        if (tid < n) {
            const float4 p = stage[tid].xyh;
            const float4 co = stage[tid].co;
            const float *a = &acc[tid * NACC];
            const float M0 = a[3], Mu = a[4], Mv = a[5], Muu = a[6], Muv = a[7], Mvv = a[8];
            const float X = p.x - tcx, Y = p.y - tcy, kk = -0.5f * co.w;
            const float Sgx = kk * (X * M0 - Mu), Sgy = kk * (Y * M0 - Mv);
            float *row = partial + (size_t)s_slot[tid] * NACC;
            row[0] = a[0];
            row[1] = a[1];
            row[2] = a[2];
            row[3] = (Sgx * co.x + Sgy * co.y) * (float)W;        // 2 * ddelx_dx = W
            row[4] = (Sgy * co.z + Sgx * co.y) * (float)H;
            row[5] = kk * (X * X * M0 - 2.f * X * Mu + Muu);
            row[6] = kk * (X * Y * M0 - X * Mv - Y * Mu + Muv);
            row[7] = kk * (Y * Y * M0 - 2.f * Y * Mv + Mvv);
            row[8] = M0;
        }
        __syncthreads();
    }
}

int launch_render_backward_mfma(const das3r_raster_args *a, const float *dL_dpix, char *geom, char *binning, char *img, const Layout &L,
                                float *partial, hipStream_t s) {
    DAS3R_LAUNCH(render_backward_mfma_kernel, dim3(xcd_grid(L)), dim3(TILE_PIX), 0, s, (const uint2 *)(img + L.pub.ranges),
                 (const uint32_t *)(binning + L.pub.point_list), a->image_width, a->image_height, L.tiles_x, pack_tiles(L),
                 (const float4 *)(geom + L.pub.xy), (const float4 *)(geom + L.pub.conic_opacity),
                 (const float4 *)(geom + L.pub.rgbd), a->bg, (const float *)(img + L.pub.final_T),
                 (const uint32_t *)(img + L.pub.n_contrib), dL_dpix, (const uint32_t *)(binning + L.b_slot), partial,
                 (uint32_t)(a->P - 1), (uint32_t)L.capacity);
    KERNEL_CHECK(s, a->debug, "render_backward_mfma");
    return DAS3R_OK;
}

}  // namespace das3r
