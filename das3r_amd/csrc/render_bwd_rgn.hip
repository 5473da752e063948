// render_bwd_rgn.hip — K7 for Gaussians of a few pixels (the DAS3R training shape): every 16-lane DPP row of a wave walks the culled list
// of a 2x2 pixel REGION — lanes = 16 entries of that list, time = the region's four pixels.
// Replaces upstream:cuda_rasterizer/backward.cu renderCUDA like the other backward kernels; same inputs, same partial rows.
//
// Why (round 6; docs/ledger.md (be)).  render_bwd_blk.hip culls per 4x4 block: a Gaussian of the DAS3R shape (one per pixel
// of every frame, live on 8.7 pixels) is listed for 2.9 - 4.4 blocks = 46 - 70 (pixel, entry) pairs, 47 - 67 are evaluated per instance and the
// kernel spends its time on pairs whose alpha is below 1/255 (padded work 5.4 - 7.7 x).  The same entry reaches 4 - 5.9 regions of 2x2 pixels =
// 16 - 24 pairs.  The decomposition of the block walk is kept — a wave = one 8x8 quadrant, wave-private 36-byte records per staged entry, the
// four rows of a wave adding to them one after the other, bucket-parallel replay from the forward's checkpoints, fixed order of every sum —
// and turned once more:
//   * a quadrant's sixteen regions are walked as four STRIPS of four (strip B = pixel rows 2 B, 2 B + 1 of the quadrant; row r of the wave
//     takes region r of the strip): sixteen lists per wave instead of four, built from ONE sixteen-bit reach mask per entry
//     (render_common.h region_mask: the forward's test of render_regions.hip) with the region's own last contributor folded in;
//   * a batch = sixteen entries x the region's FOUR pixels (block walk: sixteen): the two recurrences are the same DPP row scans, the
//     four pixel steps interleaved; the power's terms are shared between the pixels of a column / a row of the region (two dx, two dy);
//   * the geometry sums of a (region, entry) are formed directly about the splat centre from the four g = G dL/dalpha (no moments about a
//     corner, no cancellation): 18 instructions per batch;
//   * lane 16 r + s owns pixel (2 r + (s & 1), 2 (s >> 2) + ((s >> 1) & 1)) of the quadrant: its replay state (T, R) stays in its registers and
//     is read with row_newbcast:4 B + U by the walk of strip B; the constants (dL/dpixel, last contributor relative to the round) come out of
//     LDS as one broadcast ds_read_b128 per step (render_blk.h PIX = 1).
// Same arithmetic per pair as render_common.h pair_alpha, bit for bit: every pair takes the decision the forward took.
#include "render_blk.h"
#include <type_traits>

namespace das3r {

// The four pixel steps of strip B for the sixteen entries of a row's batch.  rxf, ryf: the region's corner pixel.  -> the nine sums of the
// (region, entry) pairs: lo4 = dL/dcolour (3), sum g; hi4 = sum g dx, g dy, g dx^2, g dx dy; syy = sum g dy^2  (dx = splat - pixel).
template <int B, bool MUT>
__device__ __forceinline__ void region_steps(const SplatRegs &sp, const float rxf, const float ryf, float &pT, float &pR, const char *cst,
                                             v4f &lo4, v4f &hi4, float &syy) {
    const float dx[2] = {sp.x - rxf, sp.x - (rxf + 1.0f)};
    const float dy[2] = {sp.y - ryf, sp.y - (ryf + 1.0f)};
    const float ax[2] = {__fmul_rn(sp.A, dx[0]), __fmul_rn(sp.A, dx[1])};
    const float bx[2] = {__fmul_rn(sp.B, dx[0]), __fmul_rn(sp.B, dx[1])};
    const float cyy[2] = {__fmul_rn(__fmul_rn(sp.C, dy[0]), dy[0]), __fmul_rn(__fmul_rn(sp.C, dy[1]), dy[1])};
    float am[4], Gm[4], rinv[4], Pinc[4], T[4], cd[4], w[4], wc[4], Sinc[4], Rinc[4], g[4];
    float4 pc[4];   // {d0, d1, d2, lastrel} of the step's pixel
#define ALPHA_STEP(U)                                                                                                         \
    {                                                                                                                         \
        pc[U] = *reinterpret_cast<const float4 *>(cst + U * 16);                                                              \
        const float q = __fmaf_rn(ax[U & 1], dx[U & 1], cyy[U >> 1]);                                                         \
        const float power = __fmaf_rn(-0.5f, q, -__fmul_rn(bx[U & 1], dy[U >> 1]));                                           \
        const float G = MUT ? __expf(power) * 1.0001f : __expf(power);                                                        \
        const float a1 = fminf(fminf(0.99f, __fmul_rn(sp.o, G)), pc[U].w - sp.posrel);   /* (render_blk.h block_row) */       \
        BLK_SELECT(a1, power, G, am[U], Gm[U])                                                                                \
        rinv[U] = __builtin_amdgcn_rcpf(1.f - am[U]);                                                                         \
        Pinc[U] = rinv[U];                                                                                                    \
    }
    ALPHA_STEP(0) ALPHA_STEP(1) ALPHA_STEP(2) ALPHA_STEP(3)
#undef ALPHA_STEP
    row_scan_mul_x4(Pinc[0], Pinc[1], Pinc[2], Pinc[3]);   // lane s: product of 1 / (1 - alpha) over entries 0..s of the batch
#define W_STEP(U)                                                                                                             \
    {                                                                                                                         \
        T[U] = bc<4 * B + U>(pT) * Pinc[U];   /* transmittance in front of entry s at pixel U */                              \
        w[U] = am[U] * T[U];                                                                                                  \
        cd[U] = sp.c0 * pc[U].x + sp.c1 * pc[U].y + sp.c2 * pc[U].z;                                                          \
        wc[U] = cd[U] * w[U];                                                                                                 \
        Sinc[U] = wc[U];                                                                                                      \
    }
    W_STEP(0) W_STEP(1) W_STEP(2) W_STEP(3)
#undef W_STEP
    row_scan_add_x4(Sinc[0], Sinc[1], Sinc[2], Sinc[3]);   // lane s: sum of w (c . dL/dpix) over entries 0..s of the batch
#define G_STEP(U)                                                                                                             \
    {                                                                                                                         \
        Rinc[U] = bc<4 * B + U>(pR) + Sinc[U];   /* lane 15: the pixel's R (+ tfbg) behind the next batch */                  \
        const float Rex = Rinc[U] - wc[U];       /* R (+ tfbg) behind entry s */                                              \
        const float dL_dalpha = T[U] * cd[U] - Rex * rinv[U];                                                                 \
        g[U] = Gm[U] * dL_dalpha;                                                                                             \
    }
    G_STEP(0) G_STEP(1) G_STEP(2) G_STEP(3)
#undef G_STEP
    lo4[0] = w[0] * pc[0].x + w[1] * pc[1].x + w[2] * pc[2].x + w[3] * pc[3].x;
    lo4[1] = w[0] * pc[0].y + w[1] * pc[1].y + w[2] * pc[2].y + w[3] * pc[3].y;
    lo4[2] = w[0] * pc[0].z + w[1] * pc[1].z + w[2] * pc[2].z + w[3] * pc[3].z;
    const float gx0 = g[0] + g[2], gx1 = g[1] + g[3], gy0 = g[0] + g[1], gy1 = g[2] + g[3];   // columns / rows of the region
    lo4[3] = gx0 + gx1;
    const float t0 = dx[0] * gx0, t1 = dx[1] * gx1, u0 = dy[0] * gy0, u1 = dy[1] * gy1;
    hi4[0] = t0 + t1;
    hi4[1] = u0 + u1;
    hi4[2] = dx[0] * t0 + dx[1] * t1;
    hi4[3] = dy[0] * (g[0] * dx[0] + g[1] * dx[1]) + dy[1] * (g[2] * dx[0] + g[3] * dx[1]);
    syy = dy[0] * u0 + dy[1] * u1;
    state_to_pixel_lane<4 * B + 0>(pT, pR, T[0], Rinc[0], lo4[3]);
    state_to_pixel_lane<4 * B + 1>(pT, pR, T[1], Rinc[1], lo4[3]);
    state_to_pixel_lane<4 * B + 2>(pT, pR, T[2], Rinc[2], lo4[3]);
    state_to_pixel_lane<4 * B + 3>(pT, pR, T[3], Rinc[3], lo4[3]);
}

// MB: staged list entries per round (one-byte list entries).  MUT: das3r_debug_mutate(1) — exp(power) (1 + 1e-4), the biased kernel the
// parity bars must catch.
// ABL (experiments builds, DAS3R_ABLATE with DAS3R_RENDER_BWD=fine128; results are wrong): 1 no walk, 2 no record adds, 4 no lists either, 8 nothing written out
// GEO: which sixteen of the tile's 64 regions are a wave's, and in which groups of four its passes take them (below)
template <int MB, int OCC, bool MUT = false, int ABL = 0, int GEO = 2>
__global__ void __launch_bounds__(256, OCC) render_backward_regions_kernel(
    const uint2 *__restrict__ ranges, const uint32_t *__restrict__ point_list, int W, int H, int tiles_x, int ntiles_strip /*render_common.h pack_tiles*/,
    const float4 *__restrict__ xyh, const float4 *__restrict__ conic_opacity, const float4 *__restrict__ rgbd,
    const float *__restrict__ bg, const float *__restrict__ final_T, const uint32_t *__restrict__ n_contrib,
    const float *__restrict__ dL_dpix, const uint32_t *__restrict__ slot_list, float *__restrict__ partial /*[I,9]*/,
    uint32_t last_g, uint32_t cap /*bounds of the list contents: render_common.h safe_range*/,
    const float4 *__restrict__ ckpt /*forward's checkpoints of long lists (render_common.h); used when gridDim.y > 1*/,
    unsigned long long *__restrict__ pairs /*common.h pair_counters(): null unless bench.py counts*/) {
    static_assert(MB >= 64 && MB <= 256 && MB % 8 == 0, "entries per round (one-byte list entries)");
    constexpr int LIST_STRIDE = MB + 4;                                  // bytes; the four rows of a wave read position e of their lists in one instruction
    constexpr int OFF_STAGE = 0;                                         // StagedSplat[MB]
    constexpr int OFF_ACC = OFF_STAGE + MB * (int)sizeof(StagedSplat);   // float4[2][4 waves][MB]: C0 C1 C2 M0 | Sgx Sgy Sxx Sxy (render_bwd_blk.hip)
    constexpr int OFF_ACC1 = OFF_ACC + 4 * MB * 8 * 4;                   // float[4 waves][MB]: Syy
    constexpr int OFF_SLOT = OFF_ACC1 + 4 * MB * 4;                      // uint32_t[MB]
    constexpr int OFF_LIST = OFF_SLOT + MB * 4;                          // uint8_t[4 waves][4 rows][LIST_STRIDE]: the lists of the strip being walked
    constexpr int OFF_MAX = OFF_LIST + 16 * LIST_STRIDE;                 // uint32_t[4]
    constexpr int OFF_CST = OFF_MAX + 16;                                // float4[4 waves][4 strips][4 rows][4 pixels]: d0, d1, d2, lastrel
    constexpr int LDS_BYTES = OFF_CST + 4 * 16 * 64;
    __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
    StagedSplat *const stage = reinterpret_cast<StagedSplat *>(lds + OFF_STAGE);
    float *const acc8 = reinterpret_cast<float *>(lds + OFF_ACC);
    float *const acc1 = reinterpret_cast<float *>(lds + OFF_ACC1);
    uint32_t *const s_slot = reinterpret_cast<uint32_t *>(lds + OFF_SLOT);
    uint32_t *const s_max = reinterpret_cast<uint32_t *>(lds + OFF_MAX);

    const int tile = xcd_tile(blockIdx.x, ntiles_strip, tiles_x);
    if (tile < 0) return;
    const int tid = threadIdx.x, lane = __lane_id();
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bx = tile % tiles_x, by = tile / tiles_x;
    // Which sixteen of the tile's 64 regions (8 x 8: column rx8, row ry8) are this wave's, and which four a pass B takes (row r of the wave:
    // region r of the group).  A tile's list is in depth order, and on a real sequence's depth maps that is a SPATIAL order: the 128 entries of
    // a round sit in one corner of the tile, and with a wave per quadrant one wave walks them while three wait at the round's barriers
    // (dsc: 0.707 ms; interleaved 0.608).  GEO 2 (default): a group = a 4x4 pixel block (2 x 2 regions), block (bx4, by4) belongs to wave
    // (bx4 + 2 by4) & 3 — horizontal, vertical and diagonal neighbours always to different waves, so every cluster of more than a few pixels is
    // shared by all four; a compact group's four lists are also closer in length than a strip's (tools/probes/rgn_model.py: 4 - 9 % fewer
    // batches than strips).  GEO 1: a group = a strip of four regions side by side (8 x 2 pixels), strip (row y8, half h) to wave (y8 + 2 h) & 3.
    // GEO 0: strips, a wave = one 8x8 quadrant (A-B runs).
    const int tx0 = bx * TILE_X, ty0 = by * TILE_Y;
    auto strip_y8 = [&](const int B) { return GEO == 1 ? 2 * B + (wave & 1) : 4 * (wave >> 1) + B; };
    auto strip_h = [&](const int B) { return GEO == 1 ? (((wave - strip_y8(B)) & 3) >> 1) : (wave & 1); };
    auto region_rx8 = [&](const int B, const int r) { return GEO == 2 ? 2 * ((wave - 2 * B) & 3) + (r & 1) : 4 * strip_h(B) + r; };
    auto region_ry8 = [&](const int B, const int r) { return GEO == 2 ? 2 * B + (r >> 1) : strip_y8(B); };
    const uint2 range = safe_range(ranges[tile], cap);
    if (gridDim.y > 1 && (int)blockIdx.y >= max(ckpt_buckets(range), 1)) return;   // (uniform) bucket-parallel launch: this tile's list has no bucket for this workgroup
    const int row = lane >> 4, s = lane & 15;
    uint8_t *const wave_lists = reinterpret_cast<uint8_t *>(lds + OFF_LIST) + wave * 4 * LIST_STRIDE;
    char *const cst_wave = lds + OFF_CST + wave * 16 * 64;

    // ---- this lane's pixel: pixel U = s & 3 of region `row` of strip s >> 2 ----
    const int ppx = tx0 + 2 * region_rx8(s >> 2, row) + (s & 1), ppy = ty0 + 2 * region_ry8(s >> 2, row) + ((s >> 1) & 1);
    const bool inside = ppx < W && ppy < H;
    float d0 = 0.f, d1 = 0.f, d2 = 0.f, pT, pR, my_lastrel = 0.f;
    uint32_t last_contributor;
    float my_T_final, my_tfbg;
    {
        const size_t pix = (size_t)ppy * W + ppx, plane = (size_t)H * W;
        my_T_final = inside ? final_T[pix] : 0.f;
        last_contributor = inside ? n_contrib[pix] : 0u;
        if (inside) {
            d0 = dL_dpix[pix];
            d1 = dL_dpix[plane + pix];
            d2 = dL_dpix[2 * plane + pix];
        }
        my_tfbg = my_T_final * (bg[0] * d0 + bg[1] * d1 + bg[2] * d2);
        pT = my_T_final;
        pR = my_tfbg;   // the replay state R always travels with T_final (bg . dL/dpix): only their sum is used
    }
    float *const cst_mine = reinterpret_cast<float *>(cst_wave + (((s >> 2) * 4 + row) * 4 + (s & 3)) * 16);
    *reinterpret_cast<float4 *>(cst_mine) = make_float4(d0, d1, d2, 0.f);
    // the accumulator regions are zero between rounds: whoever reads a record when the round is written out clears it
    for (int f = tid; f < 4 * MB * 8 / 4; f += TILE_PIX) reinterpret_cast<v4f *>(acc8)[f] = v4f{0.f, 0.f, 0.f, 0.f};
    for (int f = tid; f < 4 * MB; f += TILE_PIX) acc1[f] = 0.f;
    // no pixel of this tile blended anything past list position max_contrib: start the replay there
    uint32_t mx = last_contributor;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, o, 64));
    if (lane == 0) s_max[wave] = mx;
    __syncthreads();
    const uint32_t list_len = range.y - range.x;
    const uint32_t tile_contrib = min(max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3])), list_len);
    const int slices = (int)gridDim.y;
    const int nbuckets = slices > 1 ? max(ckpt_buckets(range), 1) : 1;
    int batches_done = 0;   // (wave-uniform) batches of 4 pixel steps = 256 pairs each

    for (int bk = (int)blockIdx.y; bk < nbuckets; bk += slices) {
        const uint32_t lo = slices > 1 ? (uint32_t)bk * BUCKET : 0u;                     // the bucket's list positions [lo, hi)
        const uint32_t hi = slices > 1 ? min(list_len, lo + BUCKET) : list_len;
        const uint32_t max_contrib = tile_contrib > lo ? min(tile_contrib, hi) - lo : 0u;   // entries of the bucket to replay (its first ones)
        const int rounds = ((int)max_contrib + MB - 1) / MB;
        {   // list entries beyond the last contributor receive no gradient from this tile: their partial rows are zero
            const uint32_t ntail = (hi - lo - max_contrib) * NACC;
            for (uint32_t f = tid; f < ntail; f += TILE_PIX) {
                const uint32_t t = f / NACC, q = f - t * NACC;
                partial[(size_t)min(slot_list[range.x + lo + max_contrib + t], cap - 1u) * NACC + q] = 0.f;
            }
        }
        if (max_contrib == 0u) continue;   // (uniform)
        if (slices > 1) {   // the pixels' state at the far end of the bucket
            float T0 = my_T_final, R0 = 0.f;
            if (bk < nbuckets - 1) {
                const int cpix = (ppy - by * TILE_Y) * 16 + (ppx - bx * TILE_X);
                const float4 far = ckpt_slot(const_cast<float4 *>(ckpt), range, tile, bk)[cpix];            // (T, C) in front of position hi
                const float4 fin = ckpt_slot(const_cast<float4 *>(ckpt), range, tile, nbuckets - 1)[cpix];  // final (T, C)
                T0 = far.x;
                R0 = d0 * (fin.y - far.y) + d1 * (fin.z - far.z) + d2 * (fin.w - far.w);   // (c . dL/dpix) alpha T of everything behind
            }
            pT = T0;
            pR = R0 + my_tfbg;
        }
        uint32_t g_ahead = 0u, slot_ahead = 0u;   // my entry of the next round
        if (tid < min(MB, (int)max_contrib)) {
            const uint32_t pos = range.x + lo + max_contrib - 1 - tid;
            g_ahead = point_list[pos];
            slot_ahead = slot_list[pos];
        }
        for (int i = 0; i < rounds; i++) {
            const int done_before = i * MB;
            const int n = min(MB, (int)max_contrib - done_before);
            // stage the round in reverse list order; entry j holds list position (lo + max_contrib - 1 - done_before - j)
            // (MB <= 256: a thread stages at most one entry; its list words were requested a round ahead, so that staging is ONE trip to memory)
            if (tid < n) {
                const uint32_t g = min(g_ahead, last_g);
                s_slot[tid] = min(slot_ahead, cap - 1u);
                stage[tid].xyh = xyh[(size_t)g * SPLAT_REC];           // one 64-byte record: a single cache line per splat
                stage[tid].co = conic_opacity[(size_t)g * SPLAT_REC];
                stage[tid].rgbd = rgbd[(size_t)g * SPLAT_REC];
            }
            if (i + 1 < rounds && tid < min(MB, (int)max_contrib - done_before - MB)) {
                const uint32_t pos = range.x + lo + max_contrib - 1 - (done_before + MB) - tid;
                g_ahead = point_list[pos];
                slot_ahead = slot_list[pos];
            }
            // the pixel's last contributor relative to the round's window (render_bwd_blk.hip): entry j takes part iff its posrel < lastrel
            {
                const long long base = (long long)lo + (long long)max_contrib - done_before - MB;
                const long long rel = (long long)last_contributor - base;
                my_lastrel = (float)(rel < 0 ? 0ll : (rel > MB ? (long long)MB : rel));
                cst_mine[3] = my_lastrel;
            }
            // the last contributor of each of the wave's sixteen REGIONS, relative to the round: an entry behind it gets nothing from that region
            // (every pixel's clamp makes its pairs inert) and is not listed for it
            float rgn_last[16];
            {
                float v = my_lastrel;
                v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1 /*quad_perm [1, 0, 3, 2]*/, 0xf, 0xf, true)));
                v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E /*quad_perm [2, 3, 0, 1]*/, 0xf, 0xf, true)));
#pragma unroll
                for (int k = 0; k < 16; k++) rgn_last[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16 * (k & 3) + 4 * (k >> 2)));   // k = 4 strip + row
            }
            __syncthreads();

            // ---- what each staged entry can reach of the wave's sixteen regions: one mask per entry (kept in registers, a chunk of 64 per lane) ----
            constexpr int CHUNKS = (MB + 63) / 64;
            uint32_t m16[CHUNKS];
#pragma unroll
            for (int c = 0; c < CHUNKS; c++) {
                const int j = c * 64 + lane;
                const float4 p = stage[j < n ? j : 0].xyh;
                // the forward's test (render_common.h region_mask): |centre distance| <= cull half extent + half the region, per axis
                uint32_t xb = 0, yb = 0;
                const float hx = p.z + 0.5f, hy = p.w + 0.5f;
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    xb |= (fabsf(p.x - ((float)(tx0 + 2 * k) + 0.5f)) <= hx ? 1u : 0u) << k;
                    yb |= (fabsf(p.y - ((float)(ty0 + 2 * k) + 0.5f)) <= hy ? 1u : 0u) << k;
                }
                uint32_t m = 0;
#pragma unroll
                for (int B = 0; B < 4; B++) {
                    if constexpr (GEO == 2) {
                        const uint32_t xq = (xb >> region_rx8(B, 0)) & 3u, yq = (yb >> (2 * B)) & 3u;
                        m |= (((yq & 1u) ? xq : 0u) | ((yq & 2u) ? xq << 2 : 0u)) << (4 * B);
                    } else {
                        m |= ((yb >> strip_y8(B)) & 1u) ? ((xb >> (4 * strip_h(B))) & 15u) << (4 * B) : 0u;
                    }
                }
                m16[c] = j < n ? m : 0u;   // bit 4 strip + row
            }

            // ---- the four strips, one after the other; inside a strip the rows add their (region, entry) sums to the entry's record ONE AFTER
            //      THE OTHER (two rows may hold the same entry at the same time), as in render_bwd_blk.hip ----
            auto strip = [&](auto Bc) {
                constexpr int B = decltype(Bc)::value;
                // the strip's four region lists (entries in staged order = reverse list order, kept)
                int len[4] = {0, 0, 0, 0};
#pragma unroll
                for (int c = 0; c < ((ABL & 4) ? 0 : CHUNKS); c++) {
                    if (__ballot(((m16[c] >> (4 * B)) & 15u) != 0u) == 0ull) continue;   // (uniform) nothing of this chunk reaches the strip: the common case on a spatially sorted list
                    const int j = c * 64 + lane;
                    const float posrel_j = (float)(MB - 1 - j);
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const bool hit = ((m16[c] >> (4 * B + r)) & 1u) != 0u && posrel_j < rgn_last[4 * B + r];
                        const uint64_t m = __ballot(hit);
                        const int at = len[r] + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                        if (hit) wave_lists[r * LIST_STRIDE + at] = (uint8_t)j;
                        len[r] += __popcll(m);
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");   // the wave reads its own lists back
                const int l0 = len[0], l1 = len[1], l2 = len[2], l3 = len[3];
                const int my_len = row == 0 ? l0 : (row == 1 ? l1 : (row == 2 ? l2 : l3));
                const int longest = (ABL & 1) ? 0 : __builtin_amdgcn_readfirstlane(max(max(l0, l1), max(l2, l3)));
                const uint8_t *const mine = wave_lists + row * LIST_STRIDE;
                const char *const cst = cst_wave + (4 * B + row) * 64;
                const float rxf = (float)(tx0 + 2 * region_rx8(B, row)), ryf = (float)(ty0 + 2 * region_ry8(B, row));   // the region's corner pixel
                batches_done += (longest + 15) >> 4;
                for (int b = 0; b < longest; b += 16) {
                    const int e = b + s;
                    const bool valid = e < my_len;
                    const int j = valid ? (int)mine[e] : 0;
                    SplatRegs sp;
                    {
                        const float4 p = stage[j].xyh;
                        const float4 co = stage[j].co;
                        const float4 c = lds_read4(&stage[j].rgbd);   // (b128, not b96: half the LDS cycles)
                        sp.x = p.x; sp.y = p.y;
                        sp.A = co.x; sp.B = co.y; sp.C = co.z;
                        sp.o = valid ? co.w : 0.f;   // a lane without an entry: alpha = 0 on every pixel
                        sp.c0 = c.x; sp.c1 = c.y; sp.c2 = c.z;
                        sp.posrel = (float)(MB - 1 - j);
                    }
                    v4f p_lo4, p_hi4;
                    float p_syy;
                    region_steps<B, MUT>(sp, rxf, ryf, pT, pR, cst, p_lo4, p_hi4, p_syy);
                    float *const p_rec8 = acc8 + ((size_t)wave * MB + j) * 4;
                    float *const p_rec1 = acc1 + wave * MB + j;
                    auto add_pass = [&](const int r) {
                        if (row == r && valid && !(ABL & 2)) {
                            const v4f o0 = *reinterpret_cast<const v4f *>(p_rec8), o1 = *reinterpret_cast<const v4f *>(p_rec8 + 4 * MB * 4);
                            const float o2 = *p_rec1;
                            *reinterpret_cast<v4f *>(p_rec8) = o0 + p_lo4;
                            *reinterpret_cast<v4f *>(p_rec8 + 4 * MB * 4) = o1 + p_hi4;
                            *p_rec1 = o2 + p_syy;
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // (compiler only: the next row's read stays behind this write)
                    };
                    // a row whose list is exhausted has nothing to add (uniform branches)
                    if (l0 > b) add_pass(0);
                    if (l1 > b) add_pass(1);
                    if (l2 > b) add_pass(2);
                    if (l3 > b) add_pass(3);
                }
            };
            strip(std::integral_constant<int, 0>{});
            strip(std::integral_constant<int, 1>{});
            strip(std::integral_constant<int, 2>{});
            strip(std::integral_constant<int, 3>{});
            __syncthreads();
            // ---- write the round out: the four waves' records of every staged entry -> the nine per-instance sums (render_bwd_blk.hip) ----
            for (int t0 = 0; t0 < n; t0 += TILE_PIX) {
                const int t = t0 + tid;
                if (t < n) {
                    float a[9];
#pragma unroll
                    for (int q = 0; q < 9; q++) a[q] = 0.f;
#pragma unroll
                    for (int w = 0; w < 4; w++) {
                        float *const r8 = acc8 + ((size_t)w * MB + t) * 4;
                        const v4f lo4 = *reinterpret_cast<const v4f *>(r8), hi4 = *reinterpret_cast<const v4f *>(r8 + 4 * MB * 4);
                        a[0] += lo4[0]; a[1] += lo4[1]; a[2] += lo4[2]; a[3] += lo4[3];
                        a[4] += hi4[0]; a[5] += hi4[1]; a[6] += hi4[2]; a[7] += hi4[3];
                        a[8] += acc1[w * MB + t];
                        *reinterpret_cast<v4f *>(r8) = v4f{0.f, 0.f, 0.f, 0.f};
                        *reinterpret_cast<v4f *>(r8 + 4 * MB * 4) = v4f{0.f, 0.f, 0.f, 0.f};
                        acc1[w * MB + t] = 0.f;
                    }
                    const float4 co = stage[t].co;
                    const float kh = -0.5f * co.w;
                    const float Sgx = kh * a[4], Sgy = kh * a[5];   // -1/2 o sum g dx, dy
                    float *rowp = partial + (size_t)s_slot[t] * NACC;
                    if constexpr ((ABL & 8) != 0) {
                        if (a[0] == 123.456f) rowp[0] = a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7] + a[8];
                        continue;
                    }
                    store_partial_row(rowp, a[0], a[1], a[2], (Sgx * co.x + Sgy * co.y) * (float)W /*dL/dmean2D in NDC units: 2 * (W / 2)*/,
                                      (Sgy * co.z + Sgx * co.y) * (float)H, kh * a[6], kh * a[7], kh * a[8], a[3]);
                }
            }
            // (no barrier here, round 6: MB <= 256 — entry t of a round is staged AND written out by thread t alone, so the next round's staging
            //  overwrites stage[t] / s_slot[t] behind this thread's own reads; the records it cleared are next touched behind the barrier that
            //  follows that staging; lists and masks are wave-private; nobody is still walking: every wave has passed the barrier above)
        }
    }
    if (pairs != nullptr && lane == 0 && batches_done > 0) {
        atomicAdd(pairs + 1, (unsigned long long)batches_done * 256ull);
        atomicAdd(pairs + 3, (unsigned long long)batches_done * 4ull);
    }
}

int launch_render_backward_regions(const das3r_raster_args *a, const float *dL_dpix, char *geom, char *binning, char *img, const Layout &L,
                                   float *partial, int mb, int slices, hipStream_t s) {
#define ARGS                                                                                                              \
    (const uint2 *)(img + L.pub.ranges), (const uint32_t *)(binning + L.pub.point_list), a->image_width, a->image_height, \
        L.tiles_x, pack_tiles(L), (const float4 *)(geom + L.pub.xy), (const float4 *)(geom + L.pub.conic_opacity),             \
        (const float4 *)(geom + L.pub.rgbd), a->bg, (const float *)(img + L.pub.final_T),                                 \
        (const uint32_t *)(img + L.pub.n_contrib), dL_dpix, (const uint32_t *)(binning + L.b_slot), partial,              \
        (uint32_t)(a->P - 1), (uint32_t)L.capacity, (const float4 *)(binning + L.b_ckpt), pair_counters()
#define GO(MBV, OCC) DAS3R_LAUNCH((render_backward_regions_kernel<MBV, OCC>), dim3(xcd_grid(L), std::max(slices, 1)), dim3(TILE_PIX), 0, s, ARGS)
#define GQ(MBV, OCC, G) DAS3R_LAUNCH((render_backward_regions_kernel<MBV, OCC, false, 0, G>), dim3(xcd_grid(L), std::max(slices, 1)), dim3(TILE_PIX), 0, s, ARGS)
    // DAS3R_RENDER_BWD=fine<entries per round>: LDS per workgroup = 212 MB + 4.2 KB -> 96: 24.5 KB, 128: 31.3 KB (5 per CU), 160: 38.1 (4), 192: 44.9 (3),
    // 256: 58.4 (2)
#ifdef DAS3R_EXPERIMENTS
    if (mb == 128 && switches().ablate_set) {
        const int abl = switches().ablate;
#define GA(A) DAS3R_LAUNCH((render_backward_regions_kernel<128, 5, false, A>), dim3(xcd_grid(L), std::max(slices, 1)), dim3(TILE_PIX), 0, s, ARGS)
        if (abl == 1) GA(1); else if (abl == 2) GA(2); else if (abl == 5) GA(5); else if (abl == 8) GA(8); else if (abl == 13) GA(13); else GA(0);
#undef GA
    } else
#endif
    if (switches().mutate == 1) DAS3R_LAUNCH((render_backward_regions_kernel<128, 5, true>), dim3(xcd_grid(L), std::max(slices, 1)), dim3(TILE_PIX), 0, s, ARGS);
    else if (switches().render_bwd_pix == 9) GQ(128, 5, 0);   // fine128q: strips, a wave per quadrant (A-B runs)
    else if (switches().render_bwd_pix == 8) GQ(128, 5, 1);   // fine128s: strips, interleaved
    else if (mb == 64) GO(64, 5);
    else if (mb == 96) GO(96, 5);
    else if (mb == 160) GO(160, 4);
    else if (mb == 192) GO(192, 3);
    else if (mb == 256) GO(256, 2);
    else GO(128, 5);
#undef GO
#undef GQ
#undef ARGS
    KERNEL_CHECK(s, a->debug, "render_backward_regions");
    return DAS3R_OK;
}

}  // namespace das3r
