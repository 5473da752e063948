// render_bwd_blk.hip — K7 with every 16-lane DPP row of a wave on its own 4x4 pixel BLOCK: lanes = 16 splats of the block's own
// culled list, time = the block's 16 pixels; nothing in the pair loop touches LDS or the matrix cores.
// Replaces upstream:cuda_rasterizer/backward.cu renderCUDA like render_bwd.hip / render_bwd_scan.hip; same inputs,
// same partial rows.
//
// Why a fourth decomposition.  render_bwd_scan.hip walks one culled list per 8x8 QUADRANT: at 1 M splats /
// 1080p only 36 % of the (pixel, splat) pairs it evaluates are live (tools/pair_stats_cpu.py: 137 evaluated pairs per instance
// for 49 live ones), its two per-pixel recurrences read and write the pixel state in LDS every step (two ds_read_b128 + one
// ds_write2 per 64 pairs), and its nine sums go through split-bf16 operands on the matrix cores (16 mantissa bits per factor,
// truncated: a measured bias of -7.5e-6 relative in every gradient, profiles/r03_grad_error_c4.json).  Here
//   * the unit of culling is the 4x4 block (a DPP row of the wave = one block of the wave's quadrant, as in the forward kernel
//     render_rows.hip): exact ellipse-vs-quadrant test first, then an octagon test (axis-aligned extents + the two diagonal ones)
//     per block: 4.9 block hits per instance x 16 pixels = 79 pairs, 101 with the four rows of a wave in lockstep and batches of
//     16 (x 0.75 of the quadrant walk);
//   * lane (r, s) of the walk holds splat s of row r's batch in registers for the block's 16 pixel steps; AS A PIXEL LANE the same
//     lane owns pixel s of block r: its constants (centre, dL/dpixel, T_final (bg . dL/dpix), n_contrib) and its replay state
//     (T, R) live in nine registers of that lane for the whole kernel, and step k reads them with the DPP control row_newbcast:k
//     folded into the consuming VOP2 instruction (v_subrev_f32_dpp, v_mul_f32_dpp, v_fmac_f32_dpp ...): no instruction, no LDS
//     round trip.  The recurrences run along the sixteen lanes of a row as DPP row scans (render_scan.h), four pixel steps
//     interleaved; the row's totals (lane 15) go back to pixel lane k with v_cndmask_b32_dpp under a constant lane mask;
//   * the nine sums of a (block, splat) are reductions over TIME, i.e. nine fp32 accumulators per lane — exact fp32 products,
//     no operand splitting, no MFMA: colour sums as w dL/dpix, geometry as raw moments of g = G dL/dalpha about the block's own
//     corner (weights 0..3: inline constants), turned into the sums about the splat centre once per batch;
//   * a (wave, entry) record is 36 bytes in a wave-private LDS region as before; the four rows of a wave add their batch into it
//     one after the other (two rows may hold the same entry at the same time), the four regions are added per entry when the
//     round is written out.
// Per 64 pairs: ~48 VALU + 2 transcendentals, 0 LDS, 0 MFMA; ~70 VGPRs, 27 KB of LDS per workgroup (5 workgroups per CU).
// Bucket-parallel replay of long lists as in render_bwd_scan.hip (gridDim.y > 1).  CPU model: tests/test_blk_model.py.
#include "render_scan.h"

namespace das3r {

// lane K of every 16-lane row, to all lanes of the row (DPP row_newbcast: gfx90a and later; folded into VOP2 consumers)
template <int K>
__device__ __forceinline__ float bc(const float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x150 + K, 0xf, 0xf, true));
}
// acc += x[lane K of the row] * v as ONE v_fmac_f32_dpp.  (The compiler folds a broadcast into its consumer only when that is the
// broadcast's single use; dL/dpixel is used twice per step — in c . dL/dpix and here — and came out as v_mov_b32_dpp + fmac.)
// x is a per-pixel constant: never written inside the walk, so the DPP read needs no wait states.
template <int K>
__device__ __forceinline__ void fmac_bc(float &acc, const float x, const float v) {
    asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(v), "i"(K));
}
// Pixel lane K of every row takes the row's totals (lane 15) of `t15` and `r15`; the other lanes keep theirs.  One scalar move for
// the lane mask + two v_cndmask_b32_dpp.  `order`: a value computed AFTER t15 / r15 in program order (>= 2 VALU instructions later):
// a DPP read needs two wait states behind the VALU write of its source, and the assembler does not add them inside inline asm.
template <int K>
__device__ __forceinline__ void state_to_pixel_lane(float &stT, float &stR, const float t15, const float r15, const float order) {
    constexpr unsigned long long keep = ~(0x0001000100010001ull << K);   // vcc = 1: keep the old value
    asm("s_mov_b64 vcc, %5\n\t"
        "v_cndmask_b32_dpp %0, %2, %0, vcc row_newbcast:15 row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %1, %3, %1, vcc row_newbcast:15 row_mask:0xf bank_mask:0xf"
        : "+v"(stT), "+v"(stR)
        : "v"(t15), "v"(r15), "v"(order), "s"(keep)
        : "vcc");
}

// per-pixel registers of a pixel lane (lane s of row r owns pixel s of block r: x = s & 3, y = s >> 2)
struct PixelRegs {
    float pxf, pyf;            // pixel centre
    float d0, d1, d2, tfbg;    // dL/dpixel, T_final * (bg . dL/dpixel)
    float T, R;                // replay state
    float lastrel;             // n_contrib relative to the round's staged window, as a float in [0, MB]
};
struct SplatRegs {
    float x, y;                // centre
    float A, B, C, o;          // conic, opacity (0 on lanes without an entry)
    float c0, c1, c2;          // colour
    float posrel;              // list position relative to the round's window (MB - 1 - j)
};
struct Sums {
    float C0, C1, C2, M0, Mu, Mv, Muu, Muv, Mvv;
};

// One image row KY of the block: its four pixel steps K = 4 KY .. 4 KY + 3, interleaved.  pair_alpha's arithmetic, bit for bit
// (render_common.h), so that every pair takes the decision the forward kernel took.
template <int KY>
__device__ __forceinline__ void block_row(const SplatRegs &sp, PixelRegs &px, Sums &acc) {
    const float dy = sp.y - bc<4 * KY>(px.pyf);
    const float cyy = __fmul_rn(__fmul_rn(sp.C, dy), dy);
    float am[4], Gm[4], rinv[4], Pinc[4], T[4], cd[4], w[4], wc[4], Sinc[4], Rinc[4], g[4];
#define ALPHA_STEP(U)                                                                                                         \
    {                                                                                                                         \
        constexpr int K = 4 * KY + U;                                                                                         \
        const float dx = sp.x - bc<K>(px.pxf);                                                                                \
        const float q = __fmaf_rn(__fmul_rn(sp.A, dx), dx, cyy);                                                              \
        const float power = __fmaf_rn(-0.5f, q, -__fmul_rn(__fmul_rn(sp.B, dx), dy));                                         \
        const float G = __expf(power);                                                                                        \
        /* position < n_contrib  <=>  lastrel - posrel >= 1, else <= 0 (small integers): a third operand of the alpha clamp —  */ \
        /* where the pair takes part the minimum is min(0.99, o G) as in pair_alpha, elsewhere it fails the 1/255 test         */ \
        const float a1 = fminf(fminf(0.99f, __fmul_rn(sp.o, G)), bc<K>(px.lastrel) - sp.posrel);                              \
        const bool active = (!(power > 0.0f)) & (a1 >= (1.0f / 255.0f));                                                      \
        am[U] = active ? a1 : 0.f;                                                                                            \
        Gm[U] = active ? G : 0.f;                                                                                             \
        rinv[U] = __builtin_amdgcn_rcpf(1.f - am[U]);                                                                         \
        Pinc[U] = rinv[U];                                                                                                    \
    }
    ALPHA_STEP(0) ALPHA_STEP(1) ALPHA_STEP(2) ALPHA_STEP(3)
#undef ALPHA_STEP
    row_scan_mul_x4(Pinc[0], Pinc[1], Pinc[2], Pinc[3]);   // lane s: product of 1 / (1 - alpha) over splats 0..s of the batch
#define W_STEP(U)                                                                                                             \
    {                                                                                                                         \
        constexpr int K = 4 * KY + U;                                                                                         \
        T[U] = bc<K>(px.T) * Pinc[U];               /* transmittance in front of splat s at pixel K */                        \
        w[U] = am[U] * T[U];                                                                                                  \
        cd[U] = sp.c0 * bc<K>(px.d0);                                                                                         \
        fmac_bc<K>(cd[U], px.d1, sp.c1);                                                                                      \
        fmac_bc<K>(cd[U], px.d2, sp.c2);                                                                                      \
        wc[U] = cd[U] * w[U];                                                                                                 \
        Sinc[U] = wc[U];                                                                                                      \
    }
    W_STEP(0) W_STEP(1) W_STEP(2) W_STEP(3)
#undef W_STEP
    row_scan_add_x4(Sinc[0], Sinc[1], Sinc[2], Sinc[3]);   // lane s: sum of w (c . dL/dpix) over splats 0..s of the batch
#define G_STEP(U)                                                                                                             \
    {                                                                                                                         \
        constexpr int K = 4 * KY + U;                                                                                         \
        Rinc[U] = bc<K>(px.R) + Sinc[U];            /* lane 15: the pixel's R behind the next batch */                        \
        const float Rex = Rinc[U] - wc[U];          /* R behind splat s */                                                    \
        const float dL_dalpha = T[U] * cd[U] - (Rex + bc<K>(px.tfbg)) * rinv[U];                                              \
        g[U] = Gm[U] * dL_dalpha;                                                                                             \
        fmac_bc<K>(acc.C0, px.d0, w[U]);                                                                                      \
        fmac_bc<K>(acc.C1, px.d1, w[U]);                                                                                      \
        fmac_bc<K>(acc.C2, px.d2, w[U]);                                                                                      \
        acc.M0 += g[U];                                                                                                       \
        if (U > 0) acc.Mu += (float)U * g[U];                                                                                 \
        if (KY > 0) acc.Mv += (float)KY * g[U];                                                                               \
        if (U > 0) acc.Muu += (float)(U * U) * g[U];                                                                          \
        if (U > 0 && KY > 0) acc.Muv += (float)(U * KY) * g[U];                                                               \
        if (KY > 0) acc.Mvv += (float)(KY * KY) * g[U];                                                                       \
    }
    G_STEP(0) G_STEP(1) G_STEP(2) G_STEP(3)
#undef G_STEP
    state_to_pixel_lane<4 * KY + 0>(px.T, px.R, T[0], Rinc[0], acc.M0);
    state_to_pixel_lane<4 * KY + 1>(px.T, px.R, T[1], Rinc[1], acc.M0);
    state_to_pixel_lane<4 * KY + 2>(px.T, px.R, T[2], Rinc[2], acc.M0);
    state_to_pixel_lane<4 * KY + 3>(px.T, px.R, T[3], Rinc[3], acc.M0);
}

// Octagon test of a splat against the 4x4 block whose pixel centres span [cx - 1.5, cx + 1.5] x [cy - 1.5, cy + 1.5]: the
// axis-aligned extents (hx, hy: preprocess.hip, with their margins) and the extents hd1, hd2 along (1, 1) / sqrt 2 and
// (1, -1) / sqrt 2.  Conservative: outside any of the four slabs alpha < 1 / 255 on every pixel centre of the block.
__device__ __forceinline__ bool block_hit_oct(const float x, const float y, const float hx, const float hy, const float hd1, const float hd2,
                                              const float cx, const float cy) {
    const float ddx = x - cx, ddy = y - cy;
    constexpr float RS2 = 0.70710678f, HALF_DIAG = 2.1213204f + 1e-3f;   // (1.5 + 1.5) / sqrt 2
    return (fabsf(ddx) <= hx + 1.5f) & (fabsf(ddy) <= hy + 1.5f) & (fabsf(ddx + ddy) * RS2 <= hd1 + HALF_DIAG) &
           (fabsf(ddx - ddy) * RS2 <= hd2 + HALF_DIAG);
}

template <int MB>
__global__ void __launch_bounds__(256, 5) render_backward_blk_kernel(
    const uint2 *__restrict__ ranges, const uint32_t *__restrict__ point_list, int W, int H, int tiles_x, int ntiles_strip /*tiles | strip height << 24: render_common.h xcd_tile*/,
    const float4 *__restrict__ xyh, const float4 *__restrict__ conic_opacity, const float4 *__restrict__ rgbd,
    const float *__restrict__ bg, const float *__restrict__ final_T, const uint32_t *__restrict__ n_contrib,
    const float *__restrict__ dL_dpix, const uint32_t *__restrict__ slot_list, float *__restrict__ partial /*[I,9]*/,
    uint32_t last_g, uint32_t cap /*bounds of the list contents: render_common.h safe_range*/,
    const float4 *__restrict__ ckpt /*forward's checkpoints of long lists (render_common.h); used when gridDim.y > 1*/) {
    static_assert(MB == 64 || MB == 128 || MB == 256, "entries per round");
    constexpr int LIST_STRIDE = MB + 4;                                  // bytes; the four rows of a wave read position e of their lists in one instruction
    constexpr int OFF_STAGE = 0;                                         // StagedSplat[MB]
    constexpr int OFF_ACC = OFF_STAGE + MB * (int)sizeof(StagedSplat);   // float[4 waves][MB][8]: C0 C1 C2 M0 | Sgx Sgy Sxx Sxy
    constexpr int OFF_ACC1 = OFF_ACC + 4 * MB * 8 * 4;                   // float[4 waves][MB]: Syy
    constexpr int OFF_SLOT = OFF_ACC1 + 4 * MB * 4;                      // uint32_t[MB]
    constexpr int OFF_LIST = OFF_SLOT + MB * 4;                          // uint8_t[4 waves][4 rows][LIST_STRIDE]
    constexpr int OFF_MAX = OFF_LIST + 16 * LIST_STRIDE;                 // uint32_t[4]
    constexpr int LDS_BYTES = OFF_MAX + 16;
    __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
    StagedSplat *const stage = reinterpret_cast<StagedSplat *>(lds + OFF_STAGE);
    float *const acc8 = reinterpret_cast<float *>(lds + OFF_ACC);
    float *const acc1 = reinterpret_cast<float *>(lds + OFF_ACC1);
    uint32_t *const s_slot = reinterpret_cast<uint32_t *>(lds + OFF_SLOT);
    uint32_t *const s_max = reinterpret_cast<uint32_t *>(lds + OFF_MAX);

    const int ntiles = ntiles_strip & 0xFFFFFF;
    const int tile = xcd_tile(blockIdx.x, ntiles, tiles_x, ntiles_strip >> 24);
    if (tile < 0) return;
    const int tid = threadIdx.x, lane = __lane_id();
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bx = tile % tiles_x, by = tile / tiles_x;
    const int qx0 = bx * TILE_X + ((wave & 1) << 3), qy0 = by * TILE_Y + ((wave >> 1) << 3);   // the wave's quadrant
    const float qcx = (float)qx0 + 3.5f, qcy = (float)qy0 + 3.5f;
    const uint2 range = safe_range(ranges[tile], cap);
    const int row = lane >> 4, s = lane & 15;
    uint8_t *const mine = reinterpret_cast<uint8_t *>(lds + OFF_LIST) + (wave * 4 + row) * LIST_STRIDE;
    uint8_t *const wave_lists = reinterpret_cast<uint8_t *>(lds + OFF_LIST) + wave * 4 * LIST_STRIDE;

    // ---- this lane's pixel: pixel s of block `row` of the quadrant ----
    const int ppx = qx0 + ((row & 1) << 2) + (s & 3), ppy = qy0 + ((row >> 1) << 2) + (s >> 2);
    const bool inside = ppx < W && ppy < H;
    PixelRegs px;
    uint32_t last_contributor;
    float my_T_final;
    {
        const size_t pix = (size_t)ppy * W + ppx, plane = (size_t)H * W;
        my_T_final = inside ? final_T[pix] : 0.f;
        last_contributor = inside ? n_contrib[pix] : 0u;
        px.d0 = px.d1 = px.d2 = 0.f;
        if (inside) {
            px.d0 = dL_dpix[pix];
            px.d1 = dL_dpix[plane + pix];
            px.d2 = dL_dpix[2 * plane + pix];
        }
        px.pxf = (float)ppx;
        px.pyf = (float)ppy;
        px.tfbg = my_T_final * (bg[0] * px.d0 + bg[1] * px.d1 + bg[2] * px.d2);
        px.T = my_T_final;
        px.R = 0.f;
        px.lastrel = 0.f;
    }
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
    const float bx0f = (float)(qx0 + ((row & 1) << 2)), by0f = (float)(qy0 + ((row >> 1) << 2));   // corner pixel of the lane's block: origin of its moments

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
                R0 = px.d0 * (fin.y - far.y) + px.d1 * (fin.z - far.z) + px.d2 * (fin.w - far.w);   // (c . dL/dpix) alpha T of everything behind
            }
            px.T = T0;
            px.R = R0;
        }
        for (int i = 0; i < rounds; i++) {
            const int done_before = i * MB;
            const int n = min(MB, (int)max_contrib - done_before);
            // stage the round in reverse list order; entry j holds list position (lo + max_contrib - 1 - done_before - j)
            for (int t = tid; t < n; t += TILE_PIX) {
                const uint32_t pos = range.x + lo + max_contrib - 1 - done_before - t;
                const uint32_t g = min(point_list[pos], last_g);
                s_slot[t] = min(slot_list[pos], cap - 1u);
                stage[t].xyh = xyh[(size_t)g * SPLAT_REC];           // one 64-byte record: a single cache line per splat
                stage[t].co = conic_opacity[(size_t)g * SPLAT_REC];
                stage[t].rgbd = rgbd[(size_t)g * SPLAT_REC];
            }
            // the pixel's last contributor relative to the round's window: list position of entry j = base + (MB - 1 - j) with
            // base = lo + max_contrib - done_before - MB (may be negative); entry j takes part iff its position < n_contrib
            {
                const long long base = (long long)lo + (long long)max_contrib - done_before - MB;
                const long long rel = (long long)last_contributor - base;
                px.lastrel = (float)(rel < 0 ? 0ll : (rel > MB ? (long long)MB : rel));
            }
            __syncthreads();

            // ---- the wave's four row lists (entries in staged order = reverse list order, kept) ----
            int len[4] = {0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < MB / 64; k++) {
                const int j = k * 64 + lane;
                const int jc = j < n ? j : 0;
                const float4 p = stage[jc].xyh;
                const float4 co = stage[jc].co;
                // alpha >= 1/255  <=>  q <= 2 ln(255 o): taken from the opacity (as preprocess.hip takes it), not from the stored extents
                const float tau = splat_tau(co.w);
                const bool qhit = (j < n) & quadrant_hit(p, qcx, qcy) && rect_hit_tight_tau(p, co, tau, (float)qx0, (float)qy0);
                // extents along the two diagonals from the axis-aligned ones: tau Sxx = ex2, tau Syy = ey2, tau Sxy = -B ex2 / C
                const float ex = (p.z - 0.02f) * (1.0f / 1.0005f), ey = (p.w - 0.02f) * (1.0f / 1.0005f);
                const float ex2 = ex * ex, ey2 = ey * ey, txy = -co.y * ex2 * __builtin_amdgcn_rcpf(co.z);
                const float half = 0.5f * (ex2 + ey2), slack = 2e-6f * (ex2 + ey2) + 1e-3f;   // (cancellation of long thin splats)
                const float hd1 = sqrtf(fmaxf(half + txy, 0.f) + slack) * 1.0005f + 0.05f;
                const float hd2 = sqrtf(fmaxf(half - txy, 0.f) + slack) * 1.0005f + 0.05f;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const bool hit = qhit && block_hit_oct(p.x, p.y, p.z, p.w, hd1, hd2, (float)(qx0 + ((r & 1) << 2)) + 1.5f,
                                                           (float)(qy0 + ((r >> 1) << 2)) + 1.5f);
                    const uint64_t m = __ballot(hit);
                    const int at = len[r] + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    if (hit) wave_lists[r * LIST_STRIDE + at] = (uint8_t)j;
                    len[r] += __popcll(m);
                }
            }
            const int my_len = row == 0 ? len[0] : (row == 1 ? len[1] : (row == 2 ? len[2] : len[3]));
            const int longest = __builtin_amdgcn_readfirstlane(max(max(len[0], len[1]), max(len[2], len[3])));
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");   // the wave reads its own lists back

            for (int b = 0; b < longest; b += 16) {
                const int e = b + s;
                const bool valid = e < my_len;
                const int j = valid ? (int)mine[e] : 0;
                SplatRegs sp;
                {
                    const float4 p = stage[j].xyh;
                    const float4 co = stage[j].co;
                    const float4 c = stage[j].rgbd;
                    sp.x = p.x; sp.y = p.y;
                    sp.A = co.x; sp.B = co.y; sp.C = co.z;
                    sp.o = valid ? co.w : 0.f;   // a lane without an entry: alpha = 0 on every pixel
                    sp.c0 = c.x; sp.c1 = c.y; sp.c2 = c.z;
                    sp.posrel = (float)(MB - 1 - j);
                }
                Sums a = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                block_row<0>(sp, px, a);
                block_row<1>(sp, px, a);
                block_row<2>(sp, px, a);
                block_row<3>(sp, px, a);
                // moments about the block's corner -> sums about the splat centre (dx = X - u, dy = Y - v)
                const float X = sp.x - bx0f, Y = sp.y - by0f;
                v4f lo4 = {a.C0, a.C1, a.C2, a.M0};
                v4f hi4 = {X * a.M0 - a.Mu, Y * a.M0 - a.Mv, X * X * a.M0 - 2.f * X * a.Mu + a.Muu, X * Y * a.M0 - X * a.Mv - Y * a.Mu + a.Muv};
                const float syy = Y * Y * a.M0 - 2.f * Y * a.Mv + a.Mvv;
                // the rows of the wave add to the entry's record one after the other: two rows may hold the same entry
                float *const rec8 = acc8 + ((size_t)wave * MB + j) * 8;
                float *const rec1 = acc1 + wave * MB + j;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    if (row == r && valid) {
                        const v4f o0 = *reinterpret_cast<const v4f *>(rec8), o1 = *reinterpret_cast<const v4f *>(rec8 + 4);
                        const float o2 = *rec1;
                        *reinterpret_cast<v4f *>(rec8) = o0 + lo4;
                        *reinterpret_cast<v4f *>(rec8 + 4) = o1 + hi4;
                        *rec1 = o2 + syy;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                }
            }
            __syncthreads();
            // ---- write the round out: the four waves' records of every staged entry -> the nine per-instance sums ----
            for (int t0 = 0; t0 < n; t0 += TILE_PIX) {
                const int t = t0 + tid;
                if (t < n) {
                    float a[9];
#pragma unroll
                    for (int q = 0; q < 9; q++) a[q] = 0.f;
#pragma unroll
                    for (int w = 0; w < 4; w++) {
                        float *const r8 = acc8 + ((size_t)w * MB + t) * 8;
                        const v4f lo4 = *reinterpret_cast<const v4f *>(r8), hi4 = *reinterpret_cast<const v4f *>(r8 + 4);
                        a[0] += lo4[0]; a[1] += lo4[1]; a[2] += lo4[2]; a[3] += lo4[3];
                        a[4] += hi4[0]; a[5] += hi4[1]; a[6] += hi4[2]; a[7] += hi4[3];
                        a[8] += acc1[w * MB + t];
                        *reinterpret_cast<v4f *>(r8) = v4f{0.f, 0.f, 0.f, 0.f};
                        *reinterpret_cast<v4f *>(r8 + 4) = v4f{0.f, 0.f, 0.f, 0.f};
                        acc1[w * MB + t] = 0.f;
                    }
                    const float4 co = stage[t].co;
                    const float kh = -0.5f * co.w;
                    const float Sgx = kh * a[4], Sgy = kh * a[5];   // -1/2 o sum g dx, dy
                    float *rowp = partial + (size_t)s_slot[t] * NACC;
                    rowp[0] = a[0];
                    rowp[1] = a[1];
                    rowp[2] = a[2];
                    rowp[3] = (Sgx * co.x + Sgy * co.y) * (float)W;        // dL/dmean2D in NDC units: 2 * (W / 2)
                    rowp[4] = (Sgy * co.z + Sgx * co.y) * (float)H;
                    rowp[5] = kh * a[6];
                    rowp[6] = kh * a[7];
                    rowp[7] = kh * a[8];
                    rowp[8] = a[3];
                }
            }
            __syncthreads();   // stage / s_slot / the lists are free for the next round
        }
    }
}

int launch_render_backward_blk(const das3r_raster_args *a, const float *dL_dpix, char *geom, char *binning, char *img, const Layout &L,
                               float *partial, int mb, int slices, hipStream_t s) {
#define ARGS                                                                                                              \
    (const uint2 *)(img + L.pub.ranges), (const uint32_t *)(binning + L.pub.point_list), a->image_width, a->image_height, \
        L.tiles_x, pack_tiles(L), (const float4 *)(geom + L.pub.xy), (const float4 *)(geom + L.pub.conic_opacity),             \
        (const float4 *)(geom + L.pub.rgbd), a->bg, (const float *)(img + L.pub.final_T),                                 \
        (const uint32_t *)(img + L.pub.n_contrib), dL_dpix, (const uint32_t *)(binning + L.b_slot), partial,              \
        (uint32_t)(a->P - 1), (uint32_t)L.capacity, (const float4 *)(binning + L.b_ckpt)
#define GO(MBV) DAS3R_LAUNCH((render_backward_blk_kernel<MBV>), dim3(xcd_grid(L.ntiles), slices > 1 ? slices : 1), dim3(TILE_PIX), 0, s, ARGS)
    if (mb == 64) GO(64);
    else if (mb == 256) GO(256);
    else GO(128);
#undef GO
#undef ARGS
    KERNEL_CHECK(s, a->debug, "render_backward_blk");
    return DAS3R_OK;
}

}  // namespace das3r
