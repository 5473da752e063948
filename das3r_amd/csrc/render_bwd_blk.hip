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
#include "render_blk.h"

namespace das3r {

// MB: staged list entries per round.  PIX: where the walk finds the per-pixel values (render_blk.h): 0 pixel lanes' registers (DPP
// broadcasts), 1 constants from LDS, 2 constants and state from LDS.
// ABL: timing experiments only (DAS3R_ABLATE with DAS3R_RENDER_BWD=blk128p1; results are wrong): 1 no batches at all (what the rounds
// cost without them), 2 batches without the record add, 4 nothing written out, 8 bounding-box block test only, 16 no quadrant ellipse test;
// 32 (every build; das3r_debug_mutate(1)): exp(power) (1 + 1e-4) — the mutation tests/test_gpu_fullsize.py proves the parity bars catch
template <int MB, int PIX, int ABL = 0, int OCC = 5, bool LAST = false>
__global__ void __launch_bounds__(256, OCC) render_backward_blk_kernel(
    const uint2 *__restrict__ ranges, const uint32_t *__restrict__ point_list, int W, int H, int tiles_x, int ntiles_strip /*render_common.h pack_tiles*/,
    const float4 *__restrict__ xyh, const float4 *__restrict__ conic_opacity, const float4 *__restrict__ rgbd,
    const float *__restrict__ bg, const float *__restrict__ final_T, const uint32_t *__restrict__ n_contrib,
    const float *__restrict__ dL_dpix, const uint32_t *__restrict__ slot_list, float *__restrict__ partial /*[I,9]*/,
    uint32_t last_g, uint32_t cap /*bounds of the list contents: render_common.h safe_range*/,
    const float4 *__restrict__ ckpt /*forward's checkpoints of long lists (render_common.h); used when gridDim.y > 1*/,
    unsigned long long *__restrict__ pairs_arg /*common.h pair_counters(): null unless bench.py counts*/) {
#ifdef DAS3R_EXPERIMENTS
    DECODE_PAIRS_OR_TRACE(pairs_arg)   // (tools/wg_trace.py)
#else
    unsigned long long *const pairs = pairs_arg;
#endif
    static_assert(MB >= 64 && MB <= 256 && MB % 8 == 0, "entries per round (one-byte list entries)");
    constexpr int LIST_STRIDE = MB + 4;                                  // bytes; the four rows of a wave read position e of their lists in one instruction
    constexpr int OFF_STAGE = 0;                                         // StagedSplat[MB]
    constexpr int OFF_ACC = OFF_STAGE + MB * (int)sizeof(StagedSplat);   // float4[2][4 waves][MB]: C0 C1 C2 M0 | Sgx Sgy Sxx Sxy — two arrays of
                                                                         // 16-byte rows, not one of 32-byte rows: a row's 16 lanes (random entries) then spread over 16 bank groups instead of 8
    constexpr int OFF_ACC1 = OFF_ACC + 4 * MB * 8 * 4;                   // float[4 waves][MB]: Syy
    constexpr int OFF_SLOT = OFF_ACC1 + 4 * MB * 4;                      // uint32_t[MB]
    constexpr int OFF_LIST = OFF_SLOT + MB * 4;                          // uint8_t[4 waves][4 rows][LIST_STRIDE]
    constexpr int OFF_MAX = OFF_LIST + 16 * LIST_STRIDE;                 // uint32_t[4]
    constexpr int OFF_CST = OFF_MAX + 16;                                // PIX > 0: float4[4 waves][4 rows][16 (+ pad)]: d0, d1, d2, lastrel
    constexpr int OFF_ST = OFF_CST + (PIX > 0 ? 16 * PIX_CST_ROW : 0);   // PIX == 2: float2[4 waves][4 rows][16 (+ pad)]: T, R
    constexpr int LDS_BYTES = OFF_ST + (PIX == 2 ? 16 * PIX_ST_ROW : 0);
    __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
    StagedSplat *const stage = reinterpret_cast<StagedSplat *>(lds + OFF_STAGE);
    float *const acc8 = reinterpret_cast<float *>(lds + OFF_ACC);
    float *const acc1 = reinterpret_cast<float *>(lds + OFF_ACC1);
    uint32_t *const s_slot = reinterpret_cast<uint32_t *>(lds + OFF_SLOT);
    uint32_t *const s_max = reinterpret_cast<uint32_t *>(lds + OFF_MAX);

    const int ntiles = packed_ntiles(ntiles_strip);
    const int tile = xcd_tile(blockIdx.x, ntiles_strip, tiles_x);
    if (tile < 0) return;
    PHASE_BEGIN();   // (experiments build: common.h)
#ifdef DAS3R_EXPERIMENTS
    BLK_STAMP(trace, 5, 0)
#endif
    const int tid = threadIdx.x, lane = __lane_id();
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bx = tile % tiles_x, by = tile / tiles_x;
    const int qx0 = bx * TILE_X + ((wave & 1) << 3), qy0 = by * TILE_Y + ((wave >> 1) << 3);   // the wave's quadrant
    const float qcx = (float)qx0 + 3.5f, qcy = (float)qy0 + 3.5f;
    const uint2 range = safe_range(ranges[tile], cap);
    if (gridDim.y > 1 && (int)blockIdx.y >= max(ckpt_buckets(range), 1)) return;   // (uniform) bucket-parallel launch: this tile's list has no bucket for this workgroup
    const int row = lane >> 4, s = lane & 15;
    uint8_t *const mine = reinterpret_cast<uint8_t *>(lds + OFF_LIST) + (wave * 4 + row) * LIST_STRIDE;
    uint8_t *const wave_lists = reinterpret_cast<uint8_t *>(lds + OFF_LIST) + wave * 4 * LIST_STRIDE;

    // ---- this lane's pixel: pixel s of block `row` of the quadrant ----
    const int ppx = qx0 + ((row & 1) << 2) + (s & 3), ppy = qy0 + ((row >> 1) << 2) + (s >> 2);
    const bool inside = ppx < W && ppy < H;
    PixelRegs px;
    uint32_t last_contributor;
    float my_T_final, my_tfbg;
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
        my_tfbg = my_T_final * (bg[0] * px.d0 + bg[1] * px.d1 + bg[2] * px.d2);
        px.pxf = PIX ? (float)(qx0 + ((row & 1) << 2)) : (float)ppx;
        px.pyf = PIX ? (float)(qy0 + ((row >> 1) << 2)) : (float)ppy;
        px.T = my_T_final;
        px.R = my_tfbg;   // the replay state R always travels with T_final (bg . dL/dpix): only their sum is used
        px.lastrel = 0.f;
    }
    // PIX > 0: the lane's pixel record in LDS (it is pixel s of block row `row`), and the block row the lane walks
    char *const cst_row = lds + OFF_CST + (wave * 4 + row) * PIX_CST_ROW;
    char *const st_row = lds + OFF_ST + (wave * 4 + row) * PIX_ST_ROW;
    if constexpr (PIX > 0) *reinterpret_cast<float4 *>(cst_row + s * 16) = make_float4(px.d0, px.d1, px.d2, 0.f);
    if constexpr (PIX == 2) *reinterpret_cast<float2 *>(st_row + s * 8) = make_float2(px.T, px.R);
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
    PHASE_MARK(8)   // prologue
    const uint32_t tile_contrib = min(max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3])), list_len);
    const int slices = (int)gridDim.y;
    const int nbuckets = slices > 1 ? max(ckpt_buckets(range), 1) : 1;
    const float bx0f = (float)(qx0 + ((row & 1) << 2)), by0f = (float)(qy0 + ((row >> 1) << 2));   // corner pixel of the lane's block: origin of its moments
    int batches_done = 0;   // (wave-uniform) batches of 16 pixel steps = 1024 pairs each

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
        PHASE_MARK(14)   // zero rows of the tail
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
            px.R = R0 + my_tfbg;
            if constexpr (PIX == 2) {
                *reinterpret_cast<float2 *>(st_row + s * 8) = make_float2(px.T, px.R);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");   // (wave-private rows)
            }
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
            // (MB <= 256: a thread stages at most one entry; round 6: its list words were requested a round ahead — staging is ONE trip to memory)
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
            // the pixel's last contributor relative to the round's window: list position of entry j = base + (MB - 1 - j) with
            // base = lo + max_contrib - done_before - MB (may be negative); entry j takes part iff its position < n_contrib
            {
                const long long base = (long long)lo + (long long)max_contrib - done_before - MB;
                const long long rel = (long long)last_contributor - base;
                px.lastrel = (float)(rel < 0 ? 0ll : (rel > MB ? (long long)MB : rel));
                if constexpr (PIX > 0) *reinterpret_cast<float *>(cst_row + s * 16 + 12) = px.lastrel;
            }
            // LAST (long lists: the bucket-parallel launches): the last contributor of each of the wave's four BLOCKS, relative to the round — an
            // entry behind it gets nothing from that block (every pixel's clamp makes its pairs inert) and need not be listed for it.  On the
            // random-depth benchmarks a tile's blocks stop within a few entries of each other (and the short lists of the 1080p benchmarks
            // pay 2 % for the test: not compiled in there); on a real sequence's depth maps (a tile's list is spatially sorted) 19 % of the listed
            // (entry, block) pairs of the DAS3R training shape lie behind their block's last contributor (tools/probes/train_lists.py):
            // backward 0.478 -> 0.412 ms.
            float blk_last[4] = {0.f, 0.f, 0.f, 0.f};
            if constexpr (LAST) {
                int v = (int)px.lastrel;
#pragma unroll
                for (int o = 8; o >= 1; o >>= 1) v = max(v, __shfl_xor(v, o, 64));   // (max over the 16 lanes of the row)
#pragma unroll
                for (int r = 0; r < 4; r++) blk_last[r] = (float)__builtin_amdgcn_readlane(v, 16 * r);
            }
            __syncthreads();
            PHASE_MARK(9)   // staging the round

            // ---- the wave's four row lists (entries in staged order = reverse list order, kept) ----
            int len[4] = {0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < (MB + 63) / 64; k++) {
                const int j = k * 64 + lane;
                const int jc = j < n ? j : 0;
                const float4 p = stage[jc].xyh;
                const float4 co = stage[jc].co;
                // (no exact ellipse-vs-quadrant test in front of the block tests, as render_bwd_scan.hip has it: with the octagon test per
                //  block behind it, it cost more than the few blocks it still removed were worth — 0.405 vs 0.390 ms at 1 M splats)
                const bool qhit = (j < n) && ((ABL & 16) ? rect_hit_tight_tau(p, co, splat_tau(co.w), (float)qx0, (float)qy0) : true);
                float hd1, hd2;
                diagonal_extents(p, co, hd1, hd2);
                const float posrel_j = (float)(MB - 1 - j);   // (SplatRegs::posrel of the walk: the entry takes part for a pixel iff posrel < lastrel)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const bool hit = qhit && (!LAST || posrel_j < blk_last[r]) && block_hit_oct(p.x, p.y, p.z, p.w, (ABL & 8) ? 1e30f : hd1, (ABL & 8) ? 1e30f : hd2,
                                                           (float)(qx0 + ((r & 1) << 2)) + 1.5f, (float)(qy0 + ((r >> 1) << 2)) + 1.5f);
                    const uint64_t m = __ballot(hit);
                    const int at = len[r] + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    if (hit) wave_lists[r * LIST_STRIDE + at] = (uint8_t)j;
                    len[r] += __popcll(m);
                }
            }
            const int my_len = row == 0 ? len[0] : (row == 1 ? len[1] : (row == 2 ? len[2] : len[3]));
            const int longest = (ABL & 1) ? 0 : __builtin_amdgcn_readfirstlane(max(max(len[0], len[1]), max(len[2], len[3])));
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");   // the wave reads its own lists back
            PHASE_MARK(10)   // row lists

            // The rows of a wave add their (block, entry) sums to the entry's record ONE AFTER THE OTHER (two rows may hold the same
            // entry at the same time): four dependent LDS round trips per batch, 0.079 of the kernel's 0.405 ms at 1 M splats.
            // (Taking them for the PREVIOUS batch, one between each pair of image rows of the current one, so that the vector ALU
            //  has work while a record is on its way, was slower: 0.421 -> 0.445 ms with the registers, and the twelve extra live
            //  registers spill inside the loop with the constants in LDS: 0.403 -> 1.23 ms.)
            v4f p_lo4 = {0.f, 0.f, 0.f, 0.f}, p_hi4 = p_lo4;
            float p_syy = 0.f;
            float *p_rec8 = acc8, *p_rec1 = acc1;
            bool p_valid = false;
            auto add_pass = [&](const int r) {
                if (row == r && p_valid && !(ABL & 2)) {
                    const v4f o0 = *reinterpret_cast<const v4f *>(p_rec8), o1 = *reinterpret_cast<const v4f *>(p_rec8 + 4 * MB * 4);
                    const float o2 = *p_rec1;
                    *reinterpret_cast<v4f *>(p_rec8) = o0 + p_lo4;
                    *reinterpret_cast<v4f *>(p_rec8 + 4 * MB * 4) = o1 + p_hi4;
                    *p_rec1 = o2 + p_syy;
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // (compiler only: the next row's read stays behind this write)
            };
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
                Sums a = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                block_row<0, PIX, 0, (ABL & 32) != 0>(sp, px, a, cst_row, st_row);
                block_row<1, PIX, 0, (ABL & 32) != 0>(sp, px, a, cst_row, st_row);
                block_row<2, PIX, 0, (ABL & 32) != 0>(sp, px, a, cst_row, st_row);
                block_row<3, PIX, 0, (ABL & 32) != 0>(sp, px, a, cst_row, st_row);
                // moments about the block's corner -> sums about the splat centre (dx = X - u, dy = Y - v)
                const float X = sp.x - bx0f, Y = sp.y - by0f;
                p_lo4 = v4f{a.C0, a.C1, a.C2, a.M0};
                p_hi4 = v4f{X * a.M0 - a.Mu, Y * a.M0 - a.Mv, X * X * a.M0 - 2.f * X * a.Mu + a.Muu, X * Y * a.M0 - X * a.Mv - Y * a.Mu + a.Muv};
                p_syy = Y * Y * a.M0 - 2.f * Y * a.Mv + a.Mvv;
                p_rec8 = acc8 + ((size_t)wave * MB + j) * 4;
                p_rec1 = acc1 + wave * MB + j;
                p_valid = valid;
                // a row whose list is exhausted has nothing to add (uniform branches)
                if (len[0] > b) add_pass(0);
                if (len[1] > b) add_pass(1);
                if (len[2] > b) add_pass(2);
                if (len[3] > b) add_pass(3);
            }
            PHASE_MARK(11)   // batches
            __syncthreads();
            PHASE_MARK(12)   // waiting for the slowest wave
            // ---- write the round out: the four waves' records of every staged entry -> the nine per-instance sums ----
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
                    if constexpr (ABL & 4) {
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
            PHASE_MARK(13)   // write-out
        }
    }
    PHASE_END(8)
#ifdef DAS3R_EXPERIMENTS
    BLK_STAMP(trace, 5, 1)   // stores issued
    if (trace != nullptr) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        BLK_STAMP(trace, 5, 2)
        if (threadIdx.x == 0 && blockIdx.x < (uint32_t)TRACE_WGS) trace[((size_t)5 * TRACE_WGS + blockIdx.x) * TRACE_STAMPS + 3] = range.y - range.x;
    }
#endif
    if (pairs != nullptr && lane == 0 && batches_done > 0) {
        atomicAdd(pairs + 1, (unsigned long long)batches_done * 1024ull);
        atomicAdd(pairs + 3, (unsigned long long)batches_done * 16ull);
    }
}

int launch_render_backward_blk(const das3r_raster_args *a, const float *dL_dpix, char *geom, char *binning, char *img, const Layout &L,
                               float *partial, int mb, int slices, hipStream_t s) {
#define ARGS                                                                                                              \
    (const uint2 *)(img + L.pub.ranges), (const uint32_t *)(binning + L.pub.point_list), a->image_width, a->image_height, \
        L.tiles_x, pack_tiles(L), (const float4 *)(geom + L.pub.xy), (const float4 *)(geom + L.pub.conic_opacity),             \
        (const float4 *)(geom + L.pub.rgbd), a->bg, (const float *)(img + L.pub.final_T),                                 \
        (const uint32_t *)(img + L.pub.n_contrib), dL_dpix, (const uint32_t *)(binning + L.b_slot), partial,              \
        (uint32_t)(a->P - 1), (uint32_t)L.capacity, (const float4 *)(binning + L.b_ckpt), PAIRS_ARG
#define GO(MBV, PIX, OCC)                                                                                                                             \
    do {                                                                                                                                              \
        if (slices > 1) DAS3R_LAUNCH((render_backward_blk_kernel<MBV, PIX, 0, OCC, true>), dim3(xcd_grid(L), slices), dim3(TILE_PIX), 0, s, ARGS);   \
        else DAS3R_LAUNCH((render_backward_blk_kernel<MBV, PIX, 0, OCC, false>), dim3(xcd_grid(L), 1), dim3(TILE_PIX), 0, s, ARGS);                   \
    } while (0)
    // DAS3R_RENDER_BWD=blk<entries per round>[p<PIX>][o<workgroups per CU>]: blk128 (registers), blk128p1 (constants in LDS), blk160p1o4 ...
    // default (no DAS3R_RENDER_BWD): constants in LDS for 128-entry rounds (1 M splats at 1080p: 0.381 vs 0.407 ms), registers for 192
    // (DAS3R shape: 0.497 vs 0.534 ms)
    const int pix = switches().render_bwd == 6 ? switches().render_bwd_pix : (mb == 128 ? 1 : 0), occ = switches().render_bwd_occ == 4 ? 4 : 5;
#define BY_PIX(MBV, OCC) do { if (pix == 2) GO(MBV, 2, OCC); else if (pix == 1) GO(MBV, 1, OCC); else GO(MBV, 0, OCC); } while (0)
    if (switches().mutate == 1) {   // the mutated kernel (a test aid: include/das3r_raster.h das3r_debug_mutate), in the two shipped default forms
        if (slices > 1) DAS3R_LAUNCH((render_backward_blk_kernel<192, 0, 32, 4, true>), dim3(xcd_grid(L), slices), dim3(TILE_PIX), 0, s, ARGS);
        else DAS3R_LAUNCH((render_backward_blk_kernel<128, 1, 32, 5, false>), dim3(xcd_grid(L), 1), dim3(TILE_PIX), 0, s, ARGS);
    } else
#ifdef DAS3R_EXPERIMENTS
    if (mb == 128 && pix == 1 && switches().ablate_set) {
        const int abl = switches().ablate;
#define GA(A) DAS3R_LAUNCH((render_backward_blk_kernel<128, 1, A>), dim3(xcd_grid(L), slices > 1 ? slices : 1), dim3(TILE_PIX), 0, s, ARGS)
        if (abl == 1) GA(1); else if (abl == 2) GA(2); else if (abl == 4) GA(4); else if (abl == 8) GA(8); else if (abl == 16) GA(16); else if (abl == 3) GA(3); else if (abl == 7) GA(7); else GA(0);
#undef GA
    } else
#endif
    if (mb == 64) BY_PIX(64, 5);
    else if (mb == 256) BY_PIX(256, 5);
    else if (mb == 120) BY_PIX(120, 5);
    else if (mb == 160) BY_PIX(160, 4);
    else if (mb == 192) BY_PIX(192, 4);
    else if (occ == 4) BY_PIX(128, 4);
    else BY_PIX(128, 5);
#undef BY_PIX
#undef GO
#undef ARGS
    KERNEL_CHECK(s, a->debug, "render_backward_blk");
    return DAS3R_OK;
}

}  // namespace das3r
