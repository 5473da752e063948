// render_bwd_scan.hip — K7 with the wave laid out as 4 pixels x 16 SPLATS: nothing is reduced on the vector ALU.
// Replaces upstream:cuda_rasterizer/backward.cu renderCUDA like render_bwd.hip; same inputs, same partial rows.
//
// Why a third decomposition.  With a pixel per lane (render_bwd.hip) every (splat, quadrant) iteration ends in a 64-lane
// reduction of nine sums: 30 % of that kernel, on top of ~15 products per pair that exist only to be reduced.  The first MFMA
// variant (render_bwd_mfma.hip) moves the reduction to the matrix cores but has to transpose its operands through a 44 KB LDS
// slab, and its exact-fp32 v_mfma_f32_16x16x4_f32 holds the SIMD's vector ALU for 32 cycles per instruction.
// Here the lanes are laid out the way the matrix core wants its B operand in the first place:
//     lane = 16 r + s     r = pixel row of the current step,  s = splat of the current 16-splat batch,
// so the two per-pair scalars (w = alpha T and g = G dL/dalpha) are MFMA operands the moment they are computed: split into bf16
// high + low parts (16 mantissa bits together, products accumulated in fp32) they meet ONE constant A operand per pixel group
// (dL/dpixel hi, the six moment weights about the tile centre — exact in bf16 —, dL/dpixel lo) on v_mfma_f32_16x16x32_bf16,
// which takes 16 cycles and leaves the vector ALU to the other waves.  Four MFMAs per half batch (32 pixels x 16 splats) give
// D[12 rows x 16 splats]: the colour sums (hi + lo rows) and the six moments.  What the layout costs: the per-pixel recurrences
// (T, R) now run ACROSS the sixteen lanes of a row — two 4-step DPP scans per step (product of 1/(1-alpha) for T, sum of
// w (c . dL/dpix) for R), four independent chains interleaved (render_scan.h).  Per 64 pairs: ~45 VALU + 2 transcendentals.
//
// Per round (MB list entries in reverse list order, as in render_bwd.hip) every wave culls the entries against its quadrant
// (bounding box, then the exact ellipse-vs-rectangle test), compacts the survivors to an index list in LDS and walks it 16 at a
// time; lane (r, s) keeps splat s of the batch in registers for the sixteen steps.  Per-pixel constants (dL/dpix, T_final
// (bg . dL/dpix), n_contrib) and the running state (T, R) live in a 32-byte LDS row per pixel: a step reads them with two
// broadcast ds_read_b128 (four addresses per instruction) and lane 15 of every row writes the state back (exec-masked
// ds_write2_b32).  The sums of a (wave, splat) leave the accumulator as one 36-byte record in a WAVE-PRIVATE region (a wave
// meets a splat at most once per round): no atomics anywhere; the four regions are added per splat, under the waves' hit masks,
// when the round is written out, the moments become the nine per-instance sums (once per (tile, splat)) and the row goes straight
// to the entry's emission slot.  Long lists are replayed bucket by bucket in parallel workgroups from the forward's checkpoints
// (gridDim.y > 1; common.h BUCKET).  CPU models of the arithmetic: tests/test_scan_model.py, tests/test_bucket_model.py.
#include "render_scan.h"

namespace das3r {

// MB: staged list entries per round.  ATOM: the sums of a (wave, batch) are added to ONE shared accumulator array with LDS atomics
// (small LDS footprint: more workgroups per CU, longer rounds) instead of being stored to wave-private regions and added afterwards.
// ABL: timing experiments only (DAS3R_ABLATE with DAS3R_RENDER_BWD=scan128; results are wrong): 1 no MFMAs, 2 no DPP scans, 4 no batches at
// all (what the rounds cost without them), 8 bounding-box cull only
template <int MB, bool ATOM, int ABL = 0>
__global__ void __launch_bounds__(256, (MB <= 128 ? 4 : 2)) render_backward_scan_kernel(
    const uint2 *__restrict__ ranges, const uint32_t *__restrict__ point_list, int W, int H, int tiles_x, int ntiles_strip /*render_common.h pack_tiles*/,
    const float4 *__restrict__ xyh, const float4 *__restrict__ conic_opacity, const float4 *__restrict__ rgbd,
    const float *__restrict__ bg, const float *__restrict__ final_T, const uint32_t *__restrict__ n_contrib,
    const float *__restrict__ dL_dpix, const uint32_t *__restrict__ slot_list, float *__restrict__ partial /*[I,9]*/,
    uint32_t last_g, uint32_t cap /*bounds of the list contents: render_common.h safe_range*/,
    const float4 *__restrict__ ckpt /*forward's checkpoints of long lists (render_common.h); used when gridDim.y > 1*/,
    unsigned long long *__restrict__ pairs /*common.h pair_counters(): null unless bench.py counts*/) {
    // one LDS object (cdna_hip_programming.md: a second __shared__ array changes the waits the compiler emits)
    constexpr int NROW = ATOM ? 12 : 9;   // private regions: C0 C1 C2 M0 | Mu Mv Muu Muv (two 16-byte pieces) + Mvv in its own array; shared: + C0' C1' C2'
    constexpr int NROW8 = ATOM ? 12 : 8;
    constexpr int PIX_ROW = 8 * (int)sizeof(PixRow) + 16;                // one image row of a quadrant (+16: the four rows a step reads sit on different banks)
    constexpr int PIX_WAVE = 8 * PIX_ROW;
    constexpr int OFF_STAGE = 0;                                         // StagedSplat[MB]
    constexpr int OFF_ACC = OFF_STAGE + MB * (int)sizeof(StagedSplat);   // float[ATOM ? 1 : 4 waves][MB][12 | 8]
    constexpr int OFF_ACC1 = OFF_ACC + (ATOM ? 1 : 4) * MB * NROW8 * 4;   // float[4 waves][MB]: Mvv (private regions only)
    constexpr int OFF_PIX = OFF_ACC1 + (ATOM ? 0 : 4 * MB * 4);          // 4 waves x 8 rows of PixRow[8]
    constexpr int OFF_SLOT = OFF_PIX + 4 * PIX_WAVE;                     // uint32_t[MB]
    constexpr int OFF_HIT = OFF_SLOT + MB * 4;                           // uint64_t[4][MB / 64]
    constexpr int OFF_LIST = OFF_HIT + 4 * (MB / 64) * 8;                // uint8_t / uint16_t [4][MB]
    constexpr int OFF_MAX = OFF_LIST + 4 * MB * 2;                       // uint32_t[4]
    constexpr int LDS_BYTES = OFF_MAX + 16;
    __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
    StagedSplat *const stage = reinterpret_cast<StagedSplat *>(lds + OFF_STAGE);
    float *const acc = reinterpret_cast<float *>(lds + OFF_ACC);
    float *const acc1 = reinterpret_cast<float *>(lds + OFF_ACC1);
    uint32_t *const s_slot = reinterpret_cast<uint32_t *>(lds + OFF_SLOT);
    uint64_t *const s_hit = reinterpret_cast<uint64_t *>(lds + OFF_HIT);
    uint32_t *const s_max = reinterpret_cast<uint32_t *>(lds + OFF_MAX);

    const int ntiles = packed_ntiles(ntiles_strip);
    const int tile = xcd_tile(blockIdx.x, ntiles_strip, tiles_x);
    if (tile < 0) return;
    const int tid = threadIdx.x, lane = __lane_id();
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bx = tile % tiles_x, by = tile / tiles_x;
    const int qx0 = bx * TILE_X + ((wave & 1) << 3), qy0 = by * TILE_Y + ((wave >> 1) << 3);   // the wave's quadrant
    const float qcx = (float)qx0 + 3.5f, qcy = (float)qy0 + 3.5f;
    const float tcx = (float)(bx * TILE_X) + 7.5f, tcy = (float)(by * TILE_Y) + 7.5f;   // tile centre: origin of the moments
    const uint2 range = safe_range(ranges[tile], cap);
    char *const pix_wave = lds + OFF_PIX + wave * PIX_WAVE;
    uint16_t *const list = reinterpret_cast<uint16_t *>(lds + OFF_LIST) + wave * MB;
    auto pix_at = [&](const int x, const int y) { return reinterpret_cast<PixRow *>(pix_wave + y * PIX_ROW) + x; };

    // ---- the quadrant's pixels, one per lane (lane = 8 y + x): constants and initial state into the LDS rows ----
    uint32_t last_contributor;
    float my_T_final, my_dLp0, my_dLp1, my_dLp2;   // this lane's pixel (lane = 8 y + x), kept for the per-bucket state
    {
        const int px = qx0 + (lane & 7), py = qy0 + (lane >> 3);
        const bool inside = px < W && py < H;
        const size_t pix = (size_t)py * W + px, plane = (size_t)H * W;
        const float T_final = inside ? final_T[pix] : 0.f;
        last_contributor = inside ? n_contrib[pix] : 0u;
        float dLp0 = 0.f, dLp1 = 0.f, dLp2 = 0.f;
        if (inside) {
            dLp0 = dL_dpix[pix];
            dLp1 = dL_dpix[plane + pix];
            dLp2 = dL_dpix[2 * plane + pix];
        }
        PixRow row;
        row.dLp0 = dLp0;
        row.dLp1 = dLp1;
        row.dLp2 = dLp2;
        row.tfbg = T_final * (bg[0] * dLp0 + bg[1] * dLp1 + bg[2] * dLp2);
        row.T = T_final;
        row.R = 0.f;
        row.last = last_contributor;
        row.zero = 0.f;
        *pix_at(lane & 7, lane >> 3) = row;
        my_T_final = T_final;
        my_dLp0 = dLp0;
        my_dLp1 = dLp1;
        my_dLp2 = dLp2;
    }
    // no pixel of this tile blended anything past list position max_contrib: start the replay there
    uint32_t mx = last_contributor;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, o, 64));
    if (lane == 0) s_max[wave] = mx;
    __syncthreads();
    const uint32_t list_len = range.y - range.x;
    const uint32_t tile_contrib = min(max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3])), list_len);
    // Bucket-parallel replay (gridDim.y = slices > 1): this workgroup takes buckets blockIdx.y, blockIdx.y + slices, ... of the
    // tile's list; a bucket starts from the pixel states the forward left at its far end (checkpoints).  slices == 1: the whole list
    // as one "bucket", starting from final_T.
    const int slices = (int)gridDim.y;
    const int nbuckets = slices > 1 ? max(ckpt_buckets(range), 1) : 1;
    // ---- lane (r, s) of the walk: splat s of a batch; image rows r and 4 + r of the quadrant, one column per step ----
    // (the K index of v_mfma_f32_16x16x32_bf16 is 8 (lane >> 4) + element: a lane supplies eight consecutive k = the eight pixels
    //  of ITS image row, so the sixteen steps of a batch are: columns 0..7 of rows r, then columns 0..7 of rows 4 + r)
    const int r = lane >> 4, s = lane & 15;
    const float pyfA = (float)(qy0 + r), pyfB = (float)(qy0 + 4 + r);
    const float pxf0 = (float)qx0;
    const unsigned long long lanes15 = 0x8000800080008000ull;
    // (LDS byte addresses of the lane's two image rows: the low 32 bits of a generic pointer into LDS are its LDS address)
    const uint32_t row_addr[2] = {(uint32_t)(uintptr_t)(pix_wave + r * PIX_ROW), (uint32_t)(uintptr_t)(pix_wave + (4 + r) * PIX_ROW)};
    // A operand (constant for the tile), rows q = lane & 15 of the 16 x 32 weight matrix of each half:
    //   q 0..2  bf16 high part of dL/dpixel (channel q)      q 3..8  moment weights 1, u, v, uu, uv, vv about the tile centre — half-
    //   q 9..11 bf16 low part of dL/dpixel (channel q - 9)           integers below 8 and their products: EXACT in bf16's 8 bits
    // every MFMA of a batch uses it: rows 0..2 + 9..11 of (A w) give the colour sums, rows 3..8 of (A g) the moments; the rest is ignored
    U4 Aop[2];
#pragma unroll
    for (int half = 0; half < 2; half++) {
        float wv[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const PixRow *row = pix_at(i, 4 * half + r);
            const float u = (float)(qx0 + i) - tcx, v = (float)(qy0 + 4 * half + r) - tcy;
            const int ch = s < 3 ? s : s - 9;
            const float d = (s < 3 || (s >= 9 && s < 12)) ? (ch == 0 ? row->dLp0 : ch == 1 ? row->dLp1 : row->dLp2) : 0.f;
            const float mom = s == 3 ? 1.f : s == 4 ? u : s == 5 ? v : s == 6 ? u * u : s == 7 ? u * v : s == 8 ? v * v : 0.f;
            wv[i] = s < 3 ? d : (s >= 9 && s < 12) ? lo_part(d) : mom;
        }
        Aop[half] = U4{pack_hi(wv[0], wv[1]), pack_hi(wv[2], wv[3]), pack_hi(wv[4], wv[5]), pack_hi(wv[6], wv[7])};
    }

    int batches_done = 0;   // (wave-uniform) batches of 16 steps = 1024 pairs each
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
            const int cpix = (((wave >> 1) << 3) + (lane >> 3)) * 16 + ((wave & 1) << 3) + (lane & 7);
            const float4 far = ckpt_slot(const_cast<float4 *>(ckpt), range, tile, bk)[cpix];            // (T, C) in front of position hi
            const float4 fin = ckpt_slot(const_cast<float4 *>(ckpt), range, tile, nbuckets - 1)[cpix];  // final (T, C)
            T0 = far.x;
            R0 = my_dLp0 * (fin.y - far.y) + my_dLp1 * (fin.z - far.z) + my_dLp2 * (fin.w - far.w);   // (c . dL/dpix) alpha T of everything behind
        }
        PixRow *row = pix_at(lane & 7, lane >> 3);
        row->T = T0;
        row->R = R0;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");   // (wave-private rows)
    }
    for (int i = 0; i < rounds; i++) {
        const int done_before = i * MB;
        const int n = min(MB, (int)max_contrib - done_before);
        // stage the batch in reverse list order; entry j holds list position (max_contrib - 1 - done_before - j)
        for (int t = tid; t < n; t += TILE_PIX) {
            const uint32_t pos = range.x + lo + max_contrib - 1 - done_before - t;
            const uint32_t g = min(point_list[pos], last_g);
            s_slot[t] = min(slot_list[pos], cap - 1u);
            stage[t].xyh = xyh[(size_t)g * SPLAT_REC];           // one 64-byte record: a single cache line per splat
            stage[t].co = conic_opacity[(size_t)g * SPLAT_REC];
            stage[t].rgbd = rgbd[(size_t)g * SPLAT_REC];
        }
        if constexpr (ATOM) {
            for (int f = tid; f < MB * NROW; f += TILE_PIX) acc[f] = 0.f;
        }
        __syncthreads();

        // the wave's survivors of the batch, compacted (list order = reverse list order of the tile, kept)
        int cnt = 0;
#pragma unroll
        for (int k = 0; k < MB / 64; k++) {
            const int j = k * 64 + lane;
            const int jc = j < n ? j : 0;
            const bool hit = j < n && quadrant_hit(stage[jc].xyh, qcx, qcy) && ((ABL & 8) || rect_hit_tight(stage[jc].xyh, stage[jc].co, (float)qx0, (float)qy0));
            const uint64_t m = __ballot(hit);
            if (!ATOM && lane == 0) s_hit[wave * (MB / 64) + k] = m;
            const int at = cnt + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            if (hit) list[at] = (uint16_t)j;
            cnt += __popcll(m);
        }
        cnt = (ABL & 4) ? 0 : __builtin_amdgcn_readfirstlane(cnt);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");   // the wave reads its own list back

        batches_done += (cnt + 15) >> 4;
        for (int b = 0; b < cnt; b += 16) {
            const int e = b + s;
            const bool valid = e < cnt;
            const int j = list[valid ? e : cnt - 1];
            const float4 p = stage[j].xyh;
            float4 co = stage[j].co;
            const float4 c = lds_read4(&stage[j].rgbd);   // (b128, not b96: half the LDS cycles)
            co.w = valid ? co.w : 0.f;   // an empty slot of the last batch: alpha = 0 on every pixel
            const uint32_t position = lo + max_contrib - 1 - done_before - j;   // 0-based list position of the lane's splat
            v4f Dw = {0.f, 0.f, 0.f, 0.f}, Dg = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int half = 0; half < 2; half++) {
                const float pyf = half ? pyfB : pyfA;
                const float dy = p.y - pyf;
                const char *const rows = pix_wave + (4 * half + r) * PIX_ROW;
                uint32_t whi[4], wlo[4], ghi[4], glo[4];
#pragma unroll
                for (int k4 = 0; k4 < 8; k4 += 4) {
                    float am[4], Gm[4], rinv[4], Pinc[4], w4[4], g4[4];
                    float4 pc[4];
                    float stT[4], stR[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const PixRow *row = reinterpret_cast<const PixRow *>(rows) + (k4 + u);
                        pc[u] = *reinterpret_cast<const float4 *>(&row->dLp0);
                        const float4 st = *reinterpret_cast<const float4 *>(&row->T);
                        stT[u] = st.x;
                        stR[u] = st.y;
                        // pair_alpha's arithmetic (render_common.h), dy shared by the eight steps of the half
                        const float dx = p.x - (pxf0 + (float)(k4 + u));
                        const float q = __fmaf_rn(__fmul_rn(co.x, dx), dx, __fmul_rn(__fmul_rn(co.z, dy), dy));
                        const float power = __fmaf_rn(-0.5f, q, -__fmul_rn(__fmul_rn(co.y, dx), dy));
                        const float G = __expf(power);
                        const float alpha = fminf(0.99f, __fmul_rn(co.w, G));
                        const bool active = (!(power > 0.0f)) & (alpha >= (1.0f / 255.0f)) & (position < __float_as_uint(st.z));
                        am[u] = active ? alpha : 0.f;
                        Gm[u] = active ? G : 0.f;
                        rinv[u] = __builtin_amdgcn_rcpf(1.f - am[u]);   // v_rcp_f32: T is itself a reconstruction, 1 ulp is noise
                        Pinc[u] = rinv[u];
                    }
                    if constexpr (!(ABL & 2)) row_scan_mul_x4(Pinc[0], Pinc[1], Pinc[2], Pinc[3]);   // lane s: product of 1/(1-alpha) over splats 0..s of the batch
                    float T[4], cd[4], wc[4], Sinc[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        T[u] = stT[u] * Pinc[u];                           // transmittance in front of splat s at this pixel
                        w4[u] = am[u] * T[u];
                        cd[u] = c.x * pc[u].x + c.y * pc[u].y + c.z * pc[u].z;
                        wc[u] = cd[u] * w4[u];
                        Sinc[u] = wc[u];
                    }
                    if constexpr (!(ABL & 2)) row_scan_add_x4(Sinc[0], Sinc[1], Sinc[2], Sinc[3]);   // lane s: sum of w (c . dL/dpix) over splats 0..s of the batch
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const float Rex = (stR[u] - wc[u]) + Sinc[u];      // R behind splat s: the batch's earlier (= farther) splats + state
                        const float dL_dalpha = T[u] * cd[u] - (Rex + pc[u].w) * rinv[u];
                        g4[u] = Gm[u] * dL_dalpha;
                        stR[u] += Sinc[u];
                    }
                    // the row's totals (lane 15) are the pixels' state for the next batch: PixRow.T / .R = dwords 4, 5 of the 8-dword row
                    if (k4 == 0) {
                        store2_lane15<4>(row_addr[half], T[0], stR[0], lanes15);
                        store2_lane15<12>(row_addr[half], T[1], stR[1], lanes15);
                        store2_lane15<20>(row_addr[half], T[2], stR[2], lanes15);
                        store2_lane15<28>(row_addr[half], T[3], stR[3], lanes15);
                    } else {
                        store2_lane15<36>(row_addr[half], T[0], stR[0], lanes15);
                        store2_lane15<44>(row_addr[half], T[1], stR[1], lanes15);
                        store2_lane15<52>(row_addr[half], T[2], stR[2], lanes15);
                        store2_lane15<60>(row_addr[half], T[3], stR[3], lanes15);
                    }
                    // the two scalars of the pair as bf16 high + low parts, two steps per register (element = column of the row)
#pragma unroll
                    for (int u = 0; u < 4; u += 2) {
                        const int e2 = (k4 + u) >> 1;
                        whi[e2] = pack_hi(w4[u], w4[u + 1]);
                        wlo[e2] = pack_hi(lo_part(w4[u]), lo_part(w4[u + 1]));
                        ghi[e2] = pack_hi(g4[u], g4[u + 1]);
                        glo[e2] = pack_hi(lo_part(g4[u]), lo_part(g4[u + 1]));
                    }
                }
                if constexpr (ABL & 1) {
                    asm volatile("" ::"v"(whi[0]), "v"(whi[1]), "v"(whi[2]), "v"(whi[3]), "v"(wlo[0]), "v"(wlo[1]), "v"(wlo[2]), "v"(wlo[3]));
                    asm volatile("" ::"v"(ghi[0]), "v"(ghi[1]), "v"(ghi[2]), "v"(ghi[3]), "v"(glo[0]), "v"(glo[1]), "v"(glo[2]), "v"(glo[3]));
                } else {
                    const v8bf a = as_bf8(Aop[half]);
                    Dw = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, as_bf8(U4{whi[0], whi[1], whi[2], whi[3]}), Dw, 0, 0, 0);
                    Dg = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, as_bf8(U4{ghi[0], ghi[1], ghi[2], ghi[3]}), Dg, 0, 0, 0);
                    Dw = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, as_bf8(U4{wlo[0], wlo[1], wlo[2], wlo[3]}), Dw, 0, 0, 0);
                    Dg = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, as_bf8(U4{glo[0], glo[1], glo[2], glo[3]}), Dg, 0, 0, 0);
                }
            }
            // lane (r, s) holds rows 4 r .. 4 r + 3 of both products for splat s: the twelve useful ones leave as one 16-byte row piece per lane
            //   r = 0: C0 C1 C2 (A w) | M0 (A g)      r = 1: Mu Mv Muu Muv (A g)      r = 2: Mvv (A g) | C0' C1' C2' (A w)
            if constexpr (ATOM) {
                v4f D;
                D[0] = r == 0 ? Dw[0] : Dg[0];
                D[1] = r == 1 ? Dg[1] : Dw[1];
                D[2] = r == 1 ? Dg[2] : Dw[2];
                D[3] = r == 2 ? Dw[3] : Dg[3];
                float *const a12 = acc + (size_t)j * NROW + 4 * r;
                if (valid && r < 3) {
#pragma unroll
                    for (int q = 0; q < 4; q++) __hip_atomic_fetch_add(a12 + q, D[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            } else {
                // colour rows: high-part products in lanes r = 0 (rows 0..2), low-part products in lanes r = 2 (rows 9..11): the upper
                // half of the wave hands its three values down (v_permlane32_swap), so that a (wave, entry) record is 36 bytes
                v4f D;
#pragma unroll
                for (int q = 0; q < 3; q++) {
                    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(Dw[q + 1]), 0u, false, false);
                    D[q] = Dw[q] + __uint_as_float(sw[1]);   // (lanes 0..15: own row q + row 9 + q of lanes 32..47)
                }
                D[3] = Dg[3];
                if (r == 1) D = Dg;
                if (valid && r < 2) *reinterpret_cast<v4f *>(acc + ((size_t)wave * MB + j) * 8 + 4 * r) = D;
                if (valid && r == 2) acc1[wave * MB + j] = Dg[0];
            }
        }
        __syncthreads();
        // the four waves' sums of every staged entry, moments -> the nine per-instance sums
        for (int t0 = 0; t0 < n; t0 += TILE_PIX) {
            const int t = t0 + tid;
            float a[12];
#pragma unroll
            for (int q = 0; q < 12; q++) a[q] = 0.f;
            if (t < n) {
                if constexpr (ATOM) {
#pragma unroll
                    for (int q = 0; q < NROW; q += 4) {
                        const v4f v = *reinterpret_cast<const v4f *>(acc + (size_t)t * NROW + q);
                        a[q] = v[0]; a[q + 1] = v[1]; a[q + 2] = v[2]; a[q + 3] = v[3];
                    }
                } else {
#pragma unroll
                    for (int w = 0; w < 4; w++) {
                        if ((s_hit[w * (MB / 64) + (t >> 6)] >> (t & 63)) & 1ull) {
                            const v4f lo4 = *reinterpret_cast<const v4f *>(acc + ((size_t)w * MB + t) * 8);
                            const v4f hi4 = *reinterpret_cast<const v4f *>(acc + ((size_t)w * MB + t) * 8 + 4);
                            a[0] += lo4[0]; a[1] += lo4[1]; a[2] += lo4[2]; a[3] += lo4[3];
                            a[4] += hi4[0]; a[5] += hi4[1]; a[6] += hi4[2]; a[7] += hi4[3];
                            a[8] += acc1[w * MB + t];
                        }
                    }
                }
            }
            // the row goes straight to the emission slot of its entry: rows of one tile are scattered over `partial` anyway (slots
            // follow the splat order), so bouncing them through LDS for "coalesced" 9-lane stores bought nothing but two barriers
            if (t < n) {
                const float4 p = stage[t].xyh;
                const float4 co = stage[t].co;
                const float M0 = a[3], Mu = a[4], Mv = a[5], Muu = a[6], Muv = a[7], Mvv = a[8];
                const float X = p.x - tcx, Y = p.y - tcy, kh = -0.5f * co.w;
                const float Sgx = kh * (X * M0 - Mu), Sgy = kh * (Y * M0 - Mv);   // -1/2 o sum g dx, dy  (dx = X - u)
                float *row = partial + (size_t)s_slot[t] * NACC;
                row[0] = ATOM ? a[0] + a[NROW - 3] : a[0];
                row[1] = ATOM ? a[1] + a[NROW - 2] : a[1];
                row[2] = ATOM ? a[2] + a[NROW - 1] : a[2];
                row[3] = (Sgx * co.x + Sgy * co.y) * (float)W;        // dL/dmean2D in NDC units: 2 * (W / 2)
                row[4] = (Sgy * co.z + Sgx * co.y) * (float)H;
                row[5] = kh * (X * X * M0 - 2.f * X * Mu + Muu);
                row[6] = kh * (X * Y * M0 - X * Mv - Y * Mu + Muv);
                row[7] = kh * (Y * Y * M0 - 2.f * Y * Mv + Mvv);
                row[8] = M0;
            }
        }
        __syncthreads();   // stage / s_slot / the accumulators are free for the next round
    }
    }
    if (pairs != nullptr && lane == 0 && batches_done > 0) {
        atomicAdd(pairs + 1, (unsigned long long)batches_done * 1024ull);
        atomicAdd(pairs + 3, (unsigned long long)batches_done * 16ull);
    }
}

int launch_render_backward_scan(const das3r_raster_args *a, const float *dL_dpix, char *geom, char *binning, char *img, const Layout &L,
                                float *partial, int mb, int slices, hipStream_t s) {
#define ARGS                                                                                                              \
    (const uint2 *)(img + L.pub.ranges), (const uint32_t *)(binning + L.pub.point_list), a->image_width, a->image_height, \
        L.tiles_x, pack_tiles(L), (const float4 *)(geom + L.pub.xy), (const float4 *)(geom + L.pub.conic_opacity),             \
        (const float4 *)(geom + L.pub.rgbd), a->bg, (const float *)(img + L.pub.final_T),                                 \
        (const uint32_t *)(img + L.pub.n_contrib), dL_dpix, (const uint32_t *)(binning + L.b_slot), partial,              \
        (uint32_t)(a->P - 1), (uint32_t)L.capacity, (const float4 *)(binning + L.b_ckpt), pair_counters()
#define GO(MBV, AT, AB) DAS3R_LAUNCH((render_backward_scan_kernel<MBV, AT, AB>), dim3(xcd_grid(L), slices > 1 ? slices : 1), dim3(TILE_PIX), 0, s, ARGS)
    const int abl = switches().ablate_set ? switches().ablate : 0;
    // mb: 64 / 128 / 256 private accumulator regions; 1256: 256 entries per round with the atomic flush
    if (mb == 128 && abl == 1) GO(128, false, 1);
    else if (mb == 128 && abl == 2) GO(128, false, 2);
    else if (mb == 128 && abl == 3) GO(128, false, 3);
    else if (mb == 128 && abl == 4) GO(128, false, 4);
    else if (mb == 128 && abl == 8) GO(128, false, 8);
    else if (mb == 64) GO(64, false, 0);
    else if (mb == 128) GO(128, false, 0);
    else if (mb >= 1000) GO(256, true, 0);
    else GO(256, false, 0);
#undef GO
#undef ARGS
    KERNEL_CHECK(s, a->debug, "render_backward_scan");
    return DAS3R_OK;
}

}  // namespace das3r
