// render_bwd_scan.hip — K7 with the wave laid out as 4 pixels x 16 SPLATS: nothing is reduced on the vector ALU.
// Replaces upstream:cuda_rasterizer/backward.cu renderCUDA like render_bwd.hip; same inputs, same partial rows.
//
// Why a third decomposition.  With a pixel per lane (render_bwd.hip) every (splat, quadrant) iteration ends in a 64-lane
// reduction of nine sums: 30 % of that kernel, on top of ~15 products per pair that exist only to be reduced.  The MFMA variant
// (render_bwd_mfma.hip) moves the reduction to the matrix cores but has to transpose its operands through a 44 KB LDS slab,
// because v_mfma_f32_16x16x4_f32 contracts over (lane >> 4) and the step index, never over the lanes of a DPP row.
// Here the lanes are laid out the way the matrix core wants its B operand in the first place:
//     lane = 16 r + s     r = pixel of the current 4-pixel step,  s = splat of the current 16-splat batch,
// so the two per-pair scalars (w = alpha T and g = G dL/dalpha) are MFMA operands the moment they are computed, and sixteen
// steps (the 64 pixels of the wave's 8x8 quadrant) accumulate D[9 sums x 16 splats] on the matrix pipe while the vector ALU
// evaluates the next step.  What the layout costs: the per-pixel recurrences (T, R) now run ACROSS the sixteen lanes of a row —
// two 4-step DPP scans per step (product of 1/(1-alpha) for T, sum of w (c . dL/dpix) for R), 8 full-rate instructions, against the
// 18 + 6 cross-lane and ~15 product instructions they replace.  Per 64 pairs: ~45 VALU + 2 MFMA instead of ~110 VALU.
//
// Per staged batch (MB list entries in reverse list order, as in render_bwd.hip) every wave culls the batch against its quadrant
// (ballot), compacts the survivors to a u8 index list in LDS and walks it 16 entries at a time; lane (r, s) keeps splat s of the
// batch in registers for the sixteen steps.  Per-pixel constants (dL/dpix, T_final (bg . dL/dpix), n_contrib) and the running
// state (T, R) live in a 32-byte LDS row per pixel: a step reads them with two broadcast ds_read_b128 (four addresses per
// instruction) and lane 15 of every row writes the state back.  The nine sums of a (wave, splat) leave the accumulator with plain
// LDS stores into a WAVE-PRIVATE region (a wave meets a splat at most once per batch): no atomics anywhere; the four regions are
// added per splat, under the waves' hit masks, when the batch is written out.  Geometry sums travel as raw moments of g about the
// tile centre (exact fp32 FMA chains on the matrix core) and are converted once per (tile, splat), as in render_bwd_mfma.hip.
#include "render_common.h"

namespace das3r {

namespace {

constexpr int NACC = 9;   // C0, C1, C2, M0, Mu, Mv, Muu, Muv, Mvv
typedef float v4f __attribute__((ext_vector_type(4)));

struct PixRow {   // one pixel of a wave's quadrant (32 B, two ds_read_b128)
    float dLp0, dLp1, dLp2, tfbg;   // dL/dpixel, T_final * (bg . dL/dpixel)
    float T, R;                     // replay state (render_common.h: ReplayState)
    uint32_t last;                  // n_contrib: list positions >= last take no part
    float zero;                     // (A operand of the lanes that carry no colour row)
};

// inclusive product / sum over the lanes of each 16-lane DPP row (Hillis-Steele, row_shr 1, 2, 4, 8)
// product: a lane without a source lane must keep its value, which the update_dpp builtin only offers as mov + mov_dpp + mul;
// v_mul_f32_dpp with bound_ctrl off leaves exactly those lanes unwritten.  Four independent chains per statement: a DPP read of
// a register needs two wait states after the VALU write, here the three other chains' instructions.
__device__ __forceinline__ void row_scan_mul_x4(float &x0, float &x1, float &x2, float &x3) {
    asm("s_nop 1\n\t"
        "v_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %2, %2, %2 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %3, %3, %3 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %2, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %3, %3, %3 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %2, %2, %2 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %3, %3, %3 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0"
        : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
}
// sum: the same network (a lane without a source lane keeps its value — what an inclusive scan wants).  Written out like the
// product because the compiler serialises the four chains of a group and pads every step with s_nop.
__device__ __forceinline__ void row_scan_add_x4(float &x0, float &x1, float &x2, float &x3) {
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %3, %3, %3 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %3, %3, %3 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %3, %3, %3 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0"
        : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
}

}  // namespace

template <int MB>
__global__ void __launch_bounds__(256) render_backward_scan_kernel(
    const uint2 *__restrict__ ranges, const uint32_t *__restrict__ point_list, int W, int H, int tiles_x, int ntiles,
    const float4 *__restrict__ xyh, const float4 *__restrict__ conic_opacity, const float4 *__restrict__ rgbd,
    const float *__restrict__ bg, const float *__restrict__ final_T, const uint32_t *__restrict__ n_contrib,
    const float *__restrict__ dL_dpix, const uint32_t *__restrict__ slot_list, float *__restrict__ partial /*[I,9]*/,
    uint32_t last_g, uint32_t cap /*bounds of the list contents: render_common.h safe_range*/) {
    // one LDS object (cdna_hip_programming.md: a second __shared__ array changes the waits the compiler emits)
    constexpr int OFF_STAGE = 0;                                        // StagedSplat[MB]
    constexpr int OFF_ACC8 = OFF_STAGE + MB * (int)sizeof(StagedSplat);  // float[4 waves][MB][8]: sums 0..7 of (wave, staged entry)
    constexpr int OFF_ACC1 = OFF_ACC8 + 4 * MB * 8 * 4;                  // float[4][MB]: sum 8
    constexpr int OFF_PIX = OFF_ACC1 + 4 * MB * 4;                       // PixRow[4][64]
    constexpr int OFF_SLOT = OFF_PIX + 4 * 64 * (int)sizeof(PixRow);     // uint32_t[MB]
    constexpr int OFF_HIT = OFF_SLOT + MB * 4;                           // uint64_t[4][MB / 64]
    constexpr int OFF_LIST = OFF_HIT + 4 * (MB / 64) * 8;                // uint8_t[4][MB]
    constexpr int OFF_MAX = OFF_LIST + 4 * MB;                           // uint32_t[4]
    constexpr int LDS_BYTES = OFF_MAX + 16;
    __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
    StagedSplat *const stage = reinterpret_cast<StagedSplat *>(lds + OFF_STAGE);
    float *const acc8 = reinterpret_cast<float *>(lds + OFF_ACC8);
    float *const acc1 = reinterpret_cast<float *>(lds + OFF_ACC1);
    float *const outp = acc8;   // [MB][9] rows on their way to `partial`: reuses the accumulators once they have been read
    uint32_t *const s_slot = reinterpret_cast<uint32_t *>(lds + OFF_SLOT);
    uint64_t *const s_hit = reinterpret_cast<uint64_t *>(lds + OFF_HIT);
    uint32_t *const s_max = reinterpret_cast<uint32_t *>(lds + OFF_MAX);

    const int tile = xcd_tile(blockIdx.x, ntiles);
    if (tile < 0) return;
    const int tid = threadIdx.x, lane = __lane_id();
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bx = tile % tiles_x, by = tile / tiles_x;
    const int qx0 = bx * TILE_X + ((wave & 1) << 3), qy0 = by * TILE_Y + ((wave >> 1) << 3);   // the wave's quadrant
    const float qcx = (float)qx0 + 3.5f, qcy = (float)qy0 + 3.5f;
    const float tcx = (float)(bx * TILE_X) + 7.5f, tcy = (float)(by * TILE_Y) + 7.5f;   // tile centre: origin of the moments
    const uint2 range = safe_range(ranges[tile], cap);
    PixRow *const pixrow = reinterpret_cast<PixRow *>(lds + OFF_PIX) + wave * 64;
    uint8_t *const list = reinterpret_cast<uint8_t *>(lds + OFF_LIST) + wave * MB;

    // ---- the quadrant's pixels, one per lane (lane = 8 y + x): constants and initial state into the LDS rows ----
    uint32_t last_contributor;
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
        pixrow[lane] = row;
    }
    // no pixel of this tile blended anything past list position max_contrib: start the replay there
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

    // ---- lane (r, s) of the walk: pixel r of a step, splat s of a batch ----
    const int r = lane >> 4, s = lane & 15;
    const float pxfA = (float)(qx0 + r), pxfB = (float)(qx0 + 4 + r);   // even steps: x = r, odd steps: x = 4 + r; y = step / 2
    const float pyf0 = (float)qy0;
    // A operands (constant for the tile): lane 16 r + q supplies weight q of pixel r of the step.  q < 3: dL/dpixel (read from the
    // pixel's LDS row every step), q = 3..8: moment weights 1, u, v, uu, uv, vv of the pixel's offset from the tile centre
    const char *const a1_addr = reinterpret_cast<const char *>(pixrow + r) + (s < 3 ? 4 * s : 28);   // + step * 4 rows
    float A2[16];
#pragma unroll
    for (int kk = 0; kk < 16; kk++) {
        const float u = ((kk & 1) ? pxfB : pxfA) - tcx, v = (pyf0 + (float)(kk >> 1)) - tcy;
        A2[kk] = s == 3 ? 1.f : s == 4 ? u : s == 5 ? v : s == 6 ? u * u : s == 7 ? u * v : s == 8 ? v * v : 0.f;
    }
    const PixRow *const my_rows = pixrow + r;   // + 4 * step

    for (int i = 0; i < rounds; i++) {
        const int done_before = i * MB;
        const int n = min(MB, (int)max_contrib - done_before);
        // stage the batch in reverse list order; entry j holds list position (max_contrib - 1 - done_before - j)
        if (tid < n) {
            const uint32_t pos = range.x + max_contrib - 1 - done_before - tid;
            const uint32_t g = min(point_list[pos], last_g);
            s_slot[tid] = min(slot_list[pos], cap - 1u);
            stage[tid].xyh = xyh[(size_t)g * SPLAT_REC];           // one 64-byte record: a single cache line per splat
            stage[tid].co = conic_opacity[(size_t)g * SPLAT_REC];
            stage[tid].rgbd = rgbd[(size_t)g * SPLAT_REC];
        }
        __syncthreads();

        // the wave's survivors of the batch, compacted (list order = reverse list order of the tile, kept)
        int cnt = 0;
#pragma unroll
        for (int k = 0; k < MB / 64; k++) {
            const int j = k * 64 + lane;
            const bool hit = j < n && quadrant_hit(stage[j].xyh, qcx, qcy);
            const uint64_t m = __ballot(hit);
            if (lane == 0) s_hit[wave * (MB / 64) + k] = m;
            const int at = cnt + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            if (hit) list[at] = (uint8_t)j;
            cnt += __popcll(m);
        }
        cnt = __builtin_amdgcn_readfirstlane(cnt);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");   // the wave reads its own list back

        for (int b = 0; b < cnt; b += 16) {
            const int e = b + s;
            const bool valid = e < cnt;
            const int j = list[valid ? e : cnt - 1];
            const float4 p = stage[j].xyh;
            float4 co = stage[j].co;
            const float4 c = stage[j].rgbd;
            co.w = valid ? co.w : 0.f;   // an empty slot of the last batch: alpha = 0 on every pixel
            const uint32_t position = max_contrib - 1 - done_before - j;   // 0-based list position of the lane's splat
            v4f Dw = {0.f, 0.f, 0.f, 0.f}, Dg = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k4 = 0; k4 < 16; k4 += 4) {
                float am[4], Gm[4], rinv[4], Pinc[4];
                float4 pc[4];
                float stT[4], stR[4], a1[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int kk = k4 + u;
                    const PixRow *row = my_rows + 4 * kk;
                    pc[u] = *reinterpret_cast<const float4 *>(&row->dLp0);
                    const float4 st = *reinterpret_cast<const float4 *>(&row->T);
                    stT[u] = st.x;
                    stR[u] = st.y;
                    a1[u] = *reinterpret_cast<const float *>(a1_addr + kk * 4 * (int)sizeof(PixRow));
                    float dx, dy, G, alpha;
                    const bool ok = pair_alpha(p.x, p.y, co, (kk & 1) ? pxfB : pxfA, pyf0 + (float)(kk >> 1), dx, dy, G, alpha);
                    const bool active = ok & (position < __float_as_uint(st.z));
                    am[u] = active ? alpha : 0.f;
                    Gm[u] = active ? G : 0.f;
                    rinv[u] = __builtin_amdgcn_rcpf(1.f - am[u]);   // v_rcp_f32: T is itself a reconstruction, 1 ulp is noise
                    Pinc[u] = rinv[u];
                }
                row_scan_mul_x4(Pinc[0], Pinc[1], Pinc[2], Pinc[3]);   // lane s: product of 1/(1-alpha) over splats 0..s of the batch
                float T[4], cd[4], wc[4], Sinc[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    T[u] = stT[u] * Pinc[u];                           // transmittance in front of splat s at this pixel
                    const float w = am[u] * T[u];
                    cd[u] = c.x * pc[u].x + c.y * pc[u].y + c.z * pc[u].z;
                    wc[u] = cd[u] * w;
                    Sinc[u] = wc[u];
                    Dw = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[u], w, Dw, 0, 0, 0);
                }
                row_scan_add_x4(Sinc[0], Sinc[1], Sinc[2], Sinc[3]);   // lane s: sum of w (c . dL/dpix) over splats 0..s of the batch
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int kk = k4 + u;
                    const float Rex = (stR[u] - wc[u]) + Sinc[u];      // R behind splat s: the batch's earlier (= farther) splats + state
                    const float dL_dalpha = T[u] * cd[u] - (Rex + pc[u].w) * rinv[u];
                    const float g = Gm[u] * dL_dalpha;
                    Dg = __builtin_amdgcn_mfma_f32_16x16x4f32(A2[kk], g, Dg, 0, 0, 0);
                    if (s == 15) {   // the row's totals are the pixel's state for the next batch
                        PixRow *row = pixrow + r + 4 * kk;
                        *reinterpret_cast<float2 *>(&row->T) = make_float2(T[u], stR[u] + Sinc[u]);
                    }
                }
            }
            // D[q = 4 r + reg][splat s]: rows 0..7 as two 16-byte stores per splat, row 8 on its own
            const v4f D = Dw + Dg;
            float *const a8 = acc8 + ((size_t)wave * MB + j) * 8;
            if (valid && r < 2) *reinterpret_cast<v4f *>(a8 + 4 * r) = D;
            if (valid && r == 2) acc1[wave * MB + j] = D[0];
        }
        __syncthreads();
        // the four waves' sums of every staged entry, moments -> the nine per-instance sums
        float a[NACC];
#pragma unroll
        for (int q = 0; q < NACC; q++) a[q] = 0.f;
        if (tid < n) {
#pragma unroll
            for (int w = 0; w < 4; w++) {
                if ((s_hit[w * (MB / 64) + (tid >> 6)] >> (tid & 63)) & 1ull) {
                    const v4f lo = *reinterpret_cast<const v4f *>(acc8 + ((size_t)w * MB + tid) * 8);
                    const v4f hi = *reinterpret_cast<const v4f *>(acc8 + ((size_t)w * MB + tid) * 8 + 4);
                    a[0] += lo[0]; a[1] += lo[1]; a[2] += lo[2]; a[3] += lo[3];
                    a[4] += hi[0]; a[5] += hi[1]; a[6] += hi[2]; a[7] += hi[3];
                    a[8] += acc1[w * MB + tid];
                }
            }
        }
        __syncthreads();   // every accumulator has been read: the region becomes the output rows
        if (tid < n) {
            const float4 p = stage[tid].xyh;
            const float4 co = stage[tid].co;
            const float M0 = a[3], Mu = a[4], Mv = a[5], Muu = a[6], Muv = a[7], Mvv = a[8];
            const float X = p.x - tcx, Y = p.y - tcy, kh = -0.5f * co.w;
            const float Sgx = kh * (X * M0 - Mu), Sgy = kh * (Y * M0 - Mv);   // -1/2 o sum g dx, dy  (dx = X - u)
            float *row = outp + tid * NACC;
            row[0] = a[0];
            row[1] = a[1];
            row[2] = a[2];
            row[3] = (Sgx * co.x + Sgy * co.y) * (float)W;        // dL/dmean2D in NDC units: 2 * (W / 2)
            row[4] = (Sgy * co.z + Sgx * co.y) * (float)H;
            row[5] = kh * (X * X * M0 - 2.f * X * Mu + Muu);
            row[6] = kh * (X * Y * M0 - X * Mv - Y * Mu + Muv);
            row[7] = kh * (Y * Y * M0 - 2.f * Y * Mv + Mvv);
            row[8] = M0;
        }
        __syncthreads();
        // rows go to the emission slot of their entry (36-byte row stores, 9 lanes each)
        for (int f = tid; f < n * NACC; f += TILE_PIX) {
            const int j = f / NACC, q = f - j * NACC;
            partial[(size_t)s_slot[j] * NACC + q] = outp[f];
        }
        __syncthreads();
    }
}

int launch_render_backward_scan(const das3r_raster_args *a, const float *dL_dpix, char *geom, char *binning, char *img, const Layout &L,
                                float *partial, int mb, hipStream_t s) {
#define ARGS                                                                                                              \
    (const uint2 *)(img + L.pub.ranges), (const uint32_t *)(binning + L.pub.point_list), a->image_width, a->image_height, \
        L.tiles_x, L.ntiles, (const float4 *)(geom + L.pub.xy), (const float4 *)(geom + L.pub.conic_opacity),             \
        (const float4 *)(geom + L.pub.rgbd), a->bg, (const float *)(img + L.pub.final_T),                                 \
        (const uint32_t *)(img + L.pub.n_contrib), dL_dpix, (const uint32_t *)(binning + L.b_slot), partial,              \
        (uint32_t)(a->P - 1), (uint32_t)L.capacity
    if (mb == 128) DAS3R_LAUNCH((render_backward_scan_kernel<128>), dim3(xcd_grid(L.ntiles)), dim3(TILE_PIX), 0, s, ARGS);
    else DAS3R_LAUNCH((render_backward_scan_kernel<256>), dim3(xcd_grid(L.ntiles)), dim3(TILE_PIX), 0, s, ARGS);
#undef ARGS
    KERNEL_CHECK(s, a->debug, "render_backward_scan");
    return DAS3R_OK;
}

}  // namespace das3r
