// render_bwd_stream.hip — K7 as INDEPENDENT waves: every wave64 owns one 8x8 quadrant of a tile and streams the tile's list
// by itself; no workgroup barrier after the set-up, no staging rounds, no partial batches but the last.
// Replaces upstream:cuda_rasterizer/backward.cu renderCUDA.  Arithmetic of a step: render_bwd_scan.hip (lanes =
// 4 pixels x 16 splats, recurrences as DPP row scans, the nine sums as split-bf16 products on the matrix cores).
//
// What the round-synchronous kernels pay for sharing a staged batch between the four waves of a tile (measured at 1 M splats,
// 1080p, render_bwd_scan.hip): the batch loop is 420 of 540 us but a third of its steps run on the empty slots of partial
// batches (a wave finds ~33 survivors in a 128-entry round: three 16-splat batches), 35 % of the wave time is spent waiting at
// barriers and on the staging loads in front of them, and the accumulators the waves meet in cost 24 KB of LDS (3 workgroups
// per CU).  Here a wave
//   * reads 64 list entries at a time STRAIGHT INTO REGISTERS (lane = entry; the next chunk's records and the indices of the
//     chunk after it are in flight while the current one is processed),
//   * culls them against its quadrant (bounding box, then the exact ellipse / rectangle test of render_scan.h) and appends the
//     survivors to a small queue in its own LDS (structure of arrays: a batch reads 16 consecutive 16-byte slots, no conflicts),
//   * runs a 16-splat batch whenever the queue holds 16: a batch is always full, except the very last one of the list,
//   * converts the batch's moments to the nine sums in place and stores them as one 48-byte row per (list entry, quadrant)
//     straight to global memory: rows[4 * emission slot + quadrant][12], plus a byte per (entry, quadrant) that says a row exists
//     (the byte array is zeroed by a memset in front of the kernel).  The per-Gaussian backward adds the rows that exist.
// Per wave the walk starts at ITS quadrant's deepest contributor (not the tile's).  LDS: 6.3 KB per wave, 25 KB per tile.
#include "render_scan.h"

namespace das3r {

namespace {
constexpr int QCAP = 80;   // queue capacity: 15 left over + 64 appended
constexpr int PIX_ROW = 8 * (int)sizeof(PixRow) + 16;   // one image row of a quadrant (+16: the four rows a step reads sit on different banks)
constexpr int PIX_WAVE = 8 * PIX_ROW;
constexpr int Q_XYH = PIX_WAVE, Q_CO = Q_XYH + QCAP * 16, Q_RGBP = Q_CO + QCAP * 16, Q_SLOT = Q_RGBP + QCAP * 16, Q_OUT = Q_SLOT + QCAP * 4;
constexpr int WAVE_LDS = Q_OUT + 16 * 48;   // + the batch's 16 x 12 sums on their way out
}  // namespace

// WPS: waves per SIMD the register allocation is held to (= workgroups per CU; 4: 128 VGPRs, no spills)
template <int ABL, int WPS = 4>
__global__ void __launch_bounds__(256, WPS) render_backward_stream_kernel(
    const uint2 *__restrict__ ranges, const uint32_t *__restrict__ point_list, int W, int H, int tiles_x, int ntiles_strip /*render_common.h pack_tiles*/,
    const float4 *__restrict__ xyh, const float4 *__restrict__ conic_opacity, const float4 *__restrict__ rgbd,
    const float *__restrict__ bg, const float *__restrict__ final_T, const uint32_t *__restrict__ n_contrib,
    const float *__restrict__ dL_dpix, const uint32_t *__restrict__ slot_list, float *__restrict__ rows /*[cap][4][12]*/,
    uint8_t *__restrict__ row_exists /*[cap][4], zero on entry*/, uint32_t last_g, uint32_t cap) {
    __shared__ __attribute__((aligned(16))) char lds[4 * WAVE_LDS];
    const int ntiles = packed_ntiles(ntiles_strip);
    const int tile = xcd_tile(blockIdx.x, ntiles_strip, tiles_x);
    if (tile < 0) return;
    const int lane = __lane_id();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int bx = tile % tiles_x, by = tile / tiles_x;
    const int qx0 = bx * TILE_X + ((wave & 1) << 3), qy0 = by * TILE_Y + ((wave >> 1) << 3);   // the wave's quadrant
    const float qcx = (float)qx0 + 3.5f, qcy = (float)qy0 + 3.5f;
    const float tcx = (float)(bx * TILE_X) + 7.5f, tcy = (float)(by * TILE_Y) + 7.5f;   // tile centre: origin of the moments
    const uint2 range = safe_range(ranges[tile], cap);
    char *const my = lds + wave * WAVE_LDS;
    auto pix_at = [&](const int x, const int y) { return reinterpret_cast<PixRow *>(my + y * PIX_ROW) + x; };
    float4 *const q_xyh = reinterpret_cast<float4 *>(my + Q_XYH);
    float4 *const q_co = reinterpret_cast<float4 *>(my + Q_CO);
    float4 *const q_rgbp = reinterpret_cast<float4 *>(my + Q_RGBP);
    uint32_t *const q_slot = reinterpret_cast<uint32_t *>(my + Q_SLOT);
    float *const q_out = reinterpret_cast<float *>(my + Q_OUT);

    // ---- the quadrant's pixels, one per lane (lane = 8 y + x): constants and initial state into the wave's LDS rows ----
    uint32_t start;
    {
        const int px = qx0 + (lane & 7), py = qy0 + (lane >> 3);
        const bool inside = px < W && py < H;
        const size_t pix = (size_t)py * W + px, plane = (size_t)H * W;
        const float T_final = inside ? final_T[pix] : 0.f;
        const uint32_t last_contributor = inside ? n_contrib[pix] : 0u;
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
        // no pixel of this QUADRANT blended anything past list position `start`: the wave's replay begins there
        uint32_t mx = last_contributor;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, o, 64));
        start = (uint32_t)__builtin_amdgcn_readfirstlane((int)min(mx, range.y - range.x));
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");   // (wave-private LDS: orders this wave's own stores and loads)

    // ---- lane (r, s) of a batch: splat s; image rows r and 4 + r of the quadrant, one column per step (render_bwd_scan.hip) ----
    const int r = lane >> 4, s = lane & 15;
    const float pyfA = (float)(qy0 + r), pyfB = (float)(qy0 + 4 + r);
    const float pxf0 = (float)qx0;
    const unsigned long long lanes15 = 0x8000800080008000ull;
    const uint32_t row_addr[2] = {(uint32_t)(uintptr_t)(my + r * PIX_ROW), (uint32_t)(uintptr_t)(my + (4 + r) * PIX_ROW)};
    // A operand (constant for the tile): rows q = lane & 15 of the 16 x 32 weight matrix of each half —
    //   q 0..2 bf16 high part of dL/dpixel | 3..8 moment weights 1, u, v, uu, uv, vv about the tile centre (exact in bf16) | 9..11 low part of dL/dpixel
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

    // one 16-splat batch from queue positions head .. head + 15 (entries >= fill are empty slots: only in the last batch of the list)
    auto run_batch = [&](const int head, const int fill) {
        const int e = head + s;
        const bool valid = e < fill;
        const int ec = valid ? e : head;
        const float4 p = q_xyh[ec];
        float4 co = q_co[ec];
        const float4 c = q_rgbp[ec];
        const uint32_t slot = q_slot[ec];
        co.w = valid ? co.w : 0.f;   // an empty slot: alpha = 0 on every pixel
        const uint32_t position = __float_as_uint(c.w);   // 0-based list position of the lane's splat
        v4f Dw = {0.f, 0.f, 0.f, 0.f}, Dg = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const float dy = p.y - (half ? pyfB : pyfA);
            const char *const prow = my + (4 * half + r) * PIX_ROW;
            uint32_t whi[4], wlo[4], ghi[4], glo[4];
#pragma unroll
            for (int k4 = 0; k4 < 8; k4 += 4) {
                float am[4], Gm[4], rinv[4], Pinc[4], w4[4], g4[4];
                float4 pc[4];
                float stT[4], stR[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const PixRow *row = reinterpret_cast<const PixRow *>(prow) + (k4 + u);
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
        // lane (r, s) holds rows 4 r .. 4 r + 3 of both products for splat s; the twelve useful ones meet in LDS (one 16-byte piece per
        // lane): C0 C1 C2 (A w) M0 (A g) | Mu Mv Muu Muv (A g) | Mvv (A g) C0' C1' C2' (A w) ...
        v4f D;
        D[0] = r == 0 ? Dw[0] : Dg[0];
        D[1] = r == 1 ? Dg[1] : Dw[1];
        D[2] = r == 1 ? Dg[2] : Dw[2];
        D[3] = r == 2 ? Dw[3] : Dg[3];
        if (r < 3) *reinterpret_cast<v4f *>(q_out + s * 12 + 4 * r) = D;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        // ... and every lane turns its splat's moments into its piece of the nine sums (render_bwd_mfma.hip's conversion):
        //   r = 0: dL/dcolour, dL/dmean2D.x   r = 1: dL/dmean2D.y, dL/dconic (A, B/2, C)   r = 2: dL/dopacity
        const v4f m0 = *reinterpret_cast<const v4f *>(q_out + s * 12), m1 = *reinterpret_cast<const v4f *>(q_out + s * 12 + 4),
                  m2 = *reinterpret_cast<const v4f *>(q_out + s * 12 + 8);
        const float M0 = m0[3], Mu = m1[0], Mv = m1[1], Muu = m1[2], Muv = m1[3], Mvv = m2[0];
        const float X = p.x - tcx, Y = p.y - tcy, kh = -0.5f * co.w;
        const float Sgx = kh * (X * M0 - Mu), Sgy = kh * (Y * M0 - Mv);   // -1/2 o sum g dx, dy  (dx = X - u)
        v4f o;
        if (r == 0) {
            o[0] = m0[0] + m2[1];
            o[1] = m0[1] + m2[2];
            o[2] = m0[2] + m2[3];
            o[3] = (Sgx * co.x + Sgy * co.y) * (float)W;        // dL/dmean2D in NDC units: 2 * (W / 2)
        } else {
            o[0] = r == 1 ? (Sgy * co.z + Sgx * co.y) * (float)H : M0;
            o[1] = kh * (X * X * M0 - 2.f * X * Mu + Muu);
            o[2] = kh * (X * Y * M0 - X * Mv - Y * Mu + Muv);
            o[3] = kh * (Y * Y * M0 - 2.f * Y * Mv + Mvv);
        }
        float *const dst = rows + ((size_t)slot * 4 + wave) * 12 + 4 * r;
        if (valid && r < 2) *reinterpret_cast<v4f *>(dst) = o;
        if (valid && r == 2) dst[0] = o[0];
        if (valid && r == 3) row_exists[(size_t)slot * 4 + wave] = 1;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");   // q_out is rewritten by the next batch
    };

    // ---- stream the list, deepest contributor first: 64 entries per chunk, lane = entry ----
    struct Idx { uint32_t g, slot; };
    struct Rec { float4 xyh, co, rgbd; };
    auto load_idx = [&](const uint32_t base) {
        Idx ix = {0u, 0u};
        const uint32_t t = base + lane;
        if (t < start) {
            const uint32_t pos = range.x + start - 1 - t;
            ix.g = min(point_list[pos], last_g);
            ix.slot = min(slot_list[pos], cap - 1u);
        }
        return ix;
    };
    auto load_rec = [&](const Idx ix, const uint32_t base) {
        Rec rc;
        rc.xyh = rc.co = rc.rgbd = make_float4(0.f, 0.f, 0.f, 0.f);
        if (base + lane < start) {
            rc.xyh = xyh[(size_t)ix.g * SPLAT_REC];
            rc.co = conic_opacity[(size_t)ix.g * SPLAT_REC];
            rc.rgbd = rgbd[(size_t)ix.g * SPLAT_REC];
        }
        return rc;
    };
    int fill = 0;
    Idx ix_cur = load_idx(0);
    Rec rec_next = load_rec(ix_cur, 0);
    Idx ix_next = load_idx(64);
    for (uint32_t base = 0; base < start; base += 64) {
        const Rec rc = rec_next;
        const uint32_t slot = ix_cur.slot;
        ix_cur = ix_next;
        if (base + 64 < start) rec_next = load_rec(ix_cur, base + 64);    // in flight while this chunk is processed
        if (base + 128 < start) ix_next = load_idx(base + 128);
        const uint32_t t = base + lane;
        const bool hit = t < start && quadrant_hit(rc.xyh, qcx, qcy) && ((ABL & 8) || rect_hit_tight(rc.xyh, rc.co, (float)qx0, (float)qy0));
        const uint64_t m = __ballot(hit);
        const int at = fill + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (hit) {
            q_xyh[at] = rc.xyh;
            q_co[at] = rc.co;
            q_rgbp[at] = make_float4(rc.rgbd.x, rc.rgbd.y, rc.rgbd.z, __uint_as_float(start - 1 - t));
            q_slot[at] = slot;
        }
        fill += __popcll(m);
        fill = __builtin_amdgcn_readfirstlane(fill);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        int head = 0;
        if (!(ABL & 4)) {
            for (; fill - head >= 16; head += 16) run_batch(head, fill);
        } else {
            head = fill & ~15;
        }
        if (head > 0) {   // the (< 16) entries left over move to the front of the queue
            const int rest = fill - head;
            float4 a0, a1, a2;
            uint32_t a3 = 0;
            a0 = a1 = a2 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (lane < rest) {
                a0 = q_xyh[head + lane];
                a1 = q_co[head + lane];
                a2 = q_rgbp[head + lane];
                a3 = q_slot[head + lane];
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            if (lane < rest) {
                q_xyh[lane] = a0;
                q_co[lane] = a1;
                q_rgbp[lane] = a2;
                q_slot[lane] = a3;
            }
            fill = rest;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        }
    }
    if (fill > 0 && !(ABL & 4)) run_batch(0, fill);
}

// bytes of scratch the render backward needs for `capacity` instances: [capacity][4 quadrants][12] floats + [capacity][4] bytes
size_t stream_scratch_bytes(int64_t capacity) {
    const size_t c = capacity > 0 ? (size_t)capacity : 1;
    return align_up(c * 4 * 12 * sizeof(float)) + align_up(c * 4);
}

int launch_render_backward_stream(const das3r_raster_args *a, const float *dL_dpix, char *geom, char *binning, char *img, const Layout &L,
                                  float *scratch, hipStream_t s) {
    const size_t c = L.capacity > 0 ? (size_t)L.capacity : 1;
    uint8_t *exists = reinterpret_cast<uint8_t *>(scratch) + align_up(c * 4 * 12 * sizeof(float));
    HIP_TRY(hipMemsetAsync(exists, 0, c * 4, s));
#define ARGS                                                                                                              \
    (const uint2 *)(img + L.pub.ranges), (const uint32_t *)(binning + L.pub.point_list), a->image_width, a->image_height, \
        L.tiles_x, pack_tiles(L), (const float4 *)(geom + L.pub.xy), (const float4 *)(geom + L.pub.conic_opacity),             \
        (const float4 *)(geom + L.pub.rgbd), a->bg, (const float *)(img + L.pub.final_T),                                 \
        (const uint32_t *)(img + L.pub.n_contrib), dL_dpix, (const uint32_t *)(binning + L.b_slot), scratch, exists,     \
        (uint32_t)(a->P - 1), (uint32_t)L.capacity
#define GO(AB) DAS3R_LAUNCH((render_backward_stream_kernel<AB>), dim3(xcd_grid(L)), dim3(TILE_PIX), 0, s, ARGS)
    const int abl = switches().ablate_set ? switches().ablate : 0;
    if (abl == 1) GO(1);
    else if (abl == 2) GO(2);
    else if (abl == 3) GO(3);
    else if (abl == 4) GO(4);
    else if (abl == 8) GO(8);
    else if (abl == 103) DAS3R_LAUNCH((render_backward_stream_kernel<0, 3>), dim3(xcd_grid(L)), dim3(TILE_PIX), 0, s, ARGS);
    else if (abl == 105) DAS3R_LAUNCH((render_backward_stream_kernel<0, 5>), dim3(xcd_grid(L)), dim3(TILE_PIX), 0, s, ARGS);
    else GO(0);
#undef GO
#undef ARGS
    KERNEL_CHECK(s, a->debug, "render_backward_stream");
    return DAS3R_OK;
}

}  // namespace das3r
