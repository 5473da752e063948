// render_common.h — per-(pixel, splat) arithmetic and wavefront helpers shared by the forward (K6) and backward (K7)
// compositing kernels, so both evaluate bit-identical alpha for the same pair.
//
// Work decomposition (MI355X-first): one 256-lane workgroup per 16x16 tile (the upstream tile, which also defines the
// hard 3-sigma gate), but each of its four wave64s owns one 8x8 pixel QUADRANT.  The tile's splat list is staged
// through LDS in batches of 256; every wave first culls the batch against its own quadrant with one conservative
// bounding-box test per lane (4 splats per lane) and a wavefront ballot, then walks only the surviving bits of the
// ballot masks (scalar loop, wave-uniform LDS addresses => broadcast reads).  A skipped splat provably has
// alpha < 1/255 on every pixel of the quadrant, so the image is bit-identical to walking the whole list.
#pragma once
#include "common.h"

namespace das3r {

struct StagedSplat {   // one LDS-staged entry of a tile's splat list (48 B)
    float4 xyh;        // pixel centre x, y; half extents hx, hy of the region where alpha can reach 1/255
    float4 co;         // conic A, B, C + opacity
    float4 rgbd;       // r, g, b, (depth)
};

// A staged float4 of which only x, y, z are used is fetched with ds_read_b96 — 8 LDS cycles per wave instruction against 4 for
// ds_read_b128 (MI355X_MICROARCH.md, LDS table).  Keeping the fourth component formally alive makes it the 128-bit read.
__device__ __forceinline__ float4 lds_read4(const float4 *p) {
    const float4 v = *p;
    asm volatile("" ::"v"(v.w));
    return v;
}

// The row walk of render_rows.hip reads a staged entry for EVERY position of its trips, also past the end of a row's list (the
// list byte there is stale, the arithmetic makes the pair inert): the entry it lands on must hold finite numbers — 0 x NaN is NaN —
// so the entries of a batch beyond the tile's list are filled with this one.
__device__ __forceinline__ StagedSplat null_splat() {
    StagedSplat z;
    z.xyh = make_float4(0.f, 0.f, -1e30f, -1e30f);
    z.co = make_float4(0.f, 0.f, 0.f, 0.f);
    z.rgbd = make_float4(0.f, 0.f, 0.f, 0.f);
    return z;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() fences every address space: on gfx950 (one vmcnt for loads and
// stores) that is s_waitcnt vmcnt(0) in front of s_barrier, i.e. global loads issued ahead of their use — a software prefetch —
// are waited for at the very next barrier.  Fences restricted to the LDS address space leave them in flight (s_waitcnt lgkmcnt(0)
// only).  Use it only where no global-memory hand-off between the waves of the workgroup depends on the barrier.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// XCD-aware block -> tile map.  The dispatcher places block b on XCD b % 8 (MI355X_MICROARCH.md).  The tiles are put in a LOCALITY
// ORDER — strips of 8 tile rows, column by column inside a strip, top to bottom inside a column: vertical neighbours are 1
// apart, horizontal neighbours 8 apart, against 1 / tiles_x in row-major order (where ~200 tiles in flight per XCD x 320 entries
// x 64 B already fill the L2 before the row below comes up) — so that the tiles a splat touches (2.6 on average at 1080p, mostly
// a 2 x 2 or 1 x 2 patch) are composited on the same XCD at about the same time and find the splat's record in that XCD's 4 MiB L2.
// Round 3: the order is dealt to the XCDs in CHUNKS of up to 64 tiles (an 8 x 8 tile square of the strip order), round robin,
// instead of one contiguous eighth each: a contiguous eighth of a 1080p frame is one band of the image, and an image whose
// density varies from top to bottom (sky over ground) leaves whole XCDs idle while others still work.  Fewer than 4096 tiles:
// smaller chunks, so that every XCD still gets at least eight of them.
// Packed (pack_tiles): bits 0-21 tiles, 22-24 chunk code (0 = contiguous eighths, else log2(chunk) + 1), 25-31 strip height
// (0: plain row-major order).
__device__ __forceinline__ int packed_ntiles(const int packed) { return packed & 0x3FFFFF; }
__device__ __forceinline__ int xcd_tile(const int block, const int packed, const int tiles_x) {
    const int ntiles = packed & 0x3FFFFF, code = (packed >> 22) & 7, SH = (int)((unsigned)packed >> 25);
    int k;
    if (code == 0) {
        const int per = (ntiles + 7) >> 3;
        k = (block & 7) * per + (block >> 3);
    } else {
        const int lg = code - 1, j = block >> 3;   // j-th block of XCD (block & 7): its (j >> lg)-th chunk, tile (j & mask) of it
        k = ((((j >> lg) << 3) + (block & 7)) << lg) + (j & ((1 << lg) - 1));
    }
    if (k >= ntiles) return -1;
    if (SH == 0) return k;
    const int tiles_y = ntiles / tiles_x;
    const int strip = k / (SH * tiles_x);
    const int rem = k - strip * SH * tiles_x;
    const int h = min(SH, tiles_y - SH * strip);
    const int bx = rem / h, by = SH * strip + (rem - bx * h);
    return by * tiles_x + bx;
}
// chunk code of a launch (host): DAS3R_TILE_CHUNK = 0 contiguous eighths, 1 .. 64 a fixed chunk, default: 64, halved until there
// are at least 64 chunks
static inline int tile_chunk_code(const int ntiles) {
    const int forced = switches().tile_chunk;
    if (forced == 0) return 0;
    int c = 64;
    if (forced > 0) c = forced;
    else while (c > 1 && ntiles / c < 64) c >>= 1;
    int lg = 0;
    while ((1 << lg) < c) lg++;
    return lg + 1;
}
// Strips pay when the splat records dominate a tile's traffic (measured at 1 M splats / 1080p, mean list 320: L2 misses of the
// forward 1.83 -> 1.51 x and of the backward 3.48 -> 2.97 x the algorithmic bytes, time unchanged); with short lists (100 k
// splats, mean list 32) the image rows dominate and vertically stacked tiles hit the same memory channels: 3 % slower — row-major.
// (built as an unsigned word: strip heights up to 63 keep the sign bit clear, and das3r_raster_forward refuses images of 2^22 tiles
//  or more — the tile count would run into the chunk code)
static inline int pack_tiles(const Layout &L) {
    const uint32_t sh = (L.capacity >= (int64_t)128 * L.ntiles) ? (uint32_t)std::min(switches().tile_strip, 63) : 0u;
    return (int)(((uint32_t)L.ntiles & 0x3FFFFFu) | ((uint32_t)tile_chunk_code(L.ntiles) << 22) | (sh << 25));
}
static inline int xcd_grid(const Layout &L) {
    const int code = tile_chunk_code(L.ntiles);
    if (code == 0) return ((L.ntiles + 7) / 8) * 8;
    const int ch = 1 << (code - 1), nchunks = (L.ntiles + ch - 1) / ch;
    return ((nchunks + 7) / 8) * 8 * ch;
}

// pixel owned by lane `lane` of wave `wave` inside tile (bx, by): wave -> 8x8 quadrant, lane -> pixel of the quadrant
__device__ __forceinline__ void quadrant_pixel(const int bx, const int by, const int wave, const int lane, int &px, int &py) {
    px = bx * TILE_X + ((wave & 1) << 3) + (lane & 7);
    py = by * TILE_Y + ((wave >> 1) << 3) + (lane >> 3);
}

// Gaussian falloff of splat at pixel (px,py).  Returns false when the pair is skipped
// (power > 0 or alpha < 1/255).  Explicit fma placement => same rounding in every kernel that inlines this.
__device__ __forceinline__ bool pair_alpha(const float x, const float y, const float4 co, const float px, const float py, float &dx,
                                           float &dy, float &G, float &alpha) {
    dx = x - px;
    dy = y - py;
    const float q = __fmaf_rn(__fmul_rn(co.x, dx), dx, __fmul_rn(__fmul_rn(co.z, dy), dy));
    const float power = __fmaf_rn(-0.5f, q, -__fmul_rn(__fmul_rn(co.y, dx), dy));
    G = __expf(power);  // evaluated unconditionally: the kernels' inner loops are branch-free (selects only)
    alpha = fminf(0.99f, __fmul_rn(co.w, G));
    return (!(power > 0.0f)) & (alpha >= (1.0f / 255.0f));
}

// ---- the forward kernels' per-pair decisions by arithmetic ---------------------------------------------------------------------
// On this chip v_cmp and v_cndmask issue at half the rate of a plain VALU instruction (1.76 vs 0.96 ns per wave instruction on a
// saturated SIMD: profiles/r03_valu_rate_probe.txt); a compare + select pair costs as much as four multiplies.  The helpers
// below take the same decisions as the selects they replace, bit for bit, with fma / min / med3 and the clamp output modifier.

// min(a, b) clamped to [0, 1] in ONE instruction (output modifier); a NaN comes out as 0 (DX10_CLAMP is on in compute kernels)
__device__ __forceinline__ float min_clamp01(const float a, const float b) {
    float r;
    asm("v_min_f32_e64 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// clamp(a b + c, 0, 1) in one instruction
__device__ __forceinline__ float fma_clamp01(const float a, const float b, const float c) {
    float r;
    asm("v_fma_f32 %0, %1, %2, %3 clamp" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// alpha of a pair whose clamped opacity a1 = min(0.99, o G[, ...]) and exponent `power` are known: a1 where power <= 0 and
// a1 >= 1/255 (pair_alpha's two tests), else 0.  m = min(a1 - 1/255, -power) >= 0 <=> both pass (a1 - 1/255 is exact near the
// threshold: Sterbenz); min(a1, a1 + 1e30 m) is a1 for m >= 0 and hugely negative for m < 0 (|m| >= 1e-21 whenever it is
// negative: the smallest positive power fp32 pixel offsets can produce, and one ulp of 1/255 on the other side).  a1 may be
// negative (render_rows.hip: no such list position) — the clamp takes care of it.
__device__ __forceinline__ float alpha_if_visible(const float a1, const float power) {
    const float m = fminf(a1 - (1.0f / 255.0f), -power);
    return min_clamp01(a1, __fmaf_rn(1e30f, m, a1));
}
// Per-pixel compositing state of the forward kernels.  live: 1 until the pixel stops (T would drop below 1e-4), then 0.
// lastf: staged index of the pixel's last contributor in the CURRENT batch, -1 = none yet (batch_begin / batch_end).
struct PixelBlend {
    float T, C0, C1, C2, live, lastf;
};
// Blend staged entry j (jf = (float)j, colour c, alpha `a` from alpha_if_visible) into the pixel.  The reference's loop:
//     test_T = T (1 - alpha);  if (test_T < 1e-4) { done; } else { C += c alpha T; T = test_T; last = position; }
//   * s = 1 where test_T >= 1e-4f, else 0: clamp(test_T 2^100 - c' 2^100) with c' the float below 1e-4f — the product and the
//     constant are exact (power-of-two scaling), so the sign is; one ulp of 1e-4 scaled by 2^100 is 2^63, far beyond the clamp;
//   * w = alpha s;  T (1 - w) is test_T where the pair is taken and T where it is not (alpha = 0 or stop);
//   * contributing entries are visited in staged order, so `last` is a running maximum: max(last, min(j, 1e30 w - 1)) is j where
//     w > 0 (w >= 1/255 there) and last where w = 0 (last >= -1).  (NOT med3(last, j, 1e30 w - 1): render_rows.hip reads a stale
//     list byte past the end of a row's list — an inert pair, but its j may be below `last`, and the median would take it.)
// T never drops below 1e-4 while a lane is live, and a stopped lane has alpha = 0: s = 1 there, nothing changes.
__device__ __forceinline__ void blend_pair(PixelBlend &px, const float alpha, const float4 c, const float jf) {
    const float a = alpha * px.live;
    const float test_T = px.T * (1.0f - a);
    const float s = fma_clamp01(test_T, 0x1p100f, -__uint_as_float(0x6AD1B716u) /* (float below 1e-4f = 0x38D1B716) x 2^100 */);
    const float w = a * s;
    const float wT = w * px.T;
    px.C0 = __fmaf_rn(c.x, wT, px.C0);
    px.C1 = __fmaf_rn(c.y, wT, px.C1);
    px.C2 = __fmaf_rn(c.z, wT, px.C2);
    px.T = px.T * (1.0f - w);
    px.live *= s;
    px.lastf = fmaxf(px.lastf, fminf(jf, __fmaf_rn(w, 1e30f, -1.0f)));
}
__device__ __forceinline__ void blend_batch_begin(PixelBlend &px) { px.lastf = -1.0f; }
// -> the pixel's last contributor as a 1-based list position, given the batch's first list position
__device__ __forceinline__ uint32_t blend_batch_end(const PixelBlend &px, const uint32_t last_contributor, const uint32_t batch_first) {
    return px.lastf >= 0.0f ? batch_first + (uint32_t)px.lastf + 1u : last_contributor;
}

// A failed binning (look-back timeout: reported through the self-check word, api.hip) may leave garbage in the tile lists.
// Every index read from them is kept in bounds, so that the failure surfaces as the error it is and not as a memory fault.
__device__ __forceinline__ uint2 safe_range(uint2 r, const uint32_t cap) {
    r.y = min(r.y, cap);
    r.x = min(r.x, r.y);
    return r;
}

// Conservative quadrant test: can the splat reach alpha >= 1/255 on any pixel centre of the 8x8 block centred at
// (cx, cy)?  (hx, hy) already carry their safety margin (preprocess.hip).
__device__ __forceinline__ bool quadrant_hit(const float4 xyh, const float cx, const float cy) {
    return fabsf(xyh.x - cx) <= xyh.z + 3.5f && fabsf(xyh.y - cy) <= xyh.w + 3.5f;
}

// ---- octagon test against a 4x4 block (forward: render_rows.hip, backward: render_bwd_blk.hip) --------------------------------
// The axis-aligned box (hx, hy) passes every block the splat's bounding box overlaps — for elongated, rotated splats many
// more than the ellipse reaches (tools/pair_stats_cpu.py, 1 M splats at 1080p: 5.87 block hits per instance against 4.74 exact).
// Two more slabs, along (1, 1) / sqrt 2 and (1, -1) / sqrt 2, make it an octagon: 4.92.  Their half extents come from the
// axis-aligned ones and the conic WITHOUT the determinant (which cancels in fp32 for long thin splats): with tau Sxx = ex^2,
// tau Syy = ey^2 (preprocess.hip: hx = sqrt(tau Sxx) 1.0005 + 0.02) and Sxy = -B Sxx / C,
//     hd1^2 = tau (Sxx + Syy + 2 Sxy) / 2,   hd2^2 = tau (Sxx + Syy - 2 Sxy) / 2,
// plus slack for the cancellation in the sum (2e-6 relative to the large terms) and the margins the axis-aligned extents carry.
// Conservative: outside any slab alpha < 1 / 255 on every pixel centre of the block (tests/test_blk_model.py).
__device__ __forceinline__ void diagonal_extents(const float4 xyh, const float4 co, float &hd1, float &hd2) {
    const float ex = (xyh.z - 0.02f) * (1.0f / 1.0005f), ey = (xyh.w - 0.02f) * (1.0f / 1.0005f);
    const float ex2 = ex * ex, ey2 = ey * ey, txy = -co.y * ex2 * __builtin_amdgcn_rcpf(co.z);
    const float half = 0.5f * (ex2 + ey2), slack = 2e-6f * (ex2 + ey2) + 1e-3f;
    hd1 = __builtin_amdgcn_sqrtf(fmaxf(half + txy, 0.f) + slack) * 1.0005f + 0.05f;   // (v_sqrt_f32: 1 ulp, inside the margins)
    hd2 = __builtin_amdgcn_sqrtf(fmaxf(half - txy, 0.f) + slack) * 1.0005f + 0.05f;
}
// block of pixel centres [cx - 1.5, cx + 1.5] x [cy - 1.5, cy + 1.5]
__device__ __forceinline__ bool block_hit_oct(const float x, const float y, const float hx, const float hy, const float hd1, const float hd2,
                                              const float cx, const float cy) {
    const float ddx = x - cx, ddy = y - cy;
    constexpr float RS2 = 0.70710678f, HALF_DIAG = 2.1213204f + 1e-3f;   // (1.5 + 1.5) / sqrt 2
    return (fabsf(ddx) <= hx + 1.5f) & (fabsf(ddy) <= hy + 1.5f) & (fabsf(ddx + ddy) * RS2 <= hd1 + HALF_DIAG) &
           (fabsf(ddx - ddy) * RS2 <= hd2 + HALF_DIAG);
}

// ---- 2x2 regions of an 8x8 quadrant (forward: render_regions.hip, backward: render_bwd_rgn.hip) ----------------------------------
// Which of a quadrant's sixteen 2x2 regions (bit 4 ry + rx; region centres qcx0 + 2 rx, qcy0 + 2 ry) can the entry reach?  The test of every
// forward kernel — |centre distance| <= cull half extent + half the region — evaluated ONCE per entry by the thread that stages it
// (eight compares) instead of once per entry and wave: a wave's cull is then a bit test.  Entries past the list carry extents nobody meets.
__device__ __forceinline__ uint32_t region_mask(const float4 p, const float qcx0, const float qcy0) {
    uint32_t xb = 0, yb = 0;
    const float hx = p.z + 0.5f, hy = p.w + 0.5f;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        xb |= (fabsf(p.x - (qcx0 + (float)(2 * k))) <= hx ? 1u : 0u) << k;
        yb |= (fabsf(p.y - (qcy0 + (float)(2 * k))) <= hy ? 1u : 0u) << k;
    }
    return ((yb & 1u) ? xb : 0u) | ((yb & 2u) ? xb << 4 : 0u) | ((yb & 4u) ? xb << 8 : 0u) | ((yb & 8u) ? xb << 12 : 0u);
}

// ---- checkpoints of long tile lists (common.h: BUCKET) -----------------------------------------------------------------------
// A tile whose list has len > BUCKET entries owns nb = ceil(len / BUCKET) slots of 256 float4 (one per pixel, row-major inside the
// tile), the first at block  range.x / BUCKET + tile  (distinct tiles never overlap: floor(a + b) >= floor(a) + floor(b)).
// Slot s < nb - 1 holds the pixel's (T, C0, C1, C2) in front of list position (s + 1) * BUCKET; slot nb - 1 its final values
// (C without the background).  Shorter lists write nothing.
__device__ __forceinline__ float4 *ckpt_slot(float4 *ckpt, const uint2 range, const int tile, const int s) {
    return ckpt + ((size_t)(range.x / BUCKET) + (size_t)tile + (size_t)s) * TILE_PIX;
}
__device__ __forceinline__ int ckpt_buckets(const uint2 range) { return (int)((range.y - range.x + BUCKET - 1) / BUCKET); }

// ---- single-wave bitonic network in registers -----------------------------------------------------------------------------
// Sorts N = 64 << G 64-bit words in LDS ascending, executed by ONE wave: every lane holds E = 1 << G words per pass in registers
// and takes G levels of the always-ascending bitonic network (first level of every merge mirrors the block, the rest are plain
// butterflies) before the words go back — 16 passes for 512 words (15 for 1024) against the 45 (55) LDS round trips of a
// comparison per thread and level, and no workgroup barrier (LDS is in order per wave).  Round 2's network kept the CU's LDS
// busy ~60 us of the 1 M-splat forward: 2 reads + 2 writes per comparison and level, 32 tiles per CU.
// The words are (depth bits << 32 | position) of positive finite floats, i.e. read as DOUBLES they are positive, finite and in the
// same order: a compare-exchange is v_min_f64 + v_max_f64 instead of a 64-bit compare + four selects.  Padding: the largest finite
// double.  CPU model: tests/test_sort_network.py.
__device__ __forceinline__ void ce64(double &lo, double &hi) {
    // (as builtins the compiler first canonicalises both operands — v_max_f64 x, x, x — for signalling NaNs that cannot occur here)
    double a, b;
    asm("v_min_f64 %0, %2, %3\n\tv_max_f64 %1, %2, %3" : "=&v"(a), "=&v"(b) : "v"(lo), "v"(hi));
    lo = a;
    hi = b;
}
constexpr unsigned long long SORT_PAD = 0x7FEFFFFFFFFFFFFFull;   // DBL_MAX: above every (depth bits << 32 | position) word

// levels R - 1 .. R - g of the current merge: plain butterflies on 2^g words at stride 2^(R - g); E >> g such groups per lane
template <int g, int E>
__device__ __forceinline__ void butterfly_group(double *key, const int lane, const int R) {
    const int s = R - g;
#pragma unroll
    for (int c = 0; c < (E >> g); c++) {
        const int gid = c * 64 + lane;
        const int base = ((gid >> s) << R) | (gid & ((1 << s) - 1));
        double e[1 << g];
#pragma unroll
        for (int m = 0; m < (1 << g); m++) e[m] = key[base + (m << s)];
#pragma unroll
        for (int t = g - 1; t >= 0; t--)
#pragma unroll
            for (int m = 0; m < (1 << g); m++)
                if (!((m >> t) & 1)) ce64(e[m], e[m | (1 << t)]);
#pragma unroll
        for (int m = 0; m < (1 << g); m++) key[base + (m << s)] = e[m];
    }
}

template <int G>
__device__ __forceinline__ void wave_bitonic_sort(double *key, const int lane) {
    constexpr int E = 1 << G, H = E / 2, LOGN = 6 + G;
    {   // pass 0: merges 1 .. G — every lane sorts its E contiguous words
        double v[E];
#pragma unroll
        for (int m = 0; m < E; m++) v[m] = key[lane * E + m];
#pragma unroll
        for (int lk = 1; lk <= G; lk++) {
#pragma unroll
            for (int i = 0; i < E; i++)
                if (!(i & (1 << (lk - 1)))) ce64(v[i], v[i ^ ((1 << lk) - 1)]);   // mirror inside the 2^lk block
#pragma unroll
            for (int lj = lk - 2; lj >= 0; lj--)
#pragma unroll
                for (int i = 0; i < E; i++)
                    if (!((i >> lj) & 1)) ce64(v[i], v[i | (1 << lj)]);
        }
#pragma unroll
        for (int m = 0; m < E; m++) key[lane * E + m] = v[m];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    for (int lk = G + 1; lk <= LOGN; lk++) {
        {   // mirror level of merge lk + its butterflies on bits lk - 2 .. lk - G: H words of the lower half and their mirror images
            const int sh = lk - G;
            const int block = lane >> sh, hbase = lane & ((1 << sh) - 1);
            const int lo0 = (block << lk) + hbase, hi0 = (block << lk) + ((1 << lk) - 1) - hbase;
            double a[H], b[H];
#pragma unroll
            for (int m = 0; m < H; m++) {
                a[m] = key[lo0 + (m << sh)];
                b[m] = key[hi0 - (m << sh)];
            }
#pragma unroll
            for (int m = 0; m < H; m++) ce64(a[m], b[m]);
#pragma unroll
            for (int t = G - 2; t >= 0; t--)
#pragma unroll
                for (int m = 0; m < H; m++)
                    if (!((m >> t) & 1)) {
                        ce64(a[m], a[m | (1 << t)]);
                        ce64(b[m | (1 << t)], b[m]);   // (mirrored words: the larger m is the lower address)
                    }
#pragma unroll
            for (int m = 0; m < H; m++) {
                key[lo0 + (m << sh)] = a[m];
                key[hi0 - (m << sh)] = b[m];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        for (int R = lk - G; R > 0;) {   // the remaining butterflies, G levels at a time (the last group may be smaller)
            const int g = R < G ? R : G;
            if (g == 1) butterfly_group<1, E>(key, lane, R);
            else if (g == 2) butterfly_group<2, E>(key, lane, R);
            else if (g == 3) butterfly_group<3, E>(key, lane, R);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            R -= g;
        }
    }
}

// ---- local binning front-end (local_bin.hip) -------------------------------------------------------------------------------
// Local depth order (common.h: LocalBin): the tile's list arrives in index order; put it in the exact (depth bits, index)
// order the global radix path produces, in place (point_list / slot_list are read by the backward pass and the tests).
//   n <= 256   rank sort fused with the staging of the first (only) batch: every thread loads its entry's record, counts the
//              keys below its own (n broadcast LDS reads, no barriers inside) and stores the record at that rank of `stage` —
//              one extra barrier and no extra trip to memory compared with reading a sorted list.  Returns true.
//   n <= 1024  bitonic network on 64-bit words (depth bits << 32 | position in the list: positions follow the index order,
//              so this is the (depth, index) order) in its always-ascending form (first step of every merge mirrors the
//              block, the rest are plain butterflies), padded to 512 or 1024 words and run by ONE wave with 8 / 16 words per
//              lane in registers (wave_bitonic_sort).  The sorted indices stay in s_gid for the staging loop.  The words
//              borrow the staging area.
//   longer     the same network over global memory (slow; the host is told and goes back to the global sort for the next
//              forwards); s_gid is not filled.
struct NoMark {
    __device__ __forceinline__ void operator()(int) const {}
};
// mark(k): phase clocks of the experiments build (common.h PHASE_MARK; tools/phase_clocks.py), a no-op otherwise
template <class Mark = NoMark>
__device__ __forceinline__ bool local_order_tile(const LocalBin &lb, const uint2 range, const float4 *__restrict__ xyh,
                                                 const float4 *__restrict__ conic_opacity, const float4 *__restrict__ rgbd,
                                                 StagedSplat *stage /*[TILE_PIX]*/, uint32_t *s_gid /*[LOCAL_MAX]*/, const int tid,
                                                 const Mark mark = Mark()) {
    const int n = (int)(range.y - range.x);
    uint32_t *pl = lb.point_list + range.x, *sl = lb.slot_list + range.x;
    if (n <= TILE_PIX) {   // (uniform)
        unsigned long long *s_key = reinterpret_cast<unsigned long long *>(s_gid);
        unsigned long long key = 0;
        uint32_t g = 0, slot = 0;
        float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, r2 = r0;   // (initialised: left undefined, the compiler parks them in LDS — +13 us)
        if (tid < n) {
            g = min(pl[tid], lb.last_g);
            slot = sl[tid];
            r0 = xyh[(size_t)g * SPLAT_REC];
            r1 = conic_opacity[(size_t)g * SPLAT_REC];
            r2 = rgbd[(size_t)g * SPLAT_REC];
            key = ((unsigned long long)__float_as_uint(r2.w) << 32) | (unsigned long long)g;
        }
        if (tid < ((n + 7) & ~7)) s_key[tid] = tid < n ? key : ~0ull;   // padded to a multiple of 8 with keys that count for nobody
        __syncthreads();
        if (tid < n) {
            int rank = 0;
            for (int j = 0; j < n; j += 8) {   // 8 independent broadcast reads in flight
                unsigned long long o[8];
#pragma unroll
                for (int u = 0; u < 8; u++) o[u] = s_key[j + u];
#pragma unroll
                for (int u = 0; u < 8; u++) rank += o[u] < key ? 1 : 0;
            }
            stage[rank].xyh = r0;
            stage[rank].co = r1;
            stage[rank].rgbd = r2;
            pl[rank] = g;
            sl[rank] = slot;
        } else {
            stage[tid] = null_splat();   // entries n .. 255: initialised (see null_splat)
        }
        return true;   // the compositing loop's barrier publishes the batch
    }
    if (n <= LOCAL_MAX) {
        unsigned long long *s_key = reinterpret_cast<unsigned long long *>(stage);
        uint32_t *s_slot = reinterpret_cast<uint32_t *>(stage) + 2 * LOCAL_MAX;
        const int N = n <= 2 * TILE_PIX ? 2 * TILE_PIX : LOCAL_MAX;   // (uniform) 512 or 1024 words, padded above n
        for (int i = tid; i < N; i += TILE_PIX) {
            unsigned long long k = SORT_PAD;
            if (i < n) {
                const uint32_t g = min(pl[i], lb.last_g);
                s_gid[i] = g;
                s_slot[i] = sl[i];
                k = ((unsigned long long)__float_as_uint(rgbd[(size_t)g * SPLAT_REC].w) << 32) | (unsigned long long)i;
            }
            s_key[i] = k;
        }
        __syncthreads();
        mark(6);   // list words + depth keys in LDS
        if (N == 2 * TILE_PIX) {
            // up to 512 words (the 1 M-splat 1080p benchmark: mean list 320): one wave sorts them with eight words per lane in
            // registers (wave_bitonic_sort); the other three wait at the barrier below
            if (tid < 64) wave_bitonic_sort<3>(reinterpret_cast<double *>(stage), tid);
        } else {
            // 513 .. 1024 words: round 2's network, one thread per comparison and level (sixteen words per lane in registers would
            // cost the compositing loop behind it two waves of occupancy; these lists are rare where the local order is chosen).
            // Comparison pr touches words of the 128-word chunk pr / 64 only as long as the partner distance stays below 128, and
            // the comparisons 64 w .. 64 w + 63 (+ 256) belong to wave w: those steps need no workgroup barrier.
            bool wide_before = false;   // (the barrier behind the loads above covers the first step)
            for (int lk = 1; (1 << (lk - 1)) < n; lk++)
                for (int lj = lk - 1; lj >= 0; lj--) {
                    const bool wide = lj > 6;   // words of another wave's chunk
                    if (wide || wide_before) __syncthreads();
                    else __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    wide_before = wide;
                    for (int pr = tid; pr < (N >> 1); pr += TILE_PIX) {
                        const int i = ((pr >> lj) << (lj + 1)) | (pr & ((1 << lj) - 1));   // bit lj of i is clear
                        const int q = (lj == lk - 1) ? (i ^ ((2 << lj) - 1)) : (i | (1 << lj));   // mirror in the block / butterfly
                        if (q < n) {
                            const unsigned long long a = s_key[i], b = s_key[q];
                            if (a > b) {
                                s_key[i] = b;
                                s_key[q] = a;
                            }
                        }
                    }
                }
        }
        __syncthreads();
        mark(7);   // the network
        uint32_t g[LOCAL_MAX / TILE_PIX], slot[LOCAL_MAX / TILE_PIX];
#pragma unroll
        for (int u = 0; u < LOCAL_MAX / TILE_PIX; u++) {
            const int i = u * TILE_PIX + tid;
            const int from = i < n ? (int)((uint32_t)s_key[i] & (uint32_t)(LOCAL_MAX - 1)) : 0;
            g[u] = s_gid[from];
            slot[u] = s_slot[from];
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < LOCAL_MAX / TILE_PIX; u++) {
            const int i = u * TILE_PIX + tid;
            if (i < n) {
                s_gid[i] = g[u];
                pl[i] = g[u];
                sl[i] = slot[u];
            }
        }
        __syncthreads();
        return false;
    }
    uint32_t *dk = lb.keys + range.x;
    if (tid == 0) __hip_atomic_store(lb.host_flag, lb.flag_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    for (int i = tid; i < n; i += TILE_PIX) dk[i] = __float_as_uint(rgbd[(size_t)min(pl[i], lb.last_g) * SPLAT_REC].w);
    __syncthreads();
    for (int k = 2; (k >> 1) < n; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            const int flip = (j == (k >> 1)) ? k - 1 : j;
            for (int i = tid; i < n; i += TILE_PIX) {
                const int q = i ^ flip;
                if (q > i && q < n) {
                    const uint32_t da = dk[i], db = dk[q], ga = pl[i], gb = pl[q];
                    if (da > db || (da == db && ga > gb)) {
                        dk[i] = db; dk[q] = da;
                        pl[i] = gb; pl[q] = ga;
                        const uint32_t t = sl[i];
                        sl[i] = sl[q];
                        sl[q] = t;
                    }
                }
            }
            __syncthreads();   // same workgroup, same CU: its write-through L1 keeps the exchanged words coherent
        }
    return false;
}

// ---- transposed wavefront reduction -------------------------------------------------------------------------------
// Sums 8 values across the 64 lanes in 18 cross-lane ops instead of 8 x 6: every exchange step also halves the number
// of live values per lane (gfx950 v_permlane32_swap / v_permlane16_swap, then DPP inside the 16-lane rows).
// On return every lane l holds the wave-wide total of value number (l >> 3).
template <int CTRL>
__device__ __forceinline__ float dpp_add_full(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float swap32_add(const float a, const float b) {  // lanes 0-31: sum(a) halves, lanes 32-63: sum(b) halves
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float swap16_add(const float a, const float b) {  // rows: [a0+a1, b0+b1, a2+a3, b2+b3]
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float wave_reduce8_transposed(const float v[8], const int lane) {
    const float r0 = swap32_add(v[0], v[4]), r1 = swap32_add(v[1], v[5]);
    const float r2 = swap32_add(v[2], v[6]), r3 = swap32_add(v[3], v[7]);
    const float s02 = swap16_add(r0, r2);  // rows: v0, v2, v4, v6
    const float s13 = swap16_add(r1, r3);  // rows: v1, v3, v5, v7
    const float t0 = dpp_add_full<0x128>(s02);  // row_ror:8  -> lane l: x[l] + x[l^8]
    const float t1 = dpp_add_full<0x128>(s13);
    float m = (lane & 8) ? t1 : t0;             // row r: lanes 0-7 value 2r, lanes 8-15 value 2r+1  => value index l>>3
    m = dpp_add_full<0x141>(m);                 // row_half_mirror
    m = dpp_add_full<0x1B>(m);                  // quad_perm [3,2,1,0]
    m = dpp_add_full<0xB1>(m);                  // quad_perm [1,0,3,2]
    return m;
}

// ---- backward replay of one (pixel, splat) pair, branch-free --------------------------------------------------------
// Per-pixel recurrence carried back-to-front.  Upstream keeps the colour behind the splat per channel
// (accum_rec, last_color, last_alpha: 7 floats).  Only its dot product with dL/dpixel is ever used, so the state here is
// two scalars: T (transmittance in front of the current splat) and R = sum over the splats behind of
// (c_j . dL/dpix) alpha_j T_j.  With cd = c_i . dL/dpix:
//     dL/dalpha_i = T_i * sum_ch (c_i - accum_i) dL/dpix  - T_final/(1-alpha_i) * (bg . dL/dpix)
//                 = T_i * cd - (R + T_final * (bg . dL/dpix)) / (1 - alpha_i)          (accum_i * T_i = R / (1 - alpha_i))
struct ReplayState {
    float T, R;
};
// Inactive lanes run the same arithmetic with alpha = G = 0: state and sums are unchanged.  v[0..2] dL/dcolour,
// v[3..4] dL/dmean2D (already scaled by W/2, H/2), v[5..7] dL/dconic (B entry = half the true derivative, upstream
// convention), v[8] dL/dopacity.  tfbg = T_final * (bg . dL/dpix).
__device__ __forceinline__ void replay_pair(const bool active, const float alpha, const float G, const float dx, const float dy,
                                            const float4 co, const float4 c, const float dLp0, const float dLp1, const float dLp2,
                                            const float tfbg, const float ddelx_dx, const float ddely_dy, ReplayState &st, float v[9]) {
    const float am = active ? alpha : 0.f, Gm = active ? G : 0.f;
    const float rinv = __builtin_amdgcn_rcpf(1.f - am);  // v_rcp_f32: T is itself a reconstruction, 1 ulp is noise
    st.T = st.T * rinv;
    const float w = am * st.T;                            // alpha_i * T_i = d(pixel colour)/d(c_i)
    const float cd = c.x * dLp0 + c.y * dLp1 + c.z * dLp2;
    const float dL_dalpha = st.T * cd - (st.R + tfbg) * rinv;
    st.R = st.R + cd * w;
    const float gdl = Gm * dL_dalpha;   // G * dL/dalpha (0 on inactive lanes)
    const float hd = -0.5f * co.w * gdl;  // -1/2 * G * dL/dG
    const float gx = hd * dx, gy = hd * dy;
    v[0] = w * dLp0;
    v[1] = w * dLp1;
    v[2] = w * dLp2;
    v[3] = (gx * co.x + gy * co.y) * (2.0f * ddelx_dx);
    v[4] = (gy * co.z + gx * co.y) * (2.0f * ddely_dy);
    v[5] = gx * dx;
    v[6] = gx * dy;
    v[7] = gy * dy;
    v[8] = gdl;
}

// The same replay with the six geometry sums left as raw MOMENTS of g = G * dL/dalpha about the splat centre,
//     v[3] = g dx, v[4] = g dy, v[5] = g dx^2, v[6] = g dx dy, v[7] = g dy^2, v[8] = g,
// 6 multiplies per pair instead of 14: the factors that do not depend on the pixel (conic, opacity, W/2, H/2) are applied once
// per (tile, splat) after the reduction (moments_to_sums) — every sum is linear in them.
__device__ __forceinline__ void replay_pair_moments(const bool active, const float alpha, const float G, const float dx, const float dy,
                                                    const float4 c, const float dLp0, const float dLp1, const float dLp2, const float tfbg,
                                                    ReplayState &st, float v[9]) {
    const float am = active ? alpha : 0.f, Gm = active ? G : 0.f;
    const float rinv = __builtin_amdgcn_rcpf(1.f - am);
    st.T = st.T * rinv;
    const float w = am * st.T;
    const float cd = c.x * dLp0 + c.y * dLp1 + c.z * dLp2;
    const float dL_dalpha = st.T * cd - (st.R + tfbg) * rinv;
    st.R = st.R + cd * w;
    const float g = Gm * dL_dalpha;
    const float gx = g * dx, gy = g * dy;
    v[0] = w * dLp0;
    v[1] = w * dLp1;
    v[2] = w * dLp2;
    v[3] = gx;
    v[4] = gy;
    v[5] = gx * dx;
    v[6] = gx * dy;
    v[7] = gy * dy;
    v[8] = g;
}
// in place: (colour sums, moments) of one (tile, splat) -> the nine sums of replay_pair
__device__ __forceinline__ void moments_to_sums(float *a /*[9]*/, const float4 co, const float ddelx_dx, const float ddely_dy) {
    const float hd = -0.5f * co.w;
    const float m1 = a[3], m2 = a[4];
    a[3] = hd * (co.x * m1 + co.y * m2) * (2.0f * ddelx_dx);
    a[4] = hd * (co.z * m2 + co.y * m1) * (2.0f * ddely_dy);
    a[5] = hd * a[5];
    a[6] = hd * a[6];
    a[7] = hd * a[7];
}

}  // namespace das3r
