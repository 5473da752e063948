// render_rows.hip — row-private variant of the forward compositing kernel (K6): every 16-lane DPP row of a wave64 owns a
// 4x4 pixel block and walks ITS OWN culled sub-list of the tile's splats.
//
// Why: with one sub-list per 8x8 quadrant (render_fwd.hip) a wave spends one iteration per (splat, quadrant) whatever the
// footprint; measured lane utilisation (tools/pair_stats.py) is 36 % on the 1080p benchmarks (23 of 64 pixels see
// alpha >= 1/255) and 10 % on DAS3R-shaped scenes (one tiny Gaussian per pixel: 6.5 of 64).  Four independent 4x4 blocks per
// wave cut the iteration count to 0.73-0.78x (1080p) and 0.47x (DAS3R shape): the four rows of a wave process four
// DIFFERENT splats in one pass of the instruction stream.  Measured: forward 1.12 -> 0.84 ms on the 5 M-splat DAS3R-shaped
// scene, 0.294 -> 0.273 ms at 1 M splats / 1080p, 0.045 -> 0.050 ms at 100 k (list building dominates short lists, hence the
// switch in use_row_private()).
//
// Per batch of 256 LDS-staged splats every wave builds four compacted index lists (one per row: ballot + mbcnt prefix, u8
// indices in LDS), then iterates i = 0 .. longest list; a row whose list is exhausted idles (predicated off).  The per-pixel
// arithmetic is the shared pair_alpha of render_common.h and every pixel still sees its splats in list order, so the image,
// n_contrib and final_T are bit-identical to the quadrant kernel.
//
// Round 2: four list positions per trip (independent alpha chains interleave), no per-position wave-uniform skip, padded list
// rows (bank conflicts), and — for scenes with fewer tiles than workgroup slots — a software prefetch behind an LDS-only barrier
// with two staging areas (one barrier per batch): 0.313 -> 0.272 ms at 1 M splats / 1080p, 0.78 -> 0.45 ms on the DAS3R shape.
//
// The same decomposition was tried twice for the BACKWARD kernel and lost both times at 1080p.  (1) Rows add their nine sums
// into the tile's LDS accumulators: 0.91 vs 0.605 ms at 1 M splats — LDS float atomics cost ~4 clocks per active lane whether
// or not the addresses collide (36 lanes per iteration instead of 9: 0.45 ms of the 0.91; the row-private arithmetic alone
// took 0.37 ms).  (2) Rows STORE their sums into private slots racc[16 rows][64 splats][9] (a row meets a splat once per
// batch) and the workgroup adds the listed slots per splat afterwards: no atomics, but 36 KB of slots force 64-splat batches
// and 3 workgroups per CU; 0.92 ms at 1 M splats, 1.76 vs 1.96 ms on the 5 M-splat DAS3R-shaped scene — not worth a second
// code path.
#include "render_common.h"

namespace das3r {

// pixel owned by `lane` of `wave`: wave -> 8x8 quadrant, row (lane >> 4) -> 4x4 block of the quadrant, lane & 15 -> pixel
__device__ __forceinline__ void block_pixel(const int bx, const int by, const int wave, const int lane, int &px, int &py) {
    const int row = lane >> 4;
    px = bx * TILE_X + ((wave & 1) << 3) + ((row & 1) << 2) + (lane & 3);
    py = by * TILE_Y + ((wave >> 1) << 3) + ((row >> 1) << 2) + ((lane >> 2) & 3);
}

// can the splat reach alpha >= 1/255 on a pixel centre of the 4x4 block centred at (cx, cy)?  (see quadrant_hit)
__device__ __forceinline__ bool block_hit(const float4 xyh, const float cx, const float cy) {
    return fabsf(xyh.x - cx) <= xyh.z + 1.5f && fabsf(xyh.y - cy) <= xyh.w + 1.5f;
}

// A row's list: 256 one-byte positions + one pad word.  The four rows of a wave read position t of THEIR lists in the same
// instruction: at a stride of 256 B those four bytes sit in one bank (r02 PMC: 1.8e7 conflict cycles in 2.8e7 LDS cycles at
// 1 M splats); 260 B puts them in four neighbouring banks.
constexpr int ROW_LIST_STRIDE = TILE_PIX;   // (round 3 padded the rows by 4 bytes against bank conflicts; round 4 trades that for the eighth workgroup per CU)

// Build the four per-row index lists of this wave for a staged batch of n splats.  lists: this wave's [4][256] bytes.
// Returns the four lengths (wave-uniform).  Round 3: octagon test per block (render_common.h) instead of the bounding box —
// 4.9 instead of 5.9 listed blocks per instance at 1 M splats / 1080p; a listed pair that the octagon drops has alpha < 1/255
// on every pixel of the block, so image, n_contrib and final_T do not change.
// OCT = false: bounding box only — the DAS3R shape (one tiny, nearly round Gaussian per pixel: 13 800-entry lists of which a
// block takes 8 %) pays more for the two extra slabs than they remove: 0.452 vs 0.475 ms.
template <bool OCT>
__device__ __forceinline__ void build_row_lists(const StagedSplat *stage, const int n, const float q0x, const float q0y, const int lane,
                                                uint8_t (*lists)[ROW_LIST_STRIDE], int len[4]) {
    len[0] = len[1] = len[2] = len[3] = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int s = k * 64 + lane;
        const bool valid = s < n;
        const float4 p = stage[valid ? s : 0].xyh;
        float hd1 = 1e30f, hd2 = 1e30f;
        if constexpr (OCT) diagonal_extents(p, stage[valid ? s : 0].co, hd1, hd2);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const float bcx = q0x + (float)((r & 1) << 2) + 1.5f, bcy = q0y + (float)((r >> 1) << 2) + 1.5f;
            const bool hit = valid && (OCT ? block_hit_oct(p.x, p.y, p.z, p.w, hd1, hd2, bcx, bcy) : block_hit(p, bcx, bcy));
            const uint64_t m = __ballot(hit);
            const int pos = len[r] + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            if (hit) lists[r][pos] = (uint8_t)s;
            len[r] += __popcll(m);
        }
    }
}

// PREFETCH (few tiles, long lists, globally sorted: the DAS3R shape — 416 workgroups for 1024 workgroup slots, so nobody else
// covers a workgroup's trips to memory): the records of batch i + 1 and the list words of batch i + 2 are in flight while batch i
// composites; a batch used to start with two trips in a row (list word -> record) behind the barrier.
// Round 4: the non-prefetching instantiation is compiled for EIGHT workgroups per CU instead of seven (DESIGN.md r3 section 7.5 iii:
// "320 bytes of LDS and 3 registers away") — 64 VGPRs (waves-per-EU attribute; two spills outside the walk) and exactly 20 480 B of
// LDS: the done flags of the early exit live in the tail of s_gid instead of __syncthreads_count's 256-byte scratch, the list rows
// lose their 4 bytes of bank padding.  1 M splats at 1080p: 0.2298 -> 0.2242 ms (same box, A-B-B-B).
#define FWD_WAVES(PREFETCH) __attribute__((amdgpu_waves_per_eu((PREFETCH) ? 1 : 8, 8)))
template <bool PREFETCH>
__global__ void __launch_bounds__(256) FWD_WAVES(PREFETCH) render_forward_rows_kernel(const uint2 *__restrict__ ranges, const uint32_t *__restrict__ point_list,
                                                                  int W, int H, int tiles_x, int ntiles_strip /*render_common.h pack_tiles*/, const float4 *__restrict__ xyh,
                                                                  const float4 *__restrict__ conic_opacity,
                                                                  const float4 *__restrict__ rgbd, const float *__restrict__ bg,
                                                                  float *__restrict__ final_T, uint32_t *__restrict__ n_contrib,
                                                                  float *__restrict__ out_color, const LocalBin lb,
                                                                  unsigned long long *__restrict__ pairs_arg /*common.h pair_counters()*/) {
#ifdef DAS3R_EXPERIMENTS
    DECODE_PAIRS_OR_TRACE(pairs_arg)   // (tools/wg_trace.py)
#else
    unsigned long long *const pairs = pairs_arg;
#endif
    // PREFETCH: two staging areas in turn — ONE barrier per batch (publish the batch); a wave that is done with batch i goes on
    // to stage batch i + 1 into the other area while slower waves still composite batch i (the barrier of batch i + 1 needs
    // everybody past batch i: the area being overwritten held batch i - 1).  The four quadrants of a tile rarely have equally long sub-lists, and with
    // 1.6 workgroups per CU nobody else fills the wait: PMC on the DAS3R shape showed 58 % of the wave cycles waiting.
    constexpr int NBUF = PREFETCH ? 2 : 1;
    __shared__ StagedSplat stage_all[NBUF * TILE_PIX];
    StagedSplat *stage = stage_all;
    __shared__ uint32_t s_gid[LOCAL_MAX];   // local depth order: the tile's sorted list
    __shared__ uint8_t lists[4][4][ROW_LIST_STRIDE];   // [wave][row][position]
    const int ntiles = packed_ntiles(ntiles_strip);
    const int tile = xcd_tile(blockIdx.x, ntiles_strip, tiles_x);
    if (tile < 0) return;
    PHASE_BEGIN();   // (experiments build: common.h)
#ifdef DAS3R_EXPERIMENTS
    BLK_STAMP(trace, 4, 0)
#endif
    const int tid = threadIdx.x, lane = __lane_id(), wave = tid >> 6, row = lane >> 4;
    const int bx = tile % tiles_x, by = tile / tiles_x;
    int px, py;
    block_pixel(bx, by, wave, lane, px, py);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const float q0x = (float)(bx * TILE_X + ((wave & 1) << 3)), q0y = (float)(by * TILE_Y + ((wave >> 1) << 3));
    const uint2 range = safe_range(ranges[tile], lb.cap);
    const bool sorted_here = lb.point_list != nullptr && (int)(range.y - range.x) <= LOCAL_MAX;   // (uniform)
    // local depth order: sort this tile's list first (a list of one batch is staged by the sort itself)
    const bool prestaged = lb.point_list != nullptr && local_order_tile(lb, range, xyh, conic_opacity, rgbd, stage, s_gid, threadIdx.x, [&](const int k) { (void)k; PHASE_MARK(k) });
    int toDo = (int)(range.y - range.x);
    const int rounds = (toDo + TILE_PIX - 1) / TILE_PIX;
    PHASE_MARK(0)   // prologue + local sort

    // lane state in vector registers, decisions as compare + select pairs: see render_fwd.hip (this kernel issued 1.3e8 scalar
    // instructions per launch at 1 M splats — 211 us of the CU's one scalar ALU in a 277 us kernel)
    PixelBlend pb = {1.0f, 0.f, 0.f, 0.f, inside ? 1.f : 0.f, -1.0f};   // (render_common.h blend_pair)
    float &live = pb.live, &T = pb.T, &C0 = pb.C0, &C1 = pb.C1, &C2 = pb.C2;
    uint32_t last_contributor = 0;

    const int nb = ckpt_buckets(range);                                 // (> 1: a long list, checkpointed for the bucket-parallel backward)
    const int cpix = ((py - by * TILE_Y) << 4) + (px - bx * TILE_X);      // pixel's place in a checkpoint slot
    int next_slot = 0;
    // PREFETCH state: pf = record of this thread's entry of the batch about to be staged, g_ahead = its list word one batch further
    float4 pf0 = make_float4(0.f, 0.f, 0.f, 0.f), pf1 = pf0, pf2 = pf0;
    uint32_t g_ahead = 0u;
    int positions = 0;   // (wave-uniform) list positions walked = 64 pairs each (four 4x4 blocks x 16 pixels)
    if (PREFETCH) {
        const uint32_t len = range.y - range.x;
        if ((uint32_t)tid < len) {
            const uint32_t g = min(point_list[range.x + tid], lb.last_g);
            pf0 = xyh[(size_t)g * SPLAT_REC];
            pf1 = conic_opacity[(size_t)g * SPLAT_REC];
            pf2 = rgbd[(size_t)g * SPLAT_REC];
        }
        if ((uint32_t)(TILE_PIX + tid) < len) g_ahead = point_list[range.x + TILE_PIX + tid];
    }
    for (int i = 0; i < rounds; i++, toDo -= TILE_PIX) {
        if (PREFETCH) {
            stage = stage_all + (i % NBUF) * TILE_PIX;
            if ((i & 15) == 0 && __syncthreads_count(live == 0.f) == TILE_PIX) break;   // (rare on this path: checked every 16 batches)
        } else {
            // the four waves' "all my pixels have stopped" flags live in the last eight words of s_gid, which only a list of more than
            // LOCAL_MAX - 8 entries sorted in LDS needs (such a tile goes without the early exit: the walk of stopped pixels is idle work, not wrong)
            const bool flags_free = !sorted_here || (int)(range.y - range.x) <= LOCAL_MAX - 8;   // (uniform)
            uint32_t *const s_done = s_gid + (LOCAL_MAX - 8) + ((i & 1) << 2);
            const bool wave_done = __ballot(live != 0.f) == 0ull;
            if (flags_free && lane == 0) s_done[wave] = wave_done ? 1u : 0u;
            __syncthreads();
            if (flags_free && (s_done[0] & s_done[1] & s_done[2] & s_done[3])) break;
        }
        PHASE_MARK(1)   // waiting for the workgroup's slowest wave
        if (nb > 1 && i > 0 && (i * TILE_PIX) % BUCKET == 0) ckpt_slot(lb.ckpt, range, tile, next_slot++)[cpix] = make_float4(T, C0, C1, C2);
        const uint32_t progress = range.x + i * TILE_PIX + tid;
        if (PREFETCH) {
            if (progress < range.y) {
                stage[tid].xyh = pf0;
                stage[tid].co = pf1;
                stage[tid].rgbd = pf2;
            } else {
                stage[tid] = null_splat();
            }
            if (progress + TILE_PIX < range.y) {       // batch i + 1: its list word arrived a batch ago
                const uint32_t g = min(g_ahead, lb.last_g);
                pf0 = xyh[(size_t)g * SPLAT_REC];
                pf1 = conic_opacity[(size_t)g * SPLAT_REC];
                pf2 = rgbd[(size_t)g * SPLAT_REC];
            }
            if (progress + 2 * TILE_PIX < range.y) g_ahead = point_list[progress + 2 * TILE_PIX];
        } else if (progress < range.y && !prestaged) {
            const uint32_t g = min(sorted_here ? s_gid[i * TILE_PIX + tid] : (lb.point_list ? lb.point_list[progress] : point_list[progress]), lb.last_g);
            stage[tid].xyh = xyh[(size_t)g * SPLAT_REC];           // one 64-byte record: a single cache line per splat
            stage[tid].co = conic_opacity[(size_t)g * SPLAT_REC];
            stage[tid].rgbd = rgbd[(size_t)g * SPLAT_REC];
        } else if (!prestaged) {
            stage[tid] = null_splat();   // (every staged entry is initialised: see null_splat)
        }
        if (PREFETCH) lds_barrier();   // (the loads just issued stay in flight: render_common.h)
        else __syncthreads();
        const int n = toDo < TILE_PIX ? toDo : TILE_PIX;
        PHASE_MARK(2)   // staging the batch
        int len[4];
        blend_batch_begin(pb);
        build_row_lists<!PREFETCH>(stage, n, q0x, q0y, lane, lists[wave], len);
        const int my_len = row == 0 ? len[0] : (row == 1 ? len[1] : (row == 2 ? len[2] : len[3]));
        const int longest = max(max(len[0], len[1]), max(len[2], len[3]));
        const uint8_t *mine = lists[wave][row];
        PHASE_MARK(3)   // row lists
        // Four list positions per trip: their alpha chains (subtract .. exp .. compare, ~20 dependent instructions each) are
        // independent and interleave; blending stays in list order.  A lone dependent chain leaves the SIMD idle most of the time:
        // r02 PMC on the DAS3R shape (1.6 waves per SIMD) showed 54 % of the wave cycles waiting, and removing every LDS wait
        // from the loop by software pipelining changed nothing — it was the ALU's own latency.  Measured, 1 / 2 / 4 / 8 positions
        // per trip: DAS3R shape 0.656 / 0.537 / 0.495 / 0.494 ms, 1 M splats at 1080p (7 waves per SIMD) 0.311 / 0.287 / 0.279 ms.
        // Round 3: the three tests of a position (is there one, power <= 0, alpha >= 1/255) are taken by ARITHMETIC, not by compare +
        // select pairs: on this chip v_cmp and v_cndmask issue at half the rate of a plain VALU instruction (1.76 vs 0.96 ns per
        // wave instruction on a saturated SIMD: profiles/r03_valu_rate_probe.txt), and the loop spent 12 of them per position.
        //   * "is there a position t + u in my list": rem - u is >= 1 where there is and <= 0 where not — a third operand of the
        //     alpha clamp (where it decides, the minimum fails the 1/255 test); the list byte is read regardless (in bounds: the
        //     row stride holds 260 bytes) and may name any staged entry, initialised or not: v_min3 / v_min drop a NaN operand;
        //   * the two tests of pair_alpha and the blend itself (stop at T < 1e-4, last contributor): alpha_if_visible / blend_pair
        //     (render_common.h) — 14 full-rate instructions per position where the compare + select version had 22 issue slots.
        // Same decisions as pair_alpha, bit for bit (tests: image, n_contrib and final_T against the quad kernel and the oracle).
        constexpr int UNROLL = 4;
        const float lenf = (float)my_len;
        for (int t = 0; t < longest; t += UNROLL) {
            if ((t & 15) == 0 && __ballot(live != 0.f) == 0ull) break;   // every pixel of the quadrant has stopped
            positions += UNROLL;
            int jj[UNROLL];
            float aa[UNROLL];
            const float rem = lenf - (float)t;   // positions of my list from t on
            const uint32_t packed = *reinterpret_cast<const uint32_t *>(mine + t);   // four list bytes (t is a multiple of 4, the row stride too)
#pragma unroll
            for (int u = 0; u < UNROLL; u++) {
                const int j = (int)((packed >> (8 * u)) & 0xFFu);
                const float4 p = stage[j].xyh;
                const float4 co = stage[j].co;
                const float dx = p.x - pxf, dy = p.y - pyf;
                const float q = __fmaf_rn(__fmul_rn(co.x, dx), dx, __fmul_rn(__fmul_rn(co.z, dy), dy));
                const float power = __fmaf_rn(-0.5f, q, -__fmul_rn(__fmul_rn(co.y, dx), dy));   // (pair_alpha's arithmetic)
                const float a1 = fminf(fminf(0.99f, __fmul_rn(co.w, __expf(power))), rem - (float)u);
                jj[u] = j;
                aa[u] = alpha_if_visible(a1, power);   // (a1 is negative where the list has no position)
            }
#pragma unroll
            for (int u = 0; u < UNROLL; u++) {
                // (no wave-uniform skip here: a listed position nearly always has a taker among the four rows, and every such test is a
                //  VALU -> SALU -> branch round trip)
                const int j = jj[u];
                const float4 c = lds_read4(&stage[j].rgbd);   // (b128, not b96: half the LDS cycles)
                blend_pair(pb, aa[u], c, (float)((packed >> (8 * u)) & 0xFFu));   // (v_cvt_f32_ubyte<u>: straight from the list word)
            }
        }
        last_contributor = blend_batch_end(pb, last_contributor, (uint32_t)(i * TILE_PIX));
        PHASE_MARK(4)   // the walk
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
    PHASE_MARK(5)   // output
    PHASE_END(0)
#ifdef DAS3R_EXPERIMENTS
    BLK_STAMP(trace, 4, 1)
    if (trace != nullptr) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        BLK_STAMP(trace, 4, 2)
        if (threadIdx.x == 0 && blockIdx.x < (uint32_t)TRACE_WGS) trace[((size_t)4 * TRACE_WGS + blockIdx.x) * TRACE_STAMPS + 3] = range.y - range.x;
    }
#endif
    if (pairs != nullptr && lane == 0 && positions > 0) {
        atomicAdd(pairs, (unsigned long long)positions * 64ull);
        atomicAdd(pairs + 2, (unsigned long long)positions);
    }
}

// The row lists cost ~200 instructions per wave and batch to build: worth it once a tile's list is long.  DAS3R_RENDER=quad /
// rows forces one of the two forward kernels (A-B runs, tests).
bool use_row_private(int64_t instances, int ntiles) {
    const int forced = switches().render_fwd;
    if (forced == 1 || forced == 2) return forced == 2;   // (3: render_lanes.hip where it applies, else by length)
    return instances >= (int64_t)128 * ntiles;
}

int launch_render_forward_rows(const das3r_raster_args *a, float *out_color, char *geom, char *binning, char *img, const Layout &L,
                               const LocalBin &lb, hipStream_t s) {
#define ARGS                                                                                                              \
    (const uint2 *)(img + L.pub.ranges), (const uint32_t *)(binning + L.pub.point_list), a->image_width, a->image_height,  \
        L.tiles_x, pack_tiles(L), (const float4 *)(geom + L.pub.xy), (const float4 *)(geom + L.pub.conic_opacity),         \
        (const float4 *)(geom + L.pub.rgbd), a->bg, (float *)(img + L.pub.final_T), (uint32_t *)(img + L.pub.n_contrib),   \
        out_color, lb, PAIRS_ARG
    // globally sorted lists and fewer tiles than the chip has workgroup slots (4 per CU and more): see PREFETCH
    const bool prefetch = lb.point_list == nullptr && L.ntiles <= 1024 && !switches().fwd_no_prefetch;
    if (prefetch) DAS3R_LAUNCH((render_forward_rows_kernel<true>), dim3(xcd_grid(L)), dim3(TILE_PIX), 0, s, ARGS);
    else DAS3R_LAUNCH((render_forward_rows_kernel<false>), dim3(xcd_grid(L)), dim3(TILE_PIX), 0, s, ARGS);
#undef ARGS
    KERNEL_CHECK(s, a->debug, "render_forward_rows");
    return DAS3R_OK;
}

}  // namespace das3r
