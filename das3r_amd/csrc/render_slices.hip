// render_slices.hip — forward compositing (K6) for few tiles with long, SPATIALLY SORTED lists: four lanes per pixel like
// render_lanes.hip, but a block's list of a batch is cut into chunks that ANY wave of the tile takes.  Replaces
// upstream:cuda_rasterizer/forward.cu renderCUDA like render_fwd.hip / render_rows.hip / render_lanes.hip; same inputs and outputs.
//
// Why (round 6; docs/ledger.md (bb)).  On the depth maps of a real scene a tile's depth-sorted list is spatially sorted
// too: a batch of 512 consecutive entries is a thin iso-depth LINE across the tile, the blocks on the line hold a few hundred entries
// each and the others none — tools/probes/workload_blocks.py on `dsc`: the slowest block of a batch walks 3.1 x the steps of the mean
// block (1.5 x on the shifted reliefs of the train step, 1.2 x on random depth), and with a wave per block and a barrier per batch
// (render_lanes.hip) the tile pays the slowest.  A block's list is a chain only through T = prod (1 - alpha); everything else of a
// (pixel, entry) pair — forty of fifty instructions — is independent of the pixel's state.  So:
//   * per batch every wave culls for its own block as before (lists + lengths in LDS);
//   * the sixteen lists are cut into chunks of CH entries (CH = a multiple of 8 chosen per batch so that there are at most SL_ITEMS
//     chunks); chunk number i goes to wave i mod 16, WHICHEVER block it belongs to: the wave walks it from T = 1 (no stop can be
//     taken: nobody knows the pixel's T here) and leaves, per pixel, P = prod (1 - alpha), the colour sum relative to T = 1 and the
//     chunk's last visible entry;
//   * the block's owner then composes its chunks in list order: T' = T P, C += T C_chunk, last = chunk's last — a dozen instructions
//     per chunk — as long as T P stays clear of the stop threshold for every live pixel.  Where it does not (a pixel stops inside the
//     chunk, or comes within 1e-4 relative of doing so: once or twice in a pixel's life) the owner walks THAT chunk exactly, from the
//     pixels' true state, with the stop logic of render_quad.h lanes_walk — so every stop, n_contrib and the T a stop is decided on
//     are the sequential kernels'.
// Status (measured, MI355X, same box A-B; docs/ledger.md (bb)): NOT a default.  `dsc` (5 M splats on one relief, pixels never saturate): forward
// 0.562 -> 0.412 ms, and the kernel no longer cares about depth coherence (`ds`: 0.413).  But `ds` itself loses (0.332 -> 0.413: two
// barriers and the chunk bookkeeping per batch against 1.2 x of imbalance), and so does every shape DAS3R actually trains at: there
// 70 - 97 % of the pixels STOP inside the first third of their tile's list, a block's sixteen stops fall into about half of its chunks
// while it lives, every such chunk is walked twice (once from T = 1 by whoever drew it, once exactly by the owner) and the chain is
// not shorter — train step forward 0.239 -> 0.323 ms (noise depth), 0.300 -> 0.444 (smooth depth), a whole self-consistent Sintel-shaped job
// 8.05 -> 8.57 s (on that sequence the LONG tiles — the dense region of the frame, whose chain the kernel lasts for — are exactly the ones whose pixels
// saturate: phase clocks on the tiles above 20 000 entries put 7 x the chunk walks' cycles into the owners' exact re-walks).  Selected with DAS3R_RENDER=slices only; kept because it is exact where it matters (tests) and is the measured answer to
// "split a long tile list across waves": the list can be split, the stops cannot.
// What differs from them: T after a composed chunk is fl(T fl(prod)) instead of the running product (a relative 1e-7 per chunk; the
// stop decision itself is always taken by the exact walk), and the colour is the same sum in another order.  final_T / n_contrib agree
// with render_lanes.hip except where a pixel's T comes within that rounding of 1e-4 (tests: within util.FLIP_FRACTION, as against the oracle).
#include "render_quad.h"

namespace das3r {

constexpr int SL_ITEMS = 48;               // chunk results per batch (12 KB + 1.5 KB of LDS: two workgroups per CU with the staging areas)
constexpr uint32_t SL_NONE = 0xFFFFu;

// One chunk of a block's list, four lanes per pixel, from T = 1 and without stops.  list[0 .. len): staged indices.
// -> P (the same in a quad's four lanes), this lane's colour sums, staged index of this lane's last visible entry (-1: none).
__device__ __forceinline__ void slice_walk(const StagedSplat *__restrict__ stage, const uint16_t *__restrict__ list, const int len, const float pxf,
                                           const float pyf, const int k, const float mk0, const float mk1, const float mk2, float &P, float &C0, float &C1,
                                           float &C2, float &lastf) {
    float T = 1.0f;
    C0 = C1 = C2 = 0.f;
    lastf = -1.0f;
    const float lenf = (float)len - (float)k;
    for (int t = 0; t < len; t += 4 * LN_UNROLL) {
        int j[LN_UNROLL];
        float4 c[LN_UNROLL];
        float av[LN_UNROLL];
        const float rem = lenf - (float)t;
#pragma unroll
        for (int u = 0; u < LN_UNROLL; u++) j[u] = (int)list[t + 4 * u + k];
#pragma unroll
        for (int u = 0; u < LN_UNROLL; u++) {   // (render_quad.h lanes_walk entry_alpha: the same arithmetic, bit for bit)
            const float4 p = stage[j[u]].xyh;
            const float4 co = stage[j[u]].co;
            c[u] = lds_read4(&stage[j[u]].rgbd);
            const float dx = p.x - pxf, dy = p.y - pyf;
            const float qq = __fmaf_rn(__fmul_rn(co.x, dx), dx, __fmul_rn(__fmul_rn(co.z, dy), dy));
            const float power = __fmaf_rn(-0.5f, qq, -__fmul_rn(__fmul_rn(co.y, dx), dy));
            const float a1 = fminf(fminf(0.99f, __fmul_rn(co.w, __expf(power))), rem - (float)(4 * u));
            av[u] = alpha_if_visible(a1, power);
        }
#pragma unroll
        for (int u = 0; u < LN_UNROLL; u++) {
            const float a = av[u], om = 1.0f - a;
            float f0, f1, f2;
            quad_factors(om, mk0, mk1, mk2, f0, f1, f2);
            const float x = __fmul_rn(__fmul_rn(__fmul_rn(T, f0), f1), f2);   // T in front of my entry, multiplied in list order
            const float wT = a * x;
            C0 = __fmaf_rn(c[u].x, wT, C0);
            C1 = __fmaf_rn(c[u].y, wT, C1);
            C2 = __fmaf_rn(c[u].z, wT, C2);
            lastf = max_raw(lastf, min_raw((float)j[u], __fmaf_rn(a, 1e30f, -1.0f)));
            T = quad_perm<0xFF>(__fmul_rn(x, om));                            // behind the quad's last entry
        }
    }
    P = T;
}

__global__ void __launch_bounds__(LN_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) render_forward_slices_kernel(
    const uint2 *__restrict__ ranges, const uint32_t *__restrict__ point_list, int W, int H, int tiles_x, int ntiles_strip /*render_common.h pack_tiles*/,
    const float4 *__restrict__ xyh, const float4 *__restrict__ conic_opacity, const float4 *__restrict__ rgbd, const float *__restrict__ bg,
    float *__restrict__ final_T, uint32_t *__restrict__ n_contrib, float *__restrict__ out_color, const LocalBin lb,
    unsigned long long *__restrict__ pairs /*common.h pair_counters()*/) {
    __shared__ StagedSplat stage_all[2 * LN_BATCH];
    __shared__ uint16_t lists[16][LN_LIST];          // [block][position]: staged index
    __shared__ float4 s_res[SL_ITEMS][16];           // [chunk][pixel of its block]: P, C0, C1, C2 relative to T = 1
    __shared__ uint16_t s_last[SL_ITEMS][16];        //                               staged index of the last visible entry, SL_NONE
    __shared__ uint32_t s_len[16], s_done[16];
    __shared__ uint32_t s_next[2];                   // next chunk number to hand out (one word per batch parity: re-armed a batch ahead)
    const int tile = xcd_tile(blockIdx.x, ntiles_strip, tiles_x);
    if (tile < 0) return;
    const int tid = threadIdx.x, lane = __lane_id(), wave = __builtin_amdgcn_readfirstlane(tid >> 6), k = lane & 3, pix = lane >> 2;
    const int bx = tile % tiles_x, by = tile / tiles_x;
    const int px = bx * TILE_X + ((wave & 3) << 2) + (pix & 3), py = by * TILE_Y + ((wave >> 2) << 2) + (pix >> 2);
    const bool inside = px < W && py < H;
    const float bcx = (float)(bx * TILE_X + ((wave & 3) << 2)) + 1.5f, bcy = (float)(by * TILE_Y + ((wave >> 2) << 2)) + 1.5f;
    const uint2 range = safe_range(ranges[tile], lb.cap);
    const uint32_t n = range.y - range.x;
    const int rounds = (int)((n + LN_BATCH - 1) / LN_BATCH);
    QuadLane q;   // the state of MY block's pixels (T and live the same in a quad's four lanes; C: partial sums per lane)
    q.T = 1.0f; q.live = inside ? 1.f : 0.f; q.C0 = q.C1 = q.C2 = 0.f;
    q.pxf = (float)px; q.pyf = (float)py; q.k = k; q.kf = (float)k;
    q.mk0 = k > 0 ? 0.f : 1.f; q.mk1 = k > 1 ? 0.f : 1.f; q.mk2 = k > 2 ? 0.f : 1.f;
    uint32_t last_contributor = 0;
    const int nb = ckpt_buckets(range);
    const int cpix = ((py - by * TILE_Y) << 4) + (px - bx * TILE_X);
    int next_slot = 0;
    int steps = 0;
    for (int i = tid; i < 16 * LN_LIST; i += LN_THREADS) (&lists[0][0])[i] = 0;   // (a stale list word must name a staged entry)
    if (tid < 2) s_next[tid] = 16u;   // (chunks 0 .. 15 are the waves' first ones; the counter hands out the rest)

    // staging: as in render_lanes.hip — batch i + 1 goes into the other area during batch i; here it is WRITTEN behind barrier B of
    // batch i (everybody has then finished batch i - 1, its exact re-walks included)
    uint32_t g_ahead = 0u;
    const bool loader = tid < LN_BATCH;   // (uniform per wave)
    if (loader) {
        StagedSplat rec = null_splat();
        if ((uint32_t)tid < n) {
            const uint32_t g = min(point_list[range.x + tid], lb.last_g);
            rec.xyh = xyh[(size_t)g * SPLAT_REC];
            rec.co = conic_opacity[(size_t)g * SPLAT_REC];
            rec.rgbd = rgbd[(size_t)g * SPLAT_REC];
        }
        if ((uint32_t)(LN_BATCH + tid) < n) g_ahead = point_list[range.x + LN_BATCH + tid];
        stage_all[tid] = rec;
    }
    lds_barrier();
    for (int i = 0; i < rounds; i++) {
        StagedSplat *const stage = stage_all + (i & 1) * LN_BATCH;
        const uint32_t first = (uint32_t)i * LN_BATCH;
        const bool wave_done = __ballot(q.live != 0.f) == 0ull;
        if (nb > 1 && i > 0 && first % BUCKET == 0) {   // the state in front of list position `first`
            const float q0 = quad_sum(q.C0), q1 = quad_sum(q.C1), q2 = quad_sum(q.C2);
            if (k == 0) ckpt_slot(lb.ckpt, range, tile, next_slot)[cpix] = make_float4(q.T, q0, q1, q2);
            next_slot++;
        }
        StagedSplat rec = null_splat();
        const uint32_t progress = range.x + first + LN_BATCH + tid;   // my entry of batch i + 1
        if (loader && i + 1 < rounds) {
            if (progress < range.y) {
                const uint32_t g = min(g_ahead, lb.last_g);
                rec.xyh = xyh[(size_t)g * SPLAT_REC];
                rec.co = conic_opacity[(size_t)g * SPLAT_REC];
                rec.rgbd = rgbd[(size_t)g * SPLAT_REC];
            }
            if (progress + LN_BATCH < range.y) g_ahead = point_list[progress + LN_BATCH];
        }
        // ---- my block's list of the batch ----------------------------------------------------------------------------------------
        uint16_t *const mine = lists[wave];
        int len = 0;
        if (!wave_done) {
            const int nstaged = (int)min(n - first, (uint32_t)LN_BATCH);
#pragma unroll
            for (int c = 0; c < LN_BATCH / 64; c++) {
                const int s = c * 64 + lane;
                const float4 p = stage[s].xyh;   // (entries past the list hold extents no block can meet)
                const bool hit = s < nstaged && fabsf(p.x - bcx) <= p.z + 1.5f && fabsf(p.y - bcy) <= p.w + 1.5f;
                const uint64_t m = __ballot(hit);
                const int at = len + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                if (hit) mine[at] = (uint16_t)s;
                len += __popcll(m);
            }
        }
        if (lane == 0) {
            s_len[wave] = (uint32_t)len;
            s_done[wave] = wave_done ? 1u : 0u;
        }
        lds_barrier();   // B: lists and lengths of the batch; everybody has left batch i - 1
        if (loader && i + 1 < rounds) stage_all[((i + 1) & 1) * LN_BATCH + tid] = rec;
        {   // every pixel of the tile has stopped?
            const uint32_t d = s_done[lane & 15];
            if (__ballot(d != 0u) == ~0ull) break;
        }
        // ---- the chunks: CH entries each, numbered block by block; chunk number c of the batch is wave (c mod 16)'s ---------------------
        int my_len, total;
        {
            const int l = (int)s_len[lane & 15];
            my_len = l;
            int t = l;
#pragma unroll
            for (int o = 8; o >= 1; o >>= 1) t += __shfl_xor(t, o, 64);
            total = __builtin_amdgcn_readfirstlane(t);
        }
        // at most SL_ITEMS chunks: sum ceil(len / CH) <= total / CH + 16
        const int CH = max(8, (((total + (SL_ITEMS - 16) - 1) / (SL_ITEMS - 16)) + 7) & ~7);
        int my_nch;
        {
            int qn = (int)((float)(my_len + CH - 1) * __builtin_amdgcn_rcpf((float)CH));   // (exact after the two corrections: small integers)
            if (qn * CH > my_len + CH - 1) qn--;
            if ((qn + 1) * CH <= my_len + CH - 1) qn++;
            my_nch = qn;
        }
        int my_base = my_nch;   // exclusive prefix over the sixteen blocks (lanes 0 .. 15 of every row hold blocks 0 .. 15)
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
            const int up = __shfl_up(my_base, o, 16);
            if ((lane & 15) >= o) my_base += up;
        }
        const int nitems = __builtin_amdgcn_readfirstlane(__shfl(my_base, 15, 64));
        my_base -= my_nch;
        // chunk number `wave` first, then whatever the counter hands out: the chunks differ in length (every block's last one is short) and
        // sixteen waves taking them in turn left the batch waiting for whoever drew three long ones (ds: 30 steps per batch against 17.7 of work)
        if (tid == 0) s_next[(i + 1) & 1] = 16u;   // (the other parity's word: nobody touches it before barrier B of the next batch)
        for (int item = wave; item < nitems;) {
            // the block whose chunks contain number `item`: the last one whose first chunk number is <= item (empty blocks share their successor's)
            const uint64_t le = __ballot((lane < 16) && (my_base <= item) && (my_nch > 0));
            const int b = 63 - __builtin_clzll(le);                       // (le != 0: item < nitems)
            const int c = item - __shfl(my_base, b, 64);
            const int blen = __shfl(my_len, b, 64);
            const int off = __builtin_amdgcn_readfirstlane(c * CH);
            const int clen = __builtin_amdgcn_readfirstlane(min(CH, blen - c * CH));
            const int bb = __builtin_amdgcn_readfirstlane(b);
            const float ipx = (float)(bx * TILE_X + ((bb & 3) << 2) + (pix & 3)), ipy = (float)(by * TILE_Y + ((bb >> 2) << 2) + (pix >> 2));
            float P, C0, C1, C2, lastf;
            slice_walk(stage, lists[bb] + off, clen, ipx, ipy, k, q.mk0, q.mk1, q.mk2, P, C0, C1, C2, lastf);
            steps += (clen + 3) >> 2;
            const float s0 = quad_sum(C0), s1 = quad_sum(C1), s2 = quad_sum(C2);
            const float lm = fmaxf(fmaxf(lastf, quad_perm<0xB1>(lastf)), fmaxf(quad_perm<0x4E>(lastf), quad_perm<0x1B>(lastf)));
            if (k == 0) {
                s_res[item][pix] = make_float4(P, s0, s1, s2);
                s_last[item][pix] = lm >= 0.0f ? (uint16_t)lm : (uint16_t)SL_NONE;
            }
            uint32_t nx = 0u;
            if (lane == 0) nx = atomicAdd(&s_next[i & 1], 1u);
            item = __builtin_amdgcn_readfirstlane((int)nx);
        }
        lds_barrier();   // C: the chunks' results
        if (wave_done) continue;   // (uniform; the wave has done its share of everything)
        // ---- my block: compose its chunks in list order; walk exactly the chunk in which a pixel may stop ---------------------------------
        {
            const int base = __builtin_amdgcn_readfirstlane(__shfl(my_base, wave, 64));
            const int nch = __builtin_amdgcn_readfirstlane(__shfl(my_nch, wave, 64));
            for (int c = 0; c < nch; c++) {
                const float4 r = s_res[base + c][pix];
                const uint32_t lst = s_last[base + c][pix];
                const float tn = __fmul_rn(q.T, r.x);
                // a stop needs test_T < 1e-4 somewhere in the chunk, and test_T only falls: the sequential product at the chunk's end is
                // within (entries + 1) 2^-24 <= 3.1e-5 relative of tn — at or above the bar, no pixel can have stopped
                const bool risky = (q.live != 0.f) && (tn < 0.0001f * 1.0001f);
                if (__builtin_expect(__ballot(risky) == 0ull, 1)) {
                    if (q.live != 0.f) {
                        if (k == 0) {
                            q.C0 = __fmaf_rn(q.T, r.y, q.C0);
                            q.C1 = __fmaf_rn(q.T, r.z, q.C1);
                            q.C2 = __fmaf_rn(q.T, r.w, q.C2);
                        }
                        q.T = tn;
                        if (lst != SL_NONE) last_contributor = first + lst + 1u;
                    }
                } else {
                    const int clen = min(CH, len - c * CH);
                    const float lastf = lanes_walk(stage, mine + c * CH, clen, q, steps);
                    if (lastf >= 0.0f) last_contributor = first + (uint32_t)lastf + 1u;
                }
            }
        }
    }
    const float q0 = quad_sum(q.C0), q1 = quad_sum(q.C1), q2 = quad_sum(q.C2);
    const uint32_t last = quad_max(last_contributor);
    if (k == 0)
        for (; nb > 1 && next_slot < nb; next_slot++) ckpt_slot(lb.ckpt, range, tile, next_slot)[cpix] = make_float4(q.T, q0, q1, q2);   // (final values)
    if (inside && k == 0) {
        const size_t at = (size_t)py * W + px, plane = (size_t)H * W;
        final_T[at] = q.T;
        n_contrib[at] = last;
        out_color[at] = q0 + q.T * bg[0];
        out_color[plane + at] = q1 + q.T * bg[1];
        out_color[2 * plane + at] = q2 + q.T * bg[2];
    }
    if (pairs != nullptr && lane == 0 && steps > 0) {
        atomicAdd(pairs, (unsigned long long)steps * 64ull);
        atomicAdd(pairs + 2, (unsigned long long)steps);
    }
}

int launch_render_forward_slices(const das3r_raster_args *a, float *out_color, char *geom, char *binning, char *img, const Layout &L, const LocalBin &lb,
                                 hipStream_t s) {
#define ARGS                                                                                                                                   \
    (const uint2 *)(img + L.pub.ranges), (const uint32_t *)(binning + L.pub.point_list), a->image_width, a->image_height, L.tiles_x, pack_tiles(L), \
        (const float4 *)(geom + L.pub.xy), (const float4 *)(geom + L.pub.conic_opacity), (const float4 *)(geom + L.pub.rgbd), a->bg,                \
        (float *)(img + L.pub.final_T), (uint32_t *)(img + L.pub.n_contrib), out_color, lb, pair_counters()
    DAS3R_LAUNCH(render_forward_slices_kernel, dim3(xcd_grid(L)), dim3(LN_THREADS), 0, s, ARGS);
#undef ARGS
    KERNEL_CHECK(s, a->debug, "render_forward_slices");
    return DAS3R_OK;
}

}  // namespace das3r
