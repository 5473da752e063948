// common.h — shared device/host helpers for libdas3r_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/das3r_raster.h"
#include "pretransform_math.h"

#define TILE_X 16
#define TILE_Y 16
#define TILE_PIX 256
#define WAVE 64
#define NEAR_PLANE 0.001f  // /root/reference/README.md:41-44

namespace das3r {

void set_error(const char *fmt, ...);

// Experiment / diagnostic switches, read from the environment ONCE — at the first call into the library and
// again whenever das3r_reload_switches() is called (tests and tools flip them between calls) — not on every launch.
struct Switches {
    int sort_ipl;          // DAS3R_SORT_IPL = 4 | 8 | 16: keys per lane of the radix passes (0: by size)
    bool sort_classic;     // DAS3R_SORT=classic: histogram + row scan + scatter per digit instead of the one-sweep passes
    bool rect_upstream;    // DAS3R_RECT=upstream: bin over upstream's 3-sigma square (bit-exact list tests)
    bool verbose;          // DAS3R_VERBOSE
    int binning;           // DAS3R_BINNING=local | radix | seg | seg3: 1 | -1 | 2 | 3 (0: chosen per scene; seg3 = seg with one more partition pass of bucket bits)
    bool capacity_exact;   // DAS3R_CAPACITY=exact: never lay the binning buffer out speculatively
    bool fused_emit_off;   // DAS3R_FUSED_EMIT=0
    bool no_sh_stage;      // DAS3R_NO_SH_STAGE
    int render_fwd;        // DAS3R_RENDER=quad | rows | lanes | slices | fine: 1 | 2 | 3 | 4 | 5 (0: by list length and tile count)
    int render_bwd;        // DAS3R_RENDER_BWD=dpp | mfma | scan... | stream | blk...: 1 | 2 | 3 | 5 | 6 (0: by list length)
    int render_bwd_mb;     // scan64 / scan128 / scan256, blk64 / blk128 / blk256: entries per round; scana256 / scana512: 1000 + entries, atomic flush
    int tile_chunk;        // DAS3R_TILE_CHUNK: tiles per chunk of the XCD round robin (render_common.h); -1 = default, 0 = contiguous eighths
    int scan_items;        // DAS3R_SCAN_ITEMS = 1 | 2 | 4 | 8 | 16: ranks per thread of the scan + emission kernel (0: by size)
    bool deterministic;    // DAS3R_DETERMINISTIC=1: bit-identical gradients run to run (the block-walk backward for every list length:
                           // the pixel-per-lane kernel meets its four waves with LDS float atomics, whose order varies)
    int render_bwd_occ;    // blk...o<4|5>: workgroups per CU the kernel is compiled for (register cap)
    bool tile_lpt_off;     // DAS3R_TILE_LPT=0
    int render_bwd_pix;    // blk...p<0|1|2>: where render_bwd_blk.hip keeps the per-pixel values (render_blk.h)
    bool bwd_reduce_set, bwd_reduce_shfl;   // DAS3R_BWD_REDUCE=shfl | dpp (reference reduction of the pixel-per-lane kernel)
    bool ablate_set;       // DAS3R_ABLATE (perf experiments on the pixel-per-lane kernel)
    int ablate;
    int tickets;           // DAS3R_TICKETS=always | never | <bound>: 0 | 1 << 30 | bound (-1: from the device's CU count)
    int bwd_pad_lds, fwd_pad_lds;   // DAS3R_BWD_PAD_LDS / DAS3R_FWD_PAD_LDS: extra dynamic LDS (occupancy experiments)
    int bwd_buckets;       // DAS3R_BWD_BUCKETS=0 | <slices>: bucket-parallel backward off / forced with that many slices (-1: by list length)
    int inject_fault;
    int mutate;            // das3r_debug_mutate (tests): 1 = the block-walk backward evaluates exp(power) (1 + 1e-4) — a biased kernel the parity tests must catch
    bool fwd_no_prefetch;   // DAS3R_FWD_PREFETCH=0: the rows forward kernel without its software prefetch (A-B runs)
    int tile_strip;   // DAS3R_TILE_STRIP: rows per strip of the compositing kernels' tile order (0 = row-major)      // DAS3R_INJECT_FAULT: bits OR-ed into the binning self-check word of every forward (fault-injection tests)
};
const Switches &switches();

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            das3r::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return DAS3R_ERR_HIP;                                                              \
        }                                                                                      \
    } while (0)

// After a kernel launch: always check the launch; in debug mode also synchronise so that a faulting kernel is
// reported at the call that caused it (upstream CHECK_CUDA(..., debug) behaviour).
#define KERNEL_CHECK(stream, debug, name)                                                      \
    do {                                                                                       \
        hipError_t _e = hipGetLastError();                                                     \
        if (_e == hipSuccess && (debug)) _e = hipStreamSynchronize(stream);                    \
        if (_e != hipSuccess) {                                                                \
            das3r::set_error("kernel %s failed: %s", name, hipGetErrorString(_e));             \
            return DAS3R_ERR_HIP;                                                              \
        }                                                                                      \
    } while (0)

// Optional per-kernel timing with HIP events recorded on the launch stream (das3r_profile_* in the C-ABI).
// Declare one right before a launch; when profiling is off it costs one branch.
bool profile_enabled();
void profile_begin(const char *name, hipStream_t s);
void profile_end(hipStream_t s);
struct ProfTimer {
    hipStream_t s;
    bool on;
    ProfTimer(const char *name, hipStream_t stream) : s(stream), on(profile_enabled()) {
        if (on) profile_begin(name, s);
    }
    ~ProfTimer() {
        if (on) profile_end(s);
    }
};

// Optional pair counters (das3r_pair_counters in the C-ABI; bench.py's pairs/s): a device array of four 64-bit words, or null.
//   [0] (pixel, splat) pairs the forward compositing kernel evaluated   [1] the same for the backward kernel
//   [2] forward wave iterations                                          [3] backward wave iterations (64 pairs each)
// The compositing kernels take the pointer as their last argument; each wave adds its totals once, at its end.
unsigned long long *pair_counters();
// behind the four counters: phase clocks of the compositing kernels (experiments build only: api.hip das3r_debug_phase_clocks)
constexpr int PHASE_WORDS = 16, PHASE_COPIES = 64;   // (64 copies, picked by workgroup: thousands of atomics on ONE word serialise in L2)
#ifdef DAS3R_EXPERIMENTS
// wave 0's lane 0 adds the clocks since its previous mark to phase word k — in LDS (eight words per workgroup: one thread, no
// atomics), flushed to the device words base .. base + 7 when the kernel ends.  (Marks that went straight to global atomics
// queued ahead of the kernel's own loads and showed up as "staging".)
#define PHASE_BEGIN()                                                                      \
    __shared__ unsigned long long phase_w_[8];                                             \
    unsigned long long phase_t_ = 0ull;                                                    \
    if (pairs != nullptr && threadIdx.x == 0) {                                            \
        for (int k_ = 0; k_ < 8; k_++) phase_w_[k_] = 0ull;                                \
        phase_t_ = clock64();                                                              \
    }
#define PHASE_MARK(k)                                                                      \
    if (pairs != nullptr && threadIdx.x == 0) {                                            \
        const unsigned long long now_ = clock64();                                         \
        phase_w_[(k) & 7] += now_ - phase_t_;                                              \
        phase_t_ = now_;                                                                   \
    }
#define PHASE_END(base)                                                                    \
    if (pairs != nullptr && threadIdx.x == 0)                                              \
        for (int k_ = 0; k_ < 8; k_++) atomicAdd(pairs + 4 + (blockIdx.x % PHASE_COPIES) * PHASE_WORDS + (base) + k_, phase_w_[k_]);
// Workgroup trace of the partition passes (tools/wg_trace.py): eight 100 MHz timestamps per workgroup and pass, by ticket.
constexpr int TRACE_WGS = 16384, TRACE_PASSES = 8, TRACE_STAMPS = 8;
unsigned long long *wg_trace();   // null unless das3r_debug_wg_trace(1) (api.hip)
// The compositing kernels have one diagnostic pointer argument (`pairs`): while the workgroup trace is on and the pair counters
// are off it carries the trace buffer instead, marked by its lowest bit.
static inline unsigned long long *pairs_or_trace() {
    unsigned long long *p = pair_counters();
    if (p == nullptr && wg_trace() != nullptr) p = reinterpret_cast<unsigned long long *>(reinterpret_cast<uintptr_t>(wg_trace()) | 1u);
    return p;
}
#define DECODE_PAIRS_OR_TRACE(arg)                                                                                         \
    unsigned long long *const trace = (reinterpret_cast<uintptr_t>(arg) & 1u)                                              \
                                          ? reinterpret_cast<unsigned long long *>(reinterpret_cast<uintptr_t>(arg) & ~(uintptr_t)1) \
                                          : nullptr;                                                                       \
    unsigned long long *const pairs = trace ? nullptr : (arg);
#define WG_STAMP(k)                                                                                   \
    if (trace != nullptr && threadIdx.x == 0 && s_block < (uint32_t)TRACE_WGS)                        \
        trace[((size_t)(shift >> 3) * TRACE_WGS + s_block) * TRACE_STAMPS + (k)] = wall_clock64();
// the same for scan_emit_kernel: region TRACE_PASSES - 1
#define SCAN_STAMP(k)                                                                                 \
    if (wg_trace_ptr != nullptr && threadIdx.x == 0 && s_block < (uint32_t)TRACE_WGS)                 \
        wg_trace_ptr[((size_t)(TRACE_PASSES - 1) * TRACE_WGS + s_block) * TRACE_STAMPS + (k)] = wall_clock64();
// the same for any kernel whose workgroups are numbered by blockIdx.x: region r of the trace
#define BLK_STAMP(ptr, r, k)                                                                          \
    if ((ptr) != nullptr && threadIdx.x == 0 && blockIdx.x < (uint32_t)TRACE_WGS)                      \
        (ptr)[((size_t)(r) * TRACE_WGS + blockIdx.x) * TRACE_STAMPS + (k)] = wall_clock64();
#define PAIRS_ARG pairs_or_trace()
#else
#define PAIRS_ARG pair_counters()
#define PHASE_MARK(k)
#define PHASE_BEGIN()
#define PHASE_END(base)
#define WG_STAMP(k)
#define SCAN_STAMP(k)
#define BLK_STAMP(ptr, r, k)
#endif

// every kernel launch goes through this macro so that the optional profiler sees it (name = kernel symbol)
#define DAS3R_LAUNCH(kernel, grid, block, shmem, stream, ...)                \
    do {                                                                     \
        das3r::ProfTimer _pt(#kernel, stream);                               \
        hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__); \
    } while (0)

static inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }
static inline int div_up(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ---- radix-sort geometry (sort.hip) ----
constexpr int RADIX_BITS = 8;
constexpr int RADIX_SIZE = 1 << RADIX_BITS;
constexpr int SORT_WAVES_PER_BLOCK = 4;
// keys handled by one wave ("chunk"); chosen per problem size so that small sorts still fill the chip
int sort_ipl_override();   // api.hip: DAS3R_SORT_IPL (perf experiments), 0 = none
static inline int sort_items_per_lane(int64_t n) {
    const int o = sort_ipl_override();
    return o ? o : (n >= (int64_t)(1 << 21) ? 16 : (n >= (1 << 17) ? 8 : 4));
}
static inline int sort_num_chunks(int64_t n) { return n == 0 ? 0 : div_up(n, (int64_t)WAVE * sort_items_per_lane(n)); }
// Width of a tile-id digit: the tile bits are split EVENLY over the ceil(tbits / 8) partition passes (13 bits = 7 + 6, 9 = 5 + 4)
// instead of 8 + rest: a pass costs per digit VALUE (counters to publish and to look back over, write heads kept open), so two
// narrow passes beat a wide one next to a nearly empty one.
__host__ __device__ static inline int tile_digit_width(int tbits) {
    const int passes = (tbits + 7) / 8;
    return passes > 0 ? (tbits + passes - 1) / passes : 8;
}
static inline int tile_bits(int ntiles) {
    int b = 0;
    while ((1 << b) < ntiles) b++;
    return b < 1 ? 1 : b;
}

// Long tile lists are cut into BUCKETs of list positions: the forward compositing kernels leave every pixel's (T, C) at the bucket
// boundaries in the binning buffer (checkpoints: render_common.h), so that the backward pass can replay the buckets of a tile
// in parallel workgroups (render_bwd_scan.hip) — the DAS3R shape has 416 tiles with ~14 k entries each: one workgroup per tile
// leaves the chip at 1.6 waves per SIMD.
constexpr int BUCKET = 1024;
constexpr int SPLAT_REC = 4;   // float4s per Gaussian record (xyh, conic+opacity, rgb+depth, pad): 64 bytes, one cache-line gather

struct Layout {
    das3r_raster_layout pub;
    // private scratch offsets
    size_t g_keyA, g_keyB, g_valA, g_valB, g_hist, g_totals, g_blocksums, g_count, g_off_by_gid, g_rect;   // g_rect: u32 per splat, binned tile rectangle x0 | x1 << 8 | y0 << 16 | y1 << 24 (0 = none)
    size_t b_keyA, b_keyB, b_valA, b_valB, b_hist, b_totals, b_gid_of, b_slot;
    size_t b_e2;     // u32[capacity]: ping-pong partner of b_slot for the emission slots travelling through the partition passes
    size_t b_ckpt;   // float4[(capacity / BUCKET + ntiles + 2) * 256]: per-pixel (T, C) at the bucket boundaries of long tile lists
    // single-pass radix control words (sort_onesweep.hip): [global digit histograms][tickets][status granules], contiguous
    // so that one store loop / one memset zeroes them: geom side by preprocess_kernel, binning side by a memset before emit
    size_t g_ghist, g_ticket, g_status, g_scan_status, g_ctrl_bytes;
    int64_t capacity;  // instance capacity the binning buffer was laid out for
    size_t b_ghist, b_ticket, b_status, b_ctrl_bytes;
    int tiles_x, tiles_y, ntiles, tbits, tile_passes;
    int chunksP, chunksI;
    // segmented binning path (segkey.h): the partition key is (tile id << dbits | depth bucket), kbits = tbits + dbits wide, in the
    // SAME number of passes the tile ids alone need (the backward pass lays the buffers out without knowing the path).  dbits = 0
    // everywhere else.  g_dhist: u32[256] depth histogram of the forward
    int dbits, kbits, kshift;   // kshift: the key's low bits hold the fraction of the depth map (segkey.h: min(16, 32 - kbits) bits), the
                                // partition's digits start above them
    int part_passes;            // passes of the instance partition: tile_passes, or one more when the segmented path wants more bucket bits
    size_t g_dhist;
    const uint32_t *dhist_ptr;  // round 6: where this forward's depth histogram really is (a library-owned slot: api.hip dhist_slots); null = geom + g_dhist
    size_t i_order;   // img buffer, u32[ntiles]: the tiles longest list first (render_regions.hip tile_lpt_kernel; round 6)
};
// depth-bucket bits `passes` partition passes (at most three) have room for beside the tile ids (<= 0: none)
static inline int seg_dbits(const Layout &L, int passes) { return 8 * (passes < 3 ? passes : 3) - L.tbits; }

// Local depth order (short tile lists): the binning skips the global depth sort, the tile lists arrive in index order and
// the forward compositing kernel sorts each one by (depth bits, index) itself (render_common.h: local_sort_tile) — in LDS up
// to LOCAL_MAX entries, in global memory (slow, rare) beyond.  point_list == null: the lists are already in depth order.
struct LocalBin {
    float4 *ckpt;                       // checkpoints of long tile lists (always set; see BUCKET)
    uint32_t *point_list, *slot_list;   // sorted in place
    uint32_t *keys;                     // u32[num_rendered] scratch for the lists that do not fit in LDS (the dead tile keys)
    uint32_t *host_flag;                // pinned mailbox word that receives flag_value when such a list was met
    uint32_t flag_value;                // names the shape (P, W, H) this forward belongs to, never 0
    uint32_t last_g, cap;               // P - 1 and the instances the lists hold: bounds for safe_index / safe_range (always set)
    // round 6: the forward kernel with four workgroups per tile (render_regions.hip) instead of one (render_lanes.hip): chosen by the host for
    // shapes whose tile lists are skewed (api.hip Verdict::fine, decided from the tile ranges themselves: launch_list_skew)
    bool prefer_regions;
    const uint32_t *tile_order;         // with prefer_regions: the tiles, longest list first (tile_lpt_kernel); null: the locality order
};
void compute_layout(int P, int64_t I, int W, int H, Layout *L);

// Emission fused into the preprocess kernel (speculative local-order path, grid resident: api.hip).  status == null: off.
// The control words live in a library-owned ring slot that is zero at rest (ghist, err: re-armed by tile_ranges_kernel) or
// versioned by the forward's tag (status granules), because nothing runs before this kernel that could zero them.
struct EmitArgs {
    unsigned long long *status;   // [nblocks + groups] {tag << 32 | count} granules of the chained scan
    uint32_t tag;                 // unique per forward of this host thread, never 0
    uint32_t *ghist;              // [4][256] tile-digit histograms
    uint32_t *err;                // self-check word
    uint32_t *tile_keys, *gids, *off_by_gid;
    uint32_t cap;
    int tbits;
    uint32_t *count;              // device word the partition passes read the instance count from (count[1] = flags)
    uint32_t *count_out;          // pinned host mailbox {count, flags, tag}
};
constexpr int EMIT_SLOT_WORDS = 64 + 4 * 256;          // err + padding, ghist: the part tile_ranges_kernel re-arms
constexpr int EMIT_STATUS_GRANULES = 1024;             // >= resident grid + its groups

// ---- launchers implemented in the individual .hip files ----
// binning_ctrl (may be null): the binning buffer's control words, zeroed by the same kernel when the buffer already exists
// arrive: a zeroed 64-bit device word (self re-arming); host_out / tag: pinned mailbox that receives num_rendered
int launch_preprocess(const das3r_raster_args *a, const das3r_raster_in *in, int32_t *radii, char *geom, char *img, char *binning_ctrl,
                      size_t binning_ctrl_bytes, const Layout &L, unsigned long long *arrive, uint32_t *host_out, uint32_t tag,
                      hipStream_t s, const EmitArgs *emit = nullptr, uint32_t *dhist = nullptr, uint32_t *dhist_next = nullptr);
int launch_mark_visible(int P, const float *means3D, const float *viewmatrix, uint8_t *present, hipStream_t s);
// part 0: everything that can be enqueued before num_rendered is known; part 1: the rest, which also zeroes the binning buffer's
// control words (binning_ctrl may be null)
int launch_depth_sort(int P, char *geom, const Layout &L, int part, char *binning_ctrl, size_t binning_ctrl_bytes, bool debug, hipStream_t s);
// scan of tiles_touched in depth order -> offsets / off_by_gid / count (exact path: the host then reads the count)
int launch_scan(int P, char *geom, const Layout &L, uint32_t *host_out, uint32_t tag, bool debug, hipStream_t s, bool index_order = false);
// the same scan with the instance emission fused in (hinted path; sort_onesweep.hip)
int launch_scan_emit(int P, int64_t cap, const int32_t *radii, char *geom, char *binning, const Layout &L, uint32_t *host_out, uint32_t tag,
                     bool debug, hipStream_t s, bool index_order = false);
size_t scan_status_bytes(int P);
int launch_binning_scan_emit(int P, int64_t I, const int32_t *radii, char *geom, char *binning, const Layout &L, bool ctrl_zeroed,
                             uint32_t *host_out, uint32_t tag, bool debug, hipStream_t s, bool index_order = false);
size_t onesweep_status_bytes(int64_t n, int passes);
int launch_onesweep_depth_sort(int P, char *geom, const Layout &L, int part, uint32_t *zero_ptr, uint32_t zero_words, bool debug,
                               hipStream_t s);
int launch_onesweep_partition(int64_t cap, char *geom, char *binning, const Layout &L, uint32_t **keys_final, bool debug, hipStream_t s,
                              uint32_t *ghist_override = nullptr, uint32_t *err_override = nullptr);
bool use_row_private(int64_t instances, int ntiles);  // forward: 4x4-block-per-row kernel for long tile lists (render_rows.hip)
bool use_quad_lanes(const Layout &L, const LocalBin &lb);   // forward: four lanes per pixel for few tiles with long lists (render_lanes.hip)
// das3r_raster_in.pre -> the device-side view the per-Gaussian kernels take by value (pretransform_math.h); all null when the caller handed
// over camera-frame tensors
inline PreXform pre_xform(const das3r_raster_in *in) {
    PreXform x = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (in && in->pre) {
        const das3r_pretransform *p = in->pre;
        x = PreXform{p->xyz, p->rot, p->scaling, p->opacity_raw, p->conf_flat, p->mask_index, p->R, p->t, p->Lq};
    }
    return x;
}
int launch_render_forward_regions(const das3r_raster_args *a, float *out_color, char *geom, char *binning, char *img, const Layout &L, const LocalBin &lb,
                                  hipStream_t s);   // the same shapes: sixteen lanes per pixel, a wave per 2x2 region, four workgroups per tile (render_regions.hip)
int launch_tile_lpt(char *img, const Layout &L, uint32_t cap, bool debug, hipStream_t s);   // render_regions.hip
int launch_list_skew(const char *img, const char *binning, const char *geom, const Layout &L, uint32_t cap, uint32_t last_g, uint32_t *mailbox_words, uint32_t tag,
                     bool debug, hipStream_t s);
int launch_render_forward_slices(const das3r_raster_args *a, float *out_color, char *geom, char *binning, char *img, const Layout &L, const LocalBin &lb,
                                 hipStream_t s);   // the same shapes, a block's list cut into chunks any wave takes (render_slices.hip)
int launch_render_forward_lanes(const das3r_raster_args *a, float *out_color, char *geom, char *binning, char *img, const Layout &L,
                                const LocalBin &lb, hipStream_t s);
int launch_render_forward_rows(const das3r_raster_args *a, float *out_color, char *geom, char *binning, char *img, const Layout &L,
                               const LocalBin &lb, hipStream_t s);
int launch_render_backward_mfma(const das3r_raster_args *a, const float *dL_dpix, char *geom, char *binning, char *img, const Layout &L,
                                float *partial, hipStream_t s);
// independent waves (render_bwd_stream.hip): scratch = rows[capacity][4][12] + one byte per row that exists
size_t stream_scratch_bytes(int64_t capacity);
int launch_render_backward_stream(const das3r_raster_args *a, const float *dL_dpix, char *geom, char *binning, char *img, const Layout &L,
                                  float *scratch, hipStream_t s);
// slices > 1: bucket-parallel replay (grid = tiles x slices; needs the forward's checkpoints)
int launch_render_backward_scan(const das3r_raster_args *a, const float *dL_dpix, char *geom, char *binning, char *img, const Layout &L,
                                float *partial, int mb, int slices, hipStream_t s);
// 4x4 block per DPP row, pixel state in registers, fp32 accumulators (render_bwd_blk.hip)
int launch_render_backward_blk(const das3r_raster_args *a, const float *dL_dpix, char *geom, char *binning, char *img, const Layout &L,
                               float *partial, int mb, int slices, hipStream_t s);
int launch_render_backward_regions(const das3r_raster_args *a, const float *dL_dpix, char *geom, char *binning, char *img, const Layout &L,
                               float *partial, int mb, int slices, hipStream_t s);   // render_bwd_rgn.hip
constexpr int LOCAL_MAX = 1024;   // longest tile list the forward kernels sort in LDS
// chained kernels (scan, radix passes) order their workgroups by ticket unless every workgroup of the grid is resident at once
// (api.hip: grid_is_resident)
bool grid_is_resident(int nblocks);
bool use_onesweep();  // DAS3R_SORT=classic selects the three-kernel radix passes (diagnostics / A-B)
bool use_tight_rect();  // DAS3R_RECT=upstream bins over upstream's 3-sigma square (bit-exact list tests)
// host_late / tag: pinned mailbox the last binning kernel copies the self-check word to (see api.hip)
int launch_binning(int P, int64_t I, int W, int H, const int32_t *radii, char *geom, char *binning, char *img, const Layout &L,
                   bool fused_scan, uint32_t *host_late, uint32_t tag, bool debug, hipStream_t s, uint32_t **dead_keys = nullptr,
                   uint32_t *emit_slot = nullptr /*fused emission: {err, pad, ghist} ring slot, re-armed by the last kernel*/,
                   uint32_t *seg_host_flag = nullptr, uint32_t seg_flag_value = 0 /*segmented path (L.dbits > 0): mailbox word for a segment too long for LDS*/);
// segsort.hip: exact (depth bits, index) order inside every (tile, depth bucket) segment of the partitioned list, in place
int launch_segment_sort(int64_t cap, const uint32_t *n_ptr, const uint32_t *keys_final, int fbits, uint32_t *point_list, uint32_t *slot_list,
                        const uint32_t *depth_key, uint32_t last_g, uint32_t *scratch_keys, uint32_t *host_flag, uint32_t flag_value, bool debug,
                        hipStream_t s, int dbits, uint2 *ranges, const uint32_t *err, uint32_t *host_late, uint32_t tag, uint32_t inject);
// lb.point_list != null: the tile lists are in index order and the kernel sorts them by depth first
int launch_render_forward(const das3r_raster_args *a, const float *colors_precomp, float *out_color, char *geom, char *binning,
                          char *img, const Layout &L, const LocalBin &lb, hipStream_t s);
// partial: [num_rendered, 9] per-instance sums written by the render backward, gathered by the preprocess backward
// *quad_rows (out): false = partial[I][9], one row per instance; true = the stream kernel's rows[I][4][12] + existence bytes
int launch_render_backward(const das3r_raster_args *a, const float *dL_dpix, char *geom, char *binning, char *img, const Layout &L,
                           float *partial, hipStream_t s, bool *quad_rows, int64_t num_rendered, uint32_t fwd_flags /*das3r_raster_saved.flags*/);
// pair_count.hip (measurement aid): out[0] += live pairs, out[1] += (pixel, list position) pairs below the pixel's n_contrib
int launch_count_live_pairs(const das3r_raster_args *a, char *geom, char *binning, char *img, const Layout &L, unsigned long long *out, hipStream_t s);
int launch_preprocess_backward(const das3r_raster_args *a, const das3r_raster_in *in, char *geom, char *binning, const Layout &L,
                               const das3r_raster_grads *g, const float *partial, hipStream_t s, bool quad_rows);

// ---- device helpers ----
#ifdef __HIPCC__
__device__ __forceinline__ int lane_id() { return __lane_id(); }

// Wave64 sum reduction on the DPP network (no LDS traffic).  Result valid in lane 63.
// GFX9/CDNA dpp_ctrl encodings: row_shr:n = 0x110+n, row_bcast15 = 0x142, row_bcast31 = 0x143.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_step(float v) {
    int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true);
    return v + __int_as_float(moved);
}
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
    v = dpp_step<0x111, 0xf>(v);  // row_shr:1
    v = dpp_step<0x112, 0xf>(v);  // row_shr:2
    v = dpp_step<0x114, 0xf>(v);  // row_shr:4  (lanes 4.. of each row; partial sums still correct at lane 15)
    v = dpp_step<0x118, 0xf>(v);  // row_shr:8
    v = dpp_step<0x142, 0xa>(v);  // row_bcast15 into rows 1,3
    v = dpp_step<0x143, 0xc>(v);  // row_bcast31 into rows 2,3
    return v;
}
// Sum inside each 16-lane row: lane 15 of every row ends up with its row's total (the first four steps of the above).
__device__ __forceinline__ float row_sum_to_lane15(float v) {
    v = dpp_step<0x111, 0xf>(v);  // row_shr:1
    v = dpp_step<0x112, 0xf>(v);  // row_shr:2
    v = dpp_step<0x114, 0xf>(v);  // row_shr:4
    v = dpp_step<0x118, 0xf>(v);  // row_shr:8
    return v;
}
// Portable (ds_bpermute based) full-wave sum; every lane gets the result.  Reference for the DPP path.
__device__ __forceinline__ float wave_sum_shfl(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
#endif

}  // namespace das3r
