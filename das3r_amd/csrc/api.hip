// api.hip — the C-ABI entry points of libdas3r_hip.so (include/das3r_raster.h) and the buffer layout.
#include <stdarg.h>
#include <string.h>

#include <vector>

#include "common.h"
#include <algorithm>

namespace das3r {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- per-kernel HIP-event profiler (single-threaded use: bench.py / tests) ----
struct ProfRec {
    const char *name;
    hipEvent_t start, stop;
};
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof;
bool profile_enabled() { return g_prof_on; }
// (events come from a pool that survives reports: creating two events per launch made the instrumented pass slower than the
//  timed one — the host fell behind the device and the gaps showed up inside the brackets: VERDICT r2 item 10)
static std::vector<hipEvent_t> g_event_pool;
static hipEvent_t pooled_event() {
    if (!g_event_pool.empty()) { hipEvent_t e = g_event_pool.back(); g_event_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    return hipEventCreate(&e) == hipSuccess ? e : nullptr;
}
void profile_begin(const char *name, hipStream_t s) {
    ProfRec r;
    r.name = name;
    r.start = pooled_event();
    r.stop = pooled_event();
    if (!r.start || !r.stop) return;
    (void)hipEventRecord(r.start, s);
    g_prof.push_back(r);
}
void profile_end(hipStream_t s) {
    if (!g_prof.empty()) (void)hipEventRecord(g_prof.back().stop, s);
}

static Switches g_sw;
static bool g_sw_loaded = false;
static int g_inject_fault = 0;
static int g_mutate = 0;
static void load_switches() {
    Switches w;
    memset(&w, 0, sizeof(w));
    auto env = [](const char *k) -> const char * { const char *e = getenv(k); return (e && e[0]) ? e : nullptr; };
    const char *e;
    // Switches that select between CORRECT code paths (tests force each of them) are read in every build.  Those that exist for
    // timing experiments only — some of them produce wrong results by design (DAS3R_ABLATE) — are read only by an experiments
    // build (make EXPERIMENTS=1): a stray environment variable cannot change what the shipped library computes.  Fault injection
    // for the self-check tests is a call (das3r_debug_inject_fault), not an environment variable.
#ifdef DAS3R_EXPERIMENTS
    if ((e = env("DAS3R_SORT_IPL"))) { const int v = atoi(e); w.sort_ipl = (v == 4 || v == 8 || v == 16) ? v : 0; }
    w.sort_classic = (e = env("DAS3R_SORT")) && e[0] == 'c';
    w.no_sh_stage = getenv("DAS3R_NO_SH_STAGE") != nullptr;
    if ((e = env("DAS3R_ABLATE"))) { w.ablate_set = true; w.ablate = atoi(e); }
    if ((e = env("DAS3R_BWD_PAD_LDS"))) w.bwd_pad_lds = atoi(e);
    if ((e = env("DAS3R_FWD_PAD_LDS"))) w.fwd_pad_lds = atoi(e);
    if ((e = env("DAS3R_SCAN_ITEMS"))) { const int v = atoi(e); w.scan_items = (v == 1 || v == 2 || v == 4 || v == 8 || v == 16) ? v : 0; }
    w.fwd_no_prefetch = (e = env("DAS3R_FWD_PREFETCH")) && e[0] == '0';
#endif
    w.rect_upstream = (e = env("DAS3R_RECT")) && e[0] == 'u';
    w.verbose = env("DAS3R_VERBOSE") != nullptr;
    if ((e = env("DAS3R_BINNING"))) w.binning = e[0] == 'l' ? 1 : (e[0] == 'r' ? -1 : (e[0] == 's' ? (strchr(e, '3') ? 3 : 2) : 0));
    w.capacity_exact = (e = env("DAS3R_CAPACITY")) && e[0] == 'e';
    w.fused_emit_off = (e = env("DAS3R_FUSED_EMIT")) && e[0] == '0';
    w.tile_lpt_off = (e = env("DAS3R_TILE_LPT")) && e[0] == '0';   // (A-B runs: the region forward in the locality order of the other kernels)
    if ((e = env("DAS3R_RENDER"))) w.render_fwd = e[0] == 'q' ? 1 : (e[0] == 'r' ? 2 : (e[0] == 'l' ? 3 : (e[0] == 's' ? 4 : (e[0] == 'f' ? 5 : 0))));
    if ((e = env("DAS3R_RENDER_BWD"))) {   // dpp | mfma | scan[a][64|128|256|512]
        w.render_bwd = e[0] == 'd' ? 1 : (e[0] == 'm' ? 2 : (strncmp(e, "stream", 6) == 0 ? 5 : (e[0] == 's' ? 3 : (e[0] == 'b' ? 6 : (e[0] == 'f' ? 7 : 0)))));
        if (w.render_bwd == 6) {
            const char *d = e;
            while (*d && (*d < '0' || *d > '9')) d++;
            w.render_bwd_mb = *d ? atoi(d) : 128;
            const char *pp = strchr(e, 'p');
            w.render_bwd_pix = pp ? atoi(pp + 1) : 0;
            const char *po = strchr(e, 'o');
            w.render_bwd_occ = po ? atoi(po + 1) : 5;
        }
        if (w.render_bwd == 7) {   // fine[<entries per round>]: render_bwd_rgn.hip
            const char *d = e;
            while (*d && (*d < '0' || *d > '9')) d++;
            w.render_bwd_mb = *d ? atoi(d) : 128;
            w.render_bwd_pix = strchr(e, 'q') ? 9 : (strchr(e, 's') ? 8 : 0);
        }
        if (w.render_bwd == 3) {
            const char *d = e;
            while (*d && (*d < '0' || *d > '9')) d++;
            w.render_bwd_mb = (*d ? atoi(d) : 256) + (strncmp(e, "scana", 5) == 0 ? 1000 : 0);
        }
    }
    if ((e = env("DAS3R_BWD_REDUCE"))) { w.bwd_reduce_set = true; w.bwd_reduce_shfl = e[0] == 's'; }
    w.tickets = -1;
    if ((e = env("DAS3R_TICKETS"))) w.tickets = e[0] == 'a' ? 0 : (e[0] == 'n' ? 1 << 30 : ((e[0] >= '1' && e[0] <= '9') ? atoi(e) : -1));
    w.deterministic = (e = env("DAS3R_DETERMINISTIC")) && e[0] != '0';
    w.bwd_buckets = -1;
    if ((e = env("DAS3R_BWD_BUCKETS"))) w.bwd_buckets = atoi(e);
    w.tile_strip = 8;
    w.tile_chunk = -1;
    if ((e = env("DAS3R_TILE_CHUNK"))) { const int v = atoi(e); w.tile_chunk = (v >= 0 && v <= 64 && (v & (v - 1)) == 0) ? v : -1; }
    if ((e = env("DAS3R_TILE_STRIP"))) w.tile_strip = std::max(0, std::min(63, atoi(e)));   // (7-bit field of render_common.h pack_tiles, kept below its sign bit)
    w.inject_fault = g_inject_fault;   // (not an environment variable: das3r_debug_inject_fault)
    w.mutate = g_mutate;               // (likewise: das3r_debug_mutate)
    g_sw = w;
    __atomic_store_n(&g_sw_loaded, true, __ATOMIC_RELEASE);
}
const Switches &switches() {
    if (!__atomic_load_n(&g_sw_loaded, __ATOMIC_ACQUIRE)) load_switches();
    return g_sw;
}

// Ticket-free chained kernels wait for lower-numbered workgroups, which is only safe while the whole grid is resident at once.
// The bound is taken from the device the call runs on (compute partitions have fewer CUs) and stays below what it holds
// (>= 3 workgroups per CU of the heaviest of these kernels).  DAS3R_TICKETS=always switches the short cut off.
// Round 6: "the whole grid is resident" presumes that the grid has the device to itself.  The library knows when it does
// not: every host thread that renders on a device registers there (per_device(), below; un-registered when the thread ends), and
// once a second thread has, every chained kernel of that device takes its tickets — whatever farm.run_jobs or the environment say
// (a DAS3R_TICKETS bound set by hand still wins: it is an experiment switch, and says so once on stderr).  Two PROCESSES on one GPU
// are not seen from here: tools/jobs_per_gpu.py's process mode sets DAS3R_TICKETS=always for its children.
constexpr int MAX_DEVICES = 64;
static int g_threads_on_device[MAX_DEVICES] = {};
struct ThreadRegistrar {
    bool on[MAX_DEVICES] = {};
    void add(int dev) {
        if (!on[dev]) { on[dev] = true; __atomic_fetch_add(&g_threads_on_device[dev], 1, __ATOMIC_RELAXED); }
    }
    ~ThreadRegistrar() {
        for (int d = 0; d < MAX_DEVICES; d++)
            if (on[d]) __atomic_fetch_sub(&g_threads_on_device[d], 1, __ATOMIC_RELAXED);
    }
};
static thread_local ThreadRegistrar g_registrar;
bool grid_is_resident(int nblocks) {
    const int forced = switches().tickets;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return false;
    const bool shared = __atomic_load_n(&g_threads_on_device[dev], __ATOMIC_RELAXED) > 1;
    if (forced >= 0) {
        static bool warned = false;
        if (shared && forced > 0 && !__atomic_exchange_n(&warned, true, __ATOMIC_RELAXED))
            fprintf(stderr, "[das3r] DAS3R_TICKETS is set to a bound while several host threads render on device %d: ticket-free chained "
                            "kernels are only safe when a grid has the device to itself\n", dev);
        return nblocks <= forced;
    }
    if (shared) return false;
    static thread_local int cached_dev = -1, cached_limit = 0;
    if (dev != cached_dev) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
        cached_dev = dev;
        cached_limit = cus * 2;   // the heaviest chained kernels get 3 workgroups per CU
    }
    return nblocks <= cached_limit;
}

int sort_ipl_override() { return switches().sort_ipl; }
#ifdef DAS3R_EXPERIMENTS
bool use_onesweep() { return !switches().sort_classic; }
#else
bool use_onesweep() { return true; }   // the three-kernel radix passes are built with EXPERIMENTS=1 only
#endif
bool use_tight_rect() { return !switches().rect_upstream; }

void compute_layout(int P, int64_t I, int W, int H, Layout *L) {
    memset(L, 0, sizeof(*L));
    const size_t Pn = P > 0 ? (size_t)P : 1, In = I > 0 ? (size_t)I : 1;
    L->capacity = I;
    L->tiles_x = (W + TILE_X - 1) / TILE_X;
    L->tiles_y = (H + TILE_Y - 1) / TILE_Y;
    L->ntiles = L->tiles_x * L->tiles_y;
    L->tbits = tile_bits(L->ntiles);
    L->tile_passes = (L->tbits + 7) / 8;
    L->dbits = 0;
    L->kbits = L->tbits;
    L->kshift = 0;
    L->part_passes = L->tile_passes;
    L->chunksP = sort_num_chunks(P);
    L->chunksI = sort_num_chunks(I);
    size_t o = 0;
    auto take = [&](size_t b) { size_t r = o; o += align_up(b); return r; };
    // geom
    L->g_keyA = take(4 * Pn);
    L->g_keyB = take(4 * Pn);
    L->g_valA = take(4 * Pn);
    L->g_valB = take(4 * Pn);
    L->pub.sorted_idx = L->g_valA;  // 4 passes: A -> B -> A -> B -> A
    L->pub.depth_key = L->g_keyA;   // (clobbered by the sort; kept for the layout struct only)
    L->pub.xy = take(16 * SPLAT_REC * Pn);      // one 64-byte record per Gaussian: xyh | conic+opacity | rgb+depth | pad
    L->pub.conic_opacity = L->pub.xy + 16;
    L->pub.rgbd = L->pub.xy + 32;
    L->pub.splat_stride = 16 * SPLAT_REC;
    L->pub.clamped = take(Pn);
    L->pub.tiles_touched = take(4 * Pn);
    L->pub.offsets = take(4 * Pn);
    L->g_hist = take(4 * (size_t)RADIX_SIZE * (size_t)(L->chunksP > 0 ? L->chunksP : 1));
    L->g_totals = take(4 * RADIX_SIZE);
    L->g_blocksums = take(4 * (size_t)div_up((int64_t)Pn, 4096));
    L->g_count = take(256);
    L->g_off_by_gid = take(4 * Pn);
    L->g_rect = take(4 * Pn);
    L->g_dhist = take(4 * 256);
    L->g_ghist = take(4 * 4 * RADIX_SIZE);
    L->g_ticket = take(256);
    L->g_status = take(onesweep_status_bytes((int64_t)Pn, 4));
    L->g_scan_status = take(scan_status_bytes((int)Pn));
    L->g_ctrl_bytes = o - L->g_ghist;
    L->pub.geom_bytes = o;
    // binning
    o = 0;
    L->b_keyA = take(4 * In);
    L->b_keyB = take(4 * In);
    L->b_valA = take(4 * In);
    L->b_valB = take(4 * In);
    L->b_hist = take(4 * (size_t)RADIX_SIZE * (size_t)(L->chunksI > 0 ? L->chunksI : 1));
    L->b_totals = take(4 * RADIX_SIZE);
    L->b_gid_of = take(4 * In);
    L->b_slot = take(4 * In);
    L->b_e2 = take(4 * In);
    L->b_ghist = take(4 * 4 * RADIX_SIZE);
    L->b_ticket = take(256);
    L->b_status = take(onesweep_status_bytes((int64_t)In, L->tile_passes + 1));   // (+ 1: the segmented path may take one more pass)
    L->b_ctrl_bytes = o - L->b_ghist;
    L->b_ckpt = take(((size_t)In / BUCKET + (size_t)(L->ntiles > 0 ? L->ntiles : 1) + 2) * 256 * 16);
    L->pub.binning_bytes = o;
    L->pub.point_list = (L->tile_passes & 1) ? L->b_valB : L->b_valA;
    // img
    o = 0;
    const size_t npix = (size_t)W * (size_t)H;
    L->pub.final_T = take(4 * (npix > 0 ? npix : 1));
    L->pub.n_contrib = take(4 * (npix > 0 ? npix : 1));
    L->pub.ranges = take(8 * (size_t)(L->ntiles > 0 ? L->ntiles : 1));
    L->i_order = take(4 * (size_t)(L->ntiles > 0 ? L->ntiles : 1));
    L->pub.img_bytes = o;
}

// Depth histograms of the segmented path (segkey.h), round 6: two 256-word slots per (host thread, device, stream), used in turn.  A forward's
// preprocess kernel accumulates into one and ZEROES THE OTHER for the next forward of the stream — the memset that used to precede every
// forward (a launch of its own: 5 us of every DAS3R-shaped iteration) is gone.  Launches of one stream are ordered, so the slot a forward
// zeroes is read by nobody (the previous forward's emission has finished), and the slot it reads stays as it is until the NEXT forward's
// preprocess kernel, however often its own emission is redone.  Should a slot ever be dirty — an aborted forward — the buckets come out
// less even for one forward and exact all the same: any histogram gives a monotone map, the same one in every workgroup.
static bool dhist_slots(hipStream_t s, uint32_t **use, uint32_t **next) {
    struct Slot { int dev; hipStream_t stream; uint32_t *buf; uint32_t turn; };
    static thread_local std::vector<Slot> slots;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    Slot *e = nullptr;
    for (auto &c : slots)
        if (c.dev == dev && c.stream == s) e = &c;
    if (!e) {
        uint32_t *buf = nullptr;
        if (hipMalloc((void **)&buf, 2 * 256 * sizeof(uint32_t)) != hipSuccess) return false;
        if (hipMemsetAsync(buf, 0, 2 * 256 * sizeof(uint32_t), s) != hipSuccess) return false;   // (on the stream whose kernels use it: ordered)
        slots.push_back(Slot{dev, s, buf, 0u});
        e = &slots.back();
    }
    e->turn++;
    *use = e->buf + 256 * (e->turn & 1u);
    *next = e->buf + 256 * ((e->turn + 1u) & 1u);
    return true;
}

static int validate(const das3r_raster_args *a, const das3r_raster_in *in) {
    if (!a || !in) { set_error("null args"); return DAS3R_ERR_INVALID_ARG; }
    if (a->P < 0 || a->image_width <= 0 || a->image_height <= 0) { set_error("bad extents P=%d W=%d H=%d", a->P, a->image_width, a->image_height); return DAS3R_ERR_INVALID_ARG; }
    if ((int64_t)((a->image_width + TILE_X - 1) / TILE_X) * ((a->image_height + TILE_Y - 1) / TILE_Y) >= (1ll << 22)) {   // render_common.h pack_tiles
        set_error("image of %d x %d pixels has 2^22 tiles or more", a->image_width, a->image_height);
        return DAS3R_ERR_INVALID_ARG;
    }
    if (a->P == 0) return DAS3R_OK;
    if (!a->bg || !a->viewmatrix || !a->projmatrix || !a->campos) { set_error("bg/viewmatrix/projmatrix/campos must be device pointers"); return DAS3R_ERR_INVALID_ARG; }
    if (in->pre) {   // ABI 14: raw parameters + pose; the camera-frame tensors are not looked at
        const das3r_pretransform *p = in->pre;
        if (!p->xyz || !p->rot || !p->scaling || !p->opacity_raw || !p->conf_flat || !p->R || !p->t || !p->Lq) { set_error("das3r_raster_in.pre: xyz / rot / scaling / opacity_raw / conf_flat / R / t / Lq are required"); return DAS3R_ERR_INVALID_ARG; }
        if (in->cov3D_precomp) { set_error("das3r_raster_in.pre excludes cov3D_precomp"); return DAS3R_ERR_INVALID_ARG; }
    } else if (!in->means3D || !in->opacities) { set_error("means3D/opacities are required"); return DAS3R_ERR_INVALID_ARG; }
    if ((in->shs != nullptr) == (in->colors_precomp != nullptr)) { set_error("Please provide excatly one of either SHs or precomputed colors!"); return DAS3R_ERR_INVALID_ARG; }
    const bool sr = in->pre != nullptr || (in->scales != nullptr && in->rotations != nullptr);
    if ((!in->pre && (in->scales != nullptr) != (in->rotations != nullptr)) || sr == (in->cov3D_precomp != nullptr)) {
        set_error("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!");
        return DAS3R_ERR_INVALID_ARG;
    }
    if (in->shs) {
        if (a->sh_degree < 0 || a->sh_degree > 3) { set_error("sh_degree must be 0..3 (got %d)", a->sh_degree); return DAS3R_ERR_INVALID_ARG; }
        if (a->M < (a->sh_degree + 1) * (a->sh_degree + 1) || a->M > 16) { set_error("M=%d incompatible with sh_degree=%d (need (D+1)^2 <= M <= 16)", a->M, a->sh_degree); return DAS3R_ERR_INVALID_ARG; }
    }
    if (a->tanfovx <= 0.f || a->tanfovy <= 0.f) { set_error("tanfov must be positive"); return DAS3R_ERR_INVALID_ARG; }
    return DAS3R_OK;
}

}  // namespace das3r

using namespace das3r;

extern "C" int das3r_abi_version(void) { return DAS3R_ABI_VERSION; }
extern "C" void das3r_reload_switches(void) { load_switches(); }
extern "C" void das3r_debug_inject_fault(uint32_t bits) {
    g_inject_fault = (int)bits;
    load_switches();
}
extern "C" void das3r_debug_mutate(uint32_t what) {
    g_mutate = (int)what;
    load_switches();
}
extern "C" const char *das3r_last_error(void) { return g_err; }

extern "C" int das3r_raster_get_layout(int32_t P, int64_t num_rendered, int32_t W, int32_t H, das3r_raster_layout *out) {
    if (!out || P < 0 || num_rendered < 0 || W <= 0 || H <= 0) { set_error("das3r_raster_get_layout: invalid argument"); return DAS3R_ERR_INVALID_ARG; }
    Layout L;
    compute_layout(P, num_rendered, W, H, &L);
    *out = L.pub;
    return DAS3R_OK;
}

// Host mailbox: a pinned, device-visible block per host thread and device.  The scan kernel stores {count, flags} and then the
// tag of the call (system-scope release) into words 0..2; word 10 is raised by a compositing kernel that met a tile list too
// long for its LDS sort; the last binning kernel of a forward stores {self-check word, tag} into one of CHECK_SLOTS two-word
// slots from word 16 on (slot = tag % CHECK_SLOTS).  The host polls the tags — no D2H copy kernel, no event, no parked thread
// (hipEventSynchronize's wake-up alone cost ~100 us per forward, a third of a 100 k-splat step).
struct Mailbox {
    volatile uint32_t *host = nullptr;
    uint32_t *dev = nullptr;
    uint32_t seq = 0;
};
// Everything the library remembers between calls, per host thread and per device (a thread that renders on two GPUs gets two
// of these; nothing is shared between threads: any number of them may render concurrently, each on its own stream).
constexpr uint32_t CHECK_SLOTS = 16, CHECK_WORD0 = 16, MAILBOX_BYTES = 4 * (CHECK_WORD0 + 2 * CHECK_SLOTS);
constexpr uint32_t CROWDED16 = 23u * 16u;   // (api.hip bin_and_render)
struct Verdict { int P, W, H; int64_t last_I, peak_I; int radix_left, backoff; uint32_t gen; int seg_extra; bool last_seg; bool fine; uint32_t forwards; uint32_t clean; uint32_t longest; };
struct PerDevice {
    Mailbox mb;
    uint32_t pending[CHECK_SLOTS] = {};             // per slot: tag of the forward whose self-check word has not been examined yet
    unsigned long long *arrive_ring = nullptr;      // self re-arming arrival words of the preprocess kernel's count reduction
    Verdict verdict = {0, 0, 0, -1, 0, 0, 64, 0};  // what the last forward of the current shape (P, W, H) taught us
    bool pending_fine = false, pending_valid = false;   // das3r_raster_learning(set): handed to the next shape this thread meets
    uint32_t pending_forwards = 0;
    char *emit_ring = nullptr;                      // control words of the emission fused into the preprocess kernel
    bool emit_ring_dirty = false;
    uint32_t emit_last_tag = 0;
};
static int per_device(PerDevice **out) {
    static thread_local PerDevice state[MAX_DEVICES];
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= MAX_DEVICES) { set_error("device ordinal %d not supported", dev); return DAS3R_ERR_INVALID_ARG; }
    PerDevice &T = state[dev];
    g_registrar.add(dev);   // (grid_is_resident: a second rendering thread on the device switches the ticket-free short cut off)
    if (!T.mb.host) {
        uint32_t *h = nullptr;
        HIP_TRY(hipHostMalloc((void **)&h, MAILBOX_BYTES, hipHostMallocMapped));
        memset(h, 0, MAILBOX_BYTES);
        HIP_TRY(hipHostGetDevicePointer((void **)&T.mb.dev, h, 0));
        T.mb.host = h;
    }
    *out = &T;
    return DAS3R_OK;
}
// spin until word `idx` carries `tag` (bounded: falls back to a stream synchronise, then gives up loudly)
static int mailbox_wait(Mailbox *mb, int idx, uint32_t tag, hipStream_t s) {
    for (long spins = 0; spins < 200000000L; spins++) {
        if (__atomic_load_n(&mb->host[idx], __ATOMIC_ACQUIRE) == tag) return DAS3R_OK;
        __builtin_ia32_pause();
        if ((spins & 0xFFFFF) == 0xFFFFF && hipStreamQuery(s) != hipErrorNotReady) break;   // stream drained (or failed): stop spinning
    }
    HIP_TRY(hipStreamSynchronize(s));
    if (__atomic_load_n(&mb->host[idx], __ATOMIC_ACQUIRE) == tag) return DAS3R_OK;
    set_error("the device never delivered the result of this forward to the host mailbox");
    return DAS3R_ERR_HIP;
}

// Examine the self-check word a forward's last binning kernel left in `slot` = {flags, tag} of a host mailbox (bit 1 look-back
// timeout, 2 index out of range -> write suppressed, 8 counts do not add up to the histogram; 16 = a stalled look-back was rescued,
// granule.h: informational).  wait: spin (bounded) until the word of `tag` is there.  Returns 1 when the slot does not hold
// `tag`'s word (not there yet, or — after CHECK_SLOTS later forwards, each of which examines the slot before reusing it — gone).
static unsigned long long g_stats[4] = {0, 0, 0, 0};   // forwards, self-check words examined, rescued look-back polls, failed self-checks
static void stat_add(int i) { __atomic_fetch_add(&g_stats[i], 1ull, __ATOMIC_RELAXED); }
static int examine_check_slot(volatile uint32_t *slot, uint32_t tag, bool wait, hipStream_t s) {
    uint32_t seen = __atomic_load_n(&slot[1], __ATOMIC_ACQUIRE);
    if (seen != tag && wait) {
        for (long spins = 0; spins < 200000000L && seen != tag; spins++) {
            __builtin_ia32_pause();
            seen = __atomic_load_n(&slot[1], __ATOMIC_ACQUIRE);
            if ((spins & 0xFFFFF) == 0xFFFFF && hipStreamQuery(s) != hipErrorNotReady) break;   // stream drained (or failed): stop spinning
        }
        if (seen != tag) {
            HIP_TRY(hipStreamSynchronize(s));
            seen = __atomic_load_n(&slot[1], __ATOMIC_ACQUIRE);
        }
    }
    if (seen != tag) return 1;
    const uint32_t all_flags = slot[0], flags = all_flags & ~16u;
    if (all_flags & 0x80000000u) return DAS3R_OK;   // this forward's failure has been reported already (by the backward pass / das3r_raster_check)
    stat_add(1);
    if (all_flags & 16u) stat_add(2);
    if (flags) stat_add(3);
    if ((all_flags & 16u) && switches().verbose) fprintf(stderr, "das3r: a look-back poll needed the read-modify-write path\n");
    if (flags) {
        slot[0] = all_flags | 0x80000000u;   // reported once: the thread that made the forward will not report it again at its next call
        set_error("a forward's binning failed its self-check (flags 0x%x); its output is invalid", flags);
        return DAS3R_ERR_HIP;
    }
    return DAS3R_OK;
}

// The self-check of the forward that produced `saved`, for callers that want it BEFORE they consume the image (the forward
// itself returns as soon as everything is enqueued; das3r_raster_backward runs this first).  Waits, bounded, for the word.
extern "C" int das3r_raster_check(const das3r_raster_saved *saved, das3r_stream_t stream) {
    if (!saved) { set_error("das3r_raster_check: null saved state"); return DAS3R_ERR_INVALID_ARG; }
    if (!saved->check_word || !saved->check_tag) return DAS3R_OK;   // P == 0, nothing rendered, or a caller that dropped the ticket
    const int r = examine_check_slot((volatile uint32_t *)saved->check_word, saved->check_tag, true, (hipStream_t)stream);
    return r < 0 ? r : DAS3R_OK;
}

// ABI 13: what the calling thread's library state has LEARNT about shapes on the current device is forgotten (counts, binning path, forward
// kernel choice); the mailbox, rings and tickets stay.  See include/das3r_raster.h.
extern "C" int das3r_raster_forget_shapes(void) {
    PerDevice *T = nullptr;
    int r = per_device(&T);
    if (r) return r;
    const uint32_t gen = T->verdict.gen;   // (generation numbers keep counting: a word a forward of the old shape still has on its way names nobody)
    T->verdict = Verdict{0, 0, 0, -1, 0, 0, 64, gen};
    return DAS3R_OK;
}

// ABI 13: the part of the learnt state that shows in the last bit of a result — the forward kernel of the current shape and the position in its
// re-decision schedule.  set == 0: read into state[0 .. 1]; set != 0: forget the shapes and hand state[] to the next shape this thread meets.
extern "C" int das3r_raster_learning(int32_t set, uint32_t state[2]) {
    if (!state) { set_error("das3r_raster_learning: invalid argument"); return DAS3R_ERR_INVALID_ARG; }
    PerDevice *T = nullptr;
    int r = per_device(&T);
    if (r) return r;
    if (!set) {
        state[0] = T->verdict.fine ? 1u : 0u;
        state[1] = T->verdict.forwards;
        return DAS3R_OK;
    }
    const uint32_t gen = T->verdict.gen;
    T->verdict = Verdict{0, 0, 0, -1, 0, 0, 64, gen};
    T->pending_fine = state[0] != 0u;
    T->pending_forwards = state[1];
    T->pending_valid = true;
    return DAS3R_OK;
}

extern "C" void das3r_get_stats(uint64_t out[4]) {
    for (int i = 0; i < 4; i++) out[i] = (uint64_t)__atomic_load_n(&g_stats[i], __ATOMIC_RELAXED);
}

extern "C" int64_t das3r_raster_forward(const das3r_raster_args *a, const das3r_raster_in *in, const das3r_raster_out *out,
                                        das3r_alloc_fn alloc_geom, das3r_alloc_fn alloc_binning, das3r_alloc_fn alloc_img,
                                        void *user, das3r_raster_saved *saved, das3r_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    stat_add(0);
    int rc = validate(a, in);
    if (rc) return rc;
    if (!out || !out->out_color || (a->P > 0 && !out->radii) || !alloc_geom || !alloc_binning || !alloc_img || !saved) {
        set_error("das3r_raster_forward: null output/allocator");
        return DAS3R_ERR_INVALID_ARG;
    }
    const int P = a->P, W = a->image_width, H = a->image_height;
    Layout L;
    compute_layout(P, 0, W, H, &L);
    saved->geom = alloc_geom(user, L.pub.geom_bytes);
    saved->img = alloc_img(user, L.pub.img_bytes);
    saved->binning = nullptr;
    saved->num_rendered = 0;
    saved->capacity = 0;
    saved->check_word = nullptr;
    saved->check_tag = 0;
    saved->flags = 0;
    if (!saved->geom || !saved->img) { set_error("scratch allocation failed (geom %zu B, img %zu B)", L.pub.geom_bytes, L.pub.img_bytes); return DAS3R_ERR_ALLOC; }
    if (P == 0) {
        // upstream:rasterize_points.cu skips the rasterizer when P == 0: the image stays zero (background NOT applied)
        HIP_TRY(hipMemsetAsync(out->out_color, 0, sizeof(float) * 3 * (size_t)W * H, s));
        HIP_TRY(hipMemsetAsync(saved->img + L.pub.ranges, 0, 8 * (size_t)L.ntiles, s));
        saved->binning = alloc_binning(user, 256);
        return 0;
    }
    // Host <-> device hand-offs of one forward (all through the pinned mailbox, no copies, no events):
    //   * num_rendered (sum of tiles_touched) leaves with the preprocess kernel, five kernels before the device needs it, so
    //     the binning buffer is sized exactly and the host practically never waits (upstream blocks on a cudaMemcpy in the
    //     middle of every forward); on the local-order path nothing waits for it at all (speculative capacity, below);
    //   * the binning self-check word is delivered by the last binning kernel into a slot of the mailbox; saved->check_word /
    //     check_tag name it.  It is examined by das3r_raster_backward (before anything is launched) and das3r_raster_check, which
    //     wait for it, and — for callers that use neither — without waiting at the start of this thread's later forwards;
    //     debug mode waits for it before returning.
    PerDevice *T = nullptr;
    if ((rc = per_device(&T))) return rc;
    Mailbox *mb = &T->mb;
    auto check_slot = [&](uint32_t i, bool wait) -> int {
        if (!T->pending[i]) return DAS3R_OK;
        const int r = examine_check_slot(mb->host + CHECK_WORD0 + 2 * i, T->pending[i], wait, s);
        if (r == 1) return DAS3R_OK;   // not there yet: look again next time
        T->pending[i] = 0;
        return r;
    };
    for (uint32_t i = 0; i < CHECK_SLOTS; i++)
        if ((rc = check_slot(i, false))) return rc;
    uint32_t late_tag = 0;
    // per-thread ring of self re-arming arrival words for the count reduction of the preprocess kernel
    unsigned long long *&arrive_ring = T->arrive_ring;
    constexpr uint32_t ARRIVE_SLOTS = 16, ARRIVE_WORDS = 8 + 64 * 8;   // per call: top word + 64 sub-counters, one 64-byte line each
    if (!arrive_ring) {
        HIP_TRY(hipMalloc((void **)&arrive_ring, ARRIVE_SLOTS * ARRIVE_WORDS * sizeof(unsigned long long)));
        // on the CALLER'S stream, in front of the kernel that counts in it: hipMemset runs on the null stream, which a non-blocking stream
        // does not wait for — with a second job keeping the GPU busy (round 5: farm.run_jobs) the fill could land AFTER the first preprocess
        // kernel of this thread had started counting, the last arriver never saw its total, and the forward ended with "the device never
        // delivered the result of this forward to the host mailbox" (one first-forward in a few dozen two-job runs)
        HIP_TRY(hipMemsetAsync(arrive_ring, 0, ARRIVE_SLOTS * ARRIVE_WORDS * sizeof(unsigned long long), s));
        // ... and COMPLETE before this call goes on: the ring belongs to the thread, not to the stream — the same thread's next forward may
        // be issued on another non-blocking stream, which does not wait for this one.  Once per thread and device.
        HIP_TRY(hipStreamSynchronize(s));
    }
    // Depth order: scenes with short tile lists (the 100 k-splat 1080p benchmark averages 32 entries per tile) skip the global
    // depth sort — five of the eleven binning launches, each latency-bound at that size; the instances are emitted in index
    // order and the compositing kernel sorts every tile's list itself (common.h: LocalBin).  Chosen from the instance count
    // of the previous forward of the same shape, confirmed with this forward's count; a list that outgrows LDS is still sorted
    // correctly (slowly) and sends the next forwards back to the global sort.
    Verdict &verdict = T->verdict;
    constexpr int64_t LOCAL_AVG = 384;   // mean list length up to which the local order wins (measured: 1 M splats at 1080p, mean 320: -4 %)
    const uint32_t too_long = mb->host[10];   // != 0: a forward of the shape with that generation number met a list too long for LDS
    if (too_long) mb->host[10] = 0;
    const uint32_t want_bits = mb->host[11];  // != 0: the segmented path of that generation sorted a segment long enough to want more bucket bits
    if (want_bits) mb->host[11] = 0;
    if (verdict.P != P || verdict.W != W || verdict.H != H) {
        const uint32_t gen = verdict.gen + 1u ? verdict.gen + 1u : 1u;
        verdict = Verdict{P, W, H, -1, 0, 0, 64, gen};
        if (T->pending_valid) {   // a resumed job: the forward-kernel choice and its schedule continue where the checkpoint left them
            verdict.fine = T->pending_fine;
            verdict.forwards = T->pending_forwards;
            T->pending_valid = false;
        }
    }
    // Skewed tile lists -> the forward kernel with four workgroups per tile (render_regions.hip; below, bin_and_render): decided from the
    // tile ranges of the shape's FIRST forward and of every 512th after it — a one-workgroup kernel behind the binning leaves the longest
    // list in the mailbox and the host waits for it (the first forward of a shape waits for its count anyway; afterwards one short wait
    // per 512 forwards).  Which kernel runs in which forward thus depends on the data alone, never on timing: the two kernels round a
    // pixel's T differently in the last bit, and a job must end bit-identical however it was scheduled.
    const bool decide_fine = (verdict.forwards++ & 511u) == 0u;
    // Every decision below starts a new GENERATION of the shape: the words the kernels raise (mailbox 10 / 11) name the generation they were
    // launched under, and the host runs one or two forwards ahead of the device — the forward enqueued BEFORE a decision was taken raises
    // the same word again a moment later, and answered a second time it turned "one more partition pass" straight into "global sort for
    // 64 forwards" (round 6: seen as a smooth-depth train step of 1.29 ms instead of 1.18 on some boxes, its first 64 timed steps on the
    // global sort; which box depended on how far ahead its host ran).
    auto next_gen = [&]() { verdict.gen = verdict.gen + 1u ? verdict.gen + 1u : 1u; };
    if (verdict.last_I < 0) {   // (a shape's first forward: nothing to learn from yet)
    } else if (too_long == verdict.gen && !(verdict.last_seg && verdict.seg_extra == 0 && L.tile_passes < 3 && seg_dbits(L, L.tile_passes + 1) > 0)) {
        // global sort for a while; longer every time it happens AGAIN SOON.  Round 6: a failure that comes after 256 or more clean forwards
        // on the fast path is an occasional one — a view of 45 that looks at a wall head-on (DAVIS-shaped job: one forward in ~ 350) — and
        // starts from the shortest stint again: the doubling never forgot, and by iteration 4000 such a job had spent 2 447 of its
        // iterations on the global sort (binning 0.64 ms against 0.47), tools/probes/job_binning_paths.py
        if (verdict.clean >= 256u) verdict.backoff = 64;
        verdict.radix_left = verdict.backoff;
        if (verdict.backoff < 4096) verdict.backoff *= 2;
        verdict.clean = 0;
        next_gen();
    } else if (too_long == verdict.gen || want_bits == verdict.gen) {
        verdict.seg_extra = 1;                  // the segmented path with one more partition pass of bucket bits from now on (this shape);
                                                // a segment that is too long even then sends the shape back to the global sort (above)
        next_gen();
    }
    const int forced = switches().binning;   // DAS3R_BINNING=local | radix | seg: force one (diagnostics, tests)
    bool local = use_onesweep() && (forced == 1 || (forced == 0 && verdict.radix_left == 0 && verdict.last_I <= LOCAL_AVG * L.ntiles));
    // Long lists (round 4): the segmented path — no global depth sort; the tile partition's passes carry a depth bucket in the key bits
    // the tile ids leave free, and every (tile, bucket) segment is sorted inside LDS (segkey.h, segsort.hip).  Needs free key bits
    // and segments that stay short on average; a segment that did not fit in LDS sends the shape back to the global sort for a
    // while, like a list too long for the local order does (the same mailbox word and back-off).
    // The buckets are global: they split a tile's list well when its depths are spread (random-depth benchmarks) and badly when a tile
    // sees a thin depth band of a wide scene (every real scene: its Gaussians lie on surfaces).  A forward that had to rank a long
    // segment says so (mailbox word 11), and the shape takes ONE MORE partition pass from then on: eight more bucket bits — a band
    // that held 3 % of the scene's instances in 4 of 128 buckets is cut into a thousand (r4, coherent-depth 5 M-splat scene:
    // segment sort 1.08 ms with two passes, see DESIGN.md with three).
    constexpr int64_t SEG_AVG = 384;         // mean entries per segment up to which it is taken
    const int seg_passes = L.tile_passes + ((forced == 3 || (forced == 0 && verdict.seg_extra)) ? 1 : 0);
    const int seg_bits = (use_onesweep() && seg_passes <= 3) ? seg_dbits(L, seg_passes) : 0;
    bool seg = !local && seg_bits > 0 && (forced >= 2 || (forced == 0 && verdict.radix_left == 0 && verdict.last_I >= 0 &&
                                                           (verdict.last_I >> std::min(seg_bits, 20)) <= SEG_AVG * L.ntiles));
    const uint32_t *seg_dhist = nullptr;   // where this forward's depth histogram is (set with the preprocess launch, below)
    auto apply_seg = [&]() {   // (compute_layout starts every layout without buckets)
        L.dhist_ptr = seg_dhist;
        if (seg) {
            L.dbits = seg_bits;
            L.kbits = L.tbits + seg_bits;
            L.kshift = std::min(16, 32 - L.kbits);
            L.part_passes = seg_passes;
        }
    };
    verdict.last_seg = seg;
    if (local || seg) verdict.clean++;   // (a failure of this forward is reported to the next one: reset there)
    if (forced == 0 && verdict.radix_left > 0) verdict.radix_left--;
    const uint32_t count_tag = ++mb->seq ? mb->seq : ++mb->seq;
    unsigned long long *arrive = arrive_ring + (size_t)(count_tag % ARRIVE_SLOTS) * ARRIVE_WORDS;
    // everything behind the count: scan + emission, partition, tile ranges, compositing (L laid out for `cap`, buffer allocated)
    auto bin_and_render = [&](int64_t cap, bool local_order, bool ctrl_zeroed, uint32_t *count_out = nullptr, uint32_t count_out_tag = 0,
                              uint32_t *emit_slot = nullptr /*the preprocess kernel has emitted the instances already*/) -> int {
        int r;
        const bool fused_scan = use_onesweep();   // scan + emission in one kernel; the classic path scans, then emits
        const bool seg = L.dbits > 0;             // segmented path: index-order emission with depth buckets, lists sorted by segment_sort_kernel
        if (!emit_slot) {
            if (fused_scan) {
                if ((r = launch_binning_scan_emit(P, cap, out->radii, saved->geom, saved->binning, L, ctrl_zeroed, count_out, count_out_tag, a->debug != 0,
                                                  s, local_order || seg))) return r;
            } else if ((r = launch_scan(P, saved->geom, L, nullptr, 0, a->debug != 0, s))) return r;
        }
        uint32_t *host_late = nullptr;
        if (cap > 0) {
            late_tag = ++mb->seq ? mb->seq : ++mb->seq;
            const uint32_t slot = late_tag % CHECK_SLOTS;
            if ((r = check_slot(slot, true))) return r;   // (the forward that used the slot CHECK_SLOTS forwards ago)
            T->pending[slot] = late_tag;
            host_late = mb->dev + CHECK_WORD0 + 2 * slot;
            saved->check_word = (void *)(mb->host + CHECK_WORD0 + 2 * slot);
            saved->check_tag = late_tag;
        }
        uint32_t *dead_keys = nullptr;
        if ((r = launch_binning(P, cap, W, H, out->radii, saved->geom, saved->binning, saved->img, L, fused_scan, host_late, late_tag,
                                a->debug != 0, s, &dead_keys, emit_slot, mb->dev + 10, verdict.gen))) return r;
        LocalBin lb = {(float4 *)(saved->binning + L.b_ckpt), nullptr, nullptr, nullptr, nullptr, verdict.gen, (uint32_t)(P - 1), (uint32_t)cap};
        if (decide_fine && cap > 0 && use_quad_lanes(L, lb)) {   // (shapes the one-workgroup-per-tile kernel would take: few tiles, long lists)
            const uint32_t tag = ++mb->seq ? mb->seq : ++mb->seq;
            if ((r = launch_list_skew(saved->img, saved->binning, saved->geom, L, (uint32_t)cap, (uint32_t)(P - 1), mb->dev + 13, tag, a->debug != 0, s))) return r;
            if ((r = mailbox_wait(mb, 14, tag, s))) return r;
            const int64_t longest = (int64_t)mb->host[13], mean = (verdict.last_I > 0 ? verdict.last_I : cap) / std::max(L.ntiles, 1);   // (the last forward's count; the capacity on a shape's first)
            verdict.longest = (uint32_t)longest;   // (what the backward's grid is sized by until the next look: below)
            const uint32_t crowd16 = mb->host[15];   // of 64 consecutive list entries, those in the tile's fullest quadrant, x 16 (render_regions.hip list_skew_kernel)
            // skewed: 1.8 x the mean list, the measured crossover of the forward kernels (ledger (bd)); crowded: a stretch of a list sits in part of
            // its tile (random depths: 20 of 64 in the fullest quadrant) — there the 2x2-region kernels win both ways whatever the skew (ledger (be))
            verdict.fine = (longest > mean + (mean * 4) / 5 && longest >= 4096) || crowd16 >= CROWDED16;
            if (switches().verbose) fprintf(stderr, "[das3r] tile lists: longest %lld, mean %lld, %.1f of 64 consecutive entries in one quadrant -> %s kernels\n", (long long)longest,
                                            (long long)mean, crowd16 / 16.0, verdict.fine ? "2x2-region" : "block");
        }
        lb.prefer_regions = verdict.fine;
        lb.tile_order = nullptr;
        if (verdict.fine && cap > 0 && use_quad_lanes(L, lb) && L.ntiles <= 1024 && !switches().tile_lpt_off && (switches().render_fwd == 0 || switches().render_fwd == 5)) {
            // four workgroups per tile are 3 - 4 generations of workgroups on the chip, and a tile whose list is four times the mean (every
            // real sequence has them) that starts in the last generation adds its whole chain to the kernel: longest lists first
            if ((r = launch_tile_lpt(saved->img, L, (uint32_t)cap, a->debug != 0, s))) return r;
            lb.tile_order = (const uint32_t *)(saved->img + L.i_order);
        }
        if (verdict.fine && cap > 0 && use_quad_lanes(L, lb)) {   // (the backward pass of this forward: render_bwd.hip)
            // bits 8 - 15: buckets (common.h BUCKET = 1024 list positions) of the shape's longest tile list as last measured, + 2 of headroom — the
            // bucket-parallel backward launches that many workgroups per tile instead of the AVERAGE list's (a tile of four times the mean then
            // takes four buckets per workgroup, back to back, and the kernel ends with them: self-consistent job 0.409 -> 0.313 ms)
            const uint32_t hint = std::min<uint32_t>(63u, verdict.longest / (uint32_t)BUCKET + 2u);
            saved->flags |= 1u | (hint << 8);
        }
        if (local_order && cap > 0) {
            lb.point_list = (uint32_t *)(saved->binning + L.pub.point_list);
            lb.slot_list = (uint32_t *)(saved->binning + L.b_slot);
            lb.keys = dead_keys;
            lb.host_flag = mb->dev + 10;
        }
        return launch_render_forward(a, in->colors_precomp, out->out_color, saved->geom, saved->binning, saved->img, L, lb, s);
    };
    // prefiltered: the caller's promise that no point fails the near-plane cull (upstream traps when one does: auxiliary.h in_frustum);
    // the preprocess kernel leaves this call's tag in mailbox word 12 when one did
    auto prefiltered_ok = [&]() -> int {
        if (!a->prefiltered || mb->host[12] != count_tag) return DAS3R_OK;
        set_error("Point is filtered although prefiltered is set. This shouldn't happen!");
        return DAS3R_ERR_INVALID_ARG;
    };
    int64_t I, cap;
    // Local order + a previous forward of the same shape: nothing else could be enqueued while the count is on its way, so
    // the binning buffer is laid out for that forward's count + 25 % and the whole forward is enqueued before the host looks
    // at the mailbox (the kernels clamp to the capacity).  Should the scene have grown past it, the binning and the
    // compositing are redone with the exact size.  DAS3R_CAPACITY=exact switches the speculation off.
    if ((local || seg) && verdict.last_I >= 0 && a->capacity_hint != -1 && !switches().capacity_exact) {   // (round 4: the segmented path too)
        // headroom: 25 % over the last count, 5 % over the (slowly forgotten) largest one — a camera that moves between views
        // of different density overflows rarely
        cap = std::max(verdict.last_I + verdict.last_I / 4, verdict.peak_I + verdict.peak_I / 20) + 4096;
        if (cap > (int64_t)0x7FFFFF00) cap = (int64_t)0x7FFFFF00;
        compute_layout(P, cap, W, H, &L);
        apply_seg();
        saved->binning = alloc_binning(user, L.pub.binning_bytes);
        if (!saved->binning) { set_error("scratch allocation failed (binning %zu B)", L.pub.binning_bytes); return DAS3R_ERR_ALLOC; }
        // The count is not needed before the end of this call: the preprocess kernel skips its count reduction and the scan
        // delivers it.  When the preprocess grid is resident as a whole (P <= 196 k) the scan and the emission run inside the
        // preprocess kernel itself (EmitArgs, common.h) and the separate scan + emit kernel is not launched at all; their
        // control words live in a library-owned ring slot per forward, zero at rest.  DAS3R_FUSED_EMIT=0 switches that off.
        char *&emit_ring = T->emit_ring;
        bool &emit_ring_dirty = T->emit_ring_dirty;
        uint32_t &emit_last_tag = T->emit_last_tag;
        constexpr size_t EMIT_SLOT_BYTES = sizeof(uint32_t) * EMIT_SLOT_WORDS + sizeof(unsigned long long) * EMIT_STATUS_GRANULES;
        const bool fused_emit = local && grid_is_resident(div_up(P, 256)) && !switches().fused_emit_off;
        uint32_t *emit_slot = nullptr;
        if (fused_emit) {
            if (!emit_ring) {
                HIP_TRY(hipMalloc((void **)&emit_ring, ARRIVE_SLOTS * EMIT_SLOT_BYTES));
                emit_ring_dirty = true;
            }
            if (emit_ring_dirty || count_tag < emit_last_tag) {   // first use, an aborted forward, or the tags wrapped around
                HIP_TRY(hipMemsetAsync(emit_ring, 0, ARRIVE_SLOTS * EMIT_SLOT_BYTES, s));
                emit_ring_dirty = false;
            }
            emit_last_tag = count_tag;
            emit_slot = (uint32_t *)(emit_ring + (size_t)(count_tag % ARRIVE_SLOTS) * EMIT_SLOT_BYTES);
            const EmitArgs em = {(unsigned long long *)(emit_slot + EMIT_SLOT_WORDS), count_tag, emit_slot + 64, emit_slot,
                                 (uint32_t *)(saved->binning + L.b_keyA), (uint32_t *)(saved->binning + L.b_gid_of),
                                 (uint32_t *)(saved->geom + L.g_off_by_gid), (uint32_t)cap, L.tbits, (uint32_t *)(saved->geom + L.g_count), mb->dev};
            emit_ring_dirty = true;   // until the last binning kernel of this forward is enqueued (it re-arms the slot)
            if ((rc = launch_preprocess(a, in, out->radii, saved->geom, saved->img, saved->binning + L.b_ghist, L.b_ctrl_bytes, L, nullptr, mb->dev,
                                        count_tag, s, &em))) return rc;
            if ((rc = bin_and_render(cap, true, true, nullptr, 0, emit_slot))) return rc;
            emit_ring_dirty = false;
        } else {
            uint32_t *dhist = nullptr, *dhist_next = nullptr;
            if (seg && !dhist_slots(s, &dhist, &dhist_next)) {   // the depth histogram of this forward (segkey.h): filled by the preprocess kernel, read by the emission
                dhist = (uint32_t *)(saved->geom + L.g_dhist);   // (no slot: in the geometry buffer, zeroed by a memset, as before round 6)
                HIP_TRY(hipMemsetAsync(dhist, 0, 4 * 256, s));
            }
            seg_dhist = dhist;
            apply_seg();
            if ((rc = launch_preprocess(a, in, out->radii, saved->geom, saved->img, saved->binning + L.b_ghist, L.b_ctrl_bytes, L, nullptr, mb->dev,
                                        count_tag, s, nullptr, dhist, dhist_next))) return rc;
            if ((rc = bin_and_render(cap, local, true, mb->dev, count_tag))) return rc;
        }
        if ((rc = mailbox_wait(mb, 2, count_tag, s))) return rc;
        if ((rc = prefiltered_ok())) return rc;
        I = (int64_t)mb->host[0];
        if (I > (int64_t)0x7FFFFF00) { set_error("num_rendered %lld exceeds 2^31", (long long)I); return DAS3R_ERR_OVERFLOW; }
        verdict.last_I = I;
        verdict.peak_I = std::max(I, verdict.peak_I - verdict.peak_I / 1024);
        if (I > cap) {
            cap = I;
            compute_layout(P, cap, W, H, &L);
            saved->binning = alloc_binning(user, L.pub.binning_bytes);
            if (!saved->binning) { set_error("scratch allocation failed (binning %zu B)", L.pub.binning_bytes); return DAS3R_ERR_ALLOC; }
            apply_seg();
            HIP_TRY(hipMemsetAsync(saved->geom + L.g_ghist, 0, L.g_ctrl_bytes, s));          // tickets and look-back words of the scan
            HIP_TRY(hipMemsetAsync(saved->img + L.pub.ranges, 0, 8 * (size_t)L.ntiles, s));
            if ((rc = bin_and_render(cap, local, false))) return rc;
        }
    } else {
        uint32_t *dhist = nullptr, *dhist_next = nullptr;
        if (seg && !dhist_slots(s, &dhist, &dhist_next)) {   // the depth histogram of this forward (segkey.h): filled by the preprocess kernel, read by the emission
            dhist = (uint32_t *)(saved->geom + L.g_dhist);
            HIP_TRY(hipMemsetAsync(dhist, 0, 4 * 256, s));
        }
        seg_dhist = dhist;
        if ((rc = launch_preprocess(a, in, out->radii, saved->geom, saved->img, nullptr, 0, L, arrive, mb->dev, count_tag, s, nullptr, dhist, dhist_next))) return rc;
        if (!local && !seg && (rc = launch_depth_sort(P, saved->geom, L, 0, nullptr, 0, a->debug != 0, s))) return rc;
        if ((rc = mailbox_wait(mb, 2, count_tag, s))) return rc;    // usually there already: preprocess finished long ago
        if ((rc = prefiltered_ok())) return rc;
        I = cap = (int64_t)mb->host[0];
        if (cap > (int64_t)0x7FFFFF00) { set_error("num_rendered %lld exceeds 2^31", (long long)cap); return DAS3R_ERR_OVERFLOW; }
        verdict.last_I = I;
        verdict.peak_I = std::max(I, verdict.peak_I - verdict.peak_I / 1024);
        if (local && forced == 0 && I > LOCAL_AVG * L.ntiles) {   // the scene grew: global sort after all
            local = false;
            if ((rc = launch_depth_sort(P, saved->geom, L, 0, nullptr, 0, a->debug != 0, s))) return rc;
        }
        compute_layout(P, cap, W, H, &L);
        apply_seg();
        saved->binning = alloc_binning(user, L.pub.binning_bytes);
        if (!saved->binning) { set_error("scratch allocation failed (binning %zu B)", L.pub.binning_bytes); return DAS3R_ERR_ALLOC; }
        if (!local && !seg && (rc = launch_depth_sort(P, saved->geom, L, 1, saved->binning + L.b_ghist, L.b_ctrl_bytes, a->debug != 0, s))) return rc;
        if ((rc = bin_and_render(cap, local, !local && !seg))) return rc;
    }
    if (a->debug && late_tag && (rc = check_slot(late_tag % CHECK_SLOTS, true))) return rc;   // debug: report this forward's self-check word right away
    saved->num_rendered = I;
    saved->capacity = cap;
    return I;
}

extern "C" int das3r_raster_backward(const das3r_raster_args *a, const das3r_raster_in *in, const das3r_raster_saved *saved,
                                     const float *dL_dpix, const das3r_raster_grads *g, das3r_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    int rc = validate(a, in);
    if (rc) return rc;
    const int P = a->P;
    if (P == 0) return DAS3R_OK;
    const bool chained = g && g->chain;   // (ABI 14: dL/d(camera-frame means, opacities, scales, rotations) are then not produced — may be NULL)
    if (!saved || !saved->geom || !saved->img || !dL_dpix || !g || !g->dL_dmeans2D || (!chained && (!g->dL_dopacities || !g->dL_dmeans3D)) ||
        (saved && saved->num_rendered > 0 && !g->scratch)) {
        set_error("das3r_raster_backward: null saved state / gradient buffer");
        return DAS3R_ERR_INVALID_ARG;
    }
    const bool has_sh = in->shs != nullptr, has_cov = in->cov3D_precomp != nullptr;
    if ((has_sh && !g->dL_dshs) || (!has_sh && !g->dL_dcolors_precomp) || (has_cov && !g->dL_dcov3D) ||
        (!has_cov && !chained && (!g->dL_dscales || !g->dL_drotations))) {
        set_error("das3r_raster_backward: gradient buffer missing for a provided input");
        return DAS3R_ERR_INVALID_ARG;
    }
    // the forward's binning self-check first: nothing is launched on the strength of an invalid image / list
    if ((rc = das3r_raster_check(saved, stream))) return rc;
    Layout L;
    compute_layout(P, saved->capacity > 0 ? saved->capacity : saved->num_rendered, a->image_width, a->image_height, &L);
    // scratch = per-instance partial sums [num_rendered, 9]; no accumulator needs zeroing (no atomics anywhere)
    float *partial = g->scratch;
    bool quad_rows = false;
    if (saved->num_rendered > 0) {
        if (!saved->binning) { set_error("das3r_raster_backward: binning buffer missing"); return DAS3R_ERR_INVALID_ARG; }
        if ((rc = launch_render_backward(a, dL_dpix, saved->geom, saved->binning, saved->img, L, partial, s, &quad_rows, saved->num_rendered, saved->flags))) return rc;
    }
    return launch_preprocess_backward(a, in, saved->geom, saved->binning, L, g, partial, s, quad_rows);
}

// One 36-byte row of partial sums per instance + 16 bytes (the per-Gaussian backward reads the rows of a workgroup as 16-byte
// words up to the boundary above the last row).  (The experimental stream kernel — EXPERIMENTS=1 builds, DAS3R_RENDER_BWD=stream —
// needs up to four 48-byte rows per instance: only then is the larger figure returned.)
extern "C" size_t das3r_raster_backward_scratch_bytes(int64_t capacity) {
    const size_t c = capacity > 0 ? (size_t)capacity : 1;
#ifdef DAS3R_EXPERIMENTS
    if (switches().render_bwd == 5) return std::max(c * 9 * sizeof(float) + 16, stream_scratch_bytes(capacity));
#endif
    return c * 9 * sizeof(float) + 16;
}

extern "C" int das3r_has_experiments(void) {
#ifdef DAS3R_EXPERIMENTS
    return 1;
#else
    return 0;
#endif
}

extern "C" int das3r_mark_visible(int32_t P, const float *means3D, const float *viewmatrix, const float *projmatrix,
                                  uint8_t *present, das3r_stream_t stream) {
    (void)projmatrix;
    if (P < 0 || (P > 0 && (!means3D || !viewmatrix || !present))) { set_error("das3r_mark_visible: invalid argument"); return DAS3R_ERR_INVALID_ARG; }
    return launch_mark_visible(P, means3D, viewmatrix, present, (hipStream_t)stream);
}

// ---- test aid: fill every CU's LDS with a bit pattern ----
// The compositing kernels leave parts of their LDS arrays unwritten (list tails, staged entries past a tile's list); what they
// then read must not matter.  A test fills the LDS of every CU with NaNs / all-ones first (tests/test_gpu_raster.py).
__global__ void __launch_bounds__(1024) poison_lds_kernel(uint32_t pattern, uint32_t *sink, int words) {
    extern __shared__ uint32_t lds_words[];
    for (int i = threadIdx.x; i < words; i += 1024) lds_words[i] = pattern;
    __syncthreads();
    if (lds_words[(threadIdx.x * 37) % words] != pattern) *sink = 1u;       // (keeps the stores alive)
}
extern "C" int das3r_debug_poison_lds(uint32_t pattern, das3r_stream_t stream) {
    // per device: the sink lives on the device the call runs on, and the fill is as large as that device lets one workgroup have
    static uint32_t *sinks[MAX_DEVICES] = {};
    int dev = 0, lds_bytes = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= MAX_DEVICES) { set_error("device ordinal %d not supported", dev); return DAS3R_ERR_INVALID_ARG; }
    HIP_TRY(hipDeviceGetAttribute(&lds_bytes, hipDeviceAttributeMaxSharedMemoryPerBlock, dev));
    const int words = lds_bytes / 4 - 64;   // just under the limit (160 KB per CU on MI355X)
    if (words <= 0) { set_error("das3r_debug_poison_lds: no LDS"); return DAS3R_ERR_HIP; }
    if (!sinks[dev]) HIP_TRY(hipMalloc((void **)&sinks[dev], 4));
    HIP_TRY(hipFuncSetAttribute((const void *)poison_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, words * 4));
    hipLaunchKernelGGL(poison_lds_kernel, dim3(1024), dim3(1024), words * 4, (hipStream_t)stream, pattern, sinks[dev], words);
    HIP_TRY(hipGetLastError());
    return DAS3R_OK;
}

// ---- pair counters ----
static unsigned long long *g_pairs = nullptr;
static int g_pairs_dev = -1;   // the counters live on ONE device (bench.py counts on its own GPU); launches on any other get none
namespace das3r {
unsigned long long *pair_counters() {
    if (!g_pairs) return nullptr;
    int dev = -1;
    return (hipGetDevice(&dev) == hipSuccess && dev == g_pairs_dev) ? g_pairs : nullptr;
}
}  // namespace das3r
// enable != 0: (allocate and) zero the counters, the compositing kernels count from now on; enable == 0: read them into out[4] (may be
// null), stop counting.  Single-threaded use (bench.py); synchronises the device.
extern "C" int das3r_pair_counters(int enable, uint64_t out[4]) {
    if (enable) {
        int dev = 0;
        HIP_TRY(hipGetDevice(&dev));
        if (g_pairs && dev != g_pairs_dev) { set_error("das3r_pair_counters: already counting on device %d", g_pairs_dev); return DAS3R_ERR_INVALID_ARG; }
        if (!g_pairs) HIP_TRY(hipMalloc((void **)&g_pairs, (4 + PHASE_WORDS * PHASE_COPIES) * sizeof(unsigned long long)));
        g_pairs_dev = dev;
        HIP_TRY(hipMemset(g_pairs, 0, (4 + PHASE_WORDS * PHASE_COPIES) * sizeof(unsigned long long)));
        HIP_TRY(hipDeviceSynchronize());   // (the fill is on the null stream; the kernels that count may be on any)
        return DAS3R_OK;
    }
    if (g_pairs) {
        HIP_TRY(hipDeviceSynchronize());
        if (out) HIP_TRY(hipMemcpy(out, g_pairs, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        HIP_TRY(hipFree(g_pairs));
        g_pairs = nullptr;
    } else if (out) {
        out[0] = out[1] = out[2] = out[3] = 0;
    }
    return DAS3R_OK;
}

// Live pairs of a forward (pair_count.hip): out[0] = (pixel, splat) pairs that were blended, out[1] = (pixel, list position) pairs in
// front of the pixels' last contributors (what a walk without any culling would evaluate).  Synchronises the stream; measurement only.
extern "C" int das3r_raster_count_live_pairs(const das3r_raster_args *a, const das3r_raster_saved *saved, uint64_t out[2], das3r_stream_t stream) {
    if (!a || !saved || !out) { set_error("das3r_raster_count_live_pairs: null argument"); return DAS3R_ERR_INVALID_ARG; }
    out[0] = out[1] = 0;
    if (a->P == 0 || saved->num_rendered <= 0) return DAS3R_OK;
    if (!saved->geom || !saved->binning || !saved->img) { set_error("das3r_raster_count_live_pairs: saved buffers missing"); return DAS3R_ERR_INVALID_ARG; }
    hipStream_t s = (hipStream_t)stream;
    Layout L;
    compute_layout(a->P, saved->capacity > 0 ? saved->capacity : saved->num_rendered, a->image_width, a->image_height, &L);
    unsigned long long *dev = nullptr;
    HIP_TRY(hipMalloc((void **)&dev, 2 * sizeof(unsigned long long)));
    int rc = DAS3R_OK;
    if (hipMemsetAsync(dev, 0, 2 * sizeof(unsigned long long), s) != hipSuccess) rc = DAS3R_ERR_HIP;
    if (!rc) rc = launch_count_live_pairs(a, saved->geom, saved->binning, saved->img, L, dev, s);
    if (!rc && (hipStreamSynchronize(s) != hipSuccess || hipMemcpy(out, dev, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess)) {
        set_error("das3r_raster_count_live_pairs: device error");
        rc = DAS3R_ERR_HIP;
    }
    (void)hipFree(dev);
    return rc;
}

#ifdef DAS3R_EXPERIMENTS
static unsigned long long *g_trace = nullptr;
namespace das3r { unsigned long long *wg_trace() { return g_trace; } }
// enable != 0: (allocate and) zero the trace, the partition passes stamp from now on; enable == 0: copy it to out (TRACE_PASSES x
// TRACE_WGS x TRACE_STAMPS words) and stop.
extern "C" int das3r_debug_wg_trace(int enable, uint64_t *out) {
    const size_t words = (size_t)TRACE_PASSES * TRACE_WGS * TRACE_STAMPS;
    if (enable) {
        if (!g_trace) HIP_TRY(hipMalloc((void **)&g_trace, words * 8));
        HIP_TRY(hipMemset(g_trace, 0, words * 8));
        HIP_TRY(hipDeviceSynchronize());
        return DAS3R_OK;
    }
    if (g_trace) {
        HIP_TRY(hipDeviceSynchronize());
        if (out) HIP_TRY(hipMemcpy(out, g_trace, words * 8, hipMemcpyDeviceToHost));
        HIP_TRY(hipFree(g_trace));
        g_trace = nullptr;
    }
    return DAS3R_OK;
}
// Phase clocks (make EXPERIMENTS=1 only; tools/phase_clocks.py): while the pair counters are on, wave 0 of every workgroup of
// render_forward_rows_kernel / render_backward_blk_kernel adds the shader clocks it spent in each phase of its tile to the words
// behind the pair counters (common.h PHASE_*).  Read them BEFORE das3r_pair_counters(0, ..) frees the array.
extern "C" int das3r_debug_phase_clocks(uint64_t out[16]) {
    for (int i = 0; i < PHASE_WORDS; i++) out[i] = 0;
    if (!g_pairs) return DAS3R_OK;
    HIP_TRY(hipDeviceSynchronize());
    std::vector<unsigned long long> all(PHASE_WORDS * PHASE_COPIES);
    HIP_TRY(hipMemcpy(all.data(), g_pairs + 4, all.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < all.size(); i++) out[i % PHASE_WORDS] += all[i];
    return DAS3R_OK;
}
#endif

// ---- profiling introspection ----
extern "C" void das3r_profile_enable(int on) { g_prof_on = on != 0; }

// Synchronises, then writes one line per kernel name: "<name> <launches> <total_ms>\n"; clears the records.
// Returns the number of bytes written (excluding the terminator) or a negative status.
extern "C" int das3r_profile_report(char *buf, size_t cap) {
    struct Agg { const char *name; int n; double ms; };
    std::vector<Agg> agg;
    for (auto &r : g_prof) {
        float ms = 0.f;
        if (hipEventSynchronize(r.stop) == hipSuccess && hipEventElapsedTime(&ms, r.start, r.stop) == hipSuccess) {
            bool found = false;
            for (auto &a : agg)
                if (strcmp(a.name, r.name) == 0) { a.n++; a.ms += ms; found = true; break; }
            if (!found) agg.push_back({r.name, 1, (double)ms});
        }
        g_event_pool.push_back(r.start);
        g_event_pool.push_back(r.stop);
    }
    g_prof.clear();
    size_t off = 0;
    for (auto &a : agg) {
        int w = snprintf(buf + off, off < cap ? cap - off : 0, "%s %d %.6f\n", a.name, a.n, a.ms);
        if (w < 0 || off + (size_t)w >= cap) { set_error("das3r_profile_report: buffer too small"); return DAS3R_ERR_INVALID_ARG; }
        off += (size_t)w;
    }
    if (cap) buf[off < cap ? off : cap - 1] = 0;
    return (int)off;
}
