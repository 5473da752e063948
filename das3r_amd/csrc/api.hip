// api.hip — the C-ABI entry points of libdas3r_hip.so (include/das3r_raster.h) and the buffer layout.
#include <stdarg.h>
#include <string.h>

#include <vector>

#include "common.h"

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
void profile_begin(const char *name, hipStream_t s) {
    ProfRec r;
    r.name = name;
    if (hipEventCreate(&r.start) != hipSuccess || hipEventCreate(&r.stop) != hipSuccess) return;
    hipEventRecord(r.start, s);
    g_prof.push_back(r);
}
void profile_end(hipStream_t s) {
    if (!g_prof.empty()) hipEventRecord(g_prof.back().stop, s);
}

int sort_ipl_override() {
    const char *e = getenv("DAS3R_SORT_IPL");
    const int v = e ? atoi(e) : 0;
    return (v == 4 || v == 8 || v == 16) ? v : 0;
}

bool use_onesweep() {
    const char *e = getenv("DAS3R_SORT");  // "classic" = histogram + row scan + scatter per digit
    return !(e && e[0] == 'c');
}

bool use_tight_rect() {
    const char *e = getenv("DAS3R_RECT");
    return !(e && e[0] == 'u');
}

void compute_layout(int P, int64_t I, int W, int H, Layout *L) {
    memset(L, 0, sizeof(*L));
    const size_t Pn = P > 0 ? (size_t)P : 1, In = I > 0 ? (size_t)I : 1;
    L->capacity = I;
    L->tiles_x = (W + TILE_X - 1) / TILE_X;
    L->tiles_y = (H + TILE_Y - 1) / TILE_Y;
    L->ntiles = L->tiles_x * L->tiles_y;
    L->tbits = tile_bits(L->ntiles);
    L->tile_passes = (L->tbits + 7) / 8;
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
    L->b_ghist = take(4 * 4 * RADIX_SIZE);
    L->b_ticket = take(256);
    L->b_status = take(onesweep_status_bytes((int64_t)In, L->tile_passes));
    L->b_ctrl_bytes = o - L->b_ghist;
    L->pub.binning_bytes = o;
    L->pub.point_list = (L->tile_passes & 1) ? L->b_valB : L->b_valA;
    // img
    o = 0;
    const size_t npix = (size_t)W * (size_t)H;
    L->pub.final_T = take(4 * (npix > 0 ? npix : 1));
    L->pub.n_contrib = take(4 * (npix > 0 ? npix : 1));
    L->pub.ranges = take(8 * (size_t)(L->ntiles > 0 ? L->ntiles : 1));
    L->pub.img_bytes = o;
}

static int validate(const das3r_raster_args *a, const das3r_raster_in *in) {
    if (!a || !in) { set_error("null args"); return DAS3R_ERR_INVALID_ARG; }
    if (a->P < 0 || a->image_width <= 0 || a->image_height <= 0) { set_error("bad extents P=%d W=%d H=%d", a->P, a->image_width, a->image_height); return DAS3R_ERR_INVALID_ARG; }
    if (a->P == 0) return DAS3R_OK;
    if (!a->bg || !a->viewmatrix || !a->projmatrix || !a->campos) { set_error("bg/viewmatrix/projmatrix/campos must be device pointers"); return DAS3R_ERR_INVALID_ARG; }
    if (!in->means3D || !in->opacities) { set_error("means3D/opacities are required"); return DAS3R_ERR_INVALID_ARG; }
    if ((in->shs != nullptr) == (in->colors_precomp != nullptr)) { set_error("Please provide excatly one of either SHs or precomputed colors!"); return DAS3R_ERR_INVALID_ARG; }
    const bool sr = in->scales != nullptr && in->rotations != nullptr;
    if ((in->scales != nullptr) != (in->rotations != nullptr) || sr == (in->cov3D_precomp != nullptr)) {
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
extern "C" const char *das3r_last_error(void) { return g_err; }

extern "C" int das3r_raster_get_layout(int32_t P, int64_t num_rendered, int32_t W, int32_t H, das3r_raster_layout *out) {
    if (!out || P < 0 || num_rendered < 0 || W <= 0 || H <= 0) { set_error("das3r_raster_get_layout: invalid argument"); return DAS3R_ERR_INVALID_ARG; }
    Layout L;
    compute_layout(P, num_rendered, W, H, &L);
    *out = L.pub;
    return DAS3R_OK;
}

extern "C" int64_t das3r_raster_forward(const das3r_raster_args *a, const das3r_raster_in *in, const das3r_raster_out *out,
                                        das3r_alloc_fn alloc_geom, das3r_alloc_fn alloc_binning, das3r_alloc_fn alloc_img,
                                        void *user, das3r_raster_saved *saved, das3r_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
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
    if (!saved->geom || !saved->img) { set_error("scratch allocation failed (geom %zu B, img %zu B)", L.pub.geom_bytes, L.pub.img_bytes); return DAS3R_ERR_ALLOC; }
    if (P == 0) {
        // upstream:rasterize_points.cu skips the rasterizer when P == 0: the image stays zero (background NOT applied)
        HIP_TRY(hipMemsetAsync(out->out_color, 0, sizeof(float) * 3 * (size_t)W * H, s));
        HIP_TRY(hipMemsetAsync(saved->img + L.pub.ranges, 0, 8 * (size_t)L.ntiles, s));
        saved->binning = alloc_binning(user, 256);
        return 0;
    }
    // The tile partition can only raise its error word after the count has been copied back.  It is collected without
    // ever blocking the host: copied to pinned memory at the end of this forward and examined at the start of the next one
    // (debug mode waits for it right away).  Bits: 1 look-back timeout, 2 index out of range (write suppressed), 8 counts
    // do not add up to the histogram.
    static thread_local uint32_t *h_late = nullptr;
    static thread_local hipEvent_t ev_late = nullptr;
    static thread_local bool late_pending = false;
    if (!h_late) { HIP_TRY(hipHostMalloc((void **)&h_late, 64, hipHostMallocDefault)); h_late[0] = 0; }
    if (!ev_late) HIP_TRY(hipEventCreateWithFlags(&ev_late, hipEventDisableTiming));
    if (late_pending && hipEventQuery(ev_late) == hipSuccess) {
        late_pending = false;
        if (h_late[0]) {
            set_error("the previous forward's tile partition failed its self-check (flags 0x%x); its output was invalid", h_late[0]);
            h_late[0] = 0;
            return DAS3R_ERR_HIP;
        }
    }
    // hinted path: the binning buffer can be allocated up front, and its control words are zeroed by the preprocess kernel
    // instead of a separate memset
    bool binning_ready = false;
    if (a->capacity_hint > 0 && use_onesweep() && a->capacity_hint <= (int64_t)0x7FFFFF00) {
        Layout Lb;
        compute_layout(P, a->capacity_hint, W, H, &Lb);
        saved->binning = alloc_binning(user, Lb.pub.binning_bytes);
        if (!saved->binning) { set_error("scratch allocation failed (binning %zu B)", Lb.pub.binning_bytes); return DAS3R_ERR_ALLOC; }
        if ((rc = launch_preprocess(a, in, out->radii, saved->geom, saved->img, saved->binning + Lb.b_ghist, Lb.b_ctrl_bytes, L, s))) return rc;
        binning_ready = true;
    } else if ((rc = launch_preprocess(a, in, out->radii, saved->geom, saved->img, nullptr, 0, L, s))) return rc;
    if ((rc = launch_depth_sort(P, saved->geom, L, a->debug != 0, s))) return rc;

    // num_rendered comes out of the scan of tiles_touched.  Without a capacity hint the scan runs alone and the host waits
    // for its 8-byte result (upstream does the same) before sizing the binning buffer; with a hint the scan is fused with
    // the instance emission, everything is enqueued first and the count is only collected afterwards.
    static thread_local uint32_t *h_count = nullptr;
    static thread_local hipEvent_t ev_count = nullptr;
    if (!h_count) HIP_TRY(hipHostMalloc((void **)&h_count, 64, hipHostMallocDefault));
    if (!ev_count) HIP_TRY(hipEventCreateWithFlags(&ev_count, hipEventDisableTiming));
    int64_t cap = a->capacity_hint > 0 ? a->capacity_hint : -1, I = -1;
    bool scanned = false;
    auto collect_count = [&]() -> int {
        HIP_TRY(hipMemcpyAsync(h_count, saved->geom + L.g_count, 8, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipEventRecord(ev_count, s));
        return DAS3R_OK;
    };
    for (int attempt = 0; attempt < 2; attempt++) {
        if ((cap < 0 || !use_onesweep()) && !scanned) {
            if ((rc = launch_scan(P, saved->geom, L, a->debug != 0, s))) return rc;
            if ((rc = collect_count())) return rc;
            scanned = true;
        }
        if (cap < 0) {  // exact sizing: wait for the count now
            HIP_TRY(hipEventSynchronize(ev_count));
            I = (int64_t)h_count[0];
            cap = I;
        }
        if (cap > (int64_t)0x7FFFFF00) { set_error("num_rendered %lld exceeds 2^31", (long long)cap); return DAS3R_ERR_OVERFLOW; }
        compute_layout(P, cap, W, H, &L);
        if (!(binning_ready && attempt == 0)) {
            saved->binning = alloc_binning(user, L.pub.binning_bytes);
            if (!saved->binning) { set_error("scratch allocation failed (binning %zu B)", L.pub.binning_bytes); return DAS3R_ERR_ALLOC; }
        }
        const bool fused_scan = !scanned;
        if (fused_scan) {   // the count leaves right behind the scan, ahead of the partition passes
            if ((rc = launch_binning_scan_emit(P, cap, out->radii, saved->geom, saved->binning, L, binning_ready && attempt == 0, a->debug != 0, s))) return rc;
            if ((rc = collect_count())) return rc;
            scanned = true;
        }
        if ((rc = launch_binning(P, cap, W, H, out->radii, saved->geom, saved->binning, saved->img, L, fused_scan, a->debug != 0, s))) return rc;
        if ((rc = launch_render_forward(a, in->colors_precomp, out->out_color, saved->geom, saved->binning, saved->img, L, s))) return rc;
        if (I < 0) {  // hinted path: everything is enqueued; now collect the count (available since the scan finished)
            HIP_TRY(hipEventSynchronize(ev_count));
            I = (int64_t)h_count[0];
        }
        if (h_count[1] != 0) { set_error("radix look-back timed out (flags 0x%x)", h_count[1]); return DAS3R_ERR_HIP; }
        if (I <= cap) break;
        cap = -1;  // hint too small: lists were truncated, redo binning + render with the exact size
    }
    if (use_onesweep()) {
        if (late_pending) HIP_TRY(hipEventSynchronize(ev_late));   // rare: the previous copy has not landed yet
        if (late_pending && h_late[0]) { late_pending = false; set_error("an earlier forward's tile partition failed its self-check (flags 0x%x)", h_late[0]); h_late[0] = 0; return DAS3R_ERR_HIP; }
        HIP_TRY(hipMemcpyAsync(h_late, (const uint32_t *)(saved->geom + L.g_ticket) + 8, 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipEventRecord(ev_late, s));
        late_pending = true;
        if (a->debug) {
            HIP_TRY(hipEventSynchronize(ev_late));
            late_pending = false;
            if (h_late[0]) { set_error("tile partition self-check failed (flags 0x%x)", h_late[0]); h_late[0] = 0; return DAS3R_ERR_HIP; }
        }
    }
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
    if (!saved || !saved->geom || !saved->img || !dL_dpix || !g || !g->dL_dmeans2D || !g->dL_dopacities || !g->dL_dmeans3D || (saved && saved->num_rendered > 0 && !g->scratch)) {
        set_error("das3r_raster_backward: null saved state / gradient buffer");
        return DAS3R_ERR_INVALID_ARG;
    }
    const bool has_sh = in->shs != nullptr, has_cov = in->cov3D_precomp != nullptr;
    if ((has_sh && !g->dL_dshs) || (!has_sh && !g->dL_dcolors_precomp) || (has_cov && !g->dL_dcov3D) ||
        (!has_cov && (!g->dL_dscales || !g->dL_drotations))) {
        set_error("das3r_raster_backward: gradient buffer missing for a provided input");
        return DAS3R_ERR_INVALID_ARG;
    }
    Layout L;
    compute_layout(P, saved->capacity > 0 ? saved->capacity : saved->num_rendered, a->image_width, a->image_height, &L);
    // scratch = per-instance partial sums [num_rendered, 9]; no accumulator needs zeroing (no atomics anywhere)
    float *partial = g->scratch;
    if (saved->num_rendered > 0) {
        if (!saved->binning) { set_error("das3r_raster_backward: binning buffer missing"); return DAS3R_ERR_INVALID_ARG; }
        if ((rc = launch_render_backward(a, dL_dpix, saved->geom, saved->binning, saved->img, L, partial, s))) return rc;
    }
    return launch_preprocess_backward(a, in, saved->geom, saved->binning, L, g, partial, s);
}

extern "C" int das3r_mark_visible(int32_t P, const float *means3D, const float *viewmatrix, const float *projmatrix,
                                  uint8_t *present, das3r_stream_t stream) {
    (void)projmatrix;
    if (P < 0 || (P > 0 && (!means3D || !viewmatrix || !present))) { set_error("das3r_mark_visible: invalid argument"); return DAS3R_ERR_INVALID_ARG; }
    return launch_mark_visible(P, means3D, viewmatrix, present, (hipStream_t)stream);
}

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
        hipEventDestroy(r.start);
        hipEventDestroy(r.stop);
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
