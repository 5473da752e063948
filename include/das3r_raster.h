/*
 * das3r_raster.h — C-ABI of libdas3r_hip.so: the MI355X (gfx950) differentiable Gaussian-splat
 * rasterizer + distCUDA2 that sit behind DAS3R's gaussian_renderer.render().
 *
 * Each entry point replaces one function of the reference's native extension modules (both are
 * un-vendored submodules of /root/reference — .gitmodules:1-6 — so the interfaces are cited by their
 * reference CALL SITES plus the upstream symbol they stand for):
 *
 *   das3r_raster_forward    <- diff_gaussian_rasterization._C.rasterize_gaussians
 *                              (upstream:rasterize_points.cu RasterizeGaussiansCUDA), reached from
 *                              /root/reference/gaussian_renderer/__init__.py:131-140 through
 *                              GaussianRasterizer.forward / _RasterizeGaussians.forward
 *   das3r_raster_backward   <- diff_gaussian_rasterization._C.rasterize_gaussians_backward
 *                              (upstream:rasterize_points.cu RasterizeGaussiansBackwardCUDA), reached from
 *                              loss.backward() at /root/reference/train_gui.py:579
 *   das3r_mark_visible      <- diff_gaussian_rasterization._C.mark_visible (GaussianRasterizer.markVisible)
 *   das3r_knn3_mean_dist2   <- simple_knn._C.distCUDA2, /root/reference/scene/gaussian_model.py:213,641
 *
 * Conventions: every data pointer is a DEVICE pointer to contiguous row-major fp32 unless
 * stated; matrices are in the reference's row-vector layout (p_row @ M, i.e. what
 * scene/cameras.py:90-93 stores); no exceptions cross the ABI — functions return >= 0 on success and a
 * negative das3r_status on failure, with a message available from das3r_last_error().  The library owns no
 * memory across calls: scratch comes from caller-supplied allocators (mirroring upstream's
 * std::function<char*(size_t)> resize callbacks) and stays alive in the caller's autograd context until
 * backward.  All work is enqueued on the caller's HIP stream.
 *
 * Deviations from the design-target ABI of SURVEY.md §8b, on purpose:
 *   - NO `_cpu` twins of the entry points.  The survey's table lists CPU variants next to the device ones; here the CPU
 *     side of every comparison is the oracle (oracle/raster_oracle.c, oracle/knn_oracle.c: test infrastructure, also the
 *     reported `cpu_baseline`), and the product has no CPU path at all — a missing GPU / library / device tensor is an
 *     error, never a fallback.  A `_cpu` twin inside this library would be exactly the fallback the parity rules forbid.
 *   - das3r_raster_forward returns the instance count as int64 and fills a `saved` record (num_rendered, capacity, the
 *     self-check ticket) instead of upstream's tuple; `capacity_hint` lets a caller forbid the speculative buffer layout;
 *     the k-NN entry point takes a caller-owned workspace.
 *
 * State between calls: the library owns no tensor, scratch or stream.  Per host thread and per device it keeps a pinned
 * host mailbox (instance count, error word, a ring of 16 self-check tickets), the arrival counters of the count reduction,
 * the control ring of the fused scan and the last shape's instance counts (speculative capacity).  Calls of one thread on
 * one device must be issued in stream order; different threads / devices are independent.  Experiment switches (DAS3R_*
 * environment variables) are read once and cached; das3r_reload_switches() re-reads them.
 */
#ifndef DAS3R_RASTER_H
#define DAS3R_RASTER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DAS3R_ABI_VERSION 15

typedef enum {
    DAS3R_OK = 0,
    DAS3R_ERR_INVALID_ARG = -1,
    DAS3R_ERR_HIP = -2,
    DAS3R_ERR_ALLOC = -3,
    DAS3R_ERR_OVERFLOW = -4
} das3r_status;

/* hipStream_t passed as an opaque pointer so that this header needs no HIP include. */
typedef void *das3r_stream_t;

/* Scratch allocator: must return a device pointer to at least `bytes` bytes, 256-byte aligned, valid until
 * the caller drops it (the caller keeps geom/binning/img alive for backward).  NULL = failure. */
typedef char *(*das3r_alloc_fn)(void *user, size_t bytes);

/* The 12 fields of GaussianRasterizationSettings (/root/reference/gaussian_renderer/__init__.py:62-78)
 * plus the tensor extents. */
typedef struct {
    int32_t P;              /* number of Gaussians */
    int32_t sh_degree;      /* active SH degree D (0..3) */
    int32_t M;              /* stored SH coefficients per channel (max_degree+1)^2; 0 if shs == NULL */
    int32_t image_width;
    int32_t image_height;
    float tanfovx;
    float tanfovy;
    float scale_modifier;
    const float *bg;         /* [3]    device */
    const float *viewmatrix; /* [4,4]  device, row-vector layout */
    const float *projmatrix; /* [4,4]  device, row-vector layout */
    const float *campos;     /* [3]    device */
    int32_t prefiltered;     /* !=0: the caller's promise that no Gaussian fails the near-plane cull (upstream:auxiliary.h in_frustum traps
                              * when one does); das3r_raster_forward then fails with upstream's message instead of rendering */
    int32_t debug;           /* !=0: synchronise + check after every kernel */
    int64_t capacity_hint;   /* 0: the library lays the binning buffer out as it sees fit — exactly for num_rendered (which reaches
                              * the host from the first kernel of the forward), or, for a shape it has seen before, with headroom
                              * over that forward's count so that nothing waits for the count (das3r_raster_saved.capacity tells
                              * which; a scene that outgrew the headroom is redone exactly).  -1: always exactly.  Other values
                              * are reserved. */
} das3r_raster_args;

/* ABI 14, SURVEY.md section 8(f)-1: the raw parameters of a DAS3R model and the pose of the view — what das3r_pretransform_forward (below)
 * turns into camera-frame tensors in a pass of its own (/root/reference/gaussian_renderer/__init__.py:83-97,107).  Handed to the rasterizer
 * through das3r_raster_in.pre instead, the per-Gaussian kernels of forward AND backward take the transform on their way in (the same
 * arithmetic bit for bit: csrc/pretransform_math.h): means3D / rotations / scales / opacities are never written to or read from memory.
 * The backward still returns dL/d(camera-frame means3D, scales, rotations, opacities): das3r_pretransform_backward[_adam] consumes them. */
typedef struct {
    const float *xyz;            /* [P,3] */
    const float *rot;            /* [P,4] */
    const float *scaling;        /* [P,3] log scales */
    const float *opacity_raw;    /* [P,1] logits */
    const float *conf_flat;      /* confidence map, flattened */
    const int64_t *mask_index;   /* [P] position of Gaussian i in conf_flat, or NULL = identity */
    const float *R, *t, *Lq;     /* [3,3], [3], [4,4] row-major (das3r_pose_matrices[_qt]) */
} das3r_pretransform;

/* Inputs of GaussianRasterizer.forward.  Exactly one of shs/colors_precomp and exactly one of
 * (scales,rotations)/cov3D_precomp is non-NULL — unless `pre` is given: then means3D, opacities, scales and rotations are ignored
 * (may be NULL) and cov3D_precomp must be NULL. */
typedef struct {
    const float *means3D;        /* [P,3] */
    const float *opacities;      /* [P,1] */
    const float *shs;            /* [P,M,3] or NULL */
    const float *colors_precomp; /* [P,3]   or NULL */
    const float *scales;         /* [P,3]   or NULL (already activated) */
    const float *rotations;      /* [P,4]   or NULL (w,x,y,z), NOT normalised */
    const float *cov3D_precomp;  /* [P,6]   or NULL */
    const das3r_pretransform *pre; /* ABI 14: NULL, or the raw parameters + pose (HOST struct of device pointers) */
} das3r_raster_in;

typedef struct {
    float *out_color; /* [3,H,W] planar */
    int32_t *radii;   /* [P] */
} das3r_raster_out;

/* What forward leaves behind for backward (upstream: geomBuffer, binningBuffer, imgBuffer, num_rendered). */
typedef struct {
    char *geom;
    char *binning;
    char *img;
    int64_t num_rendered;
    int64_t capacity;        /* instances the binning buffer was laid out for (>= num_rendered) */
    void *check_word;        /* HOST address of the {flags, tag} word this forward's binning self-check is delivered to ... */
    uint32_t check_tag;      /* ... and the tag that marks it as this forward's (0: nothing to check) */
    uint32_t flags;          /* ABI 14 (in what was the struct's tail padding: size and offsets unchanged).  Bit 0: this forward's tile lists are
                              * long and spatially coherent or skewed (a real sequence's depth maps) — it was composited by the 2x2-region kernel
                              * and its backward pass takes the 2x2-region kernel too (render_regions.hip / render_bwd_rgn.hip).  A caller that
                              * rebuilds this struct for the backward call hands the forward's value back; 0 is always valid (block walk). */
} das3r_raster_saved;

/* Gradient outputs of backward.  Every buffer is fully written by the call (no pre-zeroing needed). */
typedef struct {
    float *dL_dmeans2D;        /* [P,3]  (x,y) = pixel-space grad * (W/2, H/2); z = 0 */
    float *dL_dopacities;      /* [P,1] */
    float *dL_dmeans3D;        /* [P,3] */
    float *dL_dshs;            /* [P,M,3] dense; NULL when colors_precomp was used */
    float *dL_dcolors_precomp; /* [P,3]; NULL when shs was used */
    float *dL_dscales;         /* [P,3]; NULL when cov3D_precomp was used */
    float *dL_drotations;      /* [P,4]; NULL when cov3D_precomp was used */
    float *dL_dcov3D;          /* [P,6]; NULL unless cov3D_precomp was used */
    float *scratch;            /* das3r_raster_backward_scratch_bytes(saved->capacity) bytes of caller-provided scratch: partial
                                * sums per instance (the compositing kernels use no atomics) */
    const struct das3r_chain_s *chain; /* ABI 14: NULL, or "go on through the pose pre-transform" (das3r_chain below) */
} das3r_raster_grads;

/* Returns num_rendered (>= 0) or a negative das3r_status.  Fills *saved. */
int64_t das3r_raster_forward(const das3r_raster_args *args, const das3r_raster_in *in, const das3r_raster_out *out,
                             das3r_alloc_fn alloc_geom, das3r_alloc_fn alloc_binning, das3r_alloc_fn alloc_img,
                             void *alloc_user, das3r_raster_saved *saved, das3r_stream_t stream);

int das3r_raster_backward(const das3r_raster_args *args, const das3r_raster_in *in, const das3r_raster_saved *saved,
                          const float *dL_dpix /* [3,H,W] */, const das3r_raster_grads *grads, das3r_stream_t stream);

/* Bytes of device scratch das3r_raster_backward needs in grads->scratch for a forward with the given saved->capacity:
 * 36 bytes (nine partial sums) per instance + 16 (the rows are read back as 16-byte words).  16-byte alignment of the buffer
 * lets the per-Gaussian backward do that; any other alignment falls back to 4-byte loads. */
size_t das3r_raster_backward_scratch_bytes(int64_t capacity);

/* das3r_raster_forward returns as soon as its kernels are enqueued.  Its binning kernels check themselves (a bounded wait on
 * another workgroup that timed out, an index out of range, counts that do not add up) and leave one word for the host; this call
 * waits (bounded) for that word of the forward that produced `saved` and returns DAS3R_ERR_HIP if the forward's image and lists are
 * invalid.  das3r_raster_backward calls it before launching anything; callers that render without a backward (evaluation) call it
 * before they use the image.  A forward with args->debug != 0 has already waited for it.  The ticket may be examined from any
 * host thread. */
int das3r_raster_check(const das3r_raster_saved *saved, das3r_stream_t stream);

/* present[i] = (view-space z of means3D[i]) > near plane (0.001, /root/reference/README.md:41-44). */
int das3r_mark_visible(int32_t P, const float *means3D, const float *viewmatrix, const float *projmatrix,
                       uint8_t *present, das3r_stream_t stream);

/* distCUDA2: out[i] = mean of squared distances to the 3 nearest other points.  `workspace` must hold
 * das3r_knn3_workspace_bytes(P) bytes of device memory. */
size_t das3r_knn3_workspace_bytes(int32_t P);
int das3r_knn3_mean_dist2(int32_t P, const float *points /* [P,3] */, float *out /* [P] */, char *workspace,
                          das3r_stream_t stream);

/* ---- opt-in fused callers' work around the rasterizer (the default DAS3R path does not need them) ---- */

/* pose (qw,qx,qy,qz,tx,ty,tz) -> mats[28] = R (9, rotation of the normalised quaternion: get_camera_from_tensor), t (3), Lq (16,
 * left-multiplication matrix of the raw quaternion: quadmultiply) — the three small matrices das3r_pretransform_* take — and
 * the chain rule back from dL/d(mats) (the g_small of das3r_pretransform_backward) to dL/d(pose). */
int das3r_pose_matrices(const float *pose, float *mats, das3r_stream_t stream);
int das3r_pose_chain(const float *pose, const float *g_mats, float *g_pose, das3r_stream_t stream);
/* The same two with the pose given as DAS3R stores it — one row of Q (frames, 4) and one of T (frames, 3), scene/gaussian_model.py:
 * 149-184 — and its gradient written straight into the rows of two dense gradient buffers (no torch.cat, no index backward).
 * das3r_pose_chain_qt also leaves g_mats[0 .. 28) ZERO: the sums of das3r_pretransform_backward are accumulated into it, so a
 * caller-owned buffer stays zero at rest (ABI 10). */
int das3r_pose_matrices_qt(const float *q, const float *t, float *mats, das3r_stream_t stream);
int das3r_pose_chain_qt(const float *q, float *g_mats, float *g_q, float *g_t, das3r_stream_t stream);
/* ABI 15 — das3r_pose_chain_qt that first ZEROES the rows zero_q[0 .. 4) / zero_t[0 .. 3) (either may be NULL; they may be g_q / g_t
 * themselves): the rows the previous view left in dense (frames, 4) / (frames, 3) gradient buffers, which the camera optimizer (dense
 * Adam over every frame, train_gui.py:584-586) must find zero — two fill launches per iteration otherwise. */
int das3r_pose_chain_qt_rearm(const float *q, float *g_mats, float *g_q, float *g_t, float *zero_q, float *zero_t, das3r_stream_t stream);
/* §8f-1: the per-Gaussian pre-transform + activations of /root/reference/gaussian_renderer/__init__.py:83-97,107 in one
 * pass: means3D = R xyz + t, rotations = Lq rot (quadmultiply(pose[:4], .) as a 4x4 matrix), scales = exp(scaling),
 * opacities = sigmoid(opacity_raw) * conf_flat[mask_index[i]] (mask_index NULL = identity).  R [3,3], t [3], Lq [4,4]:
 * row-major device tensors.  backward: g_conf_flat (size of conf_flat) and g_small [28] = dL/dR (9), dL/dt (3),
 * dL/dLq (16) must be zero on entry. */
int das3r_pretransform_forward(int32_t P, const float *xyz, const float *rot, const float *scaling, const float *opacity_raw,
                               const float *conf_flat, const int64_t *mask_index, const float *R, const float *t, const float *Lq,
                               float *means3D, float *rotations, float *scales, float *opacities, das3r_stream_t stream);
int das3r_pretransform_backward(int32_t P, const float *xyz, const float *rot, const float *scaling, const float *opacity_raw,
                                const float *conf_flat, const int64_t *mask_index, const float *R, const float *Lq,
                                const float *g_means3D, const float *g_rot, const float *g_scales, const float *g_opac, float *g_xyz,
                                float *g_rotation, float *g_scaling, float *g_opacity_raw, float *g_conf_flat, float *g_small,
                                das3r_stream_t stream);

/* §8f-2: one multi-tensor Adam step (torch.optim.Adam semantics, no weight decay / amsgrad;
 * /root/reference/scene/gaussian_model.py:236-261).  A tensor is `rows` rows of `row_len` floats of which only the first
 * `active_len` are updated (degree-aware SH coefficients; active_len == row_len for ordinary tensors).  The first `head_len`
 * floats of a row may use a different learning rate (one [P,16,3] SH tensor whose DC part and rest part belong to the
 * reference's two parameter groups f_dc / f_rest): they step with step_size, the others with step_size_tail; head_len == 0 or
 * >= active_len means one rate (step_size) for the whole row.  `tensors` is a HOST array. */
typedef struct {
    float *param;
    const float *grad;
    float *exp_avg;
    float *exp_avg_sq;
    int64_t rows;
    int32_t row_len;
    int32_t active_len;
    float step_size;  /* lr / (1 - beta1^t) */
    float bc2_sqrt;   /* sqrt(1 - beta2^t) */
    int32_t head_len;
    float step_size_tail;
    int32_t grad_row_len; /* ABI 9: floats per row of `grad` when it is more compact than the parameter (>= active_len: e.g. the
                             gradient of the first 3 of 15 SH coefficients as [P, 3, 3]); 0 = row_len */
    int32_t state_row_len; /* ABI 14 (in what was tail padding): floats per row of exp_avg / exp_avg_sq when the MOMENTS are kept more
                              compact than the parameter (>= active_len; 0 = row_len).  The moments of SH coefficients above the active
                              degree are exactly zero: FusedAdam keeps only the active prefix — at degree 1, [P, 3, 3] beside a [P, 15, 3]
                              parameter — so that four of the step's six streams are dense (adam_kernel 0.46 -> see docs/ledger.md (bh)) */
    float *mirror;          /* ABI 14: NULL, or a second place the UPDATED active values of each row are written to: row r, column c of the active
                               prefix -> mirror[r * mirror_row_len + c].  das3r_amd/fast_step.py keeps the packed [P, K, 3] SH tensor the rasterizer
                               reads (DC + the active rest coefficients) current this way — the steps of f_dc and f_rest write into it, at column
                               offsets 0 and 3 — instead of concatenating the two parameters before every render (0.16 ms at 2.13 M Gaussians) */
    int32_t mirror_row_len;
} das3r_adam_tensor;
int das3r_adam_step(int32_t n, const das3r_adam_tensor *tensors, float beta1, float beta2, float eps, das3r_stream_t stream);
/* The same step taken iff the DEVICE scalar gate[0] > threshold (DAS3R's camera optimizer steps only when the frame PSNR exceeds
 * 26 dB, train_gui.py:584-586; deciding on the host stalls it every iteration).  state: two device int32, zero-initialised by the
 * caller: [0] counts the steps actually taken (bias corrections are computed from it on the device), [1] is scratch.
 * tensors[i].step_size / step_size_tail hold the plain learning rates here; bc2_sqrt is ignored. */
int das3r_adam_step_gated(int32_t n, const das3r_adam_tensor *tensors, float beta1, float beta2, float eps, const float *gate,
                          float threshold, int32_t *state, das3r_stream_t stream);

/* ABI 14 — das3r_raster_grads.chain: the rasterizer's backward goes on where it used to stop.  With in->pre (raw parameters + pose) its
 * per-Gaussian kernel holds dL/d(camera-frame means, scales, rotations, opacities) in registers: instead of writing them for
 * das3r_pretransform_backward_adam to read back, it applies that function's chain rule and Adam step itself (the same arithmetic, bit for
 * bit: csrc/pretransform_chain.h) — slots[0..3] = xyz, rotation, scaling, opacity exactly as for das3r_pretransform_backward_adam (they
 * must be the tensors of in->pre), g_conf_flat written at the mask positions, the 28 pose sums ADDED to g_small (zero at rest) in a fixed
 * order.  grads->dL_dmeans3D / dL_dscales / dL_drotations / dL_dopacities are then not written (may be NULL).  SH coefficients in an
 * unstaged layout only (M < 16, or the active degree below 2): otherwise DAS3R_ERR_INVALID_ARG, use the two calls. */
typedef struct das3r_chain_s {
    float *g_conf_flat;
    float *g_small;
    const struct das3r_adam_slot_s *slots;
    float beta1, beta2, eps;
} das3r_chain;

/* ABI 11: the backward of the pre-transform with the Adam step of the four tensors it differentiates — xyz, rotation, scaling, raw
 * opacity: the reference's groups "xyz", "rotation", "scaling", "opacity" (/root/reference/scene/gaussian_model.py:236-261) — taken in the
 * same pass: their gradients are neither written nor read back, the parameters are read once for both.  slots[0 .. 3] (a HOST array)
 * name the four tensors in that order (they are the xyz / rot / scaling / opacity_raw of das3r_pretransform_backward, updated in
 * place) with their moments and step_size = lr / (1 - beta1^t), bc2_sqrt = sqrt(1 - beta2^t) as in das3r_adam_tensor.  g_conf_flat and
 * g_small leave as from das3r_pretransform_backward (zero on entry): the confidence map and the camera pose have their own steps.
 * Same arithmetic as das3r_pretransform_backward followed by das3r_adam_step on the four tensors. */
typedef struct das3r_adam_slot_s {
    float *param, *exp_avg, *exp_avg_sq;
    float step_size, bc2_sqrt;
} das3r_adam_slot;
int das3r_pretransform_backward_adam(int32_t P, const float *conf_flat, const int64_t *mask_index, const float *R, const float *Lq,
                                     const float *g_means3D, const float *g_rot, const float *g_scales, const float *g_opac,
                                     float *g_conf_flat, float *g_small, const das3r_adam_slot *slots, float beta1, float beta2, float eps,
                                     das3r_stream_t stream);
/* ABI 11: g_small of das3r_pretransform_backward alone — for a pass that differentiates nothing but the camera pose (the held-out
 * pose alignment of /root/reference/train_test_psnr.py). */
int das3r_pretransform_pose_sums(int32_t P, const float *xyz, const float *rot, const float *R, const float *Lq, const float *g_means3D,
                                 const float *g_rot, float *g_small, das3r_stream_t stream);

/* ---- fused photometric loss (opt-in) ------------------------------------------------------------------
 * DAS3R's per-iteration loss (/root/reference/train_gui.py:560-571, utils/loss_utils.py:39-66): with image = render * static,
 * gt' = gt * static:  loss = mean[(1 - lambda) |image - gt'| + lambda (1 - SSIM_map(image, gt'))] over all 3*H*W elements.
 * forward: every 16x16 tile writes partials[tile][8] = {sum |d|, sum (1 - ssim), sum d^2 for channels 0, 1, 2, -, -, -}
 * (the caller adds the das3r_photometric_blocks(H, W) rows: loss = ((1-lambda) S0 + lambda S1) / (3HW), mse_c = S(2+c) / (HW))
 * and the derivative maps dmaps[4][3][H][W] the backward consumes.  backward: d_render[3,H,W], d_static[H,W] for the device
 * scalar grad_loss = dL/dloss.  render/gt are [3,H,W], static_mask [H,W], all fp32 device pointers. */
int64_t das3r_photometric_blocks(int32_t H, int32_t W);
int das3r_photometric_forward(int32_t H, int32_t W, const float *render, const float *gt, const float *static_mask, float lambda,
                              float *partials, float *dmaps, das3r_stream_t stream);
/* Adds the das3r_photometric_blocks(H, W) rows of `partials` on the device: out8 = {loss, mse_r, mse_g, mse_b, psnr_frame, 0, 0, 0}
 * with psnr_frame = mean_c 20 log10(1 / sqrt(mse_c)) (utils/image_utils.py:17-19; the 26 dB gate of train_gui.py:584) — one launch
 * instead of a dozen scalar PyTorch kernels per iteration (ABI 10). */
int das3r_photometric_finish(int32_t H, int32_t W, const float *partials, float lambda, float *out8, das3r_stream_t stream);
int das3r_photometric_backward(int32_t H, int32_t W, const float *render, const float *gt, const float *static_mask, float lambda,
                               const float *dmaps, const float *grad_loss, float *d_render, float *d_static, das3r_stream_t stream);
/* ABI 15 — das3r_photometric_backward and das3r_photometric_finish as ONE launch: the first workgroup of the backward kernel adds the
 * forward's tile sums (`partials`) into out8 on its way in.  Same values, bit for bit; one launch less between the two rasterizer passes. */
/* ABI 15 — the SSIM map itself (utils/loss_utils.py:39-66, size_average = False: 11 x 11 Gaussian window, sigma 1.5, zero padding, per channel)
 * of two [3, H, W] images, and its backward for an arbitrary upstream gradient [3, H, W] — for callers that compose the loss themselves
 * (train_gui.py:566-571 combines the map with its L1 map).  dmaps [4][3][H][W] and partials [das3r_photometric_blocks][8] are scratch the
 * forward fills (the backward reads dmaps). */
int das3r_ssim_map_forward(int32_t H, int32_t W, const float *img1, const float *img2, float *ssim_map, float *dmaps, float *partials,
                           das3r_stream_t stream);
int das3r_ssim_map_backward(int32_t H, int32_t W, const float *img1, const float *img2, const float *dmaps, const float *grad_map,
                            float *d_img1, float *d_img2, das3r_stream_t stream);
int das3r_photometric_backward_finish(int32_t H, int32_t W, const float *render, const float *gt, const float *static_mask, float lambda,
                                      const float *dmaps, const float *grad_loss, float *d_render, float *d_static, const float *partials,
                                      float *out8, das3r_stream_t stream);

/* ---- introspection (used by the parity tests and the roofline accounting) ---- */

/* Byte offsets of the saved intermediates inside geom / binning / img for given extents. */
typedef struct {
    size_t geom_bytes, binning_bytes, img_bytes;
    /* geom */
    size_t depth_key;      /* u32[P]  fp32 depth bits, 0xFFFFFFFF = culled */
    /* xy / conic_opacity / rgbd are the three float4 fields of ONE 64-byte record per Gaussian (a tile's gather of a splat
     * touches one cache line instead of three): element g of each lives at its offset + g * splat_stride bytes */
    size_t xy;             /* f32[4] per record: pixel centre x, y + culling half extents hx, hy */
    size_t conic_opacity;  /* f32[4] per record */
    size_t rgbd;           /* f32[4] per record (r,g,b,depth) */
    size_t splat_stride;   /* bytes between consecutive Gaussians' records (64) */
    size_t clamped;        /* u8[P]    bit c set = channel c clamped */
    size_t tiles_touched;  /* u32[P] */
    size_t sorted_idx;     /* u32[P]   depth rank -> gaussian index */
    size_t offsets;        /* u32[P]   exclusive instance offset per depth rank */
    /* binning */
    size_t point_list;     /* u32[I]   gaussian index per instance, ordered (tile, depth, index) */
    /* img */
    size_t final_T;        /* f32[H*W] */
    size_t n_contrib;      /* u32[H*W] */
    size_t ranges;         /* u32[tiles,2] */
} das3r_raster_layout;

int das3r_raster_get_layout(int32_t P, int64_t capacity, int32_t W, int32_t H, das3r_raster_layout *out);

/* Optional per-kernel timing: when enabled, every kernel launch of the library is bracketed by HIP events recorded on
 * the launch stream.  das3r_profile_report synchronises and writes one line per kernel, "<kernel> <launches> <total_ms>\n",
 * into buf, then clears the records; returns bytes written or a negative status.  Single-threaded use (bench.py). */
void das3r_profile_enable(int on);
int das3r_profile_report(char *buf, size_t cap);

/* Test aid: one workgroup per CU (and a few more) fills the CU's whole LDS with `pattern` (e.g. 0x7FC00000 = NaN, 0xFFFFFFFF), so that
 * a following kernel that reads LDS it has not written sees it (tests/test_gpu_raster.py). */
int das3r_debug_poison_lds(uint32_t pattern, das3r_stream_t stream);

/* Test aid: OR `bits` into the binning self-check word of every forward from now on (0 switches it off again) — how the tests
 * prove that a failed binning is reported before the backward pass launches anything.  A call, not an environment variable:
 * nothing in the environment can make the shipped library report or compute anything else than it should. */
void das3r_debug_inject_fault(uint32_t bits);

/* Test aid (mutation test of the parity suite): what = 1 makes the block-walk compositing backward
 * (render_bwd_blk.hip, the kernel of the 1 M-splat headline and of the DAS3R training shape) evaluate exp(power) (1 + 1e-4) — the
 * kind of bias a "fast exp" or a packed-record shortcut would bring in; 0 switches it off.  tests/test_gpu_fullsize.py asserts that
 * the full-size oracle comparison FAILS under it.  A call, not an environment variable (see das3r_debug_inject_fault). */
void das3r_debug_mutate(uint32_t what);

/* Pair counters of the compositing kernels (bench.py: pairs per second).  enable != 0: zero the counters and count from now on;
 * enable == 0: stop, and read into out (may be NULL): out[0] (pixel, splat) pairs the forward compositing kernel evaluated, [1] the
 * same for the backward kernel, [2] / [3] their wave iterations (64 pairs each).  Costs one atomic per wave while enabled, nothing
 * otherwise.  Single-threaded use; synchronises the device. */
int das3r_pair_counters(int enable, uint64_t out[4]);

/* ABI 12.  Measurement aid (bench.py): the LIVE (pixel, splat) pairs of the forward that produced `saved` — list position below the pixel's
 * last contributor, power <= 0, alpha >= 1/255 (what upstream:forward.cu renderCUDA blends) — into out[0], and the (pixel, list
 * position) pairs below the pixels' last contributors into out[1].  A plain walk over the saved buffers; synchronises the stream. */
int das3r_raster_count_live_pairs(const das3r_raster_args *args, const das3r_raster_saved *saved, uint64_t out[2], das3r_stream_t stream);

/* The library reads its diagnostic / experiment switches (environment variables DAS3R_*) once, at the first
 * call; a process that changes them afterwards (tests, A-B tools) calls this to have them read again. */
void das3r_reload_switches(void);

/* Process-wide counters since load: out[0] forwards, [1] binning self-check words examined by the host, [2] of those with the
 * "a stalled look-back poll was rescued by the atomic read path" note (granule.h; informational), [3] failed self-checks. */
void das3r_get_stats(uint64_t out[4]);

/* ABI 13.  Forget what the calling thread has learnt about shapes on the current device (das3r_raster.h "State between calls": the
 * instance counts a speculative capacity is laid out from, the binning path, and — round 6 — which forward compositing kernel a shape
 * gets: one workgroup per tile or four, chosen from the skew of the tile lists the shape's earlier forwards met).  Every learnt path
 * computes the same lists and the same gradients; the two forward kernels round a pixel's transmittance differently in the last bit, so
 * a job that must end bit-identical whatever ran on its thread before it calls this first (das3r_amd.farm.run_sequence_job does): its
 * forwards then see the library in the state a fresh thread finds it in.  Costs the first forwards of the next shape their short cuts. */
int das3r_raster_forget_shapes(void);

/* ABI 13.  The part of that state which shows in the last bit of a result, for checkpoints: set == 0 reads {forward kernel of the current shape
 * (0: one workgroup per tile, 1: four), forwards of the shape so far — the kernel is re-decided from the tile lists every 512th} into
 * state[0 .. 1]; set != 0 forgets the shapes (as above) and hands state[] to the next shape the calling thread meets: a job resumed from a
 * checkpoint continues with the kernel, and the re-decision schedule, the killed job had (das3r_amd.train save_checkpoint / farm resume). */
int das3r_raster_learning(int32_t set, uint32_t state[2]);

/* 1 when the library was built with the superseded experiment kernels (make EXPERIMENTS=1: DAS3R_RENDER_BWD=mfma | stream,
 * DAS3R_SORT=classic), 0 for the shipped build; the parity tests of those kernels skip themselves on 0. */
int das3r_has_experiments(void);

int das3r_abi_version(void);
const char *das3r_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* DAS3R_RASTER_H */
