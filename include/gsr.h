/*
 * gsr.h — C ABI of the MI355X (gfx950) differentiable Gaussian-splat rasterizer.
 *
 * This is the drop-in boundary for GSORB-SLAM's rasterizer core: each entry
 * point replaces one member of `CudaRasterizer::Rasterizer`
 * (Thirdparty/diff_gaussian_rasterization/cuda_rasterizer/rasterizer.h:24-99 in
 * the reference tree). Plain pointers and sizes only: no torch, no C++ types.
 * All pointers are DEVICE pointers unless stated; all floats fp32; matrices are
 * 4x4 read column-major exactly like the reference (callers pass Tcw^T and
 * (P·Tcw)^T, reference src/Camera.cc:33-36,46-47).
 *
 * Thread-safety: no global mutable state; every call works on the stream and
 * buffers it is given, so two host threads may render concurrently on one
 * device (the reference's viewer thread does, src/Viewer2.cc:256-263).
 *
 * Return value: >= 0 on success, a negative GSR_E* code otherwise.
 */
#ifndef GSR_H
#define GSR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSR_ABI_VERSION 10 /* 10: gsr_forward_args.out_sil appended (the plain 3-channel forward also stores the silhouette 1 - final T), gsr_track_loss_rows takes sil_is_transmittance, gsr_transmittance_view added; 9: gsr_pose_update_args.skip and gsr_pose_step_args.overflow_out appended (the sharded loop's overflow decision rides in its pose all-reduce); 8: gsr_forward_args.pre_Tcw / means_cam_out / raw appended (the camera transform and the map's activations inside the projection kernel), gsr_pose_step_args.sums_only and gsr_pose_finish added; 7: gsr_composite_* take the plane count of the gathered buffer, gsr_shard_order and gsr_reproj_loss added; 6: gsr_track_loss, gsr_pose_step, gsr_backward_args.fused_pose_step added; 3: out_ds / dL_dds (fused depth + silhouette channels) appended to the argument structs; 4: gsr_pixel_loss*, gsr_scale_reg* added;
                           * 5: gsr_map_prepare / gsr_map_update / gsr_map_loss_total / gsr_pose_update / gsr_pixel_loss_backward_add / gsr_composite_* added, GSR_LOSS_PARTIALS 256 -> 1024 */

#define GSR_OK 0
#define GSR_EINVAL (-1)    /* bad argument combination (NULL where data is required, sizes < 0 …) */
#define GSR_EALLOC (-2)    /* an allocation callback returned NULL */
#define GSR_EHIP (-3)      /* a HIP runtime call or kernel launch failed (see gsr_last_hip_error) */
#define GSR_EOVERFLOW (-4) /* gsr_forward_ws: binning workspace too small for num_rendered */
#define GSR_ECHANNELS (-5) /* only 3 colour channels are built (reference config.h:15) */

/* Replaces std::function<char*(size_t)> (rasterizer.h:35-37): must return a
 * device buffer of at least `bytes` bytes, 256-byte aligned, valid until the
 * matching gsr_backward has run. Contents need not be zeroed. */
typedef char* (*gsr_alloc_fn)(void* user, size_t bytes);

/* Where the projection kernel leaves the map's activations when it forms them itself (gsr_forward_args.raw, below). */
typedef struct gsr_raw_outputs {
    float* opacities;   /* [P]   sigmoid(logit) */
    float* scales;      /* [P,3] exp(log-scale) */
    float* rotations;   /* [P,4] q / max(|q|, 1e-12) */
    float reg_limit;    /* the regularisers' scale limit (gsr_map_prepare's `limit`) */
    float* reg_partial; /* NULL: no regularisers */
} gsr_raw_outputs;

/* Arguments of Rasterizer::forward (rasterizer.h:34-58), same meaning and order. */
typedef struct gsr_forward_args {
    int P;                       /* number of Gaussians (< 2^28: ids share a word with a 4-bit mask) */
    int D;                       /* active SH degree (0..3) */
    int M;                       /* SH coefficients per Gaussian (0 when colours are precomputed) */
    const float* background;     /* [3] */
    int width, height;
    const float* means3D;        /* [P,3] */
    const float* shs;            /* [P,M,3] or NULL */
    const float* colors_precomp; /* [P,3] or NULL (exactly one of shs / colors_precomp) */
    const float* opacities;      /* [P] */
    const float* scales;         /* [P,3] or NULL */
    float scale_modifier;
    const float* rotations;      /* [P,4] (r,x,y,z), used un-normalised like forward.cu:127 */
    const float* cov3D_precomp;  /* [P,6] or NULL (exactly one of scales+rotations / cov3D_precomp) */
    const float* viewmatrix;     /* [16] */
    const float* projmatrix;     /* [16] */
    const float* cam_pos;        /* [3] */
    float tan_fovx, tan_fovy;
    int prefiltered;             /* accepted for signature parity. Reference (auxiliary.h:156-160): true = the caller promises it culled
                                  * already and a splat that still fails the frustum test makes the kernel printf + __trap(). Here such a
                                  * splat is skipped (radius 0), as with false: no device trap, identical results. */
    float* out_color;            /* [3,H,W] fully written */
    float* out_depth;            /* [H,W]   fully written (median depth, forward.cu:374-379) */
    int* radii;                  /* [P] fully written, or NULL */
    /* Optional profiling hook (host array of 2*GSR_FWD_STAGES hipEvent_t, or NULL): events
     * [2i] and [2i+1] are recorded on `stream` around stage i; NULL entries are skipped. */
    void** profile_events;
    /* Tile-band sharding (multi-GPU scheme A): only tile rows [band_y0, band_y1) are binned,
     * sorted and blended, and only their pixels of out_color / out_depth are written; radii still cover
     * every splat. 0,0 (the zero-initialised default) means the whole image. A band render is bit-identical
     * to the same rows of the full render. */
    int band_y0, band_y1;
    /* Fused colour + depth / silhouette render (new capability; NULL = the reference's 3-channel render). GSORB-SLAM
     * renders every view twice with the same geometry: once with the colours, once with colors_precomp = [z, 1, 0]
     * (z = camera-frame depth) for the alpha-blended depth and the silhouette (src/Render.cc:927-981). With out_ds
     * [2,H,W] the forward blends those two channels in the same pass: out_ds[0] = sum z_i alpha_i T_i (z_i = the
     * splat's view-space depth, what preprocess computes anyway), out_ds[1] = sum alpha_i T_i; background 0. One
     * preprocess / binning / sort / gather / exp / alpha / T per pair instead of two. */
    float* out_ds;
    /* The tracking loop's camera transform inside the projection kernel (new capability; NULL = means3D is used as it is). GSORB-SLAM moves the
     * means into the camera frame before every render of a tracked pose — mc = X R^T + t through bmm, src/Render.cc:750-752 — and rasterizes with
     * an identity view. With pre_Tcw (DEVICE pointer, row-major 4x4, world -> camera) means3D holds the WORLD means, the kernel forms mc with
     * gsr_to_camera's arithmetic, uses it as the splat's mean and stores it to means_cam_out [P,3] (what gsr_backward_args.means3D then takes):
     * gsr_to_camera's launch and its 24 bytes per splat of traffic less per iteration. */
    const float* pre_Tcw;
    float* means_cam_out;
    /* The mapping loop's activations inside the projection kernel (new capability; NULL = opacities / scales / rotations are what the reference's
     * rasterizer takes). With `raw` (a HOST struct) those three hold the RAW parameters of Gaussian::GaussianOptimizer — logit opacities, log-scales,
     * un-normalised quaternions — and the kernel applies gsr_map_prepare's activations itself (sigmoid, exp, torch's normalize: the same operations),
     * stores the activated values where gsr_backward will read them, and — reg_partial != NULL — writes gsr_map_prepare's rows of the scale
     * regularisers' three sums ([3 * ceil(P / 256)], the layout gsr_map_loss_finish takes). Needs scales + rotations (not cov3D_precomp). */
    const struct gsr_raw_outputs* raw;
    /* The silhouette out of the PLAIN 3-channel forward (new capability; NULL = not wanted; only without out_ds). The silhouette GSORB-SLAM renders as colour
     * 1 of its second pass (src/Render.cc:963-975) is sum alpha_i T_i = 1 - final T of the very walk the colour pass does: with out_sil [H,W] the plain forward
     * stores 1 - T per pixel next to the colours (equal to the fused pair's out_ds[1] to rounding). A sharded tracking iteration on the surface depth needs the
     * layer's colours, surface depth and silhouette and nothing of the blended depth: it renders this way instead of the fused pair. */
    float* out_sil;
} gsr_forward_args;

/* forward stages, in launch order */
enum { GSR_FWD_PREPROCESS = 0, GSR_FWD_SCAN, GSR_FWD_FILL, GSR_FWD_SORT, GSR_FWD_BLEND, GSR_FWD_STAGES };
/* backward stages */
enum { GSR_BWD_CLEAR = 0, GSR_BWD_BLEND, GSR_BWD_SPLAT, GSR_BWD_STAGES };

/* Rasterizer::forward (rasterizer_impl.cu:199-345). Returns num_rendered (one 4-byte device→host read on
 * `stream`, like :285). `stream` is a hipStream_t (NULL = default stream).
 * Allocator contract (differs from the reference's "each callback exactly once"):
 *   geom_alloc, image_alloc: called exactly once, before any kernel.
 *   binning_alloc: called BEFORE num_rendered is known, with gsr_binning_bytes(capacity), capacity =
 *     max(P + 4096, 1.25 * num_rendered of the calling thread's previous frame on this device + 4096), and the
 *     tail kernels are enqueued behind the head at once; it is called a SECOND time, with
 *     gsr_binning_bytes(num_rendered), only if that capacity was too small (the tail is then re-run).
 *     Work on the first block may still be in flight on `stream` when the second request arrives: an adapter
 *     must keep it alive until the stream reaches that point (torch's caching allocator does: a resize_ of the
 *     same tensor frees stream-ordered) and must use the LAST block returned. 44 B per tile instance.
 * P == 0: the binning allocator is called once with gsr_binning_bytes(0). */
int gsr_forward(const gsr_forward_args* args,
                gsr_alloc_fn geom_alloc, void* geom_user,
                gsr_alloc_fn binning_alloc, void* binning_user,
                gsr_alloc_fn image_alloc, void* image_user,
                void* stream);

/* Sync-free variant for optimisation loops: the caller owns three persistent
 * workspaces (sizes from gsr_*_bytes; `binning_bytes` fixes the capacity) and
 * num_rendered never leaves the device. If the capacity is exceeded nothing is
 * rendered and the overflow flag is raised: read it with gsr_ws_status. */
int gsr_forward_ws(const gsr_forward_args* args, char* geom, char* binning, size_t binning_bytes,
                   char* image, void* stream);

/* Blocking: copies {num_rendered, overflow} of the last forward on `geom` to the host. */
int gsr_ws_status(const char* geom, void* stream, int* num_rendered, int* overflow);

struct gsr_map_update_args; /* (below: the loops' fused update) */

/* Arguments of Rasterizer::backward (rasterizer.h:60-88). R is what gsr_forward
 * returned; pass R < 0 after gsr_forward_ws (num_rendered stays on the device). The layout of the binning
 * blob follows the capacity the forward ran with, which the kernels read from the geometry header: R and
 * binning_bytes are not needed to find anything. */
typedef struct gsr_backward_args {
    int P, D, M, R;
    const float* background;
    int width, height;
    const float* means3D;
    const float* shs;
    const float* colors_precomp;
    const float* scales;
    float scale_modifier;
    const float* rotations;
    const float* cov3D_precomp;
    const float* viewmatrix;
    const float* projmatrix;
    const float* cam_pos;
    float tan_fovx, tan_fovy;
    const int* radii;            /* as written by forward, or NULL */
    char* geom_buffer;           /* the three blobs of the matching forward; geom is also scratch */
    char* binning_buffer;
    char* image_buffer;
    size_t binning_bytes;        /* unused (kept for layout compatibility of the struct) */
    const float* dL_dpix;        /* [3,H,W] */
    /* Outputs. Unlike the reference (src/Rasterizer.cu:253-261 zero-fills nine
     * tensors first) every non-NULL output is FULLY written, zeros included,
     * so callers may pass uninitialised memory. */
    float* dL_dmean2D;           /* [P,3]  (x,y in the reference's NDC-scaled units, z = 0) */
    float* dL_dconic;            /* [P,4]  (x,y,·,w) or NULL */
    float* dL_dopacity;          /* [P] */
    float* dL_dcolor;            /* [P,3] */
    float* dL_dmean3D;           /* [P,3] */
    float* dL_dcov3D;            /* [P,6] */
    float* dL_dsh;               /* [P,M,3] or NULL when M == 0 */
    float* dL_dscale;            /* [P,3] or NULL when cov3D_precomp is used */
    float* dL_drot;              /* [P,4] or NULL when cov3D_precomp is used */
    void** profile_events;       /* like gsr_forward_args.profile_events, 2*GSR_BWD_STAGES entries */
    int band_y0, band_y1;        /* must equal the matching forward's band */
    /* Which stages to run: bit 0 clear the per-splat accumulators, bit 1 blend backward (accumulates
     * into them), bit 2 per-splat stage (reads them, writes the dL_d* outputs, leaves them zero).
     * 0 = blend + per-splat + re-zero (GSR_STAGE_REZERO): the accumulators are zero after every forward,
     * and with the re-zero also after every per-splat stage, so any number of backward calls may follow one
     * forward. A caller that runs ONE backward per forward may pass GSR_STAGE_BLEND | GSR_STAGE_SPLAT and
     * save 64 bytes of stores per splat; a further backward on that state must then start with
     * GSR_STAGE_CLEAR. An explicit clear also discards a blend stage that was not followed by the
     * per-splat stage. Band sharding runs 2 on every rank, sums the accumulators across ranks
     * (gsr_acc_view + one all-reduce), then runs 4. */
    int stages;
    /* Upstream gradient of the fused depth / silhouette channels [2,H,W] (gsr_forward_args.out_ds), or NULL. dL/dalpha
     * then sums over all five channels; the depth channel's colour gradient (dL/dz_i) is folded into dL_dmean3D
     * through the third row of the view matrix (z_i is a function of the mean), everything else is written as usual. */
    const float* dL_dds;
    /* non-zero: the depth channel's colour z_i is a constant for this backward (GSORB-SLAM's tracking iterations detach the
     * [z, 1, 0] colours, src/Render.cc:949-981): its gradient is NOT folded into dL_dmean3D */
    int ds_detach_depth;
    /* Optional (NULL: off). The per-splat stage then writes NO gradient (the dL_d* outputs are ignored and may be NULL): every Gaussian's
     * raw parameters take gsr_map_update's step right there, from the gradients in registers — for a mapping iteration whose rasterizer
     * inputs are gsr_map_prepare's outputs (means3D = means_cam with an identity view matrix, colors_precomp = the rgb parameter, scales,
     * rotations, opacities the activations). Fields of the struct that name gradient tensors (dL_dmeans_cam ... dL_dscales) are ignored.
     * Needs scales + rotations, no SH, and a call that runs the per-splat stage. 112 bytes of traffic per Gaussian and one launch less
     * than gsr_backward followed by gsr_map_update. */
    const struct gsr_map_update_args* fused_map_update;
    /* 1: dL_dds holds ONE plane [1,H,W], the depth channel's upstream gradient; the silhouette's is zero (GSORB-SLAM's losses use the
     * silhouette only as a detached mask, src/Render.cc:436-471, :1088-1090): its recursion leaves the blend kernel's loop.
     * 2 (round 6): dL_dds holds ONE plane, the SILHOUETTE's upstream gradient; the depth channel's is zero — a sharded tracking iteration on the surface depth,
     * whose layer receives only what it occludes. Only without colour outputs (dL_dcolor = dL_dsh = NULL), with ds_detach_depth, and in a call that runs the
     * per-splat stage (GSR_EINVAL otherwise): the blend kernel then drops the depth channel's recursion and the colour sums. */
    int dds_depth_only;
    /* Optional (NULL: off; not together with fused_map_update). A tracking iteration's backward (the pose is the only parameter; means3D = gsr_to_camera's
     * means_cam with an identity view matrix): the per-splat stage also forms the pose sums of dL/dmeans_cam against the world-frame means (gsr_pose_grad)
     * and a one-wave kernel behind it takes the pose step (gsr_pose_update): no dL_dmean3D tensor is needed (it is still written if given). */
    const struct gsr_pose_step_args* fused_pose_step;
} gsr_backward_args;

#define GSR_STAGE_CLEAR 1
#define GSR_STAGE_BLEND 2
#define GSR_STAGE_SPLAT 4
#define GSR_STAGE_REZERO 8 /* the per-splat stage zeroes the accumulators it consumed */

/* The packed per-splat accumulators inside a geometry blob: count floats = 16 per splat (one 64-byte line:
 * moments u, u*dx, u*dy, u*dx^2, u*dx*dy, u*dy^2 of u = G * dL/dalpha, the colour sums r, g, b, 7 unused). */
int gsr_acc_view(char* geom, int P, float** acc, size_t* count);

/* Rasterizer::backward (rasterizer_impl.cu:405-498). Never allocates, never syncs. */
int gsr_backward(const gsr_backward_args* args, void* stream);

/* Rasterizer::markVisible (rasterizer_impl.cu:142-154); present is bool[P] (1 byte each). */
int gsr_mark_visible(int P, const float* means3D, const float* viewmatrix,
                     const float* projmatrix, uint8_t* present, void* stream);

/* Rasterizer::visible_filter (rasterizer_impl.cu:348-401, GSORB's radii-only
 * pass). The reference allocates geometry+image blobs it barely uses; this
 * one needs no workspace. radii [P] fully written. */
int gsr_visible_filter(int P, int width, int height, const float* means3D, const float* scales,
                       float scale_modifier, const float* rotations, const float* viewmatrix,
                       const float* projmatrix, float tan_fovx, float tan_fovy, int prefiltered,
                       int* radii, void* stream);

/* SimpleKNN::knn / distCUDA2 (reference include/simple_knn.h:15-19, src/simple_knn.cu:185-219,
 * src/spatial.cu:15-27): mean_dists[i] = mean of the three smallest squared distances from point i
 * to the other points (exact 3-NN). `workspace` is caller-owned scratch of gsr_knn_bytes(P) bytes
 * (the reference cudaMallocs its temporaries and syncs twice; this entry point does neither). */
size_t gsr_knn_bytes(int P);
int gsr_dist2(int P, const float* points /* [P,3] */, float* mean_dists /* [P] */, char* workspace,
              size_t workspace_bytes, void* stream);

/* ---- the two non-rasterizer hot spots of a mapping iteration, fused ---------------------------
 * SSIM (reference ORB_SLAM2::ssim, src/Utils.cc:77-100: five depthwise 11x11 convolutions + elementwise passes through
 * libtorch, and their autograd): mean over [C,H,W] of the SSIM map of img1 vs img2 with the 11-tap separable window
 * `taps11` (host pointer; zero padding like conv2d(padding=5)) and its gradient w.r.t. img1.
 *   gsr_ssim_forward  writes one partial sum of the map per wave of its launch (gsr_ssim_partials of them; the caller adds them
 *                     up and divides by C*H*W; H*W < 2^30) and, if dmaps != NULL, the three derivative maps [3,C,H,W] the backward needs
 *   gsr_ssim_backward dL_dimg1 [C,H,W] = (*dL_dmean / (C*H*W)) * d(sum of the map)/d(img1); dL_dmean is a DEVICE scalar
 * Adam (reference torch::optim::Adam as src/Gaussian.cc:144-175 configures it: no weight decay, no amsgrad): one
 * in-place step of one parameter tensor, `step` = the 1-based step count of that tensor; the hyper-parameters are doubles,
 * like the Python / C++ scalars the reference passes (1 - beta and the bias corrections are formed in double). Never
 * allocate, never sync. */
size_t gsr_ssim_partials(int C, int H, int W);
int gsr_ssim_forward(const float* img1, const float* img2, int C, int H, int W, const float* taps11,
                     float* partial, float* dmaps, void* stream);
int gsr_ssim_backward(const float* img1, const float* img2, const float* dmaps, int C, int H, int W,
                      const float* taps11, const float* dL_dmean, float* dL_dimg1, void* stream);
/* The camera transform mc = X R^T + t of the means (src/Render.cc:750-752: Tcw.repeat(n,1,1).bmm([x;1])) and its
 * backward (the reference leaves both to libtorch's batched GEMM and autograd). Tcw: DEVICE pointer to the 4x4 row-major pose
 * (the pose optimiser's output: no host round trip).
 *   gsr_to_camera  means_cam [n,3]
 *   gsr_pose_grad  from dL/dmeans_cam [n,3]: dL_dmeans3D [n,3] = dmc R (NULL: not wanted) and partial
 *                  [GSR_POSE_PARTIALS][12] (NULL: not wanted), row = (dL/dR row-major 3x3, dL/dt) summed over that
 *                  workgroup's splats; the caller adds the rows */
#define GSR_POSE_PARTIALS 512
int gsr_to_camera(const float* means3D /* [n,3] */, size_t n, const float* Tcw, float* means_cam, void* stream);
int gsr_pose_grad(const float* means3D /* [n,3] */, const float* dL_dmeans_cam /* [n,3] */, size_t n, const float* Tcw,
                  float* partial, float* dL_dmeans3D, void* stream);
/* rt2T (reference include/Utils.h:56-77, src/Utils.cc:170-179): un-normalised quaternion (r,x,y,z) [4] and translation [3]
 * -> Tcw [4,4] row-major, and the backward from dL/dTcw [4,4]. All pointers are device pointers. */
int gsr_pose_from_quat(const float* quat, const float* trans, float* Tcw, void* stream);
int gsr_pose_from_quat_backward(const float* quat, const float* dL_dTcw, float* dL_dquat, float* dL_dtrans, void* stream);
int gsr_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, double lr,
                  double beta1, double beta2, double eps, int step, void* stream);

/* ---- the loss terms of the two loops, fused (reference src/Render.cc:1088-1105 tracking, :436-471 mapping; src/Utils.cc:39-65:
 * libtorch tensor expressions, ~35 / ~80 small launches per iteration forwards and backwards). All pointers are DEVICE pointers.
 *   gsr_pixel_loss  mode 0 (tracking): M = sil > sil_thr && !isnan(frame_depth);
 *                       loss = w[0] * sum_M |image - frame_rgb| + w[1] * sum_M |D - frame_depth|, D = depth, or sur when depth == NULL
 *                   mode 1 (mapping): V = frame_depth > 0, S = V && sil > sil_thr;
 *                       loss = w[0] * mean |image - frame_rgb| + w[1] * sum_V |depth - frame_depth| / |V|
 *                              + w[2] * sum_S |sur - frame_depth| / max(|S|, 1)
 *                       EMPTY MASKS: a term whose mask selects no pixel is 0, in the loss and in its gradient (counts are divided as max(count, 1)).
 *                       The reference's masked_select().mean() gives NaN there (src/Render.cc:455-462, src/Utils.cc:39-56) and poisons the step; this is a
 *                       deliberate deviation (the Python harness reproduces the reference's NaN with strict_empty_terms, on the unfused path).
 *                   image, frame_rgb [3,H,W]; depth, sur, sil, frame_depth [H,W] (depth / sur / sil may be NULL: term or test absent).
 *                   partial: scratch of GSR_LOSS_PARTIALS * 5 floats; sums [8] = {sum |image - rgb|, sum |depth - fd|, its count,
 *                   sum |sur - fd|, its count (mapping), loss, 0, 0}
 *   gsr_pixel_loss_backward  dL_dimage [3,H,W] and dL_ddepth [H,W] (NULL: not wanted; zeros when depth == NULL), times *dL_dloss;
 *                   `sums` is the forward's (the mapping depth term divides by its count). sur has no gradient.
 *   gsr_scale_reg   the two scale regularisers (src/Render.cc:449-462) of log_scales [n,3]:
 *                   out [4] = {sum w, reg_scalar, sum w (max - min), w_long * reg_long + w_scalar * reg_scalar}; partial: GSR_LOSS_PARTIALS * 3 floats
 *   gsr_scale_reg_backward  dL_dlog_scales [n,3] (written, not accumulated) = *dL_dvalue * d(out[3]) / d(log_scales) */
#define GSR_LOSS_PARTIALS 1024
int gsr_pixel_loss(const float* image, const float* depth, const float* sur, const float* sil, const float* frame_rgb,
                   const float* frame_depth, int H, int W, int mode, float sil_thr, const float* w3 /* host, 3 floats */,
                   float* partial, float* sums, void* stream);
/* gsr_pixel_loss (mode 0) and gsr_pixel_loss_backward (mode 0, upstream gradient 1) in ONE pass over the render: a tracking iteration's loss and its gradient
 * planes from the same loads (src/Render.cc:1088-1105: masked L1 SUMS; the gradient needs no total). sums [8] as gsr_pixel_loss, dL_dimage [3,H,W],
 * dL_ddepth [H,W] (NULL: not wanted; zeros when depth == NULL: the median depth carries no gradient). */
int gsr_track_loss(const float* image, const float* depth, const float* sur, const float* sil, const float* frame_rgb, const float* frame_depth,
                   int H, int W, float sil_thr, const float* w3 /* host */, float* partial, float* sums, float* dL_dimage, float* dL_ddepth,
                   uint32_t* ticket /* NULL, or GSR_TICKET_WORDS DEVICE words that are zero between calls (zero them once): the sums are then finished inside the same launch */,
                   void* stream);
int gsr_pixel_loss_backward(const float* image, const float* depth, const float* sil, const float* frame_rgb, const float* frame_depth,
                            int H, int W, int mode, float sil_thr, const float* w3 /* host */, const float* sums,
                            const float* dL_dloss, float* dL_dimage, float* dL_ddepth, void* stream);
int gsr_scale_reg(const float* log_scales, size_t n, float limit, float w_long, float w_scalar, float* partial, float* out, void* stream);
int gsr_scale_reg_backward(const float* log_scales, size_t n, float limit, float w_long, float w_scalar, const float* out,
                           const float* dL_dvalue, float* dL_dlog_scales, void* stream);

/* gsr_pixel_loss_backward with a plane added to dL/dimage (`add_image` [3,H,W], e.g. the SSIM term's gradient; NULL: none) and an
 * upstream gradient that may be NULL (= 1). */
int gsr_pixel_loss_backward_add(const float* image, const float* depth, const float* sil, const float* frame_rgb, const float* frame_depth,
                                int H, int W, int mode, float sil_thr, const float* w3 /* host */, const float* sums,
                                const float* dL_dloss, const float* add_image, float* dL_dimage, float* dL_ddepth, void* stream);

/* ---- the loops without a tensor library inside the iteration (round 4; reference src/Render.cc:420-483 mapping iterations,
 * :1054-1126 tracking iterations, src/Gaussian.cc:144-175 optimisers, :97-150 pose parameters, include/Utils.h:56-77 rt2T).
 * The reference forms the rasterizer's inputs from the raw parameters with libtorch expressions (sigmoid, exp, normalize, the
 * camera transform), differentiates them with autograd and steps five Adam groups: ~60 launches per mapping iteration around
 * the two renders. These entry points are the same arithmetic as three launches; all pointers are DEVICE pointers unless noted,
 * nothing allocates or synchronises.
 *   gsr_map_prepare  n raw Gaussians -> means_cam [n,3] = xyz R^T + t (Tcw: device 4x4 row-major), opacities [n] = sigmoid(logit),
 *                    scales [n,3] = exp(log_scales), rotations [n,4] = q / max(|q|, 1e-12); any output may be NULL. With
 *                    reg_partial (scratch of 3 * ((n + 255) / 256) floats; reg_out = NULL: the rows only, for gsr_map_loss_finish) and reg_out [4] it also evaluates the two scale
 *                    regularisers (gsr_scale_reg's out: {sum w, reg_scalar, sum w (max - min), w_long * reg_long + w_scalar * reg_scalar}).
 *                    log_scales = NULL with reg_partial and reg_out and no other output: only the finish of rows that are already there (gsr_forward_args.raw wrote them).
 *   gsr_map_update   from the rasterizer's gradients (gsr_backward on camera-frame means with an identity view matrix: dL_dmean3D is
 *                    dL/dmeans_cam) to an Adam step of the five raw tensors, in place: dL/dxyz = dmc R; dL/dlogit = dopac * o (1 - o);
 *                    dL/dlog_scales = dscale * scale + the regularisers' gradient (reg_out != NULL); dL/dq through the normalisation;
 *                    dL/drgb = dL_dcolors. Adam exactly as gsr_adam_step (per tensor lr and 1-based step; order xyz, rgb, quat, logit,
 *                    log_scales). `geom` (the rasterizer's geometry blob of the iteration's forward, NULL: none): when its overflow flag
 *                    is set (gsr_forward_ws ran out of workspace) the whole update is skipped.
 *   gsr_map_loss_total  loss[0] = sums[5] (gsr_pixel_loss) + c_ssim * (1 - sum(ssim_partial) / count) + reg_out[3] (NULL: 0); NaN when
 *                    the forward on `geom` (NULL: not checked) overflowed its workspace
 *   gsr_pose_update  end of a tracking iteration (one launch): adds gsr_pose_grad's partial rows, takes them through rt2T's backward,
 *                    records history[0] = *loss and keeps the pose of the lowest loss so far in best [8] = {loss, quat[4], trans[3]}
 *                    (NaN losses never win, Render.cc:1109-1112), steps Adam on quat_trans [7] (un-normalised quaternion r,x,y,z and
 *                    translation; moments [14] = exp_avg[7], exp_avg_sq[7]; one learning rate for both groups like Gaussian.cc:149-150)
 *                    and writes the next iteration's Tcw [16]. An overflowed forward (geom) records NaN and skips the step. */
int gsr_map_prepare(size_t n, const float* xyz, const float* logit, const float* log_scales, const float* unnorm_quat, const float* Tcw,
                    float* means_cam, float* opacities, float* scales, float* rotations, float reg_limit, float w_long, float w_scalar,
                    float* reg_partial, float* reg_out, void* stream);
typedef struct gsr_map_update_args {
    size_t n;
    float *xyz, *rgb, *unnorm_quat, *logit, *log_scales;
    float* exp_avg[5];
    float* exp_avg_sq[5];
    const float *dL_dmeans_cam, *dL_dcolors, *dL_drotations, *dL_dopacities, *dL_dscales;
    const float *opacities, *scales; /* gsr_map_prepare's outputs of this iteration */
    const float* Tcw;
    const float* reg_out; /* NULL: no regulariser gradient */
    float reg_limit, w_long, w_scalar;
    const char* geom;
    double lr[5], beta1, beta2, eps;
    int step[5];
} gsr_map_update_args;
int gsr_map_update(const gsr_map_update_args* args, void* stream);
int gsr_map_loss_total(const float* sums, const float* ssim_partial, int n_partial, size_t count, float c_ssim, const float* reg_out,
                       const char* geom, float* loss, void* stream);
/* The mapping loss as two passes over the image with one single-workgroup kernel between them (instead of gsr_pixel_loss, gsr_ssim_forward,
 * gsr_ssim_backward, gsr_pixel_loss_backward_add, gsr_map_loss_total and the finish launch of gsr_map_prepare): reference src/Render.cc:436-471.
 *   gsr_map_loss_forward   SSIM forward of image vs frame_rgb ([3,H,W]; dmaps [3,3,H,W] for the backward) with gsr_pixel_loss's mode-1 sums
 *                          riding along: partial6 [6][gsr_ssim_partials(3,H,W)] = six planes of per-workgroup sums {SSIM map, |image - rgb|, |depth - fd| over fd > 0,
 *                          its count, |sur - fd| over fd > 0 && sil > sil_thr, its count} (depth / sur / sil may be NULL as in gsr_pixel_loss)
 *   gsr_map_loss_finish    sums [8] as gsr_pixel_loss (sums[6] = the SSIM map's sum), reg_out [4] as gsr_scale_reg from gsr_map_prepare's
 *                          reg_partial rows (NULL: no regularisers; call gsr_map_prepare with reg_out = NULL then), and
 *                          loss[0] = sums[5] + c_ssim * (1 - sums[6] / (3 H W)) + reg_out[3]   (NaN when the forward on `geom` overflowed)
 *   gsr_map_loss_backward  dL_dimage [3,H,W] = the SSIM term's gradient (*neg_c_ssim: DEVICE scalar, -c_ssim) + the colour L1 term's;
 *                          dL_ddepth [H,W] (NULL: not wanted) = the masked depth term's (divides by sums[2]) */
int gsr_map_loss_forward(const float* image, const float* depth, const float* sur, const float* sil, const float* frame_rgb, const float* frame_depth,
                         int H, int W, const float* taps11 /* host */, float sil_thr, float* partial6, float* dmaps, void* stream);
int gsr_map_loss_finish(const float* partial6, const float* reg_partial, size_t n_gaussians, int H, int W, const float* w3 /* host */, float c_ssim,
                        float w_long, float w_scalar, const char* geom, float* sums, float* reg_out, float* loss, void* stream);
int gsr_map_loss_backward(const float* image, const float* depth, const float* frame_rgb, const float* frame_depth, const float* dmaps, int H, int W,
                          const float* taps11 /* host */, const float* w3 /* host */, const float* neg_c_ssim, const float* sums, float* dL_dimage,
                          float* dL_ddepth, void* stream);
typedef struct gsr_pose_update_args {
    float* quat_trans;    /* [7] */
    float* moments;       /* [14] */
    float* best;          /* [8] */
    float* history;       /* [1]: this iteration's slot */
    float* Tcw;           /* [16] */
    const float* partial; /* [GSR_POSE_PARTIALS][12] */
    const float* loss;    /* [1] */
    const char* geom;     /* NULL: no overflow predicate */
    double lr, beta1, beta2, eps;
    int step;
    float* skip;          /* NULL: none. One DEVICE float: a non-zero value makes the step behave as after an overflowed forward (NaN recorded, no step).
                           * A sharded loop all-reduces the ranks' overflow flags into it (gsr_pose_step_args.overflow_out), so that every rank takes
                           * the same decision; gsr_pose_finish leaves it zero. (ABI 9) */
} gsr_pose_update_args;
int gsr_pose_update(const gsr_pose_update_args* args, void* stream);
/* gsr_pose_grad (the twelve pose sums of dL/dmeans_cam against the world-frame means) and gsr_pose_update in ONE launch: the workgroup of the sums
 * that finishes last takes the step. args->partial: scratch [GSR_POSE_PARTIALS][12]; ticket: GSR_TICKET_WORDS DEVICE words that are zero between calls (zero them once). */
#define GSR_TICKET_WORDS 144
/* gsr_backward_args.fused_pose_step */
typedef struct gsr_pose_step_args {
    const float* means_world;                  /* [P,3] the world-frame means (the rasterizer's means3D are their camera-frame images) */
    const struct gsr_pose_update_args* update; /* as for gsr_pose_update; update->partial: 64 * 12 floats that are ZERO between calls (zero them once:
                                                * the workgroups of the per-splat stage add their sums there, the step leaves them zero) */
    int sums_only;                             /* 1: the backward only ADDS the pose sums to update->partial's rows and takes no step (a sharded run sums the
                                                * ranks' rows first: all-reduce them, then gsr_pose_finish) */
    float* overflow_out;                       /* NULL, or one DEVICE float that receives this rank's overflow flag of the forward (0 / 1): placed behind the
                                                * rows, it travels in their all-reduce and comes back as gsr_pose_update_args.skip (ABI 9) */
} gsr_pose_step_args;
/* The step behind a backward with fused_pose_step.sums_only: adds the 64 accumulator rows up, takes gsr_pose_update's step with the total and leaves
 * the rows zero; sums_out (NULL or 12 DEVICE floats): the twelve pose sums the step used (dL/dR row-major, dL/dt). One single-wave launch. */
int gsr_pose_finish(const struct gsr_pose_update_args* args, float* acc_rows, float* sums_out, void* stream);
int gsr_pose_step(const float* means3D, const float* dL_dmeans_cam, size_t n, const gsr_pose_update_args* args, uint32_t* ticket, void* stream);
/* The feature reprojection term of the tracking loss, weight * Lrpj (src/Render.cc:1031-1096, _featureWeightTracking; Examples/RGB-D/replica.yaml: 0.1):
 * Lrpj = sum over the inlier matches of inv_sigma2 * |K (Xc / Xc.z) - obs|^2, Xc = R Xw + t under the DEVICE pose Tcw (the reference freezes its
 * inliers — chi-square < 5.991 — halfway through the iterations). The term depends on the pose alone and its gradient has the pose sums' own form, so it
 * is ADDED to what the pose step reads: pose_row [12] += weight * grad_scale * (dLrpj/dR row-major, dLrpj/dt) — any row of gsr_pose_grad's partial rows
 * (after that call) or of the fused pose step's accumulator rows (before gsr_backward) — and loss[0] += weight * Lrpj (gsr_track_loss's sums + 5, after
 * that call). grad_scale: 1, or 1 / world when the ranks of a sharded loop each add the term and their rows are then summed.
 * obs [M,2] pixels, Xw [M,3] world points, inv_sigma2 [M]; refresh_inliers: 0 use inliers [M] as stored, 1 recompute them from the current
 * errors and store them, 2 every match counts (inliers may be NULL). One launch, nothing allocates or synchronises. (ABI 7) */
int gsr_reproj_loss(const float* obs, const float* Xw, const float* inv_sigma2, size_t M, const float* Tcw, float fx, float fy, float cx, float cy,
                    float weight, float grad_scale, int refresh_inliers, uint8_t* inliers, float* pose_row, float* loss, void* stream);

/* ---- multi-GPU scheme B (scene shards; gsorb-slam_amd/sharded.py): compositing of the ranks' layers around the
 * two collectives of the forward and the one of the backward. The reference is single-GPU; north_star: "shard Gaussians across the GPUs,
 * RCCL all-reduce on pose / loss gradients only". All pointers are DEVICE pointers; N = H * W.
 *   gathered [world,gathered_planes,H,W]  the all-gather of every rank's (silhouette S, surface depth[, a plane that carries the order key]), in RANK order
 *   order    [world] int64  the ranks front to back (argsort of the keys)
 *   layer4   [4,H,W]        this rank's (rgb, depth) render
 * gsr_composite_forward: contrib [4,H,W] = P_own * layer4 with P_own = prod_{layers in front} (1 - S) (the caller all-reduces it),
 *   sil_total [H,W] = 1 - prod_all (1 - S), surf [H,W] (NULL: not wanted) = surface depth of the first layer, front to back, behind
 *   which the accumulated transmittance is <= 0.5 (else of the last layer that has one; has_sur = 0: zeros).
 * gsr_composite_backward_local: d_layer4 = P_own * g4 (g4 NULL: zeros) and c_own [H,W] = g4 . layer4, which the caller all-gathers.
 * gsr_composite_backward_occlusion: dS [H,W] = - sum_{k behind own} (prod_{h before k, h != own} (1 - S_h)) c_k
 *   + g_sil * prod_{h != own} (1 - S_h)   (c_all [world,H,W] in rank order; g_sil NULL: no silhouette gradient). */
int gsr_composite_forward(int world, int rank, const long long* order, const float* gathered, int gathered_planes, const float* layer4, int H, int W, int has_sur,
                          float* contrib, float* sil_total, float* surf, void* stream);
int gsr_composite_backward_local(int world, int rank, const long long* order, const float* gathered, int gathered_planes, const float* layer4, const float* g4, int H,
                                 int W, float* d_layer4, float* c_own, void* stream);
int gsr_composite_backward_occlusion(int world, int rank, const long long* order, const float* gathered, int gathered_planes, const float* c_all, const float* g_sil,
                                     int H, int W, float* dS, void* stream);
/* ---- round 6 (ABI 9): the BAND exchange of the sharded loop. Rank r composites, evaluates the loss and differentiates the composite on ITS band of pixel
 * rows only, for every rank's layer: all per-pixel work / world. Rows are image rows; [row_begin, row_end) is the rank's band.
 *   layers_all [world][6][H][W]  every rank's layer (rgb, depth, silhouette, surface depth) in RANK order, valid on the rows this rank received; own_layer [6][H][W]
 *                                stands in for layers_all[rank] (NULL layers_all with world 1)
 * gsr_band_composite_forward: rows [row_begin - halo, row_end + halo) inside the image: out_rgbd planes 0..2 = sum_k P_k rgb_k; on the band itself also plane 3
 *   (depth), out_sil = 1 - prod (1 - S), out_sur = gsr_composite_forward's surface depth. out_rgbd [4][H][W], out_sil, out_sur [H][W]: only those rows are written.
 * gsr_band_composite_backward: g4 [4][H][W] (d/d rgb, depth of the composite; band rows) -> for every rank k its layer's gradient on the band rows:
 *   d_all [world][5][H][W] = {P_k g4, dS_k} (this rank's own goes to d_own [5][H][W]); dS_k is gsr_composite_backward_occlusion's term; g_sil [H][W] (NULL: none) is the
 *   upstream gradient of the stack's silhouette.
 * gsr_shard_map_totals: rows [world][16] = every rank's {sums[8], reg_out[4], loss slot (NaN: that rank's forward overflowed), 3 unused} from
 *   gsr_map_loss_finish_rows on its band -> sums [8] / reg_out [4] (NULL: none) of the whole frame / map and the iteration's loss. */
int gsr_band_composite_forward(int world, int rank, const long long* order, const float* layers_all, const float* own_layer, int H, int W, int row_begin, int row_end,
                               int halo, float* out_rgbd, float* out_sil, float* out_sur, void* stream);
int gsr_band_composite_backward(int world, int rank, const long long* order, const float* layers_all, const float* own_layer, const float* g4, const float* g_sil, int H,
                                int W, int row_begin, int row_end, float* d_all, float* d_own, void* stream);
int gsr_shard_map_totals(int world, const float* rows, int H, int W, const float* w3 /* host */, float c_ssim, float w_long, float w_scalar, float* sums, float* reg_out,
                         float* loss, void* stream);
/* The loss kernels on a band of rows (the whole image: 0, H — what gsr_map_loss_forward / _finish / _backward / gsr_track_loss are). Planes keep their full
 * [.,H,W] layout; only the band's rows are read (the SSIM window reaches ten rows beyond: the caller provides `image` there) and written.
 *   gsr_map_loss_forward_rows   sums over the band's rows; derivative maps for the band and five rows either side; partial6 [6][gsr_map_loss_partials_rows(...)]
 *   gsr_map_loss_finish_rows    the band's raw sums (sums[5], the pixel terms' value, divides by the BAND's counts: combine the ranks' rows with gsr_shard_map_totals)
 *   gsr_map_loss_backward_rows  gradient planes on the band's rows; sums[2] must be the WHOLE frame's count of valid depth pixels
 *   gsr_track_loss_rows         masked L1 sums and gradient planes on the band's rows */
size_t gsr_map_loss_partials_rows(int H, int W, int row_begin, int row_end);
int gsr_map_loss_forward_rows(const float* image, const float* depth, const float* sur, const float* sil, const float* frame_rgb, const float* frame_depth,
                              int H, int W, const float* taps11 /* host */, float sil_thr, float* partial6, float* dmaps, int row_begin, int row_end, void* stream);
int gsr_map_loss_finish_rows(const float* partial6, const float* reg_partial, size_t n_gaussians, int H, int W, const float* w3 /* host */, float c_ssim,
                             float w_long, float w_scalar, const char* geom, float* sums, float* reg_out, float* loss, int row_begin, int row_end, void* stream);
int gsr_map_loss_backward_rows(const float* image, const float* depth, const float* frame_rgb, const float* frame_depth, const float* dmaps, int H, int W,
                               const float* taps11 /* host */, const float* w3 /* host */, const float* neg_c_ssim, const float* sums, float* dL_dimage,
                               float* dL_ddepth, int row_begin, int row_end, void* stream);
int gsr_track_loss_rows(const float* image, const float* depth, const float* sur, const float* sil, const float* frame_rgb, const float* frame_depth,
                        int H, int W, float sil_thr, const float* w3 /* host */, float* partial, float* sums, float* dL_dimage, float* dL_ddepth,
                        uint32_t* ticket, int row_begin, int row_end,
                        int sil_is_transmittance /* non-zero: `sil` holds the render's final transmittance T (gsr_transmittance_view), the silhouette is 1 - T */, void* stream);
/* The per-pixel final transmittance [H,W] the last forward on `image` left in the image blob (forward.cu's accum_alpha / final_T): 1 - T is the silhouette the
 * fused pair's second channel accumulates — a tracking iteration that needs nothing else of the pair renders the plain 3 channels and masks with this. */
int gsr_transmittance_view(char* image, int width, int height, float** final_T);

/* Front-to-back order of the cells of a k-d partition of the map (gsorb-slam_amd/sharded.py: KdPartition; one cell per rank) for the camera of
 * Tcw (DEVICE, row-major 4x4 world -> camera): order [world] (DEVICE int64) = the ranks, nearest cell first. The leaves of a BSP are ordered
 * exactly by visiting the side of every split that holds the camera centre first — for ANY view, unlike an order by nearest depth.
 * kd_nodes [world - 1][4] (DEVICE floats) = {axis 0..2, split, left, right}; a child >= 0 is a node index, a child < 0 the leaf (rank) -1 - child;
 * node 0 is the root. world <= 32. The reference renders on one GPU (src/Render.cc:927-981): nothing to compare with. */
int gsr_shard_order(int world, const float* kd_nodes, const float* Tcw, long long* order, void* stream);

/* Workspace sizes: replace required<GeometryState/ImageState/BinningState>
 * (rasterizer_impl.h:67-73). */
size_t gsr_geom_bytes(int P);
size_t gsr_image_bytes(int width, int height);
size_t gsr_binning_bytes(size_t num_rendered);

/* Test/inspection hook: re-expresses the opaque blobs in the reference's
 * GeometryState/BinningState/ImageState array layout (rasterizer_impl.h:21-65).
 * Every destination is a device pointer and may be NULL. Blocking. */
typedef struct gsr_debug_arrays {
    float* means2D;          /* [P,2] */
    float* depths;           /* [P] */
    float* conic_opacity;    /* [P,4] */
    float* rgb;              /* [P,3] colours the blend used */
    uint32_t* tiles_touched; /* [P] */
    uint32_t* point_list;    /* [R] sorted splat ids */
    uint64_t* point_list_keys; /* [R] (tile<<32 | depth bits), rebuilt */
    uint32_t* ranges;        /* [tiles,2] */
    float* final_T;          /* [H*W] */
    uint32_t* n_contrib;     /* [H*W] */
} gsr_debug_arrays;
int gsr_debug_export(int P, int width, int height, int R, const char* geom, const char* binning,
                     const char* image, const gsr_debug_arrays* out, void* stream);

/* Inspection: kernel launches this library has issued in this process so far, over all threads (a statistics counter, read by nothing on
 * a data path; bench.py reports the launches per step / per loop iteration from it). hipMemsetAsync nodes are not counted. */
unsigned long long gsr_debug_launch_count(void);

/* Text for a GSR_E* code; the HIP error string of the last failing HIP call on this thread. */
const char* gsr_error_string(int code);
const char* gsr_last_hip_error(void);
int gsr_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* GSR_H */
