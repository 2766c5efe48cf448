// gsr_api.hip — host side of the C ABI declared in include/gsr.h: argument checks,
// workspace carving and kernel launches. No global mutable state (only a
// thread-local "last HIP error" string); everything runs on the caller's stream.
#include "../../include/gsr.h"
#include "gsr_kernels.hip"
#include "gsr_knn.h"
#include "gsr_train.h"
#include "gsr_shard.h"

#include <string.h>
#include <algorithm>

namespace {

thread_local hipError_t t_last_hip = hipSuccess;

inline bool hip_ok(hipError_t e)
{
    if (e != hipSuccess) { t_last_hip = e; return false; }
    return true;
}
#define GSR_HIP(x) do { if (!hip_ok(x)) return GSR_EHIP; } while (0)
#define GSR_LAUNCHED() do { if (!hip_ok(hipGetLastError())) return GSR_EHIP; } while (0)
// every kernel launch of the library goes through here: a process-wide statistics counter (gsr_debug_launch_count: bench.py states the launches per
// step / per loop iteration from it), nothing reads it on any data path
unsigned long long g_launches = 0;
#define GSR_LAUNCH(...) do { __atomic_fetch_add(&g_launches, 1ull, __ATOMIC_RELAXED); hipLaunchKernelGGL(__VA_ARGS__); } while (0)

inline int blocks256(int n) { return (n + 255) / 256; }

// optional per-stage event pairs (profiling hook of the args structs)
struct StageTimer {
    void** ev; hipStream_t st;
    void begin(int i) const { if (ev && ev[2 * i]) (void)hipEventRecord((hipEvent_t)ev[2 * i], st); }
    void end(int i) const { if (ev && ev[2 * i + 1]) (void)hipEventRecord((hipEvent_t)ev[2 * i + 1], st); }
};

FrameParams frame_params(int P, int D, int M, int W, int H, float tfx, float tfy, float mod, int by0 = 0, int by1 = 0)
{
    FrameParams f;
    f.P = P; f.D = D; f.M = M; f.W = W; f.H = H;
    f.grid_x = (W + GSR_TILE - 1) / GSR_TILE;
    f.grid_y = (H + GSR_TILE - 1) / GSR_TILE;
    f.tan_fovx = tfx; f.tan_fovy = tfy;
    f.focal_y = H / (2.0f * tfy); // rasterizer_impl.cu:227-228
    f.focal_x = W / (2.0f * tfx);
    f.scale_modifier = mod;
    if (by0 == 0 && by1 == 0) by1 = f.grid_y;
    f.band_y0 = std::max(0, std::min(by0, f.grid_y));
    f.band_y1 = std::max(f.band_y0, std::min(by1, f.grid_y));
    f.fold_depth_color = 1;
    return f;
}

int check_forward(const gsr_forward_args* a)
{
    if (!a || a->P < 0 || a->width <= 0 || a->height <= 0) return GSR_EINVAL;
    if (a->P > (int)GSR_ID_MASK) return GSR_EINVAL; // the quad-hit log keeps ids in 28 bits (268 M splats = 30 GB of geometry blob)
    if ((int64_t)((a->width + 15) / 16) * ((a->height + 15) / 16) > (int64_t)GSR_BIN_NWIN * (GSR_BIN_WINDOW - 2)) return GSR_EINVAL; // a tile window is counted in one CU's LDS
    if (!a->out_color || !a->out_depth) return GSR_EINVAL;
    if (a->P == 0) return GSR_OK;
    if (!a->means3D || !a->opacities || !a->viewmatrix || !a->projmatrix || !a->background) return GSR_EINVAL;
    const bool has_sh = a->shs != nullptr, has_col = a->colors_precomp != nullptr;
    if (has_sh == has_col) return GSR_EINVAL; // exactly one (include/Rasterizer.cuh:310-312)
    if (has_sh && (a->M <= 0 || a->D < 0 || a->D > 3 || (a->D + 1) * (a->D + 1) > a->M || !a->cam_pos)) return GSR_EINVAL;
    const bool has_sr = a->scales != nullptr && a->rotations != nullptr, has_cov = a->cov3D_precomp != nullptr;
    if (has_sr == has_cov) return GSR_EINVAL; // exactly one (:313-316)
    if (a->out_sil && a->out_ds) return GSR_EINVAL; // (the fused pair renders the silhouette itself: out_ds[1])
    if (a->pre_Tcw && !a->means_cam_out) return GSR_EINVAL; // (the camera-frame means must go somewhere: the backward takes them)
    if (a->raw && (!has_sr || !a->raw->opacities || !a->raw->scales || !a->raw->rotations)) return GSR_EINVAL;
    return GSR_OK;
}

gsr::SplatInputs splat_inputs(const float* means3D, const float* scales, const float* rotations,
                              const float* opacities, const float* shs, const float* cov3D,
                              const float* colors, const float* view, const float* proj, const float* campos)
{
    gsr::SplatInputs in;
    in.means3D = means3D; in.scales = scales; in.rotations = rotations; in.opacities = opacities;
    in.shs = shs; in.cov3D_precomp = cov3D; in.colors_precomp = colors;
    in.view = view; in.proj = proj; in.campos = campos;
    in.pre_Tcw = nullptr; in.means_cam_out = nullptr;
    in.opac_out = in.scales_out = in.rots_out = nullptr; in.reg_limit = 0.f; in.reg_partial = nullptr;
    return in;
}

#define GSR_FILL_WX GSR_BIN_NWIN // the binning passes run one workgroup per (splat range, tile window): the layout of the bucketed records
// launch shape of the two binning passes: one workgroup per splat range x tile window
struct BinGrid {
    int rows, per, wx, nwin, twmax;
    dim3 grid;
};
BinGrid bin_grid(int P, int T, int wx, int window)
{
    BinGrid b;
    b.rows = bin_rows(P);
    b.per = ((P + b.rows - 1) / b.rows + GSR_BIN_PIECE - 1) / GSR_BIN_PIECE * GSR_BIN_PIECE; // whole pieces of the bucketed records (K_preprocess workgroups)
    b.rows = (P + b.per - 1) / b.per;
    b.wx = wx;
    (void)window;
    b.nwin = wx; // (check_forward refuses frames of more than GSR_BIN_NWIN * (GSR_BIN_WINDOW - 2) tiles)
    b.grid = dim3(b.rows * wx, 1);
    b.twmax = ((T + b.nwin - 1) / b.nwin + 2) & ~1; // even: the staged keys behind the per-tile words stay 8-byte aligned
    return b;
}

int forward_tail(const gsr_forward_args* a, const GeomView& gv, const ImageView& iv, const BinView& bv,
                 const FrameParams& f, hipStream_t st)
{
    const int P = a->P, T = f.grid_x * f.grid_y;
    const StageTimer tm{a->profile_events, st};
    tm.begin(GSR_FWD_FILL);
    const BinGrid bg = bin_grid(P, T, GSR_FILL_WX, GSR_BIN_WINDOW);
    GSR_LAUNCH(gsr::K_bin_fill, bg.grid, dim3(GSR_BINF_THREADS), (size_t)bg.twmax * 4, st, P, bg.per, T, f.grid_x, bg.wx, bg.nwin, gv,
                       iv.binmat, iv.tile_start, bv.pairs);
    GSR_LAUNCHED();
    tm.end(GSR_FWD_FILL);
    tm.begin(GSR_FWD_SORT);
#ifdef GSR_EXP_MIDSORT // timing experiment: the keys-only 16-keys-per-thread sort of the lists over 1024 entries, run beside the real one
    GSR_LAUNCH(gsr::K_tile_sort<GSR_SORT_BLOCK>, dim3(T), dim3(256), 0, st, T, iv.ranges, gv.hdr, bv.pairs, bv.point_list);
#endif
    GSR_LAUNCH(gsr::K_tile_sort_cut, dim3(T), dim3(GSR_SORT_CUT_THREADS), 0, st, T, f.grid_x, iv.ranges, gv, bv.pairs, bv.point_list,
                       bv.qhits, iv.qcount);
    GSR_LAUNCHED();
    tm.end(GSR_FWD_SORT);
    tm.begin(GSR_FWD_BLEND);
    const int Tb = (f.band_y1 - f.band_y0) * f.grid_x; // tiles of the band
    if (Tb > 0 && a->out_ds)
        GSR_LAUNCH((gsr::K_blend_fwd<GSR_ROWQ, true>), dim3(4 * Tb), dim3(64), 0, st, iv, bv, gv, a->background, a->width, a->height,
                           f.grid_x, Tb, f.band_y0 * f.grid_x, a->out_color, a->out_depth, P, a->out_ds, (float*)nullptr);
    else if (Tb > 0)
        GSR_LAUNCH((gsr::K_blend_fwd<GSR_ROWQ, false>), dim3(4 * Tb), dim3(64), 0, st, iv, bv, gv, a->background, a->width, a->height,
                           f.grid_x, Tb, f.band_y0 * f.grid_x, a->out_color, a->out_depth, P, (float*)nullptr, a->out_sil);
    else // an empty band launches no blend kernel: clear the backward accumulators here
        GSR_HIP(hipMemsetAsync(gv.acc, 0, (size_t)P * GSR_ACC_STRIDE * sizeof(float), st));
    GSR_LAUNCHED();
    tm.end(GSR_FWD_BLEND);
    return GSR_OK;
}

int forward_head(const gsr_forward_args* a, char* geom, char* image, hipStream_t st, uint32_t capacity,
                 GeomView* gv, ImageView* iv, FrameParams* fo)
{
    const int P = a->P, W = a->width, H = a->height;
    const FrameParams f = frame_params(P, a->D, a->M, W, H, a->tan_fovx, a->tan_fovy, a->scale_modifier, a->band_y0, a->band_y1);
    const int T = f.grid_x * f.grid_y;
    geom_layout(geom, P, gv);
    image_layout(image, W, H, iv);
    gsr::SplatInputs in = splat_inputs(a->means3D, a->scales, a->rotations, a->opacities, a->shs,
                                       a->cov3D_precomp, a->colors_precomp, a->viewmatrix,
                                       a->projmatrix, a->cam_pos);
    in.pre_Tcw = a->pre_Tcw; in.means_cam_out = a->means_cam_out;
    const StageTimer tm{a->profile_events, st};
    tm.begin(GSR_FWD_PREPROCESS);
    if (a->raw) {
        in.opac_out = a->raw->opacities; in.scales_out = a->raw->scales; in.rots_out = a->raw->rotations; in.reg_limit = a->raw->reg_limit; in.reg_partial = a->raw->reg_partial;
        GSR_LAUNCH(gsr::K_preprocess<true>, dim3((P + GSR_PRE_THREADS - 1) / GSR_PRE_THREADS), dim3(GSR_PRE_THREADS), 0, st, f, in, a->radii, *gv);
    } else
        GSR_LAUNCH(gsr::K_preprocess<false>, dim3((P + GSR_PRE_THREADS - 1) / GSR_PRE_THREADS), dim3(GSR_PRE_THREADS), 0, st, f, in, a->radii, *gv);
    GSR_LAUNCHED();
    tm.end(GSR_FWD_PREPROCESS);
    tm.begin(GSR_FWD_SCAN);
    const BinGrid bg = bin_grid(P, T, GSR_FILL_WX, GSR_BIN_WINDOW);
    if ((size_t)bg.twmax * 4 > 65536) { // (only frames beyond 131 000 tiles need more than the default limit of dynamic LDS)
        GSR_HIP(hipFuncSetAttribute((const void*)gsr::K_bin_count, hipFuncAttributeMaxDynamicSharedMemorySize, bg.twmax * 4));
        GSR_HIP(hipFuncSetAttribute((const void*)gsr::K_bin_fill, hipFuncAttributeMaxDynamicSharedMemorySize, bg.twmax * 4));
    }
    GSR_LAUNCH(gsr::K_bin_count, bg.grid, dim3(GSR_BINC_THREADS), (size_t)bg.twmax * 4, st, P, bg.per, T, f.grid_x, bg.wx, bg.nwin, *gv, iv->binmat);
#ifdef GSR_SEPARATE_SCAN // (the two-launch form)
    GSR_LAUNCH(gsr::K_bin_colscan, dim3((T + 31) / 32), dim3(1024), 0, st, bg.rows, T, iv->binmat, iv->tile_cnt, (uint32_t*)nullptr, (uint2*)nullptr, gv->hdr, capacity);
    GSR_LAUNCH(gsr::K_scan_tiles, dim3(1), dim3(1024), 0, st, T, iv->tile_cnt, 1, iv->tile_start, 1, iv->ranges, gv->hdr, capacity);
#else
    GSR_LAUNCH(gsr::K_bin_colscan, dim3((T + 31) / 32), dim3(1024), 0, st, bg.rows, T, iv->binmat, iv->tile_cnt, iv->tile_start, iv->ranges, gv->hdr, capacity);
#endif
    GSR_LAUNCHED();
    tm.end(GSR_FWD_SCAN);
    *fo = f;
    return GSR_OK;
}

int forward_empty(const gsr_forward_args* a, char* geom, hipStream_t st)
{
    // the reference wrapper returns zero images without calling the core (src/Rasterizer.cu:183)
    const size_t N = (size_t)a->width * a->height;
    GSR_HIP(hipMemsetAsync(a->out_color, 0, N * 3 * sizeof(float), st));
    GSR_HIP(hipMemsetAsync(a->out_depth, 0, N * sizeof(float), st));
    if (a->out_ds) GSR_HIP(hipMemsetAsync(a->out_ds, 0, N * 2 * sizeof(float), st));
    if (a->out_sil) GSR_HIP(hipMemsetAsync(a->out_sil, 0, N * sizeof(float), st));
    if (geom) GSR_HIP(hipMemsetAsync(geom, 0, sizeof(GeomHeader), st));
    return GSR_OK;
}

} // namespace

extern "C" {

size_t gsr_geom_bytes(int P) { return geom_layout(nullptr, P, nullptr); }
size_t gsr_image_bytes(int width, int height) { return image_layout(nullptr, width, height, nullptr); }
size_t gsr_binning_bytes(size_t R) { return binning_layout(nullptr, R, nullptr); }
int gsr_abi_version(void) { return GSR_ABI_VERSION; }
unsigned long long gsr_debug_launch_count(void) { return __atomic_load_n(&g_launches, __ATOMIC_RELAXED); }

const char* gsr_error_string(int code)
{
    switch (code) {
    case GSR_OK: return "ok";
    case GSR_EINVAL: return "invalid argument";
    case GSR_EALLOC: return "allocation callback returned NULL";
    case GSR_EHIP: return "HIP runtime error";
    case GSR_EOVERFLOW: return "binning workspace too small";
    case GSR_ECHANNELS: return "only 3 colour channels are supported";
    default: return code > 0 ? "ok" : "unknown error";
    }
}
const char* gsr_last_hip_error(void) { return hipGetErrorString(t_last_hip); }

// One pinned word + one event per host thread AND device (never freed: a thread renders for the life of the
// process). An event belongs to the device that was current when it was created: a thread that renders on
// cuda:0 and then on cuda:1 (RasterizeGaussiansCUDA takes device_num; the reference defines GPU0..GPU2) must not
// record device 0's event on a stream of device 1, so the slot is looked up by hipGetDevice.
struct Staging {
    uint32_t* host = nullptr;
    hipEvent_t ev = nullptr;
    bool tried = false;
    uint32_t last_R = 0; // tile instances of this thread's previous frame on this device (capacity guess)
    bool ready()
    {
        if (!tried) {
            tried = true;
            if (hipHostMalloc((void**)&host, 64, hipHostMallocDefault) != hipSuccess) host = nullptr;
            if (host && hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { (void)hipHostFree(host); host = nullptr; }
        }
        return host != nullptr;
    }
};
#define GSR_MAX_DEVICES 16
static Staging* staging_slot()
{
    static thread_local Staging t_stage[GSR_MAX_DEVICES + 1];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= GSR_MAX_DEVICES) dev = GSR_MAX_DEVICES; // shared overflow slot: blocking read only
    return &t_stage[dev];
}

int gsr_forward(const gsr_forward_args* a, gsr_alloc_fn geom_alloc, void* geom_user,
                gsr_alloc_fn binning_alloc, void* binning_user, gsr_alloc_fn image_alloc,
                void* image_user, void* stream)
{
    const int chk = check_forward(a);
    if (chk != GSR_OK) return chk;
    if (!geom_alloc || !binning_alloc || !image_alloc) return GSR_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    char* geom = geom_alloc(geom_user, gsr_geom_bytes(a->P));
    if (!geom) return GSR_EALLOC;
    char* image = image_alloc(image_user, gsr_image_bytes(a->width, a->height));
    if (!image) return GSR_EALLOC;
    if (a->P == 0) {
        if (!binning_alloc(binning_user, gsr_binning_bytes(0))) return GSR_EALLOC;
        const int rc = forward_empty(a, geom, st);
        return rc != GSR_OK ? rc : 0;
    }
    // The binning blob must be sized before num_rendered is known on the host (the one device->host read of
    // the forward, reference rasterizer_impl.cu:285). Instead of stalling the GPU while the host waits for
    // that number, the blob is requested with a capacity guessed from this thread's previous frames and the
    // tail kernels are enqueued right behind the head; the host then waits only for the head. A guess that
    // turns out too small costs one more request and a second tail (its first run exits on the overflow flag).
    // Guess: 1.25x this thread's previous frame on this device, at least P (a first frame of fat splats pays the
    // second request once; 44 B per instance: a floor of 4P would pin 176 MB at 1 M splats whatever R is).
    Staging& t_stage = *staging_slot();
    const uint32_t floor_c = (uint32_t)std::min<size_t>((size_t)a->P + 4096, 0x7FFFFFFFu);
    const uint32_t guess = std::max(floor_c, (uint32_t)std::min<size_t>((size_t)t_stage.last_R + t_stage.last_R / 4 + 4096, 0x7FFFFFFFu));
    char* binning = binning_alloc(binning_user, gsr_binning_bytes(guess));
    if (!binning) return GSR_EALLOC;
    GeomView gv; ImageView iv; FrameParams f;
    int rc = forward_head(a, geom, image, st, guess, &gv, &iv, &f);
    if (rc != GSR_OK) return rc;
    uint32_t R = 0;
    int cur_dev = 0;
    bool staged = hipGetDevice(&cur_dev) == hipSuccess && cur_dev >= 0 && cur_dev < GSR_MAX_DEVICES && t_stage.ready();
    if (staged) {
        GSR_HIP(hipMemcpyAsync(t_stage.host, &gv.hdr->num_rendered, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        if (hipEventRecord(t_stage.ev, st) != hipSuccess) { // e.g. the stream belongs to another device than the current one
            (void)hipGetLastError();
            GSR_HIP(hipStreamSynchronize(st));
            R = *t_stage.host;
            staged = false;
        }
    } else { // no pinned staging word: plain blocking read
        GSR_HIP(hipMemcpyAsync(&R, &gv.hdr->num_rendered, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        GSR_HIP(hipStreamSynchronize(st));
    }
    BinView bv;
    binning_layout(binning, guess, &bv);
    rc = forward_tail(a, gv, iv, bv, f, st);
    if (rc != GSR_OK) return rc;
    if (staged) {
        GSR_HIP(hipEventSynchronize(t_stage.ev));
        R = *t_stage.host;
    }
    if (R > 0x7FFFFFFFu) return GSR_EOVERFLOW;
    t_stage.last_R = R;
    if (R > guess) {
        binning = binning_alloc(binning_user, gsr_binning_bytes(R));
        if (!binning) return GSR_EALLOC;
        GSR_LAUNCH(gsr::K_set_capacity, dim3(1), dim3(1), 0, st, gv.hdr, R);
        GSR_LAUNCHED();
        binning_layout(binning, R, &bv);
        rc = forward_tail(a, gv, iv, bv, f, st);
        if (rc != GSR_OK) return rc;
    }
    return (int)R;
}

int gsr_forward_ws(const gsr_forward_args* a, char* geom, char* binning, size_t binning_bytes,
                   char* image, void* stream)
{
    const int chk = check_forward(a);
    if (chk != GSR_OK) return chk;
    if (!geom || !binning || !image) return GSR_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (a->P == 0) return forward_empty(a, geom, st);
    const size_t cap = binning_capacity(binning_bytes);
    if (cap == 0) return GSR_EOVERFLOW;
    GeomView gv; ImageView iv; FrameParams f;
    int rc = forward_head(a, geom, image, st, (uint32_t)(cap > 0x7FFFFFFFu ? 0x7FFFFFFFu : cap), &gv, &iv, &f);
    if (rc != GSR_OK) return rc;
    BinView bv;
    binning_layout(binning, cap, &bv); // layout fixed by the capacity, not by R
    return forward_tail(a, gv, iv, bv, f, st);
}

int gsr_ws_status(const char* geom, void* stream, int* num_rendered, int* overflow)
{
    if (!geom) return GSR_EINVAL;
    GeomHeader h;
    GSR_HIP(hipMemcpyAsync(&h, geom, sizeof(uint32_t) * 4, hipMemcpyDeviceToHost, (hipStream_t)stream));
    GSR_HIP(hipStreamSynchronize((hipStream_t)stream));
    if (num_rendered) *num_rendered = (int)h.num_rendered;
    if (overflow) *overflow = (int)h.overflow;
    return GSR_OK;
}

namespace {
int make_pose_update(const gsr_pose_update_args* a, gsr::PoseUpdate* out);
const uint32_t* overflow_flag(const char* geom)
{
    return geom ? &reinterpret_cast<const GeomHeader*>(geom)->overflow : nullptr;
}
// gsr_map_update_args -> the kernels' view of it; grads = false: the gradient tensors are not needed (gsr_backward_args.fused_map_update)
int make_map_update(const gsr_map_update_args* a, bool grads, gsr::MapUpdate* out)
{
    if (!a->xyz || !a->rgb || !a->unnorm_quat || !a->logit || !a->log_scales || !a->opacities || !a->scales || !a->Tcw || (a->n + 255) / 256 > 0x7FFFFFFFu)
        return GSR_EINVAL;
    if (grads && (!a->dL_dmeans_cam || !a->dL_dcolors || !a->dL_drotations || !a->dL_dopacities || !a->dL_dscales)) return GSR_EINVAL;
    gsr::MapUpdate u;
    u.xyz = a->xyz; u.rgb = a->rgb; u.quat = a->unnorm_quat; u.logit = a->logit; u.ls = a->log_scales;
    for (int g = 0; g < 5; g++) {
        if (!a->exp_avg[g] || !a->exp_avg_sq[g] || a->step[g] < 1) return GSR_EINVAL;
        u.m[g] = a->exp_avg[g]; u.v[g] = a->exp_avg_sq[g];
        // (bias corrections in double, like gsr_adam_step)
        const double bc1 = 1.0 - std::pow(a->beta1, (double)a->step[g]), bc2 = 1.0 - std::pow(a->beta2, (double)a->step[g]);
        u.step_size[g] = (float)(a->lr[g] / bc1); u.sqrt_bias2[g] = (float)std::sqrt(bc2);
    }
    u.dmc = a->dL_dmeans_cam; u.dcol = a->dL_dcolors; u.drot = a->dL_drotations; u.dopac = a->dL_dopacities; u.dscale = a->dL_dscales;
    u.opac = a->opacities; u.scales = a->scales; u.Tcw = a->Tcw; u.reg_out = a->reg_out; u.overflow = overflow_flag(a->geom);
    u.limit = a->reg_limit; u.w_long = a->w_long; u.w_scalar = a->w_scalar;
    u.w1 = (float)(1.0 - a->beta1); u.b2 = (float)a->beta2; u.w2 = (float)(1.0 - a->beta2); u.eps = (float)a->eps;
    *out = u;
    return GSR_OK;
}
} // namespace

int gsr_backward(const gsr_backward_args* a, void* stream)
{
    if (!a || a->P < 0 || a->width <= 0 || a->height <= 0) return GSR_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    // The fused steps are validated BEFORE anything is launched: an EINVAL must leave the accumulators as the forward left them (a retried
    // backward would otherwise count the blend stage twice).
    if (a->fused_map_update && a->fused_pose_step) return GSR_EINVAL; // one or the other (include/gsr.h)
    // dds_depth_only == 2 (dL_dds = the silhouette's plane alone) exists only in the form without colour sums and with the depth channel's colour detached
    if (a->dL_dds && a->dds_depth_only == 2 &&
        (a->dL_dcolor || a->dL_dsh || a->fused_map_update || !a->ds_detach_depth || !((a->stages ? a->stages : GSR_STAGE_SPLAT) & GSR_STAGE_SPLAT))) return GSR_EINVAL;
    if (a->dds_depth_only < 0 || a->dds_depth_only > 2) return GSR_EINVAL;
    gsr::MapUpdate mu{};
    gsr::PoseUpdate pu{};
    if (a->fused_map_update) {
        if (!a->scales || a->shs || a->fused_map_update->n != (size_t)a->P) return GSR_EINVAL;
        const int rc = make_map_update(a->fused_map_update, false, &mu);
        if (rc != GSR_OK) return rc;
    }
    if (a->fused_pose_step) {
        if (!a->fused_pose_step->update || (a->P > 0 && !a->fused_pose_step->means_world)) return GSR_EINVAL;
        const int rc = make_pose_update(a->fused_pose_step->update, &pu);
        if (rc != GSR_OK) return rc;
    }
    if (a->P == 0) { // src/Rasterizer.cu:263. An empty map still owes the tracking loop its bookkeeping: loss history, best pose, the next Tcw
        if (a->fused_pose_step && !a->fused_pose_step->sums_only && ((a->stages ? a->stages : GSR_STAGE_SPLAT) & GSR_STAGE_SPLAT)) {
            GSR_LAUNCH(gsr::K_pose_finish, dim3(1), dim3(64), 0, st, pu, const_cast<float*>(a->fused_pose_step->update->partial), (float*)nullptr);
            GSR_LAUNCHED();
        }
        return GSR_OK;
    }
    if (!a->geom_buffer || !a->binning_buffer || !a->image_buffer || !a->dL_dpix || !a->means3D ||
        !a->viewmatrix || !a->projmatrix || !a->background)
        return GSR_EINVAL;
    if ((a->scales != nullptr && a->rotations != nullptr) == (a->cov3D_precomp != nullptr)) return GSR_EINVAL;
    // K_splat_bwd writes dL_dsh[0 .. (D+1)^2) per splat and zero-fills up to M: an inconsistent D / M would write out of bounds
    if (a->shs && (!a->dL_dsh || a->M <= 0 || !a->cam_pos || a->D < 0 || a->D > 3 || (a->D + 1) * (a->D + 1) > a->M)) return GSR_EINVAL;
    const int P = a->P, W = a->width, H = a->height;
    FrameParams f = frame_params(P, a->D, a->M, W, H, a->tan_fovx, a->tan_fovy, a->scale_modifier, a->band_y0, a->band_y1);
    f.fold_depth_color = a->ds_detach_depth ? 0 : 1;
    const int T = f.grid_x * f.grid_y;
    GeomView gv; ImageView iv;
    geom_layout(a->geom_buffer, P, &gv);
    image_layout(a->image_buffer, W, H, &iv);
    // (the binning blob's layout follows the capacity the forward ran with; the kernel reads it from the header)
    const StageTimer tm{a->profile_events, st};
    const int stages = a->stages ? a->stages : (GSR_STAGE_BLEND | GSR_STAGE_SPLAT | GSR_STAGE_REZERO); // accumulators are clean by invariant
    const int Tb = (f.band_y1 - f.band_y0) * f.grid_x;
    if (stages & GSR_STAGE_CLEAR) {
        tm.begin(GSR_BWD_CLEAR);
        GSR_HIP(hipMemsetAsync(gv.acc, 0, (size_t)P * GSR_ACC_STRIDE * sizeof(float), st));
        tm.end(GSR_BWD_CLEAR);
    }
    if ((stages & GSR_STAGE_BLEND) && Tb > 0) {
        tm.begin(GSR_BWD_BLEND);
        // nobody consumes the colour sums (this call also runs the per-splat stage, which is handed no colour / SH output, and the fused depth
        // channel's colour is a constant): the fused pair's kernel without them (a tracking iteration)
        // (the fused map update steps the colours from those sums although it is handed no dL_dcolor buffer: ADVICE r4)
        const bool colour_unused = (stages & GSR_STAGE_SPLAT) && !a->dL_dcolor && !a->dL_dsh && !a->fused_map_update;
        const bool sil_only = a->dL_dds && a->dds_depth_only == 2; // dL_dds = ONE plane, the silhouette's upstream gradient (the depth channel's is zero)
        const bool no_colour = colour_unused && a->dL_dds && a->ds_detach_depth && !sil_only;
        // (round 6) the plain render without the colour sums: a tracking iteration whose depth term is the surface (median) depth passes no gradient through the
        // fused channels at all (dL_dds = NULL) — the lean body without DUAL's depth recursion: four waves per SIMD
        const bool no_colour_plain = colour_unused && !a->dL_dds;
#define GSR_BWD_DUAL(COL, SIL) GSR_LAUNCH((gsr::K_blend_bwd<GSR_ROWQ, true, COL, SIL>), dim3(4 * Tb), dim3(64), 0, st, iv, a->binning_buffer, gv, a->background, \
                                                  W, H, f.grid_x, Tb, f.band_y0 * f.grid_x, a->dL_dpix, a->dL_dds)
        if (sil_only)
            GSR_LAUNCH((gsr::K_blend_bwd<GSR_ROWQ, false, false, true>), dim3(4 * Tb), dim3(64), 0, st, iv, a->binning_buffer, gv, a->background, W, H, f.grid_x, Tb,
                       f.band_y0 * f.grid_x, a->dL_dpix, a->dL_dds);
        else if (no_colour && a->dds_depth_only) GSR_BWD_DUAL(false, false);
        else if (no_colour) GSR_BWD_DUAL(false, true);
        else if (a->dL_dds && a->dds_depth_only) GSR_BWD_DUAL(true, false);
        else if (a->dL_dds) GSR_BWD_DUAL(true, true);
#undef GSR_BWD_DUAL
        else if (no_colour_plain)
            GSR_LAUNCH((gsr::K_blend_bwd<GSR_ROWQ, false, false, false>), dim3(4 * Tb), dim3(64), 0, st, iv, a->binning_buffer, gv, a->background, W, H, f.grid_x, Tb,
                       f.band_y0 * f.grid_x, a->dL_dpix, (const float*)nullptr);
        else
            GSR_LAUNCH((gsr::K_blend_bwd<GSR_ROWQ, false>), dim3(4 * Tb), dim3(64), 0, st, iv, a->binning_buffer, gv, a->background, W, H, f.grid_x, Tb,
                               f.band_y0 * f.grid_x, a->dL_dpix, (const float*)nullptr);
        GSR_LAUNCHED();
        tm.end(GSR_BWD_BLEND);
    }
    if (stages & GSR_STAGE_SPLAT) {
        const gsr::SplatInputs in = splat_inputs(a->means3D, a->scales, a->rotations, nullptr, a->shs, a->cov3D_precomp,
                                                 a->colors_precomp, a->viewmatrix, a->projmatrix, a->cam_pos);
        gsr::SplatGrads o;
        o.dL_dmean2D = a->dL_dmean2D; o.dL_dconic = a->dL_dconic; o.dL_dopacity = a->dL_dopacity;
        o.dL_dcolor = a->dL_dcolor; o.dL_dmean3D = a->dL_dmean3D; o.dL_dcov3D = a->dL_dcov3D;
        o.dL_dsh = a->dL_dsh; o.dL_dscale = a->dL_dscale; o.dL_drot = a->dL_drot;
        tm.begin(GSR_BWD_SPLAT);
        if (a->fused_map_update) { // the per-splat stage takes the Adam step itself (include/gsr.h; arguments checked above, before the first launch)
            if (stages & GSR_STAGE_REZERO) GSR_LAUNCH((gsr::K_splat_bwd<true, true>), dim3(blocks256(P)), dim3(256), 0, st, f, in, gv, o, mu);
            else GSR_LAUNCH((gsr::K_splat_bwd<false, true>), dim3(blocks256(P)), dim3(256), 0, st, f, in, gv, o, mu);
        } else if (a->fused_pose_step) { // the per-splat stage forms the pose sums and its last workgroup takes the pose step (include/gsr.h)
            const gsr_pose_step_args* ps = a->fused_pose_step;
            static_assert(GSR_POSE_ACC_ROWS * 12 <= GSR_POSE_PARTIALS * 12, "gsr_pose_grad's scratch holds the accumulator rows");
            const gsr::PoseUpdate& u = pu;
            gsr::PoseStep k;
            k.X = ps->means_world; k.acc = const_cast<float*>(ps->update->partial); k.overflow_out = ps->sums_only ? ps->overflow_out : nullptr;
            if (stages & GSR_STAGE_REZERO) GSR_LAUNCH((gsr::K_splat_bwd_pose<true>), dim3(blocks256(P)), dim3(256), 0, st, f, in, gv, o, k);
            else GSR_LAUNCH((gsr::K_splat_bwd_pose<false>), dim3(blocks256(P)), dim3(256), 0, st, f, in, gv, o, k);
            if (!ps->sums_only) GSR_LAUNCH(gsr::K_pose_finish, dim3(1), dim3(64), 0, st, u, k.acc, (float*)nullptr);
        } else if (stages & GSR_STAGE_REZERO) GSR_LAUNCH((gsr::K_splat_bwd<true, false>), dim3(blocks256(P)), dim3(256), 0, st, f, in, gv, o, mu);
        else GSR_LAUNCH((gsr::K_splat_bwd<false, false>), dim3(blocks256(P)), dim3(256), 0, st, f, in, gv, o, mu);
        GSR_LAUNCHED();
        tm.end(GSR_BWD_SPLAT);
    }
    return GSR_OK;
}

int gsr_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                     uint8_t* present, void* stream)
{
    (void)projmatrix; // the reference's test only uses the view-space depth (auxiliary.h:154)
    if (P < 0 || (P > 0 && (!means3D || !viewmatrix || !present))) return GSR_EINVAL;
    if (P == 0) return GSR_OK;
    GSR_LAUNCH(gsr::K_mark_visible, dim3(blocks256(P)), dim3(256), 0, (hipStream_t)stream, P, means3D, viewmatrix, present);
    GSR_LAUNCHED();
    return GSR_OK;
}

int gsr_visible_filter(int P, int width, int height, const float* means3D, const float* scales,
                       float scale_modifier, const float* rotations, const float* viewmatrix,
                       const float* projmatrix, float tan_fovx, float tan_fovy, int prefiltered,
                       int* radii, void* stream)
{
    (void)prefiltered;
    if (P < 0 || width <= 0 || height <= 0) return GSR_EINVAL;
    if (P == 0) return GSR_OK;
    if (!means3D || !scales || !rotations || !viewmatrix || !projmatrix || !radii) return GSR_EINVAL;
    const FrameParams f = frame_params(P, 0, 0, width, height, tan_fovx, tan_fovy, scale_modifier);
    const gsr::SplatInputs in = splat_inputs(means3D, scales, rotations, nullptr, nullptr, nullptr, nullptr,
                                             viewmatrix, projmatrix, nullptr);
    GSR_LAUNCH(gsr::K_filter_radii, dim3(blocks256(P)), dim3(256), 0, (hipStream_t)stream, f, in, radii);
    GSR_LAUNCHED();
    return GSR_OK;
}

int gsr_acc_view(char* geom, int P, float** acc, size_t* count)
{
    if (!geom || P < 0 || !acc || !count) return GSR_EINVAL;
    GeomView gv;
    geom_layout(geom, P, &gv);
    *acc = gv.acc;
    *count = (size_t)P * GSR_ACC_STRIDE;
    return GSR_OK;
}

int gsr_transmittance_view(char* image, int width, int height, float** final_T)
{
    if (!image || width <= 0 || height <= 0 || !final_T) return GSR_EINVAL;
    ImageView iv;
    image_layout(image, width, height, &iv);
    *final_T = iv.final_T;
    return GSR_OK;
}

size_t gsr_knn_bytes(int P) { return gsr::knn_layout(nullptr, P, nullptr); }

int gsr_dist2(int P, const float* points, float* mean_dists, char* workspace, size_t workspace_bytes, void* stream)
{
    if (P < 0) return GSR_EINVAL;
    if (P == 0) return GSR_OK;
    if (!points || !mean_dists || !workspace || workspace_bytes < gsr_knn_bytes(P)) return GSR_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    gsr::KnnView k;
    gsr::knn_layout(workspace, P, &k);
    const int bits = gsr::knn_bucket_bits(P), nb = 1 << bits, shift = 30 - bits;
    const int nbox = (P + GSR_KNN_BOX - 1) / GSR_KNN_BOX;
    GSR_HIP(hipMemsetAsync(k.buckets, 0, (size_t)nb * sizeof(gsr::BucketRec), st));
    GSR_LAUNCH(gsr::K_knn_init, dim3(1), dim3(64), 0, st, k.bbox);
    GSR_LAUNCHED();
    GSR_LAUNCH(gsr::K_knn_bbox, dim3(std::min(blocks256(P), 1024)), dim3(256), 0, st, P, points, k.bbox);
    GSR_LAUNCHED();
    GSR_LAUNCH(gsr::K_knn_code, dim3(blocks256(P)), dim3(256), 0, st, P, shift, points, k.bbox, k.buckets, k.code, k.slot);
    GSR_LAUNCHED();
    GSR_LAUNCH(gsr::K_scan_tiles, dim3(1), dim3(1024), 0, st, nb, &k.buckets->cnt, 16, &k.buckets->start, 16, k.ranges, k.hdr, 0xFFFFFFFFu);
    GSR_LAUNCHED();
    GSR_LAUNCH(gsr::K_knn_fill, dim3(blocks256(P)), dim3(256), 0, st, P, shift, k.code, k.slot, k.buckets, k.pairs);
    GSR_LAUNCHED();
    GSR_LAUNCH(gsr::K_tile_sort<GSR_SORT_WAVE>, dim3(nb), dim3(GSR_SORT_SMALL_THREADS), 0, st, nb, k.ranges, k.hdr, k.pairs, k.order);
    GSR_LAUNCHED();
    GSR_LAUNCH(gsr::K_tile_sort<GSR_SORT_BLOCK>, dim3(nb), dim3(256), 0, st, nb, k.ranges, k.hdr, k.pairs, k.order);
    GSR_LAUNCHED();
    GSR_LAUNCH(gsr::K_knn_boxes, dim3(nbox), dim3(256), 0, st, P, points, k.order, k.spts, k.boxes);
    GSR_LAUNCHED();
    GSR_LAUNCH(gsr::K_knn_search, dim3(blocks256(P)), dim3(256), 0, st, P, nbox, k.spts, k.boxes, mean_dists);
    GSR_LAUNCHED();
    return GSR_OK;
}

// (the streaming SSIM kernels address a plane with 32-bit byte offsets)
static bool ssim_plane_ok(int H, int W) { return H > 0 && W > 0 && (unsigned long long)H * (unsigned long long)W < (1ull << 30); }

size_t gsr_ssim_partials(int C, int H, int W)
{
    if (C <= 0 || H <= 0 || W <= 0) return 0;
    const gsr::SsimGrid g = gsr::ssim_grid(C, H, W);
    return (size_t)C * g.nsx * g.nsy; // one row of sums per wave of the launch
}

int gsr_ssim_forward(const float* img1, const float* img2, int C, int H, int W, const float* taps11, float* partial,
                     float* dmaps, void* stream)
{
    if (!img1 || !img2 || !taps11 || !partial || C <= 0 || C > 65535 || !ssim_plane_ok(H, W)) return GSR_EINVAL;
    gsr::SsimTaps t;
    for (int k = 0; k < 11; k++) t.g[k] = taps11[k];
    const gsr::SsimGrid sg = gsr::ssim_grid(C, H, W);
    GSR_LAUNCH(gsr::K_ssim_fwd<false>, dim3((unsigned)gsr_ssim_partials(C, H, W)), dim3(64), 0, (hipStream_t)stream, img1, img2, C, H, W, t, sg, partial, dmaps,
                       gsr::MapLossPlanes{}, gsr::SsimRows{0, H, 0, H});
    GSR_LAUNCHED();
    return GSR_OK;
}

int gsr_ssim_backward(const float* img1, const float* img2, const float* dmaps, int C, int H, int W, const float* taps11,
                      const float* dL_dmean, float* dL_dimg1, void* stream)
{
    if (!img1 || !img2 || !dmaps || !taps11 || !dL_dmean || !dL_dimg1 || C <= 0 || C > 65535 || !ssim_plane_ok(H, W)) return GSR_EINVAL;
    gsr::SsimTaps t;
    for (int k = 0; k < 11; k++) t.g[k] = taps11[k];
    const gsr::SsimGrid sg = gsr::ssim_grid(C, H, W);
    GSR_LAUNCH(gsr::K_ssim_bwd<false>, dim3((unsigned)gsr_ssim_partials(C, H, W)), dim3(64), 0, (hipStream_t)stream, img1, img2, dmaps, C, H, W, t, sg, dL_dmean,
                       dL_dimg1, gsr::MapLossGrad{}, gsr::SsimRows{0, H, 0, H});
    GSR_LAUNCHED();
    return GSR_OK;
}

int gsr_to_camera(const float* means3D, size_t n, const float* Tcw, float* means_cam, void* stream)
{
    if (n == 0) return GSR_OK;
    if (!means3D || !Tcw || !means_cam || (n + 255) / 256 > 0x7FFFFFFFu) return GSR_EINVAL;
    GSR_LAUNCH(gsr::K_to_camera, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, means3D, n, Tcw, means_cam);
    GSR_LAUNCHED();
    return GSR_OK;
}

int gsr_pose_grad(const float* means3D, const float* dL_dmeans_cam, size_t n, const float* Tcw, float* partial,
                  float* dL_dmeans3D, void* stream)
{
    static_assert(GSR_POSE_PARTIALS == GSR_POSE_BLOCKS, "header and kernel agree on the number of partial rows");
    if ((!partial && !dL_dmeans3D) || !Tcw || (n > 0 && (!dL_dmeans_cam || (partial && !means3D)))) return GSR_EINVAL;
    GSR_LAUNCH(gsr::K_pose_grad, dim3(GSR_POSE_BLOCKS), dim3(256), 0, (hipStream_t)stream, means3D, dL_dmeans_cam, n, Tcw, partial,
                       dL_dmeans3D);
    GSR_LAUNCHED();
    return GSR_OK;
}

int gsr_pose_from_quat(const float* quat, const float* trans, float* Tcw, void* stream)
{
    if (!quat || !trans || !Tcw) return GSR_EINVAL;
    GSR_LAUNCH(gsr::K_rt2T, dim3(1), dim3(64), 0, (hipStream_t)stream, quat, trans, Tcw);
    GSR_LAUNCHED();
    return GSR_OK;
}

int gsr_pose_from_quat_backward(const float* quat, const float* dL_dTcw, float* dL_dquat, float* dL_dtrans, void* stream)
{
    if (!quat || !dL_dTcw || !dL_dquat || !dL_dtrans) return GSR_EINVAL;
    GSR_LAUNCH(gsr::K_rt2T_bwd, dim3(1), dim3(64), 0, (hipStream_t)stream, quat, dL_dTcw, dL_dquat, dL_dtrans);
    GSR_LAUNCHED();
    return GSR_OK;
}

int gsr_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, double lr, double beta1,
                  double beta2, double eps, int step, void* stream)
{
    if (n == 0) return GSR_OK;
    if (!param || !grad || !exp_avg || !exp_avg_sq || step < 1) return GSR_EINVAL;
    // the bias corrections in double, like the Python scalars of torch.optim.Adam
    // (and 1 - beta as well: 1.f - 0.999f is 1.3e-5 off the float nearest to 0.001)
    const double bc1 = 1.0 - std::pow(beta1, (double)step), bc2 = 1.0 - std::pow(beta2, (double)step);
    const float step_size = (float)(lr / bc1), sqrt_bc2 = (float)std::sqrt(bc2);
    const size_t blocks = (n + 1023) / 1024;
    if (blocks > 0x7FFFFFFFu) return GSR_EINVAL;
    GSR_LAUNCH(gsr::K_adam, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n, (float)(1.0 - beta1),
                       (float)beta2, (float)(1.0 - beta2), (float)eps, step_size, sqrt_bc2);
    GSR_LAUNCHED();
    return GSR_OK;
}

int gsr_pixel_loss(const float* image, const float* depth, const float* sur, const float* sil, const float* frame_rgb,
                   const float* frame_depth, int H, int W, int mode, float sil_thr, const float* w3, float* partial, float* sums, void* stream)
{
    static_assert(GSR_LOSS_PARTIALS == GSR_LOSS_BLOCKS, "header and kernels agree on the number of partial rows");
    if (!image || !frame_rgb || !frame_depth || !w3 || !partial || !sums || H <= 0 || W <= 0 || (mode != 0 && mode != 1)) return GSR_EINVAL;
    if (mode == 0 && !depth && !sur) return GSR_EINVAL;
    const size_t N = (size_t)H * W;
    const gsr::LossPlanes p{image, depth, sur, sil, frame_rgb, frame_depth};
    gsr::LossWeights w{{w3[0], w3[1], w3[2]}};
    const int nb = (int)std::min<size_t>(GSR_LOSS_BLOCKS, (N + 255) / 256);
    GSR_LAUNCH(gsr::K_loss_sums, dim3(nb), dim3(256), 0, (hipStream_t)stream, p, N, mode, sil_thr, partial);
    GSR_LAUNCHED();
    GSR_LAUNCH(gsr::K_loss_finish, dim3(1), dim3(GSR_FINISH_THREADS), 0, (hipStream_t)stream, partial, nb, mode, N, w, depth ? 0 : 1, sums);
    GSR_LAUNCHED();
    return GSR_OK;
}

int gsr_track_loss(const float* image, const float* depth, const float* sur, const float* sil, const float* frame_rgb, const float* frame_depth,
                   int H, int W, float sil_thr, const float* w3, float* partial, float* sums, float* dL_dimage, float* dL_ddepth, uint32_t* ticket,
                   void* stream)
{
    return gsr_track_loss_rows(image, depth, sur, sil, frame_rgb, frame_depth, H, W, sil_thr, w3, partial, sums, dL_dimage, dL_ddepth, ticket, 0, H, 0, stream);
}

int gsr_track_loss_rows(const float* image, const float* depth, const float* sur, const float* sil, const float* frame_rgb, const float* frame_depth,
                        int H, int W, float sil_thr, const float* w3, float* partial, float* sums, float* dL_dimage, float* dL_ddepth, uint32_t* ticket,
                        int row_begin, int row_end, int sil_is_transmittance, void* stream)
{
    if (!image || !frame_rgb || !frame_depth || !w3 || !partial || !sums || !dL_dimage || H <= 0 || W <= 0 || (!depth && !sur)) return GSR_EINVAL;
    if (row_begin < 0 || row_end > H || row_begin >= row_end) return GSR_EINVAL;
    const size_t N = (size_t)H * W, i0 = (size_t)row_begin * W, i1 = (size_t)row_end * W;
    const gsr::LossPlanes p{image, depth, sur, sil, frame_rgb, frame_depth};
    gsr::LossWeights w{{w3[0], w3[1], w3[2]}};
    const int nb = (int)std::min<size_t>(GSR_LOSS_BLOCKS, (i1 - i0 + 255) / 256);
    static_assert(GSR_FINISH_THREADS == 256, "the last workgroup of K_track_loss runs the finish");
    static_assert(GSR_TICKET_WORDS == GSR_TICKET_WORDS_DEV, "header and kernels agree on the arrival counters");
    GSR_LAUNCH(gsr::K_track_loss, dim3(nb), dim3(256), 0, (hipStream_t)stream, p, N, sil_thr, w, partial, dL_dimage, dL_ddepth, ticket, depth ? 0 : 1, sums, i0, i1, sil_is_transmittance ? 1 : 0);
    GSR_LAUNCHED();
    if (!ticket) {
        GSR_LAUNCH(gsr::K_loss_finish, dim3(1), dim3(GSR_FINISH_THREADS), 0, (hipStream_t)stream, partial, nb, 0, N, w, depth ? 0 : 1, sums);
        GSR_LAUNCHED();
    }
    return GSR_OK;
}

int gsr_pixel_loss_backward(const float* image, const float* depth, const float* sil, const float* frame_rgb, const float* frame_depth,
                            int H, int W, int mode, float sil_thr, const float* w3, const float* sums, const float* dL_dloss,
                            float* dL_dimage, float* dL_ddepth, void* stream)
{
    if (!image || !frame_rgb || !frame_depth || !w3 || !sums || !dL_dloss || !dL_dimage || H <= 0 || W <= 0 || (mode != 0 && mode != 1)) return GSR_EINVAL;
    const size_t N = (size_t)H * W;
    if ((N + 255) / 256 > 0x7FFFFFFFu) return GSR_EINVAL;
    const gsr::LossPlanes p{image, depth, nullptr, sil, frame_rgb, frame_depth};
    gsr::LossWeights w{{w3[0], w3[1], w3[2]}};
    GSR_LAUNCH(gsr::K_loss_grad, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, N, mode, sil_thr, w, sums, dL_dloss,
                       dL_dimage, dL_ddepth, (const float*)nullptr);
    GSR_LAUNCHED();
    return GSR_OK;
}

int gsr_pixel_loss_backward_add(const float* image, const float* depth, const float* sil, const float* frame_rgb, const float* frame_depth,
                                int H, int W, int mode, float sil_thr, const float* w3, const float* sums, const float* dL_dloss,
                                const float* add_image, float* dL_dimage, float* dL_ddepth, void* stream)
{
    if (!image || !frame_rgb || !frame_depth || !w3 || !sums || !dL_dimage || H <= 0 || W <= 0 || (mode != 0 && mode != 1)) return GSR_EINVAL;
    const size_t N = (size_t)H * W;
    if ((N + 255) / 256 > 0x7FFFFFFFu) return GSR_EINVAL;
    const gsr::LossPlanes p{image, depth, nullptr, sil, frame_rgb, frame_depth};
    gsr::LossWeights w{{w3[0], w3[1], w3[2]}};
    GSR_LAUNCH(gsr::K_loss_grad, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, N, mode, sil_thr, w, sums, dL_dloss,
                       dL_dimage, dL_ddepth, add_image);
    GSR_LAUNCHED();
    return GSR_OK;
}

int gsr_map_prepare(size_t n, const float* xyz, const float* logit, const float* log_scales, const float* unnorm_quat, const float* Tcw,
                    float* means_cam, float* opacities, float* scales, float* rotations, float reg_limit, float w_long, float w_scalar,
                    float* reg_partial, float* reg_out, void* stream)
{
    // (log_scales == NULL with reg_partial and reg_out: the partial sums are already there — the projection kernel wrote them, gsr_forward_args.raw — and only
    // their finish is wanted)
    const bool finish_only = !log_scales && reg_partial && reg_out && !means_cam && !opacities && !scales && !rotations;
    if ((means_cam && (!xyz || !Tcw)) || (opacities && !logit) || ((scales || reg_partial) && !log_scales && !finish_only) || (rotations && !unnorm_quat) ||
        (reg_out && !reg_partial) || (n + 255) / 256 > 0x7FFFFFFFu)
        return GSR_EINVAL;
    const unsigned rows = (unsigned)((n + 255) / 256);
    if (n > 0 && !finish_only)
        GSR_LAUNCH(gsr::K_map_prepare, dim3(rows), dim3(256), 0, (hipStream_t)stream, n, xyz, logit, log_scales, unnorm_quat, Tcw, means_cam,
                           opacities, scales, rotations, reg_limit, reg_partial);
    GSR_LAUNCHED();
    if (reg_partial && reg_out) {
        GSR_LAUNCH(gsr::K_scale_reg_finish, dim3(1), dim3(GSR_FINISH_THREADS), 0, (hipStream_t)stream, reg_partial, (int)rows, w_long, w_scalar, reg_out);
        GSR_LAUNCHED();
    }
    return GSR_OK;
}

int gsr_map_update(const gsr_map_update_args* a, void* stream)
{
    if (!a) return GSR_EINVAL;
    if (a->n == 0) return GSR_OK;
    gsr::MapUpdate u;
    const int rc = make_map_update(a, true, &u);
    if (rc != GSR_OK) return rc;
    if (a->n < (size_t)1 << 18) GSR_LAUNCH(gsr::K_map_update_small, dim3((unsigned)((a->n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a->n, u);
    else GSR_LAUNCH(gsr::K_map_update, dim3((unsigned)((a->n + 1023) / 1024)), dim3(256), 0, (hipStream_t)stream, a->n, u); // four splats per thread
    GSR_LAUNCHED();
    return GSR_OK;
}

int gsr_map_loss_total(const float* sums, const float* ssim_partial, int n_partial, size_t count, float c_ssim, const float* reg_out,
                       const char* geom, float* loss, void* stream)
{
    if (!sums || !loss || n_partial < 0 || (n_partial > 0 && (!ssim_partial || count == 0))) return GSR_EINVAL;
    GSR_LAUNCH(gsr::K_map_loss_total, dim3(1), dim3(GSR_FINISH_THREADS), 0, (hipStream_t)stream, sums, ssim_partial, n_partial,
                       count ? 1.f / (float)count : 0.f, n_partial ? c_ssim : 0.f, reg_out, overflow_flag(geom), loss);
    GSR_LAUNCHED();
    return GSR_OK;
}

int gsr_composite_forward(int world, int rank, const long long* order, const float* gathered, int gathered_planes, const float* layer4, int H, int W, int has_sur,
                          float* contrib, float* sil_total, float* surf, void* stream)
{
    if (world < 1 || rank < 0 || rank >= world || !order || !gathered || !layer4 || !contrib || !sil_total || H <= 0 || W <= 0) return GSR_EINVAL;
    if (gathered_planes < (has_sur ? 2 : 1)) return GSR_EINVAL;
    const size_t N = (size_t)H * W;
    GSR_LAUNCH(gsr::K_composite_fwd, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, world, rank, order, gathered, gathered_planes,
                       layer4, N, has_sur, contrib, sil_total, surf);
    GSR_LAUNCHED();
    return GSR_OK;
}

int gsr_composite_backward_local(int world, int rank, const long long* order, const float* gathered, int gathered_planes, const float* layer4, const float* g4, int H,
                                 int W, float* d_layer4, float* c_own, void* stream)
{
    if (world < 1 || rank < 0 || rank >= world || !order || !gathered || !layer4 || !d_layer4 || !c_own || H <= 0 || W <= 0 || gathered_planes < 1) return GSR_EINVAL;
    const size_t N = (size_t)H * W;
    GSR_LAUNCH(gsr::K_composite_bwd_local, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, world, rank, order, gathered,
                       gathered_planes, layer4, g4, N, d_layer4, c_own);
    GSR_LAUNCHED();
    return GSR_OK;
}

int gsr_composite_backward_occlusion(int world, int rank, const long long* order, const float* gathered, int gathered_planes, const float* c_all, const float* g_sil,
                                     int H, int W, float* dS, void* stream)
{
    if (world < 1 || rank < 0 || rank >= world || !order || !gathered || !c_all || !dS || H <= 0 || W <= 0 || gathered_planes < 1) return GSR_EINVAL;
    const size_t N = (size_t)H * W;
    GSR_LAUNCH(gsr::K_composite_bwd_occlusion, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, world, rank, order, gathered,
                       gathered_planes, c_all, g_sil, N, dS);
    GSR_LAUNCHED();
    return GSR_OK;
}

int gsr_band_composite_forward(int world, int rank, const long long* order, const float* layers_all, const float* own_layer, int H, int W, int row_begin, int row_end,
                               int halo, float* out_rgbd, float* out_sil, float* out_sur, void* stream)
{
    if (world < 1 || world > 32 || rank < 0 || rank >= world || !order || !own_layer || (world > 1 && !layers_all) || !out_rgbd || !out_sil || !out_sur || H <= 0 || W <= 0) return GSR_EINVAL;
    if (row_begin < 0 || row_end > H || row_begin >= row_end || halo < 0) return GSR_EINVAL;
    const int e0 = std::max(0, row_begin - halo), e1 = std::min(H, row_end + halo);
    const size_t n = (size_t)(e1 - e0) * W;
    GSR_LAUNCH(gsr::K_band_composite_fwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, world, rank, order, layers_all, own_layer, (size_t)H * W, W, e0, e1,
               row_begin, row_end, out_rgbd, out_sil, out_sur);
    GSR_LAUNCHED();
    return GSR_OK;
}

int gsr_band_composite_backward(int world, int rank, const long long* order, const float* layers_all, const float* own_layer, const float* g4, const float* g_sil, int H, int W,
                                int row_begin, int row_end, float* d_all, float* d_own, void* stream)
{
    if (world < 1 || world > 32 || rank < 0 || rank >= world || !order || !own_layer || (world > 1 && (!layers_all || !d_all)) || !g4 || !d_own || H <= 0 || W <= 0) return GSR_EINVAL;
    if (row_begin < 0 || row_end > H || row_begin >= row_end) return GSR_EINVAL;
    const size_t n = (size_t)(row_end - row_begin) * W;
    if (world <= 8)
        GSR_LAUNCH(gsr::K_band_composite_bwd<8>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, world, rank, order, layers_all, own_layer, g4, g_sil, (size_t)H * W, W,
                   row_begin, row_end, d_all, d_own);
    else
        GSR_LAUNCH(gsr::K_band_composite_bwd<32>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, world, rank, order, layers_all, own_layer, g4, g_sil, (size_t)H * W, W,
                   row_begin, row_end, d_all, d_own);
    GSR_LAUNCHED();
    return GSR_OK;
}

int gsr_shard_map_totals(int world, const float* rows, int H, int W, const float* w3, float c_ssim, float w_long, float w_scalar, float* sums, float* reg_out, float* loss,
                         void* stream)
{
    if (world < 1 || !rows || !w3 || !sums || !loss || H <= 0 || W <= 0) return GSR_EINVAL;
    gsr::ShardTotals t;
    t.w[0] = w3[0]; t.w[1] = w3[1]; t.w[2] = w3[2]; t.c_ssim = c_ssim; t.w_long = w_long; t.w_scalar = w_scalar;
    t.inv_pixels3 = 1.f / (3.f * (float)((size_t)H * W)); t.inv_count_ssim = 1.f / (float)((size_t)3 * H * W);
    GSR_LAUNCH(gsr::K_shard_map_totals, dim3(1), dim3(64), 0, (hipStream_t)stream, world, rows, t, sums, reg_out, loss);
    GSR_LAUNCHED();
    return GSR_OK;
}

int gsr_shard_order(int world, const float* kd_nodes, const float* Tcw, long long* order, void* stream)
{
    if (world < 1 || world > 32 || !order || (world > 1 && (!kd_nodes || !Tcw))) return GSR_EINVAL;
    GSR_LAUNCH(gsr::K_shard_order, dim3(1), dim3(1), 0, (hipStream_t)stream, world, kd_nodes, Tcw, order);
    GSR_LAUNCHED();
    return GSR_OK;
}

int gsr_map_loss_forward(const float* image, const float* depth, const float* sur, const float* sil, const float* frame_rgb, const float* frame_depth,
                         int H, int W, const float* taps11, float sil_thr, float* partial6, float* dmaps, void* stream)
{
    return gsr_map_loss_forward_rows(image, depth, sur, sil, frame_rgb, frame_depth, H, W, taps11, sil_thr, partial6, dmaps, 0, H, stream);
}

namespace {
// the rows whose derivative maps a band's gradient needs: the band and the window's radius either side, inside the image
inline void band_ext(int H, int row_begin, int row_end, int* e0, int* e1)
{
    *e0 = std::max(0, row_begin - GSR_SSIM_R);
    *e1 = std::min(H, row_end + GSR_SSIM_R);
}
} // namespace

size_t gsr_map_loss_partials_rows(int H, int W, int row_begin, int row_end)
{
    if (H <= 0 || W <= 0 || row_begin < 0 || row_end > H || row_begin >= row_end) return 0;
    int e0, e1;
    band_ext(H, row_begin, row_end, &e0, &e1);
    return gsr_ssim_partials(3, e1 - e0, W);
}

int gsr_map_loss_forward_rows(const float* image, const float* depth, const float* sur, const float* sil, const float* frame_rgb, const float* frame_depth,
                              int H, int W, const float* taps11, float sil_thr, float* partial6, float* dmaps, int row_begin, int row_end, void* stream)
{
    if (!image || !frame_rgb || !frame_depth || !taps11 || !partial6 || !dmaps || !ssim_plane_ok(H, W)) return GSR_EINVAL;
    if (row_begin < 0 || row_end > H || row_begin >= row_end) return GSR_EINVAL;
    gsr::SsimTaps t;
    for (int k = 0; k < 11; k++) t.g[k] = taps11[k];
    int e0, e1;
    band_ext(H, row_begin, row_end, &e0, &e1);
    const gsr::SsimGrid sg = gsr::ssim_grid(3, e1 - e0, W);
    const gsr::MapLossPlanes ml{depth, sur, sil, frame_depth, sil_thr, partial6};
    GSR_LAUNCH(gsr::K_ssim_fwd<true>, dim3(2u * (unsigned)gsr_ssim_partials(3, e1 - e0, W)), dim3(64), 0, (hipStream_t)stream, image, frame_rgb, 3, H, W, t, sg, (float*)nullptr,
                       dmaps, ml, gsr::SsimRows{e0, e1, row_begin, row_end});
    GSR_LAUNCHED();
    return GSR_OK;
}

int gsr_map_loss_finish(const float* partial6, const float* reg_partial, size_t n_gaussians, int H, int W, const float* w3, float c_ssim, float w_long,
                        float w_scalar, const char* geom, float* sums, float* reg_out, float* loss, void* stream)
{
    if (!partial6 || !w3 || !sums || !loss || H <= 0 || W <= 0 || (reg_partial && !reg_out)) return GSR_EINVAL;
    return gsr_map_loss_finish_rows(partial6, reg_partial, n_gaussians, H, W, w3, c_ssim, w_long, w_scalar, geom, sums, reg_out, loss, 0, H, stream);
}

int gsr_map_loss_finish_rows(const float* partial6, const float* reg_partial, size_t n_gaussians, int H, int W, const float* w3, float c_ssim, float w_long,
                             float w_scalar, const char* geom, float* sums, float* reg_out, float* loss, int row_begin, int row_end, void* stream)
{
    if (!partial6 || !w3 || !sums || !loss || H <= 0 || W <= 0 || (reg_partial && !reg_out)) return GSR_EINVAL;
    if (row_begin < 0 || row_end > H || row_begin >= row_end) return GSR_EINVAL;
    gsr::MapFinish m;
    m.partial6 = partial6; m.n6 = (int)gsr_map_loss_partials_rows(H, W, row_begin, row_end);
    m.reg_partial = reg_partial; m.n_reg = reg_partial ? (int)((n_gaussians + 255) / 256) : 0;
    m.inv_pixels3 = 1.f / (3.f * (float)((size_t)H * W)); m.inv_count_ssim = 1.f / (float)((size_t)3 * H * W);
    m.w[0] = w3[0]; m.w[1] = w3[1]; m.w[2] = w3[2]; m.c_ssim = c_ssim; m.w_long = w_long; m.w_scalar = w_scalar;
    m.overflow = overflow_flag(geom); m.sums = sums; m.reg_out = reg_out; m.loss = loss;
    GSR_LAUNCH(gsr::K_map_finish, dim3(1), dim3(GSR_MAP_FINISH_THREADS), 0, (hipStream_t)stream, m);
    GSR_LAUNCHED();
    return GSR_OK;
}

int gsr_map_loss_backward(const float* image, const float* depth, const float* frame_rgb, const float* frame_depth, const float* dmaps, int H, int W,
                          const float* taps11, const float* w3, const float* neg_c_ssim, const float* sums, float* dL_dimage, float* dL_ddepth, void* stream)
{
    return gsr_map_loss_backward_rows(image, depth, frame_rgb, frame_depth, dmaps, H, W, taps11, w3, neg_c_ssim, sums, dL_dimage, dL_ddepth, 0, H, stream);
}

int gsr_map_loss_backward_rows(const float* image, const float* depth, const float* frame_rgb, const float* frame_depth, const float* dmaps, int H, int W,
                               const float* taps11, const float* w3, const float* neg_c_ssim, const float* sums, float* dL_dimage, float* dL_ddepth,
                               int row_begin, int row_end, void* stream)
{
    if (!image || !frame_rgb || !frame_depth || !dmaps || !taps11 || !w3 || !neg_c_ssim || !sums || !dL_dimage || !ssim_plane_ok(H, W)) return GSR_EINVAL;
    if (row_begin < 0 || row_end > H || row_begin >= row_end) return GSR_EINVAL;
    gsr::SsimTaps t;
    for (int k = 0; k < 11; k++) t.g[k] = taps11[k];
    const gsr::SsimGrid sg = gsr::ssim_grid(3, row_end - row_begin, W);
    const gsr::MapLossGrad mg{depth, frame_depth, sums, w3[0] / (3.f * (float)((size_t)H * W)), w3[1], dL_ddepth};
    GSR_LAUNCH(gsr::K_ssim_bwd<true>, dim3((dL_ddepth ? 2u : 1u) * (unsigned)gsr_ssim_partials(3, row_end - row_begin, W)), dim3(64), 0, (hipStream_t)stream, image, frame_rgb, dmaps, 3, H, W, t,
                       sg, neg_c_ssim, dL_dimage, mg, gsr::SsimRows{row_begin, row_end, row_begin, row_end});
    GSR_LAUNCHED();
    return GSR_OK;
}

namespace {
int make_pose_update(const gsr_pose_update_args* a, gsr::PoseUpdate* out)
{
    if (!a || !a->quat_trans || !a->moments || !a->best || !a->history || !a->Tcw || !a->partial || !a->loss || a->step < 1) return GSR_EINVAL;
    gsr::PoseUpdate u;
    u.quat_trans = a->quat_trans; u.moments = a->moments; u.best = a->best; u.history = a->history; u.Tcw = a->Tcw;
    u.partial = a->partial; u.loss = a->loss; u.overflow = overflow_flag(a->geom); u.skip = a->skip;
    const double bc1 = 1.0 - std::pow(a->beta1, (double)a->step), bc2 = 1.0 - std::pow(a->beta2, (double)a->step);
    u.w1 = (float)(1.0 - a->beta1); u.b2 = (float)a->beta2; u.w2 = (float)(1.0 - a->beta2); u.eps = (float)a->eps;
    u.step_size = (float)(a->lr / bc1); u.sqrt_bias2 = (float)std::sqrt(bc2);
    *out = u;
    return GSR_OK;
}
} // namespace

int gsr_pose_update(const gsr_pose_update_args* a, void* stream)
{
    gsr::PoseUpdate u;
    const int rc = make_pose_update(a, &u);
    if (rc != GSR_OK) return rc;
    GSR_LAUNCH(gsr::K_pose_update, dim3(1), dim3(64), 0, (hipStream_t)stream, u);
    GSR_LAUNCHED();
    return GSR_OK;
}

int gsr_pose_finish(const gsr_pose_update_args* a, float* acc_rows, float* sums_out, void* stream)
{
    if (!acc_rows) return GSR_EINVAL;
    gsr::PoseUpdate u;
    const int rc = make_pose_update(a, &u);
    if (rc != GSR_OK) return rc;
    GSR_LAUNCH(gsr::K_pose_finish, dim3(1), dim3(64), 0, (hipStream_t)stream, u, acc_rows, sums_out);
    GSR_LAUNCHED();
    return GSR_OK;
}

int gsr_reproj_loss(const float* obs, const float* Xw, const float* inv_sigma2, size_t M, const float* Tcw, float fx, float fy, float cx, float cy,
                    float weight, float grad_scale, int refresh_inliers, uint8_t* inliers, float* pose_row, float* loss, void* stream)
{
    if (M > 0x7FFFFFFFu || !Tcw || !pose_row || !loss || refresh_inliers < 0 || refresh_inliers > 2) return GSR_EINVAL;
    if (M > 0 && (!obs || !Xw || !inv_sigma2 || (refresh_inliers != 2 && !inliers))) return GSR_EINVAL;
    if (M == 0) return GSR_OK;
    GSR_LAUNCH(gsr::K_reproj, dim3(1), dim3(256), 0, (hipStream_t)stream, obs, Xw, inv_sigma2, (int)M, Tcw, fx, fy, cx, cy, weight, weight * grad_scale,
                       refresh_inliers, inliers, pose_row, loss);
    GSR_LAUNCHED();
    return GSR_OK;
}

int gsr_pose_step(const float* means3D, const float* dL_dmeans_cam, size_t n, const gsr_pose_update_args* a, uint32_t* ticket, void* stream)
{
    gsr::PoseUpdate u;
    const int rc = make_pose_update(a, &u);
    if (rc != GSR_OK) return rc;
    if (!ticket || (n > 0 && (!means3D || !dL_dmeans_cam))) return GSR_EINVAL;
    GSR_LAUNCH(gsr::K_pose_step, dim3(GSR_POSE_BLOCKS), dim3(256), 0, (hipStream_t)stream, means3D, dL_dmeans_cam, n, const_cast<float*>(a->partial), ticket, u);
    GSR_LAUNCHED();
    return GSR_OK;
}

int gsr_scale_reg(const float* log_scales, size_t n, float limit, float w_long, float w_scalar, float* partial, float* out, void* stream)
{
    if (!partial || !out || (n > 0 && !log_scales)) return GSR_EINVAL;
    const int nb = (int)std::max<size_t>(1, std::min<size_t>(GSR_LOSS_BLOCKS, (n + 255) / 256));
    GSR_LAUNCH(gsr::K_scale_reg, dim3(nb), dim3(256), 0, (hipStream_t)stream, log_scales, n, limit, partial);
    GSR_LAUNCHED();
    GSR_LAUNCH(gsr::K_scale_reg_finish, dim3(1), dim3(GSR_FINISH_THREADS), 0, (hipStream_t)stream, partial, nb, w_long, w_scalar, out);
    GSR_LAUNCHED();
    return GSR_OK;
}

int gsr_scale_reg_backward(const float* log_scales, size_t n, float limit, float w_long, float w_scalar, const float* out,
                           const float* dL_dvalue, float* dL_dlog_scales, void* stream)
{
    if (n == 0) return GSR_OK;
    if (!log_scales || !out || !dL_dvalue || !dL_dlog_scales || (n + 255) / 256 > 0x7FFFFFFFu) return GSR_EINVAL;
    GSR_LAUNCH(gsr::K_scale_reg_bwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, log_scales, n, limit, w_long, w_scalar,
                       out, dL_dvalue, dL_dlog_scales);
    GSR_LAUNCHED();
    return GSR_OK;
}

int gsr_debug_export(int P, int width, int height, int R, const char* geom, const char* binning,
                     const char* image, const gsr_debug_arrays* out, void* stream)
{
    if (!out || !geom || !image || P < 0 || width <= 0 || height <= 0) return GSR_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    GeomView gv; ImageView iv; BinView bv;
    geom_layout(const_cast<char*>(geom), P, &gv);
    image_layout(const_cast<char*>(image), width, height, &iv);
    const int gx = (width + GSR_TILE - 1) / GSR_TILE, gy = (height + GSR_TILE - 1) / GSR_TILE, T = gx * gy;
    const size_t N = (size_t)width * height;
    if (P > 0) {
        GSR_LAUNCH(gsr::K_export_splats, dim3(blocks256(P)), dim3(256), 0, st, P, gx, gy, gv, out->means2D,
                           out->depths, out->conic_opacity, out->rgb, out->tiles_touched);
        GSR_LAUNCHED();
    }
    if (out->ranges) GSR_HIP(hipMemcpyAsync(out->ranges, iv.ranges, (size_t)T * 8, hipMemcpyDeviceToDevice, st));
    if (out->final_T) GSR_HIP(hipMemcpyAsync(out->final_T, iv.final_T, N * 4, hipMemcpyDeviceToDevice, st));
    if (out->n_contrib) GSR_HIP(hipMemcpyAsync(out->n_contrib, iv.n_contrib, N * 4, hipMemcpyDeviceToDevice, st));
    if (binning && R > 0) {
        GeomHeader h;
        GSR_HIP(hipMemcpyAsync(&h, geom, sizeof(uint32_t) * 4, hipMemcpyDeviceToHost, st));
        GSR_HIP(hipStreamSynchronize(st));
        binning_layout(const_cast<char*>(binning), (size_t)h.capacity, &bv);
        if (out->point_list) GSR_HIP(hipMemcpyAsync(out->point_list, bv.point_list, (size_t)R * 4, hipMemcpyDeviceToDevice, st));
        if (out->point_list_keys) {
            GSR_LAUNCH(gsr::K_export_keys, dim3(T), dim3(256), 0, st, T, iv.ranges, bv.point_list, gv, out->point_list_keys);
            GSR_LAUNCHED();
        }
    }
    GSR_HIP(hipStreamSynchronize(st));
    return GSR_OK;
}

} // extern "C"

#ifdef GSR_EXP_SORT_PHASES
extern "C" int gsr_debug_sort_phases(unsigned long long* dst, int n_words)
{
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(gsr::g_sort_phases), (size_t)n_words * 8);
}
#endif
#ifdef GSR_EXP_TIMELINE
extern "C" int gsr_debug_timeline(unsigned long long* dst, int n_words)
{
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(gsr::g_timeline), (size_t)n_words * 8);
}
#endif
