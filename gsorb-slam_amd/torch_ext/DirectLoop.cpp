// DirectLoop.cpp — SlamLoop's iterations as fixed sequences of C-ABI launches on a persistent workspace (LoopConfig::direct).
//
// Through libtorch autograd (SlamLoop.cpp) a mapping iteration at 1 M Gaussians, 1200x680 is ~60 launches and as many host-side
// operator dispatches around the rasterizer pair: activations and their backward nodes, gradient accumulation, zero fills, five
// Adam launches, allocations of every intermediate — 1.08-1.30 ms per iteration of which the rasterizer pair is 0.55
// (profiles/r04_loop.md). Here nothing inside an iteration goes through a tensor library: libtorch owns the memory (the
// parameters, the Adam moments, one workspace that lives as long as the map's size), and an iteration is
//   mapping : gsr_map_prepare -> gsr_forward_ws (fused pair) -> gsr_map_loss_forward -> gsr_map_loss_finish -> gsr_map_loss_backward ->
//             gsr_backward [with gsr_map_update fused into its per-splat stage]                                           (13 launches)
//   tracking: gsr_to_camera -> gsr_forward_ws -> gsr_pixel_loss -> gsr_pixel_loss_backward_add -> gsr_backward ->
//             gsr_pose_grad -> gsr_pose_update                                                                          (13 launches)
// with no allocation, no gradient tensor of a raw parameter, and no host synchronisation except where the reference has one
// (the tracking loop reads its loss every iteration, Render.cc:1107; the mapping loop of RenderForFrame never does: MapFrame
// reads all losses back once, at the end). Reference: src/Render.cc:420-483, :1054-1126; src/Gaussian.cc:97-175.
#include "SlamLoop.h"

#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>

#include <cmath>
#include <limits>
#include <stdexcept>
#include <string>

#include "../../include/gsr.h"

namespace ORB_SLAM2 {

namespace {
void chk(int rc, const char* what)
{
    if (rc < 0) throw std::runtime_error(std::string(what) + ": " + gsr_error_string(rc));
}
float* f(const torch::Tensor& t) { return t.data_ptr<float>(); }
char* b(const torch::Tensor& t) { return reinterpret_cast<char*>(t.data_ptr()); }
} // namespace

// Everything an iteration touches besides the parameters and their moments. Sized by the map (n) and the image; rebuilt when
// the map grows or shrinks (AddGaussians / PruneLowOpacity replace the parameter tensors anyway).
struct SlamLoop::Direct {
    int64_t n = -1;
    size_t binning_bytes = 0;
    torch::Tensor geom, image, binning;                                  // the rasterizer's three workspaces
    torch::Tensor mc, opac, scales, rots, radii;                         // what the rasterizer takes
    torch::Tensor out_color, out_sur, out_ds;                            // what it renders
    torch::Tensor g_image, g_ds, g_ssim, dmaps, ssim_partial, partial6;  // upstream gradients of the renders, SSIM / loss scratch
    torch::Tensor d_mc, d_m2d, d_col, d_opac, d_scale, d_rot;            // the rasterizer's gradients
    torch::Tensor loss_partial, sums, reg_partial, reg_out, neg_c, Tcw, bg, view, proj, campos, history;
    torch::Tensor pose, pose_moments, best, pose_partial;                // tracking: [7], [14], [8], [GSR_POSE_PARTIALS, 12]
    torch::Tensor pose_acc;                                              // [64, 12] the fused pose step's accumulator rows (zero between launches)
    torch::Tensor tickets;                                               // [2 * GSR_TICKET_WORDS] arrival counters of the kernels that finish their own sums (zero between launches)
    int64_t history_len = 0;
};

SlamLoop::~SlamLoop() = default;

void* SlamLoop::stream_() const { return (void*)c10::hip::getCurrentHIPStream(dev_.index()).stream(); }

void SlamLoop::ensure_direct_(int64_t history_len)
{
    if (!cfg_.fused_pair || !cfg_.fused_ops) throw std::runtime_error("LoopConfig::direct needs fused_pair and fused_ops");
    if (!d_) d_ = std::make_shared<Direct>();
    Direct& d = *d_;
    const auto fo = torch::TensorOptions().device(dev_).dtype(torch::kFloat32);
    const auto bo = torch::TensorOptions().device(dev_).dtype(torch::kByte);
    const int64_t n = size();
    if (!d.image.defined()) { // per image size: once
        const int64_t np = (int64_t)gsr_ssim_partials(3, H_, W_);
        d.image = torch::empty({(int64_t)gsr_image_bytes(W_, H_)}, bo);
        d.out_color = torch::empty({3, H_, W_}, fo); d.out_sur = torch::empty({1, H_, W_}, fo); d.out_ds = torch::empty({2, H_, W_}, fo);
        d.g_image = torch::empty({3, H_, W_}, fo); d.g_ds = torch::zeros({2, H_, W_}, fo); // (the silhouette plane's gradient stays zero: its mask is detached)
        d.g_ssim = torch::empty({3, H_, W_}, fo); d.dmaps = torch::empty({3, 3, H_, W_}, fo); d.ssim_partial = torch::empty({np}, fo); d.partial6 = torch::empty({np * 6}, fo);
        d.loss_partial = torch::empty({GSR_LOSS_PARTIALS * 5}, fo); d.sums = torch::empty({8}, fo); d.reg_out = torch::empty({4}, fo);
        d.neg_c = torch::full({1}, -(cfg_.im_weight_mapping * (1 - cfg_.lam)), fo);
        d.Tcw = torch::eye(4, fo); d.bg = torch::zeros({3}, fo); d.view = torch::eye(4, fo); d.campos = torch::zeros({3}, fo);
        d.proj = rasterizer_.raster_settings_.projmatrix.to(dev_, torch::kFloat32).contiguous();
        d.pose = torch::zeros({7}, fo); d.pose_moments = torch::zeros({14}, fo); d.best = torch::zeros({8}, fo);
        d.tickets = torch::zeros({2 * GSR_TICKET_WORDS}, fo.dtype(torch::kInt32));
        d.pose_acc = torch::zeros({64, 12}, fo);
    }
    if (d.n != n) { // per map size
        d.n = n;
        d.geom = torch::empty({(int64_t)gsr_geom_bytes((int)n)}, bo);
        d.mc = torch::empty({n, 3}, fo); d.opac = torch::empty({n}, fo); d.scales = torch::empty({n, 3}, fo); d.rots = torch::empty({n, 4}, fo);
        d.radii = torch::empty({n}, fo.dtype(torch::kInt32));
        d.d_mc = torch::empty({n, 3}, fo); d.d_m2d = torch::empty({n, 3}, fo); d.d_col = torch::empty({n, 3}, fo); d.d_opac = torch::empty({n}, fo);
        d.d_scale = torch::empty({n, 3}, fo); d.d_rot = torch::empty({n, 4}, fo);
        d.reg_partial = torch::empty({3 * ((n + 255) / 256) + 3}, fo);
        d.pose_partial = torch::empty({GSR_POSE_PARTIALS, 12}, fo);
        d.binning = torch::Tensor(); d.binning_bytes = 0;
        grow_binning_(cfg_.binning_capacity > 0 ? (size_t)cfg_.binning_capacity : 4 * (size_t)n + 65536); // grows at the first synchronised look at an overflow
    }
    if (d.history_len < history_len) { d.history = torch::empty({history_len}, fo); d.history_len = history_len; }
}

void SlamLoop::grow_binning_(size_t capacity)
{
    Direct& d = *d_;
    const size_t bytes = gsr_binning_bytes(capacity);
    if (bytes <= d.binning_bytes) return;
    d.binning = torch::empty({(int64_t)bytes}, torch::TensorOptions().device(dev_).dtype(torch::kByte));
    d.binning_bytes = bytes;
}

// the fused colour + depth / silhouette pass on the workspace (sync-free), and its backward
void SlamLoop::direct_forward_()
{
    Direct& d = *d_;
    const auto& s = rasterizer_.raster_settings_;
    gsr_forward_args a{};
    a.P = (int)d.n; a.D = 0; a.M = 0;
    a.background = f(d.bg); a.width = W_; a.height = H_;
    a.means3D = f(d.mc); a.colors_precomp = f(rgb); a.opacities = f(d.opac); a.scales = f(d.scales); a.scale_modifier = s.scale_modifier;
    a.rotations = f(d.rots); a.viewmatrix = f(d.view); a.projmatrix = f(d.proj); a.cam_pos = f(d.campos);
    a.tan_fovx = s.tanfovx; a.tan_fovy = s.tanfovy; a.prefiltered = 0;
    a.out_color = f(d.out_color); a.out_depth = f(d.out_sur); a.radii = d.radii.data_ptr<int>(); a.out_ds = f(d.out_ds);
    chk(gsr_forward_ws(&a, b(d.geom), b(d.binning), d.binning_bytes, b(d.image), stream_()), "gsr_forward_ws");
}

void SlamLoop::direct_backward_(bool detach_depth_colour, bool means_only, const ::gsr_map_update_args* fused, const ::gsr_pose_step_args* pose_step)
{
    Direct& d = *d_;
    const auto& s = rasterizer_.raster_settings_;
    gsr_backward_args a{};
    a.P = (int)d.n; a.D = 0; a.M = 0; a.R = -1;
    a.background = f(d.bg); a.width = W_; a.height = H_;
    a.means3D = f(d.mc); a.colors_precomp = f(rgb); a.scales = f(d.scales); a.scale_modifier = s.scale_modifier; a.rotations = f(d.rots);
    a.viewmatrix = f(d.view); a.projmatrix = f(d.proj); a.cam_pos = f(d.campos); a.tan_fovx = s.tanfovx; a.tan_fovy = s.tanfovy;
    a.radii = d.radii.data_ptr<int>();
    a.geom_buffer = b(d.geom); a.binning_buffer = b(d.binning); a.image_buffer = b(d.image);
    a.dL_dpix = f(d.g_image); a.dL_dds = f(d.g_ds); a.ds_detach_depth = detach_depth_colour ? 1 : 0;
    a.dds_depth_only = 1; // (the silhouette is a detached mask in both losses: plane 1 of g_ds would be zeros)
    a.fused_map_update = fused; // (the per-splat stage then takes the Adam step itself and writes no gradient)
    a.fused_pose_step = pose_step; // (tracking: the per-splat stage forms the pose sums and its last workgroup takes the pose step: no gradient tensor)
    if (!fused && !pose_step) a.dL_dmean3D = f(d.d_mc);
    if (!means_only && !fused) { // (tracking optimises the pose only: the per-splat stage then skips the covariance -> scale / rotation chain and 56 bytes of stores per Gaussian)
        a.dL_dmean2D = f(d.d_m2d); a.dL_dopacity = f(d.d_opac); a.dL_dcolor = f(d.d_col);
        a.dL_dscale = f(d.d_scale); a.dL_drot = f(d.d_rot);
    }
    a.stages = GSR_STAGE_BLEND | GSR_STAGE_SPLAT; // one backward per forward: the forward left the accumulators clear
    chk(gsr_backward(&a, stream_()), "gsr_backward");
}

// After a synchronised look at the workspace: did the last forward fit? If not, the workspace grows to 1.5 x what it needed.
bool SlamLoop::direct_overflowed_()
{
    int R = 0, ov = 0;
    chk(gsr_ws_status(b(d_->geom), stream_(), &R, &ov), "gsr_ws_status");
    if (ov) grow_binning_((size_t)R + (size_t)R / 2 + 65536);
    return ov != 0;
}

void SlamLoop::direct_map_iteration_(const LoopFrame& fr, float* loss_slot)
{
    Direct& d = *d_;
    void* const st = stream_();
    const size_t n = (size_t)d.n;
    const float limit = (float)(0.1 * cfg_.scene_radius), wl = (float)cfg_.reg_long_weight, wsc = (float)cfg_.reg_scalar_weight;
    chk(gsr_map_prepare(n, f(xyz), f(logit_opacities), f(log_scales), f(unnorm_quat), f(d.Tcw), f(d.mc), f(d.opac), f(d.scales), f(d.rots), limit, wl, wsc,
                        f(d.reg_partial), cfg_.fused_loss ? nullptr : f(d.reg_out), st), "gsr_map_prepare");
    direct_forward_();
    // Render.cc:436-471: lam * L1 + (1 - lam) * (1 - SSIM), masked depth L1, masked surface-depth L1 (no gradient), the regularisers
    const float w3[3] = {(float)(cfg_.im_weight_mapping * cfg_.lam), (float)cfg_.depth_weight_mapping, (float)cfg_.sur_depth_weight_mapping};
    const float c_ssim = (float)(cfg_.im_weight_mapping * (1 - cfg_.lam));
    const float *img = f(d.out_color), *dep = f(d.out_ds), *sil = f(d.out_ds) + (size_t)H_ * W_, *sur = f(d.out_sur);
    if (cfg_.fused_loss) { // two passes over the image and one single-workgroup kernel between them
        chk(gsr_map_loss_forward(img, dep, sur, sil, f(fr.rgb), f(fr.depth), H_, W_, taps_host_.data(), 0.99f, f(d.partial6), f(d.dmaps), st), "gsr_map_loss_forward");
        chk(gsr_map_loss_finish(f(d.partial6), f(d.reg_partial), n, H_, W_, w3, c_ssim, wl, wsc, b(d.geom), f(d.sums), f(d.reg_out), loss_slot, st), "gsr_map_loss_finish");
        chk(gsr_map_loss_backward(img, dep, f(fr.rgb), f(fr.depth), f(d.dmaps), H_, W_, taps_host_.data(), w3, f(d.neg_c), f(d.sums), f(d.g_image), f(d.g_ds), st),
            "gsr_map_loss_backward");
    } else {
        chk(gsr_pixel_loss(img, dep, sur, sil, f(fr.rgb), f(fr.depth), H_, W_, 1, 0.99f, w3, f(d.loss_partial), f(d.sums), st), "gsr_pixel_loss");
        chk(gsr_ssim_forward(img, f(fr.rgb), 3, H_, W_, taps_host_.data(), f(d.ssim_partial), f(d.dmaps), st), "gsr_ssim_forward");
        chk(gsr_ssim_backward(img, f(fr.rgb), f(d.dmaps), 3, H_, W_, taps_host_.data(), f(d.neg_c), f(d.g_ssim), st), "gsr_ssim_backward");
        chk(gsr_pixel_loss_backward_add(img, dep, sil, f(fr.rgb), f(fr.depth), H_, W_, 1, 0.99f, w3, f(d.sums), nullptr, f(d.g_ssim), f(d.g_image), f(d.g_ds), st),
            "gsr_pixel_loss_backward_add");
        chk(gsr_map_loss_total(f(d.sums), f(d.ssim_partial), (int)d.ssim_partial.numel(), (size_t)3 * H_ * W_, c_ssim, f(d.reg_out), b(d.geom), loss_slot, st),
            "gsr_map_loss_total");
    }
    ::gsr_map_update_args u{};
    u.n = n; u.xyz = f(xyz); u.rgb = f(rgb); u.unnorm_quat = f(unnorm_quat); u.logit = f(logit_opacities); u.log_scales = f(log_scales);
    for (int g = 0; g < 5; g++) {
        u.exp_avg[g] = f(fopt_->exp_avg(g)); u.exp_avg_sq[g] = f(fopt_->exp_avg_sq(g));
        u.lr[g] = fopt_->lr(g); u.step[g] = fopt_->advance_step(g);
    }
    u.dL_dmeans_cam = f(d.d_mc); u.dL_dcolors = f(d.d_col); u.dL_drotations = f(d.d_rot); u.dL_dopacities = f(d.d_opac); u.dL_dscales = f(d.d_scale);
    u.opacities = f(d.opac); u.scales = f(d.scales); u.Tcw = f(d.Tcw);
    u.reg_out = f(d.reg_out); u.reg_limit = limit; u.w_long = wl; u.w_scalar = wsc;
    u.geom = b(d.geom); u.beta1 = 0.9; u.beta2 = 0.999; u.eps = fopt_->eps();
    if (cfg_.fused_update) {
        direct_backward_(false, false, &u, nullptr); // backward and update in the same per-splat pass (gsr_backward_args.fused_map_update)
    } else {
        direct_backward_(false, false, nullptr, nullptr);
        chk(gsr_map_update(&u, st), "gsr_map_update");
    }
}

std::vector<double> SlamLoop::MapFrame(const LoopFrame& fr, int iters)
{
    std::vector<double> losses;
    if (iters <= 0 || size() == 0) return losses;
    if (!direct_()) {
        for (int i = 0; i < iters; i++) losses.push_back(MappingIteration(fr));
        return losses;
    }
    torch::NoGradGuard ng;
    c10::DeviceGuard guard(dev_);
    ensure_direct_(iters);
    Direct& d = *d_;
    const LoopFrame frame{fr.rgb.to(dev_, torch::kFloat32).contiguous(), fr.depth.to(dev_, torch::kFloat32).contiguous(), fr.Tcw};
    d.Tcw.copy_(fr.Tcw.to(torch::kFloat32).reshape({4, 4}));
    int done = 0;
    while (done < iters) {
        const int batch = iters - done;
        for (int i = 0; i < batch; i++) direct_map_iteration_(frame, f(d.history) + i);
        const auto h = d.history.slice(0, 0, batch).to(torch::kCPU); // the one read-back (and synchronisation) of the batch
        // an iteration whose forward ran out of workspace rendered nothing, wrote NaN and skipped its step: it is taken again
        // (its step count with it) once the workspace has grown. The first iteration on a new map size is where this can happen.
        const bool overflowed = direct_overflowed_();
        int kept = 0;
        for (int i = 0; i < batch; i++) {
            const double v = h[i].item<float>();
            if (overflowed && std::isnan(v)) continue;
            losses.push_back(v);
            kept++;
        }
        if (overflowed) for (int k = 0; k < batch - kept; k++) for (int g = 0; g < 5; g++) fopt_->retract_step(g);
        if (!overflowed) break;
        done += kept;
    }
    return losses;
}

std::vector<double> SlamLoop::direct_track_(const LoopFrame& fr, const torch::Tensor& Tcw_init, int iters, torch::Tensor* Tcw_best)
{
    torch::NoGradGuard ng;
    c10::DeviceGuard guard(dev_);
    ensure_direct_(iters);
    Direct& d = *d_;
    void* const st = stream_();
    const LoopFrame frame{fr.rgb.to(dev_, torch::kFloat32).contiguous(), fr.depth.to(dev_, torch::kFloat32).contiguous(), fr.Tcw};
    // Gaussian::InitCameraPose (Gaussian.cc:97-150): quaternion of the initial rotation, its translation, fresh moments
    const auto T0 = Tcw_init.to(torch::kCPU, torch::kFloat32);
    const auto q0 = rot_to_quat(T0.slice(0, 0, 3).slice(1, 0, 3)).reshape({4});
    d.pose.copy_(torch::cat({q0, T0.slice(0, 0, 3).slice(1, 3, 4).reshape({3})}));
    d.pose_moments.zero_();
    d.best.zero_(); d.best.slice(0, 0, 1).fill_(std::numeric_limits<float>::infinity());
    chk(gsr_pose_from_quat(f(d.pose), f(d.pose) + 4, f(d.Tcw), st), "gsr_pose_from_quat");
    // the map does not move while the pose is tracked: its activations are formed once per call
    chk(gsr_map_prepare((size_t)d.n, nullptr, f(logit_opacities), f(log_scales), f(unnorm_quat), nullptr, nullptr, f(d.opac), f(d.scales), f(d.rots), 0.f, 0.f, 0.f,
                        nullptr, nullptr, st), "gsr_map_prepare");
    const float w3[3] = {(float)cfg_.im_weight_tracking, (float)cfg_.depth_weight_tracking, 0.f};
    const float *img = f(d.out_color), *sil = f(d.out_ds) + (size_t)H_ * W_;
    const float* dep = cfg_.use_sur_depth ? nullptr : f(d.out_ds);
    const float* sur = cfg_.use_sur_depth ? f(d.out_sur) : nullptr;
    std::vector<double> history;
    double last_loss = 0.0;
    int step = 0;
    for (int it = 0; it < iters; it++) {
        chk(gsr_to_camera(f(xyz), (size_t)d.n, f(d.Tcw), f(d.mc), st), "gsr_to_camera");
        direct_forward_();
        // Render.cc:1088-1105: the masked L1 sums and their gradient planes, one pass over the render
        chk(gsr_track_loss(img, dep, sur, sil, f(frame.rgb), f(frame.depth), H_, W_, 0.99f, w3, f(d.loss_partial), f(d.sums), f(d.g_image), f(d.g_ds),
                           reinterpret_cast<uint32_t*>(d.tickets.data_ptr<int>()), st), "gsr_track_loss");
        gsr_pose_update_args u{};
        u.quat_trans = f(d.pose); u.moments = f(d.pose_moments); u.best = f(d.best); u.history = f(d.history) + it; u.Tcw = f(d.Tcw);
        u.partial = f(d.pose_partial); u.loss = f(d.sums) + 5; u.geom = b(d.geom);
        u.lr = cfg_.lr_cam_quat; u.beta1 = 0.9; u.beta2 = 0.999; u.eps = 1e-15; u.step = ++step;
        uint32_t* const tickets = reinterpret_cast<uint32_t*>(d.tickets.data_ptr<int>()) + GSR_TICKET_WORDS;
        if (cfg_.fused_update) { // the backward's per-splat stage forms the pose sums (into accumulator rows that are zero between launches), a one-wave kernel takes the step: no dL/dmeans tensor
            u.partial = f(d.pose_acc);
            const gsr_pose_step_args ps{f(xyz), &u};
            direct_backward_(true, true, nullptr, &ps); // the [z, 1, 0] colours are detached while tracking (Render.cc:949-981)
        } else {
            direct_backward_(true, true, nullptr, nullptr);
            chk(gsr_pose_step(f(xyz), f(d.d_mc), (size_t)d.n, &u, tickets, st), "gsr_pose_step"); // the pose sums and the step in one launch
        }
        const double lv = d.history.slice(0, it, it + 1).item<float>(); // Render.cc:1107: the loop looks at every loss
        if (std::isnan(lv) && direct_overflowed_()) { --step; --it; continue; } // the workspace has grown: take the iteration again
        history.push_back(lv);
        if (std::fabs(last_loss - lv) < 10e-4) break;                                                    // Render.cc:1113-1114
        last_loss = lv;
    }
    if (Tcw_best) {
        const auto bh = d.best.to(torch::kCPU);
        *Tcw_best = rt2T(bh.slice(0, 1, 5).reshape({4, 1}), bh.slice(0, 5, 8).reshape({3, 1})).to(dev_);
    }
    return history;
}

} // namespace ORB_SLAM2
