// FusedOps.cpp — see FusedOps.h.
#include "FusedOps.h"

#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>

#include <stdexcept>
#include <string>

#include "../../include/gsr.h"

namespace ORB_SLAM2 {
namespace fused {

namespace {

void* stream_of(const torch::Tensor& t) { return (void*)c10::hip::getCurrentHIPStream(t.device().index()).stream(); }
void check(int rc, const char* what)
{
    if (rc < 0) throw std::runtime_error(std::string(what) + ": " + gsr_error_string(rc));
}
const float* fp(const torch::Tensor& t) { return t.data_ptr<float>(); }

struct ToCameraFn : public torch::autograd::Function<ToCameraFn> {
    static torch::Tensor forward(torch::autograd::AutogradContext* ctx, torch::Tensor Tcw, torch::Tensor X)
    {
        const auto T = Tcw.contiguous(), Xc = X.contiguous();
        c10::DeviceGuard guard(Xc.device());
        auto out = torch::empty_like(Xc);
        check(gsr_to_camera(fp(Xc), (size_t)Xc.size(0), fp(T), out.data_ptr<float>(), stream_of(Xc)), "gsr_to_camera");
        ctx->save_for_backward({T, Xc});
        return out;
    }
    static torch::autograd::tensor_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::tensor_list g)
    {
        const auto saved = ctx->get_saved_variables();
        const auto &T = saved[0], &X = saved[1];
        const bool want_T = ctx->needs_input_grad(0), want_X = ctx->needs_input_grad(1);
        if (!want_T && !want_X) return {torch::Tensor(), torch::Tensor()};
        c10::DeviceGuard guard(X.device());
        const auto dmc = g[0].contiguous();
        torch::Tensor part, dX, dT;
        if (want_T) part = torch::empty({GSR_POSE_PARTIALS, 12}, X.options());
        if (want_X) dX = torch::empty_like(X);
        check(gsr_pose_grad(fp(X), fp(dmc), (size_t)X.size(0), fp(T), want_T ? part.data_ptr<float>() : nullptr,
                            want_X ? dX.data_ptr<float>() : nullptr, stream_of(X)), "gsr_pose_grad");
        if (want_T) {
            const auto s = part.sum(0);
            dT = torch::zeros({4, 4}, X.options());
            dT.slice(0, 0, 3).slice(1, 0, 3).copy_(s.slice(0, 0, 9).reshape({3, 3}));
            dT.slice(0, 0, 3).slice(1, 3, 4).copy_(s.slice(0, 9, 12).reshape({3, 1}));
        }
        return {dT, dX};
    }
};

struct Rt2TFn : public torch::autograd::Function<Rt2TFn> {
    static torch::Tensor forward(torch::autograd::AutogradContext* ctx, torch::Tensor quat, torch::Tensor trans)
    {
        const auto q = quat.contiguous(), t = trans.contiguous();
        c10::DeviceGuard guard(q.device());
        auto T = torch::empty({4, 4}, q.options());
        check(gsr_pose_from_quat(fp(q), fp(t), T.data_ptr<float>(), stream_of(q)), "gsr_pose_from_quat");
        ctx->save_for_backward({q});
        ctx->saved_data["qs"] = quat.sizes().vec();
        ctx->saved_data["ts"] = trans.sizes().vec();
        return T;
    }
    static torch::autograd::tensor_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::tensor_list g)
    {
        const auto q = ctx->get_saved_variables()[0];
        c10::DeviceGuard guard(q.device());
        auto dq = torch::empty({4}, q.options()), dt = torch::empty({3}, q.options());
        const auto dT = g[0].contiguous();
        check(gsr_pose_from_quat_backward(fp(q), fp(dT), dq.data_ptr<float>(), dt.data_ptr<float>(), stream_of(q)), "gsr_pose_from_quat_backward");
        return {dq.reshape(ctx->saved_data["qs"].toIntVector()), dt.reshape(ctx->saved_data["ts"].toIntVector())};
    }
};

struct SsimFn : public torch::autograd::Function<SsimFn> {
    // `need`: img1 wants a gradient — decided by the caller BEFORE apply(): inside forward() ctx->needs_input_grad() throws
    // ("Index out of range") when nothing requires grad or grad mode is off (libtorch leaves next_edges empty then), i.e.
    // exactly when SSIM is evaluated as a metric
    static torch::Tensor forward(torch::autograd::AutogradContext* ctx, torch::Tensor img1, torch::Tensor img2, std::vector<double> taps, bool need)
    {
        const auto a = img1.contiguous(), b = img2.contiguous();
        c10::DeviceGuard guard(a.device());
        const int C = (int)a.size(0), H = (int)a.size(1), W = (int)a.size(2);
        float tp[11];
        for (int i = 0; i < 11; i++) tp[i] = (float)taps[i];
        auto partial = torch::empty({(int64_t)gsr_ssim_partials(C, H, W)}, a.options());
        torch::Tensor dmaps = need ? torch::empty({3, C, H, W}, a.options()) : torch::Tensor();
        check(gsr_ssim_forward(fp(a), fp(b), C, H, W, tp, partial.data_ptr<float>(), need ? dmaps.data_ptr<float>() : nullptr, stream_of(a)),
              "gsr_ssim_forward");
        ctx->save_for_backward({a, b, need ? dmaps : a});
        ctx->saved_data["taps"] = taps;
        ctx->saved_data["need"] = need;
        return partial.sum() / (double)((int64_t)C * H * W);
    }
    static torch::autograd::tensor_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::tensor_list g)
    {
        if (!ctx->saved_data["need"].toBool()) return {torch::Tensor(), torch::Tensor(), torch::Tensor(), torch::Tensor()};
        const auto saved = ctx->get_saved_variables();
        const auto &a = saved[0], &b = saved[1], &dmaps = saved[2];
        c10::DeviceGuard guard(a.device());
        const auto taps = ctx->saved_data["taps"].toDoubleVector();
        float tp[11];
        for (int i = 0; i < 11; i++) tp[i] = (float)taps[i];
        const auto gm = g[0].to(torch::kFloat32).contiguous();
        auto out = torch::empty_like(a);
        check(gsr_ssim_backward(fp(a), fp(b), fp(dmaps), (int)a.size(0), (int)a.size(1), (int)a.size(2), tp, fp(gm), out.data_ptr<float>(), stream_of(a)),
              "gsr_ssim_backward");
        return {out, torch::Tensor(), torch::Tensor(), torch::Tensor()};
    }
};

// the pixel terms of a loop's loss: one reduction pass + one gradient pass (include/gsr.h: gsr_pixel_loss)
struct PixelLossFn : public torch::autograd::Function<PixelLossFn> {
    static torch::Tensor forward(torch::autograd::AutogradContext* ctx, torch::Tensor image, torch::Tensor depth, torch::Tensor sur, torch::Tensor sil,
                                 torch::Tensor frgb, torch::Tensor fdepth, int64_t mode, double thr, std::vector<double> w)
    {
        // (an absent plane travels as an empty tensor: autograd::Function::apply does not take undefined tensors)
        auto c = [](const torch::Tensor& t) { return t.numel() ? t.detach().contiguous() : torch::Tensor(); };
        const auto im = c(image), d = c(depth), su = c(sur), si = c(sil), fr = c(frgb), fd = c(fdepth);
        c10::DeviceGuard guard(im.device());
        const int H = (int)im.size(-2), W = (int)im.size(-1);
        // the kernels take raw pointers: every plane must live on the image's device and hold H*W (3*H*W) floats
        const int64_t N = (int64_t)H * W;
        auto plane_ok = [&](const torch::Tensor& t, int64_t n, const char* name) {
            TORCH_CHECK(!t.defined() || (t.device() == im.device() && t.scalar_type() == torch::kFloat32 && t.numel() == n),
                        "fused pixel loss: ", name, " must be a float32 tensor of ", n, " elements on ", im.device());
        };
        TORCH_CHECK(im.is_cuda() && im.scalar_type() == torch::kFloat32 && im.numel() == 3 * N, "fused pixel loss: image must be a float32 [3,H,W] tensor on the GPU");
        plane_ok(d, N, "depth"); plane_ok(su, N, "sur"); plane_ok(si, N, "sil"); plane_ok(fr, 3 * N, "frame_rgb"); plane_ok(fd, N, "frame_depth");
        const float w3[3] = {(float)w[0], (float)w[1], (float)w[2]};
        auto partial = torch::empty({GSR_LOSS_PARTIALS * 5}, im.options()), sums = torch::empty({8}, im.options());
        auto opt = [](const torch::Tensor& t) -> const float* { return t.defined() ? t.data_ptr<float>() : nullptr; };
        check(gsr_pixel_loss(fp(im), opt(d), opt(su), opt(si), fp(fr), fp(fd), H, W, (int)mode, (float)thr, w3, partial.data_ptr<float>(),
                             sums.data_ptr<float>(), stream_of(im)), "gsr_pixel_loss");
        ctx->save_for_backward({im, d.defined() ? d : im, si.defined() ? si : im, fr, fd, sums});
        ctx->saved_data["has_d"] = d.defined();
        ctx->saved_data["has_s"] = si.defined();
        ctx->saved_data["mode"] = mode;
        ctx->saved_data["thr"] = thr;
        ctx->saved_data["w"] = w;
        return sums[5].clone(); // (not a view of the buffer saved for backward: an in-place op on the loss would trip its version counter)
    }
    static torch::autograd::tensor_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::tensor_list g)
    {
        const auto saved = ctx->get_saved_variables();
        const auto &im = saved[0], &fr = saved[3], &fd = saved[4], &sums = saved[5];
        const bool has_d = ctx->saved_data["has_d"].toBool(), has_s = ctx->saved_data["has_s"].toBool();
        c10::DeviceGuard guard(im.device());
        const auto w = ctx->saved_data["w"].toDoubleVector();
        const float w3[3] = {(float)w[0], (float)w[1], (float)w[2]};
        const auto go = g[0].to(torch::kFloat32).contiguous();
        auto dimage = torch::empty_like(im);
        torch::Tensor ddepth = has_d && ctx->needs_input_grad(1) ? torch::empty_like(saved[1]) : torch::Tensor();
        check(gsr_pixel_loss_backward(fp(im), has_d ? fp(saved[1]) : nullptr, has_s ? fp(saved[2]) : nullptr, fp(fr), fp(fd), (int)im.size(-2), (int)im.size(-1),
                                      (int)ctx->saved_data["mode"].toInt(), (float)ctx->saved_data["thr"].toDouble(), w3, fp(sums), fp(go),
                                      dimage.data_ptr<float>(), ddepth.defined() ? ddepth.data_ptr<float>() : nullptr, stream_of(im)),
              "gsr_pixel_loss_backward");
        return {dimage, ddepth, torch::Tensor(), torch::Tensor(), torch::Tensor(), torch::Tensor(), torch::Tensor(), torch::Tensor(), torch::Tensor()};
    }
};

struct ScaleRegFn : public torch::autograd::Function<ScaleRegFn> {
    static torch::Tensor forward(torch::autograd::AutogradContext* ctx, torch::Tensor log_scales, double limit, double w_long, double w_scalar)
    {
        const auto ls = log_scales.detach().contiguous();
        TORCH_CHECK(ls.is_cuda() && ls.scalar_type() == torch::kFloat32 && ls.dim() == 2 && ls.size(1) == 3, "fused scale regularisers: log_scales must be a float32 [n,3] tensor on the GPU");
        c10::DeviceGuard guard(ls.device());
        auto partial = torch::empty({GSR_LOSS_PARTIALS * 3}, ls.options()), out = torch::empty({4}, ls.options());
        check(gsr_scale_reg(fp(ls), (size_t)ls.size(0), (float)limit, (float)w_long, (float)w_scalar, partial.data_ptr<float>(), out.data_ptr<float>(),
                            stream_of(ls)), "gsr_scale_reg");
        ctx->save_for_backward({ls, out});
        ctx->saved_data["limit"] = limit;
        ctx->saved_data["wl"] = w_long;
        ctx->saved_data["ws"] = w_scalar;
        return out[3].clone();
    }
    static torch::autograd::tensor_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::tensor_list g)
    {
        const auto saved = ctx->get_saved_variables();
        const auto &ls = saved[0], &out = saved[1];
        c10::DeviceGuard guard(ls.device());
        const auto go = g[0].to(torch::kFloat32).contiguous();
        auto d = torch::empty_like(ls);
        check(gsr_scale_reg_backward(fp(ls), (size_t)ls.size(0), (float)ctx->saved_data["limit"].toDouble(), (float)ctx->saved_data["wl"].toDouble(),
                                     (float)ctx->saved_data["ws"].toDouble(), fp(out), fp(go), d.data_ptr<float>(), stream_of(ls)), "gsr_scale_reg_backward");
        return {d, torch::Tensor(), torch::Tensor(), torch::Tensor()};
    }
};

} // namespace

torch::Tensor tracking_pixel_loss(const torch::Tensor& image, const torch::Tensor& depth, const torch::Tensor& sil, const torch::Tensor& frame_rgb,
                                  const torch::Tensor& frame_depth, double w_image, double w_depth, bool depth_is_surface)
{
    const auto none = torch::empty({0}, image.options());
    return PixelLossFn::apply(image, depth_is_surface ? none : depth, depth_is_surface ? depth : none, sil, frame_rgb, frame_depth, 0, 0.99,
                              std::vector<double>{w_image, w_depth, 0.0});
}
torch::Tensor mapping_pixel_loss(const torch::Tensor& image, const torch::Tensor& depth, const torch::Tensor& sur, const torch::Tensor& sil,
                                 const torch::Tensor& frame_rgb, const torch::Tensor& frame_depth, double w_l1, double w_depth, double w_sur)
{
    return PixelLossFn::apply(image, depth, sur, sil, frame_rgb, frame_depth, 1, 0.99, std::vector<double>{w_l1, w_depth, w_sur});
}
torch::Tensor scale_regularisers(const torch::Tensor& log_scales, double limit, double w_long, double w_scalar)
{
    return ScaleRegFn::apply(log_scales, limit, w_long, w_scalar);
}

torch::Tensor to_camera(const torch::Tensor& Tcw, const torch::Tensor& X) { return ToCameraFn::apply(Tcw, X); }
torch::Tensor rt2T(const torch::Tensor& quat, const torch::Tensor& trans) { return Rt2TFn::apply(quat, trans); }
torch::Tensor ssim_mean(const torch::Tensor& img1, const torch::Tensor& img2, const std::vector<float>& taps11)
{
    const bool need = img1.requires_grad() && torch::GradMode::is_enabled();
    return SsimFn::apply(img1, img2.detach(), std::vector<double>(taps11.begin(), taps11.end()), need);
}

void Adam::step()
{
    torch::NoGradGuard ng;
    if (state_.size() != groups_.size()) state_.resize(groups_.size());
    for (size_t i = 0; i < groups_.size(); i++) {
        auto& p = groups_[i].param;
        if (!p.grad().defined()) continue;
        auto& st = state_[i];
        if (!st.exp_avg.defined()) { st.exp_avg = torch::zeros_like(p); st.exp_avg_sq = torch::zeros_like(p); }
        st.step += 1;
        const auto g = p.grad().contiguous();
        c10::DeviceGuard guard(p.device());
        check(gsr_adam_step(p.data_ptr<float>(), fp(g), st.exp_avg.data_ptr<float>(), st.exp_avg_sq.data_ptr<float>(), (size_t)p.numel(), groups_[i].lr,
                            0.9, 0.999, eps_, st.step, stream_of(p)), "gsr_adam_step");
    }
}

void Adam::ensure_state_(size_t i)
{
    if (state_.size() != groups_.size()) state_.resize(groups_.size());
    auto& st = state_.at(i);
    if (!st.exp_avg.defined()) {
        torch::NoGradGuard ng;
        st.exp_avg = torch::zeros_like(groups_[i].param);
        st.exp_avg_sq = torch::zeros_like(groups_[i].param);
    }
}

void Adam::replace_extended(size_t i, const torch::Tensor& param, int64_t added)
{
    if (state_.size() != groups_.size()) state_.resize(groups_.size());
    auto& st = state_.at(i);
    if (st.exp_avg.defined()) { // (a group that has not stepped yet has no moments: they start as zeros of the new size)
        auto shape = st.exp_avg.sizes().vec();
        shape[0] = added;
        const auto z = torch::zeros(shape, st.exp_avg.options());
        st.exp_avg = torch::cat({st.exp_avg, z}, 0);
        st.exp_avg_sq = torch::cat({st.exp_avg_sq, z}, 0);
    }
    groups_.at(i).param = param;
}

void Adam::replace_selected(size_t i, const torch::Tensor& param, const torch::Tensor& keep)
{
    if (state_.size() != groups_.size()) state_.resize(groups_.size());
    auto& st = state_.at(i);
    if (st.exp_avg.defined()) {
        st.exp_avg = st.exp_avg.index_select(0, keep);
        st.exp_avg_sq = st.exp_avg_sq.index_select(0, keep);
    }
    groups_.at(i).param = param;
}

void Adam::zero_grad()
{
    for (auto& g : groups_)
        if (g.param.grad().defined()) g.param.mutable_grad() = torch::Tensor();
}

} // namespace fused
} // namespace ORB_SLAM2
