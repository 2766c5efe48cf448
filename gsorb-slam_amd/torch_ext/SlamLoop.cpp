// SlamLoop.cpp — see SlamLoop.h. Reference line numbers are those of GSORB-SLAM's src/Render.cc / src/Gaussian.cc / src/Utils.cc.
#include "SlamLoop.h"

#include <cmath>
#include <limits>
#include <stdexcept>

namespace ORB_SLAM2 {

namespace {

torch::Tensor l1_sum(const torch::Tensor& a, const torch::Tensor& b, const torch::Tensor& mask) // L1LossForTracking (Utils.cc:58-65)
{
    const auto d = torch::abs(a - b);
    return torch::where(mask, d, torch::zeros_like(d)).sum();
}

// Utils.cc:67-74: the reference's window, floor((x - 11) / 2) in the exponent (asymmetric)
torch::Tensor ssim_taps(torch::Device dev)
{
    std::vector<float> g(11);
    double s = 0;
    for (int x = 0; x < 11; x++) {
        const double e = std::floor((x - 11) / 2.0);
        g[x] = (float)std::exp(-(e * e) / (2.0 * 1.5 * 1.5));
        s += g[x];
    }
    auto t = torch::tensor(g, torch::kFloat32);
    return (t / t.sum()).to(dev);
}

// Utils.cc:77-100 with the 11x11 window applied as an 11x1 and a 1x11 pass (the window is an outer product)
torch::Tensor ssim(const torch::Tensor& img1, const torch::Tensor& img2, const torch::Tensor& taps)
{
    const double C1 = 0.01 * 0.01, C2 = 0.03 * 0.03;
    const int64_t ch = img1.size(0);
    const auto wv = taps.reshape({1, 1, 11, 1}).expand({ch, 1, 11, 1}).contiguous();
    const auto wh = taps.reshape({1, 1, 1, 11}).expand({ch, 1, 1, 11}).contiguous();
    const std::vector<int64_t> one{1, 1}, padv{5, 0}, padh{0, 5};
    auto conv = [&](const torch::Tensor& x) -> torch::Tensor {
        const torch::Tensor y = torch::conv2d(x.unsqueeze(0), wv, torch::Tensor(), one, padv, one, ch);
        return torch::conv2d(y, wh, torch::Tensor(), one, padh, one, ch).squeeze(0);
    };
    const torch::Tensor mu1 = conv(img1), mu2 = conv(img2);
    const torch::Tensor mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
    const torch::Tensor s1 = conv(img1 * img1) - mu1_sq, s2 = conv(img2 * img2) - mu2_sq, s12 = conv(img1 * img2) - mu12;
    return (((2.0 * mu12 + C1) * (2.0 * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s1 + s2 + C2))).mean();
}

GaussianRasterizationSettings make_settings(const LoopConfig& c, int W, int H, float fx, float fy, torch::Device dev)
{
    const float near = 0.01f, far = 100.0f; // include/Camera.h:15
    const float tanfovx = W / (2 * fx), tanfovy = H / (2 * fy);
    auto P = torch::zeros({4, 4}, torch::kFloat32);
    P[0][0] = 1 / tanfovx; P[1][1] = 1 / tanfovy; P[2][2] = far / (far - near); P[2][3] = -(far * near) / (far - near); P[3][2] = 1;
    GaussianRasterizationSettings s;
    s.image_height = H; s.image_width = W; s.tanfovx = tanfovx; s.tanfovy = tanfovy;
    s.bg = torch::zeros({3}, torch::TensorOptions().device(dev));
    s.scale_modifier = (float)c.scale_modifier;
    s.viewmatrix = torch::eye(4, torch::TensorOptions().device(dev)); // the means are moved to the camera frame (Render.cc:750-752)
    s.projmatrix = P.t().contiguous().to(dev);
    s.sh_degree = 1;
    s.camera_center = torch::zeros({3}, torch::TensorOptions().device(dev));
    s.prefiltered = false;
    return s;
}

std::unique_ptr<torch::optim::Adam> make_adam(const std::vector<std::pair<torch::Tensor, double>>& groups)
{
    std::vector<torch::optim::OptimizerParamGroup> pg;
    for (const auto& g : groups) {
        auto o = std::make_unique<torch::optim::AdamOptions>(g.second);
        o->eps(1e-15);
        pg.emplace_back(std::vector<torch::Tensor>{g.first}, std::move(o));
    }
    return std::make_unique<torch::optim::Adam>(pg, torch::optim::AdamOptions(0.0).eps(1e-15));
}

} // namespace

torch::Tensor rt2T(const torch::Tensor& quat, const torch::Tensor& trans)
{
    auto q = quat.reshape({4});
    q = q / torch::sqrt((q * q).sum());
    const auto r = q[0], x = q[1], y = q[2], z = q[3];
    const auto R = torch::stack({1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                                 2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                                 2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)}).reshape({3, 3});
    const auto top = torch::cat({R, trans.reshape({3, 1})}, 1);
    auto last = torch::zeros({1, 4}, top.options());
    last[0][3] = 1.0;
    return torch::cat({top, last}, 0);
}

torch::Tensor rot_to_quat(const torch::Tensor& Rin)
{
    const auto R = Rin.to(torch::kCPU, torch::kFloat64);
    auto at = [&](int i, int j) { return R[i][j].item<double>(); };
    double q[4];
    const double t = at(0, 0) + at(1, 1) + at(2, 2);
    if (t > 0) {
        const double s = std::sqrt(t + 1.0) * 2;
        q[0] = 0.25 * s; q[1] = (at(2, 1) - at(1, 2)) / s; q[2] = (at(0, 2) - at(2, 0)) / s; q[3] = (at(1, 0) - at(0, 1)) / s;
    } else {
        int i = 0;
        if (at(1, 1) > at(i, i)) i = 1;
        if (at(2, 2) > at(i, i)) i = 2;
        const int j = (i + 1) % 3, k = (i + 2) % 3;
        const double s = std::sqrt(1.0 + at(i, i) - at(j, j) - at(k, k)) * 2;
        q[0] = (at(k, j) - at(j, k)) / s; q[1 + i] = 0.25 * s; q[1 + j] = (at(j, i) + at(i, j)) / s; q[1 + k] = (at(k, i) + at(i, k)) / s;
    }
    return torch::tensor({(float)q[0], (float)q[1], (float)q[2], (float)q[3]}, torch::kFloat32);
}

SlamLoop::SlamLoop(const LoopConfig& cfg, int width, int height, float fx, float fy, torch::Device device)
    : cfg_(cfg), W_(width), H_(height), fx_(fx), fy_(fy), dev_(device),
      rasterizer_(make_settings(cfg, width, height, fx, fy, device)), taps_(ssim_taps(device))
{
    const auto t = taps_.to(torch::kCPU).contiguous();
    taps_host_.assign(t.data_ptr<float>(), t.data_ptr<float>() + 11);
}

void SlamLoop::SetMap(torch::Tensor xyz_, torch::Tensor rgb_, torch::Tensor quat_, torch::Tensor logit_, torch::Tensor logs_)
{
    auto leaf = [&](const torch::Tensor& t) { return t.to(dev_, torch::kFloat32).contiguous().detach().clone().requires_grad_(true); };
    xyz = leaf(xyz_); rgb = leaf(rgb_); unnorm_quat = leaf(quat_); logit_opacities = leaf(logit_); log_scales = leaf(logs_);
    if (cfg_.fused_ops)
        fopt_ = std::make_unique<fused::Adam>(std::vector<fused::Adam::Group>{{xyz, cfg_.lr_mean3d}, {rgb, cfg_.lr_rgb}, {unnorm_quat, cfg_.lr_rotation},
                                                                              {logit_opacities, cfg_.lr_opacities}, {log_scales, cfg_.lr_scales}}, 1e-15);
    else
        opt_ = make_adam({{xyz, cfg_.lr_mean3d}, {rgb, cfg_.lr_rgb}, {unnorm_quat, cfg_.lr_rotation}, {logit_opacities, cfg_.lr_opacities},
                          {log_scales, cfg_.lr_scales}});
}

std::tuple<torch::Tensor, torch::Tensor, torch::Tensor> SlamLoop::RenderPair(const torch::Tensor& Tcw, bool tracking)
{
    auto p = [&](const torch::Tensor& t) { return tracking ? t.detach() : t; };  // tracking optimises the pose only
    const auto X = p(xyz), C = p(rgb), Q = p(unnorm_quat), O = p(logit_opacities), S = p(log_scales);
    // Render.cc:750-758: camera-frame means and activations, formed once for both renders
    const auto mc = cfg_.fused_ops ? fused::to_camera(Tcw, X)
                                   : X.matmul(Tcw.slice(0, 0, 3).slice(1, 0, 3).t()) + Tcw.slice(0, 0, 3).slice(1, 3, 4).reshape({1, 3});
    const bool cached = tracking && act_.size() == 3;
    const auto opac = cached ? act_[0] : torch::sigmoid(O), scales = cached ? act_[1] : torch::exp(S), rots = cached ? act_[2] : torch::nn::functional::normalize(Q);
    const auto mean2D = torch::zeros_like(mc).requires_grad_(true);
    const int dn = dev_.index() < 0 ? 0 : dev_.index();
    if (cfg_.fused_pair) {
        auto [image, ds, radii, sur] = rasterizer_.forward_pair(mc, mean2D, opac, torch::Tensor(), C, scales, rots, torch::Tensor(), dn, tracking);
        (void)radii;
        return {image, sur, ds};
    }
    const auto z = mc.slice(1, 2, 3);
    auto col = torch::cat({z, torch::ones_like(z), torch::zeros_like(z)}, 1); // GSParamDepthUpdata, Render.cc:949-981
    if (tracking) col = col.detach();
    auto [dimg, r0, d0] = rasterizer_.forward(mc, mean2D, opac, torch::Tensor(), col, scales, rots, torch::Tensor(), dn);
    auto [image, r1, sur] = rasterizer_.forward(mc, torch::zeros_like(mc).requires_grad_(true), opac, torch::Tensor(), C, scales, rots, torch::Tensor(), dn);
    (void)r0; (void)d0; (void)r1;
    return {image, sur, dimg.slice(0, 0, 2)};
}

std::vector<double> SlamLoop::Track(const LoopFrame& frame, const torch::Tensor& Tcw_init, int iters, torch::Tensor* Tcw_best, const LoopMatches* matches)
{
    if (direct_()) return direct_track_(frame, Tcw_init, iters, Tcw_best, matches);
    // (the autograd path: the reprojection term as the reference's tensor expressions, Render.cc:1060-1086)
    torch::Tensor m_obs, m_X, m_s2, m_inl;
    double m_cx = 0, m_cy = 0;
    if (matches && matches->obs.defined() && matches->obs.size(0) > 0) {
        m_obs = matches->obs.to(dev_, torch::kFloat32).reshape({-1, 2}); m_X = matches->Xw.to(dev_, torch::kFloat32).reshape({-1, 3});
        m_s2 = matches->inv_sigma2.to(dev_, torch::kFloat32).reshape({-1});
        m_cx = matches->cx >= 0 ? matches->cx : (W_ - 1) / 2.0; m_cy = matches->cy >= 0 ? matches->cy : (H_ - 1) / 2.0;
    }
    // Gaussian::InitCameraPose (Gaussian.cc:97-150)
    const auto T0 = Tcw_init.to(dev_, torch::kFloat32);
    cam_quat_ = rot_to_quat(T0.slice(0, 0, 3).slice(1, 0, 3)).reshape({4, 1}).to(dev_).requires_grad_(true);
    cam_trans_ = T0.slice(0, 0, 3).slice(1, 3, 4).clone().reshape({3, 1}).requires_grad_(true);
    if (cfg_.fused_ops) fopt_pose_ = std::make_unique<fused::Adam>(std::vector<fused::Adam::Group>{{cam_quat_, cfg_.lr_cam_quat}, {cam_trans_, cfg_.lr_cam_quat}}, 1e-15);
    else opt_pose_ = make_adam({{cam_quat_, cfg_.lr_cam_quat}, {cam_trans_, cfg_.lr_cam_quat}});
    auto best_q = cam_quat_.detach().clone(), best_t = cam_trans_.detach().clone();
    double min_loss = std::numeric_limits<double>::infinity(), last_loss = 0.0;
    std::vector<double> history;
    const auto nan_mask = ~torch::isnan(frame.depth);
    // the map does not move while the pose is tracked: its activations are formed once per call, not once per iteration
    if (cfg_.fused_ops) { torch::NoGradGuard ng; act_ = {torch::sigmoid(logit_opacities), torch::exp(log_scales), torch::nn::functional::normalize(unnorm_quat)}; }
    for (int it = 0; it < iters; it++) {
        const auto Tcw = cfg_.fused_ops ? fused::rt2T(cam_quat_, cam_trans_) : rt2T(cam_quat_.clone(), cam_trans_.clone());
        auto [rimage, rsur, rdepth] = RenderPair(Tcw, true);
        torch::Tensor loss;
        if (cfg_.fused_ops) {                                                                            // Render.cc:1088-1105 as two launches
            loss = fused::tracking_pixel_loss(rimage, cfg_.use_sur_depth ? rsur[0] : rdepth[0], rdepth[1], frame.rgb, frame.depth,
                                              cfg_.im_weight_tracking, cfg_.depth_weight_tracking, cfg_.use_sur_depth);
        } else {
            const auto certain = ((rdepth[1] > 0.99) & nan_mask).detach();                               // Render.cc:1088-1090
            const auto image_l1 = l1_sum(rimage, frame.rgb, certain.unsqueeze(0).repeat({3, 1, 1}));
            const auto depth_l1 = l1_sum(cfg_.use_sur_depth ? rsur[0] : rdepth[0], frame.depth, certain);
            loss = cfg_.im_weight_tracking * image_l1 + cfg_.depth_weight_tracking * depth_l1;
        }
        if (m_obs.defined()) {
            const auto Xc = m_X.matmul(Tcw.slice(0, 0, 3).slice(1, 0, 3).t()) + Tcw.slice(0, 0, 3).slice(1, 3, 4).reshape({1, 3});
            const auto ex = fx_ * Xc.select(1, 0) / Xc.select(1, 2) + m_cx - m_obs.select(1, 0), ey = fy_ * Xc.select(1, 1) / Xc.select(1, 2) + m_cy - m_obs.select(1, 1);
            const auto werr = m_s2 * (ex * ex + ey * ey);
            if (it == (int)(iters / 2.0)) m_inl = (werr < 5.991).detach();
            loss = loss + cfg_.feature_weight_tracking * (m_inl.defined() ? werr.masked_select(m_inl).sum() : werr.sum());
        }
        loss.backward();
        torch::NoGradGuard ng;
        const double lv = loss.item<double>();
        history.push_back(lv);
        if (!std::isnan(lv) && lv < min_loss) { best_q = cam_quat_.detach().clone(); best_t = cam_trans_.detach().clone(); min_loss = lv; }
        if (std::fabs(last_loss - lv) < 10e-4) break;                                                    // Render.cc:1113-1114
        last_loss = lv;
        if (cfg_.fused_ops) { fopt_pose_->step(); fopt_pose_->zero_grad(); }
        else { opt_pose_->step(); opt_pose_->zero_grad(); }
    }
    act_.clear();
    if (Tcw_best) *Tcw_best = rt2T(best_q, best_t).detach();
    return history;
}

double SlamLoop::MappingIteration(const LoopFrame& fr)
{
    if (direct_()) {
        const auto l = MapFrame(fr, 1);
        return l.empty() ? std::numeric_limits<double>::quiet_NaN() : l[0]; // (an empty map renders nothing and has no loss)
    }
    const auto Tcw = fr.Tcw.to(dev_, torch::kFloat32);
    auto [rimage, rsur, rdepth] = RenderPair(Tcw, false);
    if (cfg_.fused_ops) { // Render.cc:436-471 as: pixel terms (2 launches), SSIM (1 + a sum), regularisers (2), a handful of scalar operations
        const auto pix = fused::mapping_pixel_loss(rimage, rdepth[0], rsur[0], rdepth[1], fr.rgb, fr.depth, cfg_.im_weight_mapping * cfg_.lam,
                                                   cfg_.depth_weight_mapping, cfg_.sur_depth_weight_mapping);
        const auto ssim_v = fused::ssim_mean(rimage, fr.rgb, taps_host_);
        const auto reg = fused::scale_regularisers(log_scales, 0.1 * cfg_.scene_radius, cfg_.reg_long_weight, cfg_.reg_scalar_weight);
        const auto loss = pix + (cfg_.im_weight_mapping * (1 - cfg_.lam)) * (1.0 - ssim_v) + reg;
        loss.backward();
        torch::NoGradGuard ng;
        fopt_->step(); fopt_->zero_grad();
        return loss.item<double>();
    }
    const auto valid = (fr.depth > 0).detach();
    const auto valid_sur = ((fr.depth > 0) & (rdepth[1] > 0.99)).detach();
    const auto ssim_v = cfg_.fused_ops ? fused::ssim_mean(rimage, fr.rgb, taps_host_) : ssim(rimage, fr.rgb, taps_);
    const auto image_loss = cfg_.lam * torch::abs(rimage - fr.rgb).mean() + (1 - cfg_.lam) * (1.0 - ssim_v);
    const auto dd = torch::abs(rdepth[0] - fr.depth);
    const auto depth_loss = torch::where(valid, dd, torch::zeros_like(dd)).sum() / valid.sum();
    const auto ds = torch::abs(rsur[0] - fr.depth);
    // (an empty selection: zero, not the reference's NaN — the same deliberate deviation as harness.py documents)
    const auto sur_loss = torch::where(valid_sur, ds, torch::zeros_like(ds)).sum() / valid_sur.sum().clamp_min(1);
    const double max_scalar = 0.1 * cfg_.scene_radius;
    const auto sc = torch::exp(log_scales);
    const auto w = (sc > max_scalar).sum(1).to(sc.dtype());                                              // Render.cc:449-462
    const auto mx = std::get<0>(sc.max(1)), mn = std::get<0>(sc.min(1));
    const auto cnt = w.sum();
    const auto reg_scalar = (w * (mx - max_scalar)).sum();
    const auto spread = (w * (mx - mn)).sum();
    const auto reg_long = torch::where(cnt > 0, spread / cnt.clamp_min(1), torch::zeros_like(spread));
    const auto loss = cfg_.im_weight_mapping * image_loss + cfg_.depth_weight_mapping * depth_loss + cfg_.sur_depth_weight_mapping * sur_loss +
                      cfg_.reg_long_weight * reg_long + cfg_.reg_scalar_weight * reg_scalar;
    loss.backward();
    torch::NoGradGuard ng;
    if (cfg_.fused_ops) { fopt_->step(); fopt_->zero_grad(); }
    else { opt_->step(); opt_->zero_grad(); }
    return loss.item<double>();
}

// ---- map growth ------------------------------------------------------------------------------------------------
std::vector<torch::Tensor*> SlamLoop::params_() { return {&xyz, &rgb, &unnorm_quat, &logit_opacities, &log_scales}; }

// The five parameter tensors are replaced by `fresh` (new leaves) in the map and in the optimiser, whose moments follow:
// extended with zeros for `added` rows (CatTensorToOptimizer, Gaussian.cc:236-258) or reduced to the rows *keep
// (PruneOptimizer, Gaussian.cc:218-234).
void SlamLoop::replace_params_(const std::vector<torch::Tensor>& fresh, int64_t added, const torch::Tensor* keep)
{
    const auto ps = params_();
    for (size_t i = 0; i < ps.size(); i++) {
        const auto leaf = fresh[i].detach().contiguous().requires_grad_(true);
        if (fopt_) {
            if (keep) fopt_->replace_selected(i, leaf, *keep);
            else fopt_->replace_extended(i, leaf, added);
        } else if (opt_) {
            auto& group = opt_->param_groups()[i];
            void* const old_key = group.params()[0].unsafeGetTensorImpl();
            auto it = opt_->state().find(old_key);
            std::unique_ptr<torch::optim::OptimizerParamState> st;
            if (it != opt_->state().end()) {
                auto as = std::make_unique<torch::optim::AdamParamState>(static_cast<torch::optim::AdamParamState&>(*it->second));
                if (keep) {
                    as->exp_avg(as->exp_avg().index_select(0, *keep));
                    as->exp_avg_sq(as->exp_avg_sq().index_select(0, *keep));
                } else {
                    auto shape = as->exp_avg().sizes().vec();
                    shape[0] = added;
                    const auto z = torch::zeros(shape, as->exp_avg().options());
                    as->exp_avg(torch::cat({as->exp_avg(), z}, 0));
                    as->exp_avg_sq(torch::cat({as->exp_avg_sq(), z}, 0));
                }
                opt_->state().erase(it);
                st = std::move(as);
            }
            group.params()[0] = leaf;
            if (st) opt_->state()[leaf.unsafeGetTensorImpl()] = std::move(st);
        }
        *ps[i] = leaf;
    }
}

void SlamLoop::AddPoints(const torch::Tensor& pts_, const torch::Tensor& cols_)
{
    torch::NoGradGuard ng;
    const auto pts = pts_.to(dev_, torch::kFloat32).contiguous(), cols = cols_.to(dev_, torch::kFloat32).contiguous();
    const int64_t n = pts.size(0);
    if (n == 0) return;
    const auto opts = torch::TensorOptions().device(dev_).dtype(torch::kFloat32);
    auto quat = torch::zeros({n, 4}, opts);
    quat.select(1, 0).fill_(1.0);                                                                       // Gaussian.cc:52-55
    const auto logit = torch::ones({n, 1}, opts);
    torch::Tensor logs;
    if (cfg_.init_scalar_method == 0 || cfg_.init_scalar_method == 1) {                                  // Gaussian.cc:59-69 (distCUDA2 on the device)
        auto sq = torch::sqrt(torch::clamp_min(distCUDA2(pts, dev_), 1e-7));
        if (cfg_.init_scalar_method == 1) sq = torch::clamp_max(sq, 8 * sq.mean());
        logs = torch::log(sq).unsqueeze(-1).repeat({1, 3});
    } else if (cfg_.init_scalar_method == 2) {                                                           // Gaussian.cc:70-74: one pixel wide at its depth
        logs = torch::log(torch::sqrt(torch::pow(pts.select(1, 2) / ((fx_ + fy_) * 0.5), 2))).unsqueeze(-1).repeat({1, 3});
    } else {
        throw std::runtime_error("Unknown Init Scalar Method");
    }
    const std::vector<torch::Tensor> fresh_rows{pts, cols, quat, logit, logs.contiguous()};
    if (!xyz.defined()) { SetMap(pts, cols, quat, logit, logs); return; }
    std::vector<torch::Tensor> fresh;
    const auto ps = params_();
    for (size_t i = 0; i < ps.size(); i++) fresh.push_back(torch::cat({ps[i]->detach(), fresh_rows[i]}, 0));
    replace_params_(fresh, n, nullptr);
}

int64_t SlamLoop::AddGaussians(const LoopFrame& frame)
{
    torch::NoGradGuard ng;
    const auto Tcw = frame.Tcw.to(dev_, torch::kFloat32);
    // (sharded: the mask is taken on the composite of all ranks' layers — the same on every rank)
    auto [rim, rsur, rdep] = shard_ ? RenderComposite(Tcw) : RenderPair(Tcw, true);
    (void)rsur;
    const auto gray = (rim[0] * 299 + rim[1] * 587 + rim[2] * 114) / 1000;                               // Render.cc:560-561
    const auto black = gray < 50 / 255.0;
    const auto diff = torch::abs(frame.depth - rdep[0]);
    const auto dmask = (diff < 0.05) & (frame.depth > 0) & (rdep[0] > 0);                                // :566-567
    double th = 0.0;
    if (dmask.any().item<bool>()) {
        const auto vals = diff.masked_select(dmask);
        th = (vals.sum() / dmask.sum()).item<double>() + cfg_.median_mul * vals.median().item<double>(); // :569-573
    }
    if (th < 0.01) th = 0.01;
    auto add = ((~(rdep[1] > 0.99)) & black & (diff > th)) | (rdep[1] < 0.8);                             // :579-581
    add = add & (frame.depth > 0);                                                                        // ProjectPixel: z > 0 (:632)
    const auto vu = torch::nonzero(add);
    const int64_t n = vu.size(0);
    if (n == 0) return 0;
    const auto v = vu.select(1, 0), u = vu.select(1, 1);
    const auto z = frame.depth.index({v, u});
    const double cx = (W_ - 1) / 2.0, cy = (H_ - 1) / 2.0;
    const auto pc = torch::stack({(u.to(torch::kFloat32) - cx) * z / fx_, (v.to(torch::kFloat32) - cy) * z / fy_, z}, 1);  // :634-637
    const auto Twc = torch::inverse(Tcw);
    const auto pw = pc.matmul(Twc.slice(0, 0, 3).slice(1, 0, 3).t()) + Twc.slice(0, 0, 3).slice(1, 3, 4).reshape({1, 3});
    const auto cols = frame.rgb.index({torch::indexing::Slice(), v, u}).t().contiguous();
    if (shard_) { // owner rule: a new Gaussian belongs to the rank whose cell holds the back-projected point
        const auto mine = torch::nonzero(shard_cells_(pw) == shard_rank_()).squeeze(-1);
        AddPoints(pw.index_select(0, mine), cols.index_select(0, mine));
        return mine.size(0);
    }
    AddPoints(pw, cols);
    return n;
}

torch::Tensor SlamLoop::ExportRows()
{
    torch::NoGradGuard ng;
    if (!fopt_) throw std::runtime_error("ExportRows needs LoopConfig::fused_ops");
    const int64_t n = size();
    std::vector<torch::Tensor> cols;
    const auto ps = params_();
    for (auto* p : ps) cols.push_back(p->detach().reshape({n, -1}));
    for (size_t i = 0; i < ps.size(); i++) cols.push_back(fopt_->exp_avg(i).reshape({n, -1}));
    for (size_t i = 0; i < ps.size(); i++) cols.push_back(fopt_->exp_avg_sq(i).reshape({n, -1}));
    return torch::cat(cols, 1);
}

void SlamLoop::ReplaceRows(const torch::Tensor& keep_, const torch::Tensor& arrivals_)
{
    torch::NoGradGuard ng;
    if (!fopt_) throw std::runtime_error("ReplaceRows needs LoopConfig::fused_ops");
    const auto keep = keep_.to(dev_, torch::kInt64).contiguous();
    const auto arr = arrivals_.to(dev_, torch::kFloat32).contiguous();
    const int64_t k = arr.size(0), nk = keep.size(0);
    const int64_t widths[5] = {3, 3, 4, 1, 3};
    if (k > 0 && arr.size(1) != 42) throw std::runtime_error("ReplaceRows: arrivals must be [k, 42]");
    std::vector<torch::Tensor> kept;
    for (auto* p : params_()) kept.push_back(p->detach().index_select(0, keep));
    replace_params_(kept, 0, &keep);                     // the rows that stay, with their moments
    if (k == 0) return;
    std::vector<torch::Tensor> grown;
    int64_t off = 0;
    const auto ps = params_();
    for (size_t i = 0; i < ps.size(); i++) {
        grown.push_back(torch::cat({ps[i]->detach(), arr.slice(1, off, off + widths[i]).reshape({k, widths[i]})}, 0));
        off += widths[i];
    }
    replace_params_(grown, k, nullptr);                  // the rows that arrive (moments: zeros for now)
    for (int part = 0; part < 2; part++) {               // ... and their own moments behind the zeros
        off = 14 * (part + 1);
        for (size_t i = 0; i < ps.size(); i++) {
            auto& m = part == 0 ? fopt_->exp_avg(i) : fopt_->exp_avg_sq(i);
            m.slice(0, nk, nk + k).copy_(arr.slice(1, off, off + widths[i]).reshape({k, widths[i]}));
            off += widths[i];
        }
    }
}

int64_t SlamLoop::PruneLowOpacity()
{
    torch::NoGradGuard ng;
    const auto remove = (torch::sigmoid(logit_opacities) < cfg_.prune_opacities).squeeze(-1);             // Gaussian.cc:180-185
    const int64_t n = remove.sum().item<int64_t>();
    if (n == 0) return 0;                                                                                  // Render.cc:606
    const auto keep = torch::nonzero(~remove).squeeze(-1);                                                 // Gaussian.cc:206-208
    std::vector<torch::Tensor> fresh;
    for (auto* p : params_()) fresh.push_back(p->detach().index_select(0, keep));
    replace_params_(fresh, 0, &keep);
    return n;
}

} // namespace ORB_SLAM2
