// FusedOps.h — libtorch (C++) autograd wrappers of the loop kernels of the C ABI (include/gsr.h: gsr_to_camera / gsr_pose_grad,
// gsr_pose_from_quat[_backward], gsr_ssim_forward / _backward, gsr_adam_step, gsr_pixel_loss[_backward], gsr_scale_reg[_backward]): what SlamLoop uses in place of the reference's
// n 4x4 bmm's (src/Render.cc:750-752), its ~120 scalar-tensor rt2T launches (include/Utils.h:56-77), its five depthwise 11x11
// convolutions (src/Utils.cc:77-100) and torch::optim::Adam's ~12 elementwise passes per tensor (src/Gaussian.cc:144-175).
// The Python twins are in gsorb-slam_amd/capi.py.
#pragma once

#include <torch/torch.h>

#include <vector>

namespace ORB_SLAM2 {
namespace fused {

// camera-frame means [n,3] of world-frame means X under Tcw [4,4]; differentiable in both
torch::Tensor to_camera(const torch::Tensor& Tcw, const torch::Tensor& X);
// Tcw [4,4] from an un-normalised quaternion (r,x,y,z) [4,1] and a translation [3,1]
torch::Tensor rt2T(const torch::Tensor& quat, const torch::Tensor& trans);
// mean SSIM of img1 vs img2 ([C,H,W]) with an 11-tap separable window (host floats); gradient w.r.t. img1
torch::Tensor ssim_mean(const torch::Tensor& img1, const torch::Tensor& img2, const std::vector<float>& taps11);

// The pixel terms of the tracking loss (src/Render.cc:1088-1105): w_image * sum_M |image - rgb| + w_depth * sum_M |depth - frame depth|,
// M = sil > 0.99 && !isnan(frame depth); depth_is_surface: `depth` is the median-depth plane (no gradient). Two launches forwards, one backwards.
torch::Tensor tracking_pixel_loss(const torch::Tensor& image, const torch::Tensor& depth, const torch::Tensor& sil, const torch::Tensor& frame_rgb,
                                  const torch::Tensor& frame_depth, double w_image, double w_depth, bool depth_is_surface);
// The pixel terms of the mapping loss (src/Render.cc:436-471): w_l1 * mean |image - rgb| + w_depth * mean_{fd > 0} |depth - fd|
// + w_sur * mean_{fd > 0 && sil > 0.99} |sur - fd| (an empty surface mask: 0; sur has no gradient)
torch::Tensor mapping_pixel_loss(const torch::Tensor& image, const torch::Tensor& depth, const torch::Tensor& sur, const torch::Tensor& sil,
                                 const torch::Tensor& frame_rgb, const torch::Tensor& frame_depth, double w_l1, double w_depth, double w_sur);
// w_long * reg_long + w_scalar * reg_scalar of src/Render.cc:449-462 for log_scales [n,3] (limit = 0.1 * scene radius)
torch::Tensor scale_regularisers(const torch::Tensor& log_scales, double limit, double w_long, double w_scalar);

// torch::optim::Adam(lr, betas (0.9, 0.999), eps) without weight decay / amsgrad, one kernel per parameter tensor
class Adam {
public:
    struct Group { torch::Tensor param; double lr; };
    Adam(std::vector<Group> groups, double eps) : groups_(std::move(groups)), eps_(eps) {}
    void step();
    void zero_grad();
    // Map growth (Gaussian::CatTensorToOptimizer / PruneOptimizer, src/Gaussian.cc:218-258): group i's parameter becomes
    // `param` (a new leaf); its moments are extended with zeros for `added` new rows / reduced to the rows `keep`
    void replace_extended(size_t i, const torch::Tensor& param, int64_t added);
    void replace_selected(size_t i, const torch::Tensor& param, const torch::Tensor& keep);
    // For the loops that step through gsr_map_update (DirectLoop.cpp): the moments of group i (created as zeros on first use),
    // its 1-based step count after `advance` more steps, its learning rate.
    torch::Tensor& exp_avg(size_t i) { ensure_state_(i); return state_[i].exp_avg; }
    torch::Tensor& exp_avg_sq(size_t i) { ensure_state_(i); return state_[i].exp_avg_sq; }
    int advance_step(size_t i) { ensure_state_(i); return ++state_[i].step; }
    void retract_step(size_t i) { ensure_state_(i); --state_[i].step; }
    double lr(size_t i) const { return groups_.at(i).lr; }
    double eps() const { return eps_; }
    const torch::Tensor& param(size_t i) const { return groups_.at(i).param; }
private:
    void ensure_state_(size_t i);
    struct State { torch::Tensor exp_avg, exp_avg_sq; int step = 0; };
    std::vector<Group> groups_;
    std::vector<State> state_;
    double eps_;
};

} // namespace fused
} // namespace ORB_SLAM2
