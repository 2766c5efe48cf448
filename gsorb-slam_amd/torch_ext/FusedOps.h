// FusedOps.h — libtorch (C++) autograd wrappers of the loop kernels of the C ABI (include/gsr.h: gsr_to_camera / gsr_pose_grad,
// gsr_pose_from_quat[_backward], gsr_ssim_forward / _backward, gsr_adam_step): what SlamLoop uses in place of the reference's
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

// torch::optim::Adam(lr, betas (0.9, 0.999), eps) without weight decay / amsgrad, one kernel per parameter tensor
class Adam {
public:
    struct Group { torch::Tensor param; double lr; };
    Adam(std::vector<Group> groups, double eps) : groups_(std::move(groups)), eps_(eps) {}
    void step();
    void zero_grad();
private:
    struct State { torch::Tensor exp_avg, exp_avg_sq; int step = 0; };
    std::vector<Group> groups_;
    std::vector<State> state_;
    double eps_;
};

} // namespace fused
} // namespace ORB_SLAM2
