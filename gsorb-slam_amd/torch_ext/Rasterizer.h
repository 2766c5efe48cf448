// Rasterizer.h — libtorch (ROCm) host layer of the MI355X rasterizer.
//
// Mirrors the reference's C++ operator interface one to one — same namespace, names,
// argument order and defaults as include/Rasterizer.cuh:24-125,284-380 and
// src/Rasterizer.cu:8-383 of the reference tree — so that src/Render.cc compiles and runs
// against it unchanged. Underneath, every call goes through the C ABI of include/gsr.h to
// the hand-written HIP kernels; torch only owns memory, streams and autograd.
//
// Differences that are NOT visible through the interface:
//   * settings scalars stay on the host (the reference round-trips 7 of them through device
//     tensors and .item(), include/Rasterizer.cuh:151-157);
//   * gradient buffers are not zero-filled first (the C ABI writes every element);
//   * the work runs on torch's current HIP stream instead of the legacy default stream;
//   * no global state: two host threads may render concurrently (reference Viewer2.cc:256-263).
#pragma once

#include <torch/torch.h>

#include <stdexcept>
#include <tuple>

namespace ORB_SLAM2 {

// include/Rasterizer.cuh:27-48 (19 parameters -> 7-tuple)
std::tuple<int, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>
RasterizeGaussiansCUDA(const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& colors,
                       const torch::Tensor& opacity, const torch::Tensor& scales, const torch::Tensor& rotations,
                       const float scale_modifier, const torch::Tensor& cov3D_precomp,
                       const torch::Tensor& viewmatrix, const torch::Tensor& projmatrix, const float tan_fovx,
                       const float tan_fovy, const int image_height, const int image_width, const torch::Tensor& sh,
                       const int degree, const torch::Tensor& campos, const bool prefiltered, const int device_num);

// include/Rasterizer.cuh:50-71 (20 parameters -> 8 gradient tensors, the reference's shapes: dL_dcov3D is [P,6] and filled
// whichever parameterisation is used, src/Rasterizer.cu:253-261,265-293); stateless: any number of calls per forward
std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor,
           torch::Tensor>
RasterizeGaussiansBackwardCUDA(const torch::Tensor& background, const torch::Tensor& means3D,
                               const torch::Tensor& radii, const torch::Tensor& colors, const torch::Tensor& scales,
                               const torch::Tensor& rotations, const float scale_modifier,
                               const torch::Tensor& cov3D_precomp, const torch::Tensor& viewmatrix,
                               const torch::Tensor& projmatrix, const float tan_fovx, const float tan_fovy,
                               const torch::Tensor& dL_dout_color, const torch::Tensor& sh, const int degree,
                               const torch::Tensor& campos, const torch::Tensor& geomBuffer, const int R,
                               const torch::Tensor& binningBuffer, const torch::Tensor& imageBuffer);

// RasterizeGaussiansBackwardCUDA with an explicit stage mask (gsr_backward_args.stages, include/gsr.h): lets a
// caller that runs one backward per forward skip the re-zero of the per-splat accumulators. This is the opt-in LEAN form the
// autograd nodes use: with scales + rotations nothing consumes dL_dcov3D (the node would hand it to the absent cov3Ds_precomp
// input), so it is returned as [0,6] and never stored (40 B per Gaussian of traffic less).
std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor,
           torch::Tensor>
RasterizeGaussiansBackwardStaged(const torch::Tensor& background, const torch::Tensor& means3D,
                                 const torch::Tensor& radii, const torch::Tensor& colors, const torch::Tensor& scales,
                                 const torch::Tensor& rotations, const float scale_modifier,
                                 const torch::Tensor& cov3D_precomp, const torch::Tensor& viewmatrix,
                                 const torch::Tensor& projmatrix, const float tan_fovx, const float tan_fovy,
                                 const torch::Tensor& dL_dout_color, const torch::Tensor& sh, const int degree,
                                 const torch::Tensor& campos, const torch::Tensor& geomBuffer, const int R,
                                 const torch::Tensor& binningBuffer, const torch::Tensor& imageBuffer, const int stages);

// include/Rasterizer.cuh:73-76
torch::Tensor markVisible(torch::Tensor& means3D, torch::Tensor& viewmatrix, torch::Tensor& projmatrix);

// include/Rasterizer.cuh:79-91
struct GaussianRasterizationSettings {
    int image_height;
    int image_width;
    float tanfovx;
    float tanfovy;
    torch::Tensor bg;
    float scale_modifier;
    torch::Tensor viewmatrix;
    torch::Tensor projmatrix;
    int sh_degree;
    torch::Tensor camera_center;
    bool prefiltered;
};

// include/Rasterizer.cuh:93-97
torch::Tensor filter_radii(torch::Tensor means3D, torch::Tensor scales, torch::Tensor rotations, int device_num,
                           GaussianRasterizationSettings raster_settings);

// include/Rasterizer.cuh:99-112
torch::Tensor RasterizeGaussiansfilterCUDA(const torch::Tensor& means3D, const torch::Tensor& scales,
                                           const torch::Tensor& rotations, const float scale_modifier,
                                           const torch::Tensor& viewmatrix, const torch::Tensor& projmatrix,
                                           const float tan_fovx, const float tan_fovy, const int image_height,
                                           const int image_width, const bool prefiltered, int device_num);

// include/Rasterizer.cuh:116-125: {color [3,H,W], radii [P] int32, depth [1,H,W]}
torch::autograd::tensor_list rasterize_gaussians(torch::Tensor means3D, torch::Tensor means2D, torch::Tensor sh,
                                                 torch::Tensor colors_precomp, torch::Tensor opacities,
                                                 torch::Tensor scales, torch::Tensor rotations,
                                                 torch::Tensor cov3Ds_precomp, int device_num,
                                                 GaussianRasterizationSettings raster_settings);

// include/Rasterizer.cuh:127-282. Differentiable inputs are the first eight tensors; depth
// and radii carry no gradient (include/Rasterizer.cuh:210-211, reference README:13).
class _RasterizeGaussians : public torch::autograd::Function<_RasterizeGaussians> {
public:
    static torch::autograd::tensor_list forward(torch::autograd::AutogradContext* ctx, torch::Tensor means3D,
                                                torch::Tensor means2D, torch::Tensor sh,
                                                torch::Tensor colors_precomp, torch::Tensor opacities,
                                                torch::Tensor scales, torch::Tensor rotations,
                                                torch::Tensor cov3Ds_precomp, torch::Tensor bg,
                                                torch::Tensor viewmatrix, torch::Tensor projmatrix,
                                                torch::Tensor camera_center, int64_t image_height,
                                                int64_t image_width, double tanfovx, double tanfovy,
                                                double scale_modifier, int64_t sh_degree, bool prefiltered,
                                                int64_t device_num);
    static torch::autograd::tensor_list backward(torch::autograd::AutogradContext* ctx,
                                                 torch::autograd::tensor_list grad_outputs);
};

// ---- The fused pair (new capability, not in the reference). GSORB-SLAM renders every view twice with the same geometry:
// colours, then colors_precomp = [z, 1, 0] for the alpha-blended depth and the silhouette (src/Render.cc:927-981).
// forward_pair does both in ONE pass of the rasterizer: {color [3,H,W], ds [2,H,W] (ds[0] = sum z_i alpha_i T_i with z_i the
// splat's view-space depth, ds[1] = sum alpha_i T_i; background 0), radii, depth [1,H,W] (median depth)}. color and ds are
// differentiable; the depth channel's gradient reaches means3D through z_i (what autograd does in the reference when the
// caller builds [z, 1, 0] from the camera-frame means).
std::tuple<int, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>
RasterizeGaussiansPairCUDA(const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& colors,
                           const torch::Tensor& opacity, const torch::Tensor& scales, const torch::Tensor& rotations,
                           const float scale_modifier, const torch::Tensor& cov3D_precomp,
                           const torch::Tensor& viewmatrix, const torch::Tensor& projmatrix, const float tan_fovx,
                           const float tan_fovy, const int image_height, const int image_width, const torch::Tensor& sh,
                           const int degree, const torch::Tensor& campos, const bool prefiltered, const int device_num);
std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor,
           torch::Tensor>
RasterizeGaussiansPairBackward(const torch::Tensor& background, const torch::Tensor& means3D,
                               const torch::Tensor& radii, const torch::Tensor& colors, const torch::Tensor& scales,
                               const torch::Tensor& rotations, const float scale_modifier,
                               const torch::Tensor& cov3D_precomp, const torch::Tensor& viewmatrix,
                               const torch::Tensor& projmatrix, const float tan_fovx, const float tan_fovy,
                               const torch::Tensor& dL_dout_color, const torch::Tensor& dL_dout_ds, const torch::Tensor& sh,
                               const int degree, const torch::Tensor& campos, const torch::Tensor& geomBuffer, const int R,
                               const torch::Tensor& binningBuffer, const torch::Tensor& imageBuffer, const int stages,
                               const bool detach_depth_color);
// detach_depth_color: the depth channel's colours z_i are constants in the backward (what GSORB-SLAM's tracking iterations do
// with their [z, 1, 0] colours, src/Render.cc:949-981); false: their gradient reaches means3D
torch::autograd::tensor_list rasterize_gaussians_pair(torch::Tensor means3D, torch::Tensor means2D, torch::Tensor sh,
                                                      torch::Tensor colors_precomp, torch::Tensor opacities,
                                                      torch::Tensor scales, torch::Tensor rotations,
                                                      torch::Tensor cov3Ds_precomp, int device_num,
                                                      GaussianRasterizationSettings raster_settings, bool detach_depth_color = false);
class _RasterizeGaussiansPair : public torch::autograd::Function<_RasterizeGaussiansPair> {
public:
    static torch::autograd::tensor_list forward(torch::autograd::AutogradContext* ctx, torch::Tensor means3D,
                                                torch::Tensor means2D, torch::Tensor sh,
                                                torch::Tensor colors_precomp, torch::Tensor opacities,
                                                torch::Tensor scales, torch::Tensor rotations,
                                                torch::Tensor cov3Ds_precomp, torch::Tensor bg,
                                                torch::Tensor viewmatrix, torch::Tensor projmatrix,
                                                torch::Tensor camera_center, int64_t image_height,
                                                int64_t image_width, double tanfovx, double tanfovy,
                                                double scale_modifier, int64_t sh_degree, bool prefiltered,
                                                int64_t device_num, bool detach_depth_color);
    static torch::autograd::tensor_list backward(torch::autograd::AutogradContext* ctx,
                                                 torch::autograd::tensor_list grad_outputs);
};

// include/Rasterizer.cuh:284-380
class GaussianRasterizer : torch::nn::Module {
public:
    GaussianRasterizer() {}
    GaussianRasterizer(GaussianRasterizationSettings raster_settings) : raster_settings_(raster_settings) {}

    torch::Tensor mark_visible(torch::Tensor positions)
    {
        torch::NoGradGuard no_grad;
        return markVisible(positions, raster_settings_.viewmatrix, raster_settings_.projmatrix);
    }

    std::tuple<torch::Tensor, torch::Tensor, torch::Tensor>
    forward(torch::Tensor means3D, torch::Tensor means2D, torch::Tensor opacities, torch::Tensor shs = torch::Tensor(),
            torch::Tensor colors_precomp = torch::Tensor(), torch::Tensor scales = torch::Tensor(),
            torch::Tensor rotations = torch::Tensor(), torch::Tensor cov3D_precomp = torch::Tensor(),
            int device_num = 0)
    {
        if ((shs.defined() && colors_precomp.defined()) || (!shs.defined() && !colors_precomp.defined()))
            throw std::invalid_argument("Please provide exactly one of either SHs or precomputed colors!");
        if (((scales.defined() || rotations.defined()) && cov3D_precomp.defined()) ||
            (!scales.defined() && !rotations.defined() && !cov3D_precomp.defined()))
            throw std::invalid_argument(
                "Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!");
        const torch::Device device = {torch::kCUDA, (c10::DeviceIndex)device_num};
        auto empty = [&](torch::Tensor& t) { if (!t.defined()) t = torch::empty({0}, torch::TensorOptions().device(device)); };
        empty(shs); empty(colors_precomp); empty(scales); empty(rotations); empty(cov3D_precomp);
        auto r = rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                     cov3D_precomp, device_num, raster_settings_);
        return {r[0], r[1], r[2]};
    }

    // the fused pair (see above): {color, ds, radii, depth}
    std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>
    forward_pair(torch::Tensor means3D, torch::Tensor means2D, torch::Tensor opacities, torch::Tensor shs = torch::Tensor(),
                 torch::Tensor colors_precomp = torch::Tensor(), torch::Tensor scales = torch::Tensor(),
                 torch::Tensor rotations = torch::Tensor(), torch::Tensor cov3D_precomp = torch::Tensor(), int device_num = 0,
                 bool detach_depth_color = false)
    {
        if ((shs.defined() && colors_precomp.defined()) || (!shs.defined() && !colors_precomp.defined()))
            throw std::invalid_argument("Please provide exactly one of either SHs or precomputed colors!");
        if (((scales.defined() || rotations.defined()) && cov3D_precomp.defined()) ||
            (!scales.defined() && !rotations.defined() && !cov3D_precomp.defined()))
            throw std::invalid_argument(
                "Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!");
        const torch::Device device = {torch::kCUDA, (c10::DeviceIndex)device_num};
        auto empty = [&](torch::Tensor& t) { if (!t.defined()) t = torch::empty({0}, torch::TensorOptions().device(device)); };
        empty(shs); empty(colors_precomp); empty(scales); empty(rotations); empty(cov3D_precomp);
        auto r = rasterize_gaussians_pair(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                          cov3D_precomp, device_num, raster_settings_, detach_depth_color);
        return {r[0], r[1], r[2], r[3]};
    }

    std::tuple<torch::Tensor> Visable(torch::Tensor means3D, torch::Tensor opacities,
                                      torch::Tensor scales = torch::Tensor(),
                                      torch::Tensor rotations = torch::Tensor(), int device_num = 0)
    {
        (void)opacities;
        const torch::Device device = {torch::kCUDA, (c10::DeviceIndex)device_num};
        if (!scales.defined()) scales = torch::empty({0}, torch::TensorOptions().device(device));
        if (!rotations.defined()) rotations = torch::empty({0}, torch::TensorOptions().device(device));
        return {filter_radii(means3D, scales, rotations, device_num, raster_settings_)};
    }

public:
    GaussianRasterizationSettings raster_settings_;
};

} // namespace ORB_SLAM2

// reference include/spatial.h:13 (global namespace there too): mean squared distance to the three
// nearest neighbours, used for the initial scale of inserted Gaussians (src/Gaussian.cc:59-69)
torch::Tensor distCUDA2(const torch::Tensor& points, torch::Device device);
