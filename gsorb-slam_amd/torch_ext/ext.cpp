// ext.cpp — the `_C` extension module of the Python operator, with exactly the entry points
// of the reference's Thirdparty/diff_gaussian_rasterization/ext.cpp:15-19 (same argument
// lists as rasterize_points.h:18-65: the Python twin has no device_num), plus the radii-only
// filter pass GSORB added on the C++ side.
#include <torch/extension.h>

#include "Rasterizer.h"

namespace {

int dev_index_of(const torch::Tensor& t) { return t.is_cuda() ? (int)t.device().index() : 0; }

std::tuple<int, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>
RasterizeGaussians(const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& colors,
                   const torch::Tensor& opacity, const torch::Tensor& scales, const torch::Tensor& rotations,
                   const float scale_modifier, const torch::Tensor& cov3D_precomp, const torch::Tensor& viewmatrix,
                   const torch::Tensor& projmatrix, const float tan_fovx, const float tan_fovy,
                   const int image_height, const int image_width, const torch::Tensor& sh, const int degree,
                   const torch::Tensor& campos, const bool prefiltered)
{
    return ORB_SLAM2::RasterizeGaussiansCUDA(background, means3D, colors, opacity, scales, rotations, scale_modifier,
                                             cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height,
                                             image_width, sh, degree, campos, prefiltered, dev_index_of(means3D));
}

std::tuple<int, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>
RasterizeGaussiansPair(const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& colors,
                       const torch::Tensor& opacity, const torch::Tensor& scales, const torch::Tensor& rotations,
                       const float scale_modifier, const torch::Tensor& cov3D_precomp, const torch::Tensor& viewmatrix,
                       const torch::Tensor& projmatrix, const float tan_fovx, const float tan_fovy,
                       const int image_height, const int image_width, const torch::Tensor& sh, const int degree,
                       const torch::Tensor& campos, const bool prefiltered)
{
    return ORB_SLAM2::RasterizeGaussiansPairCUDA(background, means3D, colors, opacity, scales, rotations, scale_modifier,
                                                 cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height,
                                                 image_width, sh, degree, campos, prefiltered, dev_index_of(means3D));
}

torch::Tensor FilterRadii(const torch::Tensor& means3D, const torch::Tensor& scales, const torch::Tensor& rotations,
                          const float scale_modifier, const torch::Tensor& viewmatrix,
                          const torch::Tensor& projmatrix, const float tan_fovx, const float tan_fovy,
                          const int image_height, const int image_width, const bool prefiltered)
{
    return ORB_SLAM2::RasterizeGaussiansfilterCUDA(means3D, scales, rotations, scale_modifier, viewmatrix, projmatrix,
                                                   tan_fovx, tan_fovy, image_height, image_width, prefiltered,
                                                   dev_index_of(means3D));
}

} // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    m.def("rasterize_gaussians", &RasterizeGaussians);
    m.def("rasterize_gaussians_backward", &ORB_SLAM2::RasterizeGaussiansBackwardCUDA);
    m.def("rasterize_gaussians_backward_staged", &ORB_SLAM2::RasterizeGaussiansBackwardStaged);
    m.def("rasterize_gaussians_pair", &RasterizeGaussiansPair);   // the fused colour + depth / silhouette render (Rasterizer.h)
    m.def("rasterize_gaussians_pair_backward", &ORB_SLAM2::RasterizeGaussiansPairBackward);
    m.def("mark_visible", &ORB_SLAM2::markVisible);
    m.def("filter_radii", &FilterRadii);
    m.def("distCUDA2", [](const torch::Tensor& points) { return distCUDA2(points, points.is_cuda() ? points.device() : torch::Device(torch::kCUDA, 0)); });
}
