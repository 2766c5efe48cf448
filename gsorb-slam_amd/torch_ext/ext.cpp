// ext.cpp — the `_C` extension module of the Python operator, with exactly the entry points
// of the reference's Thirdparty/diff_gaussian_rasterization/ext.cpp:15-19 (same argument
// lists as rasterize_points.h:18-65: the Python twin has no device_num), plus the radii-only
// filter pass GSORB added on the C++ side.
#include <torch/extension.h>
#include <torch/csrc/distributed/c10d/ProcessGroup.hpp>

#include "Rasterizer.h"
#include "SlamLoop.h"

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

// LoopConfig from keyword arguments (the field names of SlamLoop.h)
ORB_SLAM2::LoopConfig loop_config(const py::dict& kw)
{
    ORB_SLAM2::LoopConfig c;
    for (auto item : kw) {
        const std::string k = py::cast<std::string>(item.first);
        const py::handle v = item.second;
#define GSR_F(name, T) if (k == #name) { c.name = py::cast<T>(v); continue; }
        GSR_F(im_weight_mapping, double) GSR_F(depth_weight_mapping, double) GSR_F(sur_depth_weight_mapping, double) GSR_F(reg_long_weight, double)
        GSR_F(reg_scalar_weight, double) GSR_F(lam, double) GSR_F(lr_mean3d, double) GSR_F(lr_rgb, double) GSR_F(lr_rotation, double)
        GSR_F(lr_opacities, double) GSR_F(lr_scales, double) GSR_F(lr_cam_quat, double) GSR_F(im_weight_tracking, double) GSR_F(depth_weight_tracking, double) GSR_F(feature_weight_tracking, double)
        GSR_F(scale_modifier, double) GSR_F(scene_radius, double) GSR_F(prune_opacities, double) GSR_F(median_mul, double) GSR_F(init_scalar_method, int)
        GSR_F(use_sur_depth, bool) GSR_F(fused_pair, bool) GSR_F(fused_ops, bool) GSR_F(direct, bool) GSR_F(binning_capacity, int64_t) GSR_F(fused_loss, bool)
        GSR_F(fused_update, bool) GSR_F(band_exchange, bool)
#undef GSR_F
        throw std::invalid_argument("SlamLoop: unknown configuration field " + k);
    }
    return c;
}

} // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    // ORB_SLAM2::SlamLoop (torch_ext/SlamLoop.h): the tracking / mapping / map-growth loops of Render.cc as the C++ driver, for callers that hold
    // their frames as tensors — and the way a torch.distributed process group reaches the sharded loop (SetShard)
    py::class_<ORB_SLAM2::SlamLoop, std::shared_ptr<ORB_SLAM2::SlamLoop>>(m, "SlamLoop")
        .def(py::init([](int width, int height, float fx, float fy, const torch::Device& device, const py::kwargs& kw) {
                 return std::make_shared<ORB_SLAM2::SlamLoop>(loop_config(kw), width, height, fx, fy, device);
             }), py::arg("width"), py::arg("height"), py::arg("fx"), py::arg("fy"), py::arg("device"))
        .def("set_map", &ORB_SLAM2::SlamLoop::SetMap)
        .def("set_shard", [](ORB_SLAM2::SlamLoop& l, const py::object& group, int rank, int world, const torch::Tensor& kd_nodes) {
                 c10::intrusive_ptr<c10d::ProcessGroup> pg;
                 if (!group.is_none()) pg = py::cast<c10::intrusive_ptr<c10d::ProcessGroup>>(group);
                 l.SetShard(pg, rank, world, kd_nodes);
             }, py::arg("group"), py::arg("rank"), py::arg("world"), py::arg("kd_nodes"))
        .def("track", [](ORB_SLAM2::SlamLoop& l, const torch::Tensor& rgb, const torch::Tensor& depth, const torch::Tensor& Tcw_init, int iters,
                         const std::optional<torch::Tensor>& obs, const std::optional<torch::Tensor>& Xw, const std::optional<torch::Tensor>& inv_sigma2,
                         double cx, double cy) {
                 torch::Tensor best;
                 const bool with = obs.has_value() && Xw.has_value() && inv_sigma2.has_value() && obs->numel() > 0;
                 const ORB_SLAM2::LoopMatches m{with ? *obs : torch::Tensor(), with ? *Xw : torch::Tensor(), with ? *inv_sigma2 : torch::Tensor(), cx, cy};
                 const auto h = l.Track(ORB_SLAM2::LoopFrame{rgb, depth, Tcw_init}, Tcw_init, iters, &best, with ? &m : nullptr);
                 return std::make_pair(h, best);
             }, py::arg("rgb"), py::arg("depth"), py::arg("Tcw_init"), py::arg("iters"), py::arg("obs") = py::none(), py::arg("Xw") = py::none(),
             py::arg("inv_sigma2") = py::none(), py::arg("cx") = -1.0, py::arg("cy") = -1.0, py::call_guard<py::gil_scoped_release>())
        .def("map_frame", [](ORB_SLAM2::SlamLoop& l, const torch::Tensor& rgb, const torch::Tensor& depth, const torch::Tensor& Tcw, int iters) {
                 return l.MapFrame(ORB_SLAM2::LoopFrame{rgb, depth, Tcw}, iters);
             }, py::call_guard<py::gil_scoped_release>())
        .def("mapping_iteration", [](ORB_SLAM2::SlamLoop& l, const torch::Tensor& rgb, const torch::Tensor& depth, const torch::Tensor& Tcw) {
                 return l.MappingIteration(ORB_SLAM2::LoopFrame{rgb, depth, Tcw});
             }, py::call_guard<py::gil_scoped_release>())
        .def("add_gaussians", [](ORB_SLAM2::SlamLoop& l, const torch::Tensor& rgb, const torch::Tensor& depth, const torch::Tensor& Tcw) {
                 return l.AddGaussians(ORB_SLAM2::LoopFrame{rgb, depth, Tcw});
             }, py::call_guard<py::gil_scoped_release>())
        .def("prune_low_opacity", &ORB_SLAM2::SlamLoop::PruneLowOpacity, py::call_guard<py::gil_scoped_release>())
        .def("render_composite", &ORB_SLAM2::SlamLoop::RenderComposite, py::call_guard<py::gil_scoped_release>())
        .def("shard_render_step", &ORB_SLAM2::SlamLoop::ShardRenderStep, py::arg("Tcw"), py::arg("G"), py::arg("preflight") = -1, py::call_guard<py::gil_scoped_release>())
        .def("export_rows", &ORB_SLAM2::SlamLoop::ExportRows)
        .def("replace_rows", &ORB_SLAM2::SlamLoop::ReplaceRows)
        .def("last_pose_sums", &ORB_SLAM2::SlamLoop::LastPoseSums)
        .def("shard_transport", &ORB_SLAM2::SlamLoop::ShardTransport)
        .def("size", &ORB_SLAM2::SlamLoop::size)
        .def("params", [](ORB_SLAM2::SlamLoop& l) { return std::vector<torch::Tensor>{l.xyz, l.rgb, l.unnorm_quat, l.logit_opacities, l.log_scales}; });

    m.def("rasterize_gaussians", &RasterizeGaussians);
    m.def("rasterize_gaussians_backward", &ORB_SLAM2::RasterizeGaussiansBackwardCUDA);
    m.def("rasterize_gaussians_backward_staged", &ORB_SLAM2::RasterizeGaussiansBackwardStaged);
    m.def("rasterize_gaussians_pair", &RasterizeGaussiansPair);   // the fused colour + depth / silhouette render (Rasterizer.h)
    m.def("rasterize_gaussians_pair_backward", &ORB_SLAM2::RasterizeGaussiansPairBackward);
    m.def("mark_visible", &ORB_SLAM2::markVisible);
    m.def("filter_radii", &FilterRadii);
    m.def("distCUDA2", [](const torch::Tensor& points) { return distCUDA2(points, points.is_cuda() ? points.device() : torch::Device(torch::kCUDA, 0)); });
}
