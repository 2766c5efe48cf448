// Rasterizer.cpp — implementation of the libtorch host layer (see Rasterizer.h) on top of the
// C ABI (include/gsr.h). Follows the call pattern of the reference's src/Rasterizer.cu:8-383
// and Thirdparty/diff_gaussian_rasterization/rasterize_points.cu:27-215, minus their
// per-call device round trips and zero-fills.
#include "Rasterizer.h"

#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>

#include "../../include/gsr.h"

namespace ORB_SLAM2 {

namespace {

constexpr int kChannels = 3; // reference config.h:15

const float* fptr(const torch::Tensor& t) { return t.numel() == 0 ? nullptr : t.data_ptr<float>(); }

void* current_stream(const torch::Device& dev)
{
    return (void*)c10::hip::getCurrentHIPStream(dev.index()).stream();
}

void check(int rc, const char* what)
{
    if (rc >= 0) return;
    if (rc == GSR_EINVAL) throw std::invalid_argument(std::string(what) + ": " + gsr_error_string(rc));
    std::string msg = std::string(what) + ": " + gsr_error_string(rc);
    if (rc == GSR_EHIP) msg += std::string(" (") + gsr_last_hip_error() + ")";
    throw std::runtime_error(msg);
}

// replaces resizeFunctional (src/Rasterizer.cu:127-134): the blob is resized, not zeroed
char* resize_blob(void* user, size_t n)
{
    auto* t = static_cast<torch::Tensor*>(user);
    t->resize_({(int64_t)(n ? n : 1)});
    return reinterpret_cast<char*>(t->data_ptr());
}

torch::Tensor contig_f32(const torch::Tensor& t, const torch::Device& dev)
{
    if (!t.defined()) return torch::empty({0}, torch::TensorOptions().device(dev).dtype(torch::kFloat32));
    return t.to(dev, torch::kFloat32).contiguous();
}

} // namespace

namespace {
std::tuple<int, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>
forward_impl(const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& colors,
             const torch::Tensor& opacity, const torch::Tensor& scales, const torch::Tensor& rotations,
             const float scale_modifier, const torch::Tensor& cov3D_precomp,
             const torch::Tensor& viewmatrix, const torch::Tensor& projmatrix, const float tan_fovx,
             const float tan_fovy, const int image_height, const int image_width, const torch::Tensor& sh,
             const int degree, const torch::Tensor& campos, const bool prefiltered, const int device_num, torch::Tensor* out_ds)
{
    if (means3D.ndimension() != 2 || means3D.size(1) != 3) {
        AT_ERROR("means3D must have dimensions (num_points, 3)"); // src/Rasterizer.cu:158-160
    }
    const int P = (int)means3D.size(0), H = image_height, W = image_width;
    const torch::Device device(torch::kCUDA, (c10::DeviceIndex)device_num);
    c10::DeviceGuard guard(device);
    const auto fopt = torch::TensorOptions().device(device).dtype(torch::kFloat32);

    torch::Tensor out_color = torch::empty({kChannels, H, W}, fopt);
    torch::Tensor radii = torch::empty({P}, fopt.dtype(torch::kInt32));
    torch::Tensor out_depth = torch::empty({1, H, W}, fopt);
    if (out_ds) *out_ds = torch::empty({2, H, W}, fopt);
    const auto bopt = torch::TensorOptions().device(device).dtype(torch::kByte);
    torch::Tensor geomBuffer = torch::empty({0}, bopt), binningBuffer = torch::empty({0}, bopt),
                  imgBuffer = torch::empty({0}, bopt);

    const torch::Tensor bg = contig_f32(background, device), m3 = contig_f32(means3D, device),
                        col = contig_f32(colors, device), op = contig_f32(opacity, device),
                        sc = contig_f32(scales, device), rot = contig_f32(rotations, device),
                        cov = contig_f32(cov3D_precomp, device), vm = contig_f32(viewmatrix, device),
                        pm = contig_f32(projmatrix, device), shc = contig_f32(sh, device),
                        cp = contig_f32(campos, device);
    int M = 0;
    if (shc.numel() != 0) M = (int)shc.size(1);

    gsr_forward_args a{};
    a.P = P; a.D = degree; a.M = M;
    a.background = fptr(bg);
    a.width = W; a.height = H;
    a.means3D = fptr(m3); a.shs = fptr(shc); a.colors_precomp = fptr(col); a.opacities = fptr(op);
    a.scales = fptr(sc); a.scale_modifier = scale_modifier; a.rotations = fptr(rot); a.cov3D_precomp = fptr(cov);
    a.viewmatrix = fptr(vm); a.projmatrix = fptr(pm); a.cam_pos = fptr(cp);
    a.tan_fovx = tan_fovx; a.tan_fovy = tan_fovy; a.prefiltered = prefiltered ? 1 : 0;
    a.out_color = out_color.data_ptr<float>(); a.out_depth = out_depth.data_ptr<float>();
    a.radii = P ? radii.data_ptr<int>() : nullptr;
    a.profile_events = nullptr;
    a.band_y0 = a.band_y1 = 0;
    a.out_ds = out_ds ? out_ds->data_ptr<float>() : nullptr;
    const int rendered = gsr_forward(&a, resize_blob, &geomBuffer, resize_blob, &binningBuffer, resize_blob,
                                     &imgBuffer, current_stream(device));
    check(rendered, "RasterizeGaussiansCUDA");
    return std::make_tuple(rendered, out_color, radii, geomBuffer, binningBuffer, imgBuffer, out_depth);
}
} // namespace

std::tuple<int, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>
RasterizeGaussiansCUDA(const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& colors,
                       const torch::Tensor& opacity, const torch::Tensor& scales, const torch::Tensor& rotations,
                       const float scale_modifier, const torch::Tensor& cov3D_precomp,
                       const torch::Tensor& viewmatrix, const torch::Tensor& projmatrix, const float tan_fovx,
                       const float tan_fovy, const int image_height, const int image_width, const torch::Tensor& sh,
                       const int degree, const torch::Tensor& campos, const bool prefiltered, const int device_num)
{
    return forward_impl(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
                        projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos, prefiltered, device_num,
                        nullptr);
}

std::tuple<int, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>
RasterizeGaussiansPairCUDA(const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& colors,
                           const torch::Tensor& opacity, const torch::Tensor& scales, const torch::Tensor& rotations,
                           const float scale_modifier, const torch::Tensor& cov3D_precomp,
                           const torch::Tensor& viewmatrix, const torch::Tensor& projmatrix, const float tan_fovx,
                           const float tan_fovy, const int image_height, const int image_width, const torch::Tensor& sh,
                           const int degree, const torch::Tensor& campos, const bool prefiltered, const int device_num)
{
    torch::Tensor ds;
    auto r = forward_impl(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
                          projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos, prefiltered, device_num, &ds);
    return std::tuple_cat(r, std::make_tuple(ds));
}

std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor,
           torch::Tensor>
RasterizeGaussiansBackwardStaged(const torch::Tensor& background, const torch::Tensor& means3D,
                                 const torch::Tensor& radii, const torch::Tensor& colors, const torch::Tensor& scales,
                                 const torch::Tensor& rotations, const float scale_modifier,
                                 const torch::Tensor& cov3D_precomp, const torch::Tensor& viewmatrix,
                                 const torch::Tensor& projmatrix, const float tan_fovx, const float tan_fovy,
                                 const torch::Tensor& dL_dout_color, const torch::Tensor& sh, const int degree,
                                 const torch::Tensor& campos, const torch::Tensor& geomBuffer, const int R,
                                 const torch::Tensor& binningBuffer, const torch::Tensor& imageBuffer, const int stages)
{
    return RasterizeGaussiansPairBackward(background, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp,
                                          viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color, torch::Tensor(), sh, degree, campos,
                                          geomBuffer, R, binningBuffer, imageBuffer, stages, false);
}

namespace {
// `full`: the reference's return values exactly (src/Rasterizer.cu:253-293) — dL_dcov3D [P,6] is computed on the scales + rotations path too
// (there it is the intermediate computeCov2DCUDA hands to computeCov3D's backward, backward.cu:144-274 -> :278-341). false: the lean form the
// autograd nodes use — with scales + rotations nothing consumes dL_dcov3D (the node hands it to the absent cov3Ds_precomp input), so it is
// neither allocated nor stored: [0,6].
std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor,
           torch::Tensor>
backward_impl(const torch::Tensor& background, const torch::Tensor& means3D,
              const torch::Tensor& radii, const torch::Tensor& colors, const torch::Tensor& scales,
              const torch::Tensor& rotations, const float scale_modifier,
              const torch::Tensor& cov3D_precomp, const torch::Tensor& viewmatrix,
              const torch::Tensor& projmatrix, const float tan_fovx, const float tan_fovy,
              const torch::Tensor& dL_dout_color, const torch::Tensor& dL_dout_ds, const torch::Tensor& sh,
              const int degree, const torch::Tensor& campos, const torch::Tensor& geomBuffer, const int R,
              const torch::Tensor& binningBuffer, const torch::Tensor& imageBuffer, const int stages,
              const bool detach_depth_color, const bool full)
{
    const int P = (int)means3D.size(0);
    const int H = (int)dL_dout_color.size(1), W = (int)dL_dout_color.size(2);
    const torch::Device device = geomBuffer.device();
    c10::DeviceGuard guard(device);
    const auto fopt = torch::TensorOptions().device(device).dtype(torch::kFloat32);

    const torch::Tensor bg = contig_f32(background, device), m3 = contig_f32(means3D, device),
                        col = contig_f32(colors, device), sc = contig_f32(scales, device),
                        rot = contig_f32(rotations, device), cov = contig_f32(cov3D_precomp, device),
                        vm = contig_f32(viewmatrix, device), pm = contig_f32(projmatrix, device),
                        shc = contig_f32(sh, device), cp = contig_f32(campos, device),
                        gin = contig_f32(dL_dout_color, device), gds = contig_f32(dL_dout_ds, device);
    int M = 0;
    if (shc.numel() != 0) M = (int)shc.size(1);
    const bool has_sr = sc.numel() != 0 && rot.numel() != 0;

    // shapes of src/Rasterizer.cu:253-261; every element is written by the core, so no zero-fill,
    // except for the tensors the chosen parameterisation leaves untouched
    torch::Tensor dL_dmeans3D = torch::empty({P, 3}, fopt), dL_dmeans2D = torch::empty({P, 3}, fopt),
                  dL_dcolors = torch::empty({P, kChannels}, fopt), dL_dopacity = torch::empty({P, 1}, fopt),
                  dL_dcov3D = (has_sr && !full) ? torch::empty({0, 6}, fopt) : torch::empty({P, 6}, fopt), dL_dsh = torch::empty({P, M, 3}, fopt),
                  dL_dscales = has_sr ? torch::empty({P, 3}, fopt) : torch::zeros({P, 3}, fopt),
                  dL_drotations = has_sr ? torch::empty({P, 4}, fopt) : torch::zeros({P, 4}, fopt);
    if (P != 0) {
        gsr_backward_args a{};
        a.P = P; a.D = degree; a.M = M; a.R = R;
        a.background = fptr(bg); a.width = W; a.height = H;
        a.means3D = fptr(m3); a.shs = fptr(shc); a.colors_precomp = fptr(col);
        a.scales = fptr(sc); a.scale_modifier = scale_modifier; a.rotations = fptr(rot); a.cov3D_precomp = fptr(cov);
        a.viewmatrix = fptr(vm); a.projmatrix = fptr(pm); a.cam_pos = fptr(cp);
        a.tan_fovx = tan_fovx; a.tan_fovy = tan_fovy;
        a.radii = radii.numel() ? radii.data_ptr<int>() : nullptr;
        a.geom_buffer = reinterpret_cast<char*>(geomBuffer.data_ptr());
        a.binning_buffer = reinterpret_cast<char*>(binningBuffer.data_ptr());
        a.image_buffer = reinterpret_cast<char*>(imageBuffer.data_ptr());
        a.binning_bytes = 0;
        a.dL_dpix = fptr(gin);
        a.dL_dmean2D = dL_dmeans2D.data_ptr<float>(); a.dL_dconic = nullptr;
        a.dL_dopacity = dL_dopacity.data_ptr<float>(); a.dL_dcolor = dL_dcolors.data_ptr<float>();
        a.dL_dmean3D = dL_dmeans3D.data_ptr<float>(); a.dL_dcov3D = dL_dcov3D.numel() ? dL_dcov3D.data_ptr<float>() : nullptr;
        a.dL_dsh = M ? dL_dsh.data_ptr<float>() : nullptr;
        a.dL_dscale = has_sr ? dL_dscales.data_ptr<float>() : nullptr;
        a.dL_drot = has_sr ? dL_drotations.data_ptr<float>() : nullptr;
        a.profile_events = nullptr;
        a.band_y0 = a.band_y1 = 0;
        a.stages = stages;
        a.dL_dds = fptr(gds);
        a.ds_detach_depth = detach_depth_color ? 1 : 0;
        check(gsr_backward(&a, current_stream(device)), "RasterizeGaussiansBackwardCUDA");
    }
    return std::make_tuple(dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales,
                           dL_drotations);
}
} // namespace

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
                               const bool detach_depth_color)
{
    return backward_impl(background, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix, projmatrix,
                         tan_fovx, tan_fovy, dL_dout_color, dL_dout_ds, sh, degree, campos, geomBuffer, R, binningBuffer, imageBuffer,
                         stages, detach_depth_color, /*full=*/false);
}

// The reference's stateless entry point: any number of calls per forward (stages 0 = blend + per-splat +
// re-zero of the accumulators), and the reference's return values: dL_dcov3D is [P,6] and filled whichever
// parameterisation the caller uses (src/Rasterizer.cu:253-261,265-293).
std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor,
           torch::Tensor>
RasterizeGaussiansBackwardCUDA(const torch::Tensor& background, const torch::Tensor& means3D,
                               const torch::Tensor& radii, const torch::Tensor& colors, const torch::Tensor& scales,
                               const torch::Tensor& rotations, const float scale_modifier,
                               const torch::Tensor& cov3D_precomp, const torch::Tensor& viewmatrix,
                               const torch::Tensor& projmatrix, const float tan_fovx, const float tan_fovy,
                               const torch::Tensor& dL_dout_color, const torch::Tensor& sh, const int degree,
                               const torch::Tensor& campos, const torch::Tensor& geomBuffer, const int R,
                               const torch::Tensor& binningBuffer, const torch::Tensor& imageBuffer)
{
    return backward_impl(background, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix, projmatrix,
                         tan_fovx, tan_fovy, dL_dout_color, torch::Tensor(), sh, degree, campos, geomBuffer, R, binningBuffer, imageBuffer,
                         0, false, /*full=*/true);
}

torch::Tensor markVisible(torch::Tensor& means3D, torch::Tensor& viewmatrix, torch::Tensor& projmatrix)
{
    const int P = (int)means3D.size(0);
    const torch::Device device = means3D.is_cuda() ? means3D.device() : viewmatrix.device();
    c10::DeviceGuard guard(device);
    torch::Tensor present = torch::empty({P}, torch::TensorOptions().device(device).dtype(torch::kBool));
    if (P != 0) {
        const torch::Tensor m3 = contig_f32(means3D, device), vm = contig_f32(viewmatrix, device),
                            pm = contig_f32(projmatrix, device);
        check(gsr_mark_visible(P, fptr(m3), fptr(vm), fptr(pm), reinterpret_cast<uint8_t*>(present.data_ptr<bool>()),
                               current_stream(device)),
              "markVisible");
    }
    return present;
}

torch::Tensor RasterizeGaussiansfilterCUDA(const torch::Tensor& means3D, const torch::Tensor& scales,
                                           const torch::Tensor& rotations, const float scale_modifier,
                                           const torch::Tensor& viewmatrix, const torch::Tensor& projmatrix,
                                           const float tan_fovx, const float tan_fovy, const int image_height,
                                           const int image_width, const bool prefiltered, int device_num)
{
    if (means3D.ndimension() != 2 || means3D.size(1) != 3) {
        AT_ERROR("means3D must have dimensions (num_points, 3)"); // src/Rasterizer.cu:323-325
    }
    const int P = (int)means3D.size(0);
    const torch::Device device(torch::kCUDA, (c10::DeviceIndex)device_num);
    c10::DeviceGuard guard(device);
    torch::Tensor radii = torch::empty({P}, torch::TensorOptions().device(device).dtype(torch::kInt32));
    if (P != 0) {
        const torch::Tensor m3 = contig_f32(means3D, device), sc = contig_f32(scales, device),
                            rot = contig_f32(rotations, device), vm = contig_f32(viewmatrix, device),
                            pm = contig_f32(projmatrix, device);
        check(gsr_visible_filter(P, image_width, image_height, fptr(m3), fptr(sc), scale_modifier, fptr(rot),
                                 fptr(vm), fptr(pm), tan_fovx, tan_fovy, prefiltered ? 1 : 0, radii.data_ptr<int>(),
                                 current_stream(device)),
              "RasterizeGaussiansfilterCUDA");
    }
    return radii;
}

torch::Tensor filter_radii(torch::Tensor means3D, torch::Tensor scales, torch::Tensor rotations, int device_num,
                           GaussianRasterizationSettings s)
{
    return RasterizeGaussiansfilterCUDA(means3D, scales, rotations, s.scale_modifier, s.viewmatrix, s.projmatrix,
                                        s.tanfovx, s.tanfovy, s.image_height, s.image_width, s.prefiltered,
                                        device_num);
}

torch::autograd::tensor_list rasterize_gaussians(torch::Tensor means3D, torch::Tensor means2D, torch::Tensor sh,
                                                 torch::Tensor colors_precomp, torch::Tensor opacities,
                                                 torch::Tensor scales, torch::Tensor rotations,
                                                 torch::Tensor cov3Ds_precomp, int device_num,
                                                 GaussianRasterizationSettings s)
{
    const torch::Device device(torch::kCUDA, (c10::DeviceIndex)device_num);
    auto dev = [&](torch::Tensor t) { return t.defined() ? t.to(device) : t; }; // src/Rasterizer.cu:24-51
    return _RasterizeGaussians::apply(dev(means3D), dev(means2D), dev(sh), dev(colors_precomp), dev(opacities),
                                      dev(scales), dev(rotations), dev(cov3Ds_precomp), dev(s.bg),
                                      dev(s.viewmatrix), dev(s.projmatrix), dev(s.camera_center),
                                      (int64_t)s.image_height, (int64_t)s.image_width, (double)s.tanfovx,
                                      (double)s.tanfovy, (double)s.scale_modifier, (int64_t)s.sh_degree,
                                      s.prefiltered, (int64_t)device_num);
}

torch::autograd::tensor_list _RasterizeGaussians::forward(
    torch::autograd::AutogradContext* ctx, torch::Tensor means3D, torch::Tensor means2D, torch::Tensor sh,
    torch::Tensor colors_precomp, torch::Tensor opacities, torch::Tensor scales, torch::Tensor rotations,
    torch::Tensor cov3Ds_precomp, torch::Tensor bg, torch::Tensor viewmatrix, torch::Tensor projmatrix,
    torch::Tensor camera_center, int64_t image_height, int64_t image_width, double tanfovx, double tanfovy,
    double scale_modifier, int64_t sh_degree, bool prefiltered, int64_t device_num)
{
    (void)means2D; // only a gradient sink, exactly as in the reference
    camera_center = camera_center.contiguous();
    int num_rendered;
    torch::Tensor color, radii, geomBuffer, binningBuffer, imgBuffer, depth;
    std::tie(num_rendered, color, radii, geomBuffer, binningBuffer, imgBuffer, depth) = RasterizeGaussiansCUDA(
        bg, means3D, colors_precomp, opacities, scales, rotations, (float)scale_modifier, cov3Ds_precomp, viewmatrix,
        projmatrix, (float)tanfovx, (float)tanfovy, (int)image_height, (int)image_width, sh, (int)sh_degree,
        camera_center, prefiltered, (int)device_num);
    // include/Rasterizer.cuh:190-203
    ctx->save_for_backward({colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer,
                            binningBuffer, imgBuffer, bg, viewmatrix, projmatrix, camera_center});
    ctx->saved_data["num_rendered"] = num_rendered;
    ctx->saved_data["scale_modifier"] = scale_modifier;
    ctx->saved_data["tanfovx"] = tanfovx;
    ctx->saved_data["tanfovy"] = tanfovy;
    ctx->saved_data["sh_degree"] = sh_degree;
    ctx->saved_data["opacity_shape"] = opacities.sizes().vec();
    ctx->mark_non_differentiable({radii, depth});
    return {color, radii, depth};
}

torch::autograd::tensor_list _RasterizeGaussians::backward(torch::autograd::AutogradContext* ctx,
                                                           torch::autograd::tensor_list grad_outputs)
{
    auto grad_out_color = grad_outputs[0]; // radii / depth gradients are ignored (include/Rasterizer.cuh:210-211)
    const auto saved = ctx->get_saved_variables();
    const auto &colors_precomp = saved[0], &means3D = saved[1], &scales = saved[2], &rotations = saved[3],
               &cov3Ds_precomp = saved[4], &radii = saved[5], &sh = saved[6], &geomBuffer = saved[7],
               &binningBuffer = saved[8], &imgBuffer = saved[9], &bg = saved[10], &viewmatrix = saved[11],
               &projmatrix = saved[12], &camera_center = saved[13];
    torch::Tensor grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh,
        grad_scales, grad_rotations;
    // the graph node knows how often it ran: the first backward skips the re-zero of the accumulators, a
    // repeated one (retain_graph) starts with a clear
    const bool again = ctx->saved_data.count("backward_ran") != 0;
    ctx->saved_data["backward_ran"] = true;
    const int stages = again ? (GSR_STAGE_CLEAR | GSR_STAGE_BLEND | GSR_STAGE_SPLAT) : (GSR_STAGE_BLEND | GSR_STAGE_SPLAT);
    std::tie(grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh,
             grad_scales, grad_rotations) =
        RasterizeGaussiansBackwardStaged(bg, means3D, radii, colors_precomp, scales, rotations,
                                         (float)ctx->saved_data["scale_modifier"].toDouble(), cov3Ds_precomp,
                                         viewmatrix, projmatrix, (float)ctx->saved_data["tanfovx"].toDouble(),
                                         (float)ctx->saved_data["tanfovy"].toDouble(), grad_out_color, sh,
                                         (int)ctx->saved_data["sh_degree"].toInt(), camera_center, geomBuffer,
                                         (int)ctx->saved_data["num_rendered"].toInt(), binningBuffer, imgBuffer, stages);
    auto shaped = [](const torch::Tensor& g, const torch::Tensor& like) {
        return like.numel() == 0 ? torch::Tensor() : g.reshape(like.sizes());
    };
    return {grad_means3D,
            grad_means2D,
            shaped(grad_sh, sh),
            shaped(grad_colors_precomp, colors_precomp),
            grad_opacities.reshape(ctx->saved_data["opacity_shape"].toIntVector()),
            shaped(grad_scales, scales),
            shaped(grad_rotations, rotations),
            shaped(grad_cov3Ds_precomp, cov3Ds_precomp),
            torch::Tensor(), torch::Tensor(), torch::Tensor(), torch::Tensor(), torch::Tensor(), torch::Tensor(),
            torch::Tensor(), torch::Tensor(), torch::Tensor(), torch::Tensor(), torch::Tensor(), torch::Tensor()};
}

// ---- the fused pair (new capability): colours + [view depth, 1] in one pass -------------------------------------------
torch::autograd::tensor_list rasterize_gaussians_pair(torch::Tensor means3D, torch::Tensor means2D, torch::Tensor sh,
                                                      torch::Tensor colors_precomp, torch::Tensor opacities,
                                                      torch::Tensor scales, torch::Tensor rotations,
                                                      torch::Tensor cov3Ds_precomp, int device_num,
                                                      GaussianRasterizationSettings s, bool detach_depth_color)
{
    const torch::Device device(torch::kCUDA, (c10::DeviceIndex)device_num);
    auto dev = [&](torch::Tensor t) { return t.defined() ? t.to(device) : t; };
    return _RasterizeGaussiansPair::apply(dev(means3D), dev(means2D), dev(sh), dev(colors_precomp), dev(opacities),
                                          dev(scales), dev(rotations), dev(cov3Ds_precomp), dev(s.bg),
                                          dev(s.viewmatrix), dev(s.projmatrix), dev(s.camera_center),
                                          (int64_t)s.image_height, (int64_t)s.image_width, (double)s.tanfovx,
                                          (double)s.tanfovy, (double)s.scale_modifier, (int64_t)s.sh_degree,
                                          s.prefiltered, (int64_t)device_num, detach_depth_color);
}

torch::autograd::tensor_list _RasterizeGaussiansPair::forward(
    torch::autograd::AutogradContext* ctx, torch::Tensor means3D, torch::Tensor means2D, torch::Tensor sh,
    torch::Tensor colors_precomp, torch::Tensor opacities, torch::Tensor scales, torch::Tensor rotations,
    torch::Tensor cov3Ds_precomp, torch::Tensor bg, torch::Tensor viewmatrix, torch::Tensor projmatrix,
    torch::Tensor camera_center, int64_t image_height, int64_t image_width, double tanfovx, double tanfovy,
    double scale_modifier, int64_t sh_degree, bool prefiltered, int64_t device_num, bool detach_depth_color)
{
    (void)means2D;
    camera_center = camera_center.contiguous();
    ctx->saved_data["detach_depth_color"] = detach_depth_color;
    int num_rendered;
    torch::Tensor color, radii, geomBuffer, binningBuffer, imgBuffer, depth, ds;
    std::tie(num_rendered, color, radii, geomBuffer, binningBuffer, imgBuffer, depth, ds) = RasterizeGaussiansPairCUDA(
        bg, means3D, colors_precomp, opacities, scales, rotations, (float)scale_modifier, cov3Ds_precomp, viewmatrix,
        projmatrix, (float)tanfovx, (float)tanfovy, (int)image_height, (int)image_width, sh, (int)sh_degree,
        camera_center, prefiltered, (int)device_num);
    ctx->save_for_backward({colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer,
                            binningBuffer, imgBuffer, bg, viewmatrix, projmatrix, camera_center});
    ctx->saved_data["num_rendered"] = num_rendered;
    ctx->saved_data["scale_modifier"] = scale_modifier;
    ctx->saved_data["tanfovx"] = tanfovx;
    ctx->saved_data["tanfovy"] = tanfovy;
    ctx->saved_data["sh_degree"] = sh_degree;
    ctx->saved_data["opacity_shape"] = opacities.sizes().vec();
    ctx->saved_data["H"] = image_height;
    ctx->saved_data["W"] = image_width;
    ctx->mark_non_differentiable({radii, depth});
    return {color, ds, radii, depth};
}

torch::autograd::tensor_list _RasterizeGaussiansPair::backward(torch::autograd::AutogradContext* ctx,
                                                               torch::autograd::tensor_list grad_outputs)
{
    const auto saved = ctx->get_saved_variables();
    const auto &colors_precomp = saved[0], &means3D = saved[1], &scales = saved[2], &rotations = saved[3],
               &cov3Ds_precomp = saved[4], &radii = saved[5], &sh = saved[6], &geomBuffer = saved[7],
               &binningBuffer = saved[8], &imgBuffer = saved[9], &bg = saved[10], &viewmatrix = saved[11],
               &projmatrix = saved[12], &camera_center = saved[13];
    // an output the loss does not use arrives undefined: its gradient is zero
    const torch::Tensor gcol = grad_outputs[0].defined() ? grad_outputs[0] : torch::zeros({3, ctx->saved_data["H"].toInt(), ctx->saved_data["W"].toInt()}, geomBuffer.options().dtype(torch::kFloat32));
    const torch::Tensor gds = grad_outputs[1].defined() ? grad_outputs[1] : torch::zeros({2, gcol.size(1), gcol.size(2)}, gcol.options());
    torch::Tensor grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh,
        grad_scales, grad_rotations;
    const bool again = ctx->saved_data.count("backward_ran") != 0;
    ctx->saved_data["backward_ran"] = true;
    const int stages = again ? (GSR_STAGE_CLEAR | GSR_STAGE_BLEND | GSR_STAGE_SPLAT) : (GSR_STAGE_BLEND | GSR_STAGE_SPLAT);
    std::tie(grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh,
             grad_scales, grad_rotations) =
        RasterizeGaussiansPairBackward(bg, means3D, radii, colors_precomp, scales, rotations,
                                       (float)ctx->saved_data["scale_modifier"].toDouble(), cov3Ds_precomp,
                                       viewmatrix, projmatrix, (float)ctx->saved_data["tanfovx"].toDouble(),
                                       (float)ctx->saved_data["tanfovy"].toDouble(), gcol, gds, sh,
                                       (int)ctx->saved_data["sh_degree"].toInt(), camera_center, geomBuffer,
                                       (int)ctx->saved_data["num_rendered"].toInt(), binningBuffer, imgBuffer, stages,
                                       ctx->saved_data["detach_depth_color"].toBool());
    auto shaped = [](const torch::Tensor& g, const torch::Tensor& like) {
        return like.numel() == 0 ? torch::Tensor() : g.reshape(like.sizes());
    };
    return {grad_means3D,
            grad_means2D,
            shaped(grad_sh, sh),
            shaped(grad_colors_precomp, colors_precomp),
            grad_opacities.reshape(ctx->saved_data["opacity_shape"].toIntVector()),
            shaped(grad_scales, scales),
            shaped(grad_rotations, rotations),
            shaped(grad_cov3Ds_precomp, cov3Ds_precomp),
            torch::Tensor(), torch::Tensor(), torch::Tensor(), torch::Tensor(), torch::Tensor(), torch::Tensor(),
            torch::Tensor(), torch::Tensor(), torch::Tensor(), torch::Tensor(), torch::Tensor(), torch::Tensor(), torch::Tensor()};
}

} // namespace ORB_SLAM2

// reference src/spatial.cu:15-27
torch::Tensor distCUDA2(const torch::Tensor& points, torch::Device device)
{
    const int P = (int)points.size(0);
    c10::DeviceGuard guard(device);
    const torch::Tensor pts = points.to(device, torch::kFloat32).contiguous();
    torch::Tensor means = torch::empty({P}, torch::TensorOptions().device(device).dtype(torch::kFloat32));
    if (P != 0) {
        torch::Tensor ws = torch::empty({(int64_t)gsr_knn_bytes(P)}, torch::TensorOptions().device(device).dtype(torch::kByte));
        const int rc = gsr_dist2(P, pts.data_ptr<float>(), means.data_ptr<float>(), reinterpret_cast<char*>(ws.data_ptr()),
                                 (size_t)ws.numel(), (void*)c10::hip::getCurrentHIPStream(device.index()).stream());
        if (rc < 0) throw std::runtime_error(std::string("distCUDA2: ") + gsr_error_string(rc));
    }
    return means;
}
