// render_api_test.cpp — exercises the C++ drop-in API exactly the way the reference's
// Render::StartSplatting does (src/Render.cc:711-781, default path useRadiusFilter=false):
// means are moved into the camera frame with a bmm so that the pose receives its gradient
// from autograd, activations are applied to the raw parameters, and
// GaussianRasterizer::forward runs with viewmatrix = I. Reads a scene written by
// tests/test_gpu_cpp_api.py, writes image/depth/radii and the gradients of a linear loss.
#include <torch/torch.h>

#include <cstdio>
#include <fstream>
#include <iostream>
#include <vector>

#include "../../gsorb-slam_amd/torch_ext/Rasterizer.h"

using namespace ORB_SLAM2;

static torch::Tensor read_f32(std::ifstream& f, std::vector<int64_t> shape)
{
    int64_t n = 1;
    for (auto s : shape) n *= s;
    std::vector<float> buf(n);
    f.read(reinterpret_cast<char*>(buf.data()), n * 4);
    return torch::from_blob(buf.data(), shape, torch::kFloat32).clone();
}
static void write_t(std::ofstream& f, torch::Tensor t)
{
    t = t.detach().to(torch::kCPU).contiguous();
    f.write(reinterpret_cast<const char*>(t.data_ptr()), t.numel() * t.element_size());
}

int main(int argc, char** argv)
{
    if (argc < 3) { std::fprintf(stderr, "usage: %s scene.bin out.bin\n", argv[0]); return 2; }
    std::ifstream in(argv[1], std::ios::binary);
    int32_t hdr[4];
    in.read(reinterpret_cast<char*>(hdr), sizeof(hdr));
    const int64_t P = hdr[0], W = hdr[1], H = hdr[2];
    const bool direct = hdr[3] == 1; // 1: the file holds camera-frame means and ACTIVATED parameters, fed to the operator as they are
    float fl[4];
    in.read(reinterpret_cast<char*>(fl), sizeof(fl)); // tanfovx, tanfovy, unused, unused
    const auto dev = torch::Device(torch::kCUDA, 0);
    auto xyz = read_f32(in, {P, 3}).to(dev).set_requires_grad(true);          // world-frame means
    auto rgb = read_f32(in, {P, 3}).to(dev).set_requires_grad(true);
    auto unnorm_quat = read_f32(in, {P, 4}).to(dev).set_requires_grad(true);
    auto logit_opac = read_f32(in, {P, 1}).to(dev).set_requires_grad(true);
    auto log_scales = read_f32(in, {P, 3}).to(dev).set_requires_grad(true);
    auto Tcw = read_f32(in, {4, 4}).to(dev).set_requires_grad(true);
    auto proj = read_f32(in, {4, 4}).to(dev);                                 // P^T (row-major tensor)
    auto G = read_f32(in, {3, H, W}).to(dev);

    GaussianRasterizationSettings s{(int)H, (int)W, fl[0], fl[1], torch::zeros({3}, dev), 1.0f,
                                    torch::eye(4, dev), proj, 0, torch::zeros({3}, dev), false};
    GaussianRasterizer rasterizer(s);

    torch::Tensor mean3D, opacities, norm_qua, scales;
    if (direct) { // the boundary alone: no torch arithmetic between the file and the operator
        mean3D = xyz; opacities = logit_opac; norm_qua = unnorm_quat; scales = log_scales;
        Tcw.mutable_grad() = torch::zeros_like(Tcw);
    } else {
        // src/Render.cc:750-752
        auto Tb = Tcw.unsqueeze(0).repeat({P, 1, 1});
        auto m4 = torch::cat({xyz, torch::ones({P, 1}, dev)}, 1).unsqueeze(-1);
        mean3D = Tb.bmm(m4).squeeze(-1).index({torch::indexing::Slice(), torch::indexing::Slice(0, 3)});
        // :756-760
        opacities = torch::sigmoid(logit_opac);
        norm_qua = torch::nn::functional::normalize(unnorm_quat);
        scales = torch::exp(log_scales);
    }
    auto mean2D = torch::zeros_like(mean3D).set_requires_grad(true);
    mean2D.retain_grad();

    bool threw = false;
    try { rasterizer.forward(mean3D, mean2D, opacities); } catch (const std::invalid_argument&) { threw = true; }

    auto [image, radii, depth] = rasterizer.forward(mean3D, mean2D, opacities, torch::Tensor(), rgb, scales, norm_qua,
                                                    torch::Tensor(), 0);
    torch::cuda::synchronize();
    auto loss = (image * G).sum();
    loss.backward();
    auto vis = rasterizer.mark_visible(mean3D.detach());
    auto fr = std::get<0>(rasterizer.Visable(mean3D.detach(), opacities.detach(), scales.detach(), norm_qua.detach(), 0));

    std::ofstream out(argv[2], std::ios::binary);
    int32_t flags[4] = {threw ? 1 : 0, (int32_t)radii.scalar_type() == (int32_t)torch::kInt32 ? 1 : 0,
                        depth.requires_grad() ? 1 : 0, (int32_t)torch::equal(fr, radii)};
    out.write(reinterpret_cast<const char*>(flags), sizeof(flags));
    write_t(out, image); write_t(out, depth); write_t(out, radii.to(torch::kInt32));
    write_t(out, xyz.grad()); write_t(out, rgb.grad()); write_t(out, unnorm_quat.grad());
    write_t(out, logit_opac.grad()); write_t(out, log_scales.grad()); write_t(out, Tcw.grad());
    write_t(out, mean2D.grad()); write_t(out, vis.to(torch::kInt32));
    if (direct) {
        // the reference-named FREE functions (include/Rasterizer.cuh:24-71, src/Rasterizer.cu:136-305), called the way _RasterizeGaussians calls
        // them: the backward returns the reference's eight tensors, dL_dcov3D [P,6] filled on the scales + rotations path too
        torch::NoGradGuard ng;
        const torch::Tensor none;
        auto fw = RasterizeGaussiansCUDA(s.bg, mean3D.detach(), rgb.detach(), opacities.detach(), scales.detach(), norm_qua.detach(), 1.0f, none,
                                         s.viewmatrix, s.projmatrix, s.tanfovx, s.tanfovy, (int)H, (int)W, none, 0, s.camera_center, false, 0);
        auto bw = RasterizeGaussiansBackwardCUDA(s.bg, mean3D.detach(), std::get<2>(fw), rgb.detach(), scales.detach(), norm_qua.detach(), 1.0f, none,
                                                 s.viewmatrix, s.projmatrix, s.tanfovx, s.tanfovy, G, none, 0, s.camera_center, std::get<3>(fw),
                                                 std::get<0>(fw), std::get<4>(fw), std::get<5>(fw));
        const auto& dcov = std::get<4>(bw);
        int32_t shape[2] = {(int32_t)dcov.size(0), (int32_t)dcov.size(1)};
        out.write(reinterpret_cast<const char*>(shape), sizeof(shape));
        write_t(out, dcov);
        write_t(out, std::get<6>(bw)); // dL_dscales of the same call
    }
    std::printf("ok P=%ld loss=%f\n", (long)P, loss.item<float>());
    return 0;
}
