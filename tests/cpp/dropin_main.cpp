// Test infrastructure: ONE caller, written against the reference's C++ API (include/Rasterizer.cuh:76-382, include/spatial.h) exactly as src/Render.cc uses
// it — GaussianRasterizationSettings, GaussianRasterizer::forward / mark_visible / Visable, autograd through the render, distCUDA2 — and compiled TWICE:
//   tests/cpp/dropin_hip.bin   against this repository's host layer   (-DDROPIN_HEADER="Rasterizer.h",   links torch_ext/libgsr_torch.so; tests/cpp/build.py)
//   oracle/_ref/dropin_ref.bin against the reference's own host layer (-DDROPIN_HEADER="Rasterizer.cuh", the reference's src/Rasterizer.cu, spatial.cu and
//                              rasterizer translated by hipify-perl at build time; oracle/build_ref.sh)
// Same source, same input file, two output files: tests/test_gpu_reference_build.py compares them. That the file compiles and links against either side
// without an #ifdef IS the drop-in claim at this boundary.
//
//   dropin_*.bin <scene file> <output file>
// scene file (little endian): int32 P, M, W, H, D; float32 tanfovx, tanfovy, scale_modifier; then float32 arrays bg[3], view[16], proj[16], campos[3], means3D[P,3],
// opacities[P,1], scales[P,3], rotations[P,4], colors[P,3] (M == 0) or shs[P,M,3] (M > 0), G[3,H,W] (the upstream gradient of the colour image),
// points[P,3] (for distCUDA2)
// output file: named float32 / int32 blocks — "name" int64 count, data.
#include DROPIN_HEADER
#include <cstdint>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

#include <torch/torch.h>

// (the reference declares distCUDA2 in include/spatial.h; this repository's Rasterizer.h declares it too — the same signature, so a second declaration is harmless)
torch::Tensor distCUDA2(const torch::Tensor& points, torch::Device device);

namespace {

std::vector<float> read_f(std::ifstream& f, size_t n)
{
    std::vector<float> v(n);
    f.read(reinterpret_cast<char*>(v.data()), (std::streamsize)(n * sizeof(float)));
    if (!f) throw std::runtime_error("scene file too short");
    return v;
}

torch::Tensor dev(std::ifstream& f, std::vector<int64_t> shape, const torch::Device& d)
{
    size_t n = 1;
    for (auto s : shape) n *= (size_t)s;
    auto v = read_f(f, n);
    return torch::from_blob(v.data(), shape, torch::kFloat32).clone().to(d);
}

void put(std::ofstream& o, const std::string& name, const torch::Tensor& t)
{
    const auto c = t.detach().to(torch::kCPU).contiguous();
    const auto as = c.scalar_type() == torch::kFloat32 ? c : c.to(torch::kInt32);
    const int32_t len = (int32_t)name.size(), kind = as.scalar_type() == torch::kFloat32 ? 0 : 1;
    const int64_t n = as.numel();
    o.write(reinterpret_cast<const char*>(&len), 4); o.write(name.data(), len); o.write(reinterpret_cast<const char*>(&kind), 4);
    o.write(reinterpret_cast<const char*>(&n), 8); o.write(reinterpret_cast<const char*>(as.data_ptr()), n * 4);
}

} // namespace

int main(int argc, char** argv)
{
    if (argc < 3) { fprintf(stderr, "usage: %s <scene> <out>\n", argv[0]); return 2; }
    std::ifstream f(argv[1], std::ios::binary);
    if (!f) { fprintf(stderr, "cannot open %s\n", argv[1]); return 2; }
    int32_t h[5];
    f.read(reinterpret_cast<char*>(h), sizeof(h));
    const int P = h[0], M = h[1], W = h[2], H = h[3], D = h[4];
    const auto hf = read_f(f, 3);
    const torch::Device d(torch::kCUDA, 0);
    ORB_SLAM2::GaussianRasterizationSettings s;
    s.image_height = H; s.image_width = W; s.tanfovx = hf[0]; s.tanfovy = hf[1]; s.scale_modifier = hf[2];
    s.bg = dev(f, {3}, d); s.viewmatrix = dev(f, {4, 4}, d); s.projmatrix = dev(f, {4, 4}, d); s.sh_degree = D; s.camera_center = dev(f, {3}, d); s.prefiltered = false;
    auto means3D = dev(f, {P, 3}, d).requires_grad_(true);
    auto opac = dev(f, {P, 1}, d).requires_grad_(true);
    auto scales = dev(f, {P, 3}, d).requires_grad_(true);
    auto rots = dev(f, {P, 4}, d).requires_grad_(true);
    torch::Tensor colors, shs;
    if (M == 0) colors = dev(f, {P, 3}, d).requires_grad_(true);
    else shs = dev(f, {P, M, 3}, d).requires_grad_(true);
    const auto G = dev(f, {3, H, W}, d);
    const auto points = dev(f, {P, 3}, d);
    auto means2D = torch::zeros({P, 3}, torch::TensorOptions().device(d).dtype(torch::kFloat32)).requires_grad_(true); // (Render.cc: screenspace points, retain_grad)

    ORB_SLAM2::GaussianRasterizer r(s);
    // Render.cc:963-975's call shape: forward(means3D, means2D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp, device_num)
    auto out = r.forward(means3D, means2D, opac, shs, colors, scales, rots, torch::Tensor(), 0);
    auto color = std::get<0>(out), radii = std::get<1>(out), depth = std::get<2>(out);
    (color * G).sum().backward();
    std::ofstream o(argv[2], std::ios::binary);
    put(o, "color", color); put(o, "radii", radii); put(o, "depth", depth);
    put(o, "d_means3D", means3D.grad()); put(o, "d_opacities", opac.grad()); put(o, "d_scales", scales.grad()); put(o, "d_rotations", rots.grad());
    put(o, "d_means2D", means2D.grad());
    if (M == 0) put(o, "d_colors", colors.grad());
    else put(o, "d_shs", shs.grad());
    put(o, "visible", r.mark_visible(means3D.detach()).to(torch::kInt32));
    put(o, "filter_radii", std::get<0>(r.Visable(means3D.detach(), opac.detach(), scales.detach(), rots.detach(), 0)));
    put(o, "dist2", distCUDA2(points, d));
    // the argument checks of GaussianRasterizer::forward (Rasterizer.cuh:310-317): both / neither colour source, both / neither covariance source
    int thrown = 0;
    try { r.forward(means3D, means2D, opac, torch::Tensor(), torch::Tensor(), scales, rots, torch::Tensor(), 0); } catch (const std::invalid_argument&) { thrown |= 1; }
    try { r.forward(means3D, means2D, opac, shs.defined() ? shs : torch::zeros({P, 1, 3}, d), colors.defined() ? colors : torch::zeros({P, 3}, d), scales, rots, torch::Tensor(), 0); }
    catch (const std::invalid_argument&) { thrown |= 2; }
    try { r.forward(means3D, means2D, opac, shs, colors, torch::Tensor(), torch::Tensor(), torch::Tensor(), 0); } catch (const std::invalid_argument&) { thrown |= 4; }
    put(o, "invalid_argument_checks", torch::tensor({thrown}, torch::kInt32));
    printf("ok P=%d M=%d %dx%d\n", P, M, W, H);
    return 0;
}
