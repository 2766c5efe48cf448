// TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg never ship or time this as the product).
//
// A C entry to the REFERENCE's own rasterizer — CudaRasterizer::Rasterizer::forward / backward
// (Thirdparty/diff_gaussian_rasterization/cuda_rasterizer/rasterizer.h:30-85, rasterizer_impl.cu:198-345, :405-498) — built by
// oracle/build_ref.sh: the reference's .cu / .h files are translated where they lie by ROCm's own hipify-perl into a scratch
// directory, compiled by hipcc for gfx950 together with this file into oracle/_ref/libgsr_ref*.so, and the scratch directory is
// removed. Nothing of the reference is written by hand: no header, library or tool is stood in for (hipcub, HIP cooperative groups
// and glm — vendored in the reference's third_party/ — are what the translated sources include). This file only moves host
// arrays to the device, calls the reference's two entry points with the reference's own buffer-allocation callbacks, and copies
// back the outputs and the reference's state arrays (GeometryState / BinningState / ImageState, rasterizer_impl.h:29-64).
//
// Same calling convention as oracle/gsr_oracle.h (gsro_scene, host pointers), so tests/ drive both through one Python class.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <stdexcept>
#include <vector>

#include "rasterizer.h"
#include "rasterizer_impl.h"
#include "simple_knn.h"

namespace {

struct Scene { // = gsro_scene (oracle/gsr_oracle.h)
    int P, D, M, W, H;
    const float* background;
    const float* means3D;
    const float* shs;
    const float* colors_precomp;
    const float* opacities;
    const float* scales;
    float scale_modifier;
    const float* rotations;
    const float* cov3D_precomp;
    const float* viewmatrix;
    const float* projmatrix;
    const float* cam_pos;
    float tan_fovx, tan_fovy;
};

void chk(hipError_t e, const char* what)
{
    if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}

struct DevBuf {
    char* p = nullptr;
    size_t n = 0;
    char* get(size_t bytes)
    {
        if (bytes > n) {
            if (p) (void)hipFree(p);
            chk(hipMalloc(&p, bytes), "hipMalloc");
            n = bytes;
        }
        return p;
    }
    ~DevBuf() { if (p) (void)hipFree(p); }
};

template <typename T>
struct DevArr {
    T* p = nullptr;
    size_t n = 0;
    void upload(const T* host, size_t count)
    {
        release();
        if (!host || count == 0) return;
        chk(hipMalloc(&p, count * sizeof(T)), "hipMalloc");
        chk(hipMemcpy(p, host, count * sizeof(T), hipMemcpyHostToDevice), "hipMemcpy H2D");
        n = count;
    }
    void zeros(size_t count)
    {
        release();
        if (count == 0) count = 1;
        chk(hipMalloc(&p, count * sizeof(T)), "hipMalloc");
        chk(hipMemset(p, 0, count * sizeof(T)), "hipMemset");
        n = count;
    }
    void download(T* host, size_t count) const
    {
        if (host && count) chk(hipMemcpy(host, p, count * sizeof(T), hipMemcpyDeviceToHost), "hipMemcpy D2H");
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
    ~DevArr() { release(); }
};

struct State {
    DevBuf geom, binning, image;
    DevArr<float> bg, means3D, shs, colors, opac, scales, rots, cov, view, proj, campos, out_color, out_depth;
    DevArr<int> radii;
    int P = 0, M = 0, W = 0, H = 0, R = 0;
    std::vector<std::vector<char>> stages; // host copies, by the oracle's stage index (oracle/oracle.py: _STAGES)
    std::vector<size_t> counts;
};

template <typename T>
void keep(State& st, int idx, const T* dev, size_t count)
{
    st.stages[idx].resize(count * sizeof(T));
    st.counts[idx] = count;
    if (count) chk(hipMemcpy(st.stages[idx].data(), dev, count * sizeof(T), hipMemcpyDeviceToHost), "hipMemcpy D2H (stage)");
}

void upload_scene(State& st, const Scene& s)
{
    const size_t P = (size_t)s.P;
    st.bg.upload(s.background, 3);
    st.means3D.upload(s.means3D, 3 * P);
    st.shs.upload(s.shs, (size_t)3 * s.M * P);
    st.colors.upload(s.colors_precomp, 3 * P);
    st.opac.upload(s.opacities, P);
    st.scales.upload(s.scales, 3 * P);
    st.rots.upload(s.rotations, 4 * P);
    st.cov.upload(s.cov3D_precomp, 6 * P);
    st.view.upload(s.viewmatrix, 16);
    st.proj.upload(s.projmatrix, 16);
    st.campos.upload(s.cam_pos, 3);
}

} // namespace

extern "C" {

void* gsref_state_new() { return new State(); }
void gsref_state_free(void* p) { delete static_cast<State*>(p); }

// returns num_rendered (< 0: an error, printed); out_color [3,H,W], out_depth [H,W], radii [P] — HOST arrays
int gsref_forward(void* sp, const Scene* s, float* out_color, float* out_depth, int* radii)
{
    State& st = *static_cast<State*>(sp);
    try {
        st.P = s->P; st.M = s->M; st.W = s->W; st.H = s->H;
        const size_t P = (size_t)s->P, N = (size_t)s->W * s->H;
        upload_scene(st, *s);
        st.out_color.zeros(3 * N); st.out_depth.zeros(N); st.radii.zeros(P); // (rasterize_points.cu:67-69: torch::full(..., 0))
        st.stages.assign(15, {}); st.counts.assign(15, 0);
        if (P == 0) { st.R = 0; return 0; } // (rasterize_points.cu:84: the reference calls forward only if P != 0)
        std::function<char*(size_t)> ga = [&](size_t n) { return st.geom.get(n); };
        std::function<char*(size_t)> ba = [&](size_t n) { return st.binning.get(n); };
        std::function<char*(size_t)> ia = [&](size_t n) { return st.image.get(n); };
        st.R = CudaRasterizer::Rasterizer::forward(ga, ba, ia, s->P, s->D, s->M, st.bg.p, s->W, s->H, st.means3D.p, st.shs.p, st.colors.p, st.opac.p, st.scales.p,
                                                   s->scale_modifier, st.rots.p, st.cov.p, st.view.p, st.proj.p, st.campos.p, s->tan_fovx, s->tan_fovy, false,
                                                   st.out_color.p, st.out_depth.p, st.radii.p);
        chk(hipDeviceSynchronize(), "reference forward");
        st.out_color.download(out_color, 3 * N); st.out_depth.download(out_depth, N); st.radii.download(radii, P);
        // the reference's own state arrays, located by the reference's own fromChunk
        char* g = st.geom.p;
        const auto gs = CudaRasterizer::GeometryState::fromChunk(g, P);
        keep(st, 0, reinterpret_cast<const float*>(gs.means2D), 2 * P);
        keep(st, 1, gs.depths, P);
        keep(st, 2, gs.cov3D, 6 * P);
        keep(st, 3, reinterpret_cast<const float*>(gs.conic_opacity), 4 * P);
        keep(st, 4, gs.rgb, 3 * P);
        keep(st, 5, reinterpret_cast<const uint8_t*>(gs.clamped), 3 * P);
        keep(st, 6, gs.tiles_touched, P);
        keep(st, 7, gs.point_offsets, P);
        const size_t R = (size_t)st.R;
        if (R > 0) {
            char* b = st.binning.p;
            const auto bs = CudaRasterizer::BinningState::fromChunk(b, R);
            keep(st, 8, bs.point_list_keys_unsorted, R);
            keep(st, 9, bs.point_list_unsorted, R);
            keep(st, 10, bs.point_list_keys, R);
            keep(st, 11, bs.point_list, R);
        }
        char* im = st.image.p;
        const auto is = CudaRasterizer::ImageState::fromChunk(im, N);
        const size_t T = (size_t)((s->W + 15) / 16) * ((s->H + 15) / 16);
        keep(st, 12, reinterpret_cast<const uint32_t*>(is.ranges), 2 * T);
        keep(st, 13, is.accum_alpha, N);
        keep(st, 14, is.n_contrib, N);
        return st.R;
    } catch (const std::exception& e) {
        fprintf(stderr, "[gsref] forward: %s\n", e.what());
        return -1;
    }
}

const void* gsref_stage(void* sp, int idx, size_t* n)
{
    State& st = *static_cast<State*>(sp);
    if (idx < 0 || idx >= (int)st.stages.size()) { *n = 0; return nullptr; }
    *n = st.counts[idx];
    return st.stages[idx].empty() ? nullptr : st.stages[idx].data();
}

// rasterizer_impl.cu:405-498 on the state of the matching gsref_forward; the nine outputs are HOST arrays (zero-initialised on the device first,
// rasterize_points.cu:150-158)
int gsref_backward(void* sp, const Scene* s, const float* dL_dpix, float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D,
                   float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot)
{
    State& st = *static_cast<State*>(sp);
    try {
        const size_t P = (size_t)st.P, N = (size_t)st.W * st.H, M = (size_t)st.M;
        if (P == 0) return 0;
        DevArr<float> g, m2d, con, op, col, m3d, c3d, sh, sc, rt;
        g.upload(dL_dpix, 3 * N);
        m2d.zeros(3 * P); con.zeros(4 * P); op.zeros(P); col.zeros(3 * P); m3d.zeros(3 * P); c3d.zeros(6 * P); sh.zeros(3 * M * P); sc.zeros(3 * P); rt.zeros(4 * P);
        CudaRasterizer::Rasterizer::backward(s->P, s->D, s->M, st.R, st.bg.p, s->W, s->H, st.means3D.p, st.shs.p, st.colors.p, st.scales.p, s->scale_modifier, st.rots.p,
                                             st.cov.p, st.view.p, st.proj.p, st.campos.p, s->tan_fovx, s->tan_fovy, st.radii.p, st.geom.p, st.binning.p, st.image.p,
                                             g.p, m2d.p, con.p, op.p, col.p, m3d.p, c3d.p, sh.p, sc.p, rt.p);
        chk(hipDeviceSynchronize(), "reference backward");
        m2d.download(dL_dmean2D, 3 * P); con.download(dL_dconic, 4 * P); op.download(dL_dopacity, P); col.download(dL_dcolor, 3 * P); m3d.download(dL_dmean3D, 3 * P);
        c3d.download(dL_dcov3D, 6 * P); sh.download(dL_dsh, 3 * M * P); sc.download(dL_dscale, 3 * P); rt.download(dL_drot, 4 * P);
        return 0;
    } catch (const std::exception& e) {
        fprintf(stderr, "[gsref] backward: %s\n", e.what());
        return -1;
    }
}

// Wall time of the reference's own forward + backward on this GPU, inputs resident (bench.py's baseline leg): the calls of gsref_forward / gsref_backward without
// the host copies, `iters` times between HIP events after one untimed pass; ms[0] = forward, ms[1] = backward, per call. The forward includes what the reference's
// forward includes: its blocking read of num_rendered (rasterizer_impl.cu:283) and the buffer callbacks (no reallocation after the first pass here).
int gsref_time(void* sp, const Scene* s, const float* dL_dpix, int iters, float* ms)
{
    State& st = *static_cast<State*>(sp);
    try {
        const size_t P = (size_t)s->P, N = (size_t)s->W * s->H, M = (size_t)s->M;
        if (P == 0 || iters <= 0) return -1;
        st.P = s->P; st.M = s->M; st.W = s->W; st.H = s->H;
        upload_scene(st, *s);
        st.out_color.zeros(3 * N); st.out_depth.zeros(N); st.radii.zeros(P);
        DevArr<float> g, m2d, con, op, col, m3d, c3d, sh, sc, rt;
        g.upload(dL_dpix, 3 * N);
        m2d.zeros(3 * P); con.zeros(4 * P); op.zeros(P); col.zeros(3 * P); m3d.zeros(3 * P); c3d.zeros(6 * P); sh.zeros(3 * M * P); sc.zeros(3 * P); rt.zeros(4 * P);
        std::function<char*(size_t)> ga = [&](size_t n) { return st.geom.get(n); };
        std::function<char*(size_t)> ba = [&](size_t n) { return st.binning.get(n); };
        std::function<char*(size_t)> ia = [&](size_t n) { return st.image.get(n); };
        auto fwd = [&]() {
            st.R = CudaRasterizer::Rasterizer::forward(ga, ba, ia, s->P, s->D, s->M, st.bg.p, s->W, s->H, st.means3D.p, st.shs.p, st.colors.p, st.opac.p, st.scales.p,
                                                       s->scale_modifier, st.rots.p, st.cov.p, st.view.p, st.proj.p, st.campos.p, s->tan_fovx, s->tan_fovy, false,
                                                       st.out_color.p, st.out_depth.p, st.radii.p);
        };
        auto bwd = [&]() {
            // (the binding zeroes the nine gradient tensors before every backward, rasterize_points.cu:150-158: part of the reference's step)
            (void)hipMemsetAsync(m2d.p, 0, 3 * P * 4, 0); (void)hipMemsetAsync(con.p, 0, 4 * P * 4, 0); (void)hipMemsetAsync(op.p, 0, P * 4, 0); (void)hipMemsetAsync(col.p, 0, 3 * P * 4, 0);
            (void)hipMemsetAsync(m3d.p, 0, 3 * P * 4, 0); (void)hipMemsetAsync(c3d.p, 0, 6 * P * 4, 0); (void)hipMemsetAsync(sc.p, 0, 3 * P * 4, 0); (void)hipMemsetAsync(rt.p, 0, 4 * P * 4, 0);
            CudaRasterizer::Rasterizer::backward(s->P, s->D, s->M, st.R, st.bg.p, s->W, s->H, st.means3D.p, st.shs.p, st.colors.p, st.scales.p, s->scale_modifier, st.rots.p,
                                                 st.cov.p, st.view.p, st.proj.p, st.campos.p, s->tan_fovx, s->tan_fovy, st.radii.p, st.geom.p, st.binning.p, st.image.p,
                                                 g.p, m2d.p, con.p, op.p, col.p, m3d.p, c3d.p, sh.p, sc.p, rt.p);
        };
        fwd(); bwd();
        chk(hipDeviceSynchronize(), "reference warm-up");
        hipEvent_t e0, e1, e2;
        chk(hipEventCreate(&e0), "event"); chk(hipEventCreate(&e1), "event"); chk(hipEventCreate(&e2), "event");
        double tf = 0.0, tb = 0.0;
        for (int i = 0; i < iters; i++) {
            chk(hipEventRecord(e0, 0), "record"); fwd();
            chk(hipEventRecord(e1, 0), "record"); bwd();
            chk(hipEventRecord(e2, 0), "record");
            chk(hipEventSynchronize(e2), "sync");
            float a = 0.f, b = 0.f;
            chk(hipEventElapsedTime(&a, e0, e1), "elapsed"); chk(hipEventElapsedTime(&b, e1, e2), "elapsed");
            tf += a; tb += b;
        }
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(e2);
        ms[0] = (float)(tf / iters); ms[1] = (float)(tb / iters);
        return st.R;
    } catch (const std::exception& e) {
        fprintf(stderr, "[gsref] time: %s\n", e.what());
        return -1;
    }
}

// rasterizer_impl.cu:140-160 (markVisible): present [P] bytes, HOST arrays
int gsref_mark_visible(int P, const float* means3D, const float* view, const float* proj, unsigned char* present)
{
    try {
        DevArr<float> m, v, p;
        DevArr<unsigned char> out;
        m.upload(means3D, (size_t)3 * P); v.upload(view, 16); p.upload(proj, 16); out.zeros((size_t)P);
        if (P > 0) CudaRasterizer::Rasterizer::markVisible(P, m.p, v.p, p.p, reinterpret_cast<bool*>(out.p));
        chk(hipDeviceSynchronize(), "reference markVisible");
        out.download(present, (size_t)P);
        return 0;
    } catch (const std::exception& e) {
        fprintf(stderr, "[gsref] markVisible: %s\n", e.what());
        return -1;
    }
}

// rasterizer_impl.cu (Rasterizer::visible_filter, what Render.cc:784-831 calls on a 1.2x enlarged image): radii [P], HOST arrays
int gsref_visible_filter(const Scene* s, int width, int height, int* radii)
{
    try {
        const size_t P = (size_t)s->P;
        if (P == 0) return 0;
        DevBuf geom, binning, image;
        DevArr<float> m, sc, rt, v, p;
        DevArr<int> r;
        m.upload(s->means3D, 3 * P); sc.upload(s->scales, 3 * P); rt.upload(s->rotations, 4 * P); v.upload(s->viewmatrix, 16); p.upload(s->projmatrix, 16); r.zeros(P);
        std::function<char*(size_t)> ga = [&](size_t n) { return geom.get(n); };
        std::function<char*(size_t)> ba = [&](size_t n) { return binning.get(n); };
        std::function<char*(size_t)> ia = [&](size_t n) { return image.get(n); };
        CudaRasterizer::Rasterizer::visible_filter(ga, ba, ia, s->P, s->M, width, height, m.p, sc.p, s->scale_modifier, rt.p, v.p, p.p, s->tan_fovx, s->tan_fovy, false, r.p);
        chk(hipDeviceSynchronize(), "reference visible_filter");
        r.download(radii, P);
        return 0;
    } catch (const std::exception& e) {
        fprintf(stderr, "[gsref] visible_filter: %s\n", e.what());
        return -1;
    }
}

// src/simple_knn.cu:185-220 (SimpleKNN::knn, what distCUDA2 calls: src/spatial.cu:14-26): mean squared distance to the three nearest neighbours; HOST arrays
int gsref_dist2(int P, const float* points, float* mean_dists)
{
    try {
        if (P <= 0) return 0;
        DevArr<float> pts, out;
        pts.upload(points, (size_t)3 * P); out.zeros((size_t)P);
        SimpleKNN::knn(P, reinterpret_cast<float3*>(pts.p), out.p);
        chk(hipDeviceSynchronize(), "reference knn");
        out.download(mean_dists, (size_t)P);
        return 0;
    } catch (const std::exception& e) {
        fprintf(stderr, "[gsref] knn: %s\n", e.what());
        return -1;
    }
}

} // extern "C"
