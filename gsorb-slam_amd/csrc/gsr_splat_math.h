// gsr_splat_math.h — per-splat projection math (forward and backward), device side.
//
// What it computes is fixed by the reference (file:line under
// Thirdparty/diff_gaussian_rasterization/cuda_rasterizer of the reference tree):
// EWA projection forward.cu:74-113, 3D covariance :118-152, culling / radius /
// tile rectangle :188-237 and auxiliary.h:41-56,139-164, SH colour :20-71, and
// their backward passes backward.cu:20-139,144-274,278-341,346-396.
// Integer outputs (radius, tile rectangle) must match the reference's
// arithmetic bit for bit, so products and sums keep its association order and
// the library is compiled with -ffp-contract=off.
#pragma once

#include "gsr_device.h"

namespace gsr {

// column-major 3x3, m[col][row]
struct M3 { float m[3][3]; };

__device__ __forceinline__ M3 m3_mul(const M3& a, const M3& b)
{
    M3 r;
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int k = 0; k < 3; k++)
            r.m[c][k] = a.m[0][k] * b.m[c][0] + a.m[1][k] * b.m[c][1] + a.m[2][k] * b.m[c][2];
    return r;
}
__device__ __forceinline__ M3 m3_t(const M3& a)
{
    M3 r;
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int k = 0; k < 3; k++) r.m[c][k] = a.m[k][c];
    return r;
}

__device__ __forceinline__ float3 xform4x3(float3 p, const float* __restrict__ m)
{
    return make_float3(m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12],
                       m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
                       m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]);
}
__device__ __forceinline__ float4 xform4x4(float3 p, const float* __restrict__ m)
{
    return make_float4(m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12],
                       m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
                       m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14],
                       m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15]);
}

// rotation block of the quaternion (r,x,y,z), laid out like the reference's glm matrix
__device__ __forceinline__ M3 quat_R(float4 q)
{
    const float r = q.x, x = q.y, y = q.z, z = q.w;
    M3 R;
    R.m[0][0] = 1.f - 2.f * (y * y + z * z); R.m[0][1] = 2.f * (x * y - r * z); R.m[0][2] = 2.f * (x * z + r * y);
    R.m[1][0] = 2.f * (x * y + r * z); R.m[1][1] = 1.f - 2.f * (x * x + z * z); R.m[1][2] = 2.f * (y * z - r * x);
    R.m[2][0] = 2.f * (x * z - r * y); R.m[2][1] = 2.f * (y * z + r * x); R.m[2][2] = 1.f - 2.f * (x * x + y * y);
    return R;
}

// M = S*R with S diagonal: row k of R scaled by s_k (the zero products of the full
// matrix product add exact zeros and are dropped).
__device__ __forceinline__ M3 scaled_R(float3 s, const M3& R)
{
    M3 M;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        M.m[c][0] = s.x * R.m[c][0];
        M.m[c][1] = s.y * R.m[c][1];
        M.m[c][2] = s.z * R.m[c][2];
    }
    return M;
}

__device__ __forceinline__ void cov3d_from_scale_rot(float3 scale, float mod, float4 rot, float cov[6])
{
    const float3 s = make_float3(mod * scale.x, mod * scale.y, mod * scale.z);
    const M3 M = scaled_R(s, quat_R(rot));
    const M3 Sigma = m3_mul(m3_t(M), M);
    cov[0] = Sigma.m[0][0]; cov[1] = Sigma.m[0][1]; cov[2] = Sigma.m[0][2];
    cov[3] = Sigma.m[1][1]; cov[4] = Sigma.m[1][2]; cov[5] = Sigma.m[2][2];
}

struct Cov2D {
    M3 T, W, Vrk;
    float3 t;       // camera-frame mean with the fov clamp applied
    float txtz, tytz, limx, limy;
    float a, b, c;  // 2D covariance incl. the 0.3 low-pass
};

__device__ __forceinline__ Cov2D cov2d_forward(float3 mean, float fx, float fy, float tan_fovx,
                                               float tan_fovy, const float cov3D[6],
                                               const float* __restrict__ view)
{
    Cov2D o;
    float3 t = xform4x3(mean, view);
    o.limx = 1.3f * tan_fovx;
    o.limy = 1.3f * tan_fovy;
    o.txtz = t.x / t.z;
    o.tytz = t.y / t.z;
    t.x = fminf(o.limx, fmaxf(-o.limx, o.txtz)) * t.z;
    t.y = fminf(o.limy, fmaxf(-o.limy, o.tytz)) * t.z;
    o.t = t;
    M3 J;
    J.m[0][0] = fx / t.z; J.m[0][1] = 0.0f; J.m[0][2] = -(fx * t.x) / (t.z * t.z);
    J.m[1][0] = 0.0f; J.m[1][1] = fy / t.z; J.m[1][2] = -(fy * t.y) / (t.z * t.z);
    J.m[2][0] = 0.0f; J.m[2][1] = 0.0f; J.m[2][2] = 0.0f;
    o.W.m[0][0] = view[0]; o.W.m[0][1] = view[4]; o.W.m[0][2] = view[8];
    o.W.m[1][0] = view[1]; o.W.m[1][1] = view[5]; o.W.m[1][2] = view[9];
    o.W.m[2][0] = view[2]; o.W.m[2][1] = view[6]; o.W.m[2][2] = view[10];
    o.T = m3_mul(o.W, J);
    o.Vrk.m[0][0] = cov3D[0]; o.Vrk.m[0][1] = cov3D[1]; o.Vrk.m[0][2] = cov3D[2];
    o.Vrk.m[1][0] = cov3D[1]; o.Vrk.m[1][1] = cov3D[3]; o.Vrk.m[1][2] = cov3D[4];
    o.Vrk.m[2][0] = cov3D[2]; o.Vrk.m[2][1] = cov3D[4]; o.Vrk.m[2][2] = cov3D[5];
    const M3 cov = m3_mul(m3_mul(m3_t(o.T), m3_t(o.Vrk)), o.T);
    o.a = cov.m[0][0] + 0.3f;
    o.b = cov.m[0][1];
    o.c = cov.m[1][1] + 0.3f;
    return o;
}

__device__ __forceinline__ float ndc2pix(float v, int S) { return (float)(((v + 1.0) * S - 1.0) * 0.5); }

__device__ __forceinline__ void tile_rect(float px, float py, int radius, int gx, int gy, int& x0,
                                          int& y0, int& x1, int& y1)
{
    // float->int converts with saturation on gfx950 (v_cvt_i32_f32), NaN -> 0
    const float r = (float)radius;
    x0 = min(gx, max(0, (int)((px - r) / 16.0f)));
    y0 = min(gy, max(0, (int)((py - r) / 16.0f)));
    x1 = min(gx, max(0, (int)((((px + r) + 16.0f) - 1.0f) / 16.0f)));
    y1 = min(gy, max(0, (int)((((py + r) + 16.0f) - 1.0f) / 16.0f)));
}

struct Projected {
    float3 p_view;
    float px, py;
    float conic_a, conic_b, conic_c;
    int radius;
    int x0, y0, x1, y1;
};

// Everything up to the tile rectangle; false when the splat is culled.
__device__ __forceinline__ bool project_splat(float3 p, const float cov3D[6], const FrameParams& f,
                                              const float* __restrict__ view,
                                              const float* __restrict__ proj, Projected& o)
{
    o.p_view = xform4x3(p, view);
    if (o.p_view.z <= 0.2f) return false;
    const float4 hom = xform4x4(p, proj);
    const float p_w = 1.0f / (hom.w + 0.0000001f);
    const float ndc_x = hom.x * p_w, ndc_y = hom.y * p_w;
    const Cov2D c2 = cov2d_forward(p, f.focal_x, f.focal_y, f.tan_fovx, f.tan_fovy, cov3D, view);
    const float det = c2.a * c2.c - c2.b * c2.b;
    if (det == 0.0f) return false;
    const float det_inv = 1.f / det;
    o.conic_a = c2.c * det_inv;
    o.conic_b = -c2.b * det_inv;
    o.conic_c = c2.a * det_inv;
    const float mid = 0.5f * (c2.a + c2.c);
    const float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
    const float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
    const float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
    o.radius = (int)my_radius;
    o.px = ndc2pix(ndc_x, f.W);
    o.py = ndc2pix(ndc_y, f.H);
    tile_rect(o.px, o.py, o.radius, f.grid_x, f.grid_y, o.x0, o.y0, o.x1, o.y1);
    return (uint32_t)(o.x1 - o.x0) * (uint32_t)(o.y1 - o.y0) != 0u;
}

// ---- spherical harmonics -----------------------------------------------------------
__device__ constexpr float kC0 = 0.28209479177387814f;
__device__ constexpr float kC1 = 0.4886025119029199f;
__device__ constexpr float kC2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                     -1.0925484305920792f, 0.5462742152960396f};
__device__ constexpr float kC3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                     0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                     -0.5900435899266435f};

__device__ __forceinline__ float3 unit_dir(float3 p, const float* __restrict__ campos, float3& raw)
{
    raw = make_float3(p.x - campos[0], p.y - campos[1], p.z - campos[2]);
    const float len = sqrtf(raw.x * raw.x + raw.y * raw.y + raw.z * raw.z);
    return make_float3(raw.x / len, raw.y / len, raw.z / len);
}

// one colour channel; sh points at coefficient 0 of this splat, channel stride 3
__device__ __forceinline__ float sh_channel(int deg, const float* __restrict__ sh, int ch, float3 d)
{
    const float x = d.x, y = d.y, z = d.z;
#define GSR_SH(k) sh[3 * (k) + ch]
    float res = kC0 * GSR_SH(0);
    if (deg > 0) {
        res = res - kC1 * y * GSR_SH(1) + kC1 * z * GSR_SH(2) - kC1 * x * GSR_SH(3);
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            res = res + kC2[0] * xy * GSR_SH(4) + kC2[1] * yz * GSR_SH(5) +
                  kC2[2] * (2.0f * zz - xx - yy) * GSR_SH(6) + kC2[3] * xz * GSR_SH(7) +
                  kC2[4] * (xx - yy) * GSR_SH(8);
            if (deg > 2) {
                res = res + kC3[0] * y * (3.0f * xx - yy) * GSR_SH(9) + kC3[1] * xy * z * GSR_SH(10) +
                      kC3[2] * y * (4.0f * zz - xx - yy) * GSR_SH(11) +
                      kC3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * GSR_SH(12) +
                      kC3[4] * x * (4.0f * zz - xx - yy) * GSR_SH(13) + kC3[5] * z * (xx - yy) * GSR_SH(14) +
                      kC3[6] * x * (xx - 3.0f * yy) * GSR_SH(15);
            }
        }
    }
#undef GSR_SH
    return res + 0.5f;
}

// backward of one channel: writes dL_dsh[k][ch], returns d(colour_ch)/d(dir)
__device__ __forceinline__ float3 sh_channel_backward(int deg, const float* __restrict__ sh,
                                                      float* __restrict__ dsh, int ch, float3 d, float g)
{
    const float x = d.x, y = d.y, z = d.z;
#define GSR_SH(k) sh[3 * (k) + ch]
#define GSR_DSH(k) dsh[3 * (k) + ch]
    float dx = 0.f, dy = 0.f, dz = 0.f;
    GSR_DSH(0) = kC0 * g;
    if (deg > 0) {
        GSR_DSH(1) = (-kC1 * y) * g;
        GSR_DSH(2) = (kC1 * z) * g;
        GSR_DSH(3) = (-kC1 * x) * g;
        dx = -kC1 * GSR_SH(3);
        dy = -kC1 * GSR_SH(1);
        dz = kC1 * GSR_SH(2);
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            GSR_DSH(4) = (kC2[0] * xy) * g;
            GSR_DSH(5) = (kC2[1] * yz) * g;
            GSR_DSH(6) = (kC2[2] * (2.f * zz - xx - yy)) * g;
            GSR_DSH(7) = (kC2[3] * xz) * g;
            GSR_DSH(8) = (kC2[4] * (xx - yy)) * g;
            dx += kC2[0] * y * GSR_SH(4) + kC2[2] * 2.f * -x * GSR_SH(6) + kC2[3] * z * GSR_SH(7) + kC2[4] * 2.f * x * GSR_SH(8);
            dy += kC2[0] * x * GSR_SH(4) + kC2[1] * z * GSR_SH(5) + kC2[2] * 2.f * -y * GSR_SH(6) + kC2[4] * 2.f * -y * GSR_SH(8);
            dz += kC2[1] * y * GSR_SH(5) + kC2[2] * 2.f * 2.f * z * GSR_SH(6) + kC2[3] * x * GSR_SH(7);
            if (deg > 2) {
                GSR_DSH(9) = (kC3[0] * y * (3.f * xx - yy)) * g;
                GSR_DSH(10) = (kC3[1] * xy * z) * g;
                GSR_DSH(11) = (kC3[2] * y * (4.f * zz - xx - yy)) * g;
                GSR_DSH(12) = (kC3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy)) * g;
                GSR_DSH(13) = (kC3[4] * x * (4.f * zz - xx - yy)) * g;
                GSR_DSH(14) = (kC3[5] * z * (xx - yy)) * g;
                GSR_DSH(15) = (kC3[6] * x * (xx - 3.f * yy)) * g;
                dx += (kC3[0] * GSR_SH(9) * 3.f * 2.f * xy + kC3[1] * GSR_SH(10) * yz + kC3[2] * GSR_SH(11) * -2.f * xy +
                       kC3[3] * GSR_SH(12) * -3.f * 2.f * xz + kC3[4] * GSR_SH(13) * (-3.f * xx + 4.f * zz - yy) +
                       kC3[5] * GSR_SH(14) * 2.f * xz + kC3[6] * GSR_SH(15) * 3.f * (xx - yy));
                dy += (kC3[0] * GSR_SH(9) * 3.f * (xx - yy) + kC3[1] * GSR_SH(10) * xz +
                       kC3[2] * GSR_SH(11) * (-3.f * yy + 4.f * zz - xx) + kC3[3] * GSR_SH(12) * -3.f * 2.f * yz +
                       kC3[4] * GSR_SH(13) * -2.f * xy + kC3[5] * GSR_SH(14) * -2.f * yz +
                       kC3[6] * GSR_SH(15) * -3.f * 2.f * xy);
                dz += (kC3[1] * GSR_SH(10) * xy + kC3[2] * GSR_SH(11) * 4.f * 2.f * yz +
                       kC3[3] * GSR_SH(12) * 3.f * (2.f * zz - xx - yy) + kC3[4] * GSR_SH(13) * 4.f * 2.f * xz +
                       kC3[5] * GSR_SH(14) * (xx - yy));
            }
        }
    }
#undef GSR_SH
#undef GSR_DSH
    return make_float3(dx, dy, dz);
}

// d(v/|v|)/dv applied to dv (auxiliary.h:107-118)
__device__ __forceinline__ float3 dnormvdv(float3 v, float3 dv)
{
    const float sum2 = v.x * v.x + v.y * v.y + v.z * v.z;
    const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
    float3 o;
    o.x = ((+sum2 - v.x * v.x) * dv.x - v.y * v.x * dv.y - v.z * v.x * dv.z) * invsum32;
    o.y = (-v.x * v.y * dv.x + (sum2 - v.y * v.y) * dv.y - v.z * v.y * dv.z) * invsum32;
    o.z = (-v.x * v.z * dv.x - v.y * v.z * dv.y + (sum2 - v.z * v.z) * dv.z) * invsum32;
    return o;
}

} // namespace gsr
