/*
 * gsr_oracle.c — CPU restatement of the reference rasterizer. TEST
 * INFRASTRUCTURE ONLY (see gsr_oracle.h for the rules and the parity-pinning
 * statement). Plain C99, fp32 arithmetic in the reference's operation order;
 * build with -ffp-contract=off.
 *
 * Citations are relative to /root/reference; DGR = Thirdparty/
 * diff_gaussian_rasterization. Small 3x3 helpers follow the column-major
 * m[col][row] convention of the vendored glm (DGR/third_party/glm/glm/detail/
 * type_mat3x3.inl:486-520: result[c][r] = a[0][r]*b[c][0] + a[1][r]*b[c][1] +
 * a[2][r]*b[c][2]) so that index expressions can be compared one to one.
 */
#include "gsr_oracle.h"

#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef GSRO_OMP
#include <omp.h>
#endif

/* ---- DGR/cuda_rasterizer/auxiliary.h:22-39 ------------------------------------ */
static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                               -1.0925484305920792f, 0.5462742152960396f};
static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                               0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                               -0.5900435899266435f};

typedef struct { float m[3][3]; } mat3; /* m[col][row] */

static mat3 m3(float a0, float a1, float a2, float b0, float b1, float b2, float c0, float c1,
               float c2)
{
    mat3 r;
    r.m[0][0] = a0; r.m[0][1] = a1; r.m[0][2] = a2;
    r.m[1][0] = b0; r.m[1][1] = b1; r.m[1][2] = b2;
    r.m[2][0] = c0; r.m[2][1] = c1; r.m[2][2] = c2;
    return r;
}
static mat3 m3_mul(mat3 a, mat3 b)
{
    mat3 r;
    for (int c = 0; c < 3; c++)
        for (int row = 0; row < 3; row++)
            r.m[c][row] = a.m[0][row] * b.m[c][0] + a.m[1][row] * b.m[c][1] + a.m[2][row] * b.m[c][2];
    return r;
}
static mat3 m3_t(mat3 a)
{
    mat3 r;
    for (int c = 0; c < 3; c++)
        for (int row = 0; row < 3; row++) r.m[c][row] = a.m[row][c];
    return r;
}
static mat3 m3_scale(float s, mat3 a)
{
    for (int c = 0; c < 3; c++)
        for (int row = 0; row < 3; row++) a.m[c][row] = s * a.m[c][row];
    return a;
}
static float dot3(const float* a, const float* b)
{
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}

/* CUDA float->int conversion saturates and maps NaN to 0; C leaves it undefined. */
static int f2i(float f)
{
    if (f != f) return 0;
    if (f >= 2147483648.0f) return INT_MAX;
    if (f <= -2147483648.0f) return INT_MIN;
    return (int)f;
}
static uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

/* DGR/cuda_rasterizer/rasterizer_impl.cu:36-51 */
/* Test instrumentation: caps the OpenMP team (small scenes run slower on 256 threads than on 16). No-op in the
 * single-thread build. */
void gsro_set_threads(int n)
{
#ifdef GSRO_OMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

uint32_t gsro_higher_msb(uint32_t n)
{
    uint32_t msb = sizeof(n) * 4, step = msb;
    while (step > 1) {
        step /= 2;
        if (n >> msb) msb += step; else msb -= step;
    }
    if (n >> msb) msb++;
    return msb;
}

/* DGR/cuda_rasterizer/auxiliary.h:41-44: evaluated in double, returned as float */
static float ndc2pix(float v, int S) { return (float)(((v + 1.0) * S - 1.0) * 0.5); }

/* DGR/cuda_rasterizer/auxiliary.h:46-56 */
static void get_rect(const float p[2], int max_radius, uint32_t gx, uint32_t gy,
                     uint32_t rmin[2], uint32_t rmax[2])
{
    rmin[0] = umin(gx, (uint32_t)imax(0, f2i((p[0] - max_radius) / GSRO_BLOCK_X)));
    rmin[1] = umin(gy, (uint32_t)imax(0, f2i((p[1] - max_radius) / GSRO_BLOCK_Y)));
    rmax[0] = umin(gx, (uint32_t)imax(0, f2i((p[0] + max_radius + GSRO_BLOCK_X - 1) / GSRO_BLOCK_X)));
    rmax[1] = umin(gy, (uint32_t)imax(0, f2i((p[1] + max_radius + GSRO_BLOCK_Y - 1) / GSRO_BLOCK_Y)));
}

/* DGR/cuda_rasterizer/auxiliary.h:58-78 */
static void xform4x3(const float p[3], const float* m, float o[3])
{
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
}
static void xform4x4(const float p[3], const float* m, float o[4])
{
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
    o[3] = m[3] * p[0] + m[7] * p[1] + m[11] * p[2] + m[15];
}

/* DGR/cuda_rasterizer/auxiliary.h:139-164 (prefiltered trap not restated: it aborts) */
static int in_frustum(const float p[3], const float* view, float p_view[3])
{
    xform4x3(p, view, p_view);
    return !(p_view[2] <= 0.2f);
}

void gsro_mark_visible(int P, const float* means3D, const float* viewmatrix,
                       const float* projmatrix, uint8_t* present)
{
    (void)projmatrix;
    for (int i = 0; i < P; i++) {
        float pv[3];
        present[i] = (uint8_t)in_frustum(means3D + 3 * i, viewmatrix, pv);
    }
}

/* quaternion (r,x,y,z) -> glm-layout rotation, DGR/cuda_rasterizer/forward.cu:126-138 */
static mat3 quat_to_R(const float q[4])
{
    float r = q[0], x = q[1], y = q[2], z = q[3];
    return m3(1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
              2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
              2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
}

/* DGR/cuda_rasterizer/forward.cu:118-152 (quaternion used un-normalised, :127) */
static void compute_cov3d(const float scale[3], float mod, const float rot[4], float cov3D[6])
{
    mat3 S = m3(1, 0, 0, 0, 1, 0, 0, 0, 1);
    S.m[0][0] = mod * scale[0];
    S.m[1][1] = mod * scale[1];
    S.m[2][2] = mod * scale[2];
    mat3 R = quat_to_R(rot);
    mat3 M = m3_mul(S, R);
    mat3 Sigma = m3_mul(m3_t(M), M);
    cov3D[0] = Sigma.m[0][0];
    cov3D[1] = Sigma.m[0][1];
    cov3D[2] = Sigma.m[0][2];
    cov3D[3] = Sigma.m[1][1];
    cov3D[4] = Sigma.m[1][2];
    cov3D[5] = Sigma.m[2][2];
}

typedef struct {
    mat3 J, W, T, Vrk, cov;
    float t[3];
    float txtz, tytz, limx, limy;
} cov2d_ctx;

/* DGR/cuda_rasterizer/forward.cu:74-113; the same intermediate values are
 * recomputed by the backward at backward.cu:162-199. */
static void cov2d_forward(const float mean[3], float fx, float fy, float tan_fovx, float tan_fovy,
                          const float* cov3D, const float* view, cov2d_ctx* c)
{
    xform4x3(mean, view, c->t);
    c->limx = 1.3f * tan_fovx;
    c->limy = 1.3f * tan_fovy;
    c->txtz = c->t[0] / c->t[2];
    c->tytz = c->t[1] / c->t[2];
    c->t[0] = fminf(c->limx, fmaxf(-c->limx, c->txtz)) * c->t[2];
    c->t[1] = fminf(c->limy, fmaxf(-c->limy, c->tytz)) * c->t[2];
    const float* t = c->t;
    c->J = m3(fx / t[2], 0.0f, -(fx * t[0]) / (t[2] * t[2]),
              0.0f, fy / t[2], -(fy * t[1]) / (t[2] * t[2]),
              0, 0, 0);
    c->W = m3(view[0], view[4], view[8], view[1], view[5], view[9], view[2], view[6], view[10]);
    c->T = m3_mul(c->W, c->J);
    c->Vrk = m3(cov3D[0], cov3D[1], cov3D[2], cov3D[1], cov3D[3], cov3D[4], cov3D[2], cov3D[4],
                cov3D[5]);
    c->cov = m3_mul(m3_mul(m3_t(c->T), m3_t(c->Vrk)), c->T);
    c->cov.m[0][0] += 0.3f;
    c->cov.m[1][1] += 0.3f;
}

/* DGR/cuda_rasterizer/forward.cu:20-71, one colour channel at a time */
static float sh_to_channel(int deg, const float* sh /* [M][3] */, int ch, float x, float y, float z)
{
#define SH(k) sh[3 * (k) + ch]
    float result = SH_C0 * SH(0);
    if (deg > 0) {
        result = result - SH_C1 * y * SH(1) + SH_C1 * z * SH(2) - SH_C1 * x * SH(3);
        if (deg > 1) {
            float xx = x * x, yy = y * y, zz = z * z;
            float xy = x * y, yz = y * z, xz = x * z;
            result = result + SH_C2[0] * xy * SH(4) + SH_C2[1] * yz * SH(5) +
                     SH_C2[2] * (2.0f * zz - xx - yy) * SH(6) + SH_C2[3] * xz * SH(7) +
                     SH_C2[4] * (xx - yy) * SH(8);
            if (deg > 2) {
                result = result + SH_C3[0] * y * (3.0f * xx - yy) * SH(9) +
                         SH_C3[1] * xy * z * SH(10) + SH_C3[2] * y * (4.0f * zz - xx - yy) * SH(11) +
                         SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * SH(12) +
                         SH_C3[4] * x * (4.0f * zz - xx - yy) * SH(13) +
                         SH_C3[5] * z * (xx - yy) * SH(14) + SH_C3[6] * x * (xx - 3.0f * yy) * SH(15);
            }
        }
    }
#undef SH
    return result + 0.5f;
}

/* direct access to the SH colour rule for the golden-vector test (forward.cu:20-71 incl.
 * the +0.5 offset and the clamp at :63-70) */
void gsro_eval_sh(int P, int deg, int M, const float* shs, const float* dirs, float* rgb, uint8_t* clamped)
{
    for (int i = 0; i < P; i++)
        for (int ch = 0; ch < 3; ch++) {
            float v = sh_to_channel(deg, shs + (size_t)i * M * 3, ch, dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]);
            clamped[3 * i + ch] = (v < 0);
            rgb[3 * i + ch] = fmaxf(v, 0.0f);
        }
}

static void view_dir(const float pos[3], const float campos[3], float dir_orig[3], float dir[3])
{
    for (int k = 0; k < 3; k++) dir_orig[k] = pos[k] - campos[k];
    float len = sqrtf(dot3(dir_orig, dir_orig));
    for (int k = 0; k < 3; k++) dir[k] = dir_orig[k] / len;
}

/* shared by preprocess and filter_preprocess: everything up to the tile rect.
 * Returns 0 if the splat is culled. DGR/cuda_rasterizer/forward.cu:193-237. */
static int project_splat(const float* p_orig, const float* scale, float mod, const float* rot,
                         const float* cov3D_precomp_i, const float* view, const float* proj,
                         int W, int H, float fx, float fy, float tan_fovx, float tan_fovy,
                         float* cov3D_store, float p_view[3], float conic[3], float* radius,
                         float point_image[2], uint32_t rmin[2], uint32_t rmax[2])
{
    if (!in_frustum(p_orig, view, p_view)) return 0;
    float p_hom[4];
    xform4x4(p_orig, proj, p_hom);
    float p_w = 1.0f / (p_hom[3] + 0.0000001f);
    float p_proj[3] = {p_hom[0] * p_w, p_hom[1] * p_w, p_hom[2] * p_w};

    const float* cov3D;
    if (cov3D_precomp_i) cov3D = cov3D_precomp_i;
    else { compute_cov3d(scale, mod, rot, cov3D_store); cov3D = cov3D_store; }

    cov2d_ctx c;
    cov2d_forward(p_orig, fx, fy, tan_fovx, tan_fovy, cov3D, view, &c);
    float cx = c.cov.m[0][0], cy = c.cov.m[0][1], cz = c.cov.m[1][1];

    float det = cx * cz - cy * cy;
    if (det == 0.0f) return 0;
    float det_inv = 1.f / det;
    conic[0] = cz * det_inv; conic[1] = -cy * det_inv; conic[2] = cx * det_inv;

    float mid = 0.5f * (cx + cz);
    float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
    float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
    *radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
    point_image[0] = ndc2pix(p_proj[0], W);
    point_image[1] = ndc2pix(p_proj[1], H);
    uint32_t gx = (W + GSRO_BLOCK_X - 1) / GSRO_BLOCK_X, gy = (H + GSRO_BLOCK_Y - 1) / GSRO_BLOCK_Y;
    get_rect(point_image, f2i(*radius), gx, gy, rmin, rmax);
    if ((rmax[0] - rmin[0]) * (rmax[1] - rmin[1]) == 0) return 0;
    return 1;
}

void gsro_preprocess(int P, int D, int M, const float* means3D, const float* scales,
                     float scale_modifier, const float* rotations, const float* opacities,
                     const float* shs, const float* cov3D_precomp, const float* colors_precomp,
                     const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                     int W, int H, float tan_fovx, float tan_fovy, int* radii, float* means2D,
                     float* depths, float* cov3Ds, float* rgb, uint8_t* clamped,
                     float* conic_opacity, uint32_t* tiles_touched)
{
    /* DGR/cuda_rasterizer/rasterizer_impl.cu:227-228 */
    const float focal_y = H / (2.0f * tan_fovy);
    const float focal_x = W / (2.0f * tan_fovx);
#ifdef GSRO_OMP
#pragma omp parallel for schedule(static)
#endif
    for (int idx = 0; idx < P; idx++) {
        radii[idx] = 0;
        tiles_touched[idx] = 0;
        float p_view[3], conic[3], radius, pim[2];
        uint32_t rmin[2], rmax[2];
        const float* p = means3D + 3 * idx;
        if (!project_splat(p, scales ? scales + 3 * idx : NULL, scale_modifier,
                           rotations ? rotations + 4 * idx : NULL,
                           cov3D_precomp ? cov3D_precomp + 6 * idx : NULL, viewmatrix, projmatrix,
                           W, H, focal_x, focal_y, tan_fovx, tan_fovy, cov3Ds + 6 * idx, p_view,
                           conic, &radius, pim, rmin, rmax))
            continue;
        if (colors_precomp == NULL) { /* forward.cu:241-247 + :20-71 */
            float d0[3], d[3];
            view_dir(p, cam_pos, d0, d);
            for (int ch = 0; ch < 3; ch++) {
                float v = sh_to_channel(D, shs + (size_t)idx * M * 3, ch, d[0], d[1], d[2]);
                clamped[3 * idx + ch] = (v < 0);
                rgb[3 * idx + ch] = fmaxf(v, 0.0f);
            }
        }
        depths[idx] = p_view[2];
        radii[idx] = f2i(radius);
        means2D[2 * idx] = pim[0];
        means2D[2 * idx + 1] = pim[1];
        conic_opacity[4 * idx + 0] = conic[0];
        conic_opacity[4 * idx + 1] = conic[1];
        conic_opacity[4 * idx + 2] = conic[2];
        conic_opacity[4 * idx + 3] = opacities[idx];
        tiles_touched[idx] = (rmax[1] - rmin[1]) * (rmax[0] - rmin[0]);
    }
}

void gsro_filter_preprocess(int P, const float* means3D, const float* scales, float scale_modifier,
                            const float* rotations, const float* viewmatrix,
                            const float* projmatrix, int W, int H, float tan_fovx, float tan_fovy,
                            int* radii)
{
    /* DGR/cuda_rasterizer/rasterizer_impl.cu:365-366 */
    const float focal_y = H / (2.0f * tan_fovy);
    const float focal_x = W / (2.0f * tan_fovx);
    for (int idx = 0; idx < P; idx++) {
        radii[idx] = 0;
        float p_view[3], conic[3], radius, pim[2], cov3D[6];
        uint32_t rmin[2], rmax[2];
        if (!project_splat(means3D + 3 * idx, scales + 3 * idx, scale_modifier, rotations + 4 * idx,
                           NULL, viewmatrix, projmatrix, W, H, focal_x, focal_y, tan_fovx, tan_fovy,
                           cov3D, p_view, conic, &radius, pim, rmin, rmax))
            continue;
        radii[idx] = f2i(radius);
    }
}

void gsro_inclusive_sum(int P, const uint32_t* in, uint32_t* out)
{
    uint32_t s = 0;
    for (int i = 0; i < P; i++) { s += in[i]; out[i] = s; }
}

void gsro_duplicate_with_keys(int P, const float* means2D, const float* depths,
                              const uint32_t* offsets, const int* radii, int W, int H,
                              uint64_t* keys, uint32_t* values)
{
    uint32_t gx = (W + GSRO_BLOCK_X - 1) / GSRO_BLOCK_X, gy = (H + GSRO_BLOCK_Y - 1) / GSRO_BLOCK_Y;
    for (int idx = 0; idx < P; idx++) {
        if (!(radii[idx] > 0)) continue;
        uint32_t off = (idx == 0) ? 0 : offsets[idx - 1];
        uint32_t rmin[2], rmax[2];
        get_rect(means2D + 2 * idx, radii[idx], gx, gy, rmin, rmax);
        uint32_t dbits;
        memcpy(&dbits, depths + idx, 4);
        for (uint32_t y = rmin[1]; y < rmax[1]; y++)
            for (uint32_t x = rmin[0]; x < rmax[0]; x++) {
                uint64_t key = (uint64_t)(y * gx + x);
                key <<= 32;
                key |= dbits;
                keys[off] = key;
                values[off] = (uint32_t)idx;
                off++;
            }
    }
}

void gsro_sort_pairs(size_t n, const uint64_t* keys_in, const uint32_t* vals_in, uint64_t* keys_out,
                     uint32_t* vals_out, int end_bit)
{
    /* stable LSD radix sort, 8-bit digits, over bits [0, end_bit) */
    if (n == 0) return;
    uint64_t* kb[2];
    uint32_t* vb[2];
    kb[0] = (uint64_t*)malloc(n * 8); vb[0] = (uint32_t*)malloc(n * 4);
    kb[1] = (uint64_t*)malloc(n * 8); vb[1] = (uint32_t*)malloc(n * 4);
    memcpy(kb[0], keys_in, n * 8);
    memcpy(vb[0], vals_in, n * 4);
    int cur = 0;
    size_t* cnt = (size_t*)malloc(257 * sizeof(size_t));
    for (int shift = 0; shift < end_bit; shift += 8) {
        int bits = end_bit - shift < 8 ? end_bit - shift : 8;
        uint64_t mask = (1ull << bits) - 1;
        memset(cnt, 0, 257 * sizeof(size_t));
        for (size_t i = 0; i < n; i++) cnt[((kb[cur][i] >> shift) & mask) + 1]++;
        for (int d = 0; d < 256; d++) cnt[d + 1] += cnt[d];
        for (size_t i = 0; i < n; i++) {
            size_t d = (kb[cur][i] >> shift) & mask;
            kb[cur ^ 1][cnt[d]] = kb[cur][i];
            vb[cur ^ 1][cnt[d]] = vb[cur][i];
            cnt[d]++;
        }
        cur ^= 1;
    }
    memcpy(keys_out, kb[cur], n * 8);
    memcpy(vals_out, vb[cur], n * 4);
    free(cnt); free(kb[0]); free(kb[1]); free(vb[0]); free(vb[1]);
}

void gsro_identify_tile_ranges(size_t L, const uint64_t* keys, uint32_t* ranges)
{
    for (size_t idx = 0; idx < L; idx++) {
        uint32_t currtile = (uint32_t)(keys[idx] >> 32);
        if (idx == 0) ranges[2 * currtile] = 0;
        else {
            uint32_t prevtile = (uint32_t)(keys[idx - 1] >> 32);
            if (currtile != prevtile) {
                ranges[2 * prevtile + 1] = (uint32_t)idx;
                ranges[2 * currtile] = (uint32_t)idx;
            }
        }
        if (idx == L - 1) ranges[2 * currtile + 1] = (uint32_t)L;
    }
}

void gsro_render_forward(int W, int H, const uint32_t* ranges, const uint32_t* point_list,
                         const float* means2D, const float* features, const float* conic_opacity,
                         const float* depths, const float* bg, float* final_T, uint32_t* n_contrib,
                         float* out_color, float* out_depth)
{
    const int gx = (W + GSRO_BLOCK_X - 1) / GSRO_BLOCK_X, gy = (H + GSRO_BLOCK_Y - 1) / GSRO_BLOCK_Y;
#ifdef GSRO_OMP
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
#endif
    for (int ty = 0; ty < gy; ty++)
        for (int tx = 0; tx < gx; tx++) {
            const uint32_t r0 = ranges[2 * (ty * gx + tx)], r1 = ranges[2 * (ty * gx + tx) + 1];
            for (int ly = 0; ly < GSRO_BLOCK_Y; ly++)
                for (int lx = 0; lx < GSRO_BLOCK_X; lx++) {
                    const int px = tx * GSRO_BLOCK_X + lx, py = ty * GSRO_BLOCK_Y + ly;
                    if (!(px < W && py < H)) continue;
                    const int pix_id = W * py + px;
                    const float pixf[2] = {(float)px, (float)py};
                    float T = 1.0f, C[3] = {0, 0, 0}, Dp = 0.0f;
                    uint32_t contributor = 0, last_contributor = 0;
                    for (uint32_t k = r0; k < r1; k++) { /* forward.cu:339-391 */
                        contributor++;
                        const uint32_t id = point_list[k];
                        const float dx = means2D[2 * id] - pixf[0], dy = means2D[2 * id + 1] - pixf[1];
                        const float* co = conic_opacity + 4 * id;
                        const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                        if (power > 0.0f) continue;
                        const float alpha = fminf(0.99f, co[3] * expf(power));
                        if (alpha < 1.0f / 255.0f) continue;
                        const float test_T = T * (1 - alpha);
                        if (test_T < 0.0001f) break; /* done = true */
                        for (int ch = 0; ch < 3; ch++) C[ch] += features[id * 3 + ch] * alpha * T;
                        if (T > 0.5f) Dp = depths[id]; /* median depth, forward.cu:374-379 */
                        T = test_T;
                        last_contributor = contributor;
                    }
                    final_T[pix_id] = T;
                    n_contrib[pix_id] = last_contributor;
                    for (int ch = 0; ch < 3; ch++) out_color[ch * H * W + pix_id] = C[ch] + T * bg[ch];
                    out_depth[pix_id] = Dp;
                }
        }
}

/* Measurement instrumentation (not in the reference): how many (pixel, splat) pairs the forward of the last
 * gsro_forward on `st` blended (passed power <= 0, alpha >= 1/255 and the stop test) and how many it evaluated
 * (list entries walked until the pixel stopped). bench.py uses the first number for `useful_lane_frac`. */
void gsro_blend_census(int W, int H, const uint32_t* ranges, const uint32_t* point_list, const float* means2D,
                       const float* conic_opacity, const uint32_t* n_contrib, unsigned long long* blended,
                       unsigned long long* evaluated)
{
    const int gx = (W + GSRO_BLOCK_X - 1) / GSRO_BLOCK_X, gy = (H + GSRO_BLOCK_Y - 1) / GSRO_BLOCK_Y;
    unsigned long long nb = 0, ne = 0;
#ifdef GSRO_OMP
#pragma omp parallel for schedule(dynamic, 1) collapse(2) reduction(+ : nb, ne)
#endif
    for (int ty = 0; ty < gy; ty++)
        for (int tx = 0; tx < gx; tx++) {
            const uint32_t r0 = ranges[2 * (ty * gx + tx)];
            for (int ly = 0; ly < GSRO_BLOCK_Y; ly++)
                for (int lx = 0; lx < GSRO_BLOCK_X; lx++) {
                    const int px = tx * GSRO_BLOCK_X + lx, py = ty * GSRO_BLOCK_Y + ly;
                    if (!(px < W && py < H)) continue;
                    const uint32_t last = n_contrib[W * py + px];
                    ne += last;
                    for (uint32_t k = r0; k < r0 + last; k++) { /* every valid entry up to the last contributor was blended */
                        const uint32_t id = point_list[k];
                        const float dx = means2D[2 * id] - (float)px, dy = means2D[2 * id + 1] - (float)py;
                        const float* co = conic_opacity + 4 * id;
                        const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                        if (power > 0.0f) continue;
                        if (fminf(0.99f, co[3] * expf(power)) < 1.0f / 255.0f) continue;
                        nb++;
                    }
                }
        }
    *blended = nb;
    *evaluated = ne;
}

/* Measurement instrumentation (not in the reference): what the backward blend kernel's round structure would
 * cost for other PATCH GEOMETRIES. The kernel (csrc/gsr_blend.h) gives one wave an 8x8 pixel quad and cuts it into patches of
 * pw x ph pixels, one patch per group of pw*ph lanes ("row"); a round parks <= 64 quad-hit records, every row walks the list of
 * the records that reach ITS patch (exact test of the alpha >= 1/255 ellipse against the rectangle of the patch's pixel
 * centres: the kernel's patch_reach4, restated for any rectangle), the wave runs max-over-rows iterations (rounded up to an
 * even number) and turns around for a reduce phase every 64 / rows iterations (one lane per (row, iteration) pair).
 * Per geometry g (pw = geoms[2g], ph = geoms[2g+1]) out[8g..]: 0 quad hits walked, 1 patch hits, 2 wave iterations,
 * 3 reduce phases, 4 rounds, 5 lane slots of the blend loop (64 x iterations), 6 blended (pixel, splat) pairs, 7 (pixel, splat)
 * pairs inside reached patches. Records behind the quad's last contributor are dropped as the kernel drops them; the forward's
 * round alignment (qdone = multiple of 64 records) is reproduced. */
static int gsro_rect_reach(float mx, float my, float ca, float cb, float cc, float op, float X0, float Y0, float w, float h)
{
    if (op < 1.0f / 255.0f) return 0;
    if (!(ca > 0.f) || !(cc > 0.f) || !(ca * cc > cb * cb)) return 1;
    const float tau = logf(255.0f * op) + 0.01f;
    const float dxl = mx - (X0 + w - 1.f), dxh = mx - X0, dyl = my - (Y0 + h - 1.f), dyh = my - Y0;
    const float dxc = fminf(fmaxf(0.f, dxl), dxh), dyc = fminf(fmaxf(0.f, dyl), dyh);
    if (dxc == 0.f && dyc == 0.f) return 1;
    float q = 3.0e38f;
    if (dxc != 0.f) {
        const float dy = fminf(fmaxf(-cb * dxc / cc, dyl), dyh);
        q = fminf(q, 0.5f * (ca * dxc * dxc + cc * dy * dy) + cb * dxc * dy);
    }
    if (dyc != 0.f) {
        const float dx = fminf(fmaxf(-cb * dyc / ca, dxl), dxh);
        q = fminf(q, 0.5f * (ca * dx * dx + cc * dyc * dyc) + cb * dx * dyc);
    }
    return !(q > tau);
}

void gsro_geometry_census(int W, int H, const uint32_t* ranges, const uint32_t* point_list, const float* means2D,
                          const float* conic_opacity, const uint32_t* n_contrib, int ngeom, const int* geoms,
                          unsigned long long* out)
{
    const int gx = (W + GSRO_BLOCK_X - 1) / GSRO_BLOCK_X, gy = (H + GSRO_BLOCK_Y - 1) / GSRO_BLOCK_Y;
    for (int g = 0; g < ngeom; g++) {
        const int pw = geoms[2 * g], ph = geoms[2 * g + 1], ncx = 8 / pw, ncy = 8 / ph, rows = ncx * ncy, ring = 64 / rows;
        unsigned long long a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;
#ifdef GSRO_OMP
#pragma omp parallel for schedule(dynamic, 1) collapse(2) reduction(+ : a0, a1, a2, a3, a4, a5, a6, a7)
#endif
        for (int ty = 0; ty < gy; ty++)
            for (int tx = 0; tx < gx; tx++) {
                const uint32_t r0 = ranges[2 * (ty * gx + tx)], r1 = ranges[2 * (ty * gx + tx) + 1];
                const int n = (int)(r1 - r0);
                if (n <= 0) continue;
                uint64_t* pm = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)n); /* patch masks of the quad's records */
                uint32_t* ppos = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)n);
                for (int quad = 0; quad < 4; quad++) {
                    const int X0 = tx * 16 + (quad & 1) * 8, Y0 = ty * 16 + (quad >> 1) * 8;
                    uint32_t ntodo = 0;
                    for (int j = 0; j < 8; j++)
                        for (int i = 0; i < 8; i++)
                            if (X0 + i < W && Y0 + j < H && n_contrib[W * (Y0 + j) + X0 + i] > ntodo) ntodo = n_contrib[W * (Y0 + j) + X0 + i];
                    if (ntodo == 0) continue;
                    /* the quad's records, front to back (all of the tile list: the forward's round alignment needs them) */
                    int nq = 0, nlive = 0;
                    for (int k = 0; k < n; k++) {
                        const uint32_t id = point_list[r0 + k];
                        const float* co = conic_opacity + 4 * id;
                        const float mx = means2D[2 * id], my = means2D[2 * id + 1];
                        uint64_t m = 0;
                        for (int cy = 0; cy < ncy; cy++)
                            for (int cx = 0; cx < ncx; cx++)
                                if (gsro_rect_reach(mx, my, co[0], co[1], co[2], co[3], (float)(X0 + cx * pw), (float)(Y0 + cy * ph), (float)pw, (float)ph))
                                    m |= 1ull << (cy * ncx + cx);
                        if (!m) continue;
                        pm[nq] = m; ppos[nq] = (uint32_t)k; nq++;
                        if ((uint32_t)k < ntodo) nlive = nq;
                    }
                    int cq = ((nlive + 63) / 64) * 64; /* the forward stops after the round that finished the quad */
                    if (cq > nq) cq = nq;
                    for (int top = cq; top > 0; top -= 64) { /* backward rounds: records [top-64, top), back to front */
                        int c[64], count = 0;
                        for (int r = 0; r < rows; r++) c[r] = 0;
                        for (int e = top - 1; e >= 0 && e >= top - 64; e--) {
                            if (ppos[e] >= ntodo) continue; /* behind every pixel's last contributor: dropped */
                            count++;
                            const uint32_t id = point_list[r0 + ppos[e]];
                            const float* co = conic_opacity + 4 * id;
                            for (int r = 0; r < rows; r++)
                                if ((pm[e] >> r) & 1) {
                                    c[r]++; a1++; a7 += (unsigned long long)(pw * ph);
                                    const int px0 = X0 + (r % ncx) * pw, py0 = Y0 + (r / ncx) * ph;
                                    for (int j = 0; j < ph; j++)
                                        for (int i = 0; i < pw; i++) {
                                            const int px = px0 + i, py = py0 + j;
                                            if (!(px < W && py < H) || ppos[e] >= n_contrib[W * py + px]) continue;
                                            const float dx = means2D[2 * id] - (float)px, dy = means2D[2 * id + 1] - (float)py;
                                            const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                                            if (power > 0.0f) continue;
                                            if (fminf(0.99f, co[3] * expf(power)) < 1.0f / 255.0f) continue;
                                            a6++;
                                        }
                                }
                        }
                        if (count == 0) continue;
                        int maxc = 0;
                        for (int r = 0; r < rows; r++) if (c[r] > maxc) maxc = c[r];
                        maxc = (maxc + 1) & ~1;
                        a0 += (unsigned long long)count; a2 += (unsigned long long)maxc; a3 += (unsigned long long)((maxc + ring - 1) / ring);
                        a4++; a5 += 64ull * (unsigned long long)maxc;
                    }
                }
                free(pm); free(ppos);
            }
        unsigned long long* o = out + 8 * g;
        o[0] = a0; o[1] = a1; o[2] = a2; o[3] = a3; o[4] = a4; o[5] = a5; o[6] = a6; o[7] = a7;
    }
}

/* Measurement instrumentation (round 4, second census): what the forward could tell the backward. For the shipped 4x4 patches:
 * how many pixels a patch hit really blends (the reach test is geometric: an ellipse that clips a corner of the patch's rectangle may
 * contain no pixel centre; pixels already finished blend nothing), and what the blend loop would run with
 *   Z: lists without the patch hits that blend no pixel (the forward would log the live mask),
 *   P: per-PIXEL lists inside every 16-entry window of a row's list (each lane walks the entries that blend ITS pixel).
 * out[0..16]: patch hits by blended pixels; out[17]: quad hits (parked records); out[18]: quad hits without any blended pixel;
 * out[19]: wave iterations today; out[20]: reduce phases today; out[21]: wave iterations Z; out[22]: reduce phases Z;
 * out[23]: wave iterations P (sum over windows of the max over the 64 lanes); out[24]: windows P; out[25]: rounds;
 * out[26]: wave iterations of P on Z's lists; out[27]: per-round max over lanes of the pixel's count (no windows) */
void gsro_pixlist_census(int W, int H, const uint32_t* ranges, const uint32_t* point_list, const float* means2D,
                         const float* conic_opacity, const uint32_t* n_contrib, unsigned long long* out)
{
    const int gx = (W + GSRO_BLOCK_X - 1) / GSRO_BLOCK_X, gy = (H + GSRO_BLOCK_Y - 1) / GSRO_BLOCK_Y;
    unsigned long long acc[28];
    for (int i = 0; i < 28; i++) acc[i] = 0;
#ifdef GSRO_OMP
#pragma omp parallel for schedule(dynamic, 1) collapse(2) reduction(+ : acc[:28])
#endif
    for (int ty = 0; ty < gy; ty++)
        for (int tx = 0; tx < gx; tx++) {
            const uint32_t r0 = ranges[2 * (ty * gx + tx)], r1 = ranges[2 * (ty * gx + tx) + 1];
            const int n = (int)(r1 - r0);
            if (n <= 0) continue;
            uint32_t* pm = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)n);
            uint32_t* ppos = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)n);
            for (int quad = 0; quad < 4; quad++) {
                const int X0 = tx * 16 + (quad & 1) * 8, Y0 = ty * 16 + (quad >> 1) * 8;
                uint32_t ntodo = 0;
                for (int j = 0; j < 8; j++)
                    for (int i = 0; i < 8; i++)
                        if (X0 + i < W && Y0 + j < H && n_contrib[W * (Y0 + j) + X0 + i] > ntodo) ntodo = n_contrib[W * (Y0 + j) + X0 + i];
                if (ntodo == 0) continue;
                int nq = 0, nlive = 0;
                for (int k = 0; k < n; k++) {
                    const uint32_t id = point_list[r0 + k];
                    const float* co = conic_opacity + 4 * id;
                    uint32_t m = 0;
                    for (int c = 0; c < 4; c++)
                        if (gsro_rect_reach(means2D[2 * id], means2D[2 * id + 1], co[0], co[1], co[2], co[3], (float)(X0 + (c & 1) * 4), (float)(Y0 + (c >> 1) * 4), 4.f, 4.f))
                            m |= 1u << c;
                    if (!m) continue;
                    pm[nq] = m; ppos[nq] = (uint32_t)k; nq++;
                    if ((uint32_t)k < ntodo) nlive = nq;
                }
                int cq = ((nlive + 63) / 64) * 64;
                if (cq > nq) cq = nq;
                for (int top = cq; top > 0; top -= 64) {
                    int c[4] = {0, 0, 0, 0}, cz[4] = {0, 0, 0, 0}, count = 0;
                    static __thread uint16_t bm[4][64], bmz[4][64]; /* blended-pixel masks of the rows' list entries, in walking order */
                    for (int e = top - 1; e >= 0 && e >= top - 64; e--) {
                        if (ppos[e] >= ntodo) continue;
                        count++;
                        const uint32_t id = point_list[r0 + ppos[e]];
                        const float* co = conic_opacity + 4 * id;
                        int any = 0;
                        for (int r = 0; r < 4; r++)
                            if ((pm[e] >> r) & 1) {
                                const int px0 = X0 + (r & 1) * 4, py0 = Y0 + (r >> 1) * 4;
                                uint32_t mask = 0;
                                for (int p = 0; p < 16; p++) {
                                    const int px = px0 + (p & 3), py = py0 + (p >> 2);
                                    if (!(px < W && py < H) || ppos[e] >= n_contrib[W * py + px]) continue;
                                    const float dx = means2D[2 * id] - (float)px, dy = means2D[2 * id + 1] - (float)py;
                                    const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                                    if (power > 0.0f) continue;
                                    if (fminf(0.99f, co[3] * expf(power)) < 1.0f / 255.0f) continue;
                                    mask |= 1u << p;
                                }
                                acc[__builtin_popcount(mask)]++;
                                bm[r][c[r]++] = (uint16_t)mask;
                                if (mask) { bmz[r][cz[r]++] = (uint16_t)mask; any = 1; }
                            }
                        acc[17]++;
                        if (!any) acc[18]++;
                    }
                    if (count == 0) continue;
                    acc[25]++;
                    int maxc = 0, maxz = 0;
                    for (int r = 0; r < 4; r++) { if (c[r] > maxc) maxc = c[r]; if (cz[r] > maxz) maxz = cz[r]; }
                    maxc = (maxc + 1) & ~1; maxz = (maxz + 1) & ~1;
                    acc[19] += (unsigned long long)maxc; acc[20] += (unsigned long long)((maxc + 15) / 16);
                    acc[21] += (unsigned long long)maxz; acc[22] += (unsigned long long)((maxz + 15) / 16);
                    for (int variant = 0; variant < 2; variant++) { /* P on today's lists, P on Z's lists */
                        const int mx = variant ? maxz : maxc;
                        for (int w0 = 0; w0 < mx; w0 += 16) {
                            int best = 0;
                            for (int r = 0; r < 4; r++) {
                                const int cr = variant ? cz[r] : c[r];
                                for (int p = 0; p < 16; p++) {
                                    int cnt = 0;
                                    for (int k = w0; k < w0 + 16 && k < cr; k++) cnt += ((variant ? bmz[r][k] : bm[r][k]) >> p) & 1;
                                    if (cnt > best) best = cnt;
                                }
                            }
                            best = (best + 1) & ~1;
                            if (variant) acc[26] += (unsigned long long)best;
                            else { acc[23] += (unsigned long long)best; acc[24]++; }
                        }
                    }
                    {
                        int best = 0;
                        for (int r = 0; r < 4; r++)
                            for (int p = 0; p < 16; p++) {
                                int cnt = 0;
                                for (int k = 0; k < c[r]; k++) cnt += (bm[r][k] >> p) & 1;
                                if (cnt > best) best = cnt;
                            }
                        acc[27] += (unsigned long long)((best + 1) & ~1);
                    }
                }
            }
            free(pm); free(ppos);
        }
    for (int i = 0; i < 28; i++) out[i] = acc[i];
}

/* Test instrumentation (not in the reference): smallest relative distance of any
 * data-dependent branch of forward.cu:346-379 from flipping, per pixel. exp() is not
 * bit-reproducible across libm/CUDA/HIP, so a pixel whose margin is below the
 * evaluation error of exp (~1e-6) may legitimately take the other branch; the parity
 * tests mask such pixels instead of loosening the 1e-4 tolerance.
 * margin_color covers power>0, alpha<1/255 and T(1-alpha)<1e-4; margin_depth adds T>0.5. */
void gsro_render_margins(int W, int H, const uint32_t* ranges, const uint32_t* point_list,
                         const float* means2D, const float* conic_opacity, float* margin_color,
                         float* margin_depth)
{
    const int gx = (W + GSRO_BLOCK_X - 1) / GSRO_BLOCK_X, gy = (H + GSRO_BLOCK_Y - 1) / GSRO_BLOCK_Y;
#ifdef GSRO_OMP
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
#endif
    for (int ty = 0; ty < gy; ty++)
        for (int tx = 0; tx < gx; tx++) {
            const uint32_t r0 = ranges[2 * (ty * gx + tx)], r1 = ranges[2 * (ty * gx + tx) + 1];
            for (int ly = 0; ly < GSRO_BLOCK_Y; ly++)
                for (int lx = 0; lx < GSRO_BLOCK_X; lx++) {
                    const int px = tx * GSRO_BLOCK_X + lx, py = ty * GSRO_BLOCK_Y + ly;
                    if (!(px < W && py < H)) continue;
                    const float pixf[2] = {(float)px, (float)py};
                    float T = 1.0f, mc = 1e30f, md = 1e30f;
                    for (uint32_t k = r0; k < r1; k++) {
                        const uint32_t id = point_list[k];
                        const float dx = means2D[2 * id] - pixf[0], dy = means2D[2 * id + 1] - pixf[1];
                        const float* co = conic_opacity + 4 * id;
                        const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                        /* power is a sum of terms of size ~|q|: its absolute error scales with them */
                        const float pscale = 0.5f * (fabsf(co[0] * dx * dx) + fabsf(co[2] * dy * dy)) + fabsf(co[1] * dx * dy) + 1e-30f;
                        mc = fminf(mc, fabsf(power) / pscale);
                        if (power > 0.0f) continue;
                        const float araw = co[3] * expf(power);
                        /* d(alpha)/alpha = d(power): the ABSOLUTE error of power matters, and that scales with the size of its terms,
                         * not with power itself (an elongated splat far from the pixel: terms of hundreds cancelling to a power of -5) */
                        mc = fminf(mc, fabsf(araw * 255.0f - 1.0f) / fmaxf(1.0f, pscale));
                        const float alpha = fminf(0.99f, araw);
                        if (alpha < 1.0f / 255.0f) continue;
                        const float test_T = T * (1 - alpha);
                        mc = fminf(mc, fabsf(test_T * 10000.0f - 1.0f) * 0.1f); /* T is a product of many factors: 10x the error budget */
                        if (test_T < 0.0001f) break;
                        md = fminf(md, fabsf(T * 2.0f - 1.0f) * 0.1f);
                        T = test_T;
                    }
                    margin_color[W * py + px] = mc;
                    margin_depth[W * py + px] = fminf(mc, md);
                }
        }
}

void gsro_render_backward(int W, int H, int P, const uint32_t* ranges, const uint32_t* point_list,
                          const float* bg, const float* means2D, const float* conic_opacity,
                          const float* colors, const float* final_T, const uint32_t* n_contrib,
                          const float* dL_dpix, int accum_double, float* dL_dmean2D,
                          float* dL_dconic, float* dL_dopacity, float* dL_dcolor)
{
    const int gx = (W + GSRO_BLOCK_X - 1) / GSRO_BLOCK_X, gy = (H + GSRO_BLOCK_Y - 1) / GSRO_BLOCK_Y;
    /* backward.cu:458-461 */
    const float ddelx_dx = (float)(0.5 * W);
    const float ddely_dy = (float)(0.5 * H);
    double* acc = NULL; /* [P][9]: mean2D.xy, conic.xyw, opacity, colour.rgb */
    /* accum_double bit 0: per-splat sums in double; bit 1 (test instrumentation, not in the reference): ALSO the per-pixel state of
     * backward.cu:470-530 in double — the exact value of the reference's formulas for the float alphas it blended with, against
     * which the fp32 rounding of an implementation (the reference's own included) can be measured */
    const int exact_state = (accum_double & 2) != 0;
    if (accum_double) acc = (double*)calloc((size_t)P * 9, sizeof(double));

#ifdef GSRO_OMP
#define GSRO_ATOMIC _Pragma("omp atomic")
#else
#define GSRO_ATOMIC
#endif
#define ADD(slot, fptr, val)                                          \
    do {                                                              \
        if (acc) { GSRO_ATOMIC acc[(size_t)id * 9 + (slot)] += (double)(val); } \
        else { GSRO_ATOMIC *(fptr) += (float)(val); }                 \
    } while (0)

#ifdef GSRO_OMP
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
#endif
    for (int ty = 0; ty < gy; ty++)
        for (int tx = 0; tx < gx; tx++) {
            const uint32_t r0 = ranges[2 * (ty * gx + tx)], r1 = ranges[2 * (ty * gx + tx) + 1];
            for (int ly = 0; ly < GSRO_BLOCK_Y; ly++)
                for (int lx = 0; lx < GSRO_BLOCK_X; lx++) {
                    const int px = tx * GSRO_BLOCK_X + lx, py = ty * GSRO_BLOCK_Y + ly;
                    if (!(px < W && py < H)) continue;
                    const int pix_id = W * py + px;
                    const float pixf[2] = {(float)px, (float)py};
                    if (!exact_state) {
#define REAL float
                    const REAL T_final = final_T[pix_id];
                    REAL T = T_final;
                    const uint32_t last_contributor = n_contrib[pix_id];
                    REAL accum_rec[3] = {0, 0, 0}, dL_dpixel[3], last_alpha = 0, last_color[3] = {0, 0, 0};
                    for (int i = 0; i < 3; i++) dL_dpixel[i] = dL_dpix[i * H * W + pix_id];
                    /* positions >= last_contributor are skipped (backward.cu:484-486) */
                    uint32_t n = r1 - r0;
                    uint32_t start = last_contributor < n ? last_contributor : n;
                    for (uint32_t pos = start; pos-- > 0;) {
                        const uint32_t id = point_list[r0 + pos];
                        const float dx = means2D[2 * id] - pixf[0], dy = means2D[2 * id + 1] - pixf[1];
                        const float* co = conic_opacity + 4 * id;
                        const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                        if (power > 0.0f) continue;
                        const float G = expf(power);
                        const float alpha = fminf(0.99f, co[3] * G);
                        if (alpha < 1.0f / 255.0f) continue;
                        T = T / ((REAL)1 - alpha);
                        const REAL dchannel_dcolor = alpha * T;
                        REAL dL_dalpha = 0;
                        for (int ch = 0; ch < 3; ch++) {
                            const REAL c = colors[id * 3 + ch];
                            accum_rec[ch] = last_alpha * last_color[ch] + ((REAL)1 - last_alpha) * accum_rec[ch];
                            last_color[ch] = c;
                            const REAL dL_dchannel = dL_dpixel[ch];
                            dL_dalpha += (c - accum_rec[ch]) * dL_dchannel;
                            ADD(6 + ch, &dL_dcolor[id * 3 + ch], dchannel_dcolor * dL_dchannel);
                        }
                        dL_dalpha *= T;
                        last_alpha = alpha;
                        REAL bg_dot_dpixel = 0;
                        for (int i = 0; i < 3; i++) bg_dot_dpixel += bg[i] * dL_dpixel[i];
                        dL_dalpha += (-T_final / ((REAL)1 - alpha)) * bg_dot_dpixel;
                        const REAL dL_dG = co[3] * dL_dalpha;
                        const REAL gdx = (REAL)G * dx, gdy = (REAL)G * dy;
                        const REAL dG_ddelx = -gdx * co[0] - gdy * co[1];
                        const REAL dG_ddely = -gdy * co[2] - gdx * co[1];
                        ADD(0, &dL_dmean2D[3 * id + 0], dL_dG * dG_ddelx * ddelx_dx);
                        ADD(1, &dL_dmean2D[3 * id + 1], dL_dG * dG_ddely * ddely_dy);
                        ADD(2, &dL_dconic[4 * id + 0], (REAL)-0.5 * gdx * dx * dL_dG);
                        ADD(3, &dL_dconic[4 * id + 1], (REAL)-0.5 * gdx * dy * dL_dG);
                        ADD(4, &dL_dconic[4 * id + 3], (REAL)-0.5 * gdy * dy * dL_dG);
                        ADD(5, &dL_dopacity[id], G * dL_dalpha);
                    }
#undef REAL
                    } else { /* the same formulas with the per-pixel state (T, accum_rec, dL/dalpha and what is formed from it) in double */
#define REAL double
                    const REAL T_final = final_T[pix_id];
                    REAL T = T_final;
                    const uint32_t last_contributor = n_contrib[pix_id];
                    REAL accum_rec[3] = {0, 0, 0}, dL_dpixel[3], last_alpha = 0, last_color[3] = {0, 0, 0};
                    for (int i = 0; i < 3; i++) dL_dpixel[i] = dL_dpix[i * H * W + pix_id];
                    /* positions >= last_contributor are skipped (backward.cu:484-486) */
                    uint32_t n = r1 - r0;
                    uint32_t start = last_contributor < n ? last_contributor : n;
                    for (uint32_t pos = start; pos-- > 0;) {
                        const uint32_t id = point_list[r0 + pos];
                        const float dx = means2D[2 * id] - pixf[0], dy = means2D[2 * id + 1] - pixf[1];
                        const float* co = conic_opacity + 4 * id;
                        const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                        if (power > 0.0f) continue;
                        const float G = expf(power);
                        const float alpha = fminf(0.99f, co[3] * G);
                        if (alpha < 1.0f / 255.0f) continue;
                        T = T / ((REAL)1 - alpha);
                        const REAL dchannel_dcolor = alpha * T;
                        REAL dL_dalpha = 0;
                        for (int ch = 0; ch < 3; ch++) {
                            const REAL c = colors[id * 3 + ch];
                            accum_rec[ch] = last_alpha * last_color[ch] + ((REAL)1 - last_alpha) * accum_rec[ch];
                            last_color[ch] = c;
                            const REAL dL_dchannel = dL_dpixel[ch];
                            dL_dalpha += (c - accum_rec[ch]) * dL_dchannel;
                            ADD(6 + ch, &dL_dcolor[id * 3 + ch], dchannel_dcolor * dL_dchannel);
                        }
                        dL_dalpha *= T;
                        last_alpha = alpha;
                        REAL bg_dot_dpixel = 0;
                        for (int i = 0; i < 3; i++) bg_dot_dpixel += bg[i] * dL_dpixel[i];
                        dL_dalpha += (-T_final / ((REAL)1 - alpha)) * bg_dot_dpixel;
                        const REAL dL_dG = co[3] * dL_dalpha;
                        const REAL gdx = (REAL)G * dx, gdy = (REAL)G * dy;
                        const REAL dG_ddelx = -gdx * co[0] - gdy * co[1];
                        const REAL dG_ddely = -gdy * co[2] - gdx * co[1];
                        ADD(0, &dL_dmean2D[3 * id + 0], dL_dG * dG_ddelx * ddelx_dx);
                        ADD(1, &dL_dmean2D[3 * id + 1], dL_dG * dG_ddely * ddely_dy);
                        ADD(2, &dL_dconic[4 * id + 0], (REAL)-0.5 * gdx * dx * dL_dG);
                        ADD(3, &dL_dconic[4 * id + 1], (REAL)-0.5 * gdx * dy * dL_dG);
                        ADD(4, &dL_dconic[4 * id + 3], (REAL)-0.5 * gdy * dy * dL_dG);
                        ADD(5, &dL_dopacity[id], G * dL_dalpha);
                    }
#undef REAL
                    }
                }
        }
#undef ADD
    if (acc) {
        for (int id = 0; id < P; id++) {
            const double* a = acc + (size_t)id * 9;
            dL_dmean2D[3 * id + 0] += (float)a[0];
            dL_dmean2D[3 * id + 1] += (float)a[1];
            dL_dconic[4 * id + 0] += (float)a[2];
            dL_dconic[4 * id + 1] += (float)a[3];
            dL_dconic[4 * id + 3] += (float)a[4];
            dL_dopacity[id] += (float)a[5];
            for (int ch = 0; ch < 3; ch++) dL_dcolor[3 * id + ch] += (float)a[6 + ch];
        }
        free(acc);
    }
}

void gsro_cov2d_backward(int P, const float* means3D, const int* radii, const float* cov3Ds,
                         float h_x, float h_y, float tan_fovx, float tan_fovy,
                         const float* view_matrix, const float* dL_dconics, float* dL_dmeans,
                         float* dL_dcov)
{
#ifdef GSRO_OMP
#pragma omp parallel for schedule(static)
#endif
    for (int idx = 0; idx < P; idx++) {
        if (!(radii[idx] > 0)) continue;
        const float* cov3D = cov3Ds + 6 * idx;
        const float dL_dconic[3] = {dL_dconics[4 * idx], dL_dconics[4 * idx + 1], dL_dconics[4 * idx + 3]};
        cov2d_ctx k;
        cov2d_forward(means3D + 3 * idx, h_x, h_y, tan_fovx, tan_fovy, cov3D, view_matrix, &k);
        const float x_grad_mul = (k.txtz < -k.limx || k.txtz > k.limx) ? 0 : 1;
        const float y_grad_mul = (k.tytz < -k.limy || k.tytz > k.limy) ? 0 : 1;
        const mat3 T = k.T, Vrk = k.Vrk, W = k.W;
        const float* t = k.t;
        float a = k.cov.m[0][0], b = k.cov.m[0][1], c = k.cov.m[1][1];
        float denom = a * c - b * b;
        float dL_da = 0, dL_db = 0, dL_dc = 0;
        float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
        float* o = dL_dcov + 6 * idx;
        if (denom2inv != 0) {
            dL_da = denom2inv * (-c * c * dL_dconic[0] + 2 * b * c * dL_dconic[1] + (denom - a * c) * dL_dconic[2]);
            dL_dc = denom2inv * (-a * a * dL_dconic[2] + 2 * a * b * dL_dconic[1] + (denom - a * c) * dL_dconic[0]);
            dL_db = denom2inv * 2 * (b * c * dL_dconic[0] - (denom + 2 * b * b) * dL_dconic[1] + a * b * dL_dconic[2]);
            o[0] = (T.m[0][0] * T.m[0][0] * dL_da + T.m[0][0] * T.m[1][0] * dL_db + T.m[1][0] * T.m[1][0] * dL_dc);
            o[3] = (T.m[0][1] * T.m[0][1] * dL_da + T.m[0][1] * T.m[1][1] * dL_db + T.m[1][1] * T.m[1][1] * dL_dc);
            o[5] = (T.m[0][2] * T.m[0][2] * dL_da + T.m[0][2] * T.m[1][2] * dL_db + T.m[1][2] * T.m[1][2] * dL_dc);
            o[1] = 2 * T.m[0][0] * T.m[0][1] * dL_da + (T.m[0][0] * T.m[1][1] + T.m[0][1] * T.m[1][0]) * dL_db + 2 * T.m[1][0] * T.m[1][1] * dL_dc;
            o[2] = 2 * T.m[0][0] * T.m[0][2] * dL_da + (T.m[0][0] * T.m[1][2] + T.m[0][2] * T.m[1][0]) * dL_db + 2 * T.m[1][0] * T.m[1][2] * dL_dc;
            o[4] = 2 * T.m[0][2] * T.m[0][1] * dL_da + (T.m[0][1] * T.m[1][2] + T.m[0][2] * T.m[1][1]) * dL_db + 2 * T.m[1][1] * T.m[1][2] * dL_dc;
        } else {
            for (int i = 0; i < 6; i++) o[i] = 0;
        }
        float dL_dT00 = 2 * (T.m[0][0] * Vrk.m[0][0] + T.m[0][1] * Vrk.m[0][1] + T.m[0][2] * Vrk.m[0][2]) * dL_da +
                        (T.m[1][0] * Vrk.m[0][0] + T.m[1][1] * Vrk.m[0][1] + T.m[1][2] * Vrk.m[0][2]) * dL_db;
        float dL_dT01 = 2 * (T.m[0][0] * Vrk.m[1][0] + T.m[0][1] * Vrk.m[1][1] + T.m[0][2] * Vrk.m[1][2]) * dL_da +
                        (T.m[1][0] * Vrk.m[1][0] + T.m[1][1] * Vrk.m[1][1] + T.m[1][2] * Vrk.m[1][2]) * dL_db;
        float dL_dT02 = 2 * (T.m[0][0] * Vrk.m[2][0] + T.m[0][1] * Vrk.m[2][1] + T.m[0][2] * Vrk.m[2][2]) * dL_da +
                        (T.m[1][0] * Vrk.m[2][0] + T.m[1][1] * Vrk.m[2][1] + T.m[1][2] * Vrk.m[2][2]) * dL_db;
        float dL_dT10 = 2 * (T.m[1][0] * Vrk.m[0][0] + T.m[1][1] * Vrk.m[0][1] + T.m[1][2] * Vrk.m[0][2]) * dL_dc +
                        (T.m[0][0] * Vrk.m[0][0] + T.m[0][1] * Vrk.m[0][1] + T.m[0][2] * Vrk.m[0][2]) * dL_db;
        float dL_dT11 = 2 * (T.m[1][0] * Vrk.m[1][0] + T.m[1][1] * Vrk.m[1][1] + T.m[1][2] * Vrk.m[1][2]) * dL_dc +
                        (T.m[0][0] * Vrk.m[1][0] + T.m[0][1] * Vrk.m[1][1] + T.m[0][2] * Vrk.m[1][2]) * dL_db;
        float dL_dT12 = 2 * (T.m[1][0] * Vrk.m[2][0] + T.m[1][1] * Vrk.m[2][1] + T.m[1][2] * Vrk.m[2][2]) * dL_dc +
                        (T.m[0][0] * Vrk.m[2][0] + T.m[0][1] * Vrk.m[2][1] + T.m[0][2] * Vrk.m[2][2]) * dL_db;
        float dL_dJ00 = W.m[0][0] * dL_dT00 + W.m[0][1] * dL_dT01 + W.m[0][2] * dL_dT02;
        float dL_dJ02 = W.m[2][0] * dL_dT00 + W.m[2][1] * dL_dT01 + W.m[2][2] * dL_dT02;
        float dL_dJ11 = W.m[1][0] * dL_dT10 + W.m[1][1] * dL_dT11 + W.m[1][2] * dL_dT12;
        float dL_dJ12 = W.m[2][0] * dL_dT10 + W.m[2][1] * dL_dT11 + W.m[2][2] * dL_dT12;
        float tz = 1.f / t[2];
        float tz2 = tz * tz;
        float tz3 = tz2 * tz;
        float dL_dtx = x_grad_mul * -h_x * tz2 * dL_dJ02;
        float dL_dty = y_grad_mul * -h_y * tz2 * dL_dJ12;
        float dL_dtz = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * t[0]) * tz3 * dL_dJ02 + (2 * h_y * t[1]) * tz3 * dL_dJ12;
        /* transformVec4x3Transpose, auxiliary.h:90-97; overwrite (backward.cu:273) */
        const float* m = view_matrix;
        dL_dmeans[3 * idx + 0] = m[0] * dL_dtx + m[1] * dL_dty + m[2] * dL_dtz;
        dL_dmeans[3 * idx + 1] = m[4] * dL_dtx + m[5] * dL_dty + m[6] * dL_dtz;
        dL_dmeans[3 * idx + 2] = m[8] * dL_dtx + m[9] * dL_dty + m[10] * dL_dtz;
    }
}

/* DGR/cuda_rasterizer/backward.cu:20-139, one colour channel at a time.
 * Writes dL_dsh[k][ch] and returns this channel's contribution to
 * (dRGBdx,dRGBdy,dRGBdz)·dL_dRGB through out3. */
static void sh_backward_channel(int deg, const float* sh, float* dL_dsh, int ch, float x, float y,
                                float z, float dL_dRGB, float dRGBd[3])
{
#define SH(k) sh[3 * (k) + ch]
#define DSH(k) dL_dsh[3 * (k) + ch]
    float dx = 0, dy = 0, dz = 0;
    DSH(0) = SH_C0 * dL_dRGB;
    if (deg > 0) {
        float dRGBdsh1 = -SH_C1 * y, dRGBdsh2 = SH_C1 * z, dRGBdsh3 = -SH_C1 * x;
        DSH(1) = dRGBdsh1 * dL_dRGB;
        DSH(2) = dRGBdsh2 * dL_dRGB;
        DSH(3) = dRGBdsh3 * dL_dRGB;
        dx = -SH_C1 * SH(3);
        dy = -SH_C1 * SH(1);
        dz = SH_C1 * SH(2);
        if (deg > 1) {
            float xx = x * x, yy = y * y, zz = z * z;
            float xy = x * y, yz = y * z, xz = x * z;
            DSH(4) = (SH_C2[0] * xy) * dL_dRGB;
            DSH(5) = (SH_C2[1] * yz) * dL_dRGB;
            DSH(6) = (SH_C2[2] * (2.f * zz - xx - yy)) * dL_dRGB;
            DSH(7) = (SH_C2[3] * xz) * dL_dRGB;
            DSH(8) = (SH_C2[4] * (xx - yy)) * dL_dRGB;
            dx += SH_C2[0] * y * SH(4) + SH_C2[2] * 2.f * -x * SH(6) + SH_C2[3] * z * SH(7) + SH_C2[4] * 2.f * x * SH(8);
            dy += SH_C2[0] * x * SH(4) + SH_C2[1] * z * SH(5) + SH_C2[2] * 2.f * -y * SH(6) + SH_C2[4] * 2.f * -y * SH(8);
            dz += SH_C2[1] * y * SH(5) + SH_C2[2] * 2.f * 2.f * z * SH(6) + SH_C2[3] * x * SH(7);
            if (deg > 2) {
                DSH(9) = (SH_C3[0] * y * (3.f * xx - yy)) * dL_dRGB;
                DSH(10) = (SH_C3[1] * xy * z) * dL_dRGB;
                DSH(11) = (SH_C3[2] * y * (4.f * zz - xx - yy)) * dL_dRGB;
                DSH(12) = (SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy)) * dL_dRGB;
                DSH(13) = (SH_C3[4] * x * (4.f * zz - xx - yy)) * dL_dRGB;
                DSH(14) = (SH_C3[5] * z * (xx - yy)) * dL_dRGB;
                DSH(15) = (SH_C3[6] * x * (xx - 3.f * yy)) * dL_dRGB;
                dx += (SH_C3[0] * SH(9) * 3.f * 2.f * xy + SH_C3[1] * SH(10) * yz +
                       SH_C3[2] * SH(11) * -2.f * xy + SH_C3[3] * SH(12) * -3.f * 2.f * xz +
                       SH_C3[4] * SH(13) * (-3.f * xx + 4.f * zz - yy) + SH_C3[5] * SH(14) * 2.f * xz +
                       SH_C3[6] * SH(15) * 3.f * (xx - yy));
                dy += (SH_C3[0] * SH(9) * 3.f * (xx - yy) + SH_C3[1] * SH(10) * xz +
                       SH_C3[2] * SH(11) * (-3.f * yy + 4.f * zz - xx) + SH_C3[3] * SH(12) * -3.f * 2.f * yz +
                       SH_C3[4] * SH(13) * -2.f * xy + SH_C3[5] * SH(14) * -2.f * yz +
                       SH_C3[6] * SH(15) * -3.f * 2.f * xy);
                dz += (SH_C3[1] * SH(10) * xy + SH_C3[2] * SH(11) * 4.f * 2.f * yz +
                       SH_C3[3] * SH(12) * 3.f * (2.f * zz - xx - yy) + SH_C3[4] * SH(13) * 4.f * 2.f * xz +
                       SH_C3[5] * SH(14) * (xx - yy));
            }
        }
    }
#undef SH
#undef DSH
    dRGBd[0] = dx; dRGBd[1] = dy; dRGBd[2] = dz;
}

/* DGR/cuda_rasterizer/auxiliary.h:107-118 (dnormvdv, float3) */
static void dnormvdv3(const float v[3], const float dv[3], float o[3])
{
    float sum2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
    o[0] = ((+sum2 - v[0] * v[0]) * dv[0] - v[1] * v[0] * dv[1] - v[2] * v[0] * dv[2]) * invsum32;
    o[1] = (-v[0] * v[1] * dv[0] + (sum2 - v[1] * v[1]) * dv[1] - v[2] * v[1] * dv[2]) * invsum32;
    o[2] = (-v[0] * v[2] * dv[0] - v[1] * v[2] * dv[1] + (sum2 - v[2] * v[2]) * dv[2]) * invsum32;
}

/* DGR/cuda_rasterizer/backward.cu:278-341 */
static void cov3d_backward(const float scale[3], float mod, const float rot[4],
                           const float* dL_dcov3D, float* dL_dscale, float* dL_drot)
{
    float r = rot[0], x = rot[1], y = rot[2], z = rot[3];
    mat3 R = quat_to_R(rot);
    mat3 S = m3(1, 0, 0, 0, 1, 0, 0, 0, 1);
    float s[3] = {mod * scale[0], mod * scale[1], mod * scale[2]};
    S.m[0][0] = s[0]; S.m[1][1] = s[1]; S.m[2][2] = s[2];
    mat3 M = m3_mul(S, R);
    mat3 dL_dSigma = m3(dL_dcov3D[0], 0.5f * dL_dcov3D[1], 0.5f * dL_dcov3D[2],
                        0.5f * dL_dcov3D[1], dL_dcov3D[3], 0.5f * dL_dcov3D[4],
                        0.5f * dL_dcov3D[2], 0.5f * dL_dcov3D[4], dL_dcov3D[5]);
    mat3 dL_dM = m3_mul(m3_scale(2.0f, M), dL_dSigma);
    mat3 Rt = m3_t(R);
    mat3 dL_dMt = m3_t(dL_dM);
    dL_dscale[0] = dot3(Rt.m[0], dL_dMt.m[0]);
    dL_dscale[1] = dot3(Rt.m[1], dL_dMt.m[1]);
    dL_dscale[2] = dot3(Rt.m[2], dL_dMt.m[2]);
    for (int k = 0; k < 3; k++) {
        dL_dMt.m[0][k] *= s[0];
        dL_dMt.m[1][k] *= s[1];
        dL_dMt.m[2][k] *= s[2];
    }
#define D(c, rr) dL_dMt.m[c][rr]
    dL_drot[0] = 2 * z * (D(0, 1) - D(1, 0)) + 2 * y * (D(2, 0) - D(0, 2)) + 2 * x * (D(1, 2) - D(2, 1));
    dL_drot[1] = 2 * y * (D(1, 0) + D(0, 1)) + 2 * z * (D(2, 0) + D(0, 2)) + 2 * r * (D(1, 2) - D(2, 1)) - 4 * x * (D(2, 2) + D(1, 1));
    dL_drot[2] = 2 * x * (D(1, 0) + D(0, 1)) + 2 * r * (D(2, 0) - D(0, 2)) + 2 * z * (D(1, 2) + D(2, 1)) - 4 * y * (D(2, 2) + D(0, 0));
    dL_drot[3] = 2 * r * (D(0, 1) - D(1, 0)) + 2 * x * (D(2, 0) + D(0, 2)) + 2 * y * (D(1, 2) + D(2, 1)) - 4 * z * (D(1, 1) + D(0, 0));
#undef D
}

void gsro_preprocess_backward(int P, int D, int M, const float* means, const int* radii,
                              const float* shs, const uint8_t* clamped, const float* scales,
                              const float* rotations, float scale_modifier, const float* proj,
                              const float* campos, const float* dL_dmean2D, float* dL_dmeans,
                              float* dL_dcolor, const float* dL_dcov3D, float* dL_dsh,
                              float* dL_dscale, float* dL_drot)
{
#ifdef GSRO_OMP
#pragma omp parallel for schedule(static)
#endif
    for (int idx = 0; idx < P; idx++) {
        if (!(radii[idx] > 0)) continue;
        const float* m = means + 3 * idx;
        float m_hom[4];
        xform4x4(m, proj, m_hom);
        float m_w = 1.0f / (m_hom[3] + 0.0000001f);
        float mul1 = (proj[0] * m[0] + proj[4] * m[1] + proj[8] * m[2] + proj[12]) * m_w * m_w;
        float mul2 = (proj[1] * m[0] + proj[5] * m[1] + proj[9] * m[2] + proj[13]) * m_w * m_w;
        const float gx = dL_dmean2D[3 * idx], gy = dL_dmean2D[3 * idx + 1];
        float d[3];
        d[0] = (proj[0] * m_w - proj[3] * mul1) * gx + (proj[1] * m_w - proj[3] * mul2) * gy;
        d[1] = (proj[4] * m_w - proj[7] * mul1) * gx + (proj[5] * m_w - proj[7] * mul2) * gy;
        d[2] = (proj[8] * m_w - proj[11] * mul1) * gx + (proj[9] * m_w - proj[11] * mul2) * gy;
        for (int k = 0; k < 3; k++) dL_dmeans[3 * idx + k] += d[k];

        if (shs) { /* backward.cu:390-391 -> :20-139 */
            float dir_orig[3], dir[3];
            view_dir(m, campos, dir_orig, dir);
            float dRGBdx[3], dRGBdy[3], dRGBdz[3], dL_dRGB[3];
            for (int ch = 0; ch < 3; ch++) {
                dL_dRGB[ch] = dL_dcolor[3 * idx + ch] * (clamped[3 * idx + ch] ? 0 : 1);
                float g[3];
                sh_backward_channel(D, shs + (size_t)idx * M * 3, dL_dsh + (size_t)idx * M * 3, ch,
                                    dir[0], dir[1], dir[2], dL_dRGB[ch], g);
                dRGBdx[ch] = g[0]; dRGBdy[ch] = g[1]; dRGBdz[ch] = g[2];
            }
            float dL_ddir[3] = {dot3(dRGBdx, dL_dRGB), dot3(dRGBdy, dL_dRGB), dot3(dRGBdz, dL_dRGB)};
            float dm[3];
            dnormvdv3(dir_orig, dL_ddir, dm);
            for (int k = 0; k < 3; k++) dL_dmeans[3 * idx + k] += dm[k];
        }
        if (scales) /* backward.cu:394-395 */
            cov3d_backward(scales + 3 * idx, scale_modifier, rotations + 4 * idx, dL_dcov3D + 6 * idx,
                           dL_dscale + 3 * idx, dL_drot + 4 * idx);
    }
}

/* ---- whole-pipeline drivers ------------------------------------------------------ */
struct gsro_state {
    int P, W, H;
    size_t R;
    float *means2D, *depths, *cov3D, *conic_opacity, *rgb;
    uint8_t* clamped;
    uint32_t *tiles_touched, *point_offsets;
    uint64_t *keys_unsorted, *keys_sorted;
    uint32_t *values_unsorted, *point_list;
    uint32_t* ranges;
    float* final_T;
    uint32_t* n_contrib;
};

gsro_state* gsro_state_new(void) { return (gsro_state*)calloc(1, sizeof(gsro_state)); }

static void state_release(gsro_state* s)
{
    free(s->means2D); free(s->depths); free(s->cov3D); free(s->conic_opacity); free(s->rgb);
    free(s->clamped); free(s->tiles_touched); free(s->point_offsets); free(s->keys_unsorted);
    free(s->keys_sorted); free(s->values_unsorted); free(s->point_list); free(s->ranges);
    free(s->final_T); free(s->n_contrib);
    memset(s, 0, sizeof(*s));
}
void gsro_state_free(gsro_state* s) { if (s) { state_release(s); free(s); } }

int gsro_forward(gsro_state* s, const gsro_scene* a, float* out_color, float* out_depth, int* radii)
{
    state_release(s);
    const int P = a->P, W = a->W, H = a->H;
    const size_t N = (size_t)W * H, Pz = P > 0 ? (size_t)P : 1;
    const size_t tiles = (size_t)((W + 15) / 16) * ((H + 15) / 16);
    s->P = P; s->W = W; s->H = H;
    /* zero-filled blobs: src/Rasterizer.cu:127-134 */
    s->means2D = (float*)calloc(Pz * 2, 4); s->depths = (float*)calloc(Pz, 4);
    s->cov3D = (float*)calloc(Pz * 6, 4); s->conic_opacity = (float*)calloc(Pz * 4, 4);
    s->rgb = (float*)calloc(Pz * 3, 4); s->clamped = (uint8_t*)calloc(Pz * 3, 1);
    s->tiles_touched = (uint32_t*)calloc(Pz, 4); s->point_offsets = (uint32_t*)calloc(Pz, 4);
    s->ranges = (uint32_t*)calloc(tiles * 2 + 2, 4);
    s->final_T = (float*)calloc(N + 1, 4); s->n_contrib = (uint32_t*)calloc(N + 1, 4);
    /* outputs pre-filled with 0: src/Rasterizer.cu:170-172 */
    memset(out_color, 0, N * 3 * 4);
    memset(out_depth, 0, N * 4);
    if (P > 0) memset(radii, 0, (size_t)P * 4);
    if (P == 0) return 0; /* src/Rasterizer.cu:183 */

    gsro_preprocess(P, a->D, a->M, a->means3D, a->scales, a->scale_modifier, a->rotations,
                    a->opacities, a->shs, a->cov3D_precomp, a->colors_precomp, a->viewmatrix,
                    a->projmatrix, a->cam_pos, W, H, a->tan_fovx, a->tan_fovy, radii, s->means2D,
                    s->depths, s->cov3D, s->rgb, s->clamped, s->conic_opacity, s->tiles_touched);
    gsro_inclusive_sum(P, s->tiles_touched, s->point_offsets);
    const size_t R = s->point_offsets[P - 1];
    s->R = R;
    const size_t Rz = R ? R : 1;
    s->keys_unsorted = (uint64_t*)calloc(Rz, 8); s->keys_sorted = (uint64_t*)calloc(Rz, 8);
    s->values_unsorted = (uint32_t*)calloc(Rz, 4); s->point_list = (uint32_t*)calloc(Rz, 4);
    gsro_duplicate_with_keys(P, s->means2D, s->depths, s->point_offsets, radii, W, H,
                             s->keys_unsorted, s->values_unsorted);
    const int bit = (int)gsro_higher_msb((uint32_t)tiles);
    gsro_sort_pairs(R, s->keys_unsorted, s->values_unsorted, s->keys_sorted, s->point_list, 32 + bit);
    if (R > 0) gsro_identify_tile_ranges(R, s->keys_sorted, s->ranges);
    const float* feat = a->colors_precomp ? a->colors_precomp : s->rgb;
    gsro_render_forward(W, H, s->ranges, s->point_list, s->means2D, feat, s->conic_opacity,
                        s->depths, a->background, s->final_T, s->n_contrib, out_color, out_depth);
    return (int)R;
}

void gsro_backward(gsro_state* s, const gsro_scene* a, const int* radii, const float* dL_dpix,
                   int accum_double, float* dL_dmean2D, float* dL_dconic, float* dL_dopacity,
                   float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh,
                   float* dL_dscale, float* dL_drot)
{
    const int P = a->P, W = a->W, H = a->H;
    if (P == 0) return; /* src/Rasterizer.cu:263 */
    const float focal_y = H / (2.0f * a->tan_fovy);
    const float focal_x = W / (2.0f * a->tan_fovx);
    const float* color_ptr = a->colors_precomp ? a->colors_precomp : s->rgb;
    gsro_render_backward(W, H, P, s->ranges, s->point_list, a->background, s->means2D,
                         s->conic_opacity, color_ptr, s->final_T, s->n_contrib, dL_dpix,
                         accum_double, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor);
    const float* cov3D_ptr = a->cov3D_precomp ? a->cov3D_precomp : s->cov3D;
    gsro_cov2d_backward(P, a->means3D, radii, cov3D_ptr, focal_x, focal_y, a->tan_fovx, a->tan_fovy,
                        a->viewmatrix, dL_dconic, dL_dmean3D, dL_dcov3D);
    gsro_preprocess_backward(P, a->D, a->M, a->means3D, radii, a->shs, s->clamped, a->scales,
                             a->rotations, a->scale_modifier, a->projmatrix, a->cam_pos, dL_dmean2D,
                             dL_dmean3D, dL_dcolor, dL_dcov3D, dL_dsh, dL_dscale, dL_drot);
}

const void* gsro_stage(const gsro_state* s, int which, size_t* count)
{
    const size_t P = (size_t)s->P, N = (size_t)s->W * s->H;
    const size_t tiles = (size_t)((s->W + 15) / 16) * ((s->H + 15) / 16);
    size_t c = 0;
    const void* p = NULL;
    switch (which) {
    case GSRO_MEANS2D: p = s->means2D; c = P * 2; break;
    case GSRO_DEPTHS: p = s->depths; c = P; break;
    case GSRO_COV3D: p = s->cov3D; c = P * 6; break;
    case GSRO_CONIC_OPACITY: p = s->conic_opacity; c = P * 4; break;
    case GSRO_RGB: p = s->rgb; c = P * 3; break;
    case GSRO_CLAMPED: p = s->clamped; c = P * 3; break;
    case GSRO_TILES_TOUCHED: p = s->tiles_touched; c = P; break;
    case GSRO_POINT_OFFSETS: p = s->point_offsets; c = P; break;
    case GSRO_KEYS_UNSORTED: p = s->keys_unsorted; c = s->R; break;
    case GSRO_VALUES_UNSORTED: p = s->values_unsorted; c = s->R; break;
    case GSRO_KEYS_SORTED: p = s->keys_sorted; c = s->R; break;
    case GSRO_POINT_LIST: p = s->point_list; c = s->R; break;
    case GSRO_RANGES: p = s->ranges; c = tiles * 2; break;
    case GSRO_FINAL_T: p = s->final_T; c = N; break;
    case GSRO_N_CONTRIB: p = s->n_contrib; c = N; break;
    default: break;
    }
    if (count) *count = c;
    return p;
}


/* src/simple_knn.cu:131-183 reduced to its definition: for every point the three smallest squared
 * distances to the other points (self excluded by index), kept ascending like updateKBest does, and
 * their mean (best[0] + best[1] + best[2]) / 3.0f. Brute force O(P^2): the search structure of the
 * reference (Morton boxes) only prunes, it does not change the result. */
void gsro_dist2(int P, const float* pts, float* dists)
{
#ifdef GSRO_OMP
#pragma omp parallel for schedule(static)
#endif
    for (int i = 0; i < P; i++) {
        float best[3] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f};
        const float* p = pts + 3 * (size_t)i;
        for (int j = 0; j < P; j++) {
            if (j == i) continue;
            const float* q = pts + 3 * (size_t)j;
            const float dx = q[0] - p[0], dy = q[1] - p[1], dz = q[2] - p[2];
            float dist = dx * dx + dy * dy + dz * dz;
            for (int k = 0; k < 3; k++)
                if (best[k] > dist) { float t = best[k]; best[k] = dist; dist = t; }
        }
        dists[i] = (best[0] + best[1] + best[2]) / 3.0f;
    }
}
