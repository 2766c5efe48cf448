// gsr_blend.h — the two blend kernels (forward alpha compositing and its backward).
//
// Mapping (wave64-first): ONE wave per 16x16 tile, FOUR pixels per lane — lane l owns pixel
// (l&7, l>>3) of each of the tile's four 8x8 quads. Consequences:
//   * the per-quad culling decision of a splat is wave-uniform: it is a scalar branch
//     (s_bitcmp + s_cbranch), not an exec-mask divergence;
//   * a workgroup is a single wave: no s_barrier anywhere, LDS is only a broadcast buffer;
//   * every lane carries four independent pixel chains (ILP hides exp / LDS latency);
//   * the backward needs ONE cross-lane reduction per (tile, splat) — lanes first add their
//     four pixels' terms — instead of one per (quad, splat).
// Each batch of 64 list entries is gathered straight into registers (one entry per lane,
// three 16-byte loads), culled by its owning lane (conservative iso-alpha bounding box of
// the splat vs the four quads -> 4-bit mask), parked in LDS, and then only the entries with
// a non-empty mask are visited, in list order, through wave-uniform LDS broadcast reads.
//
// What is computed per (pixel, splat) pair is the reference's arithmetic
// (DGR/cuda_rasterizer/forward.cu:339-391, backward.cu:470-555).
#pragma once

#include "gsr_device.h"

namespace gsr {

#define GSR_ALPHA_MIN (1.0f / 255.0f)

// 4-bit mask of the tile's quads on which the splat can reach alpha >= 1/255 at some pixel
// centre. Conservative (axis-aligned box of the ellipse 0.5 d^T Conic d <= ln(255 op), plus
// margins far above fp32 rounding); degenerate conics and NaNs select every quad.
__device__ __forceinline__ uint32_t quad_mask(const float4 a, const float4 b, float tx0, float ty0)
{
    const float op = b.y;
    if (op < GSR_ALPHA_MIN) return 0u; // alpha = op*exp(power<=0) can never reach 1/255
    const float ca = a.z, cb = a.w, cc = b.x;
    const float det = ca * cc - cb * cb;
    if (!(det > 0.f)) return 0xFu;
    const float inv = 2.0f * (__logf(255.0f * op) + 0.01f) / det;
    const float hx = sqrtf(inv * cc) + 0.01f, hy = sqrtf(inv * ca) + 0.01f;
    const float xl = a.x - hx - tx0, xh = a.x + hx - tx0, yl = a.y - hy - ty0, yh = a.y + hy - ty0;
    const bool cx0 = !(xl > 7.f) && !(xh < 0.f), cx1 = !(xl > 15.f) && !(xh < 8.f);
    const bool cy0 = !(yl > 7.f) && !(yh < 0.f), cy1 = !(yl > 15.f) && !(yh < 8.f);
    return (cx0 && cy0 ? 1u : 0u) | (cx1 && cy0 ? 2u : 0u) | (cx0 && cy1 ? 4u : 0u) | (cx1 && cy1 ? 8u : 0u);
}

struct FwdPix {
    float T, C0, C1, C2, D;
    uint32_t last;
    bool done;
};

__device__ __forceinline__ void fwd_pixel(FwdPix& p, float pxf, float pyf, const float4 A, const float4 B,
                                          const float4 Cc, uint32_t pos)
{
    const float dx = A.x - pxf, dy = A.y - pyf;
    const float power = pair_power(dx, dy, A.z, A.w, B.x);
    const float alpha = fminf(0.99f, B.y * __expf(power));
    const bool valid = !p.done && power <= 0.0f && alpha >= GSR_ALPHA_MIN;
    const float test_T = p.T * (1.f - alpha);
    const bool stop = valid && test_T < 0.0001f;
    const bool upd = valid && !stop;
    p.done = p.done || stop;
    const float w = upd ? alpha * p.T : 0.f;
    p.C0 = fmaf(Cc.x, w, p.C0);
    p.C1 = fmaf(Cc.y, w, p.C1);
    p.C2 = fmaf(Cc.z, w, p.C2);
    p.D = (upd && p.T > 0.5f) ? B.z : p.D; // median depth (forward.cu:374-379)
    p.T = upd ? test_T : p.T;
    p.last = upd ? pos : p.last;
}

__global__ void __launch_bounds__(64)
K_blend_fwd(ImageView im, BinView bn, GeomView g, const float* __restrict__ bg, int W, int H,
            int grid_x, int ntiles, float* __restrict__ out_color, float* __restrict__ out_depth)
{
    __shared__ float4 s0[64], s1[64], s2[64];
    const uint32_t tile = xcd_remap(blockIdx.x, ntiles);
    const int tx = tile % grid_x, ty = tile / grid_x;
    const int lane = threadIdx.x;
    const int bx = tx * 16 + (lane & 7), by = ty * 16 + (lane >> 3);
    const float tx0 = (float)(tx * 16), ty0 = (float)(ty * 16);
    const uint2 range = im.ranges[tile];
    const int n = g.hdr->overflow ? 0 : (int)(range.y - range.x);
    const uint32_t* __restrict__ plist = bn.point_list + range.x;

    FwdPix p[4];
    float pxf[4], pyf[4];
    bool inside[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int px = bx + (q & 1) * 8, py = by + (q >> 1) * 8;
        inside[q] = px < W && py < H;
        pxf[q] = (float)px; pyf[q] = (float)py;
        p[q].T = 1.0f; p[q].C0 = p[q].C1 = p[q].C2 = 0.f; p[q].D = 0.f; p[q].last = 0u; p[q].done = !inside[q];
    }

    uint32_t id_next = lane < n ? plist[lane] : 0u;
    for (int base = 0; base < n; base += 64) {
        // quads whose 64 pixels are all finished take no further splats
        const uint32_t alive = (__all(p[0].done) ? 0u : 1u) | (__all(p[1].done) ? 0u : 2u) |
                               (__all(p[2].done) ? 0u : 4u) | (__all(p[3].done) ? 0u : 8u);
        if (alive == 0u) break;
        const uint32_t id = id_next;
        const bool have = base + lane < n;
        if (base + 64 + lane < n) id_next = plist[base + 64 + lane];
        uint32_t qm = 0u;
        if (have) {
            const float4 a = g.g0[id];
            float4 b = g.g1[id];
            const float4 c = g.col[id];
            qm = quad_mask(a, b, tx0, ty0) & alive;
            b.w = __uint_as_float(qm);
            s0[lane] = a; s1[lane] = b; s2[lane] = c;
        }
        __builtin_amdgcn_wave_barrier();
        unsigned long long hits = __ballot(qm != 0u);
        while (hits) {
            const int jj = (int)__builtin_ctzll(hits);
            hits &= hits - 1;
            const float4 A = s0[jj], B = s1[jj], Cc = s2[jj];
            const uint32_t m = (uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(B.w));
            const uint32_t pos = (uint32_t)(base + jj + 1);
            if (m & 1u) fwd_pixel(p[0], pxf[0], pyf[0], A, B, Cc, pos);
            if (m & 2u) fwd_pixel(p[1], pxf[1], pyf[1], A, B, Cc, pos);
            if (m & 4u) fwd_pixel(p[2], pxf[2], pyf[2], A, B, Cc, pos);
            if (m & 8u) fwd_pixel(p[3], pxf[3], pyf[3], A, B, Cc, pos);
        }
        __builtin_amdgcn_wave_barrier();
    }
    const float b0 = bg[0], b1 = bg[1], b2 = bg[2];
    const size_t HW = (size_t)H * W;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        if (!inside[q]) continue;
        const size_t pix = (size_t)(by + (q >> 1) * 8) * W + (bx + (q & 1) * 8);
        im.final_T[pix] = p[q].T;
        im.n_contrib[pix] = p[q].last;
        out_color[pix] = p[q].C0 + p[q].T * b0;
        out_color[HW + pix] = p[q].C1 + p[q].T * b1;
        out_color[2 * HW + pix] = p[q].C2 + p[q].T * b2;
        out_depth[pix] = p[q].D;
    }
}

struct BwdPix {
    float T, T_final, ar0, ar1, ar2, la, lc0, lc1, lc2, g0, g1, g2;
    uint32_t last;
};

// accumulates this pixel's nine gradient terms of the splat into v[]; returns whether the pair blended
__device__ __forceinline__ bool bwd_pixel(BwdPix& p, float pxf, float pyf, const float4 A, const float4 B,
                                          const float4 Cc, uint32_t pos, float bg_dot, float (&v)[9])
{
    const float dx = A.x - pxf, dy = A.y - pyf;
    const float power = pair_power(dx, dy, A.z, A.w, B.x);
    const float Graw = __expf(power);
    const float araw = fminf(0.99f, B.y * Graw);
    const bool valid = pos < p.last && power <= 0.0f && araw >= GSR_ALPHA_MIN;
    const float alpha = valid ? araw : 0.f, G = valid ? Graw : 0.f;
    const float ia = __builtin_amdgcn_rcpf(1.f - alpha);
    const float Tn = p.T * ia; // T <- T / (1 - alpha)
    const float dcol = alpha * Tn;
    const float n0 = fmaf(p.la, p.lc0 - p.ar0, p.ar0); // last_alpha*last_color + (1-last_alpha)*accum_rec
    const float n1 = fmaf(p.la, p.lc1 - p.ar1, p.ar1);
    const float n2 = fmaf(p.la, p.lc2 - p.ar2, p.ar2);
    float dL_dalpha = ((Cc.x - n0) * p.g0 + (Cc.y - n1) * p.g1 + (Cc.z - n2) * p.g2) * Tn;
    dL_dalpha += (-p.T_final * ia) * bg_dot;
    v[6] = fmaf(dcol, p.g0, v[6]);
    v[7] = fmaf(dcol, p.g1, v[7]);
    v[8] = fmaf(dcol, p.g2, v[8]);
    const float dL_dG = B.y * dL_dalpha;
    const float gdx = G * dx, gdy = G * dy; // zero when the pair did not blend
    v[0] = fmaf(dL_dG, -gdx * A.z - gdy * A.w, v[0]); // scaled by 0.5*W in K_splat_bwd
    v[1] = fmaf(dL_dG, -gdy * B.x - gdx * A.w, v[1]); // scaled by 0.5*H in K_splat_bwd
    const float h = -0.5f * dL_dG;
    v[2] = fmaf(h * gdx, dx, v[2]);
    v[3] = fmaf(h * gdx, dy, v[3]);
    v[4] = fmaf(h * gdy, dy, v[4]);
    v[5] = fmaf(G, dL_dalpha, v[5]);
    p.T = Tn;
    p.ar0 = valid ? n0 : p.ar0; p.ar1 = valid ? n1 : p.ar1; p.ar2 = valid ? n2 : p.ar2;
    p.lc0 = valid ? Cc.x : p.lc0; p.lc1 = valid ? Cc.y : p.lc1; p.lc2 = valid ? Cc.z : p.lc2;
    p.la = valid ? alpha : p.la;
    return valid;
}

__global__ void __launch_bounds__(64)
K_blend_bwd(ImageView im, BinView bn, GeomView g, const float* __restrict__ bg, int W, int H,
            int grid_x, int ntiles, const float* __restrict__ dL_dpix)
{
    __shared__ float4 s0[64], s1[64], s2[64];
    const uint32_t tile = xcd_remap(blockIdx.x, ntiles);
    const int tx = tile % grid_x, ty = tile / grid_x;
    const int lane = threadIdx.x;
    const int bx = tx * 16 + (lane & 7), by = ty * 16 + (lane >> 3);
    const float tx0 = (float)(tx * 16), ty0 = (float)(ty * 16);
    const uint2 range = im.ranges[tile];
    const int n = g.hdr->overflow ? 0 : (int)(range.y - range.x);
    const uint32_t* __restrict__ plist = bn.point_list + range.x;
    const size_t HW = (size_t)H * W;
    const float b0 = bg[0], b1 = bg[1], b2 = bg[2];

    BwdPix p[4];
    float pxf[4], pyf[4], bgd[4];
    uint32_t qmax[4]; // nothing at list position >= qmax[q] touches quad q
    uint32_t tmax = 0u;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int px = bx + (q & 1) * 8, py = by + (q >> 1) * 8;
        const bool inside = px < W && py < H;
        const size_t pix = (size_t)py * W + px;
        pxf[q] = (float)px; pyf[q] = (float)py;
        p[q].T_final = inside ? im.final_T[pix] : 0.f;
        p[q].T = p[q].T_final;
        p[q].last = inside ? im.n_contrib[pix] : 0u;
        p[q].g0 = inside ? dL_dpix[pix] : 0.f;
        p[q].g1 = inside ? dL_dpix[HW + pix] : 0.f;
        p[q].g2 = inside ? dL_dpix[2 * HW + pix] : 0.f;
        p[q].ar0 = p[q].ar1 = p[q].ar2 = 0.f; p[q].la = 0.f; p[q].lc0 = p[q].lc1 = p[q].lc2 = 0.f;
        bgd[q] = b0 * p[q].g0 + b1 * p[q].g1 + b2 * p[q].g2;
        qmax[q] = wave_max_u32(p[q].last);
        tmax = max(tmax, qmax[q]);
    }
    const int ntodo = min(n, (int)tmax);

    // back to front: list position of batch entry k is ntodo-1-k
    uint32_t id_next = lane < ntodo ? plist[ntodo - 1 - lane] : 0u;
    for (int base = 0; base < ntodo; base += 64) {
        const uint32_t id = id_next;
        const int k = base + lane;
        const bool have = k < ntodo;
        if (k + 64 < ntodo) id_next = plist[ntodo - 1 - (k + 64)];
        uint32_t qm = 0u;
        if (have) {
            const uint32_t pos = (uint32_t)(ntodo - 1 - k);
            const float4 a = g.g0[id];
            float4 b = g.g1[id];
            float4 c = g.col[id];
            const uint32_t reach = (pos < qmax[0] ? 1u : 0u) | (pos < qmax[1] ? 2u : 0u) |
                                   (pos < qmax[2] ? 4u : 0u) | (pos < qmax[3] ? 8u : 0u);
            qm = quad_mask(a, b, tx0, ty0) & reach;
            b.w = __uint_as_float(qm);
            c.w = __uint_as_float(id);
            s0[lane] = a; s1[lane] = b; s2[lane] = c;
        }
        __builtin_amdgcn_wave_barrier();
        unsigned long long hits = __ballot(qm != 0u);
        while (hits) {
            const int jj = (int)__builtin_ctzll(hits);
            hits &= hits - 1;
            const float4 A = s0[jj], B = s1[jj], Cc = s2[jj];
            const uint32_t m = (uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(B.w));
            const uint32_t pos = (uint32_t)(ntodo - 1 - (base + jj));
            float v[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            bool any = false;
            if (m & 1u) any |= bwd_pixel(p[0], pxf[0], pyf[0], A, B, Cc, pos, bgd[0], v);
            if (m & 2u) any |= bwd_pixel(p[1], pxf[1], pyf[1], A, B, Cc, pos, bgd[1], v);
            if (m & 4u) any |= bwd_pixel(p[2], pxf[2], pyf[2], A, B, Cc, pos, bgd[2], v);
            if (m & 8u) any |= bwd_pixel(p[3], pxf[3], pyf[3], A, B, Cc, pos, bgd[3], v);
            if (!__any(any)) continue;
#pragma unroll
            for (int i = 0; i < 9; i++) v[i] = wave_sum_to_lane63(v[i]);
            // lanes 55..63 each take one of the nine sums -> one 9-lane atomic on one cache line
#define GSR_RL63(x) __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63))
            // (readlane must be executed by the whole wave: keep it out of the selects' operands)
            const float t0 = GSR_RL63(v[0]), t1 = GSR_RL63(v[1]), t2 = GSR_RL63(v[2]), t3 = GSR_RL63(v[3]),
                        t4 = GSR_RL63(v[4]), t5 = GSR_RL63(v[5]), t6 = GSR_RL63(v[6]), t7 = GSR_RL63(v[7]);
            float mine = v[8];
            mine = lane == 62 ? t7 : mine;
            mine = lane == 61 ? t6 : mine;
            mine = lane == 60 ? t5 : mine;
            mine = lane == 59 ? t4 : mine;
            mine = lane == 58 ? t3 : mine;
            mine = lane == 57 ? t2 : mine;
            mine = lane == 56 ? t1 : mine;
            mine = lane == 55 ? t0 : mine;
#undef GSR_RL63
            const uint32_t sid = (uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(Cc.w));
            if (lane >= 55) unsafeAtomicAdd(&g.acc[(size_t)sid * GSR_ACC_STRIDE + (lane - 55)], mine);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

} // namespace gsr
