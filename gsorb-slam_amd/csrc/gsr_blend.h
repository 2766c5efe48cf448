// gsr_blend.h — the two blend kernels (forward alpha compositing and its backward).
//
// Mapping (wave64-first): ONE wave per 8x8 pixel quad, one pixel per lane; the four quads of
// a 16x16 tile are four INDEPENDENT single-wave workgroups that walk the same per-tile list.
//   * a workgroup is a single wave: no s_barrier anywhere; LDS is only a broadcast buffer
//     the wave writes and reads in program order;
//   * 4x more, 4x shorter waves than one-wave-per-tile: ~12 waves per SIMD keep the VALU
//     issuing (a wave issues at most one instruction per 4 cycles) and even out the per-tile
//     load imbalance;
//   * block ids are remapped so the four quads of a tile (and neighbouring tiles) run on
//     the same XCD and share its L2 for the per-splat gathers.
// Each batch of 64 list entries is gathered straight into registers (one entry per lane),
// culled by its owning lane with an EXACT test (does the iso-alpha ellipse alpha = 1/255
// intersect the quad's rectangle of pixel centres?), and only the surviving entries are
// parked in LDS and visited, in list order, through wave-uniform LDS broadcast reads.
// Colours are gathered only for surviving entries.
//
// Backward: every lane forms its nine partial sums for the splat, a transposing wave
// reduction (gsr_device.h: reduce9, 29 VALU ops using v_permlane{32,16}_swap) leaves each of
// the nine totals in its own lane, and those nine lanes issue ONE global atomic instruction
// that lands in the splat's 48-byte accumulator record (a single cache line).
//
// What is computed per (pixel, splat) pair is the reference's arithmetic
// (DGR/cuda_rasterizer/forward.cu:339-391, backward.cu:470-555).
#pragma once

#include "gsr_device.h"

namespace gsr {

#define GSR_ALPHA_MIN (1.0f / 255.0f)

// Can the splat reach alpha >= 1/255 on some pixel centre of the quad whose pixel centres
// span [X0, X0+7] x [Y0, Y0+7]? alpha >= 1/255  <=>  Q(d) := 0.5*(a dx^2 + c dy^2) + b dx dy
// <= ln(255*opacity), with d = splat centre - pixel. The minimum of the convex Q over the
// rectangle is attained either at d = 0 (centre inside) or on a side facing the centre,
// where it is a clamped 1-D parabola. Conservative: the continuous rectangle contains the
// pixel centres, the threshold carries a margin far above fp32 rounding, NaNs pass.
__device__ __forceinline__ bool quad_reach(const float4 a, const float4 b, float X0, float Y0)
{
    const float op = b.y;
    if (op < GSR_ALPHA_MIN) return false; // alpha = op*exp(power<=0) can never reach 1/255
    const float ca = a.z, cb = a.w, cc = b.x;
    if (!(ca > 0.f) || !(cc > 0.f)) return true;
    const float tau = __logf(255.0f * op) + 0.01f;
    const float dxl = a.x - (X0 + 7.f), dxh = a.x - X0, dyl = a.y - (Y0 + 7.f), dyh = a.y - Y0;
    const float dxc = fminf(fmaxf(0.f, dxl), dxh), dyc = fminf(fmaxf(0.f, dyl), dyh); // point of the range closest to 0
    if (dxc == 0.f && dyc == 0.f) return true;
    float q = 3.0e38f;
    if (dxc != 0.f) { // a vertical side faces the centre: minimise over dy
        const float dy = fminf(fmaxf(-cb * dxc / cc, dyl), dyh);
        q = fminf(q, 0.5f * (ca * dxc * dxc + cc * dy * dy) + cb * dxc * dy);
    }
    if (dyc != 0.f) {
        const float dx = fminf(fmaxf(-cb * dyc / ca, dxl), dxh);
        q = fminf(q, 0.5f * (ca * dx * dx + cc * dyc * dyc) + cb * dx * dyc);
    }
    return !(q > tau);
}

// Parks a surviving entry in LDS with its conic pre-multiplied by log2(e): the hit loop then forms
// power*log2(e) directly and uses the hardware exp2 (one multiply less per pixel-splat pair). Forward
// and backward stage identically, so both evaluate bit-identical alphas.
__device__ __forceinline__ void stage_entry(float4* sA, float4* sB, int lane, float4 a, float4 b)
{
    const float kLog2e = 1.4426950408889634f;
    a.z *= kLog2e; a.w *= kLog2e; b.x *= kLog2e;
    sA[lane] = a; sB[lane] = b;
}

__global__ void __launch_bounds__(64)
K_blend_fwd(ImageView im, BinView bn, GeomView g, const float* __restrict__ bg, int W, int H,
            int grid_x, int ntiles, int tile0, float* __restrict__ out_color, float* __restrict__ out_depth)
{
    __shared__ float4 sA[64], sB[64], sC[64];
    const uint32_t w = xcd_remap(blockIdx.x, 4u * (uint32_t)ntiles);
    const uint32_t tile = (uint32_t)tile0 + (w >> 2), quad = w & 3u; // tile0: first tile of the band
    const int tx = tile % grid_x, ty = tile / grid_x;
    const int lane = threadIdx.x;
    const int X0 = tx * 16 + (int)(quad & 1u) * 8, Y0 = ty * 16 + (int)(quad >> 1) * 8;
    const int px = X0 + (lane & 7), py = Y0 + (lane >> 3);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py, X0f = (float)X0, Y0f = (float)Y0;
    const uint2 range = im.ranges[tile];
    const int n = g.hdr->overflow ? 0 : (int)(range.y - range.x);
    const uint32_t* __restrict__ plist = bn.point_list + range.x;

    float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f, Dp = 0.f;
    uint32_t last = 0u;
    bool done = !inside;

    uint32_t id_next = lane < n ? plist[lane] : 0u;
    for (int base = 0; base < n; base += 64) {
        if (__all(done)) break;
        const uint32_t id = id_next;
        const bool have = base + lane < n;
        if (base + 64 + lane < n) id_next = plist[base + 64 + lane];
        bool hit = false;
        if (have) {
            const float4 a = g.g0[id];
            const float4 b = g.g1[id];
            hit = quad_reach(a, b, X0f, Y0f);
            if (hit) { stage_entry(sA, sB, lane, a, b); sC[lane] = g.col[id]; }
        }
        __builtin_amdgcn_wave_barrier();
        unsigned long long hits = __ballot(hit);
        while (hits) {
            const int jj = (int)__builtin_ctzll(hits);
            hits &= hits - 1;
            const float4 A = sA[jj], B = sB[jj], Cc = sC[jj];
            const float dx = A.x - pxf, dy = A.y - pyf;
            const float power2 = pair_power(dx, dy, A.z, A.w, B.x); // = power * log2(e): same sign as power
            const float alpha = fminf(0.99f, B.y * __builtin_amdgcn_exp2f(power2));
            const bool valid = !done && power2 <= 0.0f && alpha >= GSR_ALPHA_MIN;
            const float test_T = T * (1.f - alpha);
            const bool stop = valid && test_T < 0.0001f;
            const bool upd = valid && !stop;
            done = done || stop;
            const float wgt = upd ? alpha * T : 0.f;
            C0 = fmaf(Cc.x, wgt, C0);
            C1 = fmaf(Cc.y, wgt, C1);
            C2 = fmaf(Cc.z, wgt, C2);
            Dp = (upd && T > 0.5f) ? B.z : Dp; // median depth (forward.cu:374-379)
            T = upd ? test_T : T;
            last = upd ? (uint32_t)(base + jj + 1) : last;
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (inside) {
        const size_t pix = (size_t)py * W + px, HW = (size_t)H * W;
        im.final_T[pix] = T;
        im.n_contrib[pix] = last;
        out_color[pix] = C0 + T * bg[0];
        out_color[HW + pix] = C1 + T * bg[1];
        out_color[2 * HW + pix] = C2 + T * bg[2];
        out_depth[pix] = Dp;
    }
}

__global__ void __launch_bounds__(64)
K_blend_bwd(ImageView im, BinView bn, GeomView g, const float* __restrict__ bg, int W, int H,
            int grid_x, int ntiles, int tile0, const float* __restrict__ dL_dpix)
{
    __shared__ float4 sA[64], sB[64], sC[64];
    const uint32_t w = xcd_remap(blockIdx.x, 4u * (uint32_t)ntiles);
    const uint32_t tile = (uint32_t)tile0 + (w >> 2), quad = w & 3u;
    const int tx = tile % grid_x, ty = tile / grid_x;
    const int lane = threadIdx.x;
    const int X0 = tx * 16 + (int)(quad & 1u) * 8, Y0 = ty * 16 + (int)(quad >> 1) * 8;
    const int px = X0 + (lane & 7), py = Y0 + (lane >> 3);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py, X0f = (float)X0, Y0f = (float)Y0;
    const uint2 range = im.ranges[tile];
    const int n = g.hdr->overflow ? 0 : (int)(range.y - range.x);
    const uint32_t* __restrict__ plist = bn.point_list + range.x;
    const size_t pix = (size_t)py * W + px, HW = (size_t)H * W;

    const float T_final = inside ? im.final_T[pix] : 0.f;
    float T = T_final;
    const uint32_t last = inside ? im.n_contrib[pix] : 0u;
    const float g0 = inside ? dL_dpix[pix] : 0.f, g1 = inside ? dL_dpix[HW + pix] : 0.f,
                g2 = inside ? dL_dpix[2 * HW + pix] : 0.f;
    const float bg_dot = bg[0] * g0 + bg[1] * g1 + bg[2] * g2;
    float S0 = 0.f, S1 = 0.f, S2 = 0.f;
    const int slot = reduce9_slot_of(lane); // which of the nine sums this lane commits (-1: none)

    const int ntodo = min(n, (int)wave_max_u32(last)); // nothing at list position >= this touches the quad

    // back to front: list position of batch entry k is ntodo-1-k
    uint32_t id_next = lane < ntodo ? plist[ntodo - 1 - lane] : 0u;
    for (int base = 0; base < ntodo; base += 64) {
        const uint32_t id = id_next;
        const int k = base + lane;
        const bool have = k < ntodo;
        if (k + 64 < ntodo) id_next = plist[ntodo - 1 - (k + 64)];
        bool hit = false;
        if (have) {
            const float4 a = g.g0[id];
            const float4 b = g.g1[id];
            hit = quad_reach(a, b, X0f, Y0f);
            if (hit) {
                float4 c = g.col[id];
                c.w = __uint_as_float(id);
                stage_entry(sA, sB, lane, a, b);
                sC[lane] = c;
            }
        }
        __builtin_amdgcn_wave_barrier();
        unsigned long long hits = __ballot(hit);
        while (hits) {
            const int jj = (int)__builtin_ctzll(hits);
            hits &= hits - 1;
            const uint32_t pos = (uint32_t)(ntodo - 1 - (base + jj));
            const float4 A = sA[jj], B = sB[jj], Cc = sC[jj];
            const float dx = A.x - pxf, dy = A.y - pyf;
            const float power2 = pair_power(dx, dy, A.z, A.w, B.x); // = power * log2(e)
            const float Graw = __builtin_amdgcn_exp2f(power2);
            const float araw = fminf(0.99f, B.y * Graw);
            const bool valid = pos < last && power2 <= 0.0f && araw >= GSR_ALPHA_MIN;
            if (!__any(valid)) continue;
            const float alpha = valid ? araw : 0.f, G = valid ? Graw : 0.f;
            const float ia = __builtin_amdgcn_rcpf(1.f - alpha);
            T = T * ia; // T <- T / (1 - alpha); unchanged where the pair did not blend (alpha = 0)
            const float dcol = alpha * T;
            // S = colour accumulated behind this splat (the reference's accum_rec, updated eagerly:
            // last_alpha*last_color + (1-last_alpha)*accum_rec == fma(alpha, c - S, S) one step later)
            const float e0 = Cc.x - S0, e1 = Cc.y - S1, e2 = Cc.z - S2;
            float dL_dalpha = (e0 * g0 + e1 * g1 + e2 * g2) * T;
            dL_dalpha = fmaf(-T_final * ia, bg_dot, dL_dalpha);
            S0 = fmaf(alpha, e0, S0); S1 = fmaf(alpha, e1, S1); S2 = fmaf(alpha, e2, S2);
            // raw moments of u = G*dL/dalpha; conic, opacity and the NDC scale are applied per splat
            const float u = G * dL_dalpha, udx = u * dx, udy = u * dy;
            float v[9];
            v[0] = u;
            v[1] = udx;
            v[2] = udy;
            v[3] = udx * dx;
            v[4] = udx * dy;
            v[5] = udy * dy;
            v[6] = dcol * g0;
            v[7] = dcol * g1;
            v[8] = dcol * g2;
            const float mine = reduce9(v, lane);
            const uint32_t sid = (uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(Cc.w));
            if (slot >= 0) unsafeAtomicAdd(&g.acc[(size_t)sid * GSR_ACC_STRIDE + slot], mine);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

} // namespace gsr
