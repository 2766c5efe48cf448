// gsr_shard.h — scheme-B layer compositing (gsorb-slam_amd/sharded.py: LayerCompositor / _CompositeFn) as three elementwise
// kernels. Every rank renders its own Gaussians into a layer (rgb, depth, silhouette S, surface depth); with the layers ordered
// front to back, out = sum_k P_k L_k, P_k = prod_{h before k} (1 - S_h). The exchange is two small collectives forwards (an
// all-gather of (S, surface depth, key row), an all-reduce of the four premultiplied channels) and one backwards (an all-gather
// of g . L); what sits between them was ~15 tensor-library launches per direction and a Python loop over the ranks for the
// surface depth. `gathered` [world,planes,H,W] is the all-gather's output in RANK order (plane 0 the silhouettes, plane 1 the surface depths;
// the Python compositor gathers a third plane that carries the order key), `order` [world] the ranks front to back.
// The reference has no such exchange (it is single-GPU): north_star's "shard Gaussians, all-reduce pose / loss gradients only".
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gsr {

// contrib = P_own * layer4 (what the all-reduce sums), silhouette of the whole stack, surface depth of the first layer, front to
// back, behind which the accumulated transmittance is <= 0.5 (else of the last layer that has one)
__global__ void __launch_bounds__(256)
K_composite_fwd(int world, int rank, const long long* __restrict__ order, const float* __restrict__ gathered, int planes, const float* __restrict__ layer4,
                size_t N, int has_sur, float* __restrict__ contrib, float* __restrict__ sil_total, float* __restrict__ surf)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    float T = 1.f, P_own = 0.f, su = 0.f;
    bool found = false;
    for (int k = 0; k < world; k++) {
        const int r = (int)order[k];
        const float S = gathered[((size_t)r * planes) * N + i];
        if (r == rank) P_own = T;
        const float T_after = T * (1.f - S);
        if (has_sur) {
            const float SU = gathered[((size_t)r * planes + 1) * N + i];
            const bool has = SU > 0.f;
            if (!found && has) su = SU;
            found = found || (has && T_after <= 0.5f);
        }
        T = T_after;
    }
#pragma unroll
    for (int c = 0; c < 4; c++) contrib[c * N + i] = P_own * layer4[c * N + i];
    sil_total[i] = 1.f - T;
    if (surf) surf[i] = su;
}

// what needs nothing from the other ranks: dL/dlayer4 = P_own * g4, and c_own = g4 . layer4 (what the layers in FRONT need)
__global__ void __launch_bounds__(256)
K_composite_bwd_local(int world, int rank, const long long* __restrict__ order, const float* __restrict__ gathered, int planes, const float* __restrict__ layer4,
                      const float* __restrict__ g4, size_t N, float* __restrict__ d_layer4, float* __restrict__ c_own)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    float T = 1.f, P_own = 0.f;
    for (int k = 0; k < world; k++) {
        const int r = (int)order[k];
        if (r == rank) { P_own = T; break; }
        T *= 1.f - gathered[((size_t)r * planes) * N + i];
    }
    float c = 0.f;
#pragma unroll
    for (int ch = 0; ch < 4; ch++) {
        const float g = g4 ? g4[ch * N + i] : 0.f;
        d_layer4[ch * N + i] = P_own * g;
        c = fmaf(g, layer4[ch * N + i], c);
    }
    c_own[i] = c;
}

// the layer's occlusion of what lies behind it: dL/dS_own = - sum_{k behind own} (prod_{h before k, h != own} (1 - S_h)) c_k
//                                                          + g_sil prod_{h != own} (1 - S_h)        (c_all [world,N] in rank order)
__global__ void __launch_bounds__(256)
K_composite_bwd_occlusion(int world, int rank, const long long* __restrict__ order, const float* __restrict__ gathered, int planes, const float* __restrict__ c_all,
                          const float* __restrict__ g_sil, size_t N, float* __restrict__ dS)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    float P_excl = 1.f, acc = 0.f;
    bool behind = false;
    for (int k = 0; k < world; k++) {
        const int r = (int)order[k];
        if (behind) acc = fmaf(P_excl, c_all[(size_t)r * N + i], acc);
        if (r == rank) behind = true;
        else P_excl *= 1.f - gathered[((size_t)r * planes) * N + i];
    }
    dS[i] = (g_sil ? g_sil[i] * P_excl : 0.f) - acc;
}

// ---- round 6: the BAND exchange. Instead of every rank compositing and evaluating the loss on the whole frame behind an
// all-gather + all-reduce, rank r receives every rank's layer for ITS band of pixel rows (one grouped point-to-point exchange), composites the band,
// evaluates the loss there, takes the band's gradient back through the composite for EVERY rank's layer and returns each rank its rows (a second
// exchange): all per-pixel work / world, two collectives instead of three, and at 8 ranks a third of the bytes.
//   layers_all [world][6][H][W]  rank k's layer (rgb, depth, silhouette S, surface depth) in rank order; only the rows this rank needs are valid;
//                                `own` [6][H][W] stands in for layers_all[rank] (the rank's own render: no copy)
// K_band_composite_fwd: rows [e0, e1) (the band and, for the mapping loss's SSIM window, ten rows either side): rgb = sum_k P_k rgb_k; on the band's own
// rows [b0, b1) also depth, the stack's silhouette and the surface depth (K_composite_fwd's rule).
__global__ void __launch_bounds__(256)
K_band_composite_fwd(int world, int rank, const long long* __restrict__ order, const float* __restrict__ layers_all, const float* __restrict__ own, size_t N, int W,
                     int e0, int e1, int b0, int b1, float* __restrict__ out_rgbd, float* __restrict__ out_sil, float* __restrict__ out_sur)
{
    const size_t i = (size_t)e0 * W + (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)e1 * W) return;
    const bool band = i >= (size_t)b0 * W && i < (size_t)b1 * W;
    float T = 1.f, c0 = 0.f, c1 = 0.f, c2 = 0.f, dp = 0.f, su = 0.f;
    bool found = false;
    for (int k = 0; k < world; k++) {
        const int r = (int)order[k];
        const float* const L = r == rank ? own : layers_all + (size_t)r * 6 * N;
        const float S = L[4 * N + i];
        c0 = fmaf(T, L[i], c0); c1 = fmaf(T, L[N + i], c1); c2 = fmaf(T, L[2 * N + i], c2);
        const float T_after = T * (1.f - S);
        if (band) {
            dp = fmaf(T, L[3 * N + i], dp);
            const float SU = L[5 * N + i];
            const bool has = SU > 0.f;
            if (!found && has) su = SU;
            found = found || (has && T_after <= 0.5f);
        }
        T = T_after;
    }
    out_rgbd[i] = c0; out_rgbd[N + i] = c1; out_rgbd[2 * N + i] = c2;
    if (band) { out_rgbd[3 * N + i] = dp; out_sil[i] = 1.f - T; out_sur[i] = su; }
}
// K_band_composite_bwd: rows [b0, b1): from the loss's gradient g4 (rgb, depth) on the composite to EVERY rank's layer gradient on those rows:
// d_all[k] = {P_k g4 (4 planes), dS_k}, dS_k = -P_k B_k with B_k = c_(k+1) + (1 - S_(k+1)) B_(k+1) over the front-to-back order (c_j = g4 . layer_j):
// the occlusion term of K_composite_bwd_occlusion without a division by 1 - S; g_sil (nullptr: none), the stack's silhouette's upstream gradient, adds g_sil P_k Q_k. d_own [5][H][W] receives this rank's (no copy), d_all [world][5][H][W] the others'.
template <int MAXW>
__global__ void __launch_bounds__(256)
K_band_composite_bwd(int world, int rank, const long long* __restrict__ order, const float* __restrict__ layers_all, const float* __restrict__ own,
                     const float* __restrict__ g4, const float* __restrict__ g_sil, size_t N, int W, int b0, int b1, float* __restrict__ d_all, float* __restrict__ d_own)
{
    const size_t i = (size_t)b0 * W + (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)b1 * W) return;
    const float g0 = g4[i], g1 = g4[N + i], g2 = g4[2 * N + i], g3 = g4[3 * N + i], gs = g_sil ? g_sil[i] : 0.f;
    float S[MAXW], c[MAXW];
#pragma unroll
    for (int k = 0; k < MAXW; k++) {
        S[k] = 0.f; c[k] = 0.f;
        if (k < world) {
            const int r = (int)order[k];
            const float* const L = r == rank ? own : layers_all + (size_t)r * 6 * N;
            S[k] = L[4 * N + i];
            c[k] = fmaf(g3, L[3 * N + i], fmaf(g2, L[2 * N + i], fmaf(g1, L[N + i], g0 * L[i])));
        }
    }
    float B[MAXW]; // B_k: what the layers behind k contribute through k's transmittance, minus the silhouette's upstream gradient through the layers behind k
    float acc = 0.f, Q = 1.f; // Q: prod_{h behind k} (1 - S_h) — d(1 - prod (1 - S)) / dS_k = P_k Q_k
#pragma unroll
    for (int k = MAXW - 1; k >= 0; k--) {
        B[k] = fmaf(-gs, Q, acc);
        if (k < world) { acc = fmaf(1.f - S[k], acc, c[k]); Q *= 1.f - S[k]; }
    }
    float P = 1.f;
#pragma unroll
    for (int k = 0; k < MAXW; k++) {
        if (k < world) {
            const int r = (int)order[k];
            float* const D = r == rank ? d_own : d_all + (size_t)r * 5 * N;
            D[i] = P * g0; D[N + i] = P * g1; D[2 * N + i] = P * g2; D[3 * N + i] = P * g3;
            D[4 * N + i] = -P * B[k];
            P *= 1.f - S[k];
        }
    }
}
// The mapping loss's totals from every rank's row (gsr_map_loss_finish_rows on the rank's band: rows [world][16] = {sums[8] | reg_out[4] | the rank's loss slot:
// NaN = its forward overflowed | 3 unused}): sums[8] as gsr_pixel_loss of the WHOLE frame, reg_out[4] of the whole map, the iteration's loss (NaN if any rank's is).
struct ShardTotals {
    float w[3], c_ssim, w_long, w_scalar, inv_pixels3, inv_count_ssim;
};
__global__ void __launch_bounds__(64)
K_shard_map_totals(int world, const float* __restrict__ rows, ShardTotals t, float* __restrict__ sums, float* __restrict__ reg_out, float* __restrict__ loss)
{
    if (threadIdx.x != 0) return;
    float a[12] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    bool bad = false;
    for (int r = 0; r < world; r++) {
#pragma unroll
        for (int q = 0; q < 12; q++) a[q] += rows[r * 16 + q];
        const float l = rows[r * 16 + 12];
        bad = bad || l != l;
    }
    const float pix = t.w[0] * (a[0] * t.inv_pixels3) + t.w[1] * (a[1] / fmaxf(a[2], 1.f)) + t.w[2] * (a[3] / fmaxf(a[4], 1.f));
    const float reg = t.w_long * (a[8] > 0.f ? a[10] / a[8] : 0.f) + t.w_scalar * a[9];
    sums[0] = a[0]; sums[1] = a[1]; sums[2] = a[2]; sums[3] = a[3]; sums[4] = a[4]; sums[5] = pix; sums[6] = a[6]; sums[7] = 0.f;
    if (reg_out) { reg_out[0] = a[8]; reg_out[1] = a[9]; reg_out[2] = a[10]; reg_out[3] = reg; }
    loss[0] = bad ? __builtin_nanf("") : pix + t.c_ssim * (1.f - a[6] * t.inv_count_ssim) + reg;
}

// Front-to-back order of the cells of a k-d partition of the map for the camera of Tcw (row-major 4x4, world -> camera): the leaves of
// a BSP are ordered exactly by visiting, at every split, the side that holds the camera centre first. nodes [world - 1][4] =
// {axis, split, left, right}; a child >= 0 is a node, a child < 0 the leaf (rank) -1 - child. One thread: world <= a few dozen.
__global__ void K_shard_order(int world, const float* __restrict__ nodes, const float* __restrict__ Tcw, long long* __restrict__ order)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (world == 1) { order[0] = 0; return; }
    // camera centre c = -R^T t
    float c[3];
#pragma unroll
    for (int k = 0; k < 3; k++) c[k] = -(Tcw[0 * 4 + k] * Tcw[3] + Tcw[1 * 4 + k] * Tcw[7] + Tcw[2 * 4 + k] * Tcw[11]);
    int stack[64], sp = 0, out = 0;
    stack[sp++] = 0;
    while (sp > 0 && out < world) {
        const int n = stack[--sp];
        if (n < 0) { order[out++] = (long long)(-1 - n); continue; }
        const int axis = (int)nodes[4 * n], left = (int)nodes[4 * n + 2], right = (int)nodes[4 * n + 3];
        const bool near_left = c[axis] < nodes[4 * n + 1];
        if (sp + 2 > 64) break;
        stack[sp++] = near_left ? right : left; // far side: popped second
        stack[sp++] = near_left ? left : right;
    }
}

} // namespace gsr
