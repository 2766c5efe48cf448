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
