// gsr_train.h — the two non-rasterizer hot spots of a mapping iteration, fused:
//
//   SSIM   reference: ORB_SLAM2::ssim, src/Utils.cc:77-100 — five depthwise 11x11 convolutions (mu1, mu2, E[x^2],
//          E[y^2], E[xy]) plus a dozen elementwise passes through libtorch, and the same again backwards: 2.2 ms of a
//          6.0 ms mapping iteration at 1200x680 through MIOpen. Here: one kernel forwards (tile + halo in LDS, the
//          separable window as a row pass and a column pass, the SSIM map, its three partial derivatives w.r.t. the
//          window sums of the FIRST image) and one backwards (the transposed window over the three derivative maps).
//          The window is whatever 11 taps the caller passes (the reference's is asymmetric: harness.py:_ssim_taps).
//   Adam   reference: torch::optim::Adam, src/Gaussian.cc:144-175 (eps 1e-15, no weight decay, no amsgrad) — libtorch
//          runs it as ~12 elementwise passes per parameter tensor (0.6 ms per step at 1 M Gaussians); here one pass.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gsr {

#define GSR_SSIM_R 5 // window radius: 11 taps
#define GSR_SSIM_COLS (64 - 2 * GSR_SSIM_R) // output columns of a wave's strip: one lane per INPUT column, the five on either side are halo
#ifndef GSR_SSIM_WAVES
#define GSR_SSIM_WAVES 1024 // waves a launch aims for: one per SIMD of the chip (256 CUs x 4)
#endif
struct SsimTaps {
    float g[2 * GSR_SSIM_R + 1];
};
// How the SSIM kernels cut an image (round 5): a wave owns a strip of GSR_SSIM_COLS output columns and a segment of `rows` output rows of one
// channel and streams down the strip, one image row per step (below). nsx strips x nsy segments x C channels = the launch's single-wave workgroups
// (= gsr_ssim_partials); the segments are as tall as a launch of ~GSR_SSIM_WAVES waves allows (the ten halo rows of a segment are its overhead),
// at least 16 rows.
struct SsimGrid {
    int nsx, nsy, rows;
};
__host__ __device__ inline SsimGrid ssim_grid(int C, int H, int W)
{
    SsimGrid g;
    g.nsx = (W + GSR_SSIM_COLS - 1) / GSR_SSIM_COLS;
    int want = GSR_SSIM_WAVES / (C * g.nsx);
    const int most = (H + 15) / 16;
    want = want > most ? most : want;
    want = want < 1 ? 1 : want;
    g.rows = (H + want - 1) / want;
    g.rows = (g.rows + 2 * GSR_SSIM_R + 2 * GSR_SSIM_R) / (2 * GSR_SSIM_R + 1) * (2 * GSR_SSIM_R + 1) - 2 * GSR_SSIM_R; // rows + 10 steps: whole groups of eleven
    g.nsy = (H + g.rows - 1) / g.rows;
    return g;
}
// Which rows of the image a launch of the SSIM kernels works on (round 6: a rank of the sharded loop evaluates the loss on its band of pixel rows only).
// Outputs (the forward's derivative maps, the backward's gradient) are produced for rows [ybeg, yend) — the grid's segments cut THAT range —, the
// forward's sums take the rows [sbeg, send) only (the band; the maps of the five rows either side are what the band's gradient needs). Zero padding
// still happens at the image's own border (H). The whole image: {0, H, 0, H}.
struct SsimRows {
    int ybeg, yend, sbeg, send;
};
// sum over the wave in eight DPP adds (row_shr 1, 2, 4, 8 inside the rows of 16 lanes, row_bcast15 / row_bcast31 across them): the total is in
// lane 63. (__shfl_xor is six ds_bpermute round trips through the LDS pipe.)
template <int CTRL, int ROWS>
__device__ __forceinline__ float dpp_add_f(float v)
{
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROWS, 0xf, false));
}
__device__ __forceinline__ float wave_sum_lane63(float v)
{
    v = dpp_add_f<0x111, 0xf>(v); v = dpp_add_f<0x112, 0xf>(v); v = dpp_add_f<0x114, 0xf>(v); v = dpp_add_f<0x118, 0xf>(v);
    v = dpp_add_f<0x142, 0xa>(v); v = dpp_add_f<0x143, 0xc>(v);
    return v;
}
// "Am I the last workgroup of this launch to arrive?" for kernels that finish their own partial sums (K_track_loss, K_pose_step). Called by ONE
// thread after the workgroup's write-through stores have drained (s_waitcnt vmcnt(0) in every storing wave + __syncthreads()). One counter word
// takes ~11 ns per arrival (512-1024 workgroups on one word: 6-11 us, measured), so the arrivals are sharded: the workgroups with the same
// blockIdx.x % 8 (one XCD's, as the dispatcher places them) share a word, the last of each shard arrives at the top word. tickets: GSR_TICKET_WORDS
// (include/gsr.h) device words, zero between launches (the last arrivers put them back); words 64 bytes apart.
#define GSR_TICKET_STRIDE 16
#define GSR_TICKET_WORDS_DEV (9 * GSR_TICKET_STRIDE) // == GSR_TICKET_WORDS of include/gsr.h (checked in gsr_api.hip)
__device__ __forceinline__ bool last_arriver(uint32_t* tickets, const uint32_t nblocks)
{
    const uint32_t grp = blockIdx.x & 7u, members = (nblocks - grp + 7u) >> 3, ngroups = nblocks < 8u ? nblocks : 8u;
    uint32_t* const mine = tickets + (1u + grp) * GSR_TICKET_STRIDE;
    if (__hip_atomic_fetch_add(mine, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != members - 1u) return false;
    __hip_atomic_store(mine, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (__hip_atomic_fetch_add(tickets, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != ngroups - 1u) return false;
    __hip_atomic_store(tickets, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
}

// MAPLOSS (round 4): the pixel terms of the mapping loss (K_loss_sums, mode 1: colour L1, masked depth L1, masked surface-depth L1 and their
// counts) ride on the SSIM pass, which has both images' pixels in its hands anyway: a row of six partial sums per workgroup
// {SSIM map, |image - rgb|, |depth - fd| over fd > 0, its count, |sur - fd| over fd > 0 && sil > thr, its count} (the depth planes by the
// workgroups of channel 0), one launch instead of two and one pass over the render instead of two.
struct MapLossPlanes {
    const float *depth, *sur, *sil, *fdepth; // [H,W] each (image / frame_rgb are the SSIM kernel's img1 / img2)
    float thr;
    float* partial6;                         // [6][workgroups]
};
// zero-padded "same" cross-correlation, like conv2d(padding = 5): out[y][x] = sum_k sum_l g[k] g[l] in[y + k - 5][x + l - 5]
//
// Round 5: a STREAMING separable window. One wave per (channel, strip, segment); lane = input column; per step the wave takes ONE image row
// (requested eleven steps earlier: a register queue, so the only trips to memory a wave waits for are its first), hands it round through a
// wave-private LDS row (one 8-byte write, eleven 8-byte reads: the two images interleaved), forms the five horizontal window sums of its column,
// keeps the last eleven rows of them in registers (the loop is unrolled by eleven: the ring's slots are static) and, from the eleventh row on,
// the vertical sums of the output row five rows up, the SSIM map and its three derivative maps. No barrier, no tile: the two passes of a 16x16
// tile kernel (round 4: 32.4 us plain, 39.8 us with the mapping loss riding along at 1200x680x3) ran 26 halo rows and three barriers per 16
// output rows, 38 workgroups deep per CU; every one of them waited for its own loads. The arithmetic is the old kernels', operation for
// operation (same taps order, same fused multiply-adds: the maps are bit-identical), two window sums per instruction where they share a tap
// (v_pk_fma_f32: (mu1, mu2), (E11, E22)).
__device__ __forceinline__ v2f pk_fma(const v2f gg, const v2f x, const v2f acc) { return __builtin_elementwise_fma(gg, x, acc); }
// the eleven taps as (g, g) pairs in VECTOR registers: as kernel arguments they cost 33 scalar registers (a pair for the packed instructions, the
// plain one for the others) and the scalar file spills into vector lanes (v_readlane per use); the wave has vector registers to spare
struct SsimTapRegs {
    v2f g[2 * GSR_SSIM_R + 1];
};
__device__ __forceinline__ SsimTapRegs tap_regs(const SsimTaps& t, const bool reversed)
{
    SsimTapRegs r;
#pragma unroll
    for (int k = 0; k <= 2 * GSR_SSIM_R; k++) {
        float g = t.g[reversed ? 2 * GSR_SSIM_R - k : k];
        asm volatile("" : "+v"(g));
        r.g[k].x = g; r.g[k].y = g;
    }
    return r;
}
// element at byte offset `off` (32 bits, a vector register) from a wave-uniform base: global_load ... v_off, s[base:base+1] — the row's offset
// is ONE vector add per step for every plane the step touches instead of a 64-bit scalar add per plane
__device__ __forceinline__ float ld_off(const float* base, const uint32_t off) { return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + off); }
__device__ __forceinline__ void st_off(float* base, const uint32_t off, const float v) { *reinterpret_cast<float*>(reinterpret_cast<char*>(base) + off) = v; }

// MAPLOSS: the launch carries as many more single-wave workgroups as it has SSIM waves; they form the pixel terms (colour L1, masked depth L1,
// masked surface-depth L1 and their counts), one strided share of the pixels each, and fill rows 1..5 of the six-row partial sums; the SSIM
// waves fill row 0. (Riding on the SSIM waves themselves, as in round 4's tile kernel, the terms cost a lone wave 9 us of issue slots.)
template <bool MAPLOSS>
__global__ void __launch_bounds__(64)
K_ssim_fwd(const float* __restrict__ img1, const float* __restrict__ img2, int C, int H, int W, SsimTaps taps, SsimGrid sg,
           float* __restrict__ partial, float* __restrict__ dmaps, MapLossPlanes ml, SsimRows rw)
{
    constexpr int R = GSR_SSIM_R, NT = 2 * R + 1;
    __shared__ v2f row[2][64 + 2 * R + 2]; // (image 1, image 2) of the current row at [lane + R]; the pads stay zero
    const int lane = threadIdx.x;
    const int nss = C * sg.nsx * sg.nsy;
    const size_t plane = (size_t)H * W, N = (size_t)C * plane;
    if (MAPLOSS && (int)blockIdx.x >= nss) {
        const size_t first = (size_t)rw.sbeg * W + ((size_t)blockIdx.x - nss) * 64 + lane, stride = ((size_t)gridDim.x - nss) * 64, pend = (size_t)rw.send * W;
        float l1 = 0.f, l2 = 0.f, l3 = 0.f, l4 = 0.f, l5 = 0.f;
        const bool has_d = ml.depth != nullptr, has_s = ml.sur != nullptr, has_m = ml.sil != nullptr;
        const float* const pd = has_d ? ml.depth : ml.fdepth; // (always valid addresses: what comes back is not used)
        const float* const ps = has_s ? ml.sur : ml.fdepth;
        const float* const pm = has_m ? ml.sil : ml.fdepth;
#pragma unroll 2
        for (size_t p = first; p < pend; p += stride) {
            const float a0 = img1[p], a1 = img1[plane + p], a2 = img1[2 * plane + p], b0 = img2[p], b1 = img2[plane + p], b2 = img2[2 * plane + p];
            const float fd = ml.fdepth[p], dp = pd[p], su = ps[p], si = has_m ? pm[p] : 1.0e30f;
            l1 += (fabsf(a0 - b0) + fabsf(a1 - b1)) + fabsf(a2 - b2);
            const bool on = fd > 0.f, on2 = on && has_s && si > ml.thr;
            l2 += on && has_d ? fabsf(dp - fd) : 0.f;
            l3 += on ? 1.f : 0.f;
            l4 += on2 ? fabsf(su - fd) : 0.f;
            l5 += on2 ? 1.f : 0.f;
        }
        l1 = wave_sum_lane63(l1); l2 = wave_sum_lane63(l2); l3 = wave_sum_lane63(l3); l4 = wave_sum_lane63(l4); l5 = wave_sum_lane63(l5);
        if (lane == 63) {
            float* const o = ml.partial6 + (blockIdx.x - nss);
            o[(size_t)1 * nss] = l1; o[(size_t)2 * nss] = l2; o[(size_t)3 * nss] = l3; o[(size_t)4 * nss] = l4; o[(size_t)5 * nss] = l5;
        }
        return;
    }
    // (integer division runs on the vector pipe: without the readfirstlane the quotients, and every address and condition made from them, stay there)
    const int sx = __builtin_amdgcn_readfirstlane((int)(blockIdx.x % sg.nsx)), sy = __builtin_amdgcn_readfirstlane((int)((blockIdx.x / sg.nsx) % sg.nsy)),
              c = __builtin_amdgcn_readfirstlane((int)(blockIdx.x / (sg.nsx * sg.nsy)));
    const int gx = sx * GSR_SSIM_COLS - R + lane, y0 = rw.ybeg + sy * sg.rows;
    const int nrows = min(sg.rows, rw.yend - y0);
    const bool col_in = gx >= 0 && gx < W, col_out = lane >= R && lane < 64 - R && gx < W;
    const float* __restrict__ p1 = img1 + c * plane;
    const float* __restrict__ p2 = img2 + c * plane;
    {
        v2f z; z.x = 0.f; z.y = 0.f;
        row[0][lane] = z; row[1][lane] = z;
        if (lane < 2 * R + 2) { row[0][64 + lane] = z; row[1][64 + lane] = z; }
        lds_turn();
    }
    // The queue holds what the loads return, untouched (a select on a value just requested would wait for it): rows outside the image are
    // asked for at clamped addresses and zeroed when they are taken out.
    const uint32_t gxb = 4u * (uint32_t)min(max(gx, 0), W - 1), Wb = 4u * (uint32_t)W;
    const SsimTapRegs tg = tap_regs(taps, false);
    float qa[NT], qb[NT];
    auto request = [&](const int i, const int slot) { // image row y0 - R + i into queue slot `slot`
        const uint32_t off = gxb + (uint32_t)min(max(y0 - R + i, 0), H - 1) * Wb;
        qa[slot] = ld_off(p1, off);
        qb[slot] = ld_off(p2, off);
    };
#pragma unroll
    for (int j = 0; j < NT; j++) request(j, j);
    // row i goes through LDS one step AHEAD of its arithmetic: the eleven reads of step i + 1 are in flight while step i's sums are formed
    auto stage = [&](const int i, const int j, v2f (&n)[NT]) {
        const int gy = y0 - R + i;
        const bool in = col_in && gy >= 0 && gy < H;
        v2f ab; ab.x = in ? qa[j] : 0.f; ab.y = in ? qb[j] : 0.f;
        request(i + NT, j);
        v2f* const buf = row[i & 1];
        buf[lane + R] = ab;
        lds_turn(); // (the wave's LDS queue is in order; the compiler must not lift the neighbours' reads above the write: to it they do not alias)
#pragma unroll
        for (int k = 0; k < NT; k++) n[k] = buf[lane + k];
    };
    v2f hA[NT], hB[NT]; // the ring: (mu1, mu2) and (E11, E22) row sums of the last eleven rows, E12's in hC
    float hC[NT];
    float lsum = 0.f;
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
    v2f ncur[NT];
    stage(0, 0, ncur);
    // step i: the sums of image row y0 - R + i enter the ring (slot j = i % 11), output row y0 + i - 10 goes out (OUT: from the eleventh step on)
    auto step = [&](const int i, const int j, auto OUT) {
        v2f nn[NT];
        stage(i + 1, (j + 1) % NT, nn);
        v2f sA, sB; sA.x = 0.f; sA.y = 0.f; sB = sA;
        float sC = 0.f;
#pragma unroll
        for (int k = 0; k < NT; k++) {
            const v2f g = tg.g[k];
            const v2f sq = ncur[k] * ncur[k];
            const float xy = ncur[k].x * ncur[k].y;
            sA = pk_fma(g, ncur[k], sA); sB = pk_fma(g, sq, sB); sC = fmaf(g.x, xy, sC);
        }
        hA[j] = sA; hB[j] = sB; hC[j] = sC;
#pragma unroll
        for (int k = 0; k < NT; k++) ncur[k] = nn[k];
        if (!decltype(OUT)::value) return;
        // the window's rows are the ring's slots j + 1 ... j + 11 (mod 11), top to bottom
        v2f vA, vB; vA.x = 0.f; vA.y = 0.f; vB = vA;
        float vC = 0.f;
#pragma unroll
        for (int k = 0; k < NT; k++) {
            const v2f g = tg.g[k];
            const int sl = (j + 1 + k) % NT;
            vA = pk_fma(g, hA[sl], vA); vB = pk_fma(g, hB[sl], vB); vC = fmaf(g.x, hC[sl], vC);
        }
        const float mu1 = vA.x, mu2 = vA.y, mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
        const float s1 = vB.x - mu1_sq, s2 = vB.y - mu2_sq, s12 = vC - mu12;
        const float A = 2.f * mu12 + C1, B = 2.f * s12 + C2, Cc = mu1_sq + mu2_sq + C1, D = s1 + s2 + C2;
        // (hardware reciprocals, 1 ulp: four IEEE divisions were 40 of the step's ~230 vector instructions)
        const float rC = __builtin_amdgcn_rcpf(Cc), rD = __builtin_amdgcn_rcpf(D), inv = rC * rD, m = (A * B) * inv;
        const int orow = i - 2 * R;
        const bool ok = col_out && orow < nrows; // (the last segment of an image runs whole groups of eleven steps too)
        if (dmaps) {
            // d(map)/d(mu1), d(map)/d(E[x^2]), d(map)/d(E[xy]) with s1 = E[x^2] - mu1^2, s12 = E[xy] - mu1 mu2
            float* const o = dmaps + c * plane;
            const float d0 = 2.f * mu2 * (B - A) * inv - 2.f * mu1 * m * (rC - rD), d1 = -m * rD, d2 = 2.f * A * inv;
            if (ok) {
                const uint32_t off = gxb + (uint32_t)(y0 + orow) * Wb;
                st_off(o, off, d0); st_off(o + N, off, d1); st_off(o + 2 * N, off, d2);
            }
        }
        lsum += (ok && y0 + orow >= rw.sbeg && y0 + orow < rw.send) ? m : 0.f;
    };
    // whole groups of eleven steps (the loop's trip count is the only thing that varies): the first fills the ring and puts out one row
#pragma unroll
    for (int j = 0; j < NT; j++) {
        if (j < NT - 1) step(j, j, std::false_type{});
        else step(j, j, std::true_type{});
    }
    for (int base = NT; base < nrows + 2 * R; base += NT) {
#pragma unroll
        for (int j = 0; j < NT; j++) step(base + j, j, std::true_type{});
    }
    // one sum per wave, added up by the caller (or K_map_finish): a deterministic total
    const float t = wave_sum_lane63(lsum);
    if (lane == 63) (MAPLOSS ? ml.partial6 : partial)[blockIdx.x] = t;
}

// dL/dimg1 = scale * ( corrT(dmu1) + 2 img1 corrT(dE11) + img2 corrT(dE12) ), corrT = the transposed window:
// out[x] = sum_k g[k] d[x - k + 5]; scale = *dL_dmean / (C H W)
// MAPLOSS: the gradient of the mapping loss's pixel terms (K_loss_grad, mode 1) is added where the SSIM term's is written, and the
// workgroups of channel 0 write the depth plane's: dL/dimage and dL/ddepth of the whole image loss in one launch.
struct MapLossGrad {
    const float *depth, *fdepth; // [H,W]
    const float* sums;           // K_map_finish's (the depth term divides by its count)
    float ci, w_depth;           // w_colour / (3 H W), w_depth
    float* ddepth;               // [H,W]
};
template <bool MAPLOSS>
__global__ void __launch_bounds__(64)
K_ssim_bwd(const float* __restrict__ img1, const float* __restrict__ img2, const float* __restrict__ dmaps, int C, int H, int W,
           SsimTaps taps, SsimGrid sg, const float* __restrict__ dL_dmean, float* __restrict__ dL_dimg1, MapLossGrad mg, SsimRows rw)
{
    // the forward's streaming scheme on the three derivative maps (transposed window: the taps run backwards); the images' own pixels of the
    // OUTPUT row travel in the queue with the input row that completes its window. MAPLOSS: the depth plane's gradient is elementwise — the
    // workgroups past the SSIM waves (the launch has them only when a depth gradient is asked for) write it, a strided share of the pixels each.
    constexpr int R = GSR_SSIM_R, NT = 2 * R + 1;
    __shared__ v2f rowA[2][64 + 2 * R + 2];
    __shared__ float rowC[2][64 + 2 * R + 2];
    const int lane = threadIdx.x;
    const int nss = C * sg.nsx * sg.nsy;
    const size_t plane = (size_t)H * W, N = (size_t)C * plane;
    if (MAPLOSS && (int)blockIdx.x >= nss) {
        if (!mg.ddepth) return;
        const size_t first = (size_t)rw.ybeg * W + ((size_t)blockIdx.x - nss) * 64 + lane, stride = ((size_t)gridDim.x - nss) * 64, pend = (size_t)rw.yend * W;
        const float k = mg.w_depth / fmaxf(mg.sums[2], 1.f);
        const float* const pd = mg.depth ? mg.depth : mg.fdepth;
#pragma unroll 4
        for (size_t p = first; p < pend; p += stride) {
            const float fd = mg.fdepth[p], dd = mg.depth ? pd[p] - fd : 0.f;
            mg.ddepth[p] = fd > 0.f ? k * ((float)(dd > 0.f) - (float)(dd < 0.f)) : 0.f;
        }
        return;
    }
    const int sx = __builtin_amdgcn_readfirstlane((int)(blockIdx.x % sg.nsx)), sy = __builtin_amdgcn_readfirstlane((int)((blockIdx.x / sg.nsx) % sg.nsy)),
              c = __builtin_amdgcn_readfirstlane((int)(blockIdx.x / (sg.nsx * sg.nsy)));
    const int gx = sx * GSR_SSIM_COLS - R + lane, y0 = rw.ybeg + sy * sg.rows;
    const int nrows = min(sg.rows, rw.yend - y0);
    const bool col_in = gx >= 0 && gx < W, col_out = lane >= R && lane < 64 - R && gx < W;
    {
        v2f z; z.x = 0.f; z.y = 0.f;
        rowA[0][lane] = z; rowA[1][lane] = z; rowC[0][lane] = 0.f; rowC[1][lane] = 0.f;
        if (lane < 2 * R + 2) { rowA[0][64 + lane] = z; rowA[1][64 + lane] = z; rowC[0][64 + lane] = 0.f; rowC[1][64 + lane] = 0.f; }
        lds_turn();
    }
    const float scale = dL_dmean[0] / (float)N;
    // (the queue as in the forward: raw loads from clamped addresses, zeroed when they are taken out)
    const uint32_t gxb = 4u * (uint32_t)min(max(gx, 0), W - 1), Wb = 4u * (uint32_t)W;
    const SsimTapRegs tg = tap_regs(taps, true); // (transposed window: tap k of the loops below is g[10 - k])
    const float* __restrict__ d0 = dmaps + c * plane;
    const float* __restrict__ pi1 = img1 + c * plane;
    const float* __restrict__ pi2 = img2 + c * plane;
    float q0[NT], q1[NT], q2[NT], qi1[NT], qi2[NT];
    auto request = [&](const int i, const int slot) {
        const uint32_t off = gxb + (uint32_t)min(max(y0 - R + i, 0), H - 1) * Wb;
        q0[slot] = ld_off(d0, off);
        q1[slot] = ld_off(d0 + N, off);
        q2[slot] = ld_off(d0 + 2 * N, off);
        const uint32_t oo = gxb + (uint32_t)min(max(y0 + i - 2 * R, 0), H - 1) * Wb; // the output pixel of step i: row y0 + i - 10
        qi1[slot] = ld_off(pi1, oo);
        qi2[slot] = ld_off(pi2, oo);
    };
#pragma unroll
    for (int j = 0; j < NT; j++) request(j, j);
    // (row i goes through LDS one step ahead of its arithmetic, as in the forward; the output pixel's own values are taken out with it)
    auto stage = [&](const int i, const int j, v2f (&nA)[NT], float (&nC)[NT], float& i1, float& i2) {
        const int gy = y0 - R + i;
        const bool in = col_in && gy >= 0 && gy < H;
        v2f d01; d01.x = in ? q0[j] : 0.f; d01.y = in ? q1[j] : 0.f;
        const float d2 = in ? q2[j] : 0.f;
        i1 = qi1[j]; i2 = qi2[j];
        request(i + NT, j);
        v2f* const bA = rowA[i & 1];
        float* const bC = rowC[i & 1];
        bA[lane + R] = d01;
        bC[lane + R] = d2;
        lds_turn();
#pragma unroll
        for (int k = 0; k < NT; k++) { nA[k] = bA[lane + k]; nC[k] = bC[lane + k]; }
    };
    v2f hA[NT];
    float hC[NT];
    v2f cA[NT];
    float cC[NT], ci1, ci2;
    stage(0, 0, cA, cC, ci1, ci2);
    auto step = [&](const int i, const int j, auto OUT) {
        v2f nA[NT];
        float nC[NT], ni1, ni2;
        stage(i + 1, (j + 1) % NT, nA, nC, ni1, ni2);
        v2f sA; sA.x = 0.f; sA.y = 0.f;
        float sC = 0.f;
#pragma unroll
        for (int k = 0; k < NT; k++) { // transposed: tap k meets d[x - k + 5] = column lane + (10 - k) of the padded row
            const v2f g = tg.g[k];
            sA = pk_fma(g, cA[k], sA); sC = fmaf(g.x, cC[k], sC);
        }
        hA[j] = sA; hC[j] = sC;
        const float i1 = ci1, i2 = ci2;
#pragma unroll
        for (int k = 0; k < NT; k++) { cA[k] = nA[k]; cC[k] = nC[k]; }
        ci1 = ni1; ci2 = ni2;
        if (!decltype(OUT)::value) return;
        v2f vA; vA.x = 0.f; vA.y = 0.f;
        float vC = 0.f;
#pragma unroll
        for (int k = 0; k < NT; k++) {
            const v2f g = tg.g[k];
            const int sl = (j + 1 + k) % NT;
            vA = pk_fma(g, hA[sl], vA); vC = fmaf(g.x, hC[sl], vC);
        }
        const int orow = i - 2 * R;
        const bool ok = col_out && orow < nrows;
        float gval = scale * (vA.x + 2.f * i1 * vA.y + i2 * vC);
        if (MAPLOSS) gval += mg.ci * ((float)(i1 - i2 > 0.f) - (float)(i1 - i2 < 0.f));
        if (ok) st_off(dL_dimg1 + c * plane, gxb + (uint32_t)(y0 + orow) * Wb, gval);
    };
#pragma unroll
    for (int j = 0; j < NT; j++) {
        if (j < NT - 1) step(j, j, std::false_type{});
        else step(j, j, std::true_type{});
    }
    for (int base = NT; base < nrows + 2 * R; base += NT) {
#pragma unroll
        for (int j = 0; j < NT; j++) step(base + j, j, std::true_type{});
    }
}

// torch.optim.Adam's single-tensor update (no weight decay, no amsgrad, not maximising):
//   m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2;  p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
__global__ void __launch_bounds__(256)
K_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, size_t n,
       float w1 /* 1 - beta1 */, float b2, float w2 /* 1 - beta2 */, float eps, float step_size, float sqrt_bias2)
{
    const size_t i4 = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i4 >= n) return;
    auto one = [&](float& pp, const float gg, float& mm, float& vv) {
        mm = fmaf(w1, gg - mm, mm);                    // exp_avg.lerp_(grad, 1 - beta1)
        vv = fmaf(w2 * gg, gg, vv * b2);             // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
        const float denom = sqrtf(vv) / sqrt_bias2 + eps;
        pp = fmaf(-step_size, mm / denom, pp);        // param.addcdiv_(exp_avg, denom, value = -step_size)
    };
    if (i4 + 4 <= n && ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                         reinterpret_cast<uintptr_t>(v)) & 15u) == 0u) {
        float4 P = *reinterpret_cast<float4*>(p + i4), M = *reinterpret_cast<float4*>(m + i4), V = *reinterpret_cast<float4*>(v + i4);
        const float4 G = *reinterpret_cast<const float4*>(g + i4);
        one(P.x, G.x, M.x, V.x); one(P.y, G.y, M.y, V.y); one(P.z, G.z, M.z, V.z); one(P.w, G.w, M.w, V.w);
        *reinterpret_cast<float4*>(p + i4) = P; *reinterpret_cast<float4*>(m + i4) = M; *reinterpret_cast<float4*>(v + i4) = V;
    } else {
        for (size_t i = i4; i < n && i < i4 + 4; i++) one(p[i], g[i], m[i], v[i]);
    }
}

// Pose gradient of mc = X R^T + t (Render.cc:750-752: the means moved into the camera frame) from dL/dmc:
//   dL/dR[i][j] = sum_n dmc[n][i] X[n][j],  dL/dt[i] = sum_n dmc[n][i]
// The reference gets it from autograd through bmm; as a GEMM it is a 3 x N x 3 product that rocBLAS runs in 1.7 ms at
// N = 1 M (38 % of a tracking iteration). Here: every thread sums its splats, a DPP butterfly per wave, one row of twelve
// partial sums per workgroup (the caller adds the rows: deterministic).
#define GSR_POSE_BLOCKS 512
struct Pose34 {
    float r[9], t[3]; // R row-major, t
};
// the pose lives on the device (the optimiser's output): every wave reads its twelve numbers with scalar loads
__device__ __forceinline__ Pose34 load_pose(const float* __restrict__ Tcw)
{
    Pose34 p;
#pragma unroll
    for (int i = 0; i < 3; i++) {
#pragma unroll
        for (int j = 0; j < 3; j++) p.r[3 * i + j] = Tcw[4 * i + j];
        p.t[i] = Tcw[4 * i + 3];
    }
    return p;
}
// mc = X R^T + t, one splat per thread (as a GEMM: 63 us at 1 M splats through rocBLAS; this is 24 MB of traffic)
__global__ void __launch_bounds__(256)
K_to_camera(const float* __restrict__ X, size_t n, const float* __restrict__ Tcw, float* __restrict__ mc)
{
    const Pose34 T = load_pose(Tcw);
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float x = X[3 * i], y = X[3 * i + 1], z = X[3 * i + 2];
    mc[3 * i] = fmaf(T.r[2], z, fmaf(T.r[1], y, T.r[0] * x)) + T.t[0];
    mc[3 * i + 1] = fmaf(T.r[5], z, fmaf(T.r[4], y, T.r[3] * x)) + T.t[1];
    mc[3 * i + 2] = fmaf(T.r[8], z, fmaf(T.r[7], y, T.r[6] * x)) + T.t[2];
}
// backward: twelve pose sums per workgroup (if partial) and dL/dX = dmc R (if dX)
struct PoseUpdate;
template <bool COHERENT>
__device__ void pose_update_body(const PoseUpdate& u, int nrows, const float* presum = nullptr);
template <bool STEP>
__device__ __forceinline__ void pose_grad_body(const float* __restrict__ X, const float* __restrict__ dmc, size_t n, const float* Tcw,
                                               float* partial, float* __restrict__ dX, uint32_t* ticket, const PoseUpdate* u)
{
    const Pose34 T = load_pose(Tcw);
    __shared__ float ws[4][12];
    float a[12] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // four splats per thread and trip, all their loads requested before the first is used (one splat per trip: eight dependent trips to
    // memory per thread at 1 M splats, 8.8 us for 24 MB). The order of a thread's additions is unchanged.
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x; i0 < n; i0 += 4 * stride) {
        float g[4][3], x[4][3];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const size_t i = i0 + u * stride, ic = i < n ? i : i0;
#pragma unroll
            for (int k = 0; k < 3; k++) { g[u][k] = dmc[3 * ic + k]; x[u][k] = partial ? X[3 * ic + k] : 0.f; }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const size_t i = i0 + u * stride;
            if (i >= n) break;
            const float g0 = g[u][0], g1 = g[u][1], g2 = g[u][2];
            if (dX) {
                dX[3 * i] = fmaf(g2, T.r[6], fmaf(g1, T.r[3], g0 * T.r[0]));
                dX[3 * i + 1] = fmaf(g2, T.r[7], fmaf(g1, T.r[4], g0 * T.r[1]));
                dX[3 * i + 2] = fmaf(g2, T.r[8], fmaf(g1, T.r[5], g0 * T.r[2]));
            }
            if (partial) {
                a[0] = fmaf(g0, x[u][0], a[0]); a[1] = fmaf(g0, x[u][1], a[1]); a[2] = fmaf(g0, x[u][2], a[2]);
                a[3] = fmaf(g1, x[u][0], a[3]); a[4] = fmaf(g1, x[u][1], a[4]); a[5] = fmaf(g1, x[u][2], a[5]);
                a[6] = fmaf(g2, x[u][0], a[6]); a[7] = fmaf(g2, x[u][1], a[7]); a[8] = fmaf(g2, x[u][2], a[8]);
                a[9] += g0; a[10] += g1; a[11] += g2;
            }
        }
    }
    if (!partial) return;
#pragma unroll
    for (int q = 0; q < 12; q++) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) a[q] += __shfl_xor(a[q], off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int q = 0; q < 12; q++) ws[threadIdx.x >> 6][q] = a[q];
    }
    __syncthreads();
    const float row = threadIdx.x < 12 ? (ws[0][threadIdx.x] + ws[1][threadIdx.x]) + (ws[2][threadIdx.x] + ws[3][threadIdx.x]) : 0.f;
    if (!STEP) {
        if (threadIdx.x < 12) partial[blockIdx.x * 12 + threadIdx.x] = row;
        return;
    }
    // STEP: the workgroup that arrives last takes the pose step (K_pose_update's work, one launch less) — write-through stores of the rows, every
    // storing wave drains them, a ticket per workgroup, one acquire by the holder of the last one: the hand-over of K_bin_colscan (gsr_kernels.hip).
    // Every workgroup has read Tcw by then (it arrives after its sums): the next iteration's matrix can be written.
    __shared__ uint32_t s_ticket;
    if (threadIdx.x < 12) __hip_atomic_store(partial + blockIdx.x * 12 + threadIdx.x, row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) s_ticket = last_arriver(ticket, gridDim.x) ? 1u : 0u;
    __syncthreads();
    if (!s_ticket) return;
    if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __syncthreads();
    if (threadIdx.x < 64) pose_update_body<true>(*u, (int)gridDim.x);
}
__global__ void __launch_bounds__(256)
K_pose_grad(const float* __restrict__ X, const float* __restrict__ dmc, size_t n, const float* __restrict__ Tcw,
            float* __restrict__ partial, float* __restrict__ dX)
{
    pose_grad_body<false>(X, dmc, n, Tcw, partial, dX, nullptr, nullptr);
}

// The feature reprojection term of the tracking loss (src/Render.cc:1031-1096: the ORB matches' 3-D points under the pose being optimised against
// their observed pixels, weighted by the keypoints' inverse level variances, outliers (chi-square 5.991) frozen out halfway through the iterations).
// It depends on the pose alone: Xc = R Xw + t has the pose sums' own form, dL/dR = sum g (x) Xw, dL/dt = sum g, so the term ADDS its twelve sums to one
// row of the pose rows (gsr_pose_grad's partial rows / the fused pose step's accumulator rows) and its value to the iteration's loss, and the pose step
// that follows sees one objective. refresh: 0 use the stored inlier flags, 1 recompute them from the current errors and store them, 2 every match is one.
__global__ void __launch_bounds__(256)
K_reproj(const float* __restrict__ obs, const float* __restrict__ Xw, const float* __restrict__ inv_s2, int M, const float* __restrict__ Tcw,
         float fx, float fy, float cx, float cy, float w_loss, float w_grad, int refresh, uint8_t* __restrict__ inl, float* row, float* loss)
{
    const Pose34 T = load_pose(Tcw);
    __shared__ float ws[4][13];
    float a[13] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < M; i += 256) {
        const float x = Xw[3 * i], y = Xw[3 * i + 1], z = Xw[3 * i + 2];
        const float cxx = fmaf(T.r[2], z, fmaf(T.r[1], y, T.r[0] * x)) + T.t[0], cyy = fmaf(T.r[5], z, fmaf(T.r[4], y, T.r[3] * x)) + T.t[1],
                    czz = fmaf(T.r[8], z, fmaf(T.r[7], y, T.r[6] * x)) + T.t[2];
        const float iz = 1.0f / czz;
        const float ex = fmaf(fx, cxx * iz, cx) - obs[2 * i], ey = fmaf(fy, cyy * iz, cy) - obs[2 * i + 1], s = inv_s2[i];
        const float werr = s * (ex * ex + ey * ey);
        const bool in = refresh == 2 ? true : refresh == 1 ? werr < 5.991f : inl[i] != 0;
        if (refresh == 1) inl[i] = in ? 1 : 0;
        if (!in) continue;
        const float gx = 2.f * s * ex * fx * iz, gy = 2.f * s * ey * fy * iz, gz = -(gx * cxx + gy * cyy) * iz;
        a[0] = fmaf(gx, x, a[0]); a[1] = fmaf(gx, y, a[1]); a[2] = fmaf(gx, z, a[2]);
        a[3] = fmaf(gy, x, a[3]); a[4] = fmaf(gy, y, a[4]); a[5] = fmaf(gy, z, a[5]);
        a[6] = fmaf(gz, x, a[6]); a[7] = fmaf(gz, y, a[7]); a[8] = fmaf(gz, z, a[8]);
        a[9] += gx; a[10] += gy; a[11] += gz; a[12] += werr;
    }
#pragma unroll
    for (int q = 0; q < 13; q++) a[q] = wave_sum_lane63(a[q]);
    if ((threadIdx.x & 63) == 63) {
#pragma unroll
        for (int q = 0; q < 13; q++) ws[threadIdx.x >> 6][q] = a[q];
    }
    __syncthreads();
    if (threadIdx.x < 13) {
        const float tot = (ws[0][threadIdx.x] + ws[1][threadIdx.x]) + (ws[2][threadIdx.x] + ws[3][threadIdx.x]);
        if (threadIdx.x < 12) row[threadIdx.x] += w_grad * tot;
        else loss[0] += w_loss * tot;
    }
}

// Pose from the optimiser's parameters, rt2T of the reference (include/Utils.h:56-77, src/Utils.cc:170-179): an
// un-normalised quaternion (r, x, y, z) and a translation -> the 4x4 row-major Tcw, and its backward. In libtorch this
// is ~40 scalar-tensor kernels forwards and ~80 backwards per tracking iteration (0.4 ms of launches); one thread does it.
__global__ void K_rt2T(const float* __restrict__ quat, const float* __restrict__ trans, float* __restrict__ T)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float n = sqrtf(quat[0] * quat[0] + quat[1] * quat[1] + quat[2] * quat[2] + quat[3] * quat[3]);
    const float r = quat[0] / n, x = quat[1] / n, y = quat[2] / n, z = quat[3] / n;
    T[0] = 1.f - 2.f * (y * y + z * z); T[1] = 2.f * (x * y - r * z); T[2] = 2.f * (x * z + r * y); T[3] = trans[0];
    T[4] = 2.f * (x * y + r * z); T[5] = 1.f - 2.f * (x * x + z * z); T[6] = 2.f * (y * z - r * x); T[7] = trans[1];
    T[8] = 2.f * (x * z - r * y); T[9] = 2.f * (y * z + r * x); T[10] = 1.f - 2.f * (x * x + y * y); T[11] = trans[2];
    T[12] = 0.f; T[13] = 0.f; T[14] = 0.f; T[15] = 1.f;
}
__global__ void K_rt2T_bwd(const float* __restrict__ quat, const float* __restrict__ dT, float* __restrict__ dquat, float* __restrict__ dtrans)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float n = sqrtf(quat[0] * quat[0] + quat[1] * quat[1] + quat[2] * quat[2] + quat[3] * quat[3]);
    const float r = quat[0] / n, x = quat[1] / n, y = quat[2] / n, z = quat[3] / n;
    const float G00 = dT[0], G01 = dT[1], G02 = dT[2], G10 = dT[4], G11 = dT[5], G12 = dT[6], G20 = dT[8], G21 = dT[9], G22 = dT[10];
    // d/d(unit quaternion)
    const float dr = 2.f * (-z * G01 + y * G02 + z * G10 - x * G12 - y * G20 + x * G21);
    const float dx = 2.f * (y * G01 + z * G02 + y * G10 - 2.f * x * G11 - r * G12 + z * G20 + r * G21 - 2.f * x * G22);
    const float dy = 2.f * (-2.f * y * G00 + x * G01 + r * G02 + x * G10 + z * G12 - r * G20 + z * G21 - 2.f * y * G22);
    const float dz = 2.f * (-2.f * z * G00 - r * G01 + x * G02 + r * G10 - 2.f * z * G11 + y * G12 + x * G20 + y * G21);
    // through q / |q|
    const float dot = r * dr + x * dx + y * dy + z * dz;
    dquat[0] = (dr - r * dot) / n; dquat[1] = (dx - x * dot) / n; dquat[2] = (dy - y * dot) / n; dquat[3] = (dz - z * dot) / n;
    dtrans[0] = dT[3]; dtrans[1] = dT[7]; dtrans[2] = dT[11];
}

// =====================================================================================
// Loss terms of the two loops, fused (round 3). The reference forms them from libtorch tensor expressions — tracking
// (src/Render.cc:1088-1105, L1LossForTracking src/Utils.cc:58-65): silhouette / NaN mask, two masked L1 sums; mapping
// (src/Render.cc:436-471, L1LossForMapping src/Utils.cc:39-56): colour L1 mean, masked depth L1 mean, masked surface-depth
// L1 mean, the two scale regularisers — about 35 (tracking) and 80 (mapping) elementwise / reduction launches per iteration
// forwards and backwards, each a few microseconds of a dependent launch chain. Here: one pass over the pixels that leaves a
// row of partial sums per workgroup, one single-workgroup kernel that adds the rows (deterministic) and forms the loss, and
// one pass that writes the gradient planes (scaled by the upstream gradient, read from the device).
//   mode 0 (tracking): M = sil > thr && !isnan(frame depth);  loss = w[0] * sum_M |img - rgb| + w[1] * sum_M |d - fd|
//   mode 1 (mapping) : V = fd > 0, S = V && sil > thr;
//                      loss = w[0] * sum |img - rgb| / (3 H W) + w[1] * sum_V |d - fd| / |V| + w[2] * sum_S |sur - fd| / max(|S|, 1)
// sums[8] = {sum |img - rgb|, sum |d - fd|, count, sum |sur - fd|, count_S, loss, 0, 0}. `d` is the differentiable depth plane
// (NULL: no such term), `sur` the median-depth plane (no gradient: the rasterizer does not differentiate it).
// =====================================================================================
// Sum of K per-thread values over a 256-thread workgroup (the single-workgroup "finish" kernels: with one wave a few thousand
// partial rows are a chain of dependent loads — 17 / 38 us for 3 900 / 9 700 rows — with four waves and four rows in flight a few us).
#define GSR_FINISH_THREADS 256
template <int K>
__device__ __forceinline__ void finish_sum(float (&a)[K], float (*ws)[K])
{
#pragma unroll
    for (int q = 0; q < K; q++) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) a[q] += __shfl_xor(a[q], off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int q = 0; q < K; q++) ws[threadIdx.x >> 6][q] = a[q];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < K; q++) a[q] = (ws[0][q] + ws[1][q]) + (ws[2][q] + ws[3][q]);
}
#define GSR_LOSS_BLOCKS 1024
struct LossPlanes {
    const float* image;  // [3,H,W]
    const float* depth;  // [H,W] or nullptr
    const float* sur;    // [H,W] or nullptr
    const float* sil;    // [H,W] or nullptr (mask = all pixels that pass the frame test)
    const float* frgb;   // [3,H,W]
    const float* fdepth; // [H,W]
};
__device__ __forceinline__ float sgn(float x) { return (float)(x > 0.f) - (float)(x < 0.f); } // torch.abs' gradient: sign(x), 0 at 0
__global__ void __launch_bounds__(256)
K_loss_sums(LossPlanes p, size_t N, int mode, float thr, float* __restrict__ partial)
{
    __shared__ float ws[4][5];
    float a[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    // (every plane of the pixel is requested before the masks are looked at: behind their branches the loads were two or three dependent
    // trips to memory per pixel, and a thread takes three pixels at 1200x680: 9.5 us for 36 MB)
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (size_t)gridDim.x * 256) {
        const float fd = p.fdepth[i];
        const float sl = p.sil ? p.sil[i] : 0.f;
        const float i0 = p.image[i], i1 = p.image[N + i], i2 = p.image[2 * N + i], f0 = p.frgb[i], f1 = p.frgb[N + i], f2 = p.frgb[2 * N + i];
        const float dp = p.depth ? p.depth[i] : 0.f, sr = p.sur ? p.sur[i] : 0.f;
        const bool solid = !p.sil || sl > thr;
        const bool colour_in = mode == 0 ? (solid && fd == fd) : true;
        const bool depth_in = mode == 0 ? colour_in : fd > 0.f;
        if (colour_in) a[0] += (fabsf(i0 - f0) + fabsf(i1 - f1)) + fabsf(i2 - f2);
        if (depth_in) {
            if (p.depth) a[1] += fabsf(dp - fd);
            a[2] += 1.f;
        }
        if (mode == 1 && p.sur && depth_in && solid) { a[3] += fabsf(sr - fd); a[4] += 1.f; }
        if (mode == 0 && p.sur && depth_in) a[3] += fabsf(sr - fd); // tracking on the surface depth (use_sur_depth)
    }
#pragma unroll
    for (int q = 0; q < 5; q++) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) a[q] += __shfl_xor(a[q], off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int q = 0; q < 5; q++) ws[threadIdx.x >> 6][q] = a[q];
    }
    __syncthreads();
    if (threadIdx.x < 5) partial[blockIdx.x * 5 + threadIdx.x] = (ws[0][threadIdx.x] + ws[1][threadIdx.x]) + (ws[2][threadIdx.x] + ws[3][threadIdx.x]);
}
struct LossWeights {
    float w[3];
};
// COHERENT: the rows were written by other workgroups of the SAME launch with write-through (agent-scope) stores — the last workgroup of
// K_track_loss to arrive runs this (the hand-over of K_bin_colscan, gsr_kernels.hip): read them with agent-scope loads.
template <bool COHERENT>
__device__ __forceinline__ void loss_finish_body(const float* partial, int nblocks, int mode, size_t N, LossWeights w, int depth_from_sur, float* __restrict__ sums,
                                                 float (*ws)[5])
{
    const int lane = threadIdx.x;
    float a[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    for (int b = lane; b < nblocks; b += GSR_FINISH_THREADS) {
#pragma unroll
        for (int q = 0; q < 5; q++) a[q] += COHERENT ? __hip_atomic_load(partial + b * 5 + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : partial[b * 5 + q];
    }
    finish_sum<5>(a, ws);
    if (lane == 0) {
        float loss;
        if (mode == 0) loss = w.w[0] * a[0] + w.w[1] * (depth_from_sur ? a[3] : a[1]);
        else loss = w.w[0] * (a[0] / (3.f * (float)N)) + w.w[1] * (a[1] / fmaxf(a[2], 1.f)) + w.w[2] * (a[3] / fmaxf(a[4], 1.f)); // (an empty mask: 0, in the loss and in its gradient)
        sums[0] = a[0]; sums[1] = a[1]; sums[2] = a[2]; sums[3] = a[3]; sums[4] = a[4]; sums[5] = loss; sums[6] = 0.f; sums[7] = 0.f;
    }
}
__global__ void __launch_bounds__(GSR_FINISH_THREADS)
K_loss_finish(const float* __restrict__ partial, int nblocks, int mode, size_t N, LossWeights w, int depth_from_sur, float* __restrict__ sums)
{
    __shared__ float ws[4][5];
    loss_finish_body<false>(partial, nblocks, mode, N, w, depth_from_sur, sums, ws);
}
// A tracking iteration's loss in ONE pass over the render (round 4): K_loss_sums (mode 0) and K_loss_grad (mode 0, upstream gradient 1) read the same
// planes — the tracking loss is a masked SUM, its gradient needs no total — so the partial sums and the gradient planes come out of the same loads
// (one launch and 30 MB of reads less per iteration). Same expressions, same order of a thread's additions as the two kernels.
__global__ void __launch_bounds__(256)
K_track_loss(LossPlanes p, size_t N, float thr, LossWeights w, float* partial, float* __restrict__ dimage, float* __restrict__ ddepth,
             uint32_t* ticket, int depth_from_sur, float* __restrict__ sums, size_t i_begin, size_t i_end, int sil_is_T)
{
    // (sil_is_T: the `sil` plane holds the render's final transmittance T — what the plain forward keeps per pixel anyway — and the silhouette is 1 - T:
    // a tracking iteration on the surface depth needs nothing else of the fused pair's channels, round 6)
    // (N = H W is the planes' stride; the pixels [i_begin, i_end) are this launch's — the whole image, or one rank's band of rows in the sharded loop)
    __shared__ float ws[4][5];
    __shared__ uint32_t s_ticket;
    float a[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    for (size_t i = i_begin + (size_t)blockIdx.x * 256 + threadIdx.x; i < i_end; i += (size_t)gridDim.x * 256) {
        const float fd = p.fdepth[i];
        const float sl = p.sil ? p.sil[i] : 0.f;
        const float i0 = p.image[i], i1 = p.image[N + i], i2 = p.image[2 * N + i], f0 = p.frgb[i], f1 = p.frgb[N + i], f2 = p.frgb[2 * N + i];
        const float dp = p.depth ? p.depth[i] : 0.f, sr = p.sur ? p.sur[i] : 0.f;
        const bool solid = !p.sil || (sil_is_T ? 1.f - sl : sl) > thr;
        const bool in = solid && fd == fd; // the tracking mask: colour and depth terms alike
        if (in) {
            a[0] += (fabsf(i0 - f0) + fabsf(i1 - f1)) + fabsf(i2 - f2);
            if (p.depth) a[1] += fabsf(dp - fd);
            a[2] += 1.f;
            if (p.sur) a[3] += fabsf(sr - fd); // tracking on the surface depth (use_sur_depth)
        }
        dimage[i] = in ? w.w[0] * sgn(i0 - f0) : 0.f;
        dimage[N + i] = in ? w.w[0] * sgn(i1 - f1) : 0.f;
        dimage[2 * N + i] = in ? w.w[0] * sgn(i2 - f2) : 0.f;
        if (ddepth) ddepth[i] = (in && p.depth) ? w.w[1] * sgn(dp - fd) : 0.f;
    }
#pragma unroll
    for (int q = 0; q < 5; q++) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) a[q] += __shfl_xor(a[q], off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int q = 0; q < 5; q++) ws[threadIdx.x >> 6][q] = a[q];
    }
    __syncthreads();
    const float row = threadIdx.x < 5 ? (ws[0][threadIdx.x] + ws[1][threadIdx.x]) + (ws[2][threadIdx.x] + ws[3][threadIdx.x]) : 0.f;
    if (!ticket) { // the finish is a launch of its own (K_loss_finish)
        if (threadIdx.x < 5) partial[blockIdx.x * 5 + threadIdx.x] = row;
        return;
    }
    // The workgroup that arrives last adds the rows up (K_loss_finish's work, one launch less): write-through stores of the rows, every storing
    // wave drains them, one ticket per workgroup from a device-scope counter, one acquire by the holder of the last one — the hand-over of
    // K_bin_colscan (gsr_kernels.hip). The counter is zero between launches: the last arriver puts it back.
    if (threadIdx.x < 5) __hip_atomic_store(partial + blockIdx.x * 5 + threadIdx.x, row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) s_ticket = last_arriver(ticket, gridDim.x) ? 1u : 0u;
    __syncthreads();
    if (!s_ticket) return;
    if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __syncthreads();
    loss_finish_body<true>(partial, (int)gridDim.x, 0, N, w, depth_from_sur, sums, ws);
}
// gradient planes: dL/dimage [3,H,W] and dL/ddepth [H,W] (nullptr: not wanted), times the upstream gradient *go
__global__ void __launch_bounds__(256)
K_loss_grad(LossPlanes p, size_t N, int mode, float thr, LossWeights w, const float* __restrict__ sums, const float* __restrict__ go,
            float* __restrict__ dimage, float* __restrict__ ddepth, const float* __restrict__ add)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const float g = go ? go[0] : 1.f;
    const float fd = p.fdepth[i];
    const bool solid = !p.sil || p.sil[i] > thr;
    const bool colour_in = mode == 0 ? (solid && fd == fd) : true;
    const bool depth_in = mode == 0 ? colour_in : fd > 0.f;
    const float ci = mode == 0 ? g * w.w[0] : g * w.w[0] / (3.f * (float)N);
    const float cd = mode == 0 ? g * w.w[1] : g * w.w[1] / fmaxf(sums[2], 1.f);
    float im[3], fr[3], ad[3]; // (every load before the first store: the planes may alias as far as the compiler knows)
#pragma unroll
    for (int c = 0; c < 3; c++) { im[c] = p.image[c * N + i]; fr[c] = p.frgb[c * N + i]; ad[c] = add ? add[c * N + i] : 0.f; }
    const float dp = (ddepth && p.depth) ? p.depth[i] : 0.f;
#pragma unroll
    for (int c = 0; c < 3; c++) dimage[c * N + i] = (colour_in ? ci * sgn(im[c] - fr[c]) : 0.f) + ad[c];
    if (ddepth) ddepth[i] = (depth_in && p.depth) ? cd * sgn(dp - fd) : 0.f;
}

// The two scale regularisers of the mapping loss (src/Render.cc:449-462): with sc = exp(log_scales), limit = 0.1 * scene radius,
// w_i = number of components of sc_i above the limit (the reference gathers the rows of every such COMPONENT: a row with two
// oversized axes counts twice), mx / mn = the row's largest / smallest component:
//   reg_scalar = sum_i w_i (mx_i - limit);  reg_long = sum_i w_i (mx_i - mn_i) / sum_i w_i  (0 if nothing is oversized)
//   value = w_long * reg_long + w_scalar * reg_scalar.   out[4] = {sum w, reg_scalar, sum w (mx - mn), value}
__global__ void __launch_bounds__(256)
K_scale_reg(const float* __restrict__ ls, size_t n, float limit, float* __restrict__ partial)
{
    __shared__ float ws[4][3];
    float a[3] = {0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float s0 = expf(ls[3 * i]), s1 = expf(ls[3 * i + 1]), s2 = expf(ls[3 * i + 2]);
        const float wgt = (float)(s0 > limit) + (float)(s1 > limit) + (float)(s2 > limit);
        const float mx = fmaxf(s0, fmaxf(s1, s2)), mn = fminf(s0, fminf(s1, s2));
        a[0] += wgt; a[1] += wgt * (mx - limit); a[2] += wgt * (mx - mn);
    }
#pragma unroll
    for (int q = 0; q < 3; q++) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) a[q] += __shfl_xor(a[q], off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int q = 0; q < 3; q++) ws[threadIdx.x >> 6][q] = a[q];
    }
    __syncthreads();
    if (threadIdx.x < 3) partial[blockIdx.x * 3 + threadIdx.x] = (ws[0][threadIdx.x] + ws[1][threadIdx.x]) + (ws[2][threadIdx.x] + ws[3][threadIdx.x]);
}
__global__ void __launch_bounds__(GSR_FINISH_THREADS)
K_scale_reg_finish(const float* __restrict__ partial, int nblocks, float w_long, float w_scalar, float* __restrict__ out)
{
    __shared__ float ws[4][3];
    const int lane = threadIdx.x;
    float a[3] = {0.f, 0.f, 0.f};
    for (int b = lane; b < nblocks; b += GSR_FINISH_THREADS) {
#pragma unroll
        for (int q = 0; q < 3; q++) a[q] += partial[b * 3 + q];
    }
    finish_sum<3>(a, ws);
    if (lane == 0) {
        out[0] = a[0]; out[1] = a[1]; out[2] = a[2];
        out[3] = w_long * (a[0] > 0.f ? a[2] / a[0] : 0.f) + w_scalar * a[1];
    }
}
// d(value)/d(log_scales): through mx (its first largest component), mn (its first smallest) and sc = exp(ls); w is a count (no gradient)
__global__ void __launch_bounds__(256)
K_scale_reg_bwd(const float* __restrict__ ls, size_t n, float limit, float w_long, float w_scalar, const float* __restrict__ out,
                const float* __restrict__ go, float* __restrict__ dls)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float s[3] = {expf(ls[3 * i]), expf(ls[3 * i + 1]), expf(ls[3 * i + 2])};
    const float wgt = (float)(s[0] > limit) + (float)(s[1] > limit) + (float)(s[2] > limit);
    float d[3] = {0.f, 0.f, 0.f};
    if (wgt > 0.f) {
        int imax = 0, imin = 0;
        if (s[1] > s[imax]) imax = 1;
        if (s[2] > s[imax]) imax = 2;
        if (s[1] < s[imin]) imin = 1;
        if (s[2] < s[imin]) imin = 2;
        const float cnt = out[0], g = go[0];
        const float a = g * w_scalar * wgt, b = cnt > 0.f ? g * w_long * wgt / cnt : 0.f;
#pragma unroll
        for (int k = 0; k < 3; k++) d[k] = ((k == imax ? a + b : 0.f) - (k == imin ? b : 0.f)) * s[k];
    }
    dls[3 * i] = d[0]; dls[3 * i + 1] = d[1]; dls[3 * i + 2] = d[2];
}

// =====================================================================================
// The loops without a tensor library inside the iteration (round 4). Through libtorch a mapping iteration is ~60 launches
// (activations and their autograd, gradient accumulation, zero fills, five Adam launches) and as many host-side dispatches;
// profiled at 1 M Gaussians, 1200x680: 1.08-1.30 ms per iteration of which the rasterizer pair is 0.55. Here an iteration is
//   K_map_prepare  raw parameters -> what the rasterizer takes: camera-frame means, sigmoid / exp / unit quaternion; the
//                  scale regularisers' partial sums ride on the same pass over log_scales
//   (rasterizer forward, loss kernels, rasterizer backward)
//   K_map_update   gradients w.r.t. the rasterizer's inputs -> gradients of the raw parameters (camera transform, sigmoid,
//                  exp, normalisation, the regularisers' gradient) -> Adam step of all five tensors, in place: no gradient
//                  tensor of a raw parameter ever exists
// and a tracking iteration ends in K_pose_update (sum of the pose partials -> rt2T backward -> best-pose bookkeeping -> Adam on
// the seven pose numbers -> the next iteration's Tcw). Every update is predicated on the rasterizer's overflow flag (a
// workspace that was too small skips the iteration's step instead of applying garbage; the host sees it at its next read).
// Reference: src/Render.cc:420-483 (mapping), :1054-1126 (tracking), src/Gaussian.cc:144-175 (optimisers), :97-150 (pose).
// =====================================================================================
#define GSR_MAP_REG_ROWS(n) (((n) + 255) / 256) // partial rows K_map_prepare writes (3 floats each)
__global__ void __launch_bounds__(256)
K_map_prepare(size_t n, const float* __restrict__ xyz, const float* __restrict__ logit, const float* __restrict__ ls,
              const float* __restrict__ quat, const float* __restrict__ Tcw, float* __restrict__ mc, float* __restrict__ opac,
              float* __restrict__ scales, float* __restrict__ rots, float limit, float* __restrict__ reg_partial)
{
    __shared__ float ws[4][3];
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    float a[3] = {0.f, 0.f, 0.f};
    if (i < n) {
        // (all four parameter tensors are requested before the first activation is stored: tensor by tensor they were four dependent trips to memory)
        Pose34 T;
        float x = 0.f, y = 0.f, z = 0.f, lg = 0.f, l0 = 0.f, l1 = 0.f, l2 = 0.f;
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
        if (mc) { T = load_pose(Tcw); x = xyz[3 * i]; y = xyz[3 * i + 1]; z = xyz[3 * i + 2]; }
        if (opac) lg = logit[i];
        if (scales || reg_partial) { l0 = ls[3 * i]; l1 = ls[3 * i + 1]; l2 = ls[3 * i + 2]; }
        if (rots) q = reinterpret_cast<const float4*>(quat)[i];
        if (mc) {
            mc[3 * i] = fmaf(T.r[2], z, fmaf(T.r[1], y, T.r[0] * x)) + T.t[0];
            mc[3 * i + 1] = fmaf(T.r[5], z, fmaf(T.r[4], y, T.r[3] * x)) + T.t[1];
            mc[3 * i + 2] = fmaf(T.r[8], z, fmaf(T.r[7], y, T.r[6] * x)) + T.t[2];
        }
        if (opac) opac[i] = 1.f / (1.f + expf(-lg));
        if (scales || reg_partial) {
            const float s0 = expf(l0), s1 = expf(l1), s2 = expf(l2);
            if (scales) { scales[3 * i] = s0; scales[3 * i + 1] = s1; scales[3 * i + 2] = s2; }
            const float wgt = (float)(s0 > limit) + (float)(s1 > limit) + (float)(s2 > limit);
            const float mx = fmaxf(s0, fmaxf(s1, s2)), mn = fminf(s0, fminf(s1, s2));
            a[0] = wgt; a[1] = wgt * (mx - limit); a[2] = wgt * (mx - mn);
        }
        if (rots) { // torch::nn::functional::normalize: q / max(|q|, 1e-12)
            const float inv = 1.f / fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);
            reinterpret_cast<float4*>(rots)[i] = make_float4(q.x * inv, q.y * inv, q.z * inv, q.w * inv);
        }
    }
    if (!reg_partial) return;
#pragma unroll
    for (int q = 0; q < 3; q++) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) a[q] += __shfl_xor(a[q], off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int q = 0; q < 3; q++) ws[threadIdx.x >> 6][q] = a[q];
    }
    __syncthreads();
    if (threadIdx.x < 3) reg_partial[(size_t)blockIdx.x * 3 + threadIdx.x] = (ws[0][threadIdx.x] + ws[1][threadIdx.x]) + (ws[2][threadIdx.x] + ws[3][threadIdx.x]);
}

struct MapUpdate {
    float *xyz, *rgb, *quat, *logit, *ls;                                  // raw parameters, updated in place
    float *m[5], *v[5];                                                    // Adam moments, same order
    const float *dmc, *dcol, *drot, *dopac, *dscale;                       // dL/d(camera-frame mean, colour, unit quaternion, opacity, scale)
    const float *opac, *scales;                                            // the activations K_map_prepare wrote
    const float* Tcw;
    const float* reg_out;                                                  // K_scale_reg_finish's out[4] (nullptr: no regularisers)
    const uint32_t* overflow;                                              // the rasterizer's flag (nullptr: never skip)
    float limit, w_long, w_scalar;
    float w1, b2, w2, eps, step_size[5], sqrt_bias2[5];
};
__device__ __forceinline__ void adam_one(float& pp, const float gg, float& mm, float& vv, const MapUpdate& u, const int g)
{
    mm = fmaf(u.w1, gg - mm, mm);
    vv = fmaf(u.w2 * gg, gg, vv * u.b2);
    const float denom = sqrtf(vv) / u.sqrt_bias2[g] + u.eps;
    pp = fmaf(-u.step_size[g], mm / denom, pp);
}
// One Gaussian, its gradients w.r.t. the rasterizer's inputs given in registers (K_map_update_small loads them; the per-splat stage of
// the backward has just computed them: gsr_backward_args.fused_map_update). Every load is issued before the first store (the tensors may
// alias as far as the compiler knows: section by section it would wait for each section's round trip — at 10 k Gaussians, where nothing
// else hides it, the launch took 19 us for 0.4 MB).
struct MapRegs { // the raw parameters of one Gaussian, their Adam moments and the activations the chain rule needs
    float X[3], M0[3], V0[3], C[3], M1[3], V1[3], LS[3], M4[3], V4[3], S[3];
    float4 q, M2, V2;
    float L, M3, V3, o, cnt;
};
__device__ __forceinline__ void map_load(const size_t i, const MapUpdate& u, MapRegs& r)
{
#pragma unroll
    for (int k = 0; k < 3; k++) {
        r.X[k] = u.xyz[3 * i + k]; r.M0[k] = u.m[0][3 * i + k]; r.V0[k] = u.v[0][3 * i + k];
        r.C[k] = u.rgb[3 * i + k]; r.M1[k] = u.m[1][3 * i + k]; r.V1[k] = u.v[1][3 * i + k];
        r.LS[k] = u.ls[3 * i + k]; r.M4[k] = u.m[4][3 * i + k]; r.V4[k] = u.v[4][3 * i + k]; r.S[k] = u.scales[3 * i + k];
    }
    r.q = reinterpret_cast<const float4*>(u.quat)[i];
    r.M2 = reinterpret_cast<const float4*>(u.m[2])[i]; r.V2 = reinterpret_cast<const float4*>(u.v[2])[i];
    r.L = u.logit[i]; r.M3 = u.m[3][i]; r.V3 = u.v[3][i];
    r.o = u.opac[i];
    r.cnt = u.reg_out ? u.reg_out[0] : 0.f;
}
// keeps the requests of map_load where they are written (a value the compiler may not assume unchanged has to be there)
__device__ __forceinline__ void map_pin(MapRegs& r)
{
#define GSR_PIN(x) asm volatile("" : "+v"(x))
#pragma unroll
    for (int k = 0; k < 3; k++) {
        GSR_PIN(r.X[k]); GSR_PIN(r.M0[k]); GSR_PIN(r.V0[k]); GSR_PIN(r.C[k]); GSR_PIN(r.M1[k]); GSR_PIN(r.V1[k]);
        GSR_PIN(r.LS[k]); GSR_PIN(r.M4[k]); GSR_PIN(r.V4[k]); GSR_PIN(r.S[k]);
    }
    GSR_PIN(r.q.x); GSR_PIN(r.q.y); GSR_PIN(r.q.z); GSR_PIN(r.q.w); GSR_PIN(r.M2.x); GSR_PIN(r.M2.y); GSR_PIN(r.M2.z); GSR_PIN(r.M2.w);
    GSR_PIN(r.V2.x); GSR_PIN(r.V2.y); GSR_PIN(r.V2.z); GSR_PIN(r.V2.w); GSR_PIN(r.L); GSR_PIN(r.M3); GSR_PIN(r.V3); GSR_PIN(r.o);
#undef GSR_PIN
}
__device__ __forceinline__ void map_apply(const size_t i, const MapUpdate& u, const Pose34& T, MapRegs& r, const float (&GX)[3], const float (&GC)[3],
                                          const float4 dr, const float gop, const float (&GS)[3])
{
    float (&X)[3] = r.X, (&M0)[3] = r.M0, (&V0)[3] = r.V0, (&C)[3] = r.C, (&M1)[3] = r.M1, (&V1)[3] = r.V1, (&LS)[3] = r.LS, (&M4)[3] = r.M4, (&V4)[3] = r.V4, (&S)[3] = r.S;
    float4 &q = r.q, &M2 = r.M2, &V2 = r.V2;
    float &L = r.L, &M3 = r.M3, &V3 = r.V3;
    const float o = r.o, cnt = r.cnt;
    { // means: mc = X R^T + t  =>  dL/dX = dmc R
        const float d[3] = {fmaf(GX[2], T.r[6], fmaf(GX[1], T.r[3], GX[0] * T.r[0])), fmaf(GX[2], T.r[7], fmaf(GX[1], T.r[4], GX[0] * T.r[1])),
                            fmaf(GX[2], T.r[8], fmaf(GX[1], T.r[5], GX[0] * T.r[2]))};
#pragma unroll
        for (int k = 0; k < 3; k++) adam_one(X[k], d[k], M0[k], V0[k], u, 0);
    }
#pragma unroll
    for (int k = 0; k < 3; k++) adam_one(C[k], GC[k], M1[k], V1[k], u, 1);
    { // unit quaternion r = q / |q|: dL/dq = (dr - r (r . dr)) / |q|
        const float nn = fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f), inv = 1.f / nn;
        const float r0 = q.x * inv, r1 = q.y * inv, r2 = q.z * inv, r3 = q.w * inv;
        const float dot = r0 * dr.x + r1 * dr.y + r2 * dr.z + r3 * dr.w;
        adam_one(q.x, (dr.x - r0 * dot) * inv, M2.x, V2.x, u, 2); adam_one(q.y, (dr.y - r1 * dot) * inv, M2.y, V2.y, u, 2);
        adam_one(q.z, (dr.z - r2 * dot) * inv, M2.z, V2.z, u, 2); adam_one(q.w, (dr.w - r3 * dot) * inv, M2.w, V2.w, u, 2);
    }
    adam_one(L, gop * ((1.f - o) * o), M3, V3, u, 3); // opacity = sigmoid(logit)
    { // scale = exp(log scale), plus the regularisers' gradient (K_scale_reg_bwd with an upstream gradient of 1)
        float d[3] = {GS[0] * S[0], GS[1] * S[1], GS[2] * S[2]};
        if (u.reg_out) {
            const float wgt = (float)(S[0] > u.limit) + (float)(S[1] > u.limit) + (float)(S[2] > u.limit);
            if (wgt > 0.f) {
                int imax = 0, imin = 0;
                if (S[1] > S[imax]) imax = 1;
                if (S[2] > S[imax]) imax = 2;
                if (S[1] < S[imin]) imin = 1;
                if (S[2] < S[imin]) imin = 2;
                const float a = u.w_scalar * wgt, b = cnt > 0.f ? u.w_long * wgt / cnt : 0.f;
#pragma unroll
                for (int k = 0; k < 3; k++) d[k] += ((k == imax ? a + b : 0.f) - (k == imin ? b : 0.f)) * S[k];
            }
        }
#pragma unroll
        for (int k = 0; k < 3; k++) adam_one(LS[k], d[k], M4[k], V4[k], u, 4);
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
        u.xyz[3 * i + k] = X[k]; u.m[0][3 * i + k] = M0[k]; u.v[0][3 * i + k] = V0[k];
        u.rgb[3 * i + k] = C[k]; u.m[1][3 * i + k] = M1[k]; u.v[1][3 * i + k] = V1[k];
        u.ls[3 * i + k] = LS[k]; u.m[4][3 * i + k] = M4[k]; u.v[4][3 * i + k] = V4[k];
    }
    reinterpret_cast<float4*>(u.quat)[i] = q; reinterpret_cast<float4*>(u.m[2])[i] = M2; reinterpret_cast<float4*>(u.v[2])[i] = V2;
    u.logit[i] = L; u.m[3][i] = M3; u.v[3][i] = V3;
}
__device__ __forceinline__ void map_update_with(const size_t i, const MapUpdate& u, const Pose34& T, const float (&GX)[3], const float (&GC)[3],
                                                const float4 dr, const float gop, const float (&GS)[3])
{
    MapRegs r;
    map_load(i, u, r);
    map_apply(i, u, T, r, GX, GC, dr, gop, GS);
}
__device__ __forceinline__ void map_update_one(const size_t i, const MapUpdate& u, const Pose34& T)
{
    const float GX[3] = {u.dmc[3 * i], u.dmc[3 * i + 1], u.dmc[3 * i + 2]}, GC[3] = {u.dcol[3 * i], u.dcol[3 * i + 1], u.dcol[3 * i + 2]};
    const float GS[3] = {u.dscale[3 * i], u.dscale[3 * i + 1], u.dscale[3 * i + 2]};
    map_update_with(i, u, T, GX, GC, reinterpret_cast<const float4*>(u.drot)[i], u.dopac[i], GS);
}
// small maps: one Gaussian per thread (four per thread leaves a 10 k map with ten workgroups)
__global__ void __launch_bounds__(256)
K_map_update_small(size_t n, MapUpdate u)
{
    if (u.overflow && *u.overflow) return;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    map_update_one(i, u, load_pose(u.Tcw));
}

// Four consecutive splats per thread: every tensor is then read and written as whole float4s (a [n,3] tensor is 12 floats per
// thread: three aligned 16-byte accesses instead of twelve 4-byte ones at a stride of 12 bytes) — the update is pure HBM traffic,
// ~408 bytes per Gaussian (107 us -> see profiles/r04_loop.md at 1 M Gaussians).
template <int K>
__device__ __forceinline__ void ld4(const float* __restrict__ p, const size_t i4, float (&x)[4 * K])
{
#pragma unroll
    for (int q = 0; q < K; q++) {
        const float4 v = reinterpret_cast<const float4*>(p + K * i4)[q];
        x[4 * q] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
    }
}
template <int K>
__device__ __forceinline__ void st4(float* __restrict__ p, const size_t i4, const float (&x)[4 * K])
{
#pragma unroll
    for (int q = 0; q < K; q++) reinterpret_cast<float4*>(p + K * i4)[q] = make_float4(x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]);
}
__global__ void __launch_bounds__(256)
K_map_update(size_t n, MapUpdate u)
{
    if (u.overflow && *u.overflow) return;
    const size_t i4 = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i4 >= n) return;
    const Pose34 T = load_pose(u.Tcw);
    if (i4 + 4 > n) { // the last, partial group of four
        for (size_t i = i4; i < n; i++) map_update_one(i, u, T);
        return;
    }
    { // means
        float P[12], G[12], M[12], V[12];
        ld4<3>(u.xyz, i4, P); ld4<3>(u.dmc, i4, G); ld4<3>(u.m[0], i4, M); ld4<3>(u.v[0], i4, V);
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const float g0 = G[3 * s], g1 = G[3 * s + 1], g2 = G[3 * s + 2];
            const float d[3] = {fmaf(g2, T.r[6], fmaf(g1, T.r[3], g0 * T.r[0])), fmaf(g2, T.r[7], fmaf(g1, T.r[4], g0 * T.r[1])),
                                fmaf(g2, T.r[8], fmaf(g1, T.r[5], g0 * T.r[2]))};
#pragma unroll
            for (int k = 0; k < 3; k++) adam_one(P[3 * s + k], d[k], M[3 * s + k], V[3 * s + k], u, 0);
        }
        st4<3>(u.xyz, i4, P); st4<3>(u.m[0], i4, M); st4<3>(u.v[0], i4, V);
    }
    { // colours
        float P[12], G[12], M[12], V[12];
        ld4<3>(u.rgb, i4, P); ld4<3>(u.dcol, i4, G); ld4<3>(u.m[1], i4, M); ld4<3>(u.v[1], i4, V);
#pragma unroll
        for (int k = 0; k < 12; k++) adam_one(P[k], G[k], M[k], V[k], u, 1);
        st4<3>(u.rgb, i4, P); st4<3>(u.m[1], i4, M); st4<3>(u.v[1], i4, V);
    }
    { // quaternions
        float P[16], G[16], M[16], V[16];
        ld4<4>(u.quat, i4, P); ld4<4>(u.drot, i4, G); ld4<4>(u.m[2], i4, M); ld4<4>(u.v[2], i4, V);
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const float q0 = P[4 * s], q1 = P[4 * s + 1], q2 = P[4 * s + 2], q3 = P[4 * s + 3];
            const float nn = fmaxf(sqrtf(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3), 1e-12f), inv = 1.f / nn;
            const float r[4] = {q0 * inv, q1 * inv, q2 * inv, q3 * inv};
            const float dot = r[0] * G[4 * s] + r[1] * G[4 * s + 1] + r[2] * G[4 * s + 2] + r[3] * G[4 * s + 3];
#pragma unroll
            for (int k = 0; k < 4; k++) adam_one(P[4 * s + k], (G[4 * s + k] - r[k] * dot) * inv, M[4 * s + k], V[4 * s + k], u, 2);
        }
        st4<4>(u.quat, i4, P); st4<4>(u.m[2], i4, M); st4<4>(u.v[2], i4, V);
    }
    { // opacities
        float P[4], G[4], M[4], V[4], O[4];
        ld4<1>(u.logit, i4, P); ld4<1>(u.dopac, i4, G); ld4<1>(u.m[3], i4, M); ld4<1>(u.v[3], i4, V); ld4<1>(u.opac, i4, O);
#pragma unroll
        for (int k = 0; k < 4; k++) adam_one(P[k], G[k] * ((1.f - O[k]) * O[k]), M[k], V[k], u, 3);
        st4<1>(u.logit, i4, P); st4<1>(u.m[3], i4, M); st4<1>(u.v[3], i4, V);
    }
    { // scales
        float P[12], G[12], M[12], V[12], S[12];
        ld4<3>(u.ls, i4, P); ld4<3>(u.dscale, i4, G); ld4<3>(u.m[4], i4, M); ld4<3>(u.v[4], i4, V); ld4<3>(u.scales, i4, S);
        const float cnt = u.reg_out ? u.reg_out[0] : 0.f;
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const float sc[3] = {S[3 * s], S[3 * s + 1], S[3 * s + 2]};
            float d[3] = {G[3 * s] * sc[0], G[3 * s + 1] * sc[1], G[3 * s + 2] * sc[2]};
            if (u.reg_out) {
                const float wgt = (float)(sc[0] > u.limit) + (float)(sc[1] > u.limit) + (float)(sc[2] > u.limit);
                if (wgt > 0.f) {
                    int imax = 0, imin = 0;
                    if (sc[1] > sc[imax]) imax = 1;
                    if (sc[2] > sc[imax]) imax = 2;
                    if (sc[1] < sc[imin]) imin = 1;
                    if (sc[2] < sc[imin]) imin = 2;
                    const float a = u.w_scalar * wgt, b = cnt > 0.f ? u.w_long * wgt / cnt : 0.f;
#pragma unroll
                    for (int k = 0; k < 3; k++) d[k] += ((k == imax ? a + b : 0.f) - (k == imin ? b : 0.f)) * sc[k];
                }
            }
#pragma unroll
            for (int k = 0; k < 3; k++) adam_one(P[3 * s + k], d[k], M[3 * s + k], V[3 * s + k], u, 4);
        }
        st4<3>(u.ls, i4, P); st4<3>(u.m[4], i4, M); st4<3>(u.v[4], i4, V);
    }
}

// loss of a mapping iteration from its parts: pixel terms (K_loss_finish: sums[5]) + c_ssim * (1 - mean SSIM) + regularisers
__global__ void __launch_bounds__(GSR_FINISH_THREADS)
K_map_loss_total(const float* __restrict__ sums, const float* __restrict__ ssim_partial, int n_partial, float inv_count, float c_ssim,
                 const float* __restrict__ reg_out, const uint32_t* __restrict__ overflow, float* __restrict__ loss)
{
    __shared__ float ws[4][4];
    float a[4] = {0.f, 0.f, 0.f, 0.f}; // four independent chains per thread
    for (int b = threadIdx.x; b < n_partial; b += 4 * GSR_FINISH_THREADS) {
#pragma unroll
        for (int q = 0; q < 4; q++) a[q] += b + q * GSR_FINISH_THREADS < n_partial ? ssim_partial[b + q * GSR_FINISH_THREADS] : 0.f;
    }
    finish_sum<4>(a, ws);
    const float t = (a[0] + a[1]) + (a[2] + a[3]);
    if (threadIdx.x == 0) // (an iteration whose forward overflowed its workspace rendered nothing: NaN, and K_map_update skipped its step)
        loss[0] = (overflow && *overflow) ? __builtin_nanf("") : sums[5] + c_ssim * (1.f - t * inv_count) + (reg_out ? reg_out[3] : 0.f);
}

// The one single-workgroup kernel between the mapping loss's two passes: adds the rows of K_ssim_fwd<true> (six sums per workgroup) and of
// K_map_prepare (the regularisers' three), forms sums[8] as K_loss_finish (mode 1), reg_out[4] as K_scale_reg_finish and the iteration's loss as
// K_map_loss_total — three launches of a few microseconds of latency each before.
struct MapFinish {
    const float* partial6; int n6;
    const float* reg_partial; int n_reg;    // nullptr / 0: no regularisers
    float inv_pixels3, inv_count_ssim;      // 1 / (3 H W) twice over: the colour L1 mean and the SSIM mean
    float w[3], c_ssim, w_long, w_scalar;
    const uint32_t* overflow;
    float *sums, *reg_out, *loss;
};
#define GSR_MAP_FINISH_THREADS 1024
__global__ void __launch_bounds__(GSR_MAP_FINISH_THREADS)
K_map_finish(MapFinish m)
{
    constexpr int NT = GSR_MAP_FINISH_THREADS, NW = NT / 64;
    __shared__ float ws[NW][9];
    float a[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}; // six sums of the loss rows, three of the regularisers'
    // (1024 threads and every load of a trip requested before the first is used: one workgroup has to pull 0.3 MB through its own latency —
    // 256 threads with one load outstanding per accumulator took 20 us at 1200x680)
    // (five / four rows per thread and trip: the 9 675 loss rows and the 3 907 regulariser rows of a 1 M-Gaussian map at 1200x680 are ONE trip
    // to memory each — two rows per trip were five + two dependent trips: 10.6 us)
    constexpr int U6 = 5, U3 = 4;
    for (int b0 = 0; b0 < m.n6; b0 += U6 * NT) {
        float t[U6][6];
#pragma unroll
        for (int u = 0; u < U6; u++) {
            const int b = b0 + u * NT + (int)threadIdx.x;
#pragma unroll
            for (int q = 0; q < 6; q++) t[u][q] = b < m.n6 ? m.partial6[(size_t)q * m.n6 + b] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U6; u++)
#pragma unroll
            for (int q = 0; q < 6; q++) a[q] += t[u][q];
    }
    for (int b0 = 0; b0 < m.n_reg; b0 += U3 * NT) {
        float t[U3][3];
#pragma unroll
        for (int u = 0; u < U3; u++) {
            const int b = b0 + u * NT + (int)threadIdx.x;
#pragma unroll
            for (int q = 0; q < 3; q++) t[u][q] = b < m.n_reg ? m.reg_partial[(size_t)b * 3 + q] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U3; u++)
#pragma unroll
            for (int q = 0; q < 3; q++) a[6 + q] += t[u][q];
    }
#pragma unroll
    for (int q = 0; q < 9; q++) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) a[q] += __shfl_xor(a[q], off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int q = 0; q < 9; q++) ws[threadIdx.x >> 6][q] = a[q];
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    float r[3];
#pragma unroll
    for (int q = 0; q < 9; q++) {
        float t = 0.f;
#pragma unroll
        for (int wv = 0; wv < NW; wv++) t += ws[wv][q];
        if (q < 6) a[q] = t; else r[q - 6] = t;
    }
    const float pix = m.w[0] * (a[1] * m.inv_pixels3) + m.w[1] * (a[2] / fmaxf(a[3], 1.f)) + m.w[2] * (a[4] / fmaxf(a[5], 1.f));
    m.sums[0] = a[1]; m.sums[1] = a[2]; m.sums[2] = a[3]; m.sums[3] = a[4]; m.sums[4] = a[5]; m.sums[5] = pix; m.sums[6] = a[0]; m.sums[7] = 0.f;
    float reg = 0.f;
    if (m.reg_partial) {
        reg = m.w_long * (r[0] > 0.f ? r[2] / r[0] : 0.f) + m.w_scalar * r[1];
        m.reg_out[0] = r[0]; m.reg_out[1] = r[1]; m.reg_out[2] = r[2]; m.reg_out[3] = reg;
    }
    m.loss[0] = (m.overflow && *m.overflow) ? __builtin_nanf("") : pix + m.c_ssim * (1.f - a[0] * m.inv_count_ssim) + reg;
}

// End of a tracking iteration, one workgroup. state[16]: {quat[4], trans[3], 0, m_quat[4], m_trans[3], 0}; state2[8]: {v_quat[4], v_trans[3], 0};
// best[8]: {best loss, quat[4], trans[3]}; history[it] = the iteration's loss. Gaussian.cc:144-150 (both groups with one learning rate),
// Render.cc:1107-1118 (the pose of the lowest loss is kept; the step that follows an iteration is taken before the host looks at its loss).
struct PoseUpdate {
    float* quat_trans;  // [7] the pose parameters (un-normalised quaternion r,x,y,z; translation), updated in place
    float* moments;     // [14] exp_avg[7], exp_avg_sq[7]
    float* best;        // [8] lowest loss so far and the pose that produced it
    float* history;     // this iteration's slot
    float* Tcw;         // [16] rewritten for the next iteration
    const float* partial; // [GSR_POSE_BLOCKS][12] K_pose_grad's rows
    const float* loss;    // the iteration's loss (K_loss_finish: sums + 5)
    const uint32_t* overflow;
    float* skip;          // one float (nullptr: none): non-zero = some rank's forward overflowed (the sharded loop's all-reduced flag)
    float w1, b2, w2, eps, step_size, sqrt_bias2;
};
// presum: the twelve sums already added up (K_splat_bwd_pose's last workgroup), or nullptr: this wave adds the nrows rows of u.partial
template <bool COHERENT>
__device__ void pose_update_body(const PoseUpdate& u, const int nrows, const float* presum)
{
    const int lane = threadIdx.x; // (one wave: the threads 0..63 of the workgroup)
    // everything the single-thread tail needs is requested before the partial rows are summed (the tail's loads and stores may alias as
    // far as the compiler knows: left where they are used, they are ~40 dependent round trips: 9.5 us for a kernel that moves 25 KB)
    float pq[7], mm[7], vv[7];
#pragma unroll
    for (int k = 0; k < 7; k++) { pq[k] = u.quat_trans[k]; mm[k] = u.moments[k]; vv[k] = u.moments[7 + k]; }
    const float best0 = u.best[0], loss_in = u.loss[0];
    const bool skip = (u.overflow && *u.overflow) || (u.skip && *u.skip != 0.f);
    float a[12] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int b = lane; b < nrows; b += 64) {
#pragma unroll
        for (int q = 0; q < 12; q++) a[q] += COHERENT ? __hip_atomic_load(u.partial + b * 12 + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : u.partial[b * 12 + q];
    }
#pragma unroll
    for (int q = 0; q < 12; q++) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) a[q] += __shfl_xor(a[q], off, 64);
    }
    if (presum) {
#pragma unroll
        for (int q = 0; q < 12; q++) a[q] = presum[q];
    }
    if (lane != 0) return;
    float q[4] = {pq[0], pq[1], pq[2], pq[3]}, t[3] = {pq[4], pq[5], pq[6]};
    const float lv = (skip || loss_in != loss_in) ? __builtin_nanf("") : loss_in; // (one NaN pattern only: the loop that spins on the slot keeps another one for "not posted yet")
    __hip_atomic_store(u.history, lv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); // (the slot may be host memory the loop spins on: DirectLoop.cpp)
    if (lv == lv && lv < best0) {
        u.best[0] = lv;
#pragma unroll
        for (int k = 0; k < 4; k++) u.best[1 + k] = q[k];
#pragma unroll
        for (int k = 0; k < 3; k++) u.best[5 + k] = t[k];
    }
    if (!skip) {
        // rt2T backward (K_rt2T_bwd) from dL/dR = a[0..8] (row-major), dL/dt = a[9..11]
        const float n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
        const float r = q[0] / n, x = q[1] / n, y = q[2] / n, z = q[3] / n;
        const float G00 = a[0], G01 = a[1], G02 = a[2], G10 = a[3], G11 = a[4], G12 = a[5], G20 = a[6], G21 = a[7], G22 = a[8];
        const float dr = 2.f * (-z * G01 + y * G02 + z * G10 - x * G12 - y * G20 + x * G21);
        const float dx = 2.f * (y * G01 + z * G02 + y * G10 - 2.f * x * G11 - r * G12 + z * G20 + r * G21 - 2.f * x * G22);
        const float dy = 2.f * (-2.f * y * G00 + x * G01 + r * G02 + x * G10 + z * G12 - r * G20 + z * G21 - 2.f * y * G22);
        const float dz = 2.f * (-2.f * z * G00 - r * G01 + x * G02 + r * G10 - 2.f * z * G11 + y * G12 + x * G20 + y * G21);
        const float dot = r * dr + x * dx + y * dy + z * dz;
        const float g[7] = {(dr - r * dot) / n, (dx - x * dot) / n, (dy - y * dot) / n, (dz - z * dot) / n, a[9], a[10], a[11]};
#pragma unroll
        for (int k = 0; k < 7; k++) {
            float p = k < 4 ? q[k] : t[k - 4];
            mm[k] = fmaf(u.w1, g[k] - mm[k], mm[k]);
            vv[k] = fmaf(u.w2 * g[k], g[k], vv[k] * u.b2);
            p = fmaf(-u.step_size, mm[k] / (sqrtf(vv[k]) / u.sqrt_bias2 + u.eps), p);
            if (k < 4) q[k] = p; else t[k - 4] = p;
        }
#pragma unroll
        for (int k = 0; k < 7; k++) { u.moments[k] = mm[k]; u.moments[7 + k] = vv[k]; u.quat_trans[k] = k < 4 ? q[k] : t[k - 4]; }
    }
    { // the next iteration's pose matrix (K_rt2T)
        const float n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
        const float r = q[0] / n, x = q[1] / n, y = q[2] / n, z = q[3] / n;
        float* const T = u.Tcw;
        T[0] = 1.f - 2.f * (y * y + z * z); T[1] = 2.f * (x * y - r * z); T[2] = 2.f * (x * z + r * y); T[3] = t[0];
        T[4] = 2.f * (x * y + r * z); T[5] = 1.f - 2.f * (x * x + z * z); T[6] = 2.f * (y * z - r * x); T[7] = t[1];
        T[8] = 2.f * (x * z - r * y); T[9] = 2.f * (y * z + r * x); T[10] = 1.f - 2.f * (x * x + y * y); T[11] = t[2];
        T[12] = 0.f; T[13] = 0.f; T[14] = 0.f; T[15] = 1.f;
    }
}
__global__ void __launch_bounds__(64)
K_pose_update(PoseUpdate u)
{
    pose_update_body<false>(u, GSR_POSE_BLOCKS);
}
// gsr_pose_grad and gsr_pose_update in one launch (the last workgroup of the sums takes the step)
__global__ void __launch_bounds__(256)
K_pose_step(const float* __restrict__ X, const float* __restrict__ dmc, size_t n, float* partial, uint32_t* ticket, PoseUpdate u)
{
    pose_grad_body<true>(X, dmc, n, u.Tcw, partial, nullptr, ticket, &u);
}

} // namespace gsr
