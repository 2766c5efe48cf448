// gsr_kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels of the differentiable
// Gaussian-splat rasterizer. Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off.
//
// Forward  (replaces rasterizer_impl.cu:199-345 of the reference's DGR tree):
//   K_preprocess   per splat : project, cull, radius, tile rectangle, colour, reach word
//   K_bin_count    per range : LDS histogram of the tiles a contiguous range of splats covers -> one row of the
//                              count matrix (gsr_device.h)
//   K_bin_colscan  per column: exclusive scan down the matrix columns, column totals = tile counts
//   (tile scan)    last workgroup of K_bin_colscan: counts -> segment starts, ranges, num_rendered, overflow flag
//   K_bin_fill     per range : (depth bits<<32 | id) into the list slot an LDS cursor hands out (no global atomics)
//   K_tile_sort_all per tile : one launch: bucket sort by depth in LDS with exact in-bin ranking (one wave per short
//                              list, 256 threads for the queued longer ones, lists over 4096 through global scratch),
//                              bitonic network only for lists of tied depths -> point_list
//                              ((depth, id) ascending == the reference's stable radix order)
//                              (K_tile_sort<SMALL>: the same code launched per bucket by the k-NN path)
//   K_blend_fwd    per quad  : front-to-back alpha blend, one wave per 8x8 quad, four independent 4x4 patch
//                              rows per wave, exact culling (small splats: bit shifts on the reach word K_preprocess
//                              left in col.w), logs its hits for the backward (gsr_blend.h)
// Backward (replaces rasterizer_impl.cu:405-498):
//   K_blend_bwd    per quad  : back-to-front walk of the forward's log; the per-pixel loop parks (u, dcol) in an LDS
//                              ring, every 16 iterations one lane per (row, iteration) pair forms the nine sums and
//                              lane e merges the sums of entry e; one 9-lane atomic per (quad, splat) (gsr_blend.h)
//   K_splat_bwd    per splat : conic/mean2D/colour gradients -> mean3D, cov3D, scale, rot, SH
//
// No global sort and no (tile | depth) keys: per-tile counting replaces the reference's 64-bit radix sort of
// R instances (6 passes over 24 B/instance) by one 8 B/instance write and one LDS-resident sort per tile.
#include "gsr_device.h"
#include "gsr_splat_math.h"
#include "gsr_blend.h"
#include "gsr_train.h"

namespace gsr {

// ===================================================================================
// per-splat forward
// ===================================================================================
struct SplatInputs {
    const float* means3D;
    const float* scales;
    const float* rotations;
    const float* opacities;
    const float* shs;
    const float* cov3D_precomp;
    const float* colors_precomp;
    const float* view;
    const float* proj;
    const float* campos;
    const float* pre_Tcw;   // (NULL, or: means3D are world means, moved into the camera frame here — gsr_forward_args.pre_Tcw)
    float* means_cam_out;
    // RAW (gsr_forward_args.raw; K_preprocess<true>): opacities / scales / rotations above hold the RAW parameters (logits, log-scales, un-normalised
    // quaternions); the kernel applies gsr_map_prepare's activations, stores what the backward takes and writes the scale regularisers' partial sums
    float *opac_out, *scales_out, *rots_out;
    float reg_limit;
    float* reg_partial;
};

__device__ __forceinline__ void load_cov3d(const SplatInputs& in, const FrameParams& f, int idx, float cov[6])
{
    if (in.cov3D_precomp) {
#pragma unroll
        for (int k = 0; k < 6; k++) cov[k] = in.cov3D_precomp[6 * (size_t)idx + k];
    } else {
        const float3 s = make_float3(in.scales[3 * (size_t)idx], in.scales[3 * (size_t)idx + 1], in.scales[3 * (size_t)idx + 2]);
        const float4 q = reinterpret_cast<const float4*>(in.rotations)[idx];
        cov3d_from_scale_rot(s, f.scale_modifier, q, cov);
    }
}

#define GSR_PRE_THREADS GSR_BIN_PIECE // one workgroup = one piece of the bucketed bin records (gsr_device.h)
// Every per-splat input is requested before the first one is used (round 4): written in the order the reference computes, a
// wave made five dependent trips to memory — means; scales + rotations; the view matrix; the projection matrix behind the
// near-plane test; colour + opacity behind the visibility test (a compiler does not move a load above the branch that guards
// it). pin() keeps the requests where they are written. What the kernel's time is made of (round 4, variant builds; HIP-event
// pairs, ~2 us above rocprof's figure): loads alone 16 us, loads + stores without
// the arithmetic 25.5 (132 MB: 5.2 TB/s), loads + arithmetic without the stores 25, everything 32 — the three phases of a wave
// add up instead of overlapping (2.5 rounds of waves that start together), and neither the exact reach test (44 % of the
// instructions: 32.3 -> 32.3 us without it) nor the order of the requests (33.1 -> 32.3) is what it waits for. Two and four splats
// per thread, software-pipelined (the next splat's inputs requested before the current one's arithmetic): 33.1 / 35.6 us. Held to 72 / 64
// VGPRs for seven / eight waves per SIMD instead of six (it spills 44 / 72 bytes): 41.9 / 82 us.
__device__ __forceinline__ void pin(float& x) { asm volatile("" : "+v"(x)); }
// one splat: everything but its bin record, which is returned ({tiles of the band-clipped rectangle, depth bits, x0 | y0 << 16, x1 | y1 << 16})
template <bool RAW>
__device__ __forceinline__ uint4 preprocess_splat(const int idx, const FrameParams& f, const SplatInputs& in, int* __restrict__ radii_out, const GeomView& g, float (&reg)[3])
{
    float3 p = make_float3(in.means3D[3 * (size_t)idx], in.means3D[3 * (size_t)idx + 1], in.means3D[3 * (size_t)idx + 2]);
    float cov[6];
    float3 sc = make_float3(0.f, 0.f, 0.f);
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
    if (in.cov3D_precomp) {
#pragma unroll
        for (int k = 0; k < 6; k++) cov[k] = in.cov3D_precomp[6 * (size_t)idx + k];
    } else {
        sc = make_float3(in.scales[3 * (size_t)idx], in.scales[3 * (size_t)idx + 1], in.scales[3 * (size_t)idx + 2]);
        q = reinterpret_cast<const float4*>(in.rotations)[idx];
    }
    float opac = in.opacities[idx];
    float3 cp = make_float3(0.f, 0.f, 0.f);
    if (in.colors_precomp) cp = make_float3(in.colors_precomp[3 * (size_t)idx], in.colors_precomp[3 * (size_t)idx + 1], in.colors_precomp[3 * (size_t)idx + 2]);
    float vm[16], pm[16]; // the two matrices: scalar loads, requested with the rest
#pragma unroll
    for (int k = 0; k < 16; k++) { vm[k] = in.view[k]; pm[k] = in.proj[k]; }
    float T[12] = {1.f, 0.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 0.f, 1.f, 0.f};
    if (in.pre_Tcw) {
#pragma unroll
        for (int k = 0; k < 12; k++) T[k] = in.pre_Tcw[k];
    }
    pin(p.x); pin(p.y); pin(p.z); pin(opac); pin(cp.x); pin(cp.y); pin(cp.z);
    if (in.pre_Tcw) { // mc = X R^T + t with K_to_camera's arithmetic (gsr_train.h): the mean the rest of the pipeline sees, and the backward's
        const float x = p.x, y = p.y, z = p.z;
        p.x = fmaf(T[2], z, fmaf(T[1], y, T[0] * x)) + T[3];
        p.y = fmaf(T[6], z, fmaf(T[5], y, T[4] * x)) + T[7];
        p.z = fmaf(T[10], z, fmaf(T[9], y, T[8] * x)) + T[11];
        in.means_cam_out[3 * (size_t)idx] = p.x; in.means_cam_out[3 * (size_t)idx + 1] = p.y; in.means_cam_out[3 * (size_t)idx + 2] = p.z;
    }
    if (in.cov3D_precomp) {
#pragma unroll
        for (int k = 0; k < 6; k++) pin(cov[k]);
    } else {
        pin(sc.x); pin(sc.y); pin(sc.z); pin(q.x); pin(q.y); pin(q.z); pin(q.w);
        if (RAW) { // K_map_prepare's activations, operation for operation (gsr_train.h): sigmoid, exp, torch's normalize; the regularisers' three sums
            opac = 1.f / (1.f + expf(-opac));
            sc = make_float3(expf(sc.x), expf(sc.y), expf(sc.z));
            const float inv = 1.f / fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);
            q = make_float4(q.x * inv, q.y * inv, q.z * inv, q.w * inv);
            in.opac_out[idx] = opac;
            in.scales_out[3 * (size_t)idx] = sc.x; in.scales_out[3 * (size_t)idx + 1] = sc.y; in.scales_out[3 * (size_t)idx + 2] = sc.z;
            reinterpret_cast<float4*>(in.rots_out)[idx] = q;
            const float wgt = (float)(sc.x > in.reg_limit) + (float)(sc.y > in.reg_limit) + (float)(sc.z > in.reg_limit);
            const float mx = fmaxf(sc.x, fmaxf(sc.y, sc.z)), mn = fminf(sc.x, fminf(sc.y, sc.z));
            reg[0] = wgt; reg[1] = wgt * (mx - in.reg_limit); reg[2] = wgt * (mx - mn);
        }
        cov3d_from_scale_rot(sc, f.scale_modifier, q, cov);
    }
    Projected pr;
    const bool vis = project_splat(p, cov, f, vm, pm, pr);
    if (!vis) {
        g.g1[idx] = make_float4(0.f, 0.f, 0.f, 0.f); // radius 0 marks the splat invisible
        if (radii_out) radii_out[idx] = 0;
        return make_uint4(0u, 0u, 0u, 0u);
    }
    float4 c;
    if (in.colors_precomp) {
        c = make_float4(cp.x, cp.y, cp.z, 0.f);
    } else {
        float3 raw;
        const float3 d = unit_dir(p, in.campos, raw);
        const float* sh = in.shs + (size_t)idx * f.M * 3;
        const float r = sh_channel(f.D, sh, 0, d), gg = sh_channel(f.D, sh, 1, d), b = sh_channel(f.D, sh, 2, d);
        const uint32_t flags = (r < 0 ? 1u : 0u) | (gg < 0 ? 2u : 0u) | (b < 0 ? 4u : 0u);
        c = make_float4(fmaxf(r, 0.f), fmaxf(gg, 0.f), fmaxf(b, 0.f), __uint_as_float(flags));
    }
    // the exact patch reach of a small splat, once per splat instead of once per (tile entry, quad) in the blend (gsr_device.h)
    const uint32_t reach = splat_reach25(pr.px, pr.py, pr.conic_a, pr.conic_b, pr.conic_c, opac);
    g.reach[idx] = reach_entry(reach, pr.px, pr.py);
    g.g0[idx] = make_float4(pr.px, pr.py, pr.conic_a, pr.conic_b);
    g.g1[idx] = make_float4(pr.conic_c, opac, pr.p_view.z, __int_as_float(pr.radius));
    g.col[idx] = c;
    if (radii_out) radii_out[idx] = pr.radius;
    // only the band's tile rows are binned
    pr.y0 = max(pr.y0, f.band_y0); pr.y1 = max(pr.y0, min(pr.y1, f.band_y1));
    return make_uint4((uint32_t)((pr.x1 - pr.x0) * (pr.y1 - pr.y0)), __float_as_uint(pr.p_view.z), (uint32_t)pr.x0 | ((uint32_t)pr.y0 << 16),
                      (uint32_t)pr.x1 | ((uint32_t)pr.y1 << 16));
}

// The bin records leave BUCKETED by tile window (round 5): the workgroup's records of window w go, densely, to the piece (workgroup, w) of the
// pool, a splat whose rectangle crosses a window boundary to both pieces; the binning passes of window w then read what concerns them and
// nothing else (gsr_device.h). A returning LDS atomic hands out the slot, one barrier before the eight counts are stored: no global atomic.
template <bool RAW>
__global__ void __launch_bounds__(GSR_PRE_THREADS) __attribute__((amdgpu_waves_per_eu(RAW ? 5 : 6, 8))) // (six waves per SIMD: 80 registers — what the kernel had before it became a template; the variant with the activations takes 90)
K_preprocess(FrameParams f, SplatInputs in, int* __restrict__ radii_out, GeomView g)
{
    __shared__ uint32_t s_cnt[GSR_BIN_NWIN];
    const int idx = blockIdx.x * GSR_PRE_THREADS + threadIdx.x;
    if (threadIdx.x < GSR_BIN_NWIN) s_cnt[threadIdx.x] = 0u;
    if (idx == 0) g.hdr->ticket = 0u; // (K_bin_colscan's arrival counter: a launch boundary lies between this store and its first use)
    __syncthreads();
    float reg[3] = {0.f, 0.f, 0.f};
    const uint4 br = idx < f.P ? preprocess_splat<RAW>(idx, f, in, radii_out, g, reg) : make_uint4(0u, 0u, 0u, 0u);
    if (br.x != 0u) {
        const int T = f.grid_x * f.grid_y;
        const int x0 = (int)(br.z & 0xFFFFu), y0 = (int)(br.z >> 16), x1 = (int)(br.w & 0xFFFFu), y1 = (int)(br.w >> 16);
        const int w0 = tile_window(y0 * f.grid_x + x0, T), w1 = tile_window((y1 - 1) * f.grid_x + x1 - 1, T);
        const uint4 rec = make_uint4((uint32_t)idx, br.y, br.z, br.w);
        for (int w = w0; w <= w1; w++) {
            // (a window between the first and the last holds a tile of the rectangle unless it is narrower than a tile row: the exact test keeps
            // every record worth at least one tile instance)
            if (w != w0 && w != w1 && !window_meets_rect(w, T, f.grid_x, x0, y0, x1, y1)) continue;
            const uint32_t slot = __hip_atomic_fetch_add(&s_cnt[w], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            g.pool[((size_t)blockIdx.x * GSR_BIN_NWIN + w) * GSR_BIN_PIECE + slot] = rec;
        }
    }
    __shared__ float s_reg[GSR_PRE_THREADS / 64][3];
    if (RAW && in.reg_partial) { // K_map_prepare's row of three sums per 256 Gaussians (same order of additions: a butterfly per wave, the four waves pairwise)
#pragma unroll
        for (int q = 0; q < 3; q++) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) reg[q] += __shfl_xor(reg[q], off, 64);
        }
        if ((threadIdx.x & 63) == 0) {
#pragma unroll
            for (int q = 0; q < 3; q++) s_reg[threadIdx.x >> 6][q] = reg[q];
        }
    }
    __syncthreads();
    if (threadIdx.x < GSR_BIN_NWIN) g.pcnt[(size_t)blockIdx.x * GSR_BIN_NWIN + threadIdx.x] = (uint16_t)s_cnt[threadIdx.x];
    if (RAW && in.reg_partial && threadIdx.x < 3)
        in.reg_partial[(size_t)blockIdx.x * 3 + threadIdx.x] = (s_reg[0][threadIdx.x] + s_reg[1][threadIdx.x]) + (s_reg[2][threadIdx.x] + s_reg[3][threadIdx.x]);
}

__global__ void __launch_bounds__(256)
K_filter_radii(FrameParams f, SplatInputs in, int* __restrict__ radii_out)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= f.P) return;
    const float3 p = make_float3(in.means3D[3 * (size_t)idx], in.means3D[3 * (size_t)idx + 1], in.means3D[3 * (size_t)idx + 2]);
    float cov[6];
    load_cov3d(in, f, idx, cov);
    Projected pr;
    radii_out[idx] = project_splat(p, cov, f, in.view, in.proj, pr) ? pr.radius : 0;
}

__global__ void __launch_bounds__(256)
K_mark_visible(int P, const float* __restrict__ means3D, const float* __restrict__ view, uint8_t* __restrict__ present)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= P) return;
    const float3 p = make_float3(means3D[3 * (size_t)idx], means3D[3 * (size_t)idx + 1], means3D[3 * (size_t)idx + 2]);
    present[idx] = xform4x3(p, view).z <= 0.2f ? 0 : 1;
}

// ===================================================================================
// tile binning
// ===================================================================================
// The two passes over the splats (gsr_device.h) share the walk: one workgroup per contiguous range of `per` splats
// and tile window; a rectangle of up to 16 tiles is walked by its own lane, larger ones are taken one at a time by
// the whole wave (a fat splat would otherwise hold 63 lanes idle for its whole walk). hit(window tile, key) is
// called for every (splat, tile of the window).
struct BinWindow {
    int brow, wi, t0, tw, wy0, wy1;
};
__device__ __forceinline__ BinWindow bin_window(int T, int grid_x, int wx, int nwin)
{
    // consecutive workgroup ids go to consecutive XCDs: with wx = 8 all workgroups of one XCD work on the same eighth of the tiles
    BinWindow w;
    const int wi = blockIdx.x % wx + wx * blockIdx.y;
    w.wi = wi;
    w.brow = blockIdx.x / wx;
    w.t0 = (int)((int64_t)T * wi / nwin);
    w.tw = (int)((int64_t)T * (wi + 1) / nwin) - w.t0;
    w.wy0 = w.t0 / grid_x; w.wy1 = (w.t0 + w.tw + grid_x - 1) / grid_x; // tile rows the window touches
    return w;
}
template <int THREADS, typename Hit>
__device__ __forceinline__ void bin_walk(int P, int per, int grid_x, const BinWindow& w, const GeomView& g, Hit hit)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = THREADS / 64;
    // the pieces of this range that belong to the window: one per K_preprocess workgroup, a wave each, densely packed — the records of the
    // other seven windows are never touched (round 4 read all 4096 records of the range in each of the eight window workgroups: 125 MB of the
    // fill pass's 149 MB of fabric traffic)
    const int j0 = (int)(((int64_t)w.brow * per) / GSR_BIN_PIECE), j1 = min((P + GSR_BIN_PIECE - 1) / GSR_BIN_PIECE, (int)(((int64_t)(w.brow + 1) * per) / GSR_BIN_PIECE));
    for (int j = j0 + wave; j < j1; j += nwaves) {
        const size_t piece = (size_t)j * GSR_BIN_NWIN + w.wi;
        const int n = (int)g.pcnt[piece];
        const uint4* __restrict__ recs = g.pool + piece * GSR_BIN_PIECE;
        for (int s0 = 0; s0 < n; s0 += 64) { // whole waves make the same number of trips
            const int s = s0 + lane;
            const uint4 br = s < n ? recs[s] : make_uint4(0u, 0u, 0u, 0u);
            const int x0 = (int)(br.z & 0xFFFFu), y0 = (int)(br.z >> 16), x1 = (int)(br.w & 0xFFFFu), y1 = (int)(br.w >> 16);
            const int ntl = (x1 - x0) * (y1 - y0);
            const uint64_t key = ((uint64_t)br.y << 32) | br.x;
            const bool wide = ntl > 16 && y0 < w.wy1 && y1 > w.wy0;
            if (ntl != 0 && ntl <= 16)
                for (int y = max(y0, w.wy0); y < min(y1, w.wy1); y++)
                    for (int x = x0; x < x1; x++) {
                        const int t = y * grid_x + x - w.t0;
                        if ((uint32_t)t < (uint32_t)w.tw) hit(t, key);
                    }
            uint64_t todo = __ballot(wide);
            while (todo) { // a rectangle of more than 16 tiles is walked by the whole wave (a fat splat would otherwise hold 63 lanes idle)
                const int l = __builtin_ctzll(todo);
                todo &= todo - 1;
                const int fx0 = __shfl(x0, l, 64), fy0 = __shfl(y0, l, 64), fw = __shfl(x1, l, 64) - fx0;
                const int n2 = __shfl(ntl, l, 64);
                const uint64_t fkey = ((uint64_t)(uint32_t)__shfl((int)br.y, l, 64) << 32) | (uint32_t)__shfl((int)br.x, l, 64);
                for (int k = lane; k < n2; k += 64) {
                    const int dy = k / fw;
                    const int t = (fy0 + dy) * grid_x + fx0 + (k - dy * fw) - w.t0;
                    if ((uint32_t)t < (uint32_t)w.tw) hit(t, fkey);
                }
            }
        }
    }
}
__device__ __forceinline__ uint32_t lds_take(uint32_t* p)
{
    return __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// Count pass: histogram of the window's tiles in LDS -> the range's row of the count matrix (its columns of the window).
// (Round 5, measured and dropped: the column scan and the tile scan INSIDE this launch, handed over by "who arrived last" tickets three levels deep —
// per (window, group of 16 ranges), per window, per frame; nobody waits for anybody, the crossing data written through and read at agent scope:
// 33.6 us for the one launch against 11.3 + 9.4 for count + K_bin_colscan. A device-scope ticket and the agent-scope reads behind it cost ~2 us of
// latency each on this chip and three levels put nine of them in a row: a launch boundary is cheaper.)
// (Measured and dropped: K_preprocess and this pass in one launch, 1024-thread workgroups projecting four splats per
// thread and counting as they go — 38 us against 28.5 + 7.5 us for the two launches. Round 5, on the bucketed records: one workgroup per
// splat range that takes the eight windows' pieces one after the other, the whole matrix row in LDS — 16.1 / 20.6 us at 1024 / 512 threads
// against 11.5 for one workgroup per (range, window).)
__global__ void __launch_bounds__(GSR_BINC_THREADS)
K_bin_count(int P, int per, int T, int grid_x, int wx, int nwin, GeomView g, uint32_t* __restrict__ binmat)
{
    extern __shared__ uint32_t s_bin[];
    const BinWindow w = bin_window(T, grid_x, wx, nwin);
    for (int t = threadIdx.x; t < w.tw; t += GSR_BINC_THREADS) s_bin[t] = 0u;
    __syncthreads();
    bin_walk<GSR_BINC_THREADS>(P, per, grid_x, w, g, [&](int t, uint64_t) { (void)lds_take(&s_bin[t]); });
    __syncthreads();
    uint32_t* const row = binmat + (size_t)w.brow * T + w.t0;
    for (int t = threadIdx.x; t < w.tw; t += GSR_BINC_THREADS) row[t] = s_bin[t];
}

// Fill pass: the LDS words start as the range's cursors (segment start of the tile + instances of earlier ranges);
// every (splat, tile) takes its slot with a returning LDS atomic and stores its key there. (Sorting the keys of a
// workgroup by tile in LDS first, so that a wave stores to as few lines as possible, was measured slower — 36.6 vs
// 28.5 us at 1 M splats: a (range, window) pair holds ~3 keys per tile, no run worth coalescing.)
// (Non-temporal stores: 89 us — the keys of a segment do merge in L2 on the normal path.)
// (Round 5, measured and dropped: the first records of a wave's pieces requested before the cursors' LDS set-up and its barrier — 18.5 vs 18.2 us.)
__global__ void __launch_bounds__(GSR_BINF_THREADS)
K_bin_fill(int P, int per, int T, int grid_x, int wx, int nwin, GeomView g, const uint32_t* __restrict__ binmat,
           const uint32_t* __restrict__ tile_start, uint64_t* __restrict__ pairs)
{
    extern __shared__ uint32_t s_bin[];
    if (g.hdr->overflow) return;
    const BinWindow w = bin_window(T, grid_x, wx, nwin);
    const uint32_t* const row = binmat + (size_t)w.brow * T + w.t0;
    for (int t = threadIdx.x; t < w.tw; t += GSR_BINF_THREADS) s_bin[t] = tile_start[w.t0 + t] + row[t];
    __syncthreads();
    bin_walk<GSR_BINF_THREADS>(P, per, grid_x, w, g, [&](int t, uint64_t key) { pairs[lds_take(&s_bin[t])] = key; });
}

// One block: per-tile counts -> list segments (start), ranges, num_rendered and the overflow flag. Shared with the
// k-NN path (buckets instead of tiles; its counters sit in padded records, hence the strides, in words).
// (The rasterizer runs this scan inside K_bin_colscan's last workgroup since round 4 — write-through stores and a ticket instead of the
// release / acquire fences whose L2 write-back cost the first attempt 45 us; this kernel serves the k-NN path and -DGSR_SEPARATE_SCAN.)
// COHERENT: the counts were written by other workgroups of the SAME launch with write-through (sc1) stores: read them with agent-scope
// loads (K_bin_colscan's last workgroup, above).
template <bool COHERENT>
__device__ __forceinline__ void scan_tiles_body(int T, const uint32_t* cnt, int cnt_stride, uint32_t* __restrict__ start, int start_stride,
                                                uint2* __restrict__ ranges, GeomHeader* __restrict__ hdr, uint32_t capacity)
{
    auto ld = [&](size_t i) -> uint32_t {
        return COHERENT ? __hip_atomic_load(cnt + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : cnt[i];
    };
    // each thread owns `per` consecutive tiles (its counts stay in registers when per <= 8), the block
    // scan is one shuffle scan per wave plus one over the 16 wave totals: two barriers in all
    __shared__ uint32_t wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int per = (T + 1023) / 1024;
    const int b = tid * per, e = min(T, b + per);
    uint32_t ks[8];
    uint32_t s = 0;
    if (per <= 8) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            ks[j] = (j < per && b + j < e) ? ld((size_t)(b + j) * cnt_stride) : 0u;
            s += ks[j];
        }
    } else {
        for (int i = b; i < e; i++) s += ld((size_t)i * cnt_stride);
    }
    uint32_t inc = s; // inclusive scan inside the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)inc, off, 64);
        if (lane >= off) inc += o;
    }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    uint32_t wbase = 0, total = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const uint32_t x = wsum[j];
        if (j < wv) wbase += x;
        total += x;
    }
    uint32_t run = wbase + inc - s;
    auto emit = [&](int i, uint32_t c) {
        ranges[i] = c ? make_uint2(run, run + c) : make_uint2(0u, 0u); // empty tiles read (0,0) like the reference's memset
        start[(size_t)i * start_stride] = run;
        run += c;
    };
    if (per <= 8) {
#pragma unroll
        for (int j = 0; j < 8; j++)
            if (j < per && b + j < e) emit(b + j, ks[j]);
    } else {
        for (int i = b; i < e; i++) emit(i, ld((size_t)i * cnt_stride));
    }
#ifdef GSR_EXP_ROWFILL
    if (tid < 8) hdr->pad[tid] = 0u;
#endif
    if (tid == 0) {
        hdr->num_rendered = total;
        hdr->overflow = total > capacity ? 1u : 0u;
        hdr->capacity = capacity;
    }
}

__global__ void __launch_bounds__(1024)
K_scan_tiles(int T, const uint32_t* __restrict__ cnt, int cnt_stride, uint32_t* __restrict__ start, int start_stride,
             uint2* __restrict__ ranges, GeomHeader* __restrict__ hdr, uint32_t capacity)
{
    scan_tiles_body<false>(T, cnt, cnt_stride, start, start_stride, ranges, hdr, capacity);
}

// Exclusive scan down the columns of the count matrix, in place; column totals -> tile_cnt. One workgroup takes
// 32 tiles x 32 row groups (a wave reads two 128-byte row segments per load), every thread keeps its <= 16 rows
// in registers between the two passes.
__global__ void __launch_bounds__(1024)
K_bin_colscan(int rows, int T, uint32_t* __restrict__ binmat, uint32_t* tile_cnt, uint32_t* __restrict__ tile_start, uint2* __restrict__ ranges,
              GeomHeader* hdr, uint32_t capacity)
{
    __shared__ uint32_t part[32][33];
    __shared__ uint32_t s_ticket;
    static_assert(GSR_BIN_ROWS <= 32 * 16, "rows per thread");
    const int c = threadIdx.x & 31, q = threadIdx.x >> 5, t = blockIdx.x * 32 + c;
    const int rper = (rows + 31) / 32, r0 = q * rper, r1 = min(rows, r0 + rper);
    uint32_t v[16];
    uint32_t sum = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        v[j] = (t < T && r0 + j < r1) ? binmat[(size_t)(r0 + j) * T + t] : 0u;
        sum += v[j];
    }
    part[q][c] = sum;
    __syncthreads();
    uint32_t run = 0;
    for (int k = 0; k < q; k++) run += part[k][c];
#pragma unroll
    for (int j = 0; j < 16; j++) {
        if (t < T && r0 + j < r1) binmat[(size_t)(r0 + j) * T + t] = run;
        run += v[j];
    }
    // The workgroup that finishes LAST also turns the tile counts into list segments (K_scan_tiles' work: one launch and its 5 us less). The hand-over
    // follows the chip's rules for data that crosses workgroups inside a launch (8 XCDs, private L2s; MI355X guide, "inter-workgroup communication"): the counts
    // are stored write-through at agent scope (sc1), every storing wave drains its stores, then ONE lane takes a ticket from a device-scope counter (zeroed by
    // K_preprocess of the same forward); the workgroup holding the last ticket acquires once and reads the counts with agent-scope loads. No release fence
    // (it would write back the XCD's L2: the 45 us of the first attempt at this fusion), and the matrix itself goes out with plain stores: the next LAUNCH reads it.
    if (q == 31 && t < T) __hip_atomic_store(tile_cnt + t, run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!tile_start) return; // (no scan wanted)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) s_ticket = __hip_atomic_fetch_add(&hdr->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_ticket != gridDim.x - 1u) return;
    if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __syncthreads();
    scan_tiles_body<true>(T, tile_cnt, 1, tile_start, 1, ranges, hdr, capacity);
}

// the forward's capacity guess was too small: switch the header to the exact capacity before the tail re-runs
__global__ void K_set_capacity(GeomHeader* hdr, uint32_t capacity)
{
    hdr->capacity = capacity;
    hdr->overflow = hdr->num_rendered > capacity ? 1u : 0u;
}

// All-ascending bitonic network ("flip" then "disperse" stages): every compare-exchange
// keeps the smaller key at the lower index, so virtual +inf padding above n never moves.
__device__ __forceinline__ void cex(uint64_t& a, uint64_t& b)
{
    if (a > b) { const uint64_t t = a; a = b; b = t; }
}

// The network is run in "trips": every thread loads a group of K = 2^G keys chosen so that G consecutive
// passes of the network pair keys inside the group, applies them in registers and stores the group back —
// 1/G of the LDS traffic and of the synchronisations of one pass per round trip. A key belongs to exactly
// one group per trip, so one synchronisation per trip is enough; with one wave per block (ONEWAVE) that
// is no s_barrier at all. cap >= K, a power of two.
//   first trip          : stages k = 2 .. K on K consecutive keys (a complete sort of the group)
//   flip trip (stage k) : flip(k) then disperse(k/4) .. disperse(k/2^G): the group is {x + m*k/2^G} and its
//                         mirror images {k-1 - (x + m*k/2^G)}, m < K/2
//   disperse trip       : up to G disperse passes of adjacent strides: the group is {x + q*2^low}, q < K
// Keys live in LDS at a bank-swizzled position: key i sits at i ^ fold(i), fold = (bits 5-9) ^ (bits 10-14)
// brought down to bits 0-4. A trip makes the lanes of a wave differ in whatever index bits are NOT in the
// group's field; the fold spreads any five of them over the 32 bank pairs. The map is linear over XOR, so
// the K addresses of a group are one swizzled base XOR per-trip constants.
__device__ __forceinline__ int swz(int i) { return i ^ ((i >> 5) & 31) ^ ((i >> 10) & 31); }

// a one-wave sort may be one of several waves of its workgroup (K_tile_sort_all): it counts its own lanes only
template <bool ONEWAVE>
__device__ __forceinline__ int sort_tid() { return ONEWAVE ? (int)(threadIdx.x & 63u) : (int)threadIdx.x; }
template <bool ONEWAVE>
__device__ __forceinline__ int sort_nt() { return ONEWAVE ? 64 : (int)blockDim.x; }
template <bool ONEWAVE>
__device__ __forceinline__ void sort_sync()
{
    if (ONEWAVE) __builtin_amdgcn_wave_barrier();
    else __syncthreads();
}
// disperse passes on local bits nb-1 .. 0 of the register group (local order ascending with the index)
template <int G>
__device__ __forceinline__ void reg_disperse(uint64_t (&r)[1 << G], int nb)
{
#pragma unroll
    for (int b = G - 1; b >= 0; b--) {
        if (b >= nb) continue;
#pragma unroll
        for (int q = 0; q < (1 << G); q++)
            if (!(q & (1 << b))) cex(r[q], r[q | (1 << b)]);
    }
}
// disperse passes with strides 2^lj ... 1 (all strides < cap)
template <int G, bool ONEWAVE>
__device__ __forceinline__ void lds_disperse_from(uint64_t* s, int cap, int lj)
{
    constexpr int K = 1 << G;
    while (lj >= 0) {
        const int low = lj - G + 1 > 0 ? lj - G + 1 : 0, nb = lj - low + 1;
        for (int grp = sort_tid<ONEWAVE>(); grp < cap / K; grp += sort_nt<ONEWAVE>()) {
            const int x = swz(((grp >> low) << (low + G)) | (grp & ((1 << low) - 1)));
            uint64_t r[K];
#pragma unroll
            for (int q = 0; q < K; q++) r[q] = s[x ^ swz(q << low)];
            reg_disperse<G>(r, nb);
#pragma unroll
            for (int q = 0; q < K; q++) s[x ^ swz(q << low)] = r[q];
        }
        sort_sync<ONEWAVE>();
        lj = low - 1;
    }
}
// full sort of `cap` keys in LDS
template <int G, bool ONEWAVE>
__device__ __forceinline__ void lds_sort(uint64_t* s, int cap)
{
    constexpr int K = 1 << G, H = K / 2;
    for (int grp = sort_tid<ONEWAVE>(); grp < cap / K; grp += sort_nt<ONEWAVE>()) { // stages k = 2 .. K inside the group
        uint64_t r[K];
#pragma unroll
        for (int q = 0; q < K; q++) r[q] = s[swz(grp * K) ^ q]; // q < 32: swz(q) = q
#pragma unroll
        for (int st = 1; st <= G; st++) {
#pragma unroll
            for (int q = 0; q < K; q++)
                if (!(q & (1 << (st - 1)))) cex(r[q], r[q ^ ((1 << st) - 1)]); // flip inside blocks of 2^st
            reg_disperse<G>(r, st - 1);
        }
#pragma unroll
        for (int q = 0; q < K; q++) s[swz(grp * K) ^ q] = r[q];
    }
    sort_sync<ONEWAVE>();
    for (int k = 2 * K, lk = G + 1; k <= cap; k <<= 1, lk++) {
        const int low = lk - G; // the group spans bits low .. lk-1
        for (int grp = sort_tid<ONEWAVE>(); grp < cap / K; grp += sort_nt<ONEWAVE>()) {
            const int xl = ((grp >> low) << lk) | (grp & ((1 << low) - 1));
            // mirror image of x in its block of k: base + k-1 - offset; its field bits are all ones, so
            // "minus m << low" is an XOR as well
            const int x = swz(xl), mirror = swz((xl | (k - 1)) - (xl & (k - 1)));
            uint64_t r[K];
#pragma unroll
            for (int m = 0; m < H; m++) { r[m] = s[x ^ swz(m << low)]; r[H + m] = s[mirror ^ swz(m << low)]; }
#pragma unroll
            for (int m = 0; m < H; m++) cex(r[m], r[H + m]); // flip(k)
#pragma unroll
            for (int b = G - 2; b >= 0; b--) // disperse(k/4) ... : in the mirrored half a set bit means a LOWER index
#pragma unroll
                for (int m = 0; m < H; m++)
                    if (!(m & (1 << b))) { cex(r[m], r[m | (1 << b)]); cex(r[H + (m | (1 << b))], r[H + m]); }
#pragma unroll
            for (int m = 0; m < H; m++) { s[x ^ swz(m << low)] = r[m]; s[mirror ^ swz(m << low)] = r[H + m]; }
        }
        sort_sync<ONEWAVE>();
        lds_disperse_from<G, ONEWAVE>(s, cap, low - 1);
    }
}

// Two instantiations share the tiles: SMALL sorts tiles of <= 1024 entries with ONE wave (16 keys per lane,
// no s_barrier), the other takes the longer lists with 256 threads; each skips the other's tiles.
//
// A list that fits LDS is bucket-sorted first: the high word of a key (depth bits, or Morton code on the k-NN path)
// maps monotonically to one of CAP bins spread over the list's own [min, max], a returning LDS atomic counts the
// bin and ranks the key inside it, a scan turns counts into bin starts, and every key finds its final place as
// bin start + the number of smaller keys (full 64-bit compare) in its bin. With depths spread over the tile's
// range a bin holds one or two keys and the whole sort is ~7 LDS operations per key, against 30 LDS operations and
// 27 compare-exchanges per key of the bitonic network. A list whose keys crowd into few bins (sum of squared bin
// counts > 8 n: the ranking loops would cost more than the network — two surfaces in one tile, a far outlier) is
// binned a second time with equalised bins (every non-empty bin cut into sub-bins in proportion to its count), and
// only if that does not spread the keys either (exact depth ties) the list is sorted by the network.
#define GSR_SORT_MATES 16 // keys of a lane that look at their bin-mates together (measured 4 / 8 / 16: 25.0 / 25.9 / 23.9 us)
#define GSR_SORT_G 4              // 16 keys per thread and trip
#define GSR_SORT_SMALL_THREADS 64
#define GSR_SORT_BIG_THREADS 256  // 1024 threads per 4096-key tile measured no faster
// KIND 0: one wave per list of <= 1024 keys, 16 keys per lane (the k-NN buckets); KIND 1: 256 threads per list of <= 4096 keys;
// KIND 2: 256 threads per list of <= 1024 keys, 4 keys per thread — the rasterizer's short lists: a quarter of KIND 0's
// registers per thread (the 16-keys-per-lane wave needs ~200 VGPRs: two waves per SIMD), four times the waves per list
#define GSR_SORT_WAVE 0
#define GSR_SORT_BLOCK 1
#define GSR_SORT_BLOCK_SHORT 2
#ifndef GSR_SORT_CUT_THREADS
#define GSR_SORT_CUT_THREADS 256 // threads of K_tile_sort_cut's workgroup (KIND 2): 128 = 8 keys per thread
#endif
template <int KIND>
struct SortShared {
    static constexpr int NT = KIND == 0 ? GSR_SORT_SMALL_THREADS : KIND == 2 ? GSR_SORT_CUT_THREADS : GSR_SORT_BIG_THREADS;
    static constexpr int CAP = KIND == 1 ? GSR_SORT_CAP : GSR_SORT_SMALL;
    uint64_t s[CAP];
    __attribute__((aligned(16))) uint32_t h[CAP + 64]; // 64 spare words: one per lane for the padding keys' (zero) atomics
    uint32_t red[9][NT / 64];
};
// reach (rasterizer only): the dense per-splat reach array; every key's entry is gathered through its id as soon as the
// key is loaded (the gathers are in flight while the list is binned and ranked) and turned into the mask of the tile's
// patches, which is carried to the key's sorted position.
// Returns where the sorted ids of the list can be read back besides point_list: GSR_IDS_H (sh.h[i]; the payload words,
// if any, sit in the key array viewed as uint32_t: sort_payload(sh)[i]), GSR_IDS_S (the low words of sh.s[swz(i)]: the
// network ran, no payload) or GSR_IDS_GLOBAL (an oversize list).
#define GSR_IDS_H 0
#define GSR_IDS_S 1
#define GSR_IDS_GLOBAL 2
template <int KIND>
__device__ __forceinline__ uint32_t* sort_payload(SortShared<KIND>& sh) { return reinterpret_cast<uint32_t*>(sh.s); }
// Oversize list: the bitonic network with chunk-local stages in LDS and long-stride stages in global memory, in place in seg
// (the k-NN buckets over 4096 points; in the rasterizer only lists the depth bins cannot split: > 1024 exact depth ties).
// Not inlined: a rare path must not set the register allocation of the kernels that can reach it.
#define GSR_OVERSIZE_G 3 // 8 keys per thread and trip (16 on the LDS paths): this path only has to be correct, and its registers count for every kernel that can reach it
template <int KIND>
__device__ __forceinline__ void sort_oversize(SortShared<KIND>& sh, uint64_t* __restrict__ seg, const int n, uint32_t* __restrict__ out)
{
    constexpr int NT = SortShared<KIND>::NT, CAP = SortShared<KIND>::CAP, LOGCAP = CAP == 1024 ? 10 : CAP == 2048 ? 11 : 12;
    uint64_t* const s = sh.s;
    int tid_ = (int)threadIdx.x;
    asm volatile("" : "+v"(tid_)); // (see sort_tile)
    const int tid = tid_;
    long n2 = CAP;
    while (n2 < n) n2 <<= 1;
    const int nchunks = (int)(n2 / CAP);
    for (int c = 0; c < nchunks; c++) {
        const long base = (long)c * CAP;
        if (base >= n) break;
        for (int i = tid; i < CAP; i += NT) s[swz(i)] = base + i < n ? seg[base + i] : ~0ull;
        __syncthreads();
        lds_sort<GSR_OVERSIZE_G, false>(s, CAP);
        for (int i = tid; i < CAP; i += NT) if (base + i < n) seg[base + i] = s[swz(i)];
        __syncthreads();
    }
    for (long k = 2L * CAP; k <= n2; k <<= 1) {
        for (long i = tid; i < n2 / 2; i += NT) { // flip in global memory
            const long blk = i / (k >> 1), off = i % (k >> 1);
            const long lo = blk * k + off, hi = blk * k + (k - 1 - off);
            if (hi < n) { uint64_t a = seg[lo], b = seg[hi]; if (a > b) { seg[lo] = b; seg[hi] = a; } }
        }
        __syncthreads();
        long j = k >> 2;
        for (; j >= CAP; j >>= 1) { // disperse with stride >= chunk: global memory
            for (long i = tid; i < n2 / 2; i += NT) {
                const long lo = (i / j) * 2 * j + (i % j), hi = lo + j;
                if (hi < n) { uint64_t a = seg[lo], b = seg[hi]; if (a > b) { seg[lo] = b; seg[hi] = a; } }
            }
            __syncthreads();
        }
        for (int c = 0; c < nchunks; c++) { // remaining strides are chunk-local
            const long base = (long)c * CAP;
            if (base >= n) break;
            for (int i = tid; i < CAP; i += NT) s[swz(i)] = base + i < n ? seg[base + i] : ~0ull;
            __syncthreads();
            lds_disperse_from<GSR_OVERSIZE_G, false>(s, CAP, LOGCAP - 1); // strides CAP/2 ... 1
            for (int i = tid; i < CAP; i += NT) if (base + i < n) seg[base + i] = s[swz(i)];
            __syncthreads();
        }
    }
    for (int i = tid; i < n; i += NT) out[i] = (uint32_t)seg[i];
}

// seg[0 .. n): the keys (sorted in place only on the oversize path); out[0 .. n): the sorted ids
template <int KIND>
__device__ __forceinline__ int sort_tile(SortShared<KIND>& sh, uint64_t* __restrict__ seg, const int n, uint32_t* __restrict__ out,
                                         const uint2* __restrict__ reach = nullptr, int tx = 0, int ty = 0)
{
    constexpr int NT = SortShared<KIND>::NT, CAP = SortShared<KIND>::CAP;
    constexpr int EPT = CAP / NT, LOGCAP = CAP == 1024 ? 10 : CAP == 2048 ? 11 : 12, MATES = EPT < GSR_SORT_MATES ? EPT : GSR_SORT_MATES;
    static_assert(EPT % 4 == 0 && (1 << LOGCAP) == CAP, "whole uint4 of bins per thread");
    uint64_t* const s = sh.s;
    uint32_t* const h = sh.h;
    auto& red = sh.red;
    // (opaque to the optimiser: the rasterizer calls this in a loop over the chunks of a list, and loop-invariant code motion
    // would hoist every address derived from the thread id out of that loop and keep it live across the whole body: +20 VGPRs)
    int tid_ = sort_tid<KIND == 0>();
    asm volatile("" : "+v"(tid_));
    const int tid = tid_, lane = tid & 63, wv = tid >> 6;
    if (n <= CAP) {
        // Straight-line code: a load inside a divergent branch is waited for inside that branch, sixteen branches would be
        // sixteen serial round trips. Loads use clamped addresses and selects, only stores are predicated.
        uint64_t k[EPT];
        uint2 pay[EPT];
        uint32_t dmin = ~0u, dmax = 0u;
#pragma unroll
        for (int j = 0; j < EPT; j++) {
            const int i = j * NT + tid;
            const uint64_t v = seg[min(i, n - 1)];
            pay[j] = reach ? reach[(uint32_t)v] : make_uint2(0u, 0u);
            k[j] = i < n ? v : ~0ull; // padding keys: above every real key
            dmin = min(dmin, (uint32_t)(k[j] >> 32));
            dmax = max(dmax, i < n ? (uint32_t)(v >> 32) : 0u);
            h[i] = 0u;
        }
        dmin = wave_min_u32(dmin);
        dmax = wave_max_dpp(dmax);
        if (KIND != 0) {
            if (lane == 0) { red[0][wv] = dmin; red[1][wv] = dmax; }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < NT / 64; q++) { dmin = min(dmin, red[0][q]); dmax = max(dmax, red[1][q]); }
        } else {
            sort_sync<true>();
        } // (the barrier of the reduction also orders the zeroing of h before the atomics below)
        const int shift = max(0, 32 - __clz((int)(dmax - dmin)) - LOGCAP); // (dmax - dmin) >> shift < CAP
        uint32_t bin[EPT], rnk[EPT]; // rank of the key inside its bin, in arrival order
#pragma unroll
        for (int j = 0; j < EPT; j++) {
            const bool valid = j * NT + tid < n;
            // padding keys: a spare word per lane (four spare words for all of them were 700 returning atomics per address on a
            // 1 300-key list in 4096 slots)
            bin[j] = valid ? ((uint32_t)(k[j] >> 32) - dmin) >> shift : (uint32_t)(CAP + lane);
            rnk[j] = __hip_atomic_fetch_add(&h[bin[j]], valid ? 1u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        sort_sync<KIND == 0>();
        // counts -> bin starts: every thread owns sixteen consecutive bins. SLOT: which words of `red` a call may use
        // (every call its own: no barrier needed between the calls)
        auto block_sums = [&](const uint32_t v, const int slot, uint32_t& before, uint32_t& total) {
            const uint32_t inc = wave_scan_add(v);
            before = inc - v;
            total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
            if (KIND != 0) {
                if (lane == 63) red[slot][wv] = inc;
                __syncthreads();
                total = 0;
#pragma unroll
                for (int q = 0; q < NT / 64; q++) { if (q < wv) before += red[slot][q]; total += red[slot][q]; }
            }
        };
        uint32_t c[EPT], run, sq, nz;
        auto scan_counts = [&](const uint32_t* arr, const int slot) {
#pragma unroll
            for (int q = 0; q < EPT / 4; q++) {
                const uint4 v = reinterpret_cast<const uint4*>(arr)[tid * (EPT / 4) + q];
                c[4 * q] = v.x; c[4 * q + 1] = v.y; c[4 * q + 2] = v.z; c[4 * q + 3] = v.w;
            }
            uint32_t sum = 0, sq_own = 0, nz_own = 0;
#pragma unroll
            for (int j = 0; j < EPT; j++) { sum += c[j]; sq_own += c[j] * c[j]; nz_own += c[j] ? 1u : 0u; }
            // three block reductions behind one barrier (a barrier-separated phase costs this workgroup ~1 us: it is latency-bound)
            const uint32_t i0 = wave_scan_add(sum), i1 = wave_scan_add(sq_own), i2 = wave_scan_add(nz_own);
            run = i0 - sum;
            sq = (uint32_t)__builtin_amdgcn_readlane((int)i1, 63);
            nz = (uint32_t)__builtin_amdgcn_readlane((int)i2, 63);
            if (KIND != 0) {
                if (lane == 63) { red[slot][wv] = i0; red[slot + 1][wv] = i1; red[slot + 2][wv] = i2; }
                __syncthreads();
                sq = 0; nz = 0;
#pragma unroll
                for (int q = 0; q < NT / 64; q++) { if (q < wv) run += red[slot][q]; sq += red[slot + 1][q]; nz += red[slot + 2][q]; }
            }
        };
        scan_counts(h, 2);
        bool crowded = sq > 8u * (uint32_t)n;
        if (crowded && shift > 0) {
            // The depths crowd into few of the equal-width bins (two surfaces in one tile, a far outlier). Second attempt with
            // equalised bins: every non-empty bin is cut into a number of sub-bins proportional to its count, CAP in all,
            // and the keys are binned again by their position inside the old bin. Still monotone in the depth bits.
            uint32_t* const h2 = reinterpret_cast<uint32_t*>(s); // the keys are in registers: s is free until they are placed
            const float share = (float)((uint32_t)CAP - nz) / (float)n * 0.999f;
            uint32_t nsub[EPT], tot = 0, first, t;
#pragma unroll
            for (int j = 0; j < EPT; j++) { nsub[j] = c[j] ? 1u + (uint32_t)((float)c[j] * share) : 0u; tot += nsub[j]; }
            block_sums(tot, 5, first, t);
#pragma unroll
            for (int q = 0; q < EPT / 4; q++) { // first sub-bin | sub-bins << 16
                uint4 v;
                v.x = first | (nsub[4 * q] << 16); first += nsub[4 * q]; v.y = first | (nsub[4 * q + 1] << 16); first += nsub[4 * q + 1];
                v.z = first | (nsub[4 * q + 2] << 16); first += nsub[4 * q + 2]; v.w = first | (nsub[4 * q + 3] << 16); first += nsub[4 * q + 3];
                reinterpret_cast<uint4*>(h)[tid * (EPT / 4) + q] = v;
            }
#pragma unroll
            for (int j = 0; j < EPT; j++) h2[j * NT + tid] = 0u;
            sort_sync<KIND == 0>();
            const int down = max(0, shift - 16), frac = min(shift, 16); // position inside the old bin, in 16 bits
#pragma unroll
            for (int j = 0; j < EPT; j++) {
                const bool valid = j * NT + tid < n;
                const uint32_t m = h[min(bin[j], (uint32_t)CAP - 1u)];
                const uint32_t inside = (((uint32_t)(k[j] >> 32) - dmin) - (bin[j] << shift)) >> down;
                bin[j] = valid ? (m & 0xFFFFu) + ((inside * (m >> 16)) >> frac) : (uint32_t)(CAP + lane);
                rnk[j] = __hip_atomic_fetch_add(&h2[bin[j]], valid ? 1u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            sort_sync<KIND == 0>();
            scan_counts(h2, 6);
            crowded = sq > 8u * (uint32_t)n;
        }
        if (!crowded) {
#pragma unroll
            for (int q = 0; q < EPT / 4; q++) { // bin start | count << 16: one random read per key instead of two
                uint4 v;
                v.x = run | (c[4 * q] << 16); run += c[4 * q]; v.y = run | (c[4 * q + 1] << 16); run += c[4 * q + 1];
                v.z = run | (c[4 * q + 2] << 16); run += c[4 * q + 2]; v.w = run | (c[4 * q + 3] << 16); run += c[4 * q + 3];
                reinterpret_cast<uint4*>(h)[tid * (EPT / 4) + q] = v;
            }
            sort_sync<KIND == 0>();
            uint32_t sb[EPT]; // bin start | keys of the bin below this one << 16
#pragma unroll
            for (int j = 0; j < EPT; j++) {
                const bool valid = j * NT + tid < n;
                const uint32_t sc = h[bin[j]];
                sb[j] = sc & 0xFFFFu;
                if (valid) s[sb[j] + rnk[j]] = k[j];
                rnk[j] = valid ? rnk[j] | (sc & 0xFFFF0000u) : 0u; // rank | bin count << 16 (padding: the spare words hold anything)
            }
            sort_sync<KIND == 0>();
            // rank inside the bin = the number of smaller keys among the OTHER keys of the bin (most bins hold one key: nothing
            // to do). The keys of a lane step through their bin-mates together: the reads of one step are issued back to
            // back and waited for once; a key without a further mate reads word 0 (a broadcast, no bank conflict).
#pragma unroll
            for (int j0 = 0; j0 < EPT; j0 += MATES) {
                for (uint32_t q = 1;; q++) {
                    bool more = false;
#pragma unroll
                    for (int j = j0; j < j0 + MATES; j++) more |= q < (rnk[j] >> 16);
                    if (!__builtin_amdgcn_ballot_w64(more)) break;
                    uint64_t mate[MATES];
#pragma unroll
                    for (int j = j0; j < j0 + MATES; j++) {
                        const uint32_t cnt = rnk[j] >> 16;
                        uint32_t o = (rnk[j] & 0xFFFFu) + q;
                        o = o >= cnt ? o - cnt : o;
                        mate[j - j0] = s[q < cnt ? (sb[j] & 0xFFFFu) + o : 0u];
                    }
#pragma unroll
                    for (int j = j0; j < j0 + MATES; j++) sb[j] += (q < (rnk[j] >> 16) && mate[j - j0] < k[j]) ? 0x10000u : 0u;
                }
            }
            // ids to their places in LDS, then out in order (the counts in h are dead: every lane read them before the barrier above);
            // the payload words go to the same places of the key array, which every thread has finished reading
            if (reach) sort_sync<KIND == 0>();
            uint32_t* const hp = sort_payload(sh);
#pragma unroll
            for (int j = 0; j < EPT; j++)
                if (j * NT + tid < n) {
                    const uint32_t pos = (sb[j] & 0xFFFFu) + (sb[j] >> 16);
                    h[pos] = (uint32_t)k[j];
                    if (reach) hp[pos] = tile_mask_from_reach(pay[j], tx, ty);
                }
            sort_sync<KIND == 0>();
#pragma unroll
            for (int j = 0; j < EPT; j++) {
                const uint32_t id = h[j * NT + tid];
                if (j * NT + tid < n) out[j * NT + tid] = id;
            }
            return GSR_IDS_H;
        }
        sort_sync<KIND == 0>();
    }
    if (n <= CAP) { // crowded bins: the bitonic network
        int n2 = 1 << GSR_SORT_G;
        while (n2 < n) n2 <<= 1;
        for (int i = tid; i < n2; i += sort_nt<KIND == 0>()) s[swz(i)] = i < n ? seg[i] : ~0ull;
        sort_sync<KIND == 0>();
        lds_sort<GSR_SORT_G, KIND == 0>(s, n2);
        for (int i = tid; i < n; i += sort_nt<KIND == 0>()) out[i] = (uint32_t)s[swz(i)];
        return GSR_IDS_S;
    }
    if constexpr (KIND == GSR_SORT_WAVE) return GSR_IDS_GLOBAL; // (the one-wave sort is never handed such lists)
    else {
        sort_oversize<KIND>(sh, seg, n, out);
        return GSR_IDS_GLOBAL;
    }
}

// One workgroup per list (the k-NN path's buckets; the rasterizer's tiles go through K_tile_sort_cut).
template <int KIND>
__global__ void __launch_bounds__(KIND == 0 ? GSR_SORT_SMALL_THREADS : GSR_SORT_BIG_THREADS)
K_tile_sort(int ntiles, const uint2* __restrict__ ranges, const GeomHeader* __restrict__ hdr,
            uint64_t* __restrict__ pairs, uint32_t* __restrict__ point_list)
{
    __shared__ SortShared<KIND> sh;
    if (hdr->overflow) return;
    const uint2 r = ranges[xcd_remap(blockIdx.x, ntiles, 0u)];
    const int n = (int)(r.y - r.x);
    if (n == 0 || (n <= GSR_SORT_SMALL) != (KIND == 0)) return;
    (void)sort_tile<KIND>(sh, pairs + r.x, n, point_list + r.x);
}

// Lists of more than 1024 entries (rasterizer): the list is first cut into depth-ordered CHUNKS of at most 1024 keys, and
// every chunk is then sorted, masked and cut into the quad lists by the same code as a short list, one chunk after the
// other in the same 256-thread workgroup (~70 VGPRs, 13 KB of LDS: seven workgroups per CU whatever the frame holds; the
// round-3 build ran such lists in a second kernel with 16 keys per thread — 220 VGPRs, 48 KB of LDS, two workgroups per CU,
// 30-60 us of barrier-separated phases per list — and launched that kernel, empty, on frames without them).
// Partition = one counting sort over GSR_PART_BINS equal-width bins of the depth word: min / max, count, [equalise: every
// non-empty bin cut into sub-bins in proportion to its count, count again], scan, scatter into the tile's own share of the
// quad-hit log (dead until the first chunk is cut: a chunk's quad-list records never reach beyond the keys already
// consumed). A chunk = the bins whose first key falls into the same window of S list positions, S = 1025 - the largest
// bin count: at most 1024 keys, found with one LDS atomic-min per bin. Lists that cannot be split that way (more than 960
// keys of one depth, or more than GSR_PART_SLOTS windows) are sorted whole by the bitonic network in global memory.
#ifdef GSR_EXP_SORT_PHASES // instrumented build (scripts/sort_phases.py): wall-clock stamps of the phases of every tile's workgroup
__device__ unsigned long long g_sort_phases[16 * 16384];
#define GSR_PHASE(k) do { if (threadIdx.x == 0 && tile < 16384) g_sort_phases[16 * tile + (k)] = wall_clock64(); } while (0)
#define GSR_PHASE_ARG , const int tile
#define GSR_PHASE_PASS , tile
#else
#define GSR_PHASE(k) do { } while (0)
#define GSR_PHASE_ARG
#define GSR_PHASE_PASS
#endif
#define GSR_PART_BINS 2048
#define GSR_PART_SLOTS 256
#define GSR_PART_NONE 0xFFFFFFFFu
// Returns the window size S (> 0: chunk_first[w] = list position of window w's first key, GSR_PART_NONE for an empty window,
// keys by bin in temp[0 .. n)) or 0 (no partition: sort seg whole).
template <int KIND>
__device__ __forceinline__ int partition_list(SortShared<KIND>& sh, uint32_t* chunk_first, const uint64_t* __restrict__ seg, const int n,
                                              uint64_t* __restrict__ temp, uint32_t* __restrict__ map GSR_PHASE_ARG)
{
    constexpr int NT = SortShared<KIND>::NT, NB = GSR_PART_BINS, BPT = NB / NT, LOGNB = 11, U = 16;
    static_assert((1 << LOGNB) == NB && BPT % 4 == 0, "whole uint4 of bins per thread");
    static_assert(sizeof(sh.s) >= NB * sizeof(uint32_t), "the bins live in the sort's key array");
    uint32_t* const hb = reinterpret_cast<uint32_t*>(sh.s);
    auto& red = sh.red;
    int tid_ = (int)threadIdx.x;
    asm volatile("" : "+v"(tid_)); // (see sort_tile: nothing derived from the thread id is shared with the chunk code)
    const int tid = tid_, lane = tid & 63, wv = tid >> 6;
    int slot = 0; // every block reduction of a list its own words of `red`
    auto block_sums = [&](const uint32_t v, uint32_t& before, uint32_t& total) {
        const uint32_t inc = wave_scan_add(v);
        if (lane == 63) red[slot][wv] = inc;
        __syncthreads();
        before = inc - v; total = 0;
#pragma unroll
        for (int q = 0; q < NT / 64; q++) { if (q < wv) before += red[slot][q]; total += red[slot][q]; }
        slot++;
    };
    // ---- min / max of the depth words. Lists of up to NT * U = 4096 keys (all but the 10 M-splat experiment's) are loaded ONCE
    //      and stay in registers through the three passes; longer lists are re-read trip by trip
    const bool resident = n <= NT * U;
    uint64_t k[U];
    auto load_trip = [&](const int i0) {
#pragma unroll
        for (int u = 0; u < U; u++) k[u] = seg[min(i0 + u * NT + tid, n - 1)];
    };
    uint32_t dmin = ~0u, dmax = 0u;
    for (int i0 = 0; i0 < n; i0 += NT * U) {
        load_trip(i0);
#pragma unroll
        for (int u = 0; u < U; u++) { dmin = min(dmin, (uint32_t)(k[u] >> 32)); dmax = max(dmax, (uint32_t)(k[u] >> 32)); }
    }
    dmin = wave_min_u32(dmin); dmax = wave_max_dpp(dmax);
    if (lane == 0) { red[7][wv] = dmin; red[8][wv] = dmax; }
    for (int b = tid; b < NB; b += NT) hb[b] = 0u;
    for (int w = tid; w < GSR_PART_SLOTS; w += NT) chunk_first[w] = GSR_PART_NONE;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NT / 64; q++) { dmin = min(dmin, red[7][q]); dmax = max(dmax, red[8][q]); }
    const int shift = max(0, 32 - __clz((int)(dmax - dmin)) - LOGNB);
    const int down = max(0, shift - 16), frac = min(shift, 16);
    bool equalised = false;
    GSR_PHASE(7);
    auto bin_of = [&](const uint64_t key) -> uint32_t {
        const uint32_t rel = (uint32_t)(key >> 32) - dmin, b = rel >> shift;
        if (!equalised) return b;
        const uint32_t m = map[b];
        return (m & 0xFFFFu) + ((((rel - (b << shift)) >> down) * (m >> 16)) >> frac);
    };
    auto count_keys = [&]() {
        for (int i0 = 0; i0 < n; i0 += NT * U) {
            if (!resident) load_trip(i0);
#pragma unroll
            for (int u = 0; u < U; u++)
                if (i0 + u * NT + tid < n) (void)lds_take(&hb[bin_of(k[u])]);
        }
        __syncthreads();
    };
    uint32_t c[BPT], run, nz, mx;
    auto scan_counts = [&]() {
#pragma unroll
        for (int q = 0; q < BPT / 4; q++) {
            const uint4 v = reinterpret_cast<const uint4*>(hb)[tid * (BPT / 4) + q];
            c[4 * q] = v.x; c[4 * q + 1] = v.y; c[4 * q + 2] = v.z; c[4 * q + 3] = v.w;
        }
        uint32_t sum = 0, nz_own = 0, mx_own = 0;
#pragma unroll
        for (int j = 0; j < BPT; j++) { sum += c[j]; nz_own += c[j] ? 1u : 0u; mx_own = max(mx_own, c[j]); }
        const uint32_t i0 = wave_scan_add(sum), i1 = wave_scan_add(nz_own);
        mx_own = wave_max_dpp(mx_own);
        if (lane == 63) { red[slot][wv] = i0; red[slot + 1][wv] = i1; red[slot + 2][wv] = mx_own; }
        __syncthreads();
        run = i0 - sum; nz = 0; mx = 0;
#pragma unroll
        for (int q = 0; q < NT / 64; q++) { if (q < wv) run += red[slot][q]; nz += red[slot + 1][q]; mx = max(mx, red[slot + 2][q]); }
        slot += 3;
    };
    count_keys();
    GSR_PHASE(8);
    scan_counts();
    GSR_PHASE(9);
    if (mx > 256u && shift > 0) { // crowded bins (two surfaces in one tile, a far outlier): equalised bins, as in sort_tile
        const float share = (float)((uint32_t)NB - nz) / (float)n * 0.999f;
        uint32_t nsub[BPT], tot = 0, first, x;
#pragma unroll
        for (int j = 0; j < BPT; j++) { nsub[j] = c[j] ? 1u + (uint32_t)((float)c[j] * share) : 0u; tot += nsub[j]; }
        block_sums(tot, first, x);
#pragma unroll
        for (int j = 0; j < BPT; j++) { map[tid * BPT + j] = first | (nsub[j] << 16); first += nsub[j]; }
        __syncthreads(); // every thread has read its counts, and the map is visible to the workgroup
        for (int b = tid; b < NB; b += NT) hb[b] = 0u;
        __syncthreads();
        equalised = true;
        count_keys();
        scan_counts();
    }
    const int S = 1025 - (int)mx;
    if (S < 64 || (n + S - 1) / S > GSR_PART_SLOTS) return 0;
    __syncthreads(); // every thread has read its counts: the words turn into cursors
#pragma unroll
    for (int j = 0; j < BPT; j++) {
        hb[tid * BPT + j] = run;
        if (c[j]) (void)__hip_atomic_fetch_min(&chunk_first[run / (uint32_t)S], run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        run += c[j];
    }
    __syncthreads();
    GSR_PHASE(10);
    // ---- keys to their bins
    for (int i0 = 0; i0 < n; i0 += NT * U) {
        if (!resident) load_trip(i0);
#pragma unroll
        for (int u = 0; u < U; u++)
            if (i0 + u * NT + tid < n) temp[lds_take(&hb[bin_of(k[u])])] = k[u];
    }
    __syncthreads(); // (workgroup-scope release / acquire: the scattered keys are read back by this workgroup only)
    return S;
}

// Quad-hit lists. The blend kernels work per 8x8 quad (one wave, four 4x4 patch rows): what they walk is not the tile
// list but, per quad, the entries of the tile list that can reach the quad, each with the 4-bit mask of the patches it
// reaches: (list position, id | mask << 28), in list order. The lists are cut here, by the waves that have just sorted the
// tile. (The forward blend used to cull while it walked: every tile entry's 48-byte record gathered in each of the tile's
// four quad-waves, 9.7 M gathers for 3.4 M quad hits at 1 M splats.) Culling is exact (gsr_device.h): the sort gathers every
// key's 8-byte reach entry while it sorts (one gather per tile instance, overlapped with the binning and ranking),
// shifts a small splat's reach word into the 16-bit mask of the tile's patches and carries it to the entry's sorted
// position; only the few larger splats of a list (GSR_MASK_UNTESTED) take the closed-form quad and patch tests, all of
// them in one batch per list.
__device__ __forceinline__ uint32_t exact_tile_mask(const float4 a, const float4 b, const int tx, const int ty)
{
    // (only splats whose alpha >= 1/255 region extends beyond ~15 pixels come here: a quad such a splat reaches counts as reached in
    // all four of its patches — the exact patch test cost more here than the few border patches it removed from the blend lists)
    uint32_t m16 = 0u;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const float X0 = (float)(tx * 16 + (q & 1) * 8), Y0 = (float)(ty * 16 + (q >> 1) * 8);
        if (quad_reach(a, b, X0, Y0)) m16 |= 0x33u << (8 * (q >> 1) + 2 * (q & 1));
    }
    return m16;
}
// Where a chunk of a tile list goes: the list has n_list entries (the stride of the four quad lists in qh), the chunk starts at
// list position pos0, and qcnt[q] = records quad q's list holds so far (LDS, carried from chunk to chunk).
struct CutTarget {
    uint2* qh;
    uint32_t* qcnt;
    int n_list, pos0;
};
// ONE wave appends to the list of quad q the entries [0, m) of a chunk given by get(i) -> (id, tile mask).
template <typename Get>
__device__ __forceinline__ void cut_quad_list(Get get, const int m, const int q, const CutTarget& t)
{
    int lane_ = (int)(threadIdx.x & 63u);
    asm volatile("" : "+v"(lane_)); // (see sort_tile)
    const int lane = lane_;
    uint32_t cnt = t.qcnt[q];
    uint2* const dst = t.qh + (size_t)q * (size_t)t.n_list;
    uint2 nxt = get(min(lane, m - 1));
    for (int base = 0; base < m; base += 64) {
        const int k = base + lane;
        const uint2 e = nxt;
        nxt = get(min(k + 64, m - 1)); // the next 64 entries are in flight while these are cut (a global round trip per step otherwise)
        const uint32_t pm = k < m ? quad_mask_of_tile_mask(e.y, q) : 0u;
        const unsigned long long mk = __ballot(pm != 0u);
        if (pm != 0u) dst[cnt + (uint32_t)mbcnt64(mk)] = make_uint2((uint32_t)(t.pos0 + k), e.x | (pm << GSR_ID_BITS));
        cnt += (uint32_t)__popcll(mk);
    }
    if (lane == 0) t.qcnt[q] = cnt;
}
// A chunk whose sorted ids and mask words sit in LDS (ids[i], msk[i]): the untested entries are collected (their positions in
// `todo`, u16), tested in one batch — one gather of the two 16-byte words per such entry, UB entries per thread in flight —
// and then the four quad lists are appended to, one quad per wave.
template <int UB>
__device__ __forceinline__ void emit_from_lds(const uint32_t* ids, uint32_t* msk, uint16_t* todo, uint32_t* counter, const int m, const int tx, const int ty,
                                              const GeomView& g, const CutTarget& t)
{
    int tid_ = (int)threadIdx.x;
    asm volatile("" : "+v"(tid_)); // (see sort_tile)
    const int tid = tid_, nt = (int)blockDim.x;
    // (*counter was zeroed by the caller before the chunk was sorted: barriers in between)
    for (int i0 = 0; i0 < m; i0 += nt) { // one LDS atomic per wave and step, not per entry: a map of fat splats has every entry here
        const int i = i0 + tid;
        const bool u = i < m && (msk[min(i, m - 1)] & GSR_MASK_UNTESTED) != 0u;
        const unsigned long long mk = __ballot(u);
        if (mk) {
            uint32_t base = 0u;
            if ((tid & 63) == 0) base = __hip_atomic_fetch_add(counter, (uint32_t)__popcll(mk), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            if (u) todo[base + (uint32_t)mbcnt64(mk)] = (uint16_t)i;
        }
    }
    __syncthreads();
    const int nu = __builtin_amdgcn_readfirstlane((int)*counter);
    for (int u0 = 0; u0 < nu; u0 += nt * UB) {
        int i[UB];
        float4 a[UB], b[UB];
#pragma unroll
        for (int v = 0; v < UB; v++) {
            i[v] = todo[min(u0 + v * nt + tid, nu - 1)];
            const uint32_t id = ids[i[v]];
            a[v] = g.g0[id]; b[v] = g.g1[id];
        }
#pragma unroll
        for (int v = 0; v < UB; v++) {
            const uint32_t mk = exact_tile_mask(a[v], b[v], tx, ty);
            if (u0 + v * nt + tid < nu) msk[i[v]] = mk;
        }
    }
    if (nu > 0) __syncthreads();
    auto get = [&](int i) { return make_uint2(ids[i], msk[i]); };
    for (int q = tid >> 6; q < 4; q += nt >> 6) cut_quad_list(get, m, q, t); // (four waves: one quad each)
}
// A chunk without its mask words at hand (the bitonic network ran: exact depth ties): the sorted ids are read back from `out`
// (just written by this workgroup), the masks recomputed from the gathered reach entries, (id, mask) parked in the chunk's
// original key segment `park` (dead once it is sorted) — every thread of the workgroup, U independent chains each — and then
// one wave per quad cuts its list out of the parked pairs: coalesced, independent loads, one step ahead.
template <int U>
__device__ __forceinline__ void emit_from_global(const uint32_t* out, uint64_t* park, const int m, const int tx, const int ty, const GeomView& g, const CutTarget& t)
{
    __syncthreads(); // (workgroup-scope release / acquire: the ids were stored by this workgroup)
    int tid_ = (int)threadIdx.x;
    asm volatile("" : "+v"(tid_)); // (see sort_tile)
    const int tid = tid_, nt = (int)blockDim.x;
    for (int i0 = 0; i0 < m; i0 += nt * U) {
        uint32_t id[U];
        uint2 re[U];
#pragma unroll
        for (int u = 0; u < U; u++) id[u] = out[min(i0 + u * nt + tid, m - 1)];
#pragma unroll
        for (int u = 0; u < U; u++) re[u] = g.reach[id[u]];
#pragma unroll
        for (int u = 0; u < U; u++) {
            uint32_t mk = tile_mask_from_reach(re[u], tx, ty);
            if (mk & GSR_MASK_UNTESTED) mk = exact_tile_mask(g.g0[id[u]], g.g1[id[u]], tx, ty);
            if (i0 + u * nt + tid < m) park[i0 + u * nt + tid] = (uint64_t)id[u] | ((uint64_t)mk << 32);
        }
    }
    __syncthreads();
    auto get = [&](int i) { const uint64_t v = park[i]; return make_uint2((uint32_t)v, (uint32_t)(v >> 32)); };
    for (int q = tid >> 6; q < 4; q += nt >> 6) cut_quad_list(get, m, q, t);
}

// The rasterizer's tile sort: one 256-thread workgroup per tile, whatever the length of its list. Lists of up to 1024
// entries (every list of the 1 M-splat headline frame) are bucket-sorted in LDS in one go; longer lists are partitioned into
// depth-ordered chunks of at most 1024 keys first (partition_list) and the chunks take the same path one after the other.
// Every chunk is cut into the tile's four quad-hit lists right after it is sorted, one quad per wave.
// (History: one wave per list with 16 keys per lane — 63 us with the list cutting in; round 3: this kernel for the short lists
// plus K_tile_sort_long — 48 KB of LDS, 220 VGPRs — for the others: 37.5 + 4.8 us on the headline frame, 130 / 275 us on the
// 2 M-splat and fat-splat frames.)
#ifndef GSR_SORT_WAVES
#define GSR_SORT_WAVES 6 // waves per SIMD the register allocation is held to (80 VGPRs, no spills; 7 = 72 VGPRs spills six)
#endif
__global__ void __launch_bounds__(GSR_SORT_CUT_THREADS) __attribute__((amdgpu_waves_per_eu(GSR_SORT_WAVES, GSR_SORT_WAVES)))
K_tile_sort_cut(int T, int grid_x, const uint2* __restrict__ ranges, GeomView g, uint64_t* __restrict__ pairs,
                uint32_t* __restrict__ point_list, uint2* __restrict__ qhits, uint32_t* __restrict__ qcount)
{
    __shared__ SortShared<GSR_SORT_BLOCK_SHORT> sh;
    __shared__ uint32_t counter, qcnt[4];
    __shared__ uint32_t chunk_first[GSR_PART_SLOTS];
    if (g.hdr->overflow) return;
    const int tile = (int)xcd_remap(blockIdx.x, (uint32_t)T, GSR_XCD_SORT_TILES);
    const uint2 r = ranges[tile];
    const int n = (int)(r.y - r.x);
    uint32_t* const qc4 = qcount + 4 * (size_t)tile;
    if (n == 0) { // an empty tile has four empty lists
        if (threadIdx.x < 4u) qc4[threadIdx.x] = 0u;
        return;
    }
    const int tx = tile % grid_x, ty = tile / grid_x;
    if (threadIdx.x < 4u) qcnt[threadIdx.x] = 0u;
    GSR_PHASE(0);
    CutTarget ct;
    ct.qh = qhits + 4 * (size_t)r.x; ct.qcnt = qcnt; ct.n_list = n; ct.pos0 = 0;
#ifdef GSR_EXP_SORT_NOGATHER
    const uint2* const reach = nullptr;
#else
    const uint2* const reach = g.reach;
#endif
    // one chunk: keys src[0 .. m) = list positions pos0 .. pos0 + m
    auto chunk = [&](uint64_t* src, const int m, const int pos0) {
        ct.pos0 = pos0;
        if (threadIdx.x == 0) counter = 0u; // (emit_from_lds's to-do counter: the sort's barriers lie between this and its use)
        const int where = sort_tile<GSR_SORT_BLOCK_SHORT>(sh, src, m, point_list + r.x + pos0, reach, tx, ty);
        GSR_PHASE(4); // (last chunk's)
#ifdef GSR_EXP_SORT_NOEMIT
        return;
#endif
        if (where == GSR_IDS_H) { // ids in h, mask words in the first 4 KB of the key array, the to-do list behind them
            // (no barrier here: the threads still copying ids out of h only read, and the to-do list lies in words of the key
            // array nobody has touched since the barrier before the ranking)
            uint32_t* const msk = sort_payload(sh);
            emit_from_lds<2>(sh.h, msk, reinterpret_cast<uint16_t*>(msk + GSR_SORT_SMALL), &counter, m, tx, ty, g, ct);
        } else { // the network ran (exact depth ties), in LDS or in global memory
            // (parked in the list's original key segment, dead by now: the chunk's own keys may sit in the quad-hit log, where
            // quad 0's records are being appended while the other waves still read the parked pairs)
            emit_from_global<2>(point_list + r.x + pos0, pairs + r.x + pos0, m, tx, ty, g, ct);
        }
    };
    // (one call site of the chunk code: a short list, or a list that cannot be split — sorted whole by the network in global
    // memory, in place —, is a single chunk in its original key segment)
    uint64_t* const temp = reinterpret_cast<uint64_t*>(qhits + 4 * (size_t)r.x);     // [n] keys by bin
    uint32_t* const map = reinterpret_cast<uint32_t*>(temp + n);                      // [GSR_PART_BINS] equalisation map (8 KB <= 24 n bytes)
    int S = 0, nw = 1, w = 0;
    if (n > GSR_SORT_SMALL) {
        // (values read from LDS are vector registers to the compiler: the loop state is made scalar explicitly, or it and
        // everything derived from it — chunk pointers, lengths — stays live in VGPRs across the whole chunk body)
        S = __builtin_amdgcn_readfirstlane(partition_list<GSR_SORT_BLOCK_SHORT>(sh, chunk_first, pairs + r.x, n, temp, map GSR_PHASE_PASS));
        GSR_PHASE(1);
        if (S > 0) {
            nw = (n + S - 1) / S;
            while (w < nw && __builtin_amdgcn_readfirstlane((int)chunk_first[w]) == (int)GSR_PART_NONE) w++;
        }
    }
    while (w < nw) {
        int start = 0, end = n, w2 = nw;
        if (S > 0) {
            start = __builtin_amdgcn_readfirstlane((int)chunk_first[w]);
            w2 = w + 1;
            while (w2 < nw && __builtin_amdgcn_readfirstlane((int)chunk_first[w2]) == (int)GSR_PART_NONE) w2++;
            end = w2 < nw ? __builtin_amdgcn_readfirstlane((int)chunk_first[w2]) : n;
        }
        if (w2 >= nw) GSR_PHASE(3); // start of the last chunk
        chunk(S > 0 ? temp + start : pairs + r.x, end - start, start);
        w = w2;
        if (w < nw) __syncthreads(); // the chunk's LDS is free again
    }
    if ((threadIdx.x & 63u) == 0u) // (the thread that wrote the count: no barrier)
        for (uint32_t q = threadIdx.x >> 6; q < 4u; q += GSR_SORT_CUT_THREADS / 64) qc4[q] = qcnt[q];
    GSR_PHASE(5);
#ifdef GSR_EXP_SORT_PHASES
    if (threadIdx.x == 0 && tile < 16384) g_sort_phases[16 * tile + 6] = (unsigned long long)n;
#endif
}

// ===================================================================================
// per-splat backward (reference K11 + K12 fused; 3D covariance recomputed, not stored)
// ===================================================================================
struct SplatGrads {
    float* dL_dmean2D;
    float* dL_dconic;
    float* dL_dopacity;
    float* dL_dcolor;
    float* dL_dmean3D;
    float* dL_dcov3D;
    float* dL_dsh;
    float* dL_dscale;
    float* dL_drot;
};

__device__ __forceinline__ void st3(float* p, size_t i, float a, float b, float c)
{
    if (p) { p[3 * i] = a; p[3 * i + 1] = b; p[3 * i + 2] = c; }
}

// FUSED (gsr_backward_args.fused_map_update): the gradients are not written — the Gaussian's raw parameters take their Adam step right
// here (map_update_with, csrc/gsr_train.h: the activations' backward, the camera transform's, the regularisers' gradient), from registers:
// 56 bytes written and 56 read per Gaussian and one launch less per mapping iteration than gsr_backward + gsr_map_update.
// (the body of one thread; gmean: the splat's dL/dmean3D — zero for a splat that is culled or out of range — for K_splat_bwd_pose)
template <bool REZERO, bool FUSED>
__device__ __forceinline__ void splat_bwd_thread(const FrameParams& f, const SplatInputs& in, const GeomView& g, const SplatGrads& o, const MapUpdate& mu, float3& gmean)
{
    gmean = make_float3(0.f, 0.f, 0.f);
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= f.P) return;
    if (FUSED && mu.overflow && *mu.overflow) return; // (the forward rendered nothing: no step)
    const size_t i = (size_t)idx;
    // Two trips to memory per wave instead of six (round 4; see K_preprocess): the matrices are requested first — uniform
    // addresses, read before the kernel's first store: scalar loads (behind a store the compiler has to fetch them per lane) —
    // and everything a visible splat needs is requested in one go once its radius is known.
    float vm[16], pm[16];
#pragma unroll
    for (int k = 0; k < 16; k++) { vm[k] = in.view[k]; pm[k] = in.proj[k]; }
#pragma unroll
    for (int k = 0; k < 16; k++) { asm volatile("" : "+s"(vm[k])); asm volatile("" : "+s"(pm[k])); }
    Pose34 Tp; // (fused update: the pose the means were taken to the camera frame with)
    if (FUSED) {
        Tp = load_pose(mu.Tcw);
#pragma unroll
        for (int k = 0; k < 9; k++) Tp.r[k] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, Tp.r[k]))); // (uniform: scalar registers)
    }
    float4 gb = g.g1[idx];
    const int radius = __float_as_int(gb.w);
    if (FUSED && radius <= 0) { // invisible: zero gradients, but Adam still steps on its moments
        const float z3[3] = {0.f, 0.f, 0.f};
        map_update_with(i, mu, Tp, z3, z3, make_float4(0.f, 0.f, 0.f, 0.f), 0.f, z3);
        return;
    }
    if (radius <= 0) { // invisible: every gradient is zero (the reference leaves its zero-fill)
        st3(o.dL_dmean2D, i, 0.f, 0.f, 0.f);
        if (o.dL_dconic) reinterpret_cast<float4*>(o.dL_dconic)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (o.dL_dopacity) o.dL_dopacity[i] = 0.f;
        st3(o.dL_dcolor, i, 0.f, 0.f, 0.f);
        st3(o.dL_dmean3D, i, 0.f, 0.f, 0.f);
        if (o.dL_dcov3D) for (int k = 0; k < 6; k++) o.dL_dcov3D[6 * i + k] = 0.f;
        if (o.dL_dsh) for (int k = 0; k < f.M * 3; k++) o.dL_dsh[i * f.M * 3 + k] = 0.f;
        st3(o.dL_dscale, i, 0.f, 0.f, 0.f);
        if (o.dL_drot) reinterpret_cast<float4*>(o.dL_drot)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    float* const acc = g.acc + i * GSR_ACC_STRIDE;
    float4 q0 = reinterpret_cast<const float4*>(acc)[0], q1 = reinterpret_cast<const float4*>(acc)[1];
    float q8 = acc[8], q9 = acc[9]; // q9: dL/d(view depth as a colour) of the fused depth channel (zero without it)
    float4 ga = g.g0[idx];
    float3 mean = make_float3(in.means3D[3 * i], in.means3D[3 * i + 1], in.means3D[3 * i + 2]);
    float cov3D[6];
    float3 sc = make_float3(0.f, 0.f, 0.f);
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
    if (in.cov3D_precomp) {
#pragma unroll
        for (int k = 0; k < 6; k++) cov3D[k] = in.cov3D_precomp[6 * i + k];
    }
    if (in.scales) {
        sc = make_float3(in.scales[3 * i], in.scales[3 * i + 1], in.scales[3 * i + 2]);
        q = reinterpret_cast<const float4*>(in.rotations)[i];
    }
    pin(q0.x); pin(q0.y); pin(q0.z); pin(q0.w); pin(q1.x); pin(q1.y); pin(q1.z); pin(q1.w); pin(q8); pin(q9);
    pin(ga.x); pin(ga.y); pin(ga.z); pin(ga.w); pin(mean.x); pin(mean.y); pin(mean.z);
    if (in.cov3D_precomp) {
#pragma unroll
        for (int k = 0; k < 6; k++) pin(cov3D[k]);
    }
    if (in.scales) { pin(sc.x); pin(sc.y); pin(sc.z); pin(q.x); pin(q.y); pin(q.z); pin(q.w); }
    MapRegs mr; // (fused update: the Gaussian's raw parameters and moments travel with the same request: 168 of its 408 bytes)
#ifndef GSR_EXP_NOMAPHOIST
    if (FUSED) { map_load(i, mu, mr); map_pin(mr); }
#endif
    if (!in.cov3D_precomp) cov3d_from_scale_rot(sc, f.scale_modifier, q, cov3D);
    if (REZERO) { // consumed: leave the record clean for the next backward on this geometry blob
        float4* const ap = reinterpret_cast<float4*>(acc);
        ap[0] = ap[1] = ap[2] = ap[3] = make_float4(0.f, 0.f, 0.f, 0.f); // the whole 64-byte line
    }
    // K_blend_bwd accumulated the raw moments of u = G*dL/dalpha: {u, u dx, u dy, u dx^2, u dx dy, u dy^2};
    // the reference's per-pixel terms (backward.cu:536-554) are these moments times conic / opacity:
    const float ca = ga.z, cb = ga.w, cc = gb.x, op = gb.y;
    const float dmx = (op * -(ca * q0.y + cb * q0.z)) * (float)(0.5 * f.W);
    const float dmy = (op * -(cc * q0.z + cb * q0.y)) * (float)(0.5 * f.H);
    const float hop = -0.5f * op;
    const float dconx = hop * q0.w, dcony = hop * q1.x, dconw = hop * q1.y;
    const float dopac = q0.x;
    float3 dcol = make_float3(q1.z, q1.w, q8);
    st3(o.dL_dmean2D, i, dmx, dmy, 0.f);
    if (o.dL_dconic) reinterpret_cast<float4*>(o.dL_dconic)[i] = make_float4(dconx, dcony, 0.f, dconw);
    if (o.dL_dopacity) o.dL_dopacity[i] = dopac;
    st3(o.dL_dcolor, i, dcol.x, dcol.y, dcol.z);

    // ---- conic -> 2D covariance -> 3D covariance and mean (backward.cu:144-274) ----
    const Cov2D k = cov2d_forward(mean, f.focal_x, f.focal_y, f.tan_fovx, f.tan_fovy, cov3D, vm);
    const float x_grad_mul = (k.txtz < -k.limx || k.txtz > k.limx) ? 0.f : 1.f;
    const float y_grad_mul = (k.tytz < -k.limy || k.tytz > k.limy) ? 0.f : 1.f;
    const M3& T = k.T; const M3& Vrk = k.Vrk; const M3& Wm = k.W;
    const float a = k.a, b = k.b, c = k.c;
    const float denom = a * c - b * b;
    float dL_da = 0.f, dL_db = 0.f, dL_dc = 0.f;
    const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
    float dcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (denom2inv != 0.f) {
        dL_da = denom2inv * (-c * c * dconx + 2 * b * c * dcony + (denom - a * c) * dconw);
        dL_dc = denom2inv * (-a * a * dconw + 2 * a * b * dcony + (denom - a * c) * dconx);
        dL_db = denom2inv * 2 * (b * c * dconx - (denom + 2 * b * b) * dcony + a * b * dconw);
        dcov[0] = (T.m[0][0] * T.m[0][0] * dL_da + T.m[0][0] * T.m[1][0] * dL_db + T.m[1][0] * T.m[1][0] * dL_dc);
        dcov[3] = (T.m[0][1] * T.m[0][1] * dL_da + T.m[0][1] * T.m[1][1] * dL_db + T.m[1][1] * T.m[1][1] * dL_dc);
        dcov[5] = (T.m[0][2] * T.m[0][2] * dL_da + T.m[0][2] * T.m[1][2] * dL_db + T.m[1][2] * T.m[1][2] * dL_dc);
        dcov[1] = 2 * T.m[0][0] * T.m[0][1] * dL_da + (T.m[0][0] * T.m[1][1] + T.m[0][1] * T.m[1][0]) * dL_db + 2 * T.m[1][0] * T.m[1][1] * dL_dc;
        dcov[2] = 2 * T.m[0][0] * T.m[0][2] * dL_da + (T.m[0][0] * T.m[1][2] + T.m[0][2] * T.m[1][0]) * dL_db + 2 * T.m[1][0] * T.m[1][2] * dL_dc;
        dcov[4] = 2 * T.m[0][2] * T.m[0][1] * dL_da + (T.m[0][1] * T.m[1][2] + T.m[0][2] * T.m[1][1]) * dL_db + 2 * T.m[1][1] * T.m[1][2] * dL_dc;
    }
    if (o.dL_dcov3D) for (int n = 0; n < 6; n++) o.dL_dcov3D[6 * i + n] = dcov[n];

    // rows of (T * Vrk) for the two used columns of T
    float tv0[3], tv1[3];
#pragma unroll
    for (int n = 0; n < 3; n++) {
        tv0[n] = T.m[0][0] * Vrk.m[n][0] + T.m[0][1] * Vrk.m[n][1] + T.m[0][2] * Vrk.m[n][2];
        tv1[n] = T.m[1][0] * Vrk.m[n][0] + T.m[1][1] * Vrk.m[n][1] + T.m[1][2] * Vrk.m[n][2];
    }
    const float dL_dT00 = 2 * tv0[0] * dL_da + tv1[0] * dL_db;
    const float dL_dT01 = 2 * tv0[1] * dL_da + tv1[1] * dL_db;
    const float dL_dT02 = 2 * tv0[2] * dL_da + tv1[2] * dL_db;
    const float dL_dT10 = 2 * tv1[0] * dL_dc + tv0[0] * dL_db;
    const float dL_dT11 = 2 * tv1[1] * dL_dc + tv0[1] * dL_db;
    const float dL_dT12 = 2 * tv1[2] * dL_dc + tv0[2] * dL_db;
    const float dL_dJ00 = Wm.m[0][0] * dL_dT00 + Wm.m[0][1] * dL_dT01 + Wm.m[0][2] * dL_dT02;
    const float dL_dJ02 = Wm.m[2][0] * dL_dT00 + Wm.m[2][1] * dL_dT01 + Wm.m[2][2] * dL_dT02;
    const float dL_dJ11 = Wm.m[1][0] * dL_dT10 + Wm.m[1][1] * dL_dT11 + Wm.m[1][2] * dL_dT12;
    const float dL_dJ12 = Wm.m[2][0] * dL_dT10 + Wm.m[2][1] * dL_dT11 + Wm.m[2][2] * dL_dT12;
    const float tz = 1.f / k.t.z, tz2 = tz * tz, tz3 = tz2 * tz;
    const float hx = f.focal_x, hy = f.focal_y;
    const float dL_dtx = x_grad_mul * -hx * tz2 * dL_dJ02;
    const float dL_dty = y_grad_mul * -hy * tz2 * dL_dJ12;
    const float dL_dtz = -hx * tz2 * dL_dJ00 - hy * tz2 * dL_dJ11 + (2 * hx * k.t.x) * tz3 * dL_dJ02 + (2 * hy * k.t.y) * tz3 * dL_dJ12;
    float3 dmean = make_float3(vm[0] * dL_dtx + vm[1] * dL_dty + vm[2] * dL_dtz,
                               vm[4] * dL_dtx + vm[5] * dL_dty + vm[6] * dL_dtz,
                               vm[8] * dL_dtx + vm[9] * dL_dty + vm[10] * dL_dtz);
    // the fused depth channel blends the view depth z = vm[2] x + vm[6] y + vm[10] z + vm[14] as a colour
    // (the reference feeds it as colors_precomp[:, 0] of its second render and lets autograd take it back to the mean)
    if (f.fold_depth_color) { dmean.x += vm[2] * q9; dmean.y += vm[6] * q9; dmean.z += vm[10] * q9; }

    // ---- screen-space mean -> 3D mean (backward.cu:366-387) ----
    {
        const float* pj = pm;
        const float4 m_hom = xform4x4(mean, pj);
        const float m_w = 1.0f / (m_hom.w + 0.0000001f);
        const float mul1 = (pj[0] * mean.x + pj[4] * mean.y + pj[8] * mean.z + pj[12]) * m_w * m_w;
        const float mul2 = (pj[1] * mean.x + pj[5] * mean.y + pj[9] * mean.z + pj[13]) * m_w * m_w;
        dmean.x += (pj[0] * m_w - pj[3] * mul1) * dmx + (pj[1] * m_w - pj[3] * mul2) * dmy;
        dmean.y += (pj[4] * m_w - pj[7] * mul1) * dmx + (pj[5] * m_w - pj[7] * mul2) * dmy;
        dmean.z += (pj[8] * m_w - pj[11] * mul1) * dmx + (pj[9] * m_w - pj[11] * mul2) * dmy;
    }

    // ---- colour -> SH and view direction -> mean (backward.cu:20-139) ----
    if (in.shs && o.dL_dsh) {
        const uint32_t flags = __float_as_uint(g.col[idx].w);
        float3 raw;
        const float3 d = unit_dir(mean, in.campos, raw);
        const float gr = (flags & 1u) ? 0.f : dcol.x, gg = (flags & 2u) ? 0.f : dcol.y, gb = (flags & 4u) ? 0.f : dcol.z;
        const float* sh = in.shs + i * f.M * 3;
        float* dsh = o.dL_dsh + i * f.M * 3;
        for (int n = (f.D + 1) * (f.D + 1); n < f.M; n++) { dsh[3 * n] = 0.f; dsh[3 * n + 1] = 0.f; dsh[3 * n + 2] = 0.f; }
        const float3 d0 = sh_channel_backward(f.D, sh, dsh, 0, d, gr);
        const float3 d1 = sh_channel_backward(f.D, sh, dsh, 1, d, gg);
        const float3 d2 = sh_channel_backward(f.D, sh, dsh, 2, d, gb);
        const float3 dL_ddir = make_float3(d0.x * gr + d1.x * gg + d2.x * gb, d0.y * gr + d1.y * gg + d2.y * gb,
                                           d0.z * gr + d1.z * gg + d2.z * gb);
        const float3 dm = dnormvdv(raw, dL_ddir);
        dmean.x += dm.x; dmean.y += dm.y; dmean.z += dm.z;
    }
    st3(o.dL_dmean3D, i, dmean.x, dmean.y, dmean.z);
    gmean = dmean;

    // ---- 3D covariance -> scale, rotation (backward.cu:278-341) ----
    float ds3[3] = {0.f, 0.f, 0.f};
    float4 dq = make_float4(0.f, 0.f, 0.f, 0.f);
    if (in.scales && (FUSED || (o.dL_dscale && o.dL_drot))) {
        const float r = q.x, x = q.y, y = q.z, z = q.w;
        const M3 R = quat_R(q);
        const float3 s = make_float3(f.scale_modifier * sc.x, f.scale_modifier * sc.y, f.scale_modifier * sc.z);
        M3 M2 = scaled_R(s, R);
#pragma unroll
        for (int cc2 = 0; cc2 < 3; cc2++)
#pragma unroll
            for (int rr = 0; rr < 3; rr++) M2.m[cc2][rr] = 2.0f * M2.m[cc2][rr];
        M3 dS;
        dS.m[0][0] = dcov[0]; dS.m[0][1] = 0.5f * dcov[1]; dS.m[0][2] = 0.5f * dcov[2];
        dS.m[1][0] = 0.5f * dcov[1]; dS.m[1][1] = dcov[3]; dS.m[1][2] = 0.5f * dcov[4];
        dS.m[2][0] = 0.5f * dcov[2]; dS.m[2][1] = 0.5f * dcov[4]; dS.m[2][2] = dcov[5];
        const M3 dL_dM = m3_mul(M2, dS);
        const M3 Rt = m3_t(R);
        M3 dMt = m3_t(dL_dM);
        const float dsx = Rt.m[0][0] * dMt.m[0][0] + Rt.m[0][1] * dMt.m[0][1] + Rt.m[0][2] * dMt.m[0][2];
        const float dsy = Rt.m[1][0] * dMt.m[1][0] + Rt.m[1][1] * dMt.m[1][1] + Rt.m[1][2] * dMt.m[1][2];
        const float dsz = Rt.m[2][0] * dMt.m[2][0] + Rt.m[2][1] * dMt.m[2][1] + Rt.m[2][2] * dMt.m[2][2];
        ds3[0] = dsx; ds3[1] = dsy; ds3[2] = dsz;
#pragma unroll
        for (int n = 0; n < 3; n++) { dMt.m[0][n] *= s.x; dMt.m[1][n] *= s.y; dMt.m[2][n] *= s.z; }
#define GSR_D(cc3, rr3) dMt.m[cc3][rr3]
        dq.x = 2 * z * (GSR_D(0, 1) - GSR_D(1, 0)) + 2 * y * (GSR_D(2, 0) - GSR_D(0, 2)) + 2 * x * (GSR_D(1, 2) - GSR_D(2, 1));
        dq.y = 2 * y * (GSR_D(1, 0) + GSR_D(0, 1)) + 2 * z * (GSR_D(2, 0) + GSR_D(0, 2)) + 2 * r * (GSR_D(1, 2) - GSR_D(2, 1)) - 4 * x * (GSR_D(2, 2) + GSR_D(1, 1));
        dq.z = 2 * x * (GSR_D(1, 0) + GSR_D(0, 1)) + 2 * r * (GSR_D(2, 0) - GSR_D(0, 2)) + 2 * z * (GSR_D(1, 2) + GSR_D(2, 1)) - 4 * y * (GSR_D(2, 2) + GSR_D(0, 0));
        dq.w = 2 * r * (GSR_D(0, 1) - GSR_D(1, 0)) + 2 * x * (GSR_D(2, 0) + GSR_D(0, 2)) + 2 * y * (GSR_D(1, 2) + GSR_D(2, 1)) - 4 * z * (GSR_D(1, 1) + GSR_D(0, 0));
#undef GSR_D
    }
    if (FUSED) {
        const float gx[3] = {dmean.x, dmean.y, dmean.z}, gc[3] = {dcol.x, dcol.y, dcol.z};
#ifndef GSR_EXP_NOMAPHOIST
        map_apply(i, mu, Tp, mr, gx, gc, dq, dopac, ds3);
#else
        map_update_with(i, mu, Tp, gx, gc, dq, dopac, ds3);
#endif
        return;
    }
    st3(o.dL_dscale, i, ds3[0], ds3[1], ds3[2]);
    if (o.dL_drot) reinterpret_cast<float4*>(o.dL_drot)[i] = dq;
}

template <bool REZERO, bool FUSED = false>
__global__ void __launch_bounds__(256)
K_splat_bwd(FrameParams f, SplatInputs in, GeomView g, SplatGrads o, MapUpdate mu)
{
    float3 gm;
    splat_bwd_thread<REZERO, FUSED>(f, in, g, o, mu, gm);
}

// gsr_backward_args.fused_pose_step (a tracking iteration: the pose is the only parameter): the per-splat stage also forms the twelve pose sums of
// dL/dmeans_cam against the WORLD-frame means (K_pose_grad's: dL/dR[i][j] = sum dmc[i] X[j], dL/dt = sum dmc) — the gradient is in registers, the
// sums cost 12 bytes of loads per splat where gsr_pose_grad re-read 24 and this kernel wrote 12. Every workgroup ADDS its twelve sums to one of
// GSR_POSE_ACC_ROWS accumulator rows (float atomics, fire and forget: the backward's sums are atomics' anyway), K_pose_finish adds the rows up, takes the
// pose step (gsr_pose_update) and leaves the rows zero. (Measured first: the last workgroup taking the step — write-through rows, a ticket per
// workgroup: 49 us against 28 + 17 for the two launches; 3 907 short-lived workgroups each wait for their stores and their ticket.)
#define GSR_POSE_ACC_ROWS 64
struct PoseStep {
    const float* X;      // [P,3] world-frame means
    float* acc;          // [GSR_POSE_ACC_ROWS][12], zero between launches
    float* overflow_out; // nullptr, or where this forward's overflow flag goes as a float (the sharded loop all-reduces it with the rows)
};
template <bool REZERO>
__global__ void __launch_bounds__(256)
K_splat_bwd_pose(FrameParams f, SplatInputs in, GeomView g, SplatGrads o, PoseStep ps)
{
    __shared__ float ws[4][12];
    const int idx = blockIdx.x * 256 + threadIdx.x;
    float x[3] = {0.f, 0.f, 0.f};
    if (idx < f.P) { x[0] = ps.X[3 * (size_t)idx]; x[1] = ps.X[3 * (size_t)idx + 1]; x[2] = ps.X[3 * (size_t)idx + 2]; } // (requested before the body's loads)
    float3 gm;
    splat_bwd_thread<REZERO, false>(f, in, g, o, MapUpdate{}, gm);
    float a[12] = {gm.x * x[0], gm.x * x[1], gm.x * x[2], gm.y * x[0], gm.y * x[1], gm.y * x[2], gm.z * x[0], gm.z * x[1], gm.z * x[2], gm.x, gm.y, gm.z};
#pragma unroll
    for (int q = 0; q < 12; q++) a[q] = wave_sum_lane63(a[q]);
    if ((threadIdx.x & 63) == 63) {
#pragma unroll
        for (int q = 0; q < 12; q++) ws[threadIdx.x >> 6][q] = a[q];
    }
    __syncthreads();
    if (threadIdx.x < 12) {
        const float row = (ws[0][threadIdx.x] + ws[1][threadIdx.x]) + (ws[2][threadIdx.x] + ws[3][threadIdx.x]);
        if (row != 0.f) unsafeAtomicAdd(ps.acc + (size_t)(blockIdx.x % GSR_POSE_ACC_ROWS) * 12 + threadIdx.x, row);
    }
    // (a store at the top would turn the body's uniform matrix loads into per-lane loads: DESIGN.md section 4, "Dependent trips")
    if (ps.overflow_out && blockIdx.x == 0 && threadIdx.x == 64) *ps.overflow_out = g.hdr->overflow ? 1.f : 0.f;
}
// adds the accumulator rows up, takes the pose step, leaves the rows zero for the next backward
__global__ void __launch_bounds__(64)
K_pose_finish(PoseUpdate u, float* acc, float* sums_out)
{
    static_assert(GSR_POSE_ACC_ROWS == 64, "one accumulator row per lane");
    float r[12];
#pragma unroll
    for (int q = 0; q < 12; q++) r[q] = acc[threadIdx.x * 12 + q];
#pragma unroll
    for (int q = 0; q < 12; q++) { acc[threadIdx.x * 12 + q] = 0.f; r[q] = wave_sum_lane63(r[q]); }
    float tot[12];
#pragma unroll
    for (int q = 0; q < 12; q++) tot[q] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r[q]), 63));
    if (sums_out && threadIdx.x < 12) sums_out[threadIdx.x] = tot[threadIdx.x];
    pose_update_body<false>(u, 0, tot);
    if (u.skip && threadIdx.x == 0) *u.skip = 0.f; // (read above by the same lane; zero between iterations like the rows)
}

// ===================================================================================
// inspection kernels (tests only): opaque blobs -> the reference's array layout
// ===================================================================================
__global__ void __launch_bounds__(256)
K_export_splats(int P, int grid_x, int grid_y, GeomView g, float* means2D, float* depths,
                float* conic_opacity, float* rgb, uint32_t* tiles_touched)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= P) return;
    const float4 b = g.g1[idx];
    const int radius = __float_as_int(b.w);
    const bool vis = radius > 0;
    const float4 a = vis ? g.g0[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 c = vis ? g.col[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
    if (means2D) { means2D[2 * idx] = a.x; means2D[2 * idx + 1] = a.y; }
    if (depths) depths[idx] = vis ? b.z : 0.f;
    if (conic_opacity) reinterpret_cast<float4*>(conic_opacity)[idx] = vis ? make_float4(a.z, a.w, b.x, b.y) : make_float4(0.f, 0.f, 0.f, 0.f);
    if (rgb) { rgb[3 * idx] = c.x; rgb[3 * idx + 1] = c.y; rgb[3 * idx + 2] = c.z; }
    if (tiles_touched) {
        uint32_t t = 0;
        if (vis) { int x0, y0, x1, y1; tile_rect(a.x, a.y, radius, grid_x, grid_y, x0, y0, x1, y1); t = (uint32_t)((x1 - x0) * (y1 - y0)); }
        tiles_touched[idx] = t;
    }
}

__global__ void __launch_bounds__(256)
K_export_keys(int ntiles, const uint2* ranges, const uint32_t* point_list, GeomView g, uint64_t* keys)
{
    const int tile = blockIdx.x;
    if (tile >= ntiles) return;
    const uint2 r = ranges[tile];
    for (uint32_t k = r.x + threadIdx.x; k < r.y; k += 256)
        keys[k] = ((uint64_t)tile << 32) | __float_as_uint(g.g1[point_list[k]].z);
}

} // namespace gsr
