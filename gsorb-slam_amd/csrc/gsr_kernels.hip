// gsr_kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels of the differentiable
// Gaussian-splat rasterizer. Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off.
//
// Forward  (replaces rasterizer_impl.cu:199-345 of the reference's DGR tree):
//   K_preprocess   per splat : project, cull, radius, tile rectangle, colour; ONE count atomic per splat on the
//                              (class, anchor) counter of its tile rectangle (gsr_device.h: TileRec, Cls4Rec)
//   K_tile_runs    per tile  : the (class, anchor) runs that cover a tile -> its count and run offsets
//   K_scan_tiles   1 block   : counts -> segment starts, ranges, num_rendered, overflow flag
//   K_anchor_table per anchor: absolute run starts per (anchor, class, covered tile)
//   K_fill         per splat : (depth bits<<32 | id) at run start + rank in each of its tiles (no atomics)
//   K_tile_sort    per tile  : bitonic sort of the tile's segment in LDS -> point_list
//                              ((depth, id) ascending == the reference's stable radix order)
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

__global__ void __launch_bounds__(256)
K_preprocess(FrameParams f, SplatInputs in, int* __restrict__ radii_out, GeomView g,
             TileRec* __restrict__ tiles, Cls4Rec* __restrict__ cls4, uint32_t* __restrict__ tier2)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= f.P) return;
    const float3 p = make_float3(in.means3D[3 * (size_t)idx], in.means3D[3 * (size_t)idx + 1], in.means3D[3 * (size_t)idx + 2]);
    float cov[6];
    load_cov3d(in, f, idx, cov);
    Projected pr;
    const bool vis = project_splat(p, cov, f, in.view, in.proj, pr);
    if (!vis) {
        g.g1[idx] = make_float4(0.f, 0.f, 0.f, 0.f); // radius 0 marks the splat invisible
        g.slots[idx] = make_uint4(0u, 0u, 0u, 0u);
        if (radii_out) radii_out[idx] = 0;
        return;
    }
    float4 c;
    if (in.colors_precomp) {
        c = make_float4(in.colors_precomp[3 * (size_t)idx], in.colors_precomp[3 * (size_t)idx + 1],
                        in.colors_precomp[3 * (size_t)idx + 2], 0.f);
    } else {
        float3 raw;
        const float3 d = unit_dir(p, in.campos, raw);
        const float* sh = in.shs + (size_t)idx * f.M * 3;
        const float r = sh_channel(f.D, sh, 0, d), gg = sh_channel(f.D, sh, 1, d), b = sh_channel(f.D, sh, 2, d);
        const uint32_t flags = (r < 0 ? 1u : 0u) | (gg < 0 ? 2u : 0u) | (b < 0 ? 4u : 0u);
        c = make_float4(fmaxf(r, 0.f), fmaxf(gg, 0.f), fmaxf(b, 0.f), __uint_as_float(flags));
    }
    const float opac = in.opacities[idx];
    // the exact patch reach of a small splat, once per splat instead of once per (tile entry, quad) in the blend (gsr_device.h)
    c.w = __uint_as_float(__float_as_uint(c.w) | splat_reach25(pr.px, pr.py, pr.conic_a, pr.conic_b, pr.conic_c, opac));
    g.g0[idx] = make_float4(pr.px, pr.py, pr.conic_a, pr.conic_b);
    g.g1[idx] = make_float4(pr.conic_c, opac, pr.p_view.z, __int_as_float(pr.radius));
    g.col[idx] = c;
    if (radii_out) radii_out[idx] = pr.radius;
    // count the splat into its tiles: ONE returning atomic for a rectangle of at most 2x2 tiles (see TileRec)
    pr.y0 = max(pr.y0, f.band_y0); pr.y1 = max(pr.y0, min(pr.y1, f.band_y1)); // only the band's tile rows are binned
    const int w = pr.x1 - pr.x0, h = pr.y1 - pr.y0;
    const uint32_t zbits = __float_as_uint(pr.p_view.z);
    const uint32_t r0 = (uint32_t)pr.x0 | ((uint32_t)pr.y0 << 16), r1 = (uint32_t)pr.x1 | ((uint32_t)pr.y1 << 16);
    if (w * h == 0) { g.slots[idx] = make_uint4(0u, 0u, 0u, 0u); return; }
    if (w <= 2 && h <= 2) {
        const uint32_t rank = atomicAdd(&tiles[pr.y0 * f.grid_x + pr.x0].cls[(w - 1) + 2 * (h - 1)], 1u);
        g.slots[idx] = make_uint4(rank, zbits, r0, r1);
    } else if (w <= 4 && h <= 4) { // second tier: still one atomic
        const uint32_t rank = atomicAdd(&cls4[pr.y0 * f.grid_x + pr.x0].c[(w - 1) + 4 * (h - 1)], 1u);
        g.slots[idx] = make_uint4(rank, zbits, r0, r1);
        tier2[0] = 1u;
    } else {
        g.slots[idx] = make_uint4(0u, zbits, r0, r1);
        for (int y = pr.y0; y < pr.y1; y++)
            for (int x = pr.x0; x < pr.x1; x++) atomicAdd(&tiles[y * f.grid_x + x].cnt_big, 1u);
    }
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
// Per tile, 16 lanes per tile: cnt_small and the run offsets of its list segment.
//   first tier (TileRec): the nine (class, anchor) runs that cover the tile, off[j] = where run j+1 starts
//   relative to the segment start (lane 0 of the group);
//   second tier (Cls4Rec), only when some splat used it this frame: one lane per anchor (dx, dy) up-left of
//   the tile sums the classes of its anchor that reach the tile, a 16-lane shuffle scan orders the anchors,
//   and each lane writes the run offsets of its anchor behind the first-tier runs.
// The one-block scan below then only reads two counters per tile.
__global__ void __launch_bounds__(256)
K_tile_runs(int T, int grid_x, TileRec* __restrict__ tiles, const Cls4Rec* __restrict__ cls4,
            const uint32_t* __restrict__ tier2, uint32_t* __restrict__ run4)
{
    const int i = blockIdx.x * 16 + (threadIdx.x >> 4), nb = threadIdx.x & 15, dx = nb & 3, dy = nb >> 2;
    if (i >= T) return; // whole 16-lane groups leave together
    const int ty = i / grid_x, tx = i - ty * grid_x;
    uint32_t o1 = 0;
    if (nb == 0) {
        const bool L = tx > 0, U = ty > 0;
        const uint4 me = *reinterpret_cast<const uint4*>(tiles[i].cls);
        const uint4 le = L ? *reinterpret_cast<const uint4*>(tiles[i - 1].cls) : make_uint4(0u, 0u, 0u, 0u);
        const uint4 up = U ? *reinterpret_cast<const uint4*>(tiles[i - grid_x].cls) : make_uint4(0u, 0u, 0u, 0u);
        const uint32_t ul = (L && U) ? tiles[i - grid_x - 1].cls[3] : 0u;
        const uint32_t c[9] = {me.x, me.y, le.y, me.z, up.z, me.w, le.w, up.w, ul};
        uint32_t o = c[0];
        uint32_t off[8];
#pragma unroll
        for (int j = 0; j < 8; j++) { off[j] = o; o += c[j + 1]; }
        uint4* const dst = reinterpret_cast<uint4*>(tiles[i].off);
        dst[0] = make_uint4(off[0], off[1], off[2], off[3]);
        dst[1] = make_uint4(off[4], off[5], off[6], off[7]);
        o1 = o;
    }
    o1 = (uint32_t)__shfl((int)o1, 0, 16);
    uint32_t total = 0;
    if (tier2[0]) {
        const bool have = tx >= dx && ty >= dy;
        uint32_t cnt[16];
        uint32_t sum = 0;
        const uint4* const src = reinterpret_cast<const uint4*>(cls4[have ? i - dy * grid_x - dx : i].c);
        const uint4 q0 = src[0], q1 = src[1], q2 = src[2], q3 = src[3];
        const uint32_t raw[16] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
#pragma unroll
        for (int c = 0; c < 16; c++) {
            const int w = (c & 3) + 1, h = (c >> 2) + 1;
            cnt[c] = (have && w > dx && h > dy && !(w <= 2 && h <= 2)) ? raw[c] : 0u;
            sum += cnt[c];
        }
        uint32_t inc = sum; // inclusive scan over the 16 anchors of the tile
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) {
            const uint32_t o = (uint32_t)__shfl_up((int)inc, off, 16);
            if (nb >= off) inc += o;
        }
        total = (uint32_t)__shfl((int)inc, 15, 16);
        uint32_t o = o1 + inc - sum; // behind the first-tier runs
        uint32_t* const row = run4 + (size_t)i * GSR_RUN4 + nb * 16;
#pragma unroll
        for (int c = 0; c < 16; c++) { row[c] = o; o += cnt[c]; }
    }
    if (nb == 0) tiles[i].cnt_small = o1 + total;
}

// One block: per-tile counts (cnt_small, cnt_big) -> list segments (start, cursor of the big splats),
// ranges, num_rendered and the overflow flag. Shared with the k-NN path (buckets instead of tiles).
__global__ void __launch_bounds__(1024)
K_scan_tiles(int T, TileRec* __restrict__ tiles, uint2* __restrict__ ranges,
             GeomHeader* __restrict__ hdr, uint32_t capacity)
{
    auto counts = [&](int i, uint32_t& cs, uint32_t& cb) { cs = tiles[i].cnt_small; cb = tiles[i].cnt_big; };
    // each thread owns `per` consecutive tiles (its counts stay in registers when per <= 8), the block
    // scan is one shuffle scan per wave plus one over the 16 wave totals: two barriers in all
    __shared__ uint32_t wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int per = (T + 1023) / 1024;
    const int b = tid * per, e = min(T, b + per);
    uint32_t ks[8], kb[8];
    uint32_t s = 0;
    if (per <= 8) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            ks[j] = kb[j] = 0u;
            if (j < per && b + j < e) counts(b + j, ks[j], kb[j]);
            s += ks[j] + kb[j];
        }
    } else {
        for (int i = b; i < e; i++) { uint32_t cs, cb; counts(i, cs, cb); s += cs + cb; }
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
    auto emit = [&](int i, uint32_t cs, uint32_t cb) {
        const uint32_t c = cs + cb;
        ranges[i] = c ? make_uint2(run, run + c) : make_uint2(0u, 0u); // empty tiles read (0,0) like the reference's memset
        tiles[i].start = run;
        tiles[i].cur_big = run + cs;
        run += c;
    };
    if (per <= 8) {
#pragma unroll
        for (int j = 0; j < 8; j++)
            if (j < per && b + j < e) emit(b + j, ks[j], kb[j]);
    } else {
        for (int i = b; i < e; i++) { uint32_t cs, cb; counts(i, cs, cb); emit(i, cs, cb); }
    }
    if (tid == 0) {

        hdr->num_rendered = total;
        hdr->overflow = total > capacity ? 1u : 0u;
        hdr->capacity = capacity;
    }
}

// After the scan: the anchor table (gsr_device.h). One block per anchor tile, one thread per entry.
__global__ void __launch_bounds__(GSR_ANCHOR_ROW)
K_anchor_table(int T, int grid_x, const TileRec* __restrict__ tiles, const uint32_t* __restrict__ tier2,
               const uint32_t* __restrict__ run4, uint32_t* __restrict__ anchor)
{
    const int A = blockIdx.x, e = threadIdx.x;
    if (e >= 100) return;
    int w = 1, h = 1; // decode e -> class (w, h) and tile (dx, dy) of its rectangle
#pragma unroll
    for (int hh = 1; hh <= 4; hh++)
#pragma unroll
        for (int ww = 1; ww <= 4; ww++)
            if (e >= anchor_base(ww, hh)) { w = ww; h = hh; }
    const int k = e - anchor_base(w, h), dx = k % w, dy = k / w;
    const bool first_tier = w <= 2 && h <= 2;
    if (!first_tier && !tier2[0]) return;
    const int ax = A % grid_x, t = A + dy * grid_x + dx;
    if (ax + dx >= grid_x || t >= T) return; // no rectangle of this class is anchored here
    uint32_t pos = tiles[t].start;
    if (first_tier) {
        const int c2 = (w - 1) + 2 * (h - 1);
        const int run = (c2 == 0 ? 0 : c2 == 1 ? 1 : c2 == 2 ? 3 : 5) + (c2 == 3 ? dx + 2 * dy : dx + dy);
        if (run) pos += tiles[t].off[run - 1];
    } else pos += run4[(size_t)t * GSR_RUN4 + (dy * 4 + dx) * 16 + (w - 1) + 4 * (h - 1)];
    anchor[(size_t)A * GSR_ANCHOR_ROW + e] = pos;
}

// the forward's capacity guess was too small: switch the header to the exact capacity before the tail re-runs
__global__ void K_set_capacity(GeomHeader* hdr, uint32_t capacity)
{
    hdr->capacity = capacity;
    hdr->overflow = hdr->num_rendered > capacity ? 1u : 0u;
}

__global__ void __launch_bounds__(256)
K_fill(int P, int grid_x, GeomView g, TileRec* __restrict__ tiles, const uint32_t* __restrict__ anchor,
       uint64_t* __restrict__ pairs)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= P || g.hdr->overflow) return;
    const uint4 br = g.slots[idx]; // everything this pass needs, written by K_preprocess
    const int x0 = (int)(br.z & 0xFFFFu), y0 = (int)(br.z >> 16), x1 = (int)(br.w & 0xFFFFu), y1 = (int)(br.w >> 16);
    const uint64_t key = ((uint64_t)br.y << 32) | (uint32_t)idx;
    const int w = x1 - x0, h = y1 - y0;
    if (w * h == 0) return;
    if (w <= 4 && h <= 4) { // the rank was taken when the splat was counted: no atomics, one table row per splat
        const uint32_t* __restrict__ row = anchor + (size_t)(y0 * grid_x + x0) * GSR_ANCHOR_ROW + anchor_base(w, h);
        const int ntl = w * h;
#pragma unroll
        for (int k = 0; k < 16; k++)
            if (k < ntl) pairs[row[k] + br.x] = key;
    } else {
        for (int y = y0; y < y1; y++)
            for (int x = x0; x < x1; x++) pairs[atomicAdd(&tiles[y * grid_x + x].cur_big, 1u)] = key;
    }
}

// All-ascending bitonic network ("flip" then "disperse" stages): every compare-exchange
// keeps the smaller key at the lower index, so virtual +inf padding above n never moves.
#define GSR_SORT_CAP 4096
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
        for (int grp = threadIdx.x; grp < cap / K; grp += blockDim.x) {
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
    for (int grp = threadIdx.x; grp < cap / K; grp += blockDim.x) { // stages k = 2 .. K inside the group
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
        for (int grp = threadIdx.x; grp < cap / K; grp += blockDim.x) {
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

// Two instantiations share the tiles: SMALL sorts tiles of <= 1024 entries with ONE wave (16 keys per lane
// and trip, 8 KB of LDS, no s_barrier), the other takes the longer lists with 256 threads and 32 KB; each
// skips the other's tiles.
#define GSR_SORT_SMALL 1024
#define GSR_SORT_G 4              // 16 keys per thread and trip
#define GSR_SORT_SMALL_THREADS 64
#define GSR_SORT_BIG_THREADS 256  // 1024 threads per 4096-key tile measured no faster
template <bool SMALL>
__global__ void __launch_bounds__(SMALL ? GSR_SORT_SMALL_THREADS : GSR_SORT_BIG_THREADS)
K_tile_sort(int ntiles, const uint2* __restrict__ ranges, const GeomHeader* __restrict__ hdr,
            uint64_t* __restrict__ pairs, uint32_t* __restrict__ point_list)
{
    __shared__ uint64_t s[SMALL ? GSR_SORT_SMALL : GSR_SORT_CAP];
    const uint32_t tile = xcd_remap(blockIdx.x, ntiles);
    if (hdr->overflow) return;
    const uint2 r = ranges[tile];
    const int n = (int)(r.y - r.x);
    if (n == 0 || (n <= GSR_SORT_SMALL) != SMALL) return;
    uint64_t* seg = pairs + r.x;
    if (n <= GSR_SORT_CAP) {
        int n2 = 1 << GSR_SORT_G;
        while (n2 < n) n2 <<= 1;
        for (int i = threadIdx.x; i < n2; i += blockDim.x) s[swz(i)] = i < n ? seg[i] : ~0ull;
        sort_sync<SMALL>();
        lds_sort<GSR_SORT_G, SMALL>(s, n2);
        for (int i = threadIdx.x; i < n; i += blockDim.x) point_list[r.x + i] = (uint32_t)s[swz(i)];
        return;
    }
    // oversize tile: chunk-local stages in LDS, long-stride stages in global memory
    long n2 = GSR_SORT_CAP;
    while (n2 < n) n2 <<= 1;
    const int nchunks = (int)(n2 / GSR_SORT_CAP);
    for (int c = 0; c < nchunks; c++) {
        const long base = (long)c * GSR_SORT_CAP;
        if (base >= n) break;
        for (int i = threadIdx.x; i < GSR_SORT_CAP; i += blockDim.x) s[swz(i)] = base + i < n ? seg[base + i] : ~0ull;
        __syncthreads();
        lds_sort<GSR_SORT_G, false>(s, GSR_SORT_CAP);
        for (int i = threadIdx.x; i < GSR_SORT_CAP; i += blockDim.x) if (base + i < n) seg[base + i] = s[swz(i)];
        __syncthreads();
    }
    for (long k = 2L * GSR_SORT_CAP; k <= n2; k <<= 1) {
        for (long i = threadIdx.x; i < n2 / 2; i += blockDim.x) { // flip in global memory
            const long blk = i / (k >> 1), off = i % (k >> 1);
            const long lo = blk * k + off, hi = blk * k + (k - 1 - off);
            if (hi < n) { uint64_t a = seg[lo], b = seg[hi]; if (a > b) { seg[lo] = b; seg[hi] = a; } }
        }
        __syncthreads();
        long j = k >> 2;
        for (; j >= GSR_SORT_CAP; j >>= 1) { // disperse with stride >= chunk: global memory
            for (long i = threadIdx.x; i < n2 / 2; i += blockDim.x) {
                const long lo = (i / j) * 2 * j + (i % j), hi = lo + j;
                if (hi < n) { uint64_t a = seg[lo], b = seg[hi]; if (a > b) { seg[lo] = b; seg[hi] = a; } }
            }
            __syncthreads();
        }
        for (int c = 0; c < nchunks; c++) { // remaining strides are chunk-local
            const long base = (long)c * GSR_SORT_CAP;
            if (base >= n) break;
            for (int i = threadIdx.x; i < GSR_SORT_CAP; i += blockDim.x) s[swz(i)] = base + i < n ? seg[base + i] : ~0ull;
            __syncthreads();
            lds_disperse_from<GSR_SORT_G, false>(s, GSR_SORT_CAP, 11); // strides 2048 ... 1 (GSR_SORT_CAP / 2 = 2^11)
            for (int i = threadIdx.x; i < GSR_SORT_CAP; i += blockDim.x) if (base + i < n) seg[base + i] = s[swz(i)];
            __syncthreads();
        }
    }
    for (int i = threadIdx.x; i < n; i += blockDim.x) point_list[r.x + i] = (uint32_t)seg[i];
}

// blending kernels (K_blend_fwd, K_blend_bwd)
} // namespace gsr
#include "gsr_blend.h"
namespace gsr {

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

template <bool REZERO>
__global__ void __launch_bounds__(256)
K_splat_bwd(FrameParams f, SplatInputs in, GeomView g, SplatGrads o)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= f.P) return;
    const size_t i = (size_t)idx;
    const int radius = __float_as_int(g.g1[idx].w);
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
    const float4 q0 = reinterpret_cast<const float4*>(acc)[0], q1 = reinterpret_cast<const float4*>(acc)[1];
    const float q8 = acc[8];
    if (REZERO) { // consumed: leave the record clean for the next backward on this geometry blob
        float4* const ap = reinterpret_cast<float4*>(acc);
        ap[0] = ap[1] = ap[2] = ap[3] = make_float4(0.f, 0.f, 0.f, 0.f); // the whole 64-byte line
    }
    // K_blend_bwd accumulated the raw moments of u = G*dL/dalpha: {u, u dx, u dy, u dx^2, u dx dy, u dy^2};
    // the reference's per-pixel terms (backward.cu:536-554) are these moments times conic / opacity:
    const float4 ga = g.g0[idx], gb = g.g1[idx];
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

    const float3 mean = make_float3(in.means3D[3 * i], in.means3D[3 * i + 1], in.means3D[3 * i + 2]);
    float cov3D[6];
    load_cov3d(in, f, idx, cov3D);

    // ---- conic -> 2D covariance -> 3D covariance and mean (backward.cu:144-274) ----
    const Cov2D k = cov2d_forward(mean, f.focal_x, f.focal_y, f.tan_fovx, f.tan_fovy, cov3D, in.view);
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
    const float* vm = in.view;
    float3 dmean = make_float3(vm[0] * dL_dtx + vm[1] * dL_dty + vm[2] * dL_dtz,
                               vm[4] * dL_dtx + vm[5] * dL_dty + vm[6] * dL_dtz,
                               vm[8] * dL_dtx + vm[9] * dL_dty + vm[10] * dL_dtz);

    // ---- screen-space mean -> 3D mean (backward.cu:366-387) ----
    {
        const float* pj = in.proj;
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

    // ---- 3D covariance -> scale, rotation (backward.cu:278-341) ----
    if (in.scales && o.dL_dscale && o.dL_drot) {
        const float3 sc = make_float3(in.scales[3 * i], in.scales[3 * i + 1], in.scales[3 * i + 2]);
        const float4 q = reinterpret_cast<const float4*>(in.rotations)[i];
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
        st3(o.dL_dscale, i, dsx, dsy, dsz);
#pragma unroll
        for (int n = 0; n < 3; n++) { dMt.m[0][n] *= s.x; dMt.m[1][n] *= s.y; dMt.m[2][n] *= s.z; }
#define GSR_D(cc3, rr3) dMt.m[cc3][rr3]
        float4 dq;
        dq.x = 2 * z * (GSR_D(0, 1) - GSR_D(1, 0)) + 2 * y * (GSR_D(2, 0) - GSR_D(0, 2)) + 2 * x * (GSR_D(1, 2) - GSR_D(2, 1));
        dq.y = 2 * y * (GSR_D(1, 0) + GSR_D(0, 1)) + 2 * z * (GSR_D(2, 0) + GSR_D(0, 2)) + 2 * r * (GSR_D(1, 2) - GSR_D(2, 1)) - 4 * x * (GSR_D(2, 2) + GSR_D(1, 1));
        dq.z = 2 * x * (GSR_D(1, 0) + GSR_D(0, 1)) + 2 * r * (GSR_D(2, 0) - GSR_D(0, 2)) + 2 * z * (GSR_D(1, 2) + GSR_D(2, 1)) - 4 * y * (GSR_D(2, 2) + GSR_D(0, 0));
        dq.w = 2 * r * (GSR_D(0, 1) - GSR_D(1, 0)) + 2 * x * (GSR_D(2, 0) + GSR_D(0, 2)) + 2 * y * (GSR_D(1, 2) + GSR_D(2, 1)) - 4 * z * (GSR_D(1, 1) + GSR_D(0, 0));
#undef GSR_D
        reinterpret_cast<float4*>(o.dL_drot)[i] = dq;
    } else {
        st3(o.dL_dscale, i, 0.f, 0.f, 0.f);
        if (o.dL_drot) reinterpret_cast<float4*>(o.dL_drot)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
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
