// gsr_device.h — shared device-side definitions of the gfx950 rasterizer.
//
// Data layout in HBM (all sub-arrays 256-byte aligned inside caller-owned blobs):
//
//   geometry blob  (gsr_geom_bytes(P)), 248 B per splat:
//     GeomHeader                         256 B   {num_rendered, overflow, capacity of the binning blob}
//     rec  48 B[P]  the per-splat record the blend kernels gather, interleaved so that it costs one L2 line:
//          g0  {x, y, conic_a, conic_b}            pixel centre + half of the conic
//          g1  {conic_c, opacity, depth, radius}   radius stored as int bits (0: culled)
//          col {r, g, b, clamp-flags}              colour the blend uses (SH result or copy of colors_precomp)
//     reach uint2[P]     what culls the splat against the 4x4 patches of a tile, dense (8 bytes: one gather per tile instance
//                        by the tile sort): {reach word (below), centre's patch column | patch row << 16 (int16 each)}
//     pool uint4[ceil(P/256)][8][256]  the bin records of the count and fill passes, BUCKETED: piece (K_preprocess workgroup j, tile window w)
//                        holds, densely from its start, {splat id, depth bits, x0 | y0 << 16, x1 | y1 << 16} (band-clipped tile rectangle) of
//                        the workgroup's splats that cover a tile of window w; pcnt uint16[ceil(P/256)][8] = how many
//     acc  float[P][16]  backward accumulators, one 64-byte line per splat (48-byte records straddle lines and
//                        the L2 atomic rate drops from 20 to 13 G records/s): moments of u = G*dL/dalpha
//                        {sum u, u*dx, u*dy, u*dx^2, u*dx*dy, u*dy^2}, dcolor.rgb, 7 unused
//   image blob     (gsr_image_bytes(W,H)):
//     final_T f32[N], n_contrib u32[N], ranges uint2[T], tile_cnt u32[T], tile_start u32[T],
//     binmat u32[GSR_BIN_ROWS][T] (the count matrix of the binning, below), qcount u32[4T], qdone u32[4T]
//   binning blob   (gsr_binning_bytes(capacity)), 44 B per tile instance:
//     pairs u64[R] (depth bits << 32 | splat id, grouped per tile), point_list u32[R],
//     qhits uint2[4R] (per 8x8 quad: the tile-list entries that reach it, with their 4-bit patch masks; written by the
//                      tile sort, walked front to back by the forward blend and back to front by the backward)
//
// The reference keeps 79 B/splat + 24 B/instance + radix-sort temporaries (rasterizer_impl.h:21-65).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

// This library is written for gfx950 (MI355X) and leans on gfx9 behaviour outside the HIP memory model: hand-overs between workgroups of one launch through
// write-through (sc1) stores, `s_waitcnt vmcnt(0)` and relaxed agent-scope tickets instead of release / acquire fences (csrc/gsr_train.h: last_arriver;
// K_bin_colscan), row_bcast DPP reductions, LDS handed out in 1280-byte granules. On another target they would race or fail to assemble — silently. Refuse to build.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "gsorb-slam_amd is written for gfx950 (CDNA4); its in-launch hand-overs and DPP idioms are not portable: build with --offload-arch=gfx950"
#endif

#define GSR_TILE 16
#define GSR_TILE_PIX 256
#define GSR_ALIGN 256
#define GSR_ACC_STRIDE 16 // floats; 9 used

// Sort capacities (gsr_kernels.hip): lists (rasterizer: chunks of a list) of <= GSR_SORT_SMALL keys are bucket-sorted in LDS by 256
// threads; the k-NN path also sorts buckets of <= GSR_SORT_CAP keys with 256 threads (16 keys per thread).
#define GSR_SORT_SMALL 1024
#define GSR_SORT_CAP 4096


struct GeomHeader {
    uint32_t num_rendered;
    uint32_t overflow;
    uint32_t capacity;
    uint32_t pad[8];   // (the instrumented builds count here)
    uint32_t ticket;   // K_bin_colscan's arrival counter (zeroed by K_preprocess): the last workgroup to arrive scans the tile counts
    uint32_t pad2[52];
};
static_assert(sizeof(GeomHeader) == 256, "header is one aligned slot");

// Tile binning is a two-level counting sort without a single global atomic (the chip retires only ~20 G
// device-scope atomics/s, an LDS atomic costs ~7 ns of one CU: scripts/valu_bench2.hip). The splats are cut into
// `rows` contiguous ranges, one workgroup each:
//   K_bin_count    the workgroup histograms the tiles its splats cover in LDS and writes the histogram as row b
//                  of the count matrix binmat[rows][T]
//   K_bin_colscan  exclusive scan down every column (binmat[b][t] = instances of tile t in rows < b), column
//                  total -> tile_cnt[t]
//   K_scan_tiles   tile_cnt -> tile_start, ranges, num_rendered, overflow
//   K_bin_fill     the same workgroup over the same splats: every (splat, tile) takes its slot from an LDS cursor
//                  that starts at tile_start[t] + binmat[b][t] (returning LDS atomic) and its key goes there
// The matrix has at most GSR_BIN_ROWS rows, so the image blob's size depends on the resolution only. Frames of
// more than 8 x GSR_BIN_WINDOW tiles are refused. Round 5: the bin records leave K_preprocess bucketed by window (GeomView::pool), both
// passes run one workgroup per (splat range, window) and read only the records of their window — one pass over 16P + 8R bytes each
// instead of eight over the fill pass's 16P.
#define GSR_BIN_ROWS 512
// threads of a (splat range, window) workgroup of the count / fill pass: ~560 records in 16 pieces at 1 M splats, a wave per piece (measured,
// count / fill at 128, 256, 512, 1024 threads: 14.3 / 22.5, 11.4 / 20.5, 12.3 / 19.0, 15.1 / 20.5 us)
#ifndef GSR_BINC_THREADS
#define GSR_BINC_THREADS 256
#endif
#ifndef GSR_BINF_THREADS
#define GSR_BINF_THREADS 512
#endif
#define GSR_BIN_NWIN 8       // tile windows = XCDs: window w is tiles [T w / 8, T (w + 1) / 8), counted and filled by workgroups of XCD w
#define GSR_BIN_PIECE 256    // splats per K_preprocess workgroup = bin records per piece of the pool
#define GSR_BIN_WINDOW 40000 // tiles per window: 160 KB of LDS, the whole CU (frames of up to 320 000 tiles = 82 Mpixel; beyond: GSR_EINVAL)
// the window of tile t (window w starts at floor(T w / 8))
__host__ __device__ inline int tile_window(int t, int T) { return (int)((((int64_t)t + 1) * GSR_BIN_NWIN - 1) / T); }
// does window w hold a tile of the rectangle [x0, x1) x [y0, y1)?
__host__ __device__ inline bool window_meets_rect(int w, int T, int grid_x, int x0, int y0, int x1, int y1)
{
    const int a = (int)((int64_t)T * w / GSR_BIN_NWIN), b = (int)((int64_t)T * (w + 1) / GSR_BIN_NWIN); // tiles [a, b)
    for (int y = a / grid_x > y0 ? a / grid_x : y0; y < y1 && y * grid_x + x0 < b; y++) { // the rectangle's row segments are ascending in tile index
        const int lo = y * grid_x + x0 > a ? y * grid_x + x0 : a, hi = y * grid_x + x1 < b ? y * grid_x + x1 : b;
        if (lo < hi) return true;
    }
    return false;
}
__host__ __device__ inline int bin_rows(int P)
{
    // 4096 splats per range: 256 ranges at 1 M splats (measured fill / column scan: 28.5 / 6.7 us; 512 ranges 32.1 / 9.3, 128 ranges 32.7 / 5.1)
#define GSR_BIN_SPLATS 4096
    int r = (P + GSR_BIN_SPLATS - 1) / GSR_BIN_SPLATS;
    if (r > 128) r = (r + 255) & ~255; // whole rounds of the 256 CUs
    return r < 1 ? 1 : (r > GSR_BIN_ROWS ? GSR_BIN_ROWS : r);
}

// The per-splat records the blend kernels gather are interleaved (GSR_GSTRIDE float4 per splat): a random
// gather pulls a whole 128-byte L2 line per touched address, so g0, g1 and col of one splat share a line
// (48-byte records). Measured at 1 M splats, forward / backward blend: separate arrays 182 / 294 us,
// stride 2 (g0,g1) 172 / 289, stride 3 169 / 288, stride 4 (64-B records) 168 / 289 but slower per-splat
// kernels (step 0.703 vs 0.696 ms).
#define GSR_GSTRIDE 3
struct Strided4 {
    float4* p;
    __host__ __device__ float4& operator[](size_t i) const { return p[GSR_GSTRIDE * i]; }
};
struct GeomView {
    GeomHeader* hdr;
    Strided4 g0;
    Strided4 g1;
    Strided4 col;
    uint2* reach; // {reach word: bit 3 small, bits 4..28 the 5x5 patch window; (int16) floor(px/4) | (int16) floor(py/4) << 16}
    uint4* pool;    // [ceil(P/256)][8][256] {splat id, depth bits, x0 | y0 << 16, x1 | y1 << 16}: the bin records, bucketed by (K_preprocess workgroup, tile window)
    uint16_t* pcnt; // [ceil(P/256)][8] records in each piece
    float* acc;
};
struct ImageView {
    float* final_T;
    uint32_t* n_contrib;
    uint2* ranges;
    uint32_t* tile_cnt;   // [T] instances per tile
    uint32_t* tile_start; // [T] where the tile's list segment starts
    uint32_t* binmat;     // [GSR_BIN_ROWS][T] count matrix (after K_bin_colscan: exclusive column prefixes)
    uint32_t* qcount; // [4*T] quad-hit records per 8x8 quad (written by the tile sort)
    uint32_t* qdone;  // [4*T] how many of them the forward blend consumed before every pixel of the quad was done
};
struct BinView {
    uint64_t* pairs;
    uint32_t* point_list;
    uint2* qhits;     // [4*R] (list position, splat id | 4-bit patch mask << 28) of every list entry that can reach a quad, in
                      // list order; the quad q of a tile with range [x, x+n) owns qhits[4x + q*n .. 4x + (q+1)*n)
};

__host__ __device__ inline size_t gsr_align_up(size_t x) { return (x + GSR_ALIGN - 1) & ~(size_t)(GSR_ALIGN - 1); }

__host__ __device__ inline size_t geom_layout(char* base, int P, GeomView* v)
{
    size_t off = 0, Pz = P > 0 ? (size_t)P : 1;
    GeomView g;
    g.hdr = (GeomHeader*)(base + off); off = gsr_align_up(off + sizeof(GeomHeader));
    g.g0.p = (float4*)(base + off);
    g.g1.p = g.g0.p + 1;
    g.col.p = g.g0.p + 2;
    off = gsr_align_up(off + Pz * 16 * GSR_GSTRIDE);
    g.reach = (uint2*)(base + off); off = gsr_align_up(off + Pz * 8);
    const size_t pieces = (Pz + GSR_BIN_PIECE - 1) / GSR_BIN_PIECE * GSR_BIN_NWIN;
    g.pool = (uint4*)(base + off); off = gsr_align_up(off + pieces * GSR_BIN_PIECE * 16);
    g.pcnt = (uint16_t*)(base + off); off = gsr_align_up(off + pieces * 2);
    g.acc = (float*)(base + off); off = gsr_align_up(off + Pz * GSR_ACC_STRIDE * 4);
    if (v) *v = g;
    return off;
}
__host__ __device__ inline size_t image_layout(char* base, int W, int H, ImageView* v)
{
    size_t off = 0, N = (size_t)W * H;
    size_t T = (size_t)((W + GSR_TILE - 1) / GSR_TILE) * ((H + GSR_TILE - 1) / GSR_TILE);
    if (N == 0) N = 1;
    if (T == 0) T = 1;
    ImageView g;
    g.final_T = (float*)(base + off); off = gsr_align_up(off + N * 4);
    g.n_contrib = (uint32_t*)(base + off); off = gsr_align_up(off + N * 4);
    g.ranges = (uint2*)(base + off); off = gsr_align_up(off + T * 8);
    g.tile_cnt = (uint32_t*)(base + off); off = gsr_align_up(off + T * 4);
    g.tile_start = (uint32_t*)(base + off); off = gsr_align_up(off + T * 4);
    g.binmat = (uint32_t*)(base + off); off = gsr_align_up(off + T * GSR_BIN_ROWS * 4);
    g.qcount = (uint32_t*)(base + off); off = gsr_align_up(off + T * 16);
    g.qdone = (uint32_t*)(base + off); off = gsr_align_up(off + T * 16);
    if (v) *v = g;
    return off;
}
__host__ __device__ inline size_t binning_layout(char* base, size_t R, BinView* v)
{
    size_t off = 0;
    if (R == 0) R = 1;
    BinView g;
    g.pairs = (uint64_t*)(base + off); off = gsr_align_up(off + R * 8);
    g.point_list = (uint32_t*)(base + off); off = gsr_align_up(off + R * 4);
    g.qhits = (uint2*)(base + off); off = gsr_align_up(off + R * 32);
    if (v) *v = g;
    return off;
}
// largest R such that binning_layout(R) <= bytes
__host__ inline size_t binning_capacity(size_t bytes)
{
    if (bytes < 3 * GSR_ALIGN) return 0;
    size_t r = (bytes - 3 * GSR_ALIGN) / 44;
    while (r > 0 && binning_layout(nullptr, r, nullptr) > bytes) r--;
    return r;
}

// Camera / frame constants shared by the per-splat kernels.
struct FrameParams {
    int P, D, M;
    int W, H;
    int grid_x, grid_y;
    float tan_fovx, tan_fovy;
    float focal_x, focal_y;
    float scale_modifier;
    int band_y0, band_y1; // tile rows this call bins and blends (whole image: 0, grid_y)
    int fold_depth_color; // backward: the fused depth channel's colour gradient goes to the mean (gsr_backward_args.ds_detach_depth == 0)
};

// ---------------------------------------------------------------------------------
// Pixel/splat pair evaluation shared by the forward and backward blend kernels.
// The backward re-takes the forward's skip decisions, so both MUST produce the
// same bits: the library is built with -ffp-contract=off and the contractions
// below are explicit.
// ---------------------------------------------------------------------------------
// log2(e) * power, power = -0.5*(a dx^2 + c dy^2) - b dx dy (reference forward.cu:348, backward.cu:492),
// from a conic the staging code has already scaled: ca2 = -0.5*log2(e)*a, cb2 = -log2(e)*b,
// cc2 = -0.5*log2(e)*c. Five VALU ops; forward and backward call this one function on identically
// staged values, so both see bit-identical alphas (v_exp_f32 is a base-2 exponential).
#define GSR_LOG2E 1.4426950408889634f
__device__ __forceinline__ float pair_power2(float dx, float dy, float ca2, float cb2, float cc2)
{
    const float t = fmaf(ca2, dx, cb2 * dy);
    return fmaf(t, dx, (cc2 * dy) * dy);
}

// Exact patch reach of a small splat, decided once per splat by K_preprocess; the forward blend only shifts bits.
// ---------------------------------------------------------------------------------
#define GSR_ALPHA_MIN (1.0f / 255.0f)
// Exact reach of one splat over a WINDOW of NW x NW patches (4x4 pixels each) whose first patch has its pixel centres
// at (wx, wy) .. (wx + 3, wy + 3): bit (j * NW + i) of the result = the iso-alpha ellipse alpha = 1/255 meets patch (i, j).
// alpha >= 1/255  <=>  Q(d) := 0.5*(a dx^2 + c dy^2) + b dx dy <= log2(255*opacity) in log2 units, d = centre - pixel.
// The minimum of the convex Q over a rectangle is attained at d = 0 (centre inside), or on a side facing the centre,
// where it is a 1-D parabola clamped to the side. Conservative: the continuous rectangle contains the pixel centres,
// the threshold carries a margin far above fp32 rounding, NaNs pass. Everything that depends on one axis only is
// hoisted out of the NW x NW loop.
struct ReachSetup {
    float ax, ay, hca, hcc, cb, kx, ky, tau;
    bool pd, never, small, medium;
};
__device__ __forceinline__ ReachSetup reach_setup(float px, float py, float conic_a, float conic_b, float conic_c, float op)
{
    ReachSetup s;
    const float ca = conic_a * GSR_LOG2E, cb = conic_b * GSR_LOG2E, cc = conic_c * GSR_LOG2E;
    const float det = ca * cc - cb * cb;
    // the construction needs a positive-definite conic; an indefinite one (possible with cov3D_precomp) is never culled:
    // the reference would still blend it (forward.cu:346-358)
    s.pd = ca > 0.f && cc > 0.f && det > 0.f;
    s.never = !(op >= GSR_ALPHA_MIN) && op == op; // alpha = op * exp(power <= 0) can never reach 1/255
    s.ax = px; s.ay = py; s.hca = 0.5f * ca; s.hcc = 0.5f * cc; s.cb = cb;
    s.kx = -cb * __builtin_amdgcn_rcpf(cc); s.ky = -cb * __builtin_amdgcn_rcpf(ca);
    s.tau = __log2f(255.0f * op) + 0.0145f;
    // half extents of the ellipse Q <= tau: ex^2 = 2 tau cc / det, ey^2 = 2 tau ca / det; "small": both below ~7.6 pixels,
    // so that the ellipse cannot leave the 5x5 patches around the patch that holds its centre (SinglePixel-initialised
    // splats reach 2.2 .. 6.3 pixels)
    const float lim = 64.0f * 0.9f * det;
    s.small = s.pd && 2.f * s.tau * cc <= lim && 2.f * s.tau * ca <= lim;
    // "medium": both half extents below ~15.2 pixels — the ellipse cannot leave the 5x5 QUADS (8x8 pixels) around the quad that
    // holds its centre. Its reach is decided per quad (a quad that is reached counts as reached in all four of its patches:
    // the blend loop tests every pixel anyway); splats that have grown beyond their single-pixel initialisation live here.
    s.medium = s.pd && !s.small && 2.f * s.tau * cc <= 4.f * lim && 2.f * s.tau * ca <= 4.f * lim;
    return s;
}
// `cell`: pixels per cell side (4: patches, 8: quads) — a per-lane value, so that small and medium splats of a wave share the code
template <int NW>
__device__ __forceinline__ uint32_t window_reach(const ReachSetup& s, float wx, float wy, float cell = 4.f)
{
    float dl[NW], dh[NW], el[NW], eh[NW], Ax[NW], Bx[NW], Kx[NW], Ay[NW], By[NW], Ky[NW];
    uint32_t zc = 0u, zr = 0u; // columns / rows whose range contains the centre coordinate
#pragma unroll
    for (int i = 0; i < NW; i++) {
        dl[i] = s.ax - (wx + cell * i + (cell - 1.f)); dh[i] = s.ax - (wx + cell * i);
        const float dc = __builtin_amdgcn_fmed3f(0.f, dl[i], dh[i]); // point of the range closest to 0 (one v_med3 instead of max + min; a NaN gives the lower end, like the pair did)
        Ax[i] = dc != 0.f ? s.hca * dc * dc : 3.0e38f;    // a side faces the centre only if the centre is outside the range
        Bx[i] = s.cb * dc; Kx[i] = s.kx * dc;
        zc |= dc == 0.f ? 1u << i : 0u;
        el[i] = s.ay - (wy + cell * i + (cell - 1.f)); eh[i] = s.ay - (wy + cell * i);
        const float ec = __builtin_amdgcn_fmed3f(0.f, el[i], eh[i]);
        Ay[i] = ec != 0.f ? s.hcc * ec * ec : 3.0e38f;
        By[i] = s.cb * ec; Ky[i] = s.ky * ec;
        zr |= ec == 0.f ? 1u << i : 0u;
    }
    uint32_t mask = 0u;
#pragma unroll
    for (int j = 0; j < NW; j++) {
        if (zr & (1u << j)) mask |= zc << (j * NW); // centre inside the patch rectangle
#pragma unroll
        for (int i = 0; i < NW; i++) {
            const float dy = __builtin_amdgcn_fmed3f(Kx[i], el[j], eh[j]);         // minimiser on the vertical side x = dc_i
            const float qx = fmaf(dy, fmaf(s.hcc, dy, Bx[i]), Ax[i]);
            const float dx = __builtin_amdgcn_fmed3f(Ky[j], dl[i], dh[i]);         // minimiser on the horizontal side y = ec_j
            const float qy = fmaf(dx, fmaf(s.hca, dx, By[j]), Ay[j]);
            if (!(fminf(qx, qy) > s.tau)) mask |= 1u << (j * NW + i);
        }
    }
    return mask;
}
// The reach word K_preprocess writes into the splat's entry of the dense reach array (neither bit 2 nor bit 3: decided per
// tile instance by the tile sort, exact_tile_mask):
//   bit 3 "small"  : the 25 bits below are the splat's complete patch reach
//   bits 4..28     : reach over the 5x5 patches whose first patch is column floor(px/4) - 2, row floor(py/4) - 2
//                    of the GLOBAL patch grid (patch column c = pixels 4c .. 4c+3)
//   bit 2 "medium" : the 25 bits below are the splat's complete QUAD reach: 5x5 quads (8x8 pixels) whose first quad is column
//                    floor(px/8) - 2, row floor(py/8) - 2 of the global quad grid
#define GSR_REACH_SMALL 8u
#define GSR_REACH_MEDIUM 4u
__device__ __forceinline__ uint32_t splat_reach25(float px, float py, float conic_a, float conic_b, float conic_c, float op)
{
    const ReachSetup s = reach_setup(px, py, conic_a, conic_b, conic_c, op);
    if (s.never) return GSR_REACH_SMALL; // reaches nothing
    if (!s.small && !s.medium) return 0u;
    const float cell = s.small ? 4.f : 8.f, inv = s.small ? 0.25f : 0.125f;
    const float wx = cell * (floorf(px * inv) - 2.f), wy = cell * (floorf(py * inv) - 2.f);
    return (s.small ? GSR_REACH_SMALL : GSR_REACH_MEDIUM) | (window_reach<5>(s, wx, wy, cell) << 4);
}
// The 4x4 patches of a whole 16x16 tile (tx, ty): bit (j * 4 + i) = patch (4 tx + i, 4 ty + j), from the
// splat's entry of the dense reach array (K_preprocess). Splats that are neither "small" nor "medium" get GSR_MASK_UNTESTED.
#define GSR_MASK_UNTESTED 0x10000u
__device__ __forceinline__ uint2 reach_entry(uint32_t word, float px, float py)
{
    const int pcx = max(-32768, min(32767, (int)floorf(px * 0.25f))), pcy = max(-32768, min(32767, (int)floorf(py * 0.25f)));
    return make_uint2(word & ~3u, ((uint32_t)pcx & 0xFFFFu) | ((uint32_t)pcy << 16));
}
__device__ __forceinline__ uint32_t tile_mask_from_reach(uint2 re, int tx, int ty)
{
    if (!(re.x & (GSR_REACH_SMALL | GSR_REACH_MEDIUM))) return GSR_MASK_UNTESTED;
    const uint32_t m = re.x >> 4;
    if (!(re.x & GSR_REACH_SMALL)) { // medium: quad (2 tx + i, 2 ty + j) of the tile, reached = all four of its patches
        const int sx = 2 * tx - ((int)(short)(re.y & 0xFFFFu) >> 1) + 2; // window column / row of the tile's first quad
        const int sy = 2 * ty - (((int)re.y >> 16) >> 1) + 2;
        uint32_t out = 0u;
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const int wc = sx + i, wr = sy + j;
                const bool hit = (uint32_t)wc < 5u && (uint32_t)wr < 5u && ((m >> (5 * wr + wc)) & 1u) != 0u;
                out |= hit ? 0x33u << (8 * j + 2 * i) : 0u;
            }
        return out;
    }
    const int sx = 4 * tx - (int)(short)(re.y & 0xFFFFu) + 2; // window column / row of the tile's first patch
    const int sy = 4 * ty - ((int)re.y >> 16) + 2;
    uint32_t out = 0u;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int wr = sy + j;
        const uint32_t row5 = (uint32_t)wr < 5u ? (m >> (5 * wr)) & 31u : 0u;
        const uint32_t bits = sx >= 0 ? (sx < 5 ? row5 >> sx : 0u) : (sx > -4 ? row5 << (-sx) : 0u);
        out |= (bits & 15u) << (4 * j);
    }
    return out;
}
// the 2x2 patches of quad q (bit 0: x half, bit 1: y half) out of a tile mask: bit (j * 2 + i)
__device__ __forceinline__ uint32_t quad_mask_of_tile_mask(uint32_t m16, int q)
{
    const uint32_t lo = (m16 >> (8 * (q >> 1) + 2 * (q & 1))) & 3u, hi = (m16 >> (8 * (q >> 1) + 4 + 2 * (q & 1))) & 3u;
    return lo | (hi << 2);
}

// Wave-wide inclusive scan / reduction in eight DPP instructions (row_shr 1, 2, 4, 8 inside the rows of 16 lanes, then
// row_bcast15 / row_bcast31 across rows) instead of six dependent ds_bpermute round trips per __shfl scan.
template <int CTRL, int ROWS>
__device__ __forceinline__ uint32_t dpp_u(uint32_t old, uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, ROWS, 0xf, false);
}
__device__ __forceinline__ uint32_t wave_scan_add(uint32_t v) // inclusive
{
    v += dpp_u<0x111, 0xf>(0u, v);
    v += dpp_u<0x112, 0xf>(0u, v);
    v += dpp_u<0x114, 0xf>(0u, v);
    v += dpp_u<0x118, 0xf>(0u, v);
    v += dpp_u<0x142, 0xa>(0u, v); // lane 15 of rows 0, 2 -> rows 1, 3
    v += dpp_u<0x143, 0xc>(0u, v); // lane 31 -> rows 2, 3
    return v;
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) // the result, in every lane
{
    v = min(v, dpp_u<0x111, 0xf>(v, v)); v = min(v, dpp_u<0x112, 0xf>(v, v));
    v = min(v, dpp_u<0x114, 0xf>(v, v)); v = min(v, dpp_u<0x118, 0xf>(v, v));
    v = min(v, dpp_u<0x142, 0xa>(v, v)); v = min(v, dpp_u<0x143, 0xc>(v, v));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ uint32_t wave_max_dpp(uint32_t v)
{
    v = max(v, dpp_u<0x111, 0xf>(v, v)); v = max(v, dpp_u<0x112, 0xf>(v, v));
    v = max(v, dpp_u<0x114, 0xf>(v, v)); v = max(v, dpp_u<0x118, 0xf>(v, v));
    v = max(v, dpp_u<0x142, 0xa>(v, v)); v = max(v, dpp_u<0x143, 0xc>(v, v));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
    for (int off = 32; off > 0; off >>= 1) {
        uint32_t o = (uint32_t)__shfl_xor((int)v, off, 64);
        v = v > o ? v : o;
    }
    return v;
}

// XCD-aware block remap: consecutive block ids round-robin over the 8 XCDs (block b runs on XCD b & 7). Neighbouring tiles share splats, so
// neighbours should meet in one L2 — but the WORK must still reach all eight XCDs when it is concentrated in a part of the frame: one rank of a
// sharded map renders a k-d cell that covers a quadrant of the image, and with XCD x owning the x-th contiguous band of tiles (rounds 1-5) a
// quadrant's 3 225 blend jobs ran on four of the eight XCDs — 1 536 wave slots, two rounds: K_blend_bwd 127 us for a quarter of the
// headline's work (218 us), K_blend_fwd 55 of 94 (profiles/r06_rank_regime.md). The jobs are therefore dealt out in GROUPS of `group`
// consecutive jobs (GSR_XCD_TILES tiles), round-robin over the XCDs: every XCD samples the whole frame, a group's tiles (and the four quads of a
// tile) still share an L2. The last, partial super-group of 8 * group jobs keeps the identity map (a bijection for any nblocks).
#ifndef GSR_XCD_TILES
#define GSR_XCD_TILES 16 // tiles per group of the BLEND kernels (0: the contiguous bands of rounds 1-5). Measured, whole bench line, two boxes (gpurun_out/ab_xcd*,
                         // ms per step, bands -> 16-tile groups): headline 0.453 -> 0.449, one rank's quadrant of the 1 M map 0.260 -> 0.214, an eighth of the 2 M map at 640x480
                         // 0.233 -> 0.208, 2 M frame 0.536 -> 0.538, fat x4 0.670 -> 0.668; groups of 1 / 2 / 4 tiles cost the long-list frames 5-15 % (fat x4 0.78 / 0.75 / 0.71)
#endif
#ifndef GSR_XCD_SORT_TILES
#define GSR_XCD_SORT_TILES 0 // the tile sort keeps the contiguous bands: its keys were written by K_bin_fill's workgroup of the SAME tile window = XCD (sort in groups like the
                             // blend kernels: 2 M frame 0.538 -> 0.543, fat x4 0.668 -> 0.676; the rank regime 0.214 -> 0.209: the sort's share of the imbalance is small)
#endif
__device__ __forceinline__ uint32_t xcd_remap(uint32_t b, uint32_t nblocks, uint32_t group)
{
    if (group == 0u) {
        const uint32_t xcd = b & 7u, q = nblocks >> 3, r = nblocks & 7u;
        const uint32_t base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        return base + (b >> 3);
    }
    const uint32_t full = nblocks - nblocks % (8u * group); // jobs in whole super-groups
    if (b >= full) return b;
    const uint32_t xcd = b & 7u, slot = b >> 3;
    return ((slot / group) * 8u + xcd) * group + slot % group;
}
