// gsr_blend.h — the two blend kernels (forward alpha compositing and its backward).
//
// Mapping (wave64-first): ONE wave per 8x8 pixel quad; the four quads of a 16x16 tile are four
// INDEPENDENT single-wave workgroups that walk the same per-tile list, and inside the wave the four
// 16-lane DPP rows are four independent 4x4 pixel "patches", each with its own hit list ("patch rows").
//   * a workgroup is a single wave: no s_barrier anywhere; LDS holds the parked entries, the per-patch
//     byte lists and (backward) the accumulators, all written and read by the wave in program order;
//   * small splats light ~16 of the 64 lanes of a quad but ~8 of the 16 lanes of a patch, and a patch
//     is hit by half as many splats as the quad: the wave retires ~0.6x the iterations of a
//     one-splat-per-wave-iteration loop;
//   * block ids are remapped so the four quads of a tile (and neighbouring tiles) run on the same XCD
//     and share its L2 for the per-splat gathers.
// Culling is EXACT at both levels (does the iso-alpha ellipse alpha = 1/255 intersect the rectangle of
// pixel centres of the quad / of the patch?). The forward culls and logs its verdicts (qhits: list
// position, id, 4-bit patch mask); the backward walks the log and never culls.
//
// What is computed per (pixel, splat) pair is the reference's arithmetic
// (DGR/cuda_rasterizer/forward.cu:339-391, backward.cu:470-555).
#pragma once

#include "gsr_device.h"
#include <type_traits>

namespace gsr {

// a quad-hit record keeps the splat id in the low 28 bits of .y and the 4-bit patch mask above it
#define GSR_ID_BITS 28
#define GSR_ID_MASK 0x0FFFFFFFu

// number of set bits of m below this lane
__device__ __forceinline__ int mbcnt64(unsigned long long m)
{
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// Lane predicates as wave masks and back. Combining conditions on the MASKS is scalar-unit work (free next to a
// VALU-bound loop) and lets one v_cmp serve a condition and its negation (the compiler otherwise emits a second
// compare for !(x < c)); only v_cmp and v_cndmask — both half-rate on gfx950 — remain on the vector side.
// A single-wave workgroup exchanges data through LDS in program order (the LDS queue of a wave is in order); the only
// thing to prevent is the COMPILER moving LDS accesses across the hand-over point (the same block is viewed through
// differently typed pointers).
__device__ __forceinline__ void lds_turn()
{
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}
typedef unsigned long long wmask;
__device__ __forceinline__ wmask wm(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ bool lane_of(wmask m) { return __builtin_amdgcn_inverse_ballot_w64(m); }

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
    // the construction below needs a positive-definite conic; an indefinite one (possible with cov3D_precomp:
    // ca, cc > 0 but ca*cc <= cb^2) is not culled: the reference would still blend it (forward.cu:346-358)
    if (!(ca > 0.f) || !(cc > 0.f) || !(ca * cc > cb * cb)) return true;
    const float tau = __logf(255.0f * op) + 0.01f;
    const float dxl = a.x - (X0 + 7.f), dxh = a.x - X0, dyl = a.y - (Y0 + 7.f), dyh = a.y - Y0;
    const float dxc = fminf(fmaxf(0.f, dxl), dxh), dyc = fminf(fmaxf(0.f, dyl), dyh); // point of the range closest to 0
    if (dxc == 0.f && dyc == 0.f) return true;
    float q = 3.0e38f;
    if (dxc != 0.f) { // a vertical side faces the centre: minimise over dy
        const float dy = fminf(fmaxf(-cb * dxc * __builtin_amdgcn_rcpf(cc), dyl), dyh); // minimiser: its rounding enters q to 2nd order
        q = fminf(q, 0.5f * (ca * dxc * dxc + cc * dy * dy) + cb * dxc * dy);
    }
    if (dyc != 0.f) {
        const float dx = fminf(fmaxf(-cb * dyc * __builtin_amdgcn_rcpf(ca), dxl), dxh);
        q = fminf(q, 0.5f * (ca * dx * dx + cc * dyc * dyc) + cb * dx * dyc);
    }
    return !(q > tau);
}

// =====================================================================================
// Backward ("patch rows", deferred reduction): the wave owns an 8x8 quad, its four 16-lane rows are four
// INDEPENDENT 4x4 pixel patches, each walking its own hit list. Small splats touch few pixels of an 8x8 quad
// (~16 of 64 lanes useful on the 1 M-splat workload); a 4x4 patch is hit by half as many splats as the quad and
// uses ~8 of its 16 lanes, so the same wave retires ~1.7x fewer iterations.
//   gather : the forward logged which list entries reach the quad and which of its patches (qhits), so
//            the backward never touches the rest of the tile list: 64 records per step, back to front,
//            records behind every pixel's last contributor dropped, the rest parked in LDS;
//   lists  : one lane per parked entry appends the entry's byte offsets to the list of every patch in its mask
//            (list order is kept); the lists are padded with a dummy entry to a common even length;
//   blend  : row r walks list r (entries software-pipelined through two register sets). An iteration only does the
//            per-PIXEL arithmetic (alpha, T, the accum_rec recursion, dL/dalpha) and parks two numbers per lane in
//            an LDS ring: u = G * dL/dalpha and dcol = alpha * T;
//   reduce : every 16 iterations the wave turns around: one lane per (row, iteration) pair — 64 pairs — reads the
//            16 pixels of its pair from the ring and forms the nine per-splat sums (moments of u about the splat
//            centre, dcol . dL/dpixel) with plain register FMAs, then adds them to the entry's LDS accumulator,
//            row after row (the entries of one row's list are distinct: no atomics, no collision handling).
//            The transposition is done by LDS addressing instead of a 21-instruction DPP butterfly per iteration
//            (all DPP ops, v_cndmask, v_cmp, v_min/max run at half the rate of v_fma on gfx950:
//            scripts/valu_bench2.hip), and the 8 moment / colour products per lane leave the loop as well;
//   flush  : LDS accumulators -> one 9-lane global atomic per parked entry that was hit (7 per
//            instruction), i.e. one L2 atomic record per (quad, splat). Issuing the atomics per PATCH instead
//            would double them and hit the L2 atomic ceiling (~20 G records/s, scripts/atomic_bench2.hip).
// =====================================================================================
#define GSR_ROWQ 64 // parked entries per round of the forward = one gather step
#ifndef GSR_RING
#define GSR_RING 16 // iterations between two reduce phases (<= 16, even): one (row, ring slot) pair per lane of the first GSR_RING lanes of every row
#endif
#ifndef GSR_BSTEP
#define GSR_BSTEP 64 // records the backward takes (and parks) per round (<= 64: one parked entry per lane)
#endif
#ifndef GSR_BWD_WAVES
#define GSR_BWD_WAVES 3 // waves per SIMD the backward is compiled for (512 / GSR_BWD_WAVES registers)
#endif
#ifndef GSR_BWD_WIDE
#define GSR_BWD_WIDE 0 // lean body's reduce phase with every ring read ahead of the first use (measured slower: 219 vs 216 us)
#endif
#ifndef GSR_BWD_LEAN_ALWAYS
#define GSR_BWD_LEAN_ALWAYS 0 // experiment hook: the plain render through the lean body as well
#endif
typedef float v2f __attribute__((ext_vector_type(2)));
#define GSR_INV_NONE 0x10u // "entry not in this row" byte of INV (slots are 0..15); 0x10101010 is a finite float

// Exact cull of one parked entry (conic staged for pair_power2, i.e. in log2 units) against the 2x2
// patches of 4x4 pixel centres of the quad at (X0, Y0); same construction and margin as quad_reach.
// h[j*2+i]: x-half i, y-half j.
__device__ __forceinline__ void patch_reach4(const float4 A, const float4 B, float X0, float Y0, bool (&h)[4])
{
    const float ca = -2.f * A.z, cb = -A.w, cc = -2.f * B.x; // log2(e) * (a, b, c)
    const bool degenerate = !(ca > 0.f) || !(cc > 0.f) || !(ca * cc > cb * cb); // not positive definite: never culled
    const float tau = __log2f(255.0f * B.y) + 0.0145f;
    float dl[2], dh[2], dc[2], el[2], eh[2], ec[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        dl[i] = A.x - (X0 + 4.f * i + 3.f); dh[i] = A.x - (X0 + 4.f * i);
        dc[i] = fminf(fmaxf(0.f, dl[i]), dh[i]);
        el[i] = A.y - (Y0 + 4.f * i + 3.f); eh[i] = A.y - (Y0 + 4.f * i);
        ec[i] = fminf(fmaxf(0.f, el[i]), eh[i]);
    }
    // minimiser of the 1-D parabola on a side; its rounding error enters q only to second order
    const float kx = -cb * __builtin_amdgcn_rcpf(cc), ky = -cb * __builtin_amdgcn_rcpf(ca);
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const float dxc = dc[i], dyc = ec[j];
            const float dy = fminf(fmaxf(kx * dxc, el[j]), eh[j]);
            const float qx = 0.5f * (ca * dxc * dxc + cc * dy * dy) + cb * dxc * dy;
            const float dx = fminf(fmaxf(ky * dyc, dl[i]), dh[i]);
            const float qy = 0.5f * (ca * dx * dx + cc * dyc * dyc) + cb * dx * dyc;
            const float q = fminf(dxc != 0.f ? qx : 3.0e38f, dyc != 0.f ? qy : 3.0e38f);
            h[j * 2 + i] = degenerate || (dxc == 0.f && dyc == 0.f) || !(q > tau);
        }
}

#ifdef GSR_EXP_TIMELINE // instrumented build (scripts/timeline.py): start / end time and placement of every backward workgroup
__device__ unsigned long long g_timeline[4 * 65536];
struct TimelineMark {
    unsigned long long t0; uint32_t b;
    __device__ TimelineMark(uint32_t block) : t0(wall_clock64()), b(block) {}
    __device__ ~TimelineMark() {
        if (threadIdx.x == 0 && b < 65536u) {
            g_timeline[4 * b] = t0; g_timeline[4 * b + 1] = wall_clock64();
            g_timeline[4 * b + 2] = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_ID
            g_timeline[4 * b + 3] = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // XCC_ID
        }
    }
};
#endif
// ---- the WIDE body (plain colour render): everything a reduce phase reads goes out before its first use, the blend loop's software
//      pipeline runs across the reduce phases — 168 registers, 12 800 bytes of LDS, three waves per SIMD.
template <int Q>
__device__ __forceinline__ void blend_bwd_rgb(ImageView im, char* __restrict__ binning, GeomView g, const float* __restrict__ bg, int W, int H,
                                              int grid_x, int ntiles, int tile0, const float* __restrict__ dL_dpix)
{
    constexpr int WRING = 16;     // iterations between two reduce phases: 4 rows x 16 = one (row, iteration) pair per lane
    constexpr int WACCW = 12;     // floats per staged record (nine used): 48 bytes, so that it moves as three b128
    static_assert(Q == 64, "one lane per parked entry and one lane per (row, ring slot) pair");
    // parked entries; slot Q is a dummy (opacity 0, far away) the per-patch lists are padded with: no "row still active"
    // compare and no index select in the blend loop
    __shared__ float4 E0[Q + 1], E1[Q + 1], E2[Q + 1]; // (px, py, a2, b2) (c2, opacity, red, green) (blue, list position, view depth, splat id | patch mask << 28)
    // per-patch hit lists as BYTE OFFSETS (entry * 16 into E0/E1/E2): shifts and integer mads are half-rate VALU work
    // LDS budget: gfx950 hands LDS out in blocks of 1280 bytes (scripts/lds_granule.hip: 12 800 bytes per workgroup -> 12 single-wave
    // workgroups per CU, 12 816 -> 11), so the kernel is held at exactly ten blocks — the 12 waves per CU its registers allow.
    using blist_t = uint16_t;
    constexpr uint32_t LUNIT = 16u;
    auto loff = [](const blist_t x) -> uint32_t { return (uint32_t)x; };
    __shared__ blist_t LIST[4 * (Q + 4)];
    // One block of LDS used three ways, one after the other:
    //  UD  the ring: (u, dcol) of pixel p of pair q = row * 16 + (iteration % 16) at float2 UD[p * 65 + q]. A blend iteration
    //      writes 16 consecutive p for 4 values of q (stride 65 float2: the 16 lanes of a row fall on 16 different bank
    //      pairs), the reduce phase reads 64 consecutive q for one p: both conflict-free;
    //  ST  float4 ST[3 * 64]: the nine sums of pair q at ST[k * 64 + q], k = 0..2 (reduce phase -> merge by entry);
    //  ACC float ACC[64 * 12]: the per-entry totals of the round, staged for the coalesced flush.
    __shared__ float4 POOL[(16 * (4 * WRING + 1) * 8) / 16];
    // dL/dpixel of pixel p of patch r, one plane per channel (17: the four rows on different banks; planes instead of float4:
    // the plain render has three channels, and the reduce phase holds 12 instead of 16 registers of them)
    __shared__ float GP[3][4 * 17];
#ifdef GSR_EXP_LDSPAD // occupancy experiment: more LDS per wave, fewer waves per SIMD
    __shared__ uint32_t PADX[GSR_EXP_LDSPAD];
    if (W == -1) PADX[threadIdx.x] = 1u;
#endif
    v2f* const UD = reinterpret_cast<v2f*>(POOL);
    float4* const ST = POOL;
    // INV, per reduce phase: byte r of word e = ring slot of entry e in row r, or GSR_INV_NONE. It lives behind ST inside the
    // block (it is only alive while the block is ST); its words are finite as floats, so the ring slots it leaves behind are
    // harmless when they are read as stale slots
    uint32_t* const INV = reinterpret_cast<uint32_t*>(POOL + 3 * 64);
#ifdef GSR_EXP_TIMELINE
    TimelineMark mark(blockIdx.x);
#endif
    const uint32_t w = xcd_remap(blockIdx.x, 4u * (uint32_t)ntiles, 4u * GSR_XCD_TILES);
    const uint32_t tile = (uint32_t)tile0 + (w >> 2), quad = w & 3u;
    const int tx = tile % grid_x, ty = tile / grid_x;
    const int lane = threadIdx.x, r = lane >> 4, l = lane & 15;
    const int X0 = tx * 16 + (int)(quad & 1u) * 8, Y0 = ty * 16 + (int)(quad >> 1) * 8;
    const int X0p = X0 + (r & 1) * 4, Y0p = Y0 + (r >> 1) * 4; // origin of this row's patch
    const int px = X0p + (l & 3), py = Y0p + (l >> 2);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    // The job's first trips to memory (round 4: four dependent ones instead of six — at ~2.4 waves per SIMD nothing hides them): the tile's
    // range, the header words and the quad's record count are asked for together, the pixel's own words with them, and the first records
    // as soon as those scalars are there (clamped, always valid addresses) — not behind the wait for the pixel words that ntodo needs.
    const uint2 range = im.ranges[tile];
    const uint32_t ovf = g.hdr->overflow;
    const int cq = (int)im.qdone[4 * tile + quad]; // the records the forward took: every contributor is among them
    const int n = ovf ? 0 : (int)(range.y - range.x);
    BinView bn; // the layout of the binning blob follows the capacity the forward ran with (kept in the header)
    binning_layout(binning, (size_t)g.hdr->capacity, &bn);
    const uint2* __restrict__ qh = bn.qhits + (ovf ? (size_t)0 : 4 * (size_t)range.x + (size_t)quad * (size_t)n);
    const size_t pix = (size_t)py * W + px, HW = (size_t)H * W;

#ifdef GSR_EXP_PROLOGUE2 // experiment (round 6): a second, serialised copy of the job's dependent trips to memory (tile range -> quad-hit record -> splat record) in front
                         // of the real ones: what the kernel's time rises by is what a prologue hidden behind the previous job's last round could return at most
    {
        uint32_t t2 = tile;
        asm volatile("" : "+v"(t2));
        const uint2 r2 = im.ranges[t2];
        const uint32_t c2 = im.qdone[4 * t2 + quad];
        const uint2* q2 = bn.qhits + 4 * (size_t)r2.x + (size_t)quad * (size_t)(r2.y - r2.x);
        const uint2 e2 = q2[max((int)c2 - 1 - lane, 0)];
#if GSR_EXP_PROLOGUE2 >= 3
        const float4 z2 = g.g0[e2.y & GSR_ID_MASK];
        asm volatile("" ::"v"(z2.x));
#else
        asm volatile("" ::"v"(e2.x));
#endif
    }
#endif
    uint2 rec_c = qh[max(cq - 1 - lane, 0)];
    uint2 rec_n = qh[max(cq - 1 - (lane + Q), 0)];
    const float T_final = inside ? im.final_T[pix] : 0.f;
    float T = T_final;
    const uint32_t last = inside ? im.n_contrib[pix] : 0u;
    const float g0 = inside ? dL_dpix[pix] : 0.f, g1 = inside ? dL_dpix[HW + pix] : 0.f,
                g2 = inside ? dL_dpix[2 * HW + pix] : 0.f;
    const float bg_dot = bg[0] * g0 + bg[1] * g1 + bg[2] * g2;
    const float nTf_bg = -T_final * bg_dot;
    // colour accumulated behind the current splat (the reference's accum_rec, updated eagerly:
    // last_alpha*last_color + (1-last_alpha)*accum_rec == fma(alpha, c - S, S) one step later). Kept per
    // channel: c - S is formed BEFORE the contraction with the pixel gradient — neighbouring splats have
    // similar colours (depth renders!), and contracting first turns an exact small difference into the
    // difference of two rounded large numbers (measured: 9e-5 instead of 1e-6 on long lists).
    float S0 = 0.f, S1 = 0.f, S2 = 0.f;
    constexpr int NC = 9;                                 // sums per (quad, splat) record
    const int fe = lane / NC, fc = lane - NC * fe;        // flush lane -> (entry, component)
    GP[0][r * 17 + l] = g0; GP[1][r * 17 + l] = g1; GP[2][r * 17 + l] = g2;
    if (lane == 0) {
        E0[Q] = make_float4(-1.0e5f, -1.0e5f, -1.f, 0.f); // power2 ~ -2e10: exp2 gives 0
        E1[Q] = make_float4(-1.f, 0.f, 0.f, 0.f);
        E2[Q] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int ntodo = __builtin_amdgcn_readfirstlane(min(n, (int)wave_max_u32(last)));
    for (int i = lane; i < (int)(sizeof(POOL) / sizeof(float4)); i += 64) POOL[i] = make_float4(0.f, 0.f, 0.f, 0.f); // stale ring slots are read (never used): keep them finite
    const float X0pf = (float)X0p, Y0pf = (float)Y0p;
    v2f* const ud_w0 = UD + l * (4 * WRING + 1) + r * WRING; // where this lane parks (u, dcol) of ring slot 0

    // The forward logged the entries that reach this quad (list position, id), in list order; walk them
    // back to front. Gather pipeline: the records of the next two steps and the geometry of the next step
    // are in flight (unconditional loads from clamped, always valid addresses so that the compiler can
    // count them: the colour gather must not wait for the loads issued after it).
    if (ntodo <= 0 || cq <= 0) return;
    int k0 = 0;
    float4 a_c = g.g0[rec_c.y & GSR_ID_MASK], b_c = g.g1[rec_c.y & GSR_ID_MASK], c_c = g.col[rec_c.y & GSR_ID_MASK];
    // ---- flush: a round leaves the totals of its entries staged in the block (12 words per entry: NC sums, the splat id in
    //      the last one); they are sent seven entries per instruction, NC consecutive lanes per 64-byte accumulator record:
    //      one L2 atomic record per (quad, splat). The flush of round i runs inside round i + 1, right AFTER that round's
    //      gathers were issued and before its blend loop: the wave's next wait on vector memory (for those gathers, a whole
    //      blend loop later; atomics count in vmcnt on gfx9 and a wait behind a loop of them is a wait for all of them) then
    //      finds the atomics long acknowledged. Issued at the end of their own round they were waited for at the top of the
    //      next one — the round trip of a write-through atomic per round, per wave, with nothing else to do.
    constexpr int FPER = 64 / NC;              // entries per flush instruction
    constexpr int FITER = (64 + FPER - 1) / FPER;
    const float* const accf = reinterpret_cast<const float*>(POOL);
    auto flush = [&](const int cnt) {
        float val[FITER];
        uint32_t sid[FITER];
#pragma unroll
        for (int i = 0; i < FITER; i++) { // all LDS reads first: one round trip instead of two per instruction
            const int e = i * FPER + fe; // (e > 63 reads on into the block, never used: one base register + immediate offsets)
            val[i] = accf[e * WACCW + fc];
            sid[i] = __float_as_uint(accf[e * WACCW + WACCW - 1]);
        }
#pragma unroll
        for (int i = 0; i < FITER; i++) {
            const bool ok = lane < FPER * NC && i * FPER + fe < cnt && val[i] != 0.f;
#ifndef GSR_EXP_NOFLUSH
            if (ok) unsafeAtomicAdd(&g.acc[(size_t)sid[i] * GSR_ACC_STRIDE + fc], val[i]);
#else
            if (ok && val[i] == 123.456f) g.acc[0] = val[i];
#endif
        }
        lds_turn(); // the block becomes the ring again
    };
    int pend = 0; // entries of the previous round waiting to be flushed
    while (k0 < cq) {
        // ---- gather + compaction (records past the last contributor of every pixel are dropped): one step, <= 64 entries
        int count = 0;
        {
            const uint32_t id = rec_c.y & GSR_ID_MASK, pos = rec_c.x, pmask = rec_c.y >> GSR_ID_BITS;
            const float4 a = a_c, b = b_c, c = c_c; // the whole 48-byte record of the next step is in flight during this round
            const int k = k0 + lane;
            const bool hit = lane < Q && k < cq && pos < (uint32_t)ntodo;
            rec_c = rec_n;
            a_c = g.g0[rec_c.y & GSR_ID_MASK]; b_c = g.g1[rec_c.y & GSR_ID_MASK]; c_c = g.col[rec_c.y & GSR_ID_MASK];
            rec_n = qh[max(cq - 1 - (k + 2 * Q), 0)];
            // Wait for THIS round's gathers here, on every path (their use below is under `if (hit)`: on the path around it
            // they would count as still in flight — on the first round they are — and the compiler would place the wait
            // behind the flush's atomics, i.e. wait for those as well).
            asm volatile("" ::"v"(a.x), "v"(b.x), "v"(c.x));
            const unsigned long long m = __ballot(hit);
            if (hit) {
                const int e = mbcnt64(m);
                E0[e] = make_float4(a.x, a.y, a.z * (-0.5f * GSR_LOG2E), a.w * -GSR_LOG2E);
                E1[e] = make_float4(b.x * (-0.5f * GSR_LOG2E), b.y, c.x, c.y);
                E2[e] = make_float4(c.z, __uint_as_float(pos), b.z, __uint_as_float(id | (pmask << GSR_ID_BITS)));
            }
            count = (int)__popcll(m);
            k0 += Q;
        }
        if (pend > 0) { flush(pend); pend = 0; }
        if (count == 0) continue;
        lds_turn();
        // ---- per-patch hit lists (lane e looks at parked entry e)
        int c0, c1, c2, c3;
        {
            bool h[4] = {false, false, false, false};
            if (lane < count) { // the forward already ran the patch cull: its verdict travels in the record
                const float4 z = E2[lane];
                const uint32_t pm = __float_as_uint(z.w) >> GSR_ID_BITS;
                h[0] = (pm & 1u) != 0u; h[1] = (pm & 2u) != 0u; h[2] = (pm & 4u) != 0u; h[3] = (pm & 8u) != 0u;
            }
            const unsigned long long m0 = __ballot(h[0]), m1 = __ballot(h[1]), m2 = __ballot(h[2]), m3 = __ballot(h[3]);
            const blist_t off = (blist_t)(lane * LUNIT);
            if (h[0]) LIST[0 * (Q + 4) + mbcnt64(m0)] = off;
            if (h[1]) LIST[1 * (Q + 4) + mbcnt64(m1)] = off;
            if (h[2]) LIST[2 * (Q + 4) + mbcnt64(m2)] = off;
            if (h[3]) LIST[3 * (Q + 4) + mbcnt64(m3)] = off;
            c0 = (int)__popcll(m0); c1 = (int)__popcll(m1); c2 = (int)__popcll(m2); c3 = (int)__popcll(m3);
        }
        // the loop runs an even number of iterations (unrolled by two); shorter lists are padded with the dummy entry
        const int maxc = (max(max(c0, c1), max(c2, c3)) + 1) & ~1;
        {
            const int cr = r == 0 ? c0 : r == 1 ? c1 : r == 2 ? c2 : c3;
            for (int p = cr + l; p < maxc + 4; p += 16) LIST[r * (Q + 4) + p] = (blist_t)(Q * LUNIT);
        }
#ifdef GSR_EXP_ROWFILL // instrumented build (scripts/rowfill.py): how full the padded lists and the 16-slot reduce blocks are
        if (lane == 0) {
            atomicAdd(&g.hdr->pad[0], (uint32_t)(c0 + c1 + c2 + c3)); // list entries that do work
            atomicAdd(&g.hdr->pad[1], (uint32_t)(4 * maxc));          // row-iterations the wave runs for them
            atomicAdd(&g.hdr->pad[2], 1u);                            // rounds
            atomicAdd(&g.hdr->pad[3], (uint32_t)((maxc + 15) / 16));  // reduce phases
            atomicAdd(&g.hdr->pad[4], (uint32_t)count);               // parked entries (quad hits)
        }
#endif
        lds_turn();
        const blist_t* __restrict__ mylist = LIST + r * (Q + 4);
        // per-entry totals of this round, in the registers of lane e
        float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f, t4 = 0.f, t5 = 0.f, t6 = 0.f, t7 = 0.f, t8 = 0.f;
        // ---- blend. An iteration is split in two: what does not depend on the pixel's running state (alpha and the
        //      Gaussian weight of the entry at this pixel) and what does (T, accum_rec, dL/dalpha). The loop handles two
        //      entries per trip and evaluates both first halves before the two second halves: a wave issues in order and
        //      every instruction of a half depends on the one before it (exp2, min, compare, select, rcp ...), so two
        //      independent chains in flight nearly double what one wave gets out of the SIMD — it shares it with only
        //      two others (LDS-limited occupancy).
        auto alpha_part = [&](const float4 A, const float4 B, const float4 Cz, float& alpha, float& G) {
            const float dx = A.x - pxf, dy = A.y - pyf;
            const float power2 = pair_power2(dx, dy, A.z, A.w, B.x); // = power * log2(e)
            const float Graw = __builtin_amdgcn_exp2f(power2);
            const float araw = fminf(0.99f, B.y * Graw);
            const bool valid = lane_of(wm(__float_as_uint(Cz.y) < last) & wm(power2 <= 0.0f) & wm(araw >= GSR_ALPHA_MIN));
            alpha = valid ? araw : 0.f;
            G = valid ? Graw : 0.f;
        };
        auto state_part = [&](v2f* const slot, const float alpha, const float G, const float ia, const float4 B, const float4 Cz) {
            T = T * ia;
            const float e0 = B.z - S0, e1 = B.w - S1, e2 = Cz.x - S2;
            const float eg = fmaf(e2, g2, fmaf(e1, g1, e0 * g0)); // (colour - accum_rec) . dL_dpix
            const float dL_dalpha = fmaf(nTf_bg, ia, eg * T); // - T_final/(1-alpha) * (bg . dL_dpix)
            S0 = fmaf(alpha, e0, S0); S1 = fmaf(alpha, e1, S1); S2 = fmaf(alpha, e2, S2);
            v2f ud;
            ud.x = G * dL_dalpha;
            ud.y = alpha * T;
            *slot = ud;
        };
        // ---- reduce + merge, every 16 iterations. (1) lane (r, l) sums the 16 pixels of the pair (row r, ring slot l) =
        //      list position b0 + l of row r into nine numbers; (2) the sums go to ST, and every row publishes where its
        //      entries sit (INV); (3) lane e collects the sums of entry e from the (at most four) rows that hold it.
        //      No read-modify-write on shared data anywhere: nothing to serialise, nothing to make atomic.
        auto gp_load = [&](const int p) -> float4 {
            // (measured and dropped in round 4: pixel-major float4 entries, one ds_read_b128 per pixel instead of three or four ds_read_b32, with
            // byte-sized lists to stay inside ten LDS blocks: plain kernel 221 -> 238 us (8 spills), fused pair 266 -> 266)
#ifdef GSR_EXP_NOGP // timing experiment (wrong colour sums): the reduce phase without its 48 dL/dpixel reads = 96 of its ~263 LDS cycles
            return make_float4(pxf, pyf, 1.f, 0.f);
#else
            return make_float4(GP[0][r * 17 + p], GP[1][r * 17 + p], GP[2][r * 17 + p], 0.f);
#endif
        };
        auto reduce = [&](const int b0, const int nb) {
            lds_turn();
            const uint32_t o = loff(mylist[min(b0 + l, maxc + 3)]); // padded lists: always a valid entry (slots >= nb: not published)
            const float2 c = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(E0) + o); // splat centre
            // all sixteen ring reads and the first dL/dpixel reads go out before the first use: with ~3 waves per SIMD
            // an LDS round trip per pixel would be the longest thing in this phase
            v2f ud[16];
#pragma unroll
            for (int p = 0; p < 16; p++) ud[p] = UD[p * (4 * WRING + 1) + lane];
            float4 gq[4];
#pragma unroll
            for (int p = 0; p < 4; p++) gq[p] = gp_load(p);
            __builtin_amdgcn_sched_barrier(0);
            float dxk[4], dyk[4];
            const float x0 = X0pf, y0 = Y0pf;
#pragma unroll
            for (int k = 0; k < 4; k++) { dxk[k] = c.x - (x0 + (float)k); dyk[k] = c.y - (y0 + (float)k); }
            // moments of u about the splat centre over the 4x4 patch, through its column and row sums (dx depends on the
            // column i = p & 3 only, dy on the row j = p >> 2 only): 71 instead of 128 instructions
            float q0 = 0.f, q1 = 0.f, q2 = 0.f;
            float col[4], row[4], wj[4];
#pragma unroll
            for (int i = 0; i < 4; i++) col[i] = (ud[i].x + ud[4 + i].x) + (ud[8 + i].x + ud[12 + i].x);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                row[j] = (ud[4 * j].x + ud[4 * j + 1].x) + (ud[4 * j + 2].x + ud[4 * j + 3].x);
                wj[j] = fmaf(dxk[3], ud[4 * j + 3].x, fmaf(dxk[2], ud[4 * j + 2].x, fmaf(dxk[1], ud[4 * j + 1].x, dxk[0] * ud[4 * j].x)));
            }
            const float m0 = (row[0] + row[1]) + (row[2] + row[3]);
            const float m1 = fmaf(dxk[3], col[3], fmaf(dxk[2], col[2], fmaf(dxk[1], col[1], dxk[0] * col[0])));
            const float m2 = fmaf(dyk[3], row[3], fmaf(dyk[2], row[2], fmaf(dyk[1], row[1], dyk[0] * row[0])));
            const float m3 = fmaf(dxk[3] * dxk[3], col[3], fmaf(dxk[2] * dxk[2], col[2], fmaf(dxk[1] * dxk[1], col[1], (dxk[0] * dxk[0]) * col[0])));
            const float m4 = fmaf(dyk[3], wj[3], fmaf(dyk[2], wj[2], fmaf(dyk[1], wj[1], dyk[0] * wj[0])));
            const float m5 = fmaf(dyk[3] * dyk[3], row[3], fmaf(dyk[2] * dyk[2], row[2], fmaf(dyk[1] * dyk[1], row[1], (dyk[0] * dyk[0]) * row[0])));
#pragma unroll
            for (int p = 0; p < 16; p++) {
                const float4 gp = gq[p & 3];
                if (p + 4 < 16) gq[p & 3] = gp_load(p + 4);
                q0 = fmaf(ud[p].y, gp.x, q0); q1 = fmaf(ud[p].y, gp.y, q1); q2 = fmaf(ud[p].y, gp.z, q2);
            }
            lds_turn(); // every lane has read its column of the ring: the block turns into ST
            ST[0 * 64 + lane] = make_float4(m0, m1, m2, m3);
            ST[1 * 64 + lane] = make_float4(m4, m5, q0, q1);
            ST[2 * 64 + lane] = make_float4(q2, 0.f, 0.f, 0.f);
            INV[lane] = GSR_INV_NONE * 0x01010101u;
            if (lane < 4) INV[64 + lane] = GSR_INV_NONE * 0x01010101u;
            if (l < nb) reinterpret_cast<uint8_t*>(INV)[(o >> 2) + r] = (uint8_t)l; // o >> 2 = entry * 4; the dummy (index Q) lands in the slack
            lds_turn();
            const uint32_t inv = INV[lane]; // lane e: where entry e sits in the four rows
            float4 s0[4], s1[4], s2[4];
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
                const uint32_t q = rr * WRING + ((inv >> (8 * rr)) & 15u);
                s0[rr] = ST[0 * 64 + q]; s1[rr] = ST[1 * 64 + q]; s2[rr] = ST[2 * 64 + q];
            }
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
                const float f = ((inv >> (8 * rr)) & GSR_INV_NONE) == 0u ? 1.f : 0.f;
                t0 = fmaf(f, s0[rr].x, t0); t1 = fmaf(f, s0[rr].y, t1); t2 = fmaf(f, s0[rr].z, t2); t3 = fmaf(f, s0[rr].w, t3);
                t4 = fmaf(f, s1[rr].x, t4); t5 = fmaf(f, s1[rr].y, t5);
                t6 = fmaf(f, s1[rr].z, t6); t7 = fmaf(f, s1[rr].w, t7); t8 = fmaf(f, s2[rr].x, t8);
            }
            lds_turn(); // the block is the ring again
        };
        // software pipeline over PAIRS of entries, unrolled by two pairs so that the register sets alternate without copies:
        // while one pair is blended the next pair's entries and the list offsets of the pair after are in flight
#define GSR_LOAD3(A, B, C, off) \
        A = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(E0) + (off)); \
        B = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(E1) + (off)); \
        C = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(E2) + (off))
        uint32_t o0 = loff(mylist[0]), o1 = loff(mylist[1]), o2 = loff(mylist[2]), o3 = loff(mylist[3]);
        float4 A0, B0, C0, A1, B1, C1, A2, B2, C2, A3, B3, C3;
        GSR_LOAD3(A0, B0, C0, o0); GSR_LOAD3(A1, B1, C1, o1);
        int ring = 0;
        auto pair = [&](const int it, const float4 Aa, const float4 Ba, const float4 Ca, const float4 Ab, const float4 Bb, const float4 Cb) {
            float al0, G0, al1, G1;
            alpha_part(Aa, Ba, Ca, al0, G0);
            alpha_part(Ab, Bb, Cb, al1, G1);
            const float ia0 = __builtin_amdgcn_rcpf(1.f - al0), ia1 = __builtin_amdgcn_rcpf(1.f - al1);
            state_part(ud_w0 + ring, al0, G0, ia0, Ba, Ca);
            state_part(ud_w0 + ring + 1, al1, G1, ia1, Bb, Cb);
            ring += 2;
            if (ring == WRING || it + 2 >= maxc) {
#ifndef GSR_EXP_NOREDUCE
                reduce(it + 2 - ring, ring);
#endif
                ring = 0;
            }
        };
#ifdef GSR_EXP_NOLOOP
        if (maxc == 12345)
#endif
        for (int it = 0; it < maxc; it += 4) {
            GSR_LOAD3(A2, B2, C2, o2); GSR_LOAD3(A3, B3, C3, o3);
            o0 = loff(mylist[it + 4]); o1 = loff(mylist[it + 5]); // the lists are padded up to maxc + 3
            __builtin_amdgcn_sched_barrier(0); // keep the prefetch above the pair it overlaps with
            pair(it, A0, B0, C0, A1, B1, C1);
            if (it + 2 >= maxc) break;
            GSR_LOAD3(A0, B0, C0, o0); GSR_LOAD3(A1, B1, C1, o1);
            o2 = loff(mylist[it + 6]); o3 = loff(mylist[it + 7]);
            __builtin_amdgcn_sched_barrier(0);
            pair(it + 2, A2, B2, C2, A3, B3, C3);
        }
#undef GSR_LOAD3
        // ---- lane e holds the totals of entry e: stage them for the flush (which runs inside the next round, see above)
        ST[lane * 3 + 0] = make_float4(t0, t1, t2, t3);
        ST[lane * 3 + 1] = make_float4(t4, t5, t6, t7);
        ST[lane * 3 + 2] = make_float4(t8, 0.f, 0.f, __uint_as_float(__float_as_uint(E2[lane].w) & GSR_ID_MASK)); // (slots >= count: never flushed)
        pend = count;
        lds_turn();
    }
    if (pend > 0) flush(pend);
}

// ---- the LEAN body (fused pair): the same rounds with a smaller footprint — the list is walked in blocks of GSR_RING iterations, the
//      reduce phase takes the patch one pixel row at a time and collects two rows of sums at a time: <= 146 registers without spills
//      where the wide body needs 168 and spills 6-9 with the pair's extra channels; without the colour sums (tracking) 118 registers and
//      7.5 KB of LDS: four waves per SIMD. Built for four waves per SIMD with a ring of 12 and 54 records per round
//      (-DGSR_BWD_WAVES=4 -DGSR_RING=12 -DGSR_BSTEP=54 -DGSR_BWD_LEAN_ALWAYS=1) the plain render fits 127 registers and 10 144 bytes, 16
//      waves per CU, and takes 211-215 us (ring 12, 48 records: 222) against the wide body's 209-211 at 12: DESIGN.md section 4.
// DUAL: the fused colour + depth / silhouette render (gsr_forward_args.out_ds): two more channels ride on the same
// alphas — the splat's view depth z and the constant 1 (what the reference renders in a second pass with colours
// [z, 1, 0], src/Render.cc:949-981). dL_dds [2,H,W] is their upstream gradient; the z channel adds a tenth sum per
// (quad, splat): dL/dz-colour, which K_splat_bwd folds into the mean.
// COLORS = false: nobody consumes the colour sums (a tracking iteration: the pose is the only parameter, the colours and the depth channel's
// colour are constants) — the reduce phase then skips its dL/dpixel reads and three or four of its nine or ten sums, the records are six floats.
// SIL = false (with DUAL): dL_dds holds the depth plane only, the silhouette's upstream gradient is zero (both loops use the silhouette as a
// detached mask); with SIL the plane is read once per pixel and folded into the background factor of dL/dalpha (below): no work in the loop.
template <int Q, bool DUAL, bool COLORS, bool SIL>
__device__ __forceinline__ void blend_bwd_lean(ImageView im, char* __restrict__ binning, GeomView g, const float* __restrict__ bg, int W, int H,
                                               int grid_x, int ntiles, int tile0, const float* __restrict__ dL_dpix, const float* __restrict__ dL_dds)
{
    static_assert(Q == 64 && GSR_RING <= 16 && GSR_RING % 2 == 0 && GSR_BSTEP <= 64, "one lane per parked entry and one lane per (row, ring slot) pair");
    constexpr int QB = GSR_BSTEP; // parked entries per round
    // parked entries; slot Q is a dummy (opacity 0, far away) the per-patch lists are padded with: no "row still active"
    // compare and no index select in the blend loop
    __shared__ float4 E0[QB + 1], E1[QB + 1], E2[QB + 1]; // (px, py, a2, b2) (c2, opacity, red, green) (blue, list position, view depth, splat id | patch mask << 28)
    // per-patch hit lists as BYTE OFFSETS (entry * 16 into E0/E1/E2): shifts and integer mads are half-rate VALU work
    // LDS budget: gfx950 hands LDS out in blocks of 1280 bytes (scripts/lds_granule.hip: 12 800 bytes per workgroup -> 12 single-wave
    // workgroups per CU, 12 816 -> 11), so the kernel is held at exactly ten blocks — the 12 waves per CU its registers allow.
    // The fused pair pays for its fourth dL/dpixel channel with byte-sized list entries (one shift per entry read).
    constexpr bool BYTE_LISTS = DUAL;
    using blist_t = typename std::conditional<BYTE_LISTS, uint8_t, uint16_t>::type;
    constexpr uint32_t LUNIT = BYTE_LISTS ? 1u : 16u;
    auto loff = [](const blist_t x) -> uint32_t { return BYTE_LISTS ? (uint32_t)x << 4 : (uint32_t)x; };
    __shared__ blist_t LIST[4 * (QB + 4)];
    // One block of LDS used three ways, one after the other:
    //  UD  the ring: (u, dcol) of pixel p of pair q = row * 16 + (iteration % 16) at float2 UD[p * 65 + q]. A blend iteration
    //      writes 16 consecutive p for 4 values of q (stride 65 float2: the 16 lanes of a row fall on 16 different bank
    //      pairs), the reduce phase reads 64 consecutive q for one p: both conflict-free;
    //  ST  float4 ST[3 * 64]: the nine sums of pair q at ST[k * 64 + q], k = 0..2 (reduce phase -> merge by entry);
    //  ACC float ACC[64 * 12]: the per-entry totals of the round, staged for the coalesced flush.
    // (without the colour sums the ring holds u alone, the records are two float4)
    using ring_t = typename std::conditional<COLORS, v2f, float>::type;
    constexpr int STN = COLORS ? 3 : 2;   // float4 per (row, slot) pair in ST and per entry in the staged totals
    constexpr int RING_BYTES = 16 * (4 * GSR_RING + 1) * (int)sizeof(ring_t), ST_BYTES = STN * 64 * 16 + 68 * 4;
    __shared__ float4 POOL[((RING_BYTES > ST_BYTES ? RING_BYTES : ST_BYTES) + 15) / 16];
    // dL/dpixel of pixel p of patch r for the reduce phases, packed for 8-byte LDS reads (ds_read_b64 moves 256 B per LDS cycle, ds_read_b32
    // and ds_read2_b32 half that): (g0, g1) per pixel, and g2 of two neighbouring pixels together (fused pair: (g2, g3) per pixel)
    __shared__ float2 G01[64];
    __shared__ float2 G2X[DUAL ? 64 : 32];
#ifdef GSR_EXP_LDSPAD // occupancy experiment: more LDS per wave, fewer waves per SIMD
    __shared__ uint32_t PADX[GSR_EXP_LDSPAD];
    if (W == -1) PADX[threadIdx.x] = 1u;
#endif
    ring_t* const UD = reinterpret_cast<ring_t*>(POOL);
    float4* const ST = POOL;
    // INV, per reduce phase: byte r of word e = ring slot of entry e in row r, or GSR_INV_NONE. It lives behind ST inside the
    // block (it is only alive while the block is ST); its words are finite as floats, so the ring slots it leaves behind are
    // harmless when they are read as stale slots
    uint32_t* const INV = reinterpret_cast<uint32_t*>(POOL + STN * 64);
#ifdef GSR_EXP_TIMELINE
    TimelineMark mark(blockIdx.x);
#endif
    const uint32_t w = xcd_remap(blockIdx.x, 4u * (uint32_t)ntiles, 4u * GSR_XCD_TILES);
    const uint32_t tile = (uint32_t)tile0 + (w >> 2), quad = w & 3u;
    const int tx = tile % grid_x, ty = tile / grid_x;
    const int lane = threadIdx.x, r = lane >> 4, l = lane & 15;
    const int X0 = tx * 16 + (int)(quad & 1u) * 8, Y0 = ty * 16 + (int)(quad >> 1) * 8;
    const int X0p = X0 + (r & 1) * 4, Y0p = Y0 + (r >> 1) * 4; // origin of this row's patch
    const int px = X0p + (l & 3), py = Y0p + (l >> 2);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    // The job's first trips to memory (round 4: four dependent ones instead of six — at ~2.4 waves per SIMD nothing hides them): the tile's
    // range, the header words and the quad's record count are asked for together, the pixel's own words with them, and the first records
    // as soon as those scalars are there (clamped, always valid addresses) — not behind the wait for the pixel words that ntodo needs.
    const uint2 range = im.ranges[tile];
    const uint32_t ovf = g.hdr->overflow;
    const int cq = (int)im.qdone[4 * tile + quad]; // the records the forward took: every contributor is among them
    const int n = ovf ? 0 : (int)(range.y - range.x);
    BinView bn; // the layout of the binning blob follows the capacity the forward ran with (kept in the header)
    binning_layout(binning, (size_t)g.hdr->capacity, &bn);
    const uint2* __restrict__ qh = bn.qhits + (ovf ? (size_t)0 : 4 * (size_t)range.x + (size_t)quad * (size_t)n);
    const size_t pix = (size_t)py * W + px, HW = (size_t)H * W;

    uint2 rec_c = qh[max(cq - 1 - lane, 0)];
    uint2 rec_n = qh[max(cq - 1 - (lane + GSR_BSTEP), 0)];
    const float T_final = inside ? im.final_T[pix] : 0.f;
    float T = T_final;
    const uint32_t last = inside ? im.n_contrib[pix] : 0u;
    const float g0 = inside ? dL_dpix[pix] : 0.f, g1 = inside ? dL_dpix[HW + pix] : 0.f,
                g2 = inside ? dL_dpix[2 * HW + pix] : 0.f;
    // (their background is 0; without DUAL but with SIL, dL_dds IS the silhouette's plane: a sharded tracking iteration on the surface depth, round 6)
    const float g3 = DUAL && inside ? dL_dds[pix] : 0.f, g4 = SIL && inside ? dL_dds[DUAL ? HW + pix : pix] : 0.f;
    const float bg_dot = bg[0] * g0 + bg[1] * g1 + bg[2] * g2;
    // The silhouette channel (colour 1 on every splat) needs no recursion of its own: what it accumulates is 1 - T_final, so its share of
    // dL/dalpha_i, g4 T_i (1 - accum_rec_i) = g4 T_final / (1 - alpha_i), has the background term's form with colour -1 and rides in its factor.
    const float nTf_bg = -T_final * (bg_dot - g4);
    // colour accumulated behind the current splat (the reference's accum_rec, updated eagerly:
    // last_alpha*last_color + (1-last_alpha)*accum_rec == fma(alpha, c - S, S) one step later). Kept per
    // channel: c - S is formed BEFORE the contraction with the pixel gradient — neighbouring splats have
    // similar colours (depth renders!), and contracting first turns an exact small difference into the
    // difference of two rounded large numbers (measured: 9e-5 instead of 1e-6 on long lists).
    float S0 = 0.f, S1 = 0.f, S2 = 0.f, S3 = 0.f;
    constexpr int NC = !COLORS ? 6 : DUAL ? 10 : 9;       // sums per (quad, splat) record
    if (COLORS) {
        G01[lane] = make_float2(g0, g1);
        if (DUAL) G2X[DUAL ? lane : 0] = make_float2(g2, g3);
        else reinterpret_cast<float*>(G2X)[lane] = g2;
    }
    if (lane == 0) {
        E0[QB] = make_float4(-1.0e5f, -1.0e5f, -1.f, 0.f); // power2 ~ -2e10: exp2 gives 0
        E1[QB] = make_float4(-1.f, 0.f, 0.f, 0.f);
        E2[QB] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int ntodo = __builtin_amdgcn_readfirstlane(min(n, (int)wave_max_u32(last)));
    for (int i = lane; i < (int)(sizeof(POOL) / sizeof(float4)); i += 64) POOL[i] = make_float4(0.f, 0.f, 0.f, 0.f); // stale ring slots are read (never used): keep them finite
    const float X0pf = (float)X0p, Y0pf = (float)Y0p;
    ring_t* const ud_w0 = UD + l * (4 * GSR_RING + 1) + r * GSR_RING; // where this lane parks (u, dcol) of ring slot 0
    const int rslot = r * GSR_RING + min(l, GSR_RING - 1);         // the (row, ring slot) pair this lane reduces (lanes l >= GSR_RING idle: they re-read the last slot)

    // The forward logged the entries that reach this quad (list position, id), in list order; walk them
    // back to front. Gather pipeline: the records of the next two steps and the geometry of the next step
    // are in flight (unconditional loads from clamped, always valid addresses so that the compiler can
    // count them: the colour gather must not wait for the loads issued after it).
    if (ntodo <= 0 || cq <= 0) return;
    int k0 = 0;
    float4 a_c = g.g0[rec_c.y & GSR_ID_MASK], b_c = g.g1[rec_c.y & GSR_ID_MASK], c_c = g.col[rec_c.y & GSR_ID_MASK];
    // ---- flush: a round leaves the totals of its entries staged in the block (12 words per entry: NC sums, the splat id in
    //      the last one); they are sent seven entries per instruction, NC consecutive lanes per 64-byte accumulator record:
    //      one L2 atomic record per (quad, splat). The flush of round i runs inside round i + 1, right AFTER that round's
    //      gathers were issued and before its blend loop: the wave's next wait on vector memory (for those gathers, a whole
    //      blend loop later; atomics count in vmcnt on gfx9 and a wait behind a loop of them is a wait for all of them) then
    //      finds the atomics long acknowledged. Issued at the end of their own round they were waited for at the top of the
    //      next one — the round trip of a write-through atomic per round, per wave, with nothing else to do.
    constexpr int ACCW = 4 * STN;              // floats per staged record (NC sums, the splat id in the last one): whole float4s
    constexpr int FPER = 64 / NC;              // entries per flush instruction
    constexpr int FITER = (QB + FPER - 1) / FPER;
    const float* const accf = reinterpret_cast<const float*>(POOL);
    auto flush = [&](const int cnt) {
        // flush lane -> (entry, component), formed here: kept in registers (with the record address that follows from it) it would be
        // three of the 128 through the whole kernel
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int fe = ln / NC, fc = ln - NC * fe;
        float val[FITER];
        uint32_t sid[FITER];
#pragma unroll
        for (int i = 0; i < FITER; i++) { // all LDS reads first: one round trip instead of two per instruction
            const int e = i * FPER + fe; // (e > 63 reads on into the block, never used: one base register + immediate offsets)
            val[i] = accf[e * ACCW + fc];
            sid[i] = __float_as_uint(accf[e * ACCW + ACCW - 1]);
        }
#pragma unroll
        for (int i = 0; i < FITER; i++) {
            const bool ok = lane < FPER * NC && fe < cnt - i * FPER && val[i] != 0.f; // (a scalar subtraction per instruction, not FITER lane constants)
#ifndef GSR_EXP_NOFLUSH
            if (ok) unsafeAtomicAdd(&g.acc[(size_t)sid[i] * GSR_ACC_STRIDE + fc], val[i]);
#else
            if (ok && val[i] == 123.456f) g.acc[0] = val[i];
#endif
        }
        lds_turn(); // the block becomes the ring again
    };
    int pend = 0; // entries of the previous round waiting to be flushed
    while (k0 < cq) {
        // ---- gather + compaction (records past the last contributor of every pixel are dropped): one step, <= 64 entries
        int count = 0;
        {
            const uint32_t id = rec_c.y & GSR_ID_MASK, pos = rec_c.x, pmask = rec_c.y >> GSR_ID_BITS;
            const float4 a = a_c, b = b_c, c = c_c; // the whole 48-byte record of the next step is in flight during this round
            const int k = k0 + lane;
            const bool hit = lane < GSR_BSTEP && k < cq && pos < (uint32_t)ntodo;
            rec_c = rec_n;
            a_c = g.g0[rec_c.y & GSR_ID_MASK]; b_c = g.g1[rec_c.y & GSR_ID_MASK]; c_c = g.col[rec_c.y & GSR_ID_MASK];
            rec_n = qh[max(cq - 1 - (k + 2 * GSR_BSTEP), 0)];
            // Wait for THIS round's gathers here, on every path (their use below is under `if (hit)`: on the path around it
            // they would count as still in flight — on the first round they are — and the compiler would place the wait
            // behind the flush's atomics, i.e. wait for those as well).
            asm volatile("" ::"v"(a.x), "v"(b.x), "v"(c.x));
            const unsigned long long m = __ballot(hit);
            if (hit) {
                const int e = mbcnt64(m);
                E0[e] = make_float4(a.x, a.y, a.z * (-0.5f * GSR_LOG2E), a.w * -GSR_LOG2E);
                E1[e] = make_float4(b.x * (-0.5f * GSR_LOG2E), b.y, c.x, c.y);
                E2[e] = make_float4(c.z, __uint_as_float(pos), b.z, __uint_as_float(id | (pmask << GSR_ID_BITS)));
            }
            count = (int)__popcll(m);
            k0 += GSR_BSTEP;
        }
        if (pend > 0) { flush(pend); pend = 0; }
        if (count == 0) continue;
        lds_turn();
        // ---- per-patch hit lists (lane e looks at parked entry e)
        int c0, c1, c2, c3;
        {
            bool h[4] = {false, false, false, false};
            if (lane < count) { // the forward already ran the patch cull: its verdict travels in the record
                const float4 z = E2[lane];
                const uint32_t pm = __float_as_uint(z.w) >> GSR_ID_BITS;
                h[0] = (pm & 1u) != 0u; h[1] = (pm & 2u) != 0u; h[2] = (pm & 4u) != 0u; h[3] = (pm & 8u) != 0u;
            }
            const unsigned long long m0 = __ballot(h[0]), m1 = __ballot(h[1]), m2 = __ballot(h[2]), m3 = __ballot(h[3]);
            const blist_t off = (blist_t)(lane * LUNIT);
            if (h[0]) LIST[0 * (QB + 4) + mbcnt64(m0)] = off;
            if (h[1]) LIST[1 * (QB + 4) + mbcnt64(m1)] = off;
            if (h[2]) LIST[2 * (QB + 4) + mbcnt64(m2)] = off;
            if (h[3]) LIST[3 * (QB + 4) + mbcnt64(m3)] = off;
            c0 = (int)__popcll(m0); c1 = (int)__popcll(m1); c2 = (int)__popcll(m2); c3 = (int)__popcll(m3);
        }
        // the loop runs an even number of iterations (unrolled by two); shorter lists are padded with the dummy entry
        const int maxc = (max(max(c0, c1), max(c2, c3)) + 1) & ~1;
        {
            const int cr = r == 0 ? c0 : r == 1 ? c1 : r == 2 ? c2 : c3;
            for (int p = cr + l; p < maxc + 4; p += 16) LIST[r * (QB + 4) + p] = (blist_t)(QB * LUNIT);
        }
#ifdef GSR_EXP_ROWFILL // instrumented build (scripts/rowfill.py): how full the padded lists and the 16-slot reduce blocks are
        if (lane == 0) {
            atomicAdd(&g.hdr->pad[0], (uint32_t)(c0 + c1 + c2 + c3)); // list entries that do work
            atomicAdd(&g.hdr->pad[1], (uint32_t)(4 * maxc));          // row-iterations the wave runs for them
            atomicAdd(&g.hdr->pad[2], 1u);                            // rounds
            atomicAdd(&g.hdr->pad[3], (uint32_t)((maxc + 15) / 16));  // reduce phases
            atomicAdd(&g.hdr->pad[4], (uint32_t)count);               // parked entries (quad hits)
        }
#endif
        lds_turn();
        const blist_t* __restrict__ mylist = LIST + r * (QB + 4);
        // per-entry totals of this round, in the registers of lane e
        float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f, t4 = 0.f, t5 = 0.f, t6 = 0.f, t7 = 0.f, t8 = 0.f, t9 = 0.f;
        // ---- blend. An iteration is split in two: what does not depend on the pixel's running state (alpha and the
        //      Gaussian weight of the entry at this pixel) and what does (T, accum_rec, dL/dalpha). The loop handles two
        //      entries per trip and evaluates both first halves before the two second halves: a wave issues in order and
        //      every instruction of a half depends on the one before it (exp2, min, compare, select, rcp ...), so two
        //      independent chains in flight nearly double what one wave gets out of the SIMD — it shares it with only
        //      two others (LDS-limited occupancy).
        auto alpha_part = [&](const float4 A, const float4 B, const float4 Cz, float& alpha, float& G) {
            const float dx = A.x - pxf, dy = A.y - pyf;
            const float power2 = pair_power2(dx, dy, A.z, A.w, B.x); // = power * log2(e)
            const float Graw = __builtin_amdgcn_exp2f(power2);
            const float araw = fminf(0.99f, B.y * Graw);
            const bool valid = lane_of(wm(__float_as_uint(Cz.y) < last) & wm(power2 <= 0.0f) & wm(araw >= GSR_ALPHA_MIN));
            alpha = valid ? araw : 0.f;
            G = valid ? Graw : 0.f;
        };
        auto state_part = [&](ring_t* const slot, const float alpha, const float G, const float ia, const float4 B, const float4 Cz) {
            T = T * ia;
            const float e0 = B.z - S0, e1 = B.w - S1, e2 = Cz.x - S2;
            float eg = fmaf(e2, g2, fmaf(e1, g1, e0 * g0)); // (colour - accum_rec) . dL_dpix
            if (DUAL) { // the depth channel: colour z (the silhouette's term sits in nTf_bg)
                const float e3 = Cz.z - S3;
                eg = fmaf(e3, g3, eg);
                S3 = fmaf(alpha, e3, S3);
            }
            const float dL_dalpha = fmaf(nTf_bg, ia, eg * T); // - T_final/(1-alpha) * (bg . dL_dpix)
            S0 = fmaf(alpha, e0, S0); S1 = fmaf(alpha, e1, S1); S2 = fmaf(alpha, e2, S2);
            if constexpr (COLORS) {
                v2f ud;
                ud.x = G * dL_dalpha;
                ud.y = alpha * T;
                *slot = ud;
            } else
                *slot = G * dL_dalpha;
        };
        // ---- reduce + merge, every 16 iterations. (1) lane (r, l) sums the 16 pixels of the pair (row r, ring slot l) =
        //      list position b0 + l of row r into nine numbers; (2) the sums go to ST, and every row publishes where its
        //      entries sit (INV); (3) lane e collects the sums of entry e from the (at most four) rows that hold it.
        //      No read-modify-write on shared data anywhere: nothing to serialise, nothing to make atomic.
        auto reduce = [&](const int b0, const int nb, auto&& mid) {
            lds_turn();
            const uint32_t o = loff(mylist[min(b0 + l, maxc + 3)]); // padded lists: always a valid entry (slots >= nb: not published)
            const float2 c = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(E0) + o); // splat centre
            // The 16 pixels of the pair are taken one pixel ROW (four pixels) at a time, the next row's ring entries and dL/dpixel in flight
            // while this one is summed: eight instead of 32 ring registers live at once (the kernel is compiled for four waves per SIMD:
            // 128 registers), every LDS read 8 bytes wide.
            // moments of u about the splat centre over the 4x4 patch through its column and row sums (dx depends on the column
            // i = p & 3 only, dy on the row j = p >> 2 only)
            // WIDE (three waves per SIMD: registers to spare): all sixteen ring reads go out before the first use and dL/dpixel runs one pixel row
            // ahead; otherwise one pixel row of each is in flight while the previous one is summed.
            constexpr bool WIDE = GSR_BWD_WIDE;
            constexpr int AHEAD = WIDE ? 3 : 1;
            float uu[4][4], dd[4][4];
            float2 ga[4][4], gb[4][DUAL ? 4 : 2];
            auto issue = [&](const int j) {
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const ring_t x = UD[(4 * j + i) * (4 * GSR_RING + 1) + rslot];
                    if constexpr (COLORS) { uu[j][i] = x.x; dd[j][i] = x.y; }
                    else uu[j][i] = x;
                }
            };
            auto issue_g = [&](const int j) {
                if (COLORS) {
#pragma unroll
                    for (int i = 0; i < 4; i++) ga[j][i] = G01[r * 16 + 4 * j + i];
                    if (DUAL) {
#pragma unroll
                        for (int i = 0; i < 4; i++) gb[j][DUAL ? i : 0] = G2X[DUAL ? r * 16 + 4 * j + i : 0];
                    } else {
                        gb[j][0] = G2X[r * 8 + 2 * j]; gb[j][1] = G2X[r * 8 + 2 * j + 1];
                    }
                }
            };
#pragma unroll
            for (int j = 0; j < AHEAD; j++) issue(j);
            if (WIDE) issue_g(0);
            float x0 = X0pf, y0 = Y0pf;
            if (DUAL) asm volatile("" : "+v"(x0), "+v"(y0));
            float dxk[4];
#pragma unroll
            for (int k = 0; k < 4; k++) dxk[k] = c.x - (x0 + (float)k);
            float q0 = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f;
            float col[4] = {0.f, 0.f, 0.f, 0.f};
            float m0 = 0.f, m2 = 0.f, m4 = 0.f, m5 = 0.f;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (j + AHEAD < 4) issue(j + AHEAD);
                if (WIDE) { if (j + 1 < 4) issue_g(j + 1); }
                else issue_g(j); // asked for here, used after the moments' arithmetic below
                __builtin_amdgcn_sched_barrier(0);
                const float dy = c.y - (y0 + (float)j);
                const float rw = (uu[j][0] + uu[j][1]) + (uu[j][2] + uu[j][3]);
                const float wj = fmaf(dxk[3], uu[j][3], fmaf(dxk[2], uu[j][2], fmaf(dxk[1], uu[j][1], dxk[0] * uu[j][0])));
#pragma unroll
                for (int i = 0; i < 4; i++) col[i] += uu[j][i];
                m0 += rw;
                m2 = fmaf(dy, rw, m2);
                m4 = fmaf(dy, wj, m4);
                m5 = fmaf(dy * dy, rw, m5);
                if (COLORS) {
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        q0 = fmaf(dd[j][i], ga[j][i].x, q0); q1 = fmaf(dd[j][i], ga[j][i].y, q1);
                        if (DUAL) { q2 = fmaf(dd[j][i], gb[j][DUAL ? i : 0].x, q2); q3 = fmaf(dd[j][i], gb[j][DUAL ? i : 0].y, q3); }
                        else q2 = fmaf(dd[j][i], (i & 1) ? gb[j][i >> 1].y : gb[j][i >> 1].x, q2);
                    }
                }
            }
            const float m1 = fmaf(dxk[3], col[3], fmaf(dxk[2], col[2], fmaf(dxk[1], col[1], dxk[0] * col[0])));
            const float m3 = fmaf(dxk[3] * dxk[3], col[3], fmaf(dxk[2] * dxk[2], col[2], fmaf(dxk[1] * dxk[1], col[1], (dxk[0] * dxk[0]) * col[0])));
            lds_turn(); // every lane has read its column of the ring: the block turns into ST
            ST[0 * 64 + lane] = make_float4(m0, m1, m2, m3);
            ST[1 * 64 + lane] = make_float4(m4, m5, q0, q1);
            if (COLORS) ST[2 * 64 + lane] = make_float4(q2, q3, 0.f, 0.f);
            INV[lane] = GSR_INV_NONE * 0x01010101u;
            if (lane < 4) INV[64 + lane] = GSR_INV_NONE * 0x01010101u;
            if (l < nb) reinterpret_cast<uint8_t*>(INV)[(o >> 2) + r] = (uint8_t)l; // o >> 2 = entry * 4; the dummy (index QB) lands in the slack
            lds_turn();
            mid(); // (the moments' registers are dead here: the next block's first entries are asked for now, not across the whole phase)
            const uint32_t inv = INV[lane]; // lane e: where entry e sits in the four rows
            // the sums of the (at most four) rows that hold entry e, two rows in flight at a time
            float4 s0[2], s1[2], s2[2];
            auto fetch = [&](const int rr, const int s) {
                const uint32_t q = rr * 16 + ((inv >> (8 * rr)) & 15u);
                s0[s] = ST[0 * 64 + q]; s1[s] = ST[1 * 64 + q];
                if (COLORS) s2[s] = ST[2 * 64 + q];
            };
            fetch(0, 0);
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
                const int s = rr & 1;
                if (rr < 3) fetch(rr + 1, s ^ 1);
                __builtin_amdgcn_sched_barrier(0);
                const float f = ((inv >> (8 * rr)) & GSR_INV_NONE) == 0u ? 1.f : 0.f;
                t0 = fmaf(f, s0[s].x, t0); t1 = fmaf(f, s0[s].y, t1); t2 = fmaf(f, s0[s].z, t2); t3 = fmaf(f, s0[s].w, t3);
                t4 = fmaf(f, s1[s].x, t4); t5 = fmaf(f, s1[s].y, t5);
                if (COLORS) {
                    t6 = fmaf(f, s1[s].z, t6); t7 = fmaf(f, s1[s].w, t7); t8 = fmaf(f, s2[s].x, t8);
                    if (DUAL) t9 = fmaf(f, s2[s].y, t9);
                }
            }
            lds_turn(); // the block is the ring again
        };
        // The list is walked in BLOCKS of GSR_RING iterations, a reduce phase after each. Inside a block: a software pipeline over PAIRS of
        // entries through two register sets (A, B) that alternate without copies — while one pair is blended the next pair's entries and the
        // list offsets of the pair after are in flight. The last pair of a block asks for nothing: the next block's first pair is asked for
        // from inside the reduce phase, once its moment registers are dead (no 24 entry registers live across the phase's widest part: the
        // kernel is held at 128 registers, four waves per SIMD).
#define GSR_LOAD3(A, B, C, off) \
        A = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(E0) + (off)); \
        B = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(E1) + (off)); \
        C = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(E2) + (off))
        uint32_t oa0 = loff(mylist[0]), oa1 = loff(mylist[1]), ob0, ob1;
        float4 A0, B0, C0, A1, B1, C1;
        GSR_LOAD3(A0, B0, C0, oa0); GSR_LOAD3(A1, B1, C1, oa1);
        auto pair = [&](ring_t* const slot, const float4 Aa, const float4 Ba, const float4 Ca, const float4 Ab, const float4 Bb, const float4 Cb) {
            float al0, G0, al1, G1;
            alpha_part(Aa, Ba, Ca, al0, G0);
            alpha_part(Ab, Bb, Cb, al1, G1);
            const float ia0 = __builtin_amdgcn_rcpf(1.f - al0), ia1 = __builtin_amdgcn_rcpf(1.f - al1);
            state_part(slot, al0, G0, ia0, Ba, Ca);
            state_part(slot + 1, al1, G1, ia1, Bb, Cb);
        };
#ifdef GSR_EXP_NOLOOP
        if (maxc == 12345)
#endif
        for (int b0 = 0; b0 < maxc; b0 += GSR_RING) {
            const int nb = min(GSR_RING, maxc - b0); // even: maxc is
            const blist_t* const bl = mylist + b0;
            ob0 = loff(bl[2]); ob1 = loff(bl[3]); // (the lists are padded up to maxc + 3)
            for (int i = 0; i < nb; i += 4) {
                // (set B lives inside a trip: assigned and used under the same condition, it must not look loop-carried to the compiler —
                // it would stay allocated across the whole reduce phase)
                float4 A2, B2, C2, A3, B3, C3;
                if (i + 2 < nb) { GSR_LOAD3(A2, B2, C2, ob0); GSR_LOAD3(A3, B3, C3, ob1); }
                oa0 = loff(bl[i + 4]); oa1 = loff(bl[i + 5]);
                __builtin_amdgcn_sched_barrier(0); // keep the prefetch above the pair it overlaps with
                pair(ud_w0 + i, A0, B0, C0, A1, B1, C1);
                if (i + 2 >= nb) break;
                if (i + 4 < nb) { GSR_LOAD3(A0, B0, C0, oa0); GSR_LOAD3(A1, B1, C1, oa1); }
                ob0 = loff(bl[i + 6]); ob1 = loff(bl[i + 7]);
                __builtin_amdgcn_sched_barrier(0);
                pair(ud_w0 + i + 2, A2, B2, C2, A3, B3, C3);
            }
            oa0 = loff(bl[nb]); oa1 = loff(bl[nb + 1]); // first pair of the next block (padding behind the last one)
#ifndef GSR_EXP_NOREDUCE
            reduce(b0, nb, [&]() { GSR_LOAD3(A0, B0, C0, oa0); GSR_LOAD3(A1, B1, C1, oa1); });
#else
            GSR_LOAD3(A0, B0, C0, oa0); GSR_LOAD3(A1, B1, C1, oa1);
#endif
        }
#undef GSR_LOAD3
        // ---- lane e holds the totals of entry e: stage them for the flush (which runs inside the next round, see above)
        const float sidf = __uint_as_float(__float_as_uint(E2[min(lane, QB)].w) & GSR_ID_MASK); // (slots >= count: never flushed)
        ST[lane * STN + 0] = make_float4(t0, t1, t2, t3);
        if (COLORS) {
            ST[lane * STN + 1] = make_float4(t4, t5, t6, t7);
            ST[lane * STN + (COLORS ? 2 : 0)] = make_float4(t8, t9, 0.f, sidf);
        } else
            ST[lane * STN + 1] = make_float4(t4, t5, 0.f, sidf);
        pend = count;
        lds_turn();
    }
    if (pend > 0) flush(pend);
}

// The kernel: the plain render runs the wide body (three waves per SIMD at 168 registers), the fused pair the lean one. Both were measured in both
// roles (round 5, 1 M splats, 1200x680, backward blend alone): plain render wide 209-211 us / lean 216; fused pair wide 252-253 (6-9 spilled registers) /
// lean 248; tracking's form without the colour sums wide 3 waves / lean 4 waves per SIMD: 0.517 -> 0.497 ms per iteration.
template <int Q, bool DUAL, bool COLORS = true, bool SIL = true>
__attribute__((amdgpu_waves_per_eu(GSR_BWD_WAVES, GSR_BWD_WAVES))) __global__ void __launch_bounds__(64)
K_blend_bwd(ImageView im, char* __restrict__ binning, GeomView g, const float* __restrict__ bg, int W, int H,
                 int grid_x, int ntiles, int tile0, const float* __restrict__ dL_dpix, const float* __restrict__ dL_dds)
{
    if constexpr (DUAL || !COLORS || GSR_BWD_LEAN_ALWAYS) blend_bwd_lean<Q, DUAL, COLORS, SIL>(im, binning, g, bg, W, H, grid_x, ntiles, tile0, dL_dpix, dL_dds);
    else blend_bwd_rgb<Q>(im, binning, g, bg, W, H, grid_x, ntiles, tile0, dL_dpix);
}

// =====================================================================================
// Forward ("patch rows"), same wave/row mapping as K_blend_bwd: the wave owns an 8x8 quad, its four
// 16-lane rows are four independent 4x4 patches with their own hit lists. The quad's list of hits (list
// position, id | patch mask) was cut by the tile sort (gsr_kernels.hip: emit_quad_lists); per round the
// wave takes 64 of them, front to back: gather (records two steps ahead, the whole 48-byte record one
// step ahead; every record is a hit: no culling here), patch lists from the masks, blend (row r walks
// list r, entries software-pipelined). A row whose 16 pixels are all done idles; the wave leaves when
// every pixel is done and notes how many records it took (qdone): the backward starts there.
// =====================================================================================
template <int Q, bool DUAL>
__global__ void __launch_bounds__(64)
K_blend_fwd(ImageView im, BinView bn, GeomView g, const float* __restrict__ bg, int W, int H,
                 int grid_x, int ntiles, int tile0, float* __restrict__ out_color, float* __restrict__ out_depth, int P,
                 float* __restrict__ out_ds, float* __restrict__ out_sil)
{
    static_assert(Q == 64, "one parked entry per lane");
    // parked entries; slot Q is a dummy that no pixel can see (opacity 0, far away): the per-patch lists are padded with
    // it, so the blend loop needs neither an "is this row still active" compare nor an index select
    __shared__ float4 E0[Q + 1], E1[Q + 1], E2[Q + 1]; // (px, py, a2, b2) (c2, opacity, red, green) (blue, depth, list position + 1, id)
    // per-patch hit lists: BYTE OFFSETS of the entries (index * 16: shifts and integer mads are half-rate on gfx950,
    // LDS loads are not VALU work at all), 4 slots of slack behind the longest list for the software pipeline
    __shared__ uint16_t LIST[4 * (Q + 4)];
    const uint32_t w = xcd_remap(blockIdx.x, 4u * (uint32_t)ntiles, 4u * GSR_XCD_TILES);
    const uint32_t tile = (uint32_t)tile0 + (w >> 2), quad = w & 3u;
    const int tx = tile % grid_x, ty = tile / grid_x;
    const int lane = threadIdx.x, r = lane >> 4, l = lane & 15;
    const int X0 = tx * 16 + (int)(quad & 1u) * 8, Y0 = ty * 16 + (int)(quad >> 1) * 8;
    const int px = X0 + (r & 1) * 4 + (l & 3), py = Y0 + (r >> 1) * 4 + (l >> 2);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    // (the range, the overflow flag and the quad's record count are asked for together: one trip to memory before the records, not two)
    const uint2 range = im.ranges[tile];
    const uint32_t ovf = g.hdr->overflow;
    const int cq0 = (int)im.qcount[4 * tile + quad];
    const int n = ovf ? 0 : (int)(range.y - range.x);

    float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f, C3 = 0.f, C4 = 0.f, Dp = 0.f;
    uint32_t last = 0u;
    wmask m_done = wm(!inside); // lanes whose pixel is finished (or outside the image)
    const int cq = n > 0 ? cq0 : 0;
    const uint2* __restrict__ qh = bn.qhits + 4 * (size_t)range.x + (size_t)quad * (size_t)n;
    constexpr uint32_t DUMMY = (uint32_t)Q * 16u;
    if (lane == 0) {
        E0[Q] = make_float4(-1.0e5f, -1.0e5f, -1.f, 0.f); // power2 ~ -2e10: exp2 gives 0
        E1[Q] = make_float4(-1.f, 0.f, 0.f, 0.f);
        E2[Q] = make_float4(0.f, 0.f, 0.f, 0.f);
    }

    int k0 = 0;
    if (cq > 0) {
        // unconditional loads from clamped, always valid addresses: the compiler can count them, and the gathers of the next
        // step do not wait for the loads issued after them
        uint2 rec_c = qh[min(lane, cq - 1)], rec_n = qh[min(lane + 64, cq - 1)];
        float4 a_c = g.g0[rec_c.y & GSR_ID_MASK], b_c = g.g1[rec_c.y & GSR_ID_MASK], c_c = g.col[rec_c.y & GSR_ID_MASK];
        while (k0 < cq) {
            const wmask dmask = m_done;
            if (dmask == ~0ull) break;
            // ---- gather: one step, <= 64 entries, entry e in the LDS slot of lane e
            const int count = min(64, cq - k0);
            uint32_t pm = 0u;
            {
                const uint2 rec = rec_c;
                const float4 a = a_c, b = b_c, c = c_c;
                rec_c = rec_n;
                a_c = g.g0[rec_c.y & GSR_ID_MASK]; b_c = g.g1[rec_c.y & GSR_ID_MASK]; c_c = g.col[rec_c.y & GSR_ID_MASK];
                rec_n = qh[min(k0 + lane + 128, cq - 1)];
                if (lane < count) {
                    E0[lane] = make_float4(a.x, a.y, a.z * (-0.5f * GSR_LOG2E), a.w * -GSR_LOG2E);
                    E1[lane] = make_float4(b.x * (-0.5f * GSR_LOG2E), b.y, c.x, c.y);
                    E2[lane] = make_float4(c.z, b.z, __uint_as_float(rec.x + 1u), __uint_as_float(rec.y & GSR_ID_MASK));
                    pm = rec.y >> GSR_ID_BITS;
                }
                k0 += 64;
            }
            // ---- per-patch hit lists (lane e looks at parked entry e; its patch mask came with the record)
            int c0, c1, c2, c3;
            {
                const bool h0 = (pm & 1u) != 0u, h1 = (pm & 2u) != 0u, h2 = (pm & 4u) != 0u, h3 = (pm & 8u) != 0u;
                const unsigned long long m0 = __ballot(h0), m1 = __ballot(h1), m2 = __ballot(h2), m3 = __ballot(h3);
                const uint16_t off = (uint16_t)(lane * 16);
                if (h0) LIST[0 * (Q + 4) + mbcnt64(m0)] = off;
                if (h1) LIST[1 * (Q + 4) + mbcnt64(m1)] = off;
                if (h2) LIST[2 * (Q + 4) + mbcnt64(m2)] = off;
                if (h3) LIST[3 * (Q + 4) + mbcnt64(m3)] = off;
                c0 = (int)__popcll(m0); c1 = (int)__popcll(m1); c2 = (int)__popcll(m2); c3 = (int)__popcll(m3);
            }
            // ---- blend: row r walks its own list; the wave runs as long as its longest unfinished row (an even number of
            //      iterations: the loop is unrolled by two), shorter lists are padded with the dummy entry
            const int e0 = ((dmask >> 0) & 0xFFFFull) == 0xFFFFull ? 0 : c0, e1 = ((dmask >> 16) & 0xFFFFull) == 0xFFFFull ? 0 : c1;
            const int e2 = ((dmask >> 32) & 0xFFFFull) == 0xFFFFull ? 0 : c2, e3 = ((dmask >> 48) & 0xFFFFull) == 0xFFFFull ? 0 : c3;
            const int maxc = (max(max(e0, e1), max(e2, e3)) + 1) & ~1;
            {
                const int cr = r == 0 ? c0 : r == 1 ? c1 : r == 2 ? c2 : c3;
                for (int p = cr + l; p < maxc + 4; p += 16) LIST[r * (Q + 4) + p] = (uint16_t)DUMMY;
            }
            lds_turn();
            const uint16_t* __restrict__ mylist = LIST + r * (Q + 4);
            // The loop reads 36 of an entry's 48 bytes: the list position (for n_contrib) and the depth (for the median depth) of
            // the LAST entry that updated a pixel are looked up once per round through the iteration index instead of being
            // carried through every iteration (the forward is bound by the LDS pipe: 0.81 busy, profiles/r03a_pmc.md)
            int li = -1, di = -1;
            // MEDIAN: some pixel of the quad still has T > 0.5 at the start of the round (T only falls: decided once per round, wave-uniform). Once
            // none has, the median-depth bookkeeping (a compare and a select, both half-rate) leaves the loop — on the bench frame a pixel is below
            // 0.5 after one or two of its ~50 contributors.
            // (Round 5, measured and dropped: the step on packed fp32 instructions — (dx, dy) one v_pk_add, (b2 dy, c2 dy) one v_pk_mul, (red, green) and
            // (blue, depth) one v_pk_fma each, bit-identical results, 47 -> 41 vector instructions per two entries at the same seven waves per SIMD:
            // 95.1 us against 93.1. Fewer instructions are not what this loop is short of.)
            auto step = [&](auto MEDIAN, const int it, const float4 A, const float4 B, const float Cb) {
                const float dx = A.x - pxf, dy = A.y - pyf;
                const float power2 = pair_power2(dx, dy, A.z, A.w, B.x); // = power * log2(e): same sign as power
                const float alpha = fminf(0.99f, B.y * __builtin_amdgcn_exp2f(power2));
                const wmask valid = ~m_done & wm(power2 <= 0.0f) & wm(alpha >= GSR_ALPHA_MIN);
                const float test_T = T * (1.f - alpha);
                const wmask lt = wm(test_T < 0.0001f), upd = valid & ~lt;
                m_done |= valid & lt;
                const bool u = lane_of(upd);
                const float wgt = u ? alpha * T : 0.f;
                C0 = fmaf(B.z, wgt, C0);
                C1 = fmaf(B.w, wgt, C1);
                C2 = fmaf(Cb, wgt, C2);
                if (DUAL) C4 += wgt;                              // accumulated opacity (the depth channel: below, per round)
                if (decltype(MEDIAN)::value) di = lane_of(upd & wm(T > 0.5f)) ? it : di; // median depth (forward.cu:374-379)
                T = u ? test_T : T;
                li = u ? it : li;
                return wgt;
            };
            if (maxc > 0) {
                uint32_t o0 = mylist[0], o1 = mylist[1];
                float4 A0 = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(E0) + o0);
                float4 B0 = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(E1) + o0);
                float Z0 = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(E2) + o0), Zd0 = 0.f, Zd1 = 0.f;
                if (DUAL) Zd0 = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(E2) + o0 + 4);
                float4 A1, B1;
                float Z1;
                auto run = [&](auto MED) {
                for (int it = 0; it < maxc; it += 2) {
                    A1 = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(E0) + o1);
                    B1 = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(E1) + o1);
                    Z1 = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(E2) + o1);
                    if (DUAL) Zd1 = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(E2) + o1 + 4);
                    o0 = mylist[it + 2];
                    __builtin_amdgcn_sched_barrier(0); // keep the prefetch above the iteration it overlaps with
                    const float w0 = step(MED, it, A0, B0, Z0);
                    if (DUAL) C3 = fmaf(Zd0, w0, C3);             // alpha-blended view depth
                    A0 = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(E0) + o0);
                    B0 = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(E1) + o0);
                    Z0 = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(E2) + o0);
                    if (DUAL) Zd0 = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(E2) + o0 + 4);
                    o1 = mylist[it + 3];
                    __builtin_amdgcn_sched_barrier(0);
                    const float w1 = step(MED, it + 1, A1, B1, Z1);
                    if (DUAL) C3 = fmaf(Zd1, w1, C3);
                }
                };
                if (wm(T > 0.5f) & ~m_done) run(std::true_type{});
                else run(std::false_type{});
                // the round's last updates: list position + 1 and depth of the entries behind the two indices
                const uint32_t ol = mylist[max(li, 0)], od = mylist[max(di, 0)];
                const uint32_t lpos = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(E2) + ol + 8);
                const float ddep = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(E2) + od + 4);
                last = li >= 0 ? lpos : last;
                Dp = di >= 0 ? ddep : Dp;
            }
            lds_turn();
        }
    }
    if (lane == 0) im.qdone[4 * tile + quad] = (uint32_t)min(k0, cq);
    if (inside) {
        const size_t pix = (size_t)py * W + px, HW = (size_t)H * W;
        im.final_T[pix] = T;
        im.n_contrib[pix] = last;
        out_color[pix] = C0 + T * bg[0];
        out_color[HW + pix] = C1 + T * bg[1];
        out_color[2 * HW + pix] = C2 + T * bg[2];
        out_depth[pix] = Dp;
        if (DUAL) { out_ds[pix] = C3; out_ds[HW + pix] = C4; }
        else if (out_sil) out_sil[pix] = 1.f - T; // (gsr_forward_args.out_sil: sum alpha T of this walk)
    }
    { // The backward accumulators (64 bytes per splat) must be zero when the forward is done. Clearing them is
      // pure memory traffic and this kernel is pure VALU work, so every block clears its share here for free
      // (inside K_preprocess the same stores cost ~10 us at 1 M splats). Last thing the wave does: nothing waits for them.
        const size_t total = (size_t)P * (GSR_ACC_STRIDE / 4), per = (total + gridDim.x - 1) / gridDim.x;
        const size_t b0 = (size_t)blockIdx.x * per, b1 = b0 + per < total ? b0 + per : total;
        float4* const acc4 = reinterpret_cast<float4*>(g.acc);
        for (size_t i = b0 + threadIdx.x; i < b1; i += 64) acc4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

} // namespace gsr
