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

namespace gsr {

#define GSR_ALPHA_MIN (1.0f / 255.0f)
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
// Backward ("patch rows"): the wave still owns an 8x8 quad, but its four 16-lane DPP rows are four
// INDEPENDENT 4x4 pixel patches, each walking its own hit list. Small splats touch few pixels of
// an 8x8 quad (~16 of 64 lanes useful on the 1 M-splat workload); a 4x4 patch is hit by half as many
// splats as the quad and uses ~8 of its 16 lanes, so the same wave retires ~1.7x fewer iterations.
//   gather : the forward logged which list entries reach the quad and which of its patches (qhits), so
//            the backward never touches the rest of the tile list: 64 records per step, back to front,
//            records behind every pixel's last contributor dropped, the rest parked in LDS;
//   lists  : one lane per parked entry appends the entry's index to the byte list of every patch in
//            its mask (list order is kept);
//   blend  : row r walks list r (entries software-pipelined through two register sets); per iteration
//            the nine partial sums are reduced inside the 16-lane row (transposing, bank-masked DPP
//            adds) and nine lanes per row add them into the entry's LDS accumulator with a plain
//            read-modify-write — the four rows usually work on four different splats; the iterations
//            in which two rows meet on one entry are found beforehand and use ds_add_f32 instead
//            (LDS float atomics retire ~1 lane per 3 cycles: using them always costs +100 us);
//   flush  : LDS accumulators -> one 9-lane global atomic per parked entry that was hit (7 per
//            instruction), i.e. the same number of L2 atomic records as the quad kernel. Issuing the
//            atomics per PATCH instead would double them and hit the L2 atomic ceiling (~20 G records/s,
//            scripts/atomic_bench2.hip).
// =====================================================================================
#define GSR_FWDQ 96 // forward: gathers until more than 32 entries are parked (2-3 steps of ~22 quad hits)
#define GSR_ROWQ 64 // parked entries per round = one gather step; 64 beats 96 and 128 (LDS 5.9 KB per wave)
typedef float v2f __attribute__((ext_vector_type(2)));

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

// Transposing reduction of nine values inside each 16-lane row. Stage s pairs lanes through a DPP
// permutation that flips bit (3-s) of the lane number; the lane keeps one value of a pair and hands the
// other to its partner, so the live registers go 9 -> 5 -> 3 -> 2 -> 1 (21 VALU with the bank-masked first two
// stages of row_reduce9, 27 with selects throughout). On return lane l of the
// row holds the row total of value rows_slot_of(l).
__device__ __forceinline__ int rows_slot_of(int l)
{
    if (l & 1) return l == 1 ? 8 : -1;
    return ((l >> 3) & 1) | (((l >> 2) & 1) << 1) | (((l >> 1) & 1) << 2);
}
template <int CTRL>
__device__ __forceinline__ float tr_pair(bool hi, float even, float odd)
{
    const float keep = hi ? odd : even, give = hi ? even : odd;
    return keep + dpp_f<CTRL>(give);
}
// `early` is any value that must already be in a register when the reduction starts (the caller's LDS
// accumulator read: naming it here keeps the compiler from sinking that load behind the reduction).
__device__ __forceinline__ float row_reduce9(const float (&v)[9], int l, float early)
{
    // Stages 1 and 2 select by DPP bank (a bank = 4 lanes): the lanes of banks {0,1} / {2,3} (stage 1) and
    // {0,2} / {1,3} (stage 2) are exactly the lanes that keep the even / odd value of a pair, so two
    // bank-masked v_add_f32_dpp writing one register replace two v_cndmask + one add. Hand-written
    // because the compiler cannot express a partial-bank destination; s_nop covers the VALU-write ->
    // DPP-read hazard at the block entry, inside the block every DPP source is >= 4 instructions old.
    float a0, a1, a2, a3, a4, c0, c1, c2;
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %8, %8 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %0, %9, %9 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %1, %10, %10 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %1, %11, %11 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %2, %12, %12 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %2, %13, %13 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %3, %14, %14 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %3, %15, %15 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %4, %16, %16 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %5, %0, %0 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %5, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %6, %2, %2 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %6, %3, %3 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %7, %4, %4 row_half_mirror row_mask:0xf bank_mask:0xf"
        : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(c0), "=&v"(c1), "=&v"(c2)
        : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]), "v"(early));
    const bool b1 = (l & 2) != 0, b0 = (l & 1) != 0;
    const float d0 = tr_pair<0x1B>(b1, c0, c1);                                             // quad_perm [3,2,1,0]  l <-> l^3
    const float d1 = c2 + dpp_f<0x1B>(c2);
    return tr_pair<0xB1>(b0, d0, d1);                                                       // quad_perm [1,0,3,2]  l <-> l^1
}

template <int Q>
__global__ void __launch_bounds__(64)
K_blend_bwd(ImageView im, char* __restrict__ binning, GeomView g, const float* __restrict__ bg, int W, int H,
                 int grid_x, int ntiles, int tile0, const float* __restrict__ dL_dpix)
{
    // parked entries; slot Q is a dummy (opacity 0, far away) the per-patch lists are padded with: no "row still active"
    // compare and no index select in the blend loop; what the idle rows add to its accumulator record is never flushed
    __shared__ float4 E0[Q + 1], E1[Q + 1], E2[Q + 1]; // (px, py, a2, b2) (c2, opacity, red, green) (blue, list position, splat id, patch mask)
    __shared__ float ACC[(Q + 1) * 9];
    // per-patch hit lists as BYTE OFFSETS (entry * 16 into E0/E1/E2, entry * 36 into ACC): shifts and integer mads are
    // half-rate VALU work on gfx950, a second 2-byte LDS load is not VALU work at all
    __shared__ uint16_t LIST[4 * (Q + 4)], LISTA[4 * (Q + 4)];
    const uint32_t w = xcd_remap(blockIdx.x, 4u * (uint32_t)ntiles);
    const uint32_t tile = (uint32_t)tile0 + (w >> 2), quad = w & 3u;
    const int tx = tile % grid_x, ty = tile / grid_x;
    const int lane = threadIdx.x, r = lane >> 4, l = lane & 15;
    const int X0 = tx * 16 + (int)(quad & 1u) * 8, Y0 = ty * 16 + (int)(quad >> 1) * 8;
    const int px = X0 + (r & 1) * 4 + (l & 3), py = Y0 + (r >> 1) * 4 + (l >> 2);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const uint2 range = im.ranges[tile];
    const int n = g.hdr->overflow ? 0 : (int)(range.y - range.x);
    BinView bn; // the layout of the binning blob follows the capacity the forward ran with (kept in the header)
    binning_layout(binning, (size_t)g.hdr->capacity, &bn);
    const size_t pix = (size_t)py * W + px, HW = (size_t)H * W;

    const float T_final = inside ? im.final_T[pix] : 0.f;
    float T = T_final;
    const uint32_t last = inside ? im.n_contrib[pix] : 0u;
    const float g0 = inside ? dL_dpix[pix] : 0.f, g1 = inside ? dL_dpix[HW + pix] : 0.f,
                g2 = inside ? dL_dpix[2 * HW + pix] : 0.f;
    const float bg_dot = bg[0] * g0 + bg[1] * g1 + bg[2] * g2;
    const v2f g01 = {g0, g1};
    const float nTf_bg = -T_final * bg_dot;
    // colour accumulated behind the current splat (the reference's accum_rec, updated eagerly:
    // last_alpha*last_color + (1-last_alpha)*accum_rec == fma(alpha, c - S, S) one step later). Kept per
    // channel: c - S is formed BEFORE the contraction with the pixel gradient — neighbouring splats have
    // similar colours (depth renders!), and contracting first turns an exact small difference into the
    // difference of two rounded large numbers (measured: 9e-5 instead of 1e-6 on long lists).
    float S0 = 0.f, S1 = 0.f, S2 = 0.f;
    const int slot = rows_slot_of(l);
    const bool has_slot = slot >= 0; // the nine lanes of a row that end up holding a row total
    const uint32_t slot_b = has_slot ? 4u * (uint32_t)slot : 0u;
    const int fe = (lane * 57) >> 9, fc = lane - 9 * fe; // lane / 9, lane % 9: flush lane -> (entry, component)
    for (int i = lane; i < (Q + 1) * 9; i += 64) ACC[i] = 0.f;
    if (lane == 0) {
        E0[Q] = make_float4(-1.0e5f, -1.0e5f, -1.f, 0.f); // power2 ~ -2e10: exp2 gives 0
        E1[Q] = make_float4(-1.f, 0.f, 0.f, 0.f);
        E2[Q] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int ntodo = __builtin_amdgcn_readfirstlane(min(n, (int)wave_max_u32(last)));

    // The forward logged the entries that reach this quad (list position, id), in list order; walk them
    // back to front. Gather pipeline: the records of the next two steps and the geometry of the next step
    // are in flight (unconditional loads from clamped, always valid addresses so that the compiler can
    // count them: the colour gather must not wait for the loads issued after it).
    const int cq = (int)im.qcount[4 * tile + quad];
    const uint2* __restrict__ qh = bn.qhits + 4 * (size_t)range.x + (size_t)quad * (size_t)n;
    if (ntodo <= 0 || cq <= 0) return;
    int k0 = 0;
    uint2 rec_c = qh[max(cq - 1 - lane, 0)];
    uint2 rec_n = qh[max(cq - 1 - (lane + 64), 0)];
    float4 a_c = g.g0[rec_c.y & GSR_ID_MASK], b_c = g.g1[rec_c.y & GSR_ID_MASK];
    while (k0 < cq) {
        // ---- gather + compaction (records past the last contributor of every pixel are dropped)
        int count = 0;
        do {
            const uint32_t id = rec_c.y & GSR_ID_MASK, pos = rec_c.x, pmask = rec_c.y >> GSR_ID_BITS;
            const float4 a = a_c, b = b_c;
            const int k = k0 + lane;
            const bool hit = k < cq && pos < (uint32_t)ntodo;
            float4 c;
            if (hit) c = g.col[id];
            rec_c = rec_n;
            a_c = g.g0[rec_c.y & GSR_ID_MASK]; b_c = g.g1[rec_c.y & GSR_ID_MASK];
            rec_n = qh[max(cq - 1 - (k + 128), 0)];
            const unsigned long long m = __ballot(hit);
            if (hit) {
                const int e = count + mbcnt64(m);
                E0[e] = make_float4(a.x, a.y, a.z * (-0.5f * GSR_LOG2E), a.w * -GSR_LOG2E);
                E1[e] = make_float4(b.x * (-0.5f * GSR_LOG2E), b.y, c.x, c.y);
                E2[e] = make_float4(c.z, __uint_as_float(pos), __uint_as_float(id), __uint_as_float(pmask));
            }
            count += (int)__popcll(m);
            k0 += 64;
        } while (k0 < cq && count <= Q - 64);
        if (count == 0) continue;
        __builtin_amdgcn_wave_barrier();
        // ---- per-patch hit lists
        int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
        for (int eb = 0; eb < count; eb += 64) {
            const int e = eb + lane;
            bool h[4] = {false, false, false, false};
            if (e < count) { // the forward already ran the patch cull: its verdict travels in the record
                const uint32_t pm = __float_as_uint(E2[e].w);
                h[0] = (pm & 1u) != 0u; h[1] = (pm & 2u) != 0u; h[2] = (pm & 4u) != 0u; h[3] = (pm & 8u) != 0u;
            }
            const unsigned long long m0 = __ballot(h[0]), m1 = __ballot(h[1]), m2 = __ballot(h[2]), m3 = __ballot(h[3]);
            const uint16_t off = (uint16_t)(e * 16), offa = (uint16_t)(e * 36);
            if (h[0]) { const int p = 0 * (Q + 4) + c0 + mbcnt64(m0); LIST[p] = off; LISTA[p] = offa; }
            if (h[1]) { const int p = 1 * (Q + 4) + c1 + mbcnt64(m1); LIST[p] = off; LISTA[p] = offa; }
            if (h[2]) { const int p = 2 * (Q + 4) + c2 + mbcnt64(m2); LIST[p] = off; LISTA[p] = offa; }
            if (h[3]) { const int p = 3 * (Q + 4) + c3 + mbcnt64(m3); LIST[p] = off; LISTA[p] = offa; }
            c0 += (int)__popcll(m0); c1 += (int)__popcll(m1); c2 += (int)__popcll(m2); c3 += (int)__popcll(m3);
        }
        __builtin_amdgcn_wave_barrier();
        // ---- blend: row r walks its own list; the loop runs an even number of iterations (unrolled by two), shorter
        //      lists are padded with the dummy entry
        const int maxc = (max(max(c0, c1), max(c2, c3)) + 1) & ~1;
        // iterations in which two rows work on the same parked entry: those must accumulate atomically
        unsigned long long cm[(Q + 63) / 64];
#pragma unroll
        for (int b = 0; b < (Q + 63) / 64; b++) {
            const int t = b * 64 + lane;
            bool coll = false;
            if (t < maxc) {
                const int v0 = t < c0 ? (int)LIST[0 * (Q + 4) + t] : 0x10000, v1 = t < c1 ? (int)LIST[1 * (Q + 4) + t] : 0x10001;
                const int v2 = t < c2 ? (int)LIST[2 * (Q + 4) + t] : 0x10002, v3 = t < c3 ? (int)LIST[3 * (Q + 4) + t] : 0x10003;
                coll = v0 == v1 || v0 == v2 || v0 == v3 || v1 == v2 || v1 == v3 || v2 == v3;
            }
            cm[b] = __ballot(coll);
        }
        {
            const int cr = r == 0 ? c0 : r == 1 ? c1 : r == 2 ? c2 : c3;
            for (int p = cr + l; p < maxc + 4; p += 16) { LIST[r * (Q + 4) + p] = (uint16_t)(Q * 16); LISTA[r * (Q + 4) + p] = (uint16_t)(Q * 36); }
        }
        __builtin_amdgcn_wave_barrier();
        const uint16_t* __restrict__ mylist = LIST + r * (Q + 4);
        const uint16_t* __restrict__ mylista = LISTA + r * (Q + 4);
        // One iteration on an entry already in registers. Accumulation is a plain LDS read-modify-write by the nine
        // slot lanes of the row (LDS float atomics cost ~240 cycles per wave instruction: scripts/valu_bench2.hip);
        // only the iterations flagged in cm (two rows on one entry) use the atomic.
        auto step = [&](const int it, const uint32_t offa, const float4 A, const float4 B, const float4 Cz) {
            float* const accp = reinterpret_cast<float*>(reinterpret_cast<char*>(ACC) + offa + slot_b);
            const float acc_old = *accp; // every lane reads (the other seven of a row re-read component 0: same address, broadcast)
            __builtin_amdgcn_sched_barrier(0); // issue the accumulator read here, a whole iteration ahead of its use
            const float dx = A.x - pxf, dy = A.y - pyf;
            const float power2 = pair_power2(dx, dy, A.z, A.w, B.x); // = power * log2(e)
            const float Graw = __builtin_amdgcn_exp2f(power2);
            const float araw = fminf(0.99f, B.y * Graw);
            const bool valid = lane_of(wm(__float_as_uint(Cz.y) < last) & wm(power2 <= 0.0f) & wm(araw >= GSR_ALPHA_MIN));
            const float alpha = valid ? araw : 0.f, G = valid ? Graw : 0.f;
            const float ia = __builtin_amdgcn_rcpf(1.f - alpha);
            T = T * ia;
            const float dcol = alpha * T;
            const float e0 = B.z - S0, e1 = B.w - S1, e2 = Cz.x - S2;
            const float eg = fmaf(e2, g2, fmaf(e1, g1, e0 * g0)); // (colour - accum_rec) . dL_dpix
            const float dL_dalpha = fmaf(nTf_bg, ia, eg * T); // - T_final/(1-alpha) * (bg . dL_dpix)
            S0 = fmaf(alpha, e0, S0); S1 = fmaf(alpha, e1, S1); S2 = fmaf(alpha, e2, S2);
            const float u = G * dL_dalpha;
            const float udx = u * dx, udy = u * dy;
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
            const float mine = row_reduce9(v, l, acc_old);
            const bool collide = Q <= 64 ? ((cm[0] >> it) & 1ull) != 0ull : ((cm[it >> 6] >> (it & 63)) & 1ull) != 0ull;
            if (has_slot) {
                if (!collide) *accp = acc_old + mine;
                else unsafeAtomicAdd(accp, mine);
            }
        };
        // software pipeline, unrolled by two so that the two register sets alternate without copies:
        // the entry of the next iteration and the list offsets of the one after are always in flight
        uint32_t o0 = mylist[0], oa0 = mylista[0], o1 = mylist[1], oa1 = mylista[1];
        float4 A0 = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(E0) + o0);
        float4 B0 = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(E1) + o0);
        float4 C0 = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(E2) + o0);
        float4 A1, B1, C1;
        for (int it = 0; it < maxc; it += 2) {
            A1 = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(E0) + o1);
            B1 = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(E1) + o1);
            C1 = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(E2) + o1);
            o0 = mylist[it + 2];
            const uint32_t oa_cur0 = oa0;
            oa0 = mylista[it + 2];
            step(it, oa_cur0, A0, B0, C0);
            A0 = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(E0) + o0);
            B0 = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(E1) + o0);
            C0 = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(E2) + o0);
            o1 = mylist[it + 3];
            const uint32_t oa_cur1 = oa1;
            oa1 = mylista[it + 3];
            step(it + 1, oa_cur1, A1, B1, C1);
        }
        __builtin_amdgcn_wave_barrier();
        // ---- flush: seven parked entries per instruction, nine consecutive lanes per 64-byte record
        for (int fb = 0; fb < count; fb += 7) {
            const int e = fb + fe;
            if (lane < 63 && e < count) {
                const float val = ACC[e * 9 + fc];
                if (val != 0.f) {
                    ACC[e * 9 + fc] = 0.f;
                    unsafeAtomicAdd(&g.acc[(size_t)__float_as_uint(E2[e].z) * GSR_ACC_STRIDE + fc], val);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// =====================================================================================
// Forward ("patch rows"), same wave/row mapping as K_blend_bwd: the wave owns an 8x8 quad, its four
// 16-lane rows are four independent 4x4 patches with their own hit lists. Per round: gather (64 list
// entries per step, exact quad cull, survivors compacted into LDS until more than Q-64 are parked),
// patch lists (exact patch cull; the entry is logged for the backward as (list position, id | mask<<28)),
// blend (row r walks list r, entries software-pipelined). A row whose 16 pixels are all done idles; the
// wave leaves when every pixel is done.
// =====================================================================================
template <int Q>
__global__ void __launch_bounds__(64)
K_blend_fwd(ImageView im, BinView bn, GeomView g, const float* __restrict__ bg, int W, int H,
                 int grid_x, int ntiles, int tile0, float* __restrict__ out_color, float* __restrict__ out_depth, int P)
{
    // parked entries; slot Q is a dummy that no pixel can see (opacity 0, far away): the per-patch lists are padded with
    // it, so the blend loop needs neither an "is this row still active" compare nor an index select
    __shared__ float4 E0[Q + 1], E1[Q + 1], E2[Q + 1]; // (px, py, a2, b2) (c2, opacity, red, green) (blue, depth, list position + 1, id)
    // per-patch hit lists: BYTE OFFSETS of the entries (index * 16: shifts and integer mads are half-rate on gfx950,
    // LDS loads are not VALU work at all), 4 slots of slack behind the longest list for the software pipeline
    __shared__ uint16_t LIST[4 * (Q + 4)];
    const uint32_t w = xcd_remap(blockIdx.x, 4u * (uint32_t)ntiles);
    const uint32_t tile = (uint32_t)tile0 + (w >> 2), quad = w & 3u;
    const int tx = tile % grid_x, ty = tile / grid_x;
    const int lane = threadIdx.x, r = lane >> 4, l = lane & 15;
    const int X0 = tx * 16 + (int)(quad & 1u) * 8, Y0 = ty * 16 + (int)(quad >> 1) * 8;
    const int px = X0 + (r & 1) * 4 + (l & 3), py = Y0 + (r >> 1) * 4 + (l >> 2);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py, X0f = (float)X0, Y0f = (float)Y0;
    const uint2 range = im.ranges[tile];
    const int n = g.hdr->overflow ? 0 : (int)(range.y - range.x);
    const uint32_t* __restrict__ plist = bn.point_list + range.x;

    float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f, Dp = 0.f;
    uint32_t last = 0u;
    wmask m_done = wm(!inside); // lanes whose pixel is finished (or outside the image)
    uint2* __restrict__ qh = bn.qhits + 4 * (size_t)range.x + (size_t)quad * (size_t)n;
    int qc = 0;
    constexpr uint32_t DUMMY = (uint32_t)Q * 16u;
    if (lane == 0) {
        E0[Q] = make_float4(-1.0e5f, -1.0e5f, -1.f, 0.f); // power2 ~ -2e10: exp2 gives 0
        E1[Q] = make_float4(-1.f, 0.f, 0.f, 0.f);
        E2[Q] = make_float4(0.f, 0.f, 0.f, 0.f);
    }

    if (n > 0) {
        int base = 0;
        uint32_t id_c = plist[min(lane, n - 1)], id_n = plist[min(lane + 64, n - 1)];
        float4 a_c = g.g0[id_c], b_c = g.g1[id_c];
        while (base < n) {
            const wmask dmask = m_done;
            if (dmask == ~0ull) break;
            // ---- gather + quad cull + compaction
            int count = 0;
            do {
                const uint32_t id = id_c;
                const float4 a = a_c, b = b_c;
                const int k = base + lane;
                const bool hit = k < n && quad_reach(a, b, X0f, Y0f);
                float4 c;
                if (hit) c = g.col[id];
                id_c = id_n;
                a_c = g.g0[id_c]; b_c = g.g1[id_c];
                id_n = plist[min(k + 128, n - 1)];
                const unsigned long long m = __ballot(hit);
                if (hit) {
                    const int e = count + mbcnt64(m);
                    E0[e] = make_float4(a.x, a.y, a.z * (-0.5f * GSR_LOG2E), a.w * -GSR_LOG2E);
                    E1[e] = make_float4(b.x * (-0.5f * GSR_LOG2E), b.y, c.x, c.y);
                    E2[e] = make_float4(c.z, b.z, __uint_as_float((uint32_t)k + 1u), __uint_as_float(id));
                }
                count += (int)__popcll(m);
                base += 64;
            } while (base < n && count <= Q - 64);
            if (count == 0) continue;
            __builtin_amdgcn_wave_barrier();
            // ---- per-patch hit lists + the log for the backward
            int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
            for (int eb = 0; eb < count; eb += 64) {
                const int e = eb + lane;
                bool h[4] = {false, false, false, false};
                float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
                if (e < count) { patch_reach4(E0[e], E1[e], X0f, Y0f, h); z = E2[e]; }
                const unsigned long long m0 = __ballot(h[0]), m1 = __ballot(h[1]), m2 = __ballot(h[2]), m3 = __ballot(h[3]);
                const uint16_t off = (uint16_t)(e * 16);
                if (h[0]) LIST[0 * (Q + 4) + c0 + mbcnt64(m0)] = off;
                if (h[1]) LIST[1 * (Q + 4) + c1 + mbcnt64(m1)] = off;
                if (h[2]) LIST[2 * (Q + 4) + c2 + mbcnt64(m2)] = off;
                if (h[3]) LIST[3 * (Q + 4) + c3 + mbcnt64(m3)] = off;
                c0 += (int)__popcll(m0); c1 += (int)__popcll(m1); c2 += (int)__popcll(m2); c3 += (int)__popcll(m3);
                const uint32_t pm = (h[0] ? 1u : 0u) | (h[1] ? 2u : 0u) | (h[2] ? 4u : 0u) | (h[3] ? 8u : 0u);
                const unsigned long long ma = m0 | m1 | m2 | m3;
                if (pm) qh[qc + mbcnt64(ma)] = make_uint2(__float_as_uint(z.z) - 1u, __float_as_uint(z.w) | (pm << GSR_ID_BITS));
                qc += (int)__popcll(ma);
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
            __builtin_amdgcn_wave_barrier();
            const uint16_t* __restrict__ mylist = LIST + r * (Q + 4);
            auto step = [&](const float4 A, const float4 B, const float4 Cz) {
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
                C2 = fmaf(Cz.x, wgt, C2);
                Dp = lane_of(upd & wm(T > 0.5f)) ? Cz.y : Dp; // median depth (forward.cu:374-379)
                T = u ? test_T : T;
                last = u ? __float_as_uint(Cz.z) : last;
            };
            if (maxc > 0) {
                uint32_t o0 = mylist[0], o1 = mylist[1];
                float4 A0 = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(E0) + o0);
                float4 B0 = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(E1) + o0);
                float4 Z0 = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(E2) + o0);
                float4 A1, B1, Z1;
                for (int it = 0; it < maxc; it += 2) {
                    A1 = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(E0) + o1);
                    B1 = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(E1) + o1);
                    Z1 = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(E2) + o1);
                    o0 = mylist[it + 2];
                    __builtin_amdgcn_sched_barrier(0); // keep the prefetch above the iteration it overlaps with
                    step(A0, B0, Z0);
                    A0 = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(E0) + o0);
                    B0 = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(E1) + o0);
                    Z0 = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(E2) + o0);
                    o1 = mylist[it + 3];
                    __builtin_amdgcn_sched_barrier(0);
                    step(A1, B1, Z1);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (lane == 0) im.qcount[4 * tile + quad] = (uint32_t)qc;
    if (inside) {
        const size_t pix = (size_t)py * W + px, HW = (size_t)H * W;
        im.final_T[pix] = T;
        im.n_contrib[pix] = last;
        out_color[pix] = C0 + T * bg[0];
        out_color[HW + pix] = C1 + T * bg[1];
        out_color[2 * HW + pix] = C2 + T * bg[2];
        out_depth[pix] = Dp;
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
