// Issue cost of the VALU instructions the blend kernels are made of, on one SIMD of gfx950, as a function of
// the number of resident waves per SIMD. Answers: is a wave64 v_fma_f32 2 or 4 cycles? does v_pk_fma_f32
// do two lanes' worth of work in the slot of one? what do v_exp_f32 / v_rcp_f32 / DPP adds / v_cndmask cost?
// Every body is 32 independent instructions in inline asm (the compiler cannot fold or reorder them),
// repeated ITER times; cycles are read with s_memtime inside the wave (shader clock).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_bench scripts/valu_bench.hip && /tmp/valu_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>

#define ITER 2000

#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)

enum Kind { FMA, PKFMA, PKMUL, PKADD, EXP, RCP, DPPADD_ROR, DPPADD_QUAD, DPPADD_BANK, CNDMASK, CMP, MINF, MIX_FWD, MUL, CND_VCCDEF, CND_E64, CMP_CND, CND_ZERO, ADDF, SUBSGPR, ANDB, MADI24, FMA_DEP2, NKIND };
static const char* kname[NKIND] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_exp_f32", "v_rcp_f32",
    "v_add_f32_dpp row_ror:8", "v_add_f32_dpp quad_perm", "v_add_f32_dpp bank_mask", "v_cndmask_b32", "v_cmp_lt_f32", "v_min_f32",
    "mix: 12 pk_fma + 2 exp + 18 plain", "v_mul_f32", "v_cndmask vcc (vcc written before loop)", "v_cndmask_e64 s[10:11]", "v_cmp + v_cndmask pairs (16+16)", "v_cndmask 0, v, vcc (other dst)", "v_add_f32", "v_sub_f32 v, s, v (SGPR operand)", "v_and_b32", "v_mad_i32_i24", "v_fma_f32 2 chains (dependent)"};

template <int K>
__global__ void __launch_bounds__(256) bench(float* out, unsigned long long* cyc, float seed)
{
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b0 = a0 * 0.5f, b1 = a1 * 0.5f, b2 = a2 * 0.5f, b3 = a3 * 0.5f, b4 = a4 * 0.5f, b5 = a5 * 0.5f, b6 = a6 * 0.5f, b7 = a7 * 0.5f;
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f p0 = {a0, b0}, p1 = {a1, b1}, p2 = {a2, b2}, p3 = {a3, b3}, p4 = {a4, b4}, p5 = {a5, b5}, p6 = {a6, b6}, p7 = {a7, b7};
    const float c = 0.999f, d = 1e-6f;
    const v2f c2 = {c, c}, d2 = {d, d};
    const unsigned vlo = __builtin_amdgcn_readfirstlane(0x5555aaaau ^ (unsigned)(seed > 2.f)), vhi = __builtin_amdgcn_readfirstlane(0xaaaa5555u ^ (unsigned)(seed > 3.f));
    const unsigned long long vccinit = ((unsigned long long)vhi << 32) | vlo;
    if (K == CND_VCCDEF || K == CND_ZERO) asm volatile("s_mov_b64 vcc, %0" :: "s"(vccinit) : "vcc");
    unsigned long long t0 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll 1
    for (int it = 0; it < ITER; it++) {
        if (K == FMA) {
            REP4(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));)
        } else if (K == MUL) {
            REP4(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                         "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));)
        } else if (K == PKFMA) {
            REP4(asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                         "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c2), "v"(d2));)
        } else if (K == PKMUL) {
            REP4(asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                         "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c2));)
        } else if (K == PKADD) {
            REP4(asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                         "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(d2));)
        } else if (K == EXP) {
            REP4(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                         "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (K == RCP) {
            REP4(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                         "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (K == DPPADD_ROR) {
            REP4(asm volatile("v_add_f32_dpp %0, %8, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %9, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                         "v_add_f32_dpp %2, %10, %2 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %11, %3 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                         "v_add_f32_dpp %4, %12, %4 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %13, %5 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                         "v_add_f32_dpp %6, %14, %6 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %15, %7 row_ror:8 row_mask:0xf bank_mask:0xf"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                         : "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(b4), "v"(b5), "v"(b6), "v"(b7));)
        } else if (K == DPPADD_QUAD) {
            REP4(asm volatile("v_add_f32_dpp %0, %8, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %9, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                         "v_add_f32_dpp %2, %10, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %11, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                         "v_add_f32_dpp %4, %12, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %13, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                         "v_add_f32_dpp %6, %14, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %15, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                         : "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(b4), "v"(b5), "v"(b6), "v"(b7));)
        } else if (K == DPPADD_BANK) {
            REP4(asm volatile("v_add_f32_dpp %0, %8, %8 row_half_mirror row_mask:0xf bank_mask:0x5\n v_add_f32_dpp %0, %9, %9 row_half_mirror row_mask:0xf bank_mask:0xa\n"
                         "v_add_f32_dpp %1, %10, %10 row_half_mirror row_mask:0xf bank_mask:0x5\n v_add_f32_dpp %1, %11, %11 row_half_mirror row_mask:0xf bank_mask:0xa\n"
                         "v_add_f32_dpp %2, %12, %12 row_half_mirror row_mask:0xf bank_mask:0x5\n v_add_f32_dpp %2, %13, %13 row_half_mirror row_mask:0xf bank_mask:0xa\n"
                         "v_add_f32_dpp %3, %14, %14 row_half_mirror row_mask:0xf bank_mask:0x5\n v_add_f32_dpp %3, %15, %15 row_half_mirror row_mask:0xf bank_mask:0xa"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                         : "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(b4), "v"(b5), "v"(b6), "v"(b7));)
        } else if (K == CNDMASK) {
            REP4(asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                         "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "vcc");)
        } else if (K == CMP) {
            REP4(asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cmp_lt_f32 vcc, %1, %8\n v_cmp_lt_f32 vcc, %2, %8\n v_cmp_lt_f32 vcc, %3, %8\n"
                         "v_cmp_lt_f32 vcc, %4, %8\n v_cmp_lt_f32 vcc, %5, %8\n v_cmp_lt_f32 vcc, %6, %8\n v_cmp_lt_f32 vcc, %7, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "vcc");)
        } else if (K == MINF) {
            REP4(asm volatile("v_min_f32 %0, %0, %8\n v_min_f32 %1, %1, %8\n v_min_f32 %2, %2, %8\n v_min_f32 %3, %3, %8\n"
                         "v_min_f32 %4, %4, %8\n v_min_f32 %5, %5, %8\n v_min_f32 %6, %6, %8\n v_min_f32 %7, %7, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));)
        } else if (K == CND_VCCDEF) {
            REP4(asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                         "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "s"(vccinit));)
        } else if (K == CND_E64) {
            REP4(asm volatile("v_cndmask_b32_e64 %0, %0, %8, %9\n v_cndmask_b32_e64 %1, %1, %8, %9\n v_cndmask_b32_e64 %2, %2, %8, %9\n v_cndmask_b32_e64 %3, %3, %8, %9\n"
                         "v_cndmask_b32_e64 %4, %4, %8, %9\n v_cndmask_b32_e64 %5, %5, %8, %9\n v_cndmask_b32_e64 %6, %6, %8, %9\n v_cndmask_b32_e64 %7, %7, %8, %9"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "s"(vccinit));)
        } else if (K == CMP_CND) {
            REP4(asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %8, vcc\n"
                         "v_cmp_lt_f32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %8, vcc\n v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %8, vcc"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "vcc");)
        } else if (K == CND_ZERO) {
            REP4(asm volatile("v_cndmask_b32 %0, 0, %8, vcc\n v_cndmask_b32 %1, 0, %9, vcc\n v_cndmask_b32 %2, 0, %10, vcc\n v_cndmask_b32 %3, 0, %11, vcc\n"
                         "v_cndmask_b32 %4, 0, %12, vcc\n v_cndmask_b32 %5, 0, %13, vcc\n v_cndmask_b32 %6, 0, %14, vcc\n v_cndmask_b32 %7, 0, %15, vcc"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                         : "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(b4), "v"(b5), "v"(b6), "v"(b7), "s"(vccinit));)
        } else if (K == ADDF) {
            REP4(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                         "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(d));)
        } else if (K == SUBSGPR) {
            REP4(asm volatile("v_sub_f32 %0, %8, %0\n v_sub_f32 %1, %8, %1\n v_sub_f32 %2, %8, %2\n v_sub_f32 %3, %8, %3\n"
                         "v_sub_f32 %4, %8, %4\n v_sub_f32 %5, %8, %5\n v_sub_f32 %6, %8, %6\n v_sub_f32 %7, %8, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(seed));)
        } else if (K == ANDB) {
            REP4(asm volatile("v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n"
                         "v_and_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));)
        } else if (K == MADI24) {
            REP4(asm volatile("v_mad_i32_i24 %0, %0, 36, %8\n v_mad_i32_i24 %1, %1, 36, %8\n v_mad_i32_i24 %2, %2, 36, %8\n v_mad_i32_i24 %3, %3, 36, %8\n"
                         "v_mad_i32_i24 %4, %4, 36, %8\n v_mad_i32_i24 %5, %5, 36, %8\n v_mad_i32_i24 %6, %6, 36, %8\n v_mad_i32_i24 %7, %7, 36, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));)
        } else if (K == FMA_DEP2) {
            REP4(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n"
                         "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));)
        } else if (K == MIX_FWD) { // the shape of a two-entries-per-iteration forward body
            asm volatile("v_pk_fma_f32 %0, %0, %16, %17\n v_pk_fma_f32 %1, %1, %16, %17\n v_pk_fma_f32 %2, %2, %16, %17\n v_pk_fma_f32 %3, %3, %16, %17\n"
                         "v_pk_fma_f32 %4, %4, %16, %17\n v_pk_fma_f32 %5, %5, %16, %17\n v_exp_f32 %8, %8\n v_exp_f32 %9, %9\n"
                         "v_pk_fma_f32 %6, %6, %16, %17\n v_pk_fma_f32 %7, %7, %16, %17\n v_pk_fma_f32 %0, %0, %16, %17\n v_pk_fma_f32 %1, %1, %16, %17\n"
                         "v_pk_fma_f32 %2, %2, %16, %17\n v_pk_fma_f32 %3, %3, %16, %17\n"
                         "v_min_f32 %10, %10, %18\n v_min_f32 %11, %11, %18\n v_cmp_lt_f32 vcc, %12, %18\n v_cndmask_b32 %12, %12, %18, vcc\n"
                         "v_cmp_lt_f32 vcc, %13, %18\n v_cndmask_b32 %13, %13, %18, vcc\n v_cmp_lt_f32 vcc, %14, %18\n v_cndmask_b32 %14, %14, %18, vcc\n"
                         "v_cmp_lt_f32 vcc, %15, %18\n v_cndmask_b32 %15, %15, %18, vcc\n v_mul_f32 %10, %10, %18\n v_mul_f32 %11, %11, %18\n"
                         "v_fma_f32 %12, %12, %18, %19\n v_fma_f32 %13, %13, %18, %19\n v_fma_f32 %14, %14, %18, %19\n v_fma_f32 %15, %15, %18, %19\n"
                         "v_cndmask_b32 %10, %10, %18, vcc\n v_cndmask_b32 %11, %11, %18, vcc"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7),
                           "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                         : "v"(c2), "v"(d2), "v"(c), "v"(d) : "vcc");
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p4.y + p5.x + p5.y +
              p6.x + p6.y + p7.x + p7.y + b0 + b1 + b2 + b3 + b4 + b5 + b6 + b7;
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int K>
static void run(float* out, unsigned long long* cyc, unsigned long long* hcyc)
{
    const int wps_list[] = {1, 2, 4, 8};
    printf("%-38s", kname[K]);
    for (int wi = 0; wi < 4; wi++) {
        const int wps = wps_list[wi];
        const int blocks = 256 * wps; // 256 threads = 4 waves = one per SIMD; wps blocks per CU
        hipEvent_t a, b;
        (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        hipLaunchKernelGGL(bench<K>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.0f);
        (void)hipEventRecord(a);
        hipLaunchKernelGGL(bench<K>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.0f);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms;
        (void)hipEventElapsedTime(&ms, a, b);
        (void)hipMemcpy(hcyc, cyc, (size_t)blocks * 4 * 8, hipMemcpyDeviceToHost);
        double mean = 0;
        for (int i = 0; i < blocks * 4; i++) mean += (double)hcyc[i];
        mean /= blocks * 4;
        const double ninstr = 32.0 * ITER;
        // wall-clock rate per SIMD: wave-instructions per second per SIMD
        const double per_simd_ginstr = ninstr * wps / (ms * 1e-3) / 1e9;
        printf(" | w%d: %5.2f clk/instr/wave, %5.2f Ginstr/s/SIMD", wps, mean / ninstr, per_simd_ginstr);
    }
    printf("\n");
}

int main()
{
    float* out; unsigned long long *cyc, *hcyc;
    (void)hipMalloc(&out, (size_t)256 * 8 * 256 * 4);
    (void)hipMalloc(&cyc, (size_t)256 * 8 * 4 * 8);
    hcyc = (unsigned long long*)malloc((size_t)256 * 8 * 4 * 8);
    printf("clk/instr/wave uses __builtin_readcyclecounter (s_memtime, 100 MHz-domain or shader clock: compare rows, and use the wall-clock column)\n");
    run<FMA>(out, cyc, hcyc); run<MUL>(out, cyc, hcyc); run<PKFMA>(out, cyc, hcyc); run<PKMUL>(out, cyc, hcyc); run<PKADD>(out, cyc, hcyc);
    run<EXP>(out, cyc, hcyc); run<RCP>(out, cyc, hcyc); run<DPPADD_ROR>(out, cyc, hcyc); run<DPPADD_QUAD>(out, cyc, hcyc);
    run<DPPADD_BANK>(out, cyc, hcyc); run<CNDMASK>(out, cyc, hcyc); run<CMP>(out, cyc, hcyc); run<MINF>(out, cyc, hcyc); run<MIX_FWD>(out, cyc, hcyc);
    run<CND_VCCDEF>(out, cyc, hcyc); run<CND_E64>(out, cyc, hcyc); run<CMP_CND>(out, cyc, hcyc); run<CND_ZERO>(out, cyc, hcyc); run<ADDF>(out, cyc, hcyc);
    run<SUBSGPR>(out, cyc, hcyc); run<ANDB>(out, cyc, hcyc); run<MADI24>(out, cyc, hcyc); run<FMA_DEP2>(out, cyc, hcyc);
    return 0;
}
