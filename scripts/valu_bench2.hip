// Second instruction-cost table for gfx950 (see valu_bench.hip): which operand kinds / opcodes fall into the
// half-rate class. Each kernel body = 32 instructions (the 8-instruction pattern x4), 2000 iterations,
// 4 waves per SIMD resident; the table prints wave-instructions per second per SIMD (wall clock).
//   hipcc --offload-arch=gfx950 -O3 -o build/valu_bench2 scripts/valu_bench2.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define ITER 2000
#define REP4(x) x x x x

#define REGS "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
#define INS "v"(c), "v"(d), "s"(sc), "s"(mask), "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(idx)
// operands: %0..%7 accumulators, %8 c (vgpr), %9 d (vgpr), %10 sc (sgpr float), %11 mask (sgpr pair), %12..%15 b0..b3, %16 idx

#define DEFK(NAME, P0, P1, P2, P3, P4, P5, P6, P7)                                                              \
    __global__ void __launch_bounds__(256) NAME(float* out, float seed)                                          \
    {                                                                                                            \
        float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        float b0 = a0 * 0.5f, b1 = a1 * 0.5f, b2 = a2 * 0.5f, b3 = a3 * 0.5f;                                     \
        const float c = 0.999f, d = 1e-6f;                                                                       \
        const float sc = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, seed * 0.25f))); \
        const unsigned lo = __builtin_amdgcn_readfirstlane(0x5555aaaau ^ (unsigned)(seed > 2.f));                \
        const unsigned long long mask = ((unsigned long long)lo << 32) | lo;                                     \
        int idx = (threadIdx.x * 4) ^ 64;                                                                        \
        _Pragma("unroll 1") for (int it = 0; it < ITER; it++) {                                                  \
            REP4(asm volatile(P0 "\n" P1 "\n" P2 "\n" P3 "\n" P4 "\n" P5 "\n" P6 "\n" P7 : REGS : INS : "vcc");)     \
        }                                                                                                        \
        out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0 + b1 + b2 + b3; \
    }

#define SAME8(NAME, F) DEFK(NAME, F(0), F(1), F(2), F(3), F(4), F(5), F(6), F(7))

#define F_ADD_V(i) "v_add_f32 %" #i ", %" #i ", %9"
#define F_ADD_S(i) "v_add_f32 %" #i ", %10, %" #i
#define F_ADD_LIT(i) "v_add_f32 %" #i ", 0x3f7d70a4, %" #i
#define F_ADD_INL(i) "v_add_f32 %" #i ", 1.0, %" #i
#define F_MUL_S(i) "v_mul_f32 %" #i ", %10, %" #i
#define F_FMA_S(i) "v_fma_f32 %" #i ", %" #i ", %10, %9"
#define F_FMAC(i) "v_fmac_f32 %" #i ", %8, %9"
#define F_SUB_V(i) "v_sub_f32 %" #i ", %" #i ", %9"
#define F_MIN_V(i) "v_min_f32 %" #i ", %" #i ", %8"
#define F_MAX_V(i) "v_max_f32 %" #i ", %" #i ", %9"
#define F_MED3(i) "v_med3_f32 %" #i ", %" #i ", %8, %9"
#define F_LSHL(i) "v_lshlrev_b32 %" #i ", 1, %" #i
#define F_ADDU(i) "v_add_u32 %" #i ", %" #i ", %16"
#define F_ADDU_S(i) "v_add_u32 %" #i ", %10, %" #i
#define F_AND_LIT(i) "v_and_b32 %" #i ", 0xff, %" #i
#define F_MOV(i) "v_mov_b32 %" #i ", %8"
#define F_MOVDPP(i) "v_mov_b32_dpp %" #i ", %8 row_ror:8 row_mask:0xf bank_mask:0xf"
#define F_CMP_E64(i) "v_cmp_lt_f32_e64 s[20:21], %" #i ", %8"
#define F_CMP_E32(i) "v_cmp_lt_f32 vcc, %" #i ", %8"
#define F_CND_E64(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, %11"
#define F_MADU24(i) "v_mad_u32_u24 %" #i ", %" #i ", 36, %16"
#define F_MULU24(i) "v_mul_u32_u24 %" #i ", %" #i ", 36"
#define F_CVT(i) "v_cvt_f32_i32 %" #i ", %" #i
#define F_BFE(i) "v_bfe_u32 %" #i ", %" #i ", 4, 8"
#define F_PERM(i) "v_perm_b32 %" #i ", %" #i ", %8, %9"
#define F_SWAP32(i) "v_permlane32_swap_b32 %" #i ", %12"
#define F_SWAP16(i) "v_permlane16_swap_b32 %" #i ", %12"
#define F_LOG(i) "v_log_f32 %" #i ", %" #i
#define F_SQRT(i) "v_sqrt_f32 %" #i ", %" #i
#define F_MULLEG(i) "v_mul_legacy_f32 %" #i ", %" #i ", %8"
#define F_FMA_MIX(i) "v_fma_f32 %" #i ", %" #i ", %8, %9 clamp"
#define F_ADD_NEG(i) "v_add_f32_e64 %" #i ", %" #i ", -%9"
#define F_MUL_ABS(i) "v_mul_f32_e64 %" #i ", |%" #i "|, %8"

SAME8(k_add_v, F_ADD_V) SAME8(k_add_s, F_ADD_S) SAME8(k_add_lit, F_ADD_LIT) SAME8(k_add_inl, F_ADD_INL) SAME8(k_mul_s, F_MUL_S)
SAME8(k_fma_s, F_FMA_S) SAME8(k_fmac, F_FMAC) SAME8(k_sub_v, F_SUB_V) SAME8(k_min_v, F_MIN_V) SAME8(k_max_v, F_MAX_V) SAME8(k_med3, F_MED3)
SAME8(k_lshl, F_LSHL) SAME8(k_addu, F_ADDU) SAME8(k_addu_s, F_ADDU_S) SAME8(k_and_lit, F_AND_LIT) SAME8(k_mov, F_MOV) SAME8(k_movdpp, F_MOVDPP)
SAME8(k_cmp_e64, F_CMP_E64) SAME8(k_cmp_e32, F_CMP_E32) SAME8(k_cnd_e64, F_CND_E64) SAME8(k_madu24, F_MADU24) SAME8(k_mulu24, F_MULU24)
SAME8(k_cvt, F_CVT) SAME8(k_bfe, F_BFE) SAME8(k_perm, F_PERM) SAME8(k_swap32, F_SWAP32) SAME8(k_swap16, F_SWAP16) SAME8(k_log, F_LOG) SAME8(k_sqrt, F_SQRT)
SAME8(k_fma_clamp, F_FMA_MIX) SAME8(k_add_neg, F_ADD_NEG) SAME8(k_mul_abs, F_MUL_ABS)
// patterns
DEFK(k_salu_vcc_cnd, "s_and_b64 vcc, %11, %11", "v_cndmask_b32 %0, 0, %12, vcc", "v_add_f32 %1, %1, %9", "v_add_f32 %2, %2, %9",
     "v_cndmask_b32 %3, 0, %13, vcc", "v_add_f32 %4, %4, %9", "v_add_f32 %5, %5, %9", "v_add_f32 %6, %6, %9")
DEFK(k_salu_s_cnd, "s_and_b64 s[20:21], %11, %11", "v_cndmask_b32_e64 %0, 0, %12, s[20:21]", "v_add_f32 %1, %1, %9", "v_add_f32 %2, %2, %9",
     "v_cndmask_b32_e64 %3, 0, %13, s[20:21]", "v_add_f32 %4, %4, %9", "v_add_f32 %5, %5, %9", "v_add_f32 %6, %6, %9")
DEFK(k_cmp_vcc_cnd2, "v_cmp_lt_f32 vcc, %7, %8", "v_cndmask_b32 %0, 0, %12, vcc", "v_add_f32 %1, %1, %9", "v_add_f32 %2, %2, %9",
     "v_cndmask_b32 %3, 0, %13, vcc", "v_add_f32 %4, %4, %9", "v_add_f32 %5, %5, %9", "v_add_f32 %6, %6, %9")
DEFK(k_adds_only7, "s_nop 0", "v_add_f32 %0, %0, %9", "v_add_f32 %1, %1, %9", "v_add_f32 %2, %2, %9",
     "v_add_f32 %3, %3, %9", "v_add_f32 %4, %4, %9", "v_add_f32 %5, %5, %9", "v_add_f32 %6, %6, %9")
DEFK(k_ds_read_b128, "ds_read_b128 v[40:43], %16", "v_add_f32 %0, %0, %9", "v_add_f32 %1, %1, %9", "v_add_f32 %2, %2, %9",
     "ds_read_b128 v[44:47], %16 offset:1024", "v_add_f32 %4, %4, %9", "v_add_f32 %5, %5, %9", "s_waitcnt lgkmcnt(0)")
DEFK(k_ds_rmw, "ds_read_b32 v40, %16", "v_add_f32 %0, %0, %9", "v_add_f32 %1, %1, %9", "s_waitcnt lgkmcnt(0)",
     "v_add_f32 v40, v40, %9", "ds_write_b32 %16, v40", "v_add_f32 %5, %5, %9", "v_add_f32 %6, %6, %9")
DEFK(k_ds_add_f32, "ds_add_f32 %16, %12", "v_add_f32 %0, %0, %9", "v_add_f32 %1, %1, %9", "v_add_f32 %2, %2, %9",
     "v_add_f32 %3, %3, %9", "v_add_f32 %4, %4, %9", "v_add_f32 %5, %5, %9", "v_add_f32 %6, %6, %9")
DEFK(k_ds_add_u32, "ds_add_u32 %16, %16", "v_add_f32 %0, %0, %9", "v_add_f32 %1, %1, %9", "v_add_f32 %2, %2, %9",
     "v_add_f32 %3, %3, %9", "v_add_f32 %4, %4, %9", "v_add_f32 %5, %5, %9", "v_add_f32 %6, %6, %9")
DEFK(k_ds_add_rtn_u32, "ds_add_rtn_u32 v40, %16, %16", "v_add_f32 %0, %0, %9", "v_add_f32 %1, %1, %9", "v_add_f32 %2, %2, %9",
     "v_add_f32 %3, %3, %9", "v_add_f32 %4, %4, %9", "v_add_f32 %5, %5, %9", "s_waitcnt lgkmcnt(0)")
DEFK(k_ds_add_rtn_u32_x4, "ds_add_rtn_u32 v40, %16, %16", "ds_add_rtn_u32 v41, %16, %16 offset:1024", "ds_add_rtn_u32 v42, %16, %16 offset:2048", "ds_add_rtn_u32 v43, %16, %16 offset:3072",
     "v_add_f32 %3, %3, %9", "v_add_f32 %4, %4, %9", "v_add_f32 %5, %5, %9", "s_waitcnt lgkmcnt(0)")
DEFK(k_ds_swizzle, "ds_swizzle_b32 v40, %12 offset:swizzle(SWAP,8)", "v_add_f32 %0, %0, %9", "v_add_f32 %1, %1, %9", "v_add_f32 %2, %2, %9",
     "ds_swizzle_b32 v41, %13 offset:swizzle(SWAP,4)", "v_add_f32 %4, %4, %9", "v_add_f32 %5, %5, %9", "s_waitcnt lgkmcnt(0)")
DEFK(k_ds_bpermute, "ds_bpermute_b32 v40, %16, %12", "v_add_f32 %0, %0, %9", "v_add_f32 %1, %1, %9", "v_add_f32 %2, %2, %9",
     "ds_bpermute_b32 v41, %16, %13", "v_add_f32 %4, %4, %9", "v_add_f32 %5, %5, %9", "s_waitcnt lgkmcnt(0)")
DEFK(k_mfma_16x16x4, "v_mfma_f32_16x16x4_f32 v[48:51], %12, %13, v[48:51]", "v_add_f32 %0, %0, %9", "v_add_f32 %1, %1, %9", "v_add_f32 %2, %2, %9",
     "v_add_f32 %3, %3, %9", "v_add_f32 %4, %4, %9", "v_add_f32 %5, %5, %9", "v_add_f32 %6, %6, %9")
DEFK(k_mfma_only, "v_mfma_f32_16x16x4_f32 v[48:51], %12, %13, v[48:51]", "v_mfma_f32_16x16x4_f32 v[52:55], %12, %13, v[52:55]",
     "v_mfma_f32_16x16x4_f32 v[56:59], %12, %13, v[56:59]", "v_mfma_f32_16x16x4_f32 v[60:63], %12, %13, v[60:63]",
     "v_mfma_f32_16x16x4_f32 v[64:67], %12, %13, v[64:67]", "v_mfma_f32_16x16x4_f32 v[68:71], %12, %13, v[68:71]",
     "v_mfma_f32_16x16x4_f32 v[72:75], %12, %13, v[72:75]", "v_mfma_f32_16x16x4_f32 v[76:79], %12, %13, v[76:79]")
DEFK(k_mfma_4x4, "v_mfma_f32_4x4x1_16b_f32 v[48:51], %12, %13, v[48:51]", "v_add_f32 %0, %0, %9", "v_add_f32 %1, %1, %9", "v_add_f32 %2, %2, %9",
     "v_add_f32 %3, %3, %9", "v_add_f32 %4, %4, %9", "v_add_f32 %5, %5, %9", "v_add_f32 %6, %6, %9")

typedef void (*kern_t)(float*, float);
struct Entry { const char* name; kern_t k; };
#define E(n, k) {n, k}

int main()
{
    float* out;
    (void)hipMalloc(&out, (size_t)256 * 8 * 256 * 4);
    Entry tab[] = {
        E("v_add_f32 v,v,v", k_add_v), E("v_add_f32 v,SGPR,v", k_add_s), E("v_add_f32 v,literal,v", k_add_lit), E("v_add_f32 v,1.0,v", k_add_inl),
        E("v_mul_f32 v,SGPR,v", k_mul_s), E("v_fma_f32 v,v,SGPR,v", k_fma_s), E("v_fmac_f32", k_fmac), E("v_sub_f32 v,v,v", k_sub_v),
        E("v_add_f32_e64 v,v,-v (neg mod)", k_add_neg), E("v_mul_f32_e64 v,|v|,v (abs mod)", k_mul_abs), E("v_fma_f32 clamp", k_fma_clamp),
        E("v_min_f32 v,v,v", k_min_v), E("v_max_f32 v,v,v", k_max_v), E("v_med3_f32", k_med3),
        E("v_lshlrev_b32", k_lshl), E("v_add_u32 v,v,v", k_addu), E("v_add_u32 v,SGPR,v", k_addu_s), E("v_and_b32 v,0xff,v", k_and_lit), E("v_mov_b32", k_mov),
        E("v_mov_b32_dpp", k_movdpp), E("v_cmp_lt_f32_e64 sgpr", k_cmp_e64), E("v_cmp_lt_f32 vcc", k_cmp_e32), E("v_cndmask_b32_e64 sgpr", k_cnd_e64),
        E("v_mad_u32_u24", k_madu24), E("v_mul_u32_u24", k_mulu24), E("v_cvt_f32_i32", k_cvt), E("v_bfe_u32", k_bfe), E("v_perm_b32", k_perm),
        E("v_permlane32_swap_b32", k_swap32), E("v_permlane16_swap_b32", k_swap16), E("v_log_f32", k_log), E("v_sqrt_f32", k_sqrt),
        E("baseline: 7 v_add + s_nop per 8", k_adds_only7),
        E("s_and vcc; cnd vcc; 2 add; cnd vcc; 3 add", k_salu_vcc_cnd), E("s_and s[]; cnd_e64; 2 add; cnd_e64; 3 add", k_salu_s_cnd),
        E("v_cmp vcc; cnd vcc; 2 add; cnd vcc; 3 add", k_cmp_vcc_cnd2),
        E("2 ds_read_b128 + 5 add + wait", k_ds_read_b128), E("ds_read_b32;2 add;wait;add;ds_write;2 add", k_ds_rmw), E("ds_add_f32 + 7 add", k_ds_add_f32), E("ds_add_u32 + 7 add", k_ds_add_u32), E("ds_add_rtn_u32 + 6 add + wait", k_ds_add_rtn_u32),
        E("4 ds_add_rtn_u32 + 3 add + wait", k_ds_add_rtn_u32_x4),
        E("2 ds_swizzle + 5 add + wait", k_ds_swizzle), E("2 ds_bpermute + 5 add + wait", k_ds_bpermute),
        E("1 mfma16x16x4f32 + 7 add", k_mfma_16x16x4), E("8 mfma16x16x4f32", k_mfma_only), E("1 mfma4x4x1_16b + 7 add", k_mfma_4x4),
    };
    printf("%-46s %s\n", "pattern (8 slots, x4 per iteration)", "G wave-instr/s/SIMD at 1,2,4,8 waves/SIMD  [ns per 8-slot pattern at 4 waves/SIMD]");
    for (auto& e : tab) {
        printf("%-46s", e.name);
        const int wl[] = {1, 2, 4, 8};
        for (int wi = 0; wi < 4; wi++) {
            const int wps = wl[wi], blocks = 256 * wps;
            hipEvent_t a, b;
            (void)hipEventCreate(&a); (void)hipEventCreate(&b);
            hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
            (void)hipEventRecord(a);
            hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
            (void)hipEventRecord(b);
            (void)hipEventSynchronize(b);
            float ms;
            (void)hipEventElapsedTime(&ms, a, b);
            const double g = 32.0 * ITER * wps / (ms * 1e-3) / 1e9;
            printf(" %5.2f", g);
            if (wps == 4) printf(" [%5.1f]", ms * 1e6 / (4.0 * ITER * wps));
        }
        printf("\n");
    }
    return 0;
}
