// row_newbcast (DPP ctrl 0x150 + n: lane n of every 16-lane row to the whole row) on gfx950: semantics check and
// what a blend-loop-shaped body costs when the per-entry operands come (a) from LDS, every lane of a row reading the
// same 48 bytes (three ds_read_b128), or (b) from the registers of the lane that holds the entry, through DPP operands.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o build/dpp_bcast_bench scripts/dpp_bcast_bench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
template <int I> __device__ __forceinline__ float bc(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x150 + I, 0xf, 0xf, true));
}
__global__ void k_check(const float* in, float* out)
{
    const float v = in[threadIdx.x];
    float* o = out + threadIdx.x * 16;
    o[0] = bc<0>(v); o[1] = bc<1>(v); o[2] = bc<2>(v); o[3] = bc<3>(v); o[4] = bc<4>(v); o[5] = bc<5>(v); o[6] = bc<6>(v); o[7] = bc<7>(v);
    o[8] = bc<8>(v); o[9] = bc<9>(v); o[10] = bc<10>(v); o[11] = bc<11>(v); o[12] = bc<12>(v); o[13] = bc<13>(v); o[14] = bc<14>(v); o[15] = bc<15>(v);
}
#define ITER 400
// body: the forward blend's arithmetic on one entry (dx, dy, power, exp2, alpha, test_T, colour accumulation)
#define BODY(AX, AY, CA, CB, CC, OP, R, G, B)                                               \
    {                                                                                       \
        const float dx = (AX) - pxf, dy = (AY) - pyf;                                       \
        const float t = fmaf((CA), dx, (CB) * dy);                                          \
        const float p2 = fmaf(t, dx, ((CC) * dy) * dy);                                     \
        const float al = fminf(0.99f, (OP) * __builtin_amdgcn_exp2f(p2));                   \
        const float tt = T * (1.f - al);                                                    \
        const bool u = p2 <= 0.f && al >= 0.0039f && tt >= 1e-4f;                           \
        const float w = u ? al * T : 0.f;                                                   \
        C0 = fmaf((R), w, C0); C1 = fmaf((G), w, C1); C2 = fmaf((B), w, C2);                \
        T = u ? tt : T;                                                                     \
    }
template <int WAVES>
__global__ void __launch_bounds__(64) k_lds(const float4* in, float* out)
{
    __shared__ float4 E0[64], E1[64], E2[64];
    __shared__ unsigned short LIST[4 * 16 * 8]; // byte offsets, like the blend kernels' per-patch lists
    __shared__ char pad[WAVES == 3 ? 10000 : 1]; // 3 waves/SIMD like the backward: LDS-limited
    const int lane = threadIdx.x, r = lane >> 4;
    E0[lane] = in[lane]; E1[lane] = in[64 + lane]; E2[lane] = in[128 + lane];
    for (int i = lane; i < 4 * 16 * 8; i += 64) LIST[i] = (unsigned short)(((i * 7) & 63) * 16);
    if (in[0].x == 12345.f) pad[lane] = 1;
    __syncthreads();
    const float pxf = (float)(lane & 3), pyf = (float)((lane >> 2) & 3);
    float T = 1.f, C0 = 0.f, C1 = 0.f, C2 = 0.f;
#pragma unroll 1
    for (int it = 0; it < ITER; it++) {
        const unsigned short* l16 = LIST + r * 128 + (it & 7) * 16;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const unsigned o = l16[i];
            const float4 A = *(const float4*)((const char*)E0 + o), Bq = *(const float4*)((const char*)E1 + o), Cq = *(const float4*)((const char*)E2 + o);
            BODY(A.x, A.y, A.z, A.w, Bq.x, Bq.y, Bq.z, Bq.w, Cq.x)
        }
    }
    out[blockIdx.x * 64 + lane] = T + C0 + C1 + C2 + (float)pad[0];
}
template <int WAVES>
__global__ void __launch_bounds__(64) k_dpp(const float4* in, float* out)
{
    __shared__ float4 E0[64], E1[64], E2[64];
    __shared__ char pad[WAVES == 3 ? 10000 : 1];
    const int lane = threadIdx.x;
    E0[lane] = in[lane]; E1[lane] = in[64 + lane]; E2[lane] = in[128 + lane];
    if (in[0].x == 12345.f) pad[lane] = 1;
    __syncthreads();
    const float pxf = (float)(lane & 3), pyf = (float)((lane >> 2) & 3);
    float T = 1.f, C0 = 0.f, C1 = 0.f, C2 = 0.f;
#pragma unroll 1
    for (int it = 0; it < ITER; it++) {
        const int e = (lane + it) & 63;
        const float4 A = E0[e], Bq = E1[e], Cq = E2[e]; // my entry: once per 16 iterations
#define IT(i) BODY(bc<i>(A.x), bc<i>(A.y), bc<i>(A.z), bc<i>(A.w), bc<i>(Bq.x), bc<i>(Bq.y), bc<i>(Bq.z), bc<i>(Bq.w), bc<i>(Cq.x))
        IT(0) IT(1) IT(2) IT(3) IT(4) IT(5) IT(6) IT(7) IT(8) IT(9) IT(10) IT(11) IT(12) IT(13) IT(14) IT(15)
    }
    out[blockIdx.x * 64 + lane] = T + C0 + C1 + C2 + (float)pad[0];
}
template <typename K> float run(K k, const float4* in, float* out, int blocks)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, in, out);
    hipEventRecord(a);
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, in, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / 5;
}
int main()
{
    std::vector<float> h(64), o(64 * 16);
    for (int i = 0; i < 64; i++) h[i] = 100.f + i;
    float *din, *dout;
    hipMalloc(&din, 4096 * 4); hipMalloc(&dout, 1 << 24);
    hipMemcpy(din, h.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, din, dout);
    hipMemcpy(o.data(), dout, 64 * 16 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; l++) for (int n = 0; n < 16; n++) bad += o[l * 16 + n] != 100.f + (l & ~15) + n;
    printf("row_newbcast semantics: %s (%d mismatches)\n", bad ? "WRONG" : "ok", bad);
    std::vector<float> e(192 * 4);
    for (int i = 0; i < 192 * 4; i++) e[i] = 0.01f * (i % 37) - 0.1f;
    hipMemcpy(din, e.data(), e.size() * 4, hipMemcpyHostToDevice);
    const int blocks = 1024 * 12; // 12 single-wave workgroups per SIMD
    const double iters = (double)blocks * ITER * 16;
    float t;
    t = run(k_lds<8>, (const float4*)din, dout, blocks); printf("LDS operands,  <=8 waves/SIMD: %.3f ms  %.2f ns per wave-iteration per SIMD\n", t, t * 1e6 / (iters / 1024));
    t = run(k_dpp<8>, (const float4*)din, dout, blocks); printf("DPP operands,  <=8 waves/SIMD: %.3f ms  %.2f ns per wave-iteration per SIMD\n", t, t * 1e6 / (iters / 1024));
    t = run(k_lds<3>, (const float4*)din, dout, blocks); printf("LDS operands,    3 waves/SIMD: %.3f ms  %.2f ns per wave-iteration per SIMD\n", t, t * 1e6 / (iters / 1024));
    t = run(k_dpp<3>, (const float4*)din, dout, blocks); printf("DPP operands,    3 waves/SIMD: %.3f ms  %.2f ns per wave-iteration per SIMD\n", t, t * 1e6 / (iters / 1024));
    return bad != 0;
}
