// How many single-wave workgroups fit on a CU as a function of their LDS size? (The LDS allocation granule of gfx950 is
// not in the guides.) Every workgroup spins for a fixed time; n workgroups per CU take one spin if they all fit, two if not.
// hipcc --offload-arch=gfx950 -O2 -o build/lds_granule scripts/lds_granule.hip && build/lds_granule
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(64) spin(long long ticks, int* sink)
{
    extern __shared__ int lds[];
    lds[threadIdx.x] = threadIdx.x;
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (lds[threadIdx.x] == -1) *sink = 1;
}
int main()
{
    int* sink; hipMalloc(&sink, 4);
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const int cus = pr.multiProcessorCount;
    hipFuncSetAttribute((const void*)spin, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const long long ticks = 100 * 100; // wall_clock64 runs at 100 MHz: 100 us
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int sizes[] = {10240, 11264, 11520, 11776, 12288, 12544, 12800, 12816, 13056, 13072, 13312, 13344, 13824, 14080, 14336, 14848, 15360, 16384};
    printf("CUs %d; rows: LDS bytes per workgroup; columns: workgroups per CU launched -> kernel time in spins\n", cus);
    for (int s : sizes) {
        printf("%6d B:", s);
        for (int n = 8; n <= 16; n++) {
            hipLaunchKernelGGL(spin, dim3(cus * n), dim3(64), s, 0, ticks, sink);
            hipDeviceSynchronize();
            hipEventRecord(a);
            hipLaunchKernelGGL(spin, dim3(cus * n), dim3(64), s, 0, ticks, sink);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            printf(" %d:%.1f", n, ms / 0.1f);
        }
        printf("\n");
    }
    return 0;
}
