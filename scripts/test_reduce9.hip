// standalone check of reduce9 (run on the GPU box): hipcc --offload-arch=gfx950 -O3 -I gsorb-slam_amd/csrc scripts/test_reduce9.hip -o /tmp/tr && /tmp/tr
#include "gsr_device.h"
#include <stdio.h>
#include <stdlib.h>
__global__ void k(const float* in, float* out, int* slots)
{
    const int lane = threadIdx.x;
    float v[9];
    for (int i = 0; i < 9; i++) v[i] = in[i * 64 + lane];
    out[lane] = reduce9(v, lane);
    slots[lane] = reduce9_slot_of(lane);
}
int main()
{
    float h[9 * 64], *d, *o; int* s;
    double ref[9] = {0};
    for (int i = 0; i < 9; i++) for (int l = 0; l < 64; l++) { h[i * 64 + l] = (float)((rand() % 2001) - 1000); ref[i] += h[i * 64 + l]; }
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, 64 * 4); hipMalloc(&s, 64 * 4);
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, s);
    float ho[64]; int hs[64];
    hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost); hipMemcpy(hs, s, sizeof(hs), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int kk = 0; kk < 9; kk++) {
        int l = reduce9_lane_of(kk);
        if (hs[l] != kk || ho[l] != (float)ref[kk]) { bad++; printf("value %d: lane %d slot %d got %g want %g\n", kk, l, hs[l], ho[l], ref[kk]); }
    }
    int owners = 0; for (int l = 0; l < 64; l++) owners += hs[l] >= 0;
    printf("reduce9 %s (owners=%d)\n", bad == 0 && owners == 9 ? "PASS" : "FAIL", owners);
    if (bad) { for (int l = 0; l < 64; l++) printf("%g ", ho[l]); printf("\n"); for (int kk = 0; kk < 9; kk++) printf("%g ", ref[kk]); printf("\n"); }
    return bad != 0;
}
