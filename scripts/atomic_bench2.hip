// Ceiling of the per-splat gradient accumulation: 9 consecutive floats per record, random records.
//   f9   : one record per wave instruction (9 active lanes)        — the 8x8-quad blend kernel
//   f36  : four records per wave instruction (36 active lanes)     — a 4x4-patch x 4-splat kernel
// for record strides of 12 floats (48 B, half the records straddle a 64-B line) and 16 floats.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__device__ uint32_t hash(uint32_t x){x^=x>>16;x*=0x7feb352d;x^=x>>15;x*=0x846ca68b;x^=x>>16;return x;}
__global__ void k_f9(float* acc,int nsplat,int stride,int per){int i=blockIdx.x*blockDim.x+threadIdx.x; int wave=i>>6, lane=i&63;
  for(int k=0;k<per;k++){uint32_t t=hash(wave*per+k)%nsplat; if(lane<9) unsafeAtomicAdd(&acc[(size_t)t*stride+lane],1.0f);} }
__global__ void k_f36(float* acc,int nsplat,int stride,int per){int i=blockIdx.x*blockDim.x+threadIdx.x; int wave=i>>6, lane=i&63; int grp=lane/9, sub=lane%9;
  for(int k=0;k<per;k++){uint32_t t=hash((wave*per+k)*4+grp)%nsplat; if(lane<36) unsafeAtomicAdd(&acc[(size_t)t*stride+sub],1.0f);} }
int main(){ float* acc; const int NS=1000000; hipMalloc(&acc,(size_t)NS*16*4);
 hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b); float ms;
 for(int stride: {12,16}) for(int rep=0;rep<2;rep++){ hipMemset(acc,0,(size_t)NS*64); hipEventRecord(a); hipLaunchKernelGGL(k_f9,dim3(8000000/4/8),dim3(256),0,0,acc,NS,stride,8); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms,a,b);
   printf("f9  stride %2d: 8M records in %.3f ms -> %.2f G records/s\n",stride,ms,8.0e6/ms/1e6);}
 for(int stride: {12,16}) for(int rep=0;rep<2;rep++){ hipMemset(acc,0,(size_t)NS*64); hipEventRecord(a); hipLaunchKernelGGL(k_f36,dim3(2000000/4/8),dim3(256),0,0,acc,NS,stride,8); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms,a,b);
   printf("f36 stride %2d: 8M records in %.3f ms -> %.2f G records/s\n",stride,ms,8.0e6/ms/1e6);}
 return 0; }
