#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
__device__ uint32_t hash(uint32_t x){x^=x>>16;x*=0x7feb352d;x^=x>>15;x*=0x846ca68b;x^=x>>16;return x;}
__global__ void k_u32(uint32_t* c,int nctr,int stride,int per){int i=blockIdx.x*blockDim.x+threadIdx.x; for(int k=0;k<per;k++){uint32_t t=hash(i*per+k)%nctr; atomicAdd(&c[(size_t)t*stride],1u);} }
__global__ void k_u32ret(uint32_t* c,uint32_t* o,int nctr,int stride,int per){int i=blockIdx.x*blockDim.x+threadIdx.x; uint32_t s=0; for(int k=0;k<per;k++){uint32_t t=hash(i*per+k)%nctr; s+=atomicAdd(&c[(size_t)t*stride],1u);} o[i]=s; }
// 9-lane float atomics: each wave does `per` atomics instrs with 9 active lanes to a random splat
__global__ void k_f9(float* acc,int nsplat,int per){int i=blockIdx.x*blockDim.x+threadIdx.x; int wave=i>>6, lane=i&63; for(int k=0;k<per;k++){uint32_t t=hash(wave*per+k)%nsplat; if(lane<9) unsafeAtomicAdd(&acc[(size_t)t*12+lane],1.0f);} }
__global__ void k_f1(float* acc,int nsplat,int per){int i=blockIdx.x*blockDim.x+threadIdx.x; for(int k=0;k<per;k++){uint32_t t=hash(i*per+k)%(nsplat*9); unsafeAtomicAdd(&acc[t],1.0f);} }
int main(){ uint32_t* c; float* acc; uint32_t* o; hipMalloc(&c,3225*64*4+1024); hipMalloc(&acc,(size_t)1000000*12*4); hipMalloc(&o,4<<20);
 hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b); float ms;
 for(int stride: {1,16,32}) for(int rep=0;rep<2;rep++){ hipMemset(c,0,3225*64*4); hipEventRecord(a); hipLaunchKernelGGL(k_u32,dim3(1000000/256),dim3(256),0,0,c,3225,stride,2); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms,a,b); printf("u32 noret 2M atomics over 3225 ctrs stride %d: %.3f ms -> %.2f G/s\n",stride,ms,2.0e6/ms/1e6);}
 for(int stride: {1,16}) for(int rep=0;rep<2;rep++){ hipMemset(c,0,3225*64*4); hipEventRecord(a); hipLaunchKernelGGL(k_u32ret,dim3(1000000/256),dim3(256),0,0,c,o,3225,stride,2); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms,a,b); printf("u32 ret   2M atomics over 3225 ctrs stride %d: %.3f ms -> %.2f G/s\n",stride,ms,2.0e6/ms/1e6);}
 for(int rep=0;rep<2;rep++){ hipMemset(acc,0,(size_t)1000000*48); hipEventRecord(a); hipLaunchKernelGGL(k_f9,dim3(4000000/4),dim3(256),0,0,acc,1000000,1); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms,a,b); printf("f32 9-lane: 4M wave-atomics (36M lane ops) over 1M splats: %.3f ms -> %.2f G waveops/s\n",ms,4.0e6/ms/1e6);}
 for(int rep=0;rep<2;rep++){ hipMemset(acc,0,(size_t)1000000*48); hipEventRecord(a); hipLaunchKernelGGL(k_f1,dim3(36000000/256/4),dim3(256),0,0,acc,1000000,4); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms,a,b); printf("f32 scattered 36M lane atomics: %.3f ms -> %.2f G/s\n",ms,36.0e6/ms/1e6);}
 return 0; }
