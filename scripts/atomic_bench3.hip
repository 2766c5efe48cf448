// returning atomics: 32-bit vs 64-bit, on 64-byte records (one hot word per record)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__device__ uint32_t hash(uint32_t x){x^=x>>16;x*=0x7feb352d;x^=x>>15;x*=0x846ca68b;x^=x>>16;return x;}
template<typename T> __global__ void k_ret(T* c, T* o, int nrec, int per){int i=blockIdx.x*blockDim.x+threadIdx.x; T s=0; for(int k=0;k<per;k++){uint32_t t=hash(i*per+k)%nrec; s+=atomicAdd(&c[(size_t)t*(64/sizeof(T))],(T)1);} o[i]=s;}
template<typename T> __global__ void k_noret(T* c, int nrec, int per){int i=blockIdx.x*blockDim.x+threadIdx.x; for(int k=0;k<per;k++){uint32_t t=hash(i*per+k)%nrec; atomicAdd(&c[(size_t)t*(64/sizeof(T))],(T)1);} }
int main(){ void* c; void* o; (void)hipMalloc(&c,4096*64); (void)hipMalloc(&o,8<<20);
 hipEvent_t a,b; (void)hipEventCreate(&a); (void)hipEventCreate(&b); float ms;
 for(int nrec: {1634,3225}) for(int rep=0;rep<2;rep++){
  (void)hipMemset(c,0,4096*64); (void)hipEventRecord(a); hipLaunchKernelGGL(k_ret<uint32_t>,dim3(1000000/256),dim3(256),0,0,(uint32_t*)c,(uint32_t*)o,nrec,2); (void)hipEventRecord(b); (void)hipEventSynchronize(b); (void)hipEventElapsedTime(&ms,a,b); printf("u32 ret   2M over %d recs: %.3f ms %.2f G/s\n",nrec,ms,2e6/ms/1e6);
  (void)hipMemset(c,0,4096*64); (void)hipEventRecord(a); hipLaunchKernelGGL(k_ret<unsigned long long>,dim3(1000000/256),dim3(256),0,0,(unsigned long long*)c,(unsigned long long*)o,nrec,2); (void)hipEventRecord(b); (void)hipEventSynchronize(b); (void)hipEventElapsedTime(&ms,a,b); printf("u64 ret   2M over %d recs: %.3f ms %.2f G/s\n",nrec,ms,2e6/ms/1e6);
  (void)hipMemset(c,0,4096*64); (void)hipEventRecord(a); hipLaunchKernelGGL(k_noret<uint32_t>,dim3(1000000/256),dim3(256),0,0,(uint32_t*)c,nrec,2); (void)hipEventRecord(b); (void)hipEventSynchronize(b); (void)hipEventElapsedTime(&ms,a,b); printf("u32 noret 2M over %d recs: %.3f ms %.2f G/s\n",nrec,ms,2e6/ms/1e6);
  (void)hipMemset(c,0,4096*64); (void)hipEventRecord(a); hipLaunchKernelGGL(k_noret<unsigned long long>,dim3(1000000/256),dim3(256),0,0,(unsigned long long*)c,nrec,2); (void)hipEventRecord(b); (void)hipEventSynchronize(b); (void)hipEventElapsedTime(&ms,a,b); printf("u64 noret 2M over %d recs: %.3f ms %.2f G/s\n",nrec,ms,2e6/ms/1e6);
 }
 return 0; }
