// Device-scope vs workgroup-scope returning atomics. MI355X has one L2 per XCD; an atomic whose scope does
// not reach beyond the XCD can be executed in that L2. Each XCD gets its own counter array here (a block
// learns its XCD from HW_REG_XCC_ID), so the narrow scope is also CORRECT: only one L2 ever sees an address.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__device__ uint32_t hash(uint32_t x){x^=x>>16;x*=0x7feb352d;x^=x>>15;x*=0x846ca68b;x^=x>>16;return x;}
__device__ uint32_t xcc_id(){ uint32_t v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xf; }
template<int SCOPE> __global__ void k(uint32_t* c, uint32_t* o, int nrec, int per){
  int i=blockIdx.x*blockDim.x+threadIdx.x; uint32_t x=xcc_id(); uint32_t* base=c+(size_t)x*nrec*16; uint32_t s=0;
  for(int k=0;k<per;k++){uint32_t t=hash(i*per+k)%nrec; s+=__hip_atomic_fetch_add(&base[(size_t)t*16],1u,__ATOMIC_RELAXED,SCOPE);} o[i]=s+x; }
__global__ void kx(uint32_t* o){ if(threadIdx.x==0) o[blockIdx.x]=xcc_id(); }
int main(){ uint32_t* c; uint32_t* o; const int nrec=3225*4; (void)hipMalloc(&c,(size_t)8*nrec*64); (void)hipMalloc(&o,8<<20);
 hipEvent_t a,b; (void)hipEventCreate(&a); (void)hipEventCreate(&b); float ms;
 hipLaunchKernelGGL(kx,dim3(16),dim3(64),0,0,o); uint32_t h[16]; (void)hipMemcpy(h,o,64,hipMemcpyDeviceToHost); printf("xcc of blocks 0..15:"); for(int i=0;i<16;i++) printf(" %u",h[i]); printf("\n");
 for(int rep=0;rep<2;rep++){
  (void)hipMemset(c,0,(size_t)8*nrec*64); (void)hipEventRecord(a); hipLaunchKernelGGL(k<__HIP_MEMORY_SCOPE_AGENT>,dim3(1000000/256),dim3(256),0,0,c,o,nrec,2); (void)hipEventRecord(b); (void)hipEventSynchronize(b); (void)hipEventElapsedTime(&ms,a,b); printf("agent scope     2M ret atomics: %.3f ms %.2f G/s\n",ms,2e6/ms/1e6);
  (void)hipMemset(c,0,(size_t)8*nrec*64); (void)hipEventRecord(a); hipLaunchKernelGGL(k<__HIP_MEMORY_SCOPE_WORKGROUP>,dim3(1000000/256),dim3(256),0,0,c,o,nrec,2); (void)hipEventRecord(b); (void)hipEventSynchronize(b); (void)hipEventElapsedTime(&ms,a,b); printf("workgroup scope 2M ret atomics: %.3f ms %.2f G/s\n",ms,2e6/ms/1e6);
  // check: the eight arrays together hold every increment
  static uint32_t hc[8*3225*4*16]; (void)hipMemcpy(hc,c,sizeof(hc),hipMemcpyDeviceToHost); unsigned long long tot=0; for(size_t i=0;i<sizeof(hc)/4;i+=16) tot+=hc[i]; printf("   sum of counters %llu (expect 2000000)\n",tot);
 }
 return 0; }
