// microbenchmark: dependent-chain latencies that bound the BFS (not part of the product)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)
__device__ __forceinline__ uint64_t rt(){ return __builtin_amdgcn_s_memrealtime(); }
template<int MODE>
__global__ void chase(uint32_t* buf, uint32_t mask, int iters, uint64_t* out){
  if (threadIdx.x!=0) return;
  uint32_t p = 1;
  uint64_t t0 = rt();
  for(int i=0;i<iters;i++){
    uint32_t v;
    if (MODE==0) v = buf[p];
    else if (MODE==1) v = __hip_atomic_load(&buf[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (MODE==2) v = atomicCAS(&buf[p], 0xFFFFFFFFu, 0u);   // never matches: returns value
    else if (MODE==3) { uint32_t exp=0xFFFFFFFFu; __hip_atomic_compare_exchange_strong(&buf[p], &exp, 0u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); v=exp; }
    else if (MODE==4) v = __hip_atomic_load(&buf[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else { v = __hip_atomic_fetch_add(&buf[p], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    p = v & mask;
  }
  uint64_t t1 = rt();
  out[0] = t1-t0; out[1]=p;
}
__global__ void barr(int iters, uint64_t* out){
  uint64_t t0=rt();
  for(int i=0;i<iters;i++) __syncthreads();
  uint64_t t1=rt();
  if(threadIdx.x==0) out[0]=t1-t0;
}
int main(){
  uint64_t* d_out; CK(hipMalloc(&d_out,16));
  for (int big=0; big<3; big++){
    size_t n = big==0 ? (1u<<18) : big==1 ? (1u<<26) : (1ull<<32);   // 1 MB, 256 MB, 16 GB of u32
    uint32_t* buf; CK(hipMalloc(&buf, n*4));
    // fill with a pseudo-random permutation-ish chain: buf[i] = hash(i) & mask
    std::vector<uint32_t> h;
    size_t fill = n > (1u<<26) ? (1u<<26) : n;
    h.resize(fill);
    for(size_t r=0;r<n/fill;r++){
      for(size_t i=0;i<fill;i++){ uint64_t x=(i+r*fill)*0x9E3779B97F4A7C15ull; x^=x>>29; x*=0xBF58476D1CE4E5B9ull; x^=x>>32; h[i]=(uint32_t)x & (uint32_t)(n-1) & ~15u; }
      CK(hipMemcpy(buf+r*fill,h.data(),fill*4,hipMemcpyHostToDevice));
    }
    const char* names[]={"plain","agent-load(sc1)","agent-CAS","wg-CAS","wg-load","wg-fetch_add0"};
    for(int mode=0;mode<6;mode++){
      int iters=20000; uint64_t o[2];
      for(int rep=0;rep<2;rep++){
      switch(mode){
        case 0: hipLaunchKernelGGL(chase<0>,1,64,0,0,buf,(uint32_t)(n-1),iters,d_out); break;
        case 1: hipLaunchKernelGGL(chase<1>,1,64,0,0,buf,(uint32_t)(n-1),iters,d_out); break;
        case 2: hipLaunchKernelGGL(chase<2>,1,64,0,0,buf,(uint32_t)(n-1),iters,d_out); break;
        case 3: hipLaunchKernelGGL(chase<3>,1,64,0,0,buf,(uint32_t)(n-1),iters,d_out); break;
        case 4: hipLaunchKernelGGL(chase<4>,1,64,0,0,buf,(uint32_t)(n-1),iters,d_out); break;
        default: hipLaunchKernelGGL(chase<5>,1,64,0,0,buf,(uint32_t)(n-1),iters,d_out); break;
      }
      CK(hipDeviceSynchronize()); CK(hipMemcpy(o,d_out,16,hipMemcpyDeviceToHost));
      }
      printf("buf %6zu MB  %-16s %.1f ns/op\n", n*4>>20, names[mode], (double)o[0]*10.0/iters);
    }
    CK(hipFree(buf));
  }
  for (int th : {64,256,1024}) { uint64_t o[2]; hipLaunchKernelGGL(barr,1,th,0,0,10000,d_out); CK(hipDeviceSynchronize()); CK(hipMemcpy(o,d_out,16,hipMemcpyDeviceToHost)); printf("__syncthreads %d threads: %.1f ns\n", th, (double)o[0]*10.0/10000); }
  return 0;
}
