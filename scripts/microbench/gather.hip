// microbenchmark: random 16-byte gathers (not part of the product)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)
__device__ __forceinline__ uint64_t mix(uint64_t x){ x^=x>>33; x*=0xff51afd7ed558ccdull; x^=x>>33; x*=0xc4ceb9fe1a85ec53ull; x^=x>>33; return x; }
template<int NL>
__global__ void gather(const uint4* buf, uint64_t mask, int rounds, uint64_t* out, uint32_t* sink){
  uint64_t t0 = __builtin_amdgcn_s_memrealtime();
  uint32_t acc=0;
  uint64_t gid = blockIdx.x*(uint64_t)blockDim.x+threadIdx.x;
  for(int r=0;r<rounds;r++){
    uint4 v[NL];
#pragma unroll
    for(int i=0;i<NL;i++){ uint64_t a = mix(gid*1315423911ull + r*NL + i + 12345) & mask; v[i] = buf[a]; }
#pragma unroll
    for(int i=0;i<NL;i++) acc += v[i].x ^ v[i].w;
  }
  uint64_t t1 = __builtin_amdgcn_s_memrealtime();
  if(threadIdx.x==0 && blockIdx.x==0){ out[0]=t1-t0; }
  if(acc==0x12345) sink[0]=acc;
}
int main(){
  uint64_t* d_out; uint32_t* sink; CK(hipMalloc(&d_out,16)); CK(hipMalloc(&sink,4));
  size_t sizes[3] = {64ull<<20, 2ull<<30, 16ull<<30};
  for(int si=0;si<3;si++){
    uint4* buf; CK(hipMalloc(&buf, sizes[si])); CK(hipMemset(buf, 1, sizes[si]));
    uint64_t mask = sizes[si]/16-1;
    int cfgs[][2] = {{1,64},{1,512},{256,256},{2048,256}};
    for(auto& c: cfgs){
      for(int nl : {1,6}){
        int rounds = 200; uint64_t o;
        for(int rep=0;rep<2;rep++){
          if(nl==1) hipLaunchKernelGGL(gather<1>,c[0],c[1],0,0,buf,mask,rounds,d_out,sink);
          else hipLaunchKernelGGL(gather<6>,c[0],c[1],0,0,buf,mask,rounds,d_out,sink);
          CK(hipDeviceSynchronize());
        }
        CK(hipMemcpy(&o,d_out,8,hipMemcpyDeviceToHost));
        double us = o*0.01; double loads=(double)c[0]*c[1]*nl*rounds;
        printf("buf %5zu MB grid %4d x %3d NL=%d: %.2f us/round (block0)  %.2f Gloads/s chip\n", sizes[si]>>20, c[0], c[1], nl, us/rounds, loads/us/1e3);
      }
    }
    CK(hipFree(buf));
  }
  return 0;
}
