// which way do the gfx950 wave-wide DPP shifts move data?  prints lane 5's and lane 63/0's view
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *out)
{
    const unsigned x = threadIdx.x;
    out[threadIdx.x] = __builtin_amdgcn_update_dpp(1000u, x, 0x130, 0xF, 0xF, false);        // wave_shl:1
    out[64 + threadIdx.x] = __builtin_amdgcn_update_dpp(1000u, x, 0x138, 0xF, 0xF, false);   // wave_shr:1
}
int main()
{
    unsigned *d, h[128];
    hipMalloc(&d, sizeof h);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("wave_shl:1  lane0<-%u lane5<-%u lane31<-%u lane32<-%u lane63<-%u\n", h[0], h[5], h[31], h[32], h[63]);
    printf("wave_shr:1  lane0<-%u lane5<-%u lane31<-%u lane32<-%u lane63<-%u\n", h[64], h[69], h[95], h[96], h[127]);
    return 0;
}
