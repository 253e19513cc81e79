// Does a load issued early really arrive while the wave does other things?  One workgroup of 512
// threads; waves 0-1 (112 lanes) issue two random 16-byte loads per lane from a big buffer, then
// the workgroup spends `delay` iterations of barriers + ALU work, then the issuing waves wait for
// the data.  Prints the issue->data-in-hand time and the time spent in the final wait.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }

__global__ void __launch_bounds__(512) k(const uint4 *buf, uint64_t n, int rounds, int delay, int other_traffic, const uint64_t *small,
                                         unsigned long long *out)
{
    __shared__ uint32_t sink[512];
    const uint32_t tid = threadIdx.x;
    unsigned long long t_total = 0, t_wait = 0;
    uint32_t acc = 0;
    for (int r = 0; r < rounds; r++) {
        const bool node = tid < 112;
        uint4 a0 = make_uint4(0, 0, 0, 0), a1 = a0;
        unsigned long long t0 = 0;
        if (node) {
            const uint64_t i0 = mix((uint64_t)r * 1000003 + tid * 2 + blockIdx.x * 7777777) % n;
            const uint64_t i1 = mix((uint64_t)r * 1000003 + tid * 2 + 1 + blockIdx.x * 7777777) % n;
            t0 = __builtin_amdgcn_s_memrealtime();
            a0 = buf[i0];
            a1 = buf[i1];
            asm volatile("" ::: "memory");
        }
        for (int d = 0; d < delay; d++) {  // ~0.1 us each: a barrier and some ALU
            uint32_t x = tid + d;
            for (int q = 0; q < 20; q++) x = x * 1664525u + 1013904223u;
            sink[tid] = x;
            if (other_traffic && tid >= 256 && tid < 256 + 112) acc += (uint32_t)small[(x >> 8) & 0xFFFF];  // another half loads from a small, hot buffer
            __syncthreads();
        }
        if (node) {
            const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // wait here
            const unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
            acc += a0.x + a1.y;
            asm volatile("" :: "v"(acc));
            t_total += t2 - t0;
            t_wait += t2 - t1;
        }
        __syncthreads();
    }
    sink[tid] = acc;
    if (tid == 0) { out[blockIdx.x * 2] = t_total; out[blockIdx.x * 2 + 1] = t_wait; }
}

int main()
{
    const uint64_t n = (8ull << 30) / 16;  // 8 GB
    uint4 *buf; uint64_t *small; unsigned long long *out;
    (void)hipMalloc(&buf, n * 16); (void)hipMemset(buf, 1, n * 16);
    (void)hipMalloc(&small, 65536 * 8); (void)hipMemset(small, 1, 65536 * 8);
    (void)hipMalloc(&out, 64);
    const int rounds = 2000;
    for (int other = 0; other < 2; other++)
        for (int delay : {0, 5, 10, 20, 40}) {
            for (int blocks : {1, 2}) {
                hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, buf, n, rounds, delay, other, small, out);
                (void)hipDeviceSynchronize();
                unsigned long long h[4];
                (void)hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
                printf("other=%d delay=%2d blocks=%d: issue->use %.2f us, of which final wait %.2f us\n", other, delay, blocks,
                       h[0] * 0.01 / rounds, h[1] * 0.01 / rounds);
            }
        }
    return 0;
}
