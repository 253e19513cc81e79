// LDS operation rates on one CU with random addresses over a 4096-entry table (the merge kernel's
// access pattern): non-returning add, returning add, CAS, 8-byte read, 4-byte read.
// One workgroup of 512 threads (8 waves) or two (16 waves on the CU is not guaranteed; see grid).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int OP>
__global__ void __launch_bounds__(512) k(int iters, unsigned long long *out, uint32_t *sink)
{
    __shared__ uint64_t key[4096];
    __shared__ uint32_t cnt[4096];
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < 4096; i += 512) { key[i] = i; cnt[i] = 0; }
    __syncthreads();
    uint32_t x = tid * 2654435761u + 12345u, acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) {
        x = x * 1664525u + 1013904223u;
        const uint32_t s = (x >> 12) & 4095u;
        if (OP == 0) atomicAdd(&cnt[s], 1u);
        if (OP == 1) acc += atomicAdd(&cnt[s], 1u);
        if (OP == 2) acc += (uint32_t)atomicCAS(reinterpret_cast<unsigned long long *>(&key[s]), ~0ull, (unsigned long long)tid);
        if (OP == 3) acc += (uint32_t)key[s];
        if (OP == 4) acc += cnt[s];
        if (OP == 5) { acc += (uint32_t)key[s]; atomicAdd(&cnt[s], 1u); acc += cnt[(s + 1) & 4095u]; }
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    sink[blockIdx.x * 512 + tid] = acc + cnt[tid];
    if (tid == 0) out[blockIdx.x] = t1 - t0;
}

int main()
{
    unsigned long long *out; uint32_t *sink;
    (void)hipMalloc(&out, 8 * 1024); (void)hipMalloc(&sink, 4 * 512 * 1024);
    const int iters = 20000;
    const char *names[] = {"ds_add (no return)", "ds_add_rtn", "ds_cmpswap_b64", "ds_read_b64", "ds_read_b32", "read64+add+read32"};
    for (int blocks : {1, 512}) {
#define RUN(OP) do { hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(512), 0, 0, iters, out, sink); (void)hipDeviceSynchronize(); \
        unsigned long long h; (void)hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost); \
        const double cyc = h * 10e-9 * 2.4e9; /* 100 MHz ticks -> 2.4 GHz cycles (approx.) */ \
        printf("blocks=%3d %-20s: %.1f cycles per wave-instruction per workgroup of 8 waves -> %.2f lanes/clk/workgroup\n", blocks, names[OP], cyc / iters / 8, 512.0 * iters / cyc); } while (0)
        RUN(0); RUN(1); RUN(2); RUN(3); RUN(4); RUN(5);
    }
    return 0;
}
