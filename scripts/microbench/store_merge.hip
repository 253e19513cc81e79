// Do 16-byte stores that fill a 32-byte sector (a 128-byte line) a few microseconds apart reach HBM merged?  The level-1 kernel of the
// counting pipeline (k_sk1w_extract) writes 16-byte records into 512 streams per workgroup, a stream getting one record every few
// microseconds, and rocprofv3's WRITE_SIZE reads 2.36 x its payload.  Here: 256 workgroups x 16 waves, every wave stores 64 records
// into 64 of its workgroup's 512 streams per step (consecutive places in a stream, as the kernel does), then idles `gap` steps of
// arithmetic.  Variants: plain store, nontemporal, sc1, sc0 sc1; records of 16 or 32 bytes.  Run under
//   rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out -- ./store_merge
// and compare WRITE_SIZE (KB) with the payload printed here.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int MODE>
__global__ void __launch_bounds__(1024) k_store(uint4 *out, uint32_t cap, int steps, int gap, uint32_t *sink)
{
    __shared__ uint32_t fill[512];
    const uint32_t tid = threadIdx.x;
    if (tid < 512) fill[tid] = 0;
    __syncthreads();
    uint32_t x = (blockIdx.x * 1024u + tid) * 2654435761u + 17u, acc = 0;
    for (int s = 0; s < steps; s++) {
        x = x * 1664525u + 1013904223u;
        const uint32_t b = (x >> 9) & 511u;
        const uint32_t pos = atomicAdd(&fill[b], MODE == 4 ? 2u : 1u);
        if (pos + 1 < cap) {
            uint4 *p = out + ((uint64_t)blockIdx.x * 512u + b) * cap + pos;
            const uint4 v = make_uint4(x, s, tid, b);
            if (MODE == 0) *p = v;
            if (MODE == 1) { typedef uint32_t u4 __attribute__((ext_vector_type(4))); u4 w = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(w, reinterpret_cast<u4 *>(p)); }
            typedef uint32_t u4v __attribute__((ext_vector_type(4)));
            const u4v vv = {v.x, v.y, v.z, v.w};
            if (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(vv) : "memory");
            if (MODE == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(vv) : "memory");
            if (MODE == 4) { p[0] = v; p[1] = v; }  // a 32-byte record
        }
        for (int g = 0; g < gap; g++) { x = x * 1664525u + 1013904223u; acc += x >> 7; }
    }
    if (acc == 0x12345u) sink[0] = acc;
}

int main(int argc, char **argv)
{
    const int steps = 400, gap = argc > 1 ? atoi(argv[1]) : 1500;
    const uint32_t cap = 1024;  // records a stream (a wave adds ~ steps / 8 to each of its workgroup's 512)
    uint4 *out; uint32_t *sink;
    const size_t bytes = (size_t)256 * 512 * cap * sizeof(uint4);
    (void)hipMalloc(&out, bytes); (void)hipMalloc(&sink, 64);
    (void)hipMemset(out, 0, bytes);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const char *names[] = {"plain", "nontemporal", "sc1", "sc0 sc1", "32-byte records (two plain stores)"};
#define RUN(M) do { (void)hipEventRecord(e0); hipLaunchKernelGGL(k_store<M>, dim3(256), dim3(1024), 0, 0, out, cap, steps, gap, sink); (void)hipEventRecord(e1); \
        (void)hipDeviceSynchronize(); float ms; (void)hipEventElapsedTime(&ms, e0, e1); \
        printf("k_store<%d> %-36s: %.2f ms, payload %.1f MB (%d records of %d bytes), one record a stream every %.1f us\n", M, names[M], ms, \
               256.0 * 1024 * steps * (M == 4 ? 32 : 16) / 1e6, 256 * 1024 * steps, M == 4 ? 32 : 16, ms * 1e3 / (1024.0 * steps / 512)); } while (0)
    RUN(0); RUN(1); RUN(2); RUN(3); RUN(4);
    return 0;
}
