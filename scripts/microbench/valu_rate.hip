// Issue rate of the vector instructions the counting kernels are made of (they are bound by vector issue: DESIGN.md §3.1).
// One workgroup of 256 / 512 / 1024 threads = 1 / 2 / 4 waves per SIMD; every wave runs 32 x iters instruction slots on two
// independent register chains.  Output: s_memtime ticks per slot, per wave and per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP16(X) X X X X X X X X X X X X X X X X

template <int OP>
__global__ void __launch_bounds__(1024) k(int iters, unsigned long long *out, uint32_t *sink, uint32_t seed)
{
    __shared__ uint32_t lds[4096];
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < 4096; i += blockDim.x) lds[i] = i * 4;
    __syncthreads();
    uint32_t a = tid * 2654435761u + seed, b = a ^ 0x12345u, c = a + 77u, d = b + 99u;
    uint64_t p = ((uint64_t)a << 32) | b, q = ((uint64_t)c << 32) | d;
    uint32_t sh = (seed & 7u) + 1u, addr = (tid * 4u) & 255u;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        if (OP == 0) { REP16(asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(c) : "v"(d));) }
        if (OP == 1) { REP16(asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(p) : "v"(sh)); asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(q) : "v"(sh));) }
        if (OP == 2) { REP16(asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(p) : "v"(sh)); asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(q) : "v"(sh));) }
        if (OP == 3) { REP16(asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a) : "v"(b)); asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(c) : "v"(d));) }
        if (OP == 4) { REP16(asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a) : "v"(b)); asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(c) : "v"(d));) }
        if (OP == 5) { REP16(asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(p) : "v"(a), "v"(b) : "vcc"); asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q) : "v"(c), "v"(d) : "vcc");) }
        if (OP == 6) { REP16(asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(sh)); asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(c) : "v"(d), "v"(sh));) }
        if (OP == 7) { REP16(asm volatile("v_bfrev_b32 %0, %0" : "+v"(a)); asm volatile("v_bfrev_b32 %0, %0" : "+v"(c));) }
        if (OP == 8) { REP16(asm volatile("v_cmp_lt_u64 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc" : : "v"(p), "v"(q), "v"(a), "v"(b) : "vcc"); asm volatile("v_cmp_lt_u64 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc" : : "v"(q), "v"(p), "v"(c), "v"(d) : "vcc");) }
        if (OP == 9) { REP16(asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a) : "v"(b)); asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(c) : "v"(d));) }
        if (OP == 10) { REP16(asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(d)); asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(c) : "v"(d), "v"(b));) }
        if (OP == 11) { REP16(asm volatile("v_lshl_add_u64 %0, %0, 3, %1" : "+v"(p) : "v"(q)); asm volatile("v_lshl_add_u64 %0, %0, 2, %1" : "+v"(q) : "v"(p));) }
        if (OP == 12) { REP16(asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(a) : "v"(addr)); asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(c) : "v"(addr));) }
        if (OP == 13) { REP16(asm volatile("ds_bpermute_b32 %0, %1, %0" : "+v"(a) : "v"(addr)); asm volatile("ds_bpermute_b32 %0, %1, %0" : "+v"(c) : "v"(addr));) asm volatile("s_waitcnt lgkmcnt(0)"); }
        if (OP == 14) { REP16(asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(b)); asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(c) : "v"(d));) }
        if (OP == 15) { REP16(asm volatile("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)" : "+v"(addr)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(c) : "v"(d));) }
        if (OP == 16) { REP16(asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a) : "v"(b)); asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(c) : "v"(d), "v"(b));) }
        if (OP == 17) { REP16(asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(p)); asm volatile("v_lshrrev_b64 %0, 1, %0" : "+v"(q));) }
        if (OP == 18) { REP16(asm volatile("v_add_co_u32 %0, vcc, %0, %1\n v_addc_co_u32 %2, vcc, %2, %3, vcc" : "+v"(a), "+v"(c) : "v"(b), "v"(d) : "vcc"); asm volatile("v_sub_co_u32 %0, vcc, %0, %1\n v_subb_co_u32 %2, vcc, %2, %3, vcc" : "+v"(b), "+v"(d) : "v"(a), "v"(c) : "vcc");) }
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    sink[blockIdx.x * 1024 + tid] = a + b + c + d + (uint32_t)p + (uint32_t)(p >> 32) + (uint32_t)q + (uint32_t)(q >> 32) + addr;
    if (tid == 0) out[blockIdx.x] = t1 - t0;
}

int main()
{
    unsigned long long *out; uint32_t *sink;
    (void)hipMalloc(&out, 8 * 1024); (void)hipMalloc(&sink, 4 * 256 * 1024);
    const int iters = 2000;
    const char *names[] = {"v_add_u32", "v_lshlrev_b64 (reg shift)", "v_lshrrev_b64 (reg shift)", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u64_u32", "v_alignbit_b32", "v_bfrev_b32",
                           "v_cmp_lt_u64 + v_cndmask (2 instr)", "v_mul_u32_u24", "v_mad_u32_u24", "v_lshl_add_u64", "ds_bpermute + wait (latency)", "ds_bpermute x32 then wait", "v_mov_dpp row_shr",
                           "ds_read_b32 dependent + wait, and a v_add (2 instr)", "v_xor / v_and_or", "v_lshl/lshr_b64 by constant 1", "64-bit add/sub with carry (2 instr)"};
#define RUN(OP) do { printf("%-52s:", names[OP]); for (int th : {256, 512, 1024}) { hipLaunchKernelGGL(k<OP>, dim3(1), dim3(th), 0, 0, iters, out, sink, 3u); (void)hipDeviceSynchronize(); \
        hipLaunchKernelGGL(k<OP>, dim3(1), dim3(th), 0, 0, iters, out, sink, 5u); (void)hipDeviceSynchronize(); \
        unsigned long long h; (void)hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost); \
        printf("  %d waves/SIMD: %6.2f ticks per slot and wave, %5.2f per slot and SIMD;", th / 256, (double)h / iters / 32.0, (double)h / iters / 32.0 / (th / 256)); } printf("\n"); } while (0)
    RUN(0); RUN(1); RUN(2); RUN(3); RUN(4); RUN(5); RUN(6); RUN(7); RUN(8); RUN(9); RUN(10); RUN(11); RUN(12); RUN(13); RUN(14); RUN(15); RUN(16); RUN(17); RUN(18);
    return 0;
}
