// Issue-rate microbenchmark of the integer / fp64 VALU instructions that big-integer modular
// multiplication on gfx950 can be built from.  Each kernel runs 8 independent dependency chains
// per lane so the stream is issue-bound.  Prints lane-operations per second and the equivalent
// cycles per wave64 instruction per SIMD at the measured kernel clock.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters, uint32_t seed, unsigned long long* clk) {
    uint32_t a[8], b[8];
    uint64_t w[8];
    double d[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 2654435761u + i * 40503u + seed; b[i] = a[i] ^ 0x9e3779b9u; w[i] = ((uint64_t)a[i] << 32) | b[i]; d[i] = (double)a[i] * 1e-9; }
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[i]) : "v"(a[i]), "v"(b[i]) : "vcc");
                if (OP == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 2) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 3) asm volatile("v_add_co_u32 %0, vcc, %0, %1\n\tv_addc_co_u32 %2, vcc, %2, %1, vcc" : "+v"(a[i]), "+v"(b[i]), "+v"(a[(i + 1) & 7]) : : "vcc");
                if (OP == 4) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(w[i]) : "v"(w[(i + 1) & 7]));
                if (OP == 5) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 6) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 7) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
                if (OP == 8) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 9) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 10) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 11) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a[i]) : "v"(b[i]) : "vcc");
                if (OP == 12) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_addc_co_u32 %3, vcc, 0, %3, vcc" : "+v"(w[i]), "+v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(b[i]), "v"(b[(i+1)&7]) : "vcc");
                if (OP == 13) asm volatile("v_mad_i32_i24 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 14) asm volatile("v_lshrrev_b64 %0, 29, %0" : "+v"(w[i]));
                if (OP == 15) asm volatile("v_alignbit_b32 %0, %0, %1, 29" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 16) asm volatile("v_and_b32 %0, 0x1fffffff, %0" : "+v"(a[i]));
                if (OP == 17) asm volatile("v_lshrrev_b32 %0, 29, %0" : "+v"(a[i]));
                if (OP == 18) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 19) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 20) asm volatile("v_bfe_u32 %0, %0, 3, 29" : "+v"(a[i]));
                if (OP == 21) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b[i]) : "vcc");
                if (OP == 22) asm volatile("v_mad_u64_u32 %0, vcc, %1, 8, %0" : "+v"(w[i]) : "v"(a[i]) : "vcc");
                if (OP == 23) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 24) asm volatile("v_ashrrev_i32 %0, 29, %0" : "+v"(a[i]));
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    uint32_t acc = 0;
    for (int i = 0; i < 8; ++i) acc ^= a[i] ^ b[i] ^ (uint32_t)w[i] ^ (uint32_t)(w[i] >> 32) ^ (uint32_t)d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int OP>
int run(const char* name, int per_iter_mult) {
    const int blocks = 256 * 8, threads = 256, iters = 2000;
    uint32_t* out; unsigned long long* clk;
    CHECK(hipMalloc(&out, blocks * threads * 4));
    CHECK(hipMalloc(&clk, 16));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, out, 10, 1u, clk);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, out, iters, 2u, clk);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[2]; CHECK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
    double insts = (double)blocks * threads / 64.0 * iters * 32.0 * per_iter_mult;  // wave-instructions
    double ghz = (double)h[0] / ((double)h[1] / 100e6) / 1e9;
    double per_simd_per_s = insts / (ms * 1e-3) / (256.0 * 4.0);
    printf("%-34s %8.3f ms  %7.2f T lane-op/s  clock %.2f GHz  %.2f cycles/wave-inst/SIMD\n", name, ms,
           insts * 64 / (ms * 1e-3) / 1e12, ghz, ghz * 1e9 / per_simd_per_s);
    hipFree(out); hipFree(clk);
    return 0;
}

int main() {
    run<0>("v_mad_u64_u32", 1);
    run<1>("v_mul_lo_u32", 1);
    run<2>("v_mul_hi_u32", 1);
    run<3>("v_add_co + v_addc_co (pair)", 2);
    run<4>("v_lshl_add_u64", 1);
    run<5>("v_mov_b32", 1);
    run<6>("v_mad_u32_u24", 1);
    run<13>("v_mad_i32_i24", 1);
    run<9>("v_mul_hi_u32_u24", 1);
    run<7>("v_fma_f64", 1);
    run<8>("v_add3_u32", 1);
    run<10>("v_add_u32", 1);
    run<11>("v_addc_co_u32", 1);
    run<12>("v_mad_u64_u32 + v_addc_co (pair)", 2);
    run<14>("v_lshrrev_b64", 1);
    run<15>("v_alignbit_b32", 1);
    run<16>("v_and_b32 (literal)", 1);
    run<17>("v_lshrrev_b32", 1);
    run<18>("v_sub_u32", 1);
    run<19>("v_and_or_b32", 1);
    run<20>("v_bfe_u32", 1);
    run<21>("v_cndmask_b32", 1);
    run<22>("v_mad_u64_u32 (inline constant)", 1);
    run<23>("v_lshl_add_u32", 1);
    run<24>("v_ashrrev_i32", 1);
    return 0;
}
