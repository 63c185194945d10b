// Dependency-chain-free field-multiplication kernels: the integer-VALU ceiling that the NTT,
// quotient and MSM kernels are priced against.  Each thread runs 4 independent Montgomery
// product chains so the issue stream is never latency-bound.
#include <algorithm>
#include <vector>

#include "common.hpp"

namespace plonk {

template <class F>
__global__ __launch_bounds__(256) void field_mul_bench_kernel(elem_t<F>* x, int iters) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    elem_t<F> a0 = ld(x + 4 * i), a1 = ld(x + 4 * i + 1), a2 = ld(x + 4 * i + 2), a3 = ld(x + 4 * i + 3);
    for (int k = 0; k < iters; ++k) {
        a0 = f_mul<F>(a0, a1);
        a1 = f_mul<F>(a1, a2);
        a2 = f_mul<F>(a2, a3);
        a3 = f_mul<F>(a3, a0);
    }
    st(x + 4 * i, a0); st(x + 4 * i + 1, a1); st(x + 4 * i + 2, a2); st(x + 4 * i + 3, a3);
}

template <class F>
static double run_bench(Ctx* ctx) {
    const int blocks = 256 * 8, threads = 256, iters = 256;
    const size_t n = (size_t)blocks * threads * 4;
    DevBuf<elem_t<F>> buf(n);
    std::vector<elem_t<F>> h(n);
    uint64_t s = 0x9E3779B97F4A7C15ull;
    for (size_t i = 0; i < n; ++i)
        for (int j = 0; j < F::N; ++j) {
            s = s * 6364136223846793005ull + 1442695040888963407ull;
            h[i].l[j] = (uint32_t)(s >> 33) & (j == F::N - 1 ? 0x0fffffffu : 0xffffffffu);
        }
    HIP_CHECK(hipMemcpy(buf.p, h.data(), n * sizeof(elem_t<F>), hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    HIP_CHECK(hipEventCreate(&e0));
    HIP_CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(field_mul_bench_kernel<F>, dim3(blocks), dim3(threads), 0, ctx->stream, buf.p, 8);
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        HIP_CHECK(hipEventRecord(e0, ctx->stream));
        hipLaunchKernelGGL(field_mul_bench_kernel<F>, dim3(blocks), dim3(threads), 0, ctx->stream, buf.p, iters);
        HIP_CHECK(hipEventRecord(e1, ctx->stream));
        HIP_CHECK(hipEventSynchronize(e1));
        float ms = 0;
        HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    const double muls = (double)blocks * threads * 4.0 * iters;
    return muls / (best * 1e-3);
}

double bench_field_mul(Ctx* ctx, uint32_t which) {
    return which == 0 ? run_bench<FrP>(ctx) : run_bench<FqP>(ctx);
}

}  // namespace plonk
