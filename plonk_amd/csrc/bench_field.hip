// Dependency-chain-free field-multiplication kernels: the integer-VALU ceiling that the NTT,
// quotient and MSM kernels are priced against.  Each thread runs 4 independent Montgomery
// product chains so the issue stream is never latency-bound.  which = 0 / 1: Fr / Fq in the
// radix-2^29 form the NTT and MSM kernels multiply in; which = 2 / 3: Fr / Fq in the saturated
// 32-bit-limb form.
#include <algorithm>
#include <vector>

#include "common.hpp"
#include "fieldx.cuh"

namespace plonk {

template <class F>
struct SatMul {
    typedef elem_t<F> elem;
    static constexpr int WORDS = F::N;
    static constexpr uint32_t TOP_MASK = 0x0fffffffu, MASK = 0xffffffffu;
    static __device__ __forceinline__ elem mul(const elem& a, const elem& b) { return f_mul<F>(a, b); }
};
template <class X>
struct LazyMul {
    typedef xe_t<X> elem;
    static constexpr int WORDS = X::L;
    static constexpr uint32_t TOP_MASK = 0x003fffffu, MASK = XMASK;
    static __device__ __forceinline__ elem mul(const elem& a, const elem& b) { return x_mul<X>(a, b); }
};

// ---- variants measured against x_mul and not used by the kernels: none of them beats it (one Karatsuba level saves
// multiply-adds but pays more in the 64-bit additions that recombine the halves) ----
// One Karatsuba level on the product half: the 14 x 14 limb product as three 7 x 7 products (147 multiply-adds
// instead of 196), recombined with 64-bit column additions; the Montgomery reduction half is x_mul's.
template <class X>
__device__ __forceinline__ xe_t<X> x_mul_karatsuba(const xe_t<X>& a, const xe_t<X>& b) {
    constexpr int L = X::L, H = L / 2;
    static_assert(L % 2 == 0, "even limb count");
    uint64_t lo[2 * H - 1], hi[2 * H - 1], mid[2 * H - 1];
#pragma unroll
    for (int k = 0; k < 2 * H - 1; ++k) lo[k] = hi[k] = mid[k] = 0;
    uint32_t sa[H], sb[H];
#pragma unroll
    for (int i = 0; i < H; ++i) {
        sa[i] = a.l[i] + a.l[i + H];   // below 2^30: seven products of 2^60 fit a 64-bit column
        sb[i] = b.l[i] + b.l[i + H];
    }
#pragma unroll
    for (int i = 0; i < H; ++i)
#pragma unroll
        for (int j = 0; j < H; ++j) {
            lo[i + j] += (uint64_t)a.l[i] * b.l[j];
            hi[i + j] += (uint64_t)a.l[i + H] * b.l[j + H];
            mid[i + j] += (uint64_t)sa[i] * sb[j];
        }
    uint64_t c[2 * L];
#pragma unroll
    for (int k = 0; k < 2 * L; ++k) c[k] = 0;
#pragma unroll
    for (int k = 0; k < 2 * H - 1; ++k) {
        c[k] += lo[k];
        c[k + L] += hi[k];
        c[k + H] += mid[k] - lo[k] - hi[k];
    }
#pragma unroll
    for (int i = 0; i < L; ++i) {
        const uint32_t m = ((uint32_t)c[i] * X::INV) & XMASK;
#pragma unroll
        for (int j = 0; j < L; ++j) c[i + j] += (uint64_t)m * X::p(j);
        c[i + 1] += c[i] >> 29;
    }
    xe_t<X> r;
#pragma unroll
    for (int k = 0; k < L - 1; ++k) {
        r.l[k] = (uint32_t)c[L + k] & XMASK;
        c[L + k + 1] += c[L + k] >> 29;
    }
    r.l[L - 1] = (uint32_t)c[2 * L - 1];
    return r;
}
template <class X>
struct KaratsubaMul {
    typedef xe_t<X> elem;
    static constexpr int WORDS = X::L;
    static constexpr uint32_t TOP_MASK = 0x003fffffu, MASK = XMASK;
    static __device__ __forceinline__ elem mul(const elem& a, const elem& b) { return x_mul_karatsuba<X>(a, b); }
};
// squarings only (x_sqr: 105 + 196 multiply-adds for Fq)
template <class X>
struct LazySqr {
    typedef xe_t<X> elem;
    static constexpr int WORDS = X::L;
    static constexpr uint32_t TOP_MASK = 0x003fffffu, MASK = XMASK;
    static __device__ __forceinline__ elem mul(const elem& a, const elem& b) { return x_sqr<X>(x_add_lazy<X>(a, b)); }
};
// the mix of a mixed addition: 7 products, 2 squarings and one two-product reduction per 10 products
template <class X>
struct MaddMix {
    typedef xe_t<X> elem;
    static constexpr int WORDS = X::L;
    static constexpr uint32_t TOP_MASK = 0x003fffffu, MASK = XMASK;
};

template <class X>
__global__ __launch_bounds__(256) void madd_mix_bench_kernel(xe_t<X>* x, int iters) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    xe_t<X> a0 = x[4 * i], a1 = x[4 * i + 1], a2 = x[4 * i + 2], a3 = x[4 * i + 3];
    for (int k = 0; k < iters; ++k) {   // 10 products per iteration in the proportions of ptx_madd
        a0 = x_mul<X>(a0, a1);
        a1 = x_mul<X>(a1, a2);
        a2 = x_sqr<X>(a2);
        a3 = x_mul<X>(a3, a0);
        a0 = x_mul<X>(a0, a2);
        a1 = x_sqr<X>(a1);
        a2 = x_mul2<X>(a2, a3, a0, a1);
        a3 = x_mul<X>(a3, a1);
        a0 = x_mul<X>(a0, a3);
    }
    x[4 * i] = a0; x[4 * i + 1] = a1; x[4 * i + 2] = a2; x[4 * i + 3] = a3;
}

template <class M>
__global__ __launch_bounds__(256) void field_mul_bench_kernel(typename M::elem* x, int iters) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    typename M::elem a0 = x[4 * i], a1 = x[4 * i + 1], a2 = x[4 * i + 2], a3 = x[4 * i + 3];
    for (int k = 0; k < iters; ++k) {
        a0 = M::mul(a0, a1);
        a1 = M::mul(a1, a2);
        a2 = M::mul(a2, a3);
        a3 = M::mul(a3, a0);
    }
    x[4 * i] = a0; x[4 * i + 1] = a1; x[4 * i + 2] = a2; x[4 * i + 3] = a3;
}

// `lds_bytes` of dynamic LDS per workgroup (unused by the kernel) bound the workgroups a CU holds: 80 KiB leaves two
// waves per SIMD, 160 KiB one -- the product rate at the occupancy of a register-heavy kernel.
template <class M>
static double run_bench(Ctx* ctx, size_t lds_bytes = 0) {
    typedef typename M::elem elem;
    if (lds_bytes)
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(field_mul_bench_kernel<M>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)lds_bytes));
    const int blocks = 256 * 8, threads = 256, iters = 256;
    const size_t n = (size_t)blocks * threads * 4;
    DevBuf<elem> buf(n);
    std::vector<elem> h(n);
    uint64_t s = 0x9E3779B97F4A7C15ull;
    for (size_t i = 0; i < n; ++i)
        for (int j = 0; j < M::WORDS; ++j) {
            s = s * 6364136223846793005ull + 1442695040888963407ull;
            h[i].l[j] = (uint32_t)(s >> 33) & (j == M::WORDS - 1 ? M::TOP_MASK : M::MASK);
        }
    HIP_CHECK(hipMemcpy(buf.p, h.data(), n * sizeof(elem), hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    HIP_CHECK(hipEventCreate(&e0));
    HIP_CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(field_mul_bench_kernel<M>, dim3(blocks), dim3(threads), lds_bytes, ctx->stream, buf.p, 8);
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        HIP_CHECK(hipEventRecord(e0, ctx->stream));
        hipLaunchKernelGGL(field_mul_bench_kernel<M>, dim3(blocks), dim3(threads), lds_bytes, ctx->stream, buf.p, iters);
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

// Raw issue rate of v_mad_u64_u32 (the instruction the 29-bit-limb products are made of): 8
// independent accumulators per lane, lane-operations per second.
__global__ __launch_bounds__(256) void mad_rate_kernel(uint32_t* out, int iters, uint32_t seed) {
    uint32_t a[8], b[8];
    uint64_t w[8];
    for (int i = 0; i < 8; ++i) {
        a[i] = threadIdx.x * 2654435761u + i * 40503u + seed;
        b[i] = a[i] ^ 0x9e3779b9u;
        w[i] = ((uint64_t)a[i] << 32) | b[i];
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[i]) : "v"(a[i]), "v"(b[i]) : "vcc");
        }
    }
    uint32_t acc = 0;
    for (int i = 0; i < 8; ++i) acc ^= (uint32_t)w[i] ^ (uint32_t)(w[i] >> 32);
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

static double run_mad_rate(Ctx* ctx) {
    const int blocks = 256 * 8, threads = 256, iters = 2000;
    DevBuf<uint32_t> out((size_t)blocks * threads);
    hipEvent_t e0, e1;
    HIP_CHECK(hipEventCreate(&e0));
    HIP_CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(mad_rate_kernel, dim3(blocks), dim3(threads), 0, ctx->stream, out.p, 10, 1u);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        HIP_CHECK(hipEventRecord(e0, ctx->stream));
        hipLaunchKernelGGL(mad_rate_kernel, dim3(blocks), dim3(threads), 0, ctx->stream, out.p, iters, 2u + rep);
        HIP_CHECK(hipEventRecord(e1, ctx->stream));
        HIP_CHECK(hipEventSynchronize(e1));
        float ms = 0;
        HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return (double)blocks * threads * iters * 32.0 / (best * 1e-3);
}

// products per second when they come in the proportions of one mixed addition (10 per iteration)
template <class X>
static double run_madd_mix(Ctx* ctx) {
    typedef xe_t<X> elem;
    const int blocks = 256 * 8, threads = 256, iters = 64;
    const size_t n = (size_t)blocks * threads * 4;
    DevBuf<elem> buf(n);
    std::vector<elem> h(n);
    uint64_t s = 0x9E3779B97F4A7C15ull;
    for (size_t i = 0; i < n; ++i)
        for (int j = 0; j < X::L; ++j) {
            s = s * 6364136223846793005ull + 1442695040888963407ull;
            h[i].l[j] = (uint32_t)(s >> 33) & (j == X::L - 1 ? 0x003fffffu : XMASK);
        }
    HIP_CHECK(hipMemcpy(buf.p, h.data(), n * sizeof(elem), hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    HIP_CHECK(hipEventCreate(&e0));
    HIP_CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(madd_mix_bench_kernel<X>, dim3(blocks), dim3(threads), 0, ctx->stream, buf.p, 4);
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        HIP_CHECK(hipEventRecord(e0, ctx->stream));
        hipLaunchKernelGGL(madd_mix_bench_kernel<X>, dim3(blocks), dim3(threads), 0, ctx->stream, buf.p, iters);
        HIP_CHECK(hipEventRecord(e1, ctx->stream));
        HIP_CHECK(hipEventSynchronize(e1));
        float ms = 0;
        HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return (double)blocks * threads * 10.0 * iters / (best * 1e-3);
}

double bench_field_mul(Ctx* ctx, uint32_t which) {
    switch (which) {
        case 5: return run_bench<KaratsubaMul<FqX>>(ctx);
        case 6: return run_bench<LazySqr<FqX>>(ctx);
        case 7: return run_madd_mix<FqX>(ctx);

        case 8: return run_bench<LazyMul<FqX>>(ctx, 80 * 1024);    // two waves per SIMD
        case 9: return run_bench<LazyMul<FqX>>(ctx, 160 * 1024);   // one
        case 10: return run_bench<LazyMul<FrX>>(ctx, 80 * 1024);
        case 11: return run_bench<LazyMul<FrX>>(ctx, 160 * 1024);
        case 4: return run_mad_rate(ctx);
        case 0: return run_bench<LazyMul<FrX>>(ctx);
        case 1: return run_bench<LazyMul<FqX>>(ctx);
        case 2: return run_bench<SatMul<FrP>>(ctx);
        case 3: return run_bench<SatMul<FqP>>(ctx);
        default: throw Error(PLONK_E_INVALID, "bench: unknown field selector");
    }
}

}  // namespace plonk
