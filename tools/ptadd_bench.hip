// Development aid: latency / throughput of chained XYZZ point additions in the radix-2^29 form,
// with the products expanded in place (MulInline) or through the out-of-line routine (MulCall).
// usage: ptadd_bench            (prints a small table)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../plonk_amd/csrc/g1x.cuh"
using namespace plonk;

template <class M>
__global__ __launch_bounds__(256) void chain_kernel(const g1x_xyzz_t* in, g1x_xyzz_t* out, int iters) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    PtX acc = ld_ptx(in + gid), q = ld_ptx(in + gid + 1);
#pragma unroll 1
    for (int i = 0; i < iters; ++i) ptx_add<M>(acc, q);
    st_ptx(out + gid, acc);
}

template <class M>
__global__ __launch_bounds__(256) void madd_kernel(const g1x_xyzz_t* in, g1x_xyzz_t* out, int iters) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    PtX acc = ld_ptx(in + gid);
    AffX q;
    q.x = acc.y; q.y = acc.x;
#pragma unroll 1
    for (int i = 0; i < iters; ++i) ptx_madd(acc, q);
    st_ptx(out + gid, acc);
}

template <class K>
static float run(K kernel, int blocks, int threads, const g1x_xyzz_t* in, g1x_xyzz_t* out, int iters) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), 0, 0, in, out, iters);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), 0, 0, in, out, iters);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main() {
    const int maxl = 1024 * 256 * 4 + 1;
    std::vector<g1x_xyzz_t> h(maxl);
    uint32_t s = 12345;
    for (auto& p : h)
        for (fqx_slot* f : {&p.x, &p.y, &p.zz, &p.zzz})
            for (int k = 0; k < 16; ++k) { s = s * 1664525u + 1013904223u; f->w[k] = k < 14 ? (s >> 4) & ((1u << 28) - 1u) : 0u; }
    g1x_xyzz_t *in, *out;
    hipMalloc(&in, maxl * sizeof(g1x_xyzz_t));
    hipMalloc(&out, maxl * sizeof(g1x_xyzz_t));
    hipMemcpy(in, h.data(), maxl * sizeof(g1x_xyzz_t), hipMemcpyHostToDevice);
    const int iters = 64;
    printf("%-10s %8s %8s %12s %14s\n", "variant", "blocks", "threads", "us/add(lat)", "Gadd/s");
    for (int blocks : {1, 256, 1024, 2048, 4096}) {
        for (int threads : {64, 256}) {
            float t1 = run(chain_kernel<MulInline>, blocks, threads, in, out, iters);
            float t2 = run(chain_kernel<MulCall>, blocks, threads, in, out, iters);
            float t3 = run(madd_kernel<MulInline>, blocks, threads, in, out, iters);
            const double lanes = (double)blocks * threads;
            printf("%-10s %8d %8d %12.2f %14.3f\n", "inline", blocks, threads, t1 * 1e3 / iters, lanes * iters / (t1 * 1e-3) / 1e9);
            printf("%-10s %8d %8d %12.2f %14.3f\n", "call", blocks, threads, t2 * 1e3 / iters, lanes * iters / (t2 * 1e-3) / 1e9);
            printf("%-10s %8d %8d %12.2f %14.3f\n", "madd", blocks, threads, t3 * 1e3 / iters, lanes * iters / (t3 * 1e-3) / 1e9);
        }
    }
    return 0;
}
