// Multi-scalar multiplication over BLS12-381 G1 for gfx950: the device side of
// G1Projective::msm (backend_new/src/curve.rs:169-173) and Commitment::commit (kzg.rs:231-245).
//
// Pippenger bucket method with signed c-bit digits:
//   1. digits   : scalar -> canonical form -> W = ceil(256/c) signed digits in [-2^(c-1), 2^(c-1)];
//                 per-(window, |digit|) histogram.
//   2. scan     : exclusive prefix sum of the histogram, one workgroup per window.
//   3. scatter  : point indices grouped by (window, |digit|); bit 31 carries the digit's sign.
//   4. buckets  : one lane per bucket sums its points with XYZZ mixed additions; buckets larger
//                 than BIG_BUCKET are summed by a whole workgroup instead (skewed scalars such as
//                 0/1/-1 selector vectors put most points of a window into one bucket).
//   5. reduce   : sum_b b * B_b per window: lanes take chunks of L buckets (running sums), lift
//                 their chunk by (t*L) * chunk_sum with double-and-add, then a tree over LDS.
//   6. the W window sums are combined (c doublings each) and normalised on the host: O(256)
//      point operations.
// Integer VALU work throughout; bytes moved are negligible next to the field multiplications.
#include <algorithm>
#include <vector>

#include "common.hpp"
#include "g1.cuh"
#include "msm.hpp"

namespace plonk {

static constexpr uint32_t BIG_BUCKET = 192;   // points; above this a bucket gets a workgroup
static constexpr uint32_t SIGN_BIT = 0x80000000u;

// ---- 1. digits ----
__global__ __launch_bounds__(256) void msm_digits_kernel(const fr_t* __restrict__ scalars, uint32_t n, uint32_t c,
                                                          uint32_t W, uint32_t nb, int32_t* __restrict__ digits,
                                                          uint32_t* __restrict__ counts) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fr_t s = f_from_mont<FrP>(ld(scalars + i));
    const uint32_t mask = (1u << c) - 1u, half = 1u << (c - 1);
    uint32_t carry = 0;
    for (uint32_t w = 0; w < W; ++w) {
        uint32_t raw = (s.l[0] & mask) + carry;
        int32_t d;
        if (raw > half) { d = (int32_t)raw - (int32_t)(1u << c); carry = 1; }
        else { d = (int32_t)raw; carry = 0; }
        digits[(size_t)w * n + i] = d;
        if (d != 0) atomicAdd(&counts[(size_t)w * nb + (uint32_t)(d < 0 ? -d : d)], 1u);
        // s >>= c
#pragma unroll
        for (int k = 0; k < 7; ++k) s.l[k] = (s.l[k] >> c) | (s.l[k + 1] << (32 - c));
        s.l[7] >>= c;
    }
}

// ---- 2. scan (one workgroup of 1024 lanes per window) ----
__global__ __launch_bounds__(1024) void msm_scan_kernel(const uint32_t* __restrict__ counts, uint32_t nb,
                                                         uint32_t* __restrict__ offsets, uint32_t* __restrict__ cursor) {
    __shared__ uint32_t part[1024];
    const uint32_t w = blockIdx.x, tid = threadIdx.x;
    const uint32_t per = (nb + 1023) / 1024;
    const uint32_t lo = min(nb, tid * per), hi = min(nb, lo + per);
    uint32_t sum = 0;
    for (uint32_t k = lo; k < hi; ++k) sum += counts[(size_t)w * nb + k];
    part[tid] = sum;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {
        uint32_t v = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    uint32_t run = part[tid] - sum;  // exclusive
    for (uint32_t k = lo; k < hi; ++k) {
        offsets[(size_t)w * nb + k] = run;
        cursor[(size_t)w * nb + k] = run;
        run += counts[(size_t)w * nb + k];
    }
}

// ---- 3. scatter ----
__global__ __launch_bounds__(256) void msm_scatter_kernel(const int32_t* __restrict__ digits, uint32_t n, uint32_t W,
                                                           uint32_t nb, uint32_t* __restrict__ cursor,
                                                           uint32_t* __restrict__ sorted) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (uint32_t w = 0; w < W; ++w) {
        const int32_t d = digits[(size_t)w * n + i];
        if (d == 0) continue;
        const uint32_t mag = (uint32_t)(d < 0 ? -d : d);
        const uint32_t pos = atomicAdd(&cursor[(size_t)w * nb + mag], 1u);
        sorted[(size_t)w * n + pos] = i | (d < 0 ? SIGN_BIT : 0u);
    }
}

__device__ __forceinline__ g1_affine_t load_signed_point(const g1_affine_t* points, uint32_t entry) {
    g1_affine_t p = ld_affine(points + (entry & ~SIGN_BIT));
    if (entry & SIGN_BIT) p.y = f_neg<FqP>(p.y);  // (0,0) stays (0,0)
    return p;
}

// ---- 4a. small buckets: one lane per bucket ----
__global__ __launch_bounds__(256) void msm_bucket_kernel(const g1_affine_t* __restrict__ points,
                                                          const uint32_t* __restrict__ sorted,
                                                          const uint32_t* __restrict__ offsets,
                                                          const uint32_t* __restrict__ counts, uint32_t n, uint32_t nb,
                                                          uint32_t total_buckets, g1_xyzz_t* __restrict__ buckets,
                                                          uint32_t* __restrict__ big_list, uint32_t* __restrict__ big_count) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total_buckets) return;
    const uint32_t nbk = nb - 1;
    const uint32_t w = gid / nbk, b = gid % nbk + 1;
    const uint32_t cnt = counts[(size_t)w * nb + b];
    g1_xyzz_t acc = xyzz_identity();
    if (cnt > BIG_BUCKET) {
        big_list[atomicAdd(big_count, 1u)] = gid;
    } else {
        const uint32_t* src = sorted + (size_t)w * n + offsets[(size_t)w * nb + b];
        for (uint32_t k = 0; k < cnt; ++k) {
            const g1_affine_t p = load_signed_point(points, src[k]);
            xyzz_madd(acc, p);
        }
    }
    st_xyzz(buckets + gid, acc);
}

// ---- 4b. big buckets: one workgroup per bucket ----
__global__ __launch_bounds__(256) void msm_big_bucket_kernel(const g1_affine_t* __restrict__ points,
                                                              const uint32_t* __restrict__ sorted,
                                                              const uint32_t* __restrict__ offsets,
                                                              const uint32_t* __restrict__ counts, uint32_t n, uint32_t nb,
                                                              const uint32_t* __restrict__ big_list,
                                                              const uint32_t* __restrict__ big_count,
                                                              g1_xyzz_t* __restrict__ buckets) {
    __shared__ g1_xyzz_t red[256];
    const uint32_t tid = threadIdx.x;
    const uint32_t nbig = *big_count;
    for (uint32_t item = blockIdx.x; item < nbig; item += gridDim.x) {
        const uint32_t gid = big_list[item];
        const uint32_t nbk = nb - 1;
        const uint32_t w = gid / nbk, b = gid % nbk + 1;
        const uint32_t cnt = counts[(size_t)w * nb + b];
        const uint32_t* src = sorted + (size_t)w * n + offsets[(size_t)w * nb + b];
        g1_xyzz_t acc = xyzz_identity();
        for (uint32_t k = tid; k < cnt; k += 256) {
            const g1_affine_t p = load_signed_point(points, src[k]);
            xyzz_madd(acc, p);
        }
        st_xyzz(&red[tid], acc);
        __syncthreads();
        for (uint32_t s = 128; s > 0; s >>= 1) {
            if (tid < s) {
                const g1_xyzz_t o = ld_xyzz(&red[tid + s]);
                xyzz_add(acc, o);
                st_xyzz(&red[tid], acc);
            }
            __syncthreads();
        }
        if (tid == 0) st_xyzz(buckets + gid, acc);
        __syncthreads();
    }
}

// ---- 5a. chunked running sums: lane (w, t) covers buckets b in (t*L, (t+1)*L] ----
__global__ __launch_bounds__(256) void msm_chunk_reduce_kernel(const g1_xyzz_t* __restrict__ buckets, uint32_t nbk,
                                                                uint32_t log_l, uint32_t chunks_per_window,
                                                                uint32_t total_chunks, g1_xyzz_t* __restrict__ partial) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total_chunks) return;
    const uint32_t w = gid / chunks_per_window, t = gid % chunks_per_window;
    const uint32_t L = 1u << log_l;
    const g1_xyzz_t* B = buckets + (size_t)w * nbk + ((size_t)t << log_l);
    g1_xyzz_t running = xyzz_identity(), acc = xyzz_identity();
    for (uint32_t j = L; j-- > 0;) {
        const g1_xyzz_t bj = ld_xyzz(B + j);
        xyzz_add(running, bj);
        xyzz_add(acc, running);
    }
    // acc = sum (b - t*L) * B_b ; add (t*L) * running
    const uint32_t k = t << log_l;
    if (k != 0 && !xyzz_is_inf(running)) {
        g1_xyzz_t r = xyzz_identity();
        for (int bit = 31 - __clz(k); bit >= 0; --bit) {
            r = xyzz_double(r);
            if ((k >> bit) & 1u) xyzz_add(r, running);
        }
        xyzz_add(acc, r);
    }
    st_xyzz(partial + gid, acc);
}

// ---- 5b. per-window tree over the chunk partials ----
__global__ __launch_bounds__(256) void msm_window_reduce_kernel(const g1_xyzz_t* __restrict__ partial,
                                                                 uint32_t chunks_per_window,
                                                                 g1_xyzz_t* __restrict__ window_sums) {
    __shared__ g1_xyzz_t red[256];
    const uint32_t w = blockIdx.x, tid = threadIdx.x;
    g1_xyzz_t acc = xyzz_identity();
    for (uint32_t k = tid; k < chunks_per_window; k += 256) {
        const g1_xyzz_t o = ld_xyzz(partial + (size_t)w * chunks_per_window + k);
        xyzz_add(acc, o);
    }
    st_xyzz(&red[tid], acc);
    __syncthreads();
    for (uint32_t s = 128; s > 0; s >>= 1) {
        if (tid < s) {
            const g1_xyzz_t o = ld_xyzz(&red[tid + s]);
            xyzz_add(acc, o);
            st_xyzz(&red[tid], acc);
        }
        __syncthreads();
    }
    if (tid == 0) st_xyzz(window_sums + w, acc);
}

// ------------------------------------------------------------------------------------------
uint32_t msm_window_bits(size_t n) {
    uint32_t lg = 0;
    while (((size_t)1 << (lg + 1)) <= n) ++lg;
    int c = (int)lg - 4;
    return (uint32_t)std::max(4, std::min(16, c));
}

static host::G1XyzzH xyzz_to_host(const g1_xyzz_t& d) {
    host::G1XyzzH h;
    memcpy(h.x.l, d.x.l, 48); memcpy(h.y.l, d.y.l, 48); memcpy(h.zz.l, d.zz.l, 48); memcpy(h.zzz.l, d.zzz.l, 48);
    return h;
}

struct MsmWorkspace {
    DevBuf<int32_t> digits;
    DevBuf<uint32_t> sorted, counts, offsets, cursor, big_list, big_count;
    DevBuf<g1_xyzz_t> buckets, partial, window_sums;
};

static MsmWorkspace* msm_workspace(Ctx* ctx) {
    if (!ctx->msm_ws) ctx->msm_ws = std::shared_ptr<void>(new MsmWorkspace(), [](void* p) { delete static_cast<MsmWorkspace*>(p); });
    return static_cast<MsmWorkspace*>(ctx->msm_ws.get());
}

// Queues the whole MSM on the ctx stream and returns the affine result after one stream sync.
host::G1AffineH msm_run(Ctx* ctx, const g1_affine_t* points_dev, const fr_t* scalars_dev, size_t n) {
    host::G1AffineH ident;
    ident.x = host::Fq::zero(); ident.y = host::Fq::zero(); ident.inf = true;
    if (n == 0) return ident;
    PLONK_REQUIRE(n < ((size_t)1 << 31), "msm: too many points");
    MsmWorkspace* ws = msm_workspace(ctx);
    const uint32_t c = msm_window_bits(n);
    const uint32_t W = (256 + c - 1) / c;
    const uint32_t nbk = 1u << (c - 1), nb = nbk + 1;
    const uint32_t total_buckets = W * nbk;
    const uint32_t log_l = std::min(3u, c - 1);
    const uint32_t chunks = nbk >> log_l, total_chunks = W * chunks;
    ws->digits.ensure((size_t)W * n);
    ws->sorted.ensure((size_t)W * n);
    ws->counts.ensure((size_t)W * nb);
    ws->offsets.ensure((size_t)W * nb);
    ws->cursor.ensure((size_t)W * nb);
    ws->big_list.ensure(total_buckets);
    ws->big_count.ensure(4);
    ws->buckets.ensure(total_buckets);
    ws->partial.ensure(total_chunks);
    ws->window_sums.ensure(W);
    hipStream_t st = ctx->stream;
    HIP_CHECK(hipMemsetAsync(ws->counts.p, 0, (size_t)W * nb * sizeof(uint32_t), st));
    HIP_CHECK(hipMemsetAsync(ws->big_count.p, 0, 16, st));
    const uint32_t nblk = (uint32_t)((n + 255) / 256);
    hipLaunchKernelGGL(msm_digits_kernel, dim3(nblk), dim3(256), 0, st, scalars_dev, (uint32_t)n, c, W, nb, ws->digits.p,
                       ws->counts.p);
    hipLaunchKernelGGL(msm_scan_kernel, dim3(W), dim3(1024), 0, st, ws->counts.p, nb, ws->offsets.p, ws->cursor.p);
    hipLaunchKernelGGL(msm_scatter_kernel, dim3(nblk), dim3(256), 0, st, ws->digits.p, (uint32_t)n, W, nb, ws->cursor.p,
                       ws->sorted.p);
    hipLaunchKernelGGL(msm_bucket_kernel, dim3((total_buckets + 255) / 256), dim3(256), 0, st, points_dev, ws->sorted.p,
                       ws->offsets.p, ws->counts.p, (uint32_t)n, nb, total_buckets, ws->buckets.p, ws->big_list.p,
                       ws->big_count.p);
    hipLaunchKernelGGL(msm_big_bucket_kernel, dim3(512), dim3(256), 0, st, points_dev, ws->sorted.p, ws->offsets.p,
                       ws->counts.p, (uint32_t)n, nb, ws->big_list.p, ws->big_count.p, ws->buckets.p);
    hipLaunchKernelGGL(msm_chunk_reduce_kernel, dim3((total_chunks + 255) / 256), dim3(256), 0, st, ws->buckets.p, nbk,
                       log_l, chunks, total_chunks, ws->partial.p);
    hipLaunchKernelGGL(msm_window_reduce_kernel, dim3(W), dim3(256), 0, st, ws->partial.p, chunks, ws->window_sums.p);
    HIP_CHECK(hipGetLastError());
    std::vector<g1_xyzz_t> sums(W);
    HIP_CHECK(hipMemcpyAsync(sums.data(), ws->window_sums.p, W * sizeof(g1_xyzz_t), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    host::G1XyzzH acc = xyzz_to_host(sums[W - 1]);
    for (int w = (int)W - 2; w >= 0; --w) {
        for (uint32_t k = 0; k < c; ++k) acc = host::g1_double(acc);
        acc = host::g1_add(acc, xyzz_to_host(sums[w]));
    }
    return host::g1_to_affine(acc);
}

// ------------------------------------------------------------------------------------------
// SRS: g1_powers resident on the device as (x, y) Montgomery pairs, (0, 0) for the identity.

static void g1_pack(const plonk_g1* in, size_t n, std::vector<g1_affine_t>& out) {
    out.resize(n);
    for (size_t i = 0; i < n; ++i) {
        if (in[i].infinity) {
            memset(&out[i], 0, sizeof(g1_affine_t));
        } else {
            memcpy(out[i].x.l, in[i].x, 48);
            memcpy(out[i].y.l, in[i].y, 48);
        }
    }
}

void g1_unpack(const g1_affine_t& in, plonk_g1* out) {
    memset(out, 0, sizeof(*out));
    bool zero = true;
    for (int k = 0; k < 12; ++k) zero = zero && in.x.l[k] == 0 && in.y.l[k] == 0;
    if (zero) {
        out->infinity = 1;
        return;
    }
    memcpy(out->x, in.x.l, 48);
    memcpy(out->y, in.y.l, 48);
}

void g1_from_host(const host::G1AffineH& a, plonk_g1* out) {
    memset(out, 0, sizeof(*out));
    if (a.inf) {
        out->infinity = 1;
        return;
    }
    memcpy(out->x, a.x.l, 48);
    memcpy(out->y, a.y.l, 48);
}

Srs* srs_upload(Ctx* ctx, const plonk_g1* pts, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        if (pts[i].infinity) continue;
        host::G1AffineH a;
        memcpy(a.x.l, pts[i].x, 48);
        memcpy(a.y.l, pts[i].y, 48);
        a.inf = false;
        PLONK_REQUIRE(!host::Fq::geq_p(a.x) && !host::Fq::geq_p(a.y), "srs: coordinate not reduced");
        if (i < 4 || i + 1 == n) PLONK_REQUIRE(host::g1_on_curve(a), "srs: point not on the curve");
    }
    std::unique_ptr<Srs> s(new Srs());
    s->device = ctx->device;
    s->n = n;
    std::vector<g1_affine_t> packed;
    g1_pack(pts, n, packed);
    s->points.alloc(n);
    if (n) HIP_CHECK(hipMemcpy(s->points.p, packed.data(), n * sizeof(g1_affine_t), hipMemcpyHostToDevice));
    return s.release();
}

// ---- fixed-base generation of [tau^i]_1 ----
// table[w][d-1] = d * 2^(8w) * G, w < 32, d in 1..255 (affine)
static const std::vector<g1_affine_t>& fixed_base_table() {
    static std::vector<g1_affine_t> table;
    if (!table.empty()) return table;
    using namespace host;
    std::vector<g1_affine_t> t(32 * 255);
    G1XyzzH base = g1_from_affine(g1_generator());
    for (int w = 0; w < 32; ++w) {
        G1XyzzH acc = g1_identity();
        for (int d = 1; d <= 255; ++d) {
            acc = g1_add(acc, base);
            G1AffineH a = g1_to_affine(acc);
            memcpy(t[w * 255 + d - 1].x.l, a.x.l, 48);
            memcpy(t[w * 255 + d - 1].y.l, a.y.l, 48);
        }
        base = g1_add(acc, base);  // 256 * base
    }
    table.swap(t);
    return table;
}

__global__ __launch_bounds__(256) void srs_powers_kernel(fr_t tau, uint64_t first, uint32_t n,
                                                          const g1_affine_t* __restrict__ table,
                                                          g1_xyzz_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // tau^(first + i) by square-and-multiply
    fr_t r = f_one<FrP>(), b = tau;
    for (uint64_t e = first + i; e; e >>= 1) {
        if (e & 1u) r = fr_mul(r, b);
        b = fr_mul(b, b);
    }
    fr_t s = f_from_mont<FrP>(r);
    g1_xyzz_t acc = xyzz_identity();
    for (uint32_t w = 0; w < 32; ++w) {
        const uint32_t d = s.l[0] & 0xffu;
#pragma unroll
        for (int k = 0; k < 7; ++k) s.l[k] = (s.l[k] >> 8) | (s.l[k + 1] << 24);
        s.l[7] >>= 8;
        if (d) {
            const g1_affine_t p = ld_affine(table + w * 255 + d - 1);
            xyzz_madd(acc, p);
        }
    }
    st_xyzz(out + i, acc);
}

__device__ __noinline__ fq_t fq_inverse(const fq_t& a) {
    // a^(q-2): q-2 differs from q only in limb 0 (0xffffaaab - 2)
    fq_t r = f_one<FqP>();
#pragma unroll 1
    for (int limb = 11; limb >= 0; --limb) {
        uint32_t e = FqP::p(0);
        switch (limb) {
            case 11: e = FqP::p(11); break; case 10: e = FqP::p(10); break; case 9: e = FqP::p(9); break;
            case 8: e = FqP::p(8); break; case 7: e = FqP::p(7); break; case 6: e = FqP::p(6); break;
            case 5: e = FqP::p(5); break; case 4: e = FqP::p(4); break; case 3: e = FqP::p(3); break;
            case 2: e = FqP::p(2); break; case 1: e = FqP::p(1); break; default: e = FqP::p(0) - 2u; break;
        }
#pragma unroll 1
        for (int bit = 31; bit >= 0; --bit) {
            r = fq_mul(r, r);
            if ((e >> bit) & 1u) r = fq_mul(r, a);
        }
    }
    return r;
}

// XYZZ -> affine for chunks of CH consecutive points per lane (Montgomery batch inversion through
// a global scratch array of prefix products).
__global__ __launch_bounds__(256) void xyzz_to_affine_kernel(const g1_xyzz_t* __restrict__ in, uint32_t n, uint32_t ch,
                                                              fq_t* __restrict__ prefix, g1_affine_t* __restrict__ out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lo = t * ch;
    if (lo >= n) return;
    const uint32_t hi = min(n, lo + ch);
    fq_t run = f_one<FqP>();
    for (uint32_t i = lo; i < hi; ++i) {
        st(prefix + i, run);  // product of the non-identity entries before i
        const fq_t zz = ld(&in[i].zz);
        if (!f_is_zero<FqP>(zz)) run = fq_mul(run, fq_mul(zz, ld(&in[i].zzz)));
    }
    fq_t inv = fq_inverse(run);
    for (uint32_t i = hi; i-- > lo;) {
        const fq_t zz = ld(&in[i].zz);
        g1_affine_t a;
        if (f_is_zero<FqP>(zz)) {
            a.x = f_zero<FqP>();
            a.y = f_zero<FqP>();
        } else {
            const fq_t zzz = ld(&in[i].zzz);
            const fq_t tinv = fq_mul(inv, ld(prefix + i));  // 1 / (zz * zzz)
            inv = fq_mul(inv, fq_mul(zz, zzz));
            a.x = fq_mul(ld(&in[i].x), fq_mul(tinv, zzz));
            a.y = fq_mul(ld(&in[i].y), fq_mul(tinv, zz));
        }
        st(&out[i].x, a.x);
        st(&out[i].y, a.y);
    }
}

Srs* srs_from_tau(Ctx* ctx, const host::fr_h& tau, size_t first, size_t n) {
    PLONK_REQUIRE(n < ((size_t)1 << 31), "srs: too many points");
    std::unique_ptr<Srs> s(new Srs());
    s->device = ctx->device;
    s->n = n;
    s->points.alloc(n);
    if (n == 0) return s.release();
    const std::vector<g1_affine_t>& table = fixed_base_table();
    DevBuf<g1_affine_t> table_dev(table.size());
    HIP_CHECK(hipMemcpy(table_dev.p, table.data(), table.size() * sizeof(g1_affine_t), hipMemcpyHostToDevice));
    DevBuf<g1_xyzz_t> tmp(n);
    DevBuf<fq_t> prefix(n);
    const uint32_t nblk = (uint32_t)((n + 255) / 256);
    hipLaunchKernelGGL(srs_powers_kernel, dim3(nblk), dim3(256), 0, ctx->stream, to_dev(tau), (uint64_t)first, (uint32_t)n,
                       table_dev.p, tmp.p);
    const uint32_t ch = 16;
    const uint32_t lanes = (uint32_t)((n + ch - 1) / ch);
    hipLaunchKernelGGL(xyzz_to_affine_kernel, dim3((lanes + 255) / 256), dim3(256), 0, ctx->stream, tmp.p, (uint32_t)n, ch,
                       prefix.p, s->points.p);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return s.release();
}

}  // namespace plonk
