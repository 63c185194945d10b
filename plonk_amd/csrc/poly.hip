// Device kernels for the O(n) polynomial work between the NTT / MSM phases of the prover
// (backend_new/src/plonk/prover.rs): blinding (:205-211,:318-320), the round-2 grand product
// (:265-302), Horner evaluations at zeta (:595-604; polynomial.rs:103-113), the round-5 linear
// combinations (:742-807) and division by (X - zeta) (polynomial.rs:188-204).  Field arithmetic is
// exact, so any association order gives the same canonical element as the reference's serial loops.
#include <algorithm>
#include <vector>

#include "common.hpp"
#include "poly.hpp"

namespace plonk {

static constexpr uint32_t CH = 32;  // elements per lane in the chunked scans

__device__ __forceinline__ fr_t fr_pow_u64(fr_t b, uint64_t e) {
    fr_t r = f_one<FrP>();
    while (e) {
        if (e & 1ull) r = fr_mul(r, b);
        b = fr_mul(b, b);
        e >>= 1;
    }
    return r;
}

// ---- inversion of a nonzero Montgomery residue by the binary "almost inverse" (Kaliski): the loop
//      below ends with r = -a^-1 2^k mod p, 255 <= k <= 510, after k rounds of shifts, one 256-bit
//      subtraction and one addition each -- about a fifth of the dependent instructions of the
//      exponentiation a^(p-2), which matters because one lane's inversion is the longest chain of
//      the round-2 kernel.  With a = x R the wanted x^-1 R is (p - r) * 2^(512 - k). ----
struct U256 {
    uint32_t l[8];
};
__device__ __forceinline__ U256 u256_shr1(const U256& a) {
    U256 r;
#pragma unroll
    for (int i = 0; i < 7; ++i) r.l[i] = (a.l[i] >> 1) | (a.l[i + 1] << 31);
    r.l[7] = a.l[7] >> 1;
    return r;
}
__device__ __forceinline__ U256 u256_shl1(const U256& a) {
    U256 r;
    r.l[0] = a.l[0] << 1;
#pragma unroll
    for (int i = 1; i < 8; ++i) r.l[i] = (a.l[i] << 1) | (a.l[i - 1] >> 31);
    return r;
}
__device__ __forceinline__ U256 u256_add(const U256& a, const U256& b) {
    U256 r;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        c += (uint64_t)a.l[i] + b.l[i];
        r.l[i] = (uint32_t)c;
        c >>= 32;
    }
    return r;
}
// a - b and whether it borrowed (a < b)
__device__ __forceinline__ U256 u256_sub(const U256& a, const U256& b, bool& borrow) {
    U256 r;
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        c += (int64_t)a.l[i] - (int64_t)b.l[i];
        r.l[i] = (uint32_t)c;
        c >>= 32;
    }
    borrow = c != 0;
    return r;
}
__device__ __forceinline__ U256 u256_sel(bool take_a, const U256& a, const U256& b) {
    U256 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.l[i] = take_a ? a.l[i] : b.l[i];
    return r;
}

__device__ __noinline__ fr_t fr_inverse(const fr_t& a) {
    U256 u, v, r, s, pm;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        pm.l[i] = FrP::p(i);
        u.l[i] = pm.l[i];
        v.l[i] = a.l[i];
        r.l[i] = 0;
        s.l[i] = i == 0 ? 1u : 0u;
    }
    uint32_t k = 0;
#pragma unroll 1
    for (;;) {
        uint32_t any = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) any |= v.l[i];
        if (any == 0 || k >= 512u) break;   // k <= 510 for any nonzero residue below p; the cap bounds the loop on other input
        const bool u_even = (u.l[0] & 1u) == 0, v_even = (v.l[0] & 1u) == 0;
        bool u_lt_v, v_lt_u;
        const U256 umv = u256_sub(u, v, u_lt_v);
        const U256 vmu = u256_sub(v, u, v_lt_u);
        const bool case_a = u_even, case_b = !u_even && v_even;
        const bool case_c = !u_even && !v_even && v_lt_u;    // u > v
        const bool case_d = !u_even && !v_even && !v_lt_u;   // u <= v (equal only at u = v = 1: v becomes 0)
        const U256 t = u256_add(r, s);
        u = case_a ? u256_shr1(u) : (case_c ? u256_shr1(umv) : u);
        v = case_b ? u256_shr1(v) : (case_d ? u256_shr1(vmu) : v);
        if (case_c) r = t;
        if (case_d) s = t;
        if (case_a || case_c) s = u256_shl1(s);
        else r = u256_shl1(r);
        ++k;
    }
    bool lt;
    const U256 r_red = u256_sub(r, pm, lt);
    r = u256_sel(lt, r, r_red);                 // r mod p
    const U256 res = u256_sub(pm, r, lt);       // p - r = a^-1 2^k mod p
    // times 2^(512 - k): as a Montgomery product with the Montgomery form of that power of two
    fr_t pow2 = f_one<FrP>(), out;
#pragma unroll 1
    for (uint32_t i = 0; i < 512u - k; ++i) pow2 = fr_add(pow2, pow2);
#pragma unroll
    for (int i = 0; i < 8; ++i) out.l[i] = res.l[i];
    return fr_mul(out, pow2);
}

// coeffs[j] -= bl[j], coeffs[n + j] += bl[j], j < k:  adds (bl[k-1] X^(k-1) + ... + bl[0]) * (X^n - 1)
__global__ void blind_kernel(fr_t* coeffs, uint32_t n, BlindArgs b) {
    const uint32_t j = threadIdx.x;
    if (j >= b.k) return;
    const fr_t v = b.bl[j];
    st(coeffs + j, fr_sub(ld(coeffs + j), v));
    st(coeffs + n + j, fr_add(ld(coeffs + n + j), v));
}

// coeffs[idx] += v (or -= v)
__global__ void poke_kernel(fr_t* coeffs, uint32_t idx, fr_t v, int subtract) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        fr_t c = ld(coeffs + idx);
        st(coeffs + idx, subtract ? fr_sub(c, v) : fr_add(c, v));
    }
}

// Round-2 factors (prover.rs:281-288): num_i, den_i for i < n.  w_lo/w_hi: two-level powers of omega.
__global__ __launch_bounds__(256) void perm_terms_kernel(PermArgs p) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.n) return;
    const fr_t a = ld(p.a + i), b = ld(p.b + i), c = ld(p.c + i);
    fr_t w = ld(p.w_lo + (i & ((1u << p.log_l) - 1u)));
    if (i >> p.log_l) w = fr_mul(w, ld(p.w_hi + (i >> p.log_l)));
    const fr_t bw = fr_mul(p.beta, w);
    const fr_t ag = fr_add(a, p.gamma), bg = fr_add(b, p.gamma), cg = fr_add(c, p.gamma);
    fr_t num = fr_mul(fr_mul(fr_add(ag, bw), fr_add(bg, fr_mul(p.k1, bw))), fr_add(cg, fr_mul(p.k2, bw)));
    fr_t den = fr_mul(fr_mul(fr_add(ag, fr_mul(p.beta, ld(p.s1 + i))), fr_add(bg, fr_mul(p.beta, ld(p.s2 + i)))),
                      fr_add(cg, fr_mul(p.beta, ld(p.s3 + i))));
    st(p.num + i, num);
    st(p.den + i, den);
}

// den[i] <- num[i] / den[i] by Montgomery's trick over a whole workgroup: every lane multiplies up its
// RATIO_CH consecutive denominators, the 256 lane products are scanned in LDS from both sides, ONE
// lane inverts the workgroup's product (a 255-bit exponentiation: the longest dependent chain of the
// kernel, so it is paid once per 256 lanes, not once per lane), and every lane unwinds its chunk.
// A zero denominator sets *flag (prover.rs:290 panics "Denominator is zero").
static constexpr uint32_t RATIO_CH = 16;

__global__ __launch_bounds__(256) void ratio_kernel(const fr_t* __restrict__ num, fr_t* __restrict__ den, uint32_t n,
                                                     fr_t* __restrict__ scratch, uint32_t* __restrict__ flag) {
    __shared__ fr_t pre[256];   // inclusive prefix products of the lane products
    __shared__ fr_t suf[256];   // inclusive suffix products
    __shared__ fr_t inv_all;
    const uint32_t tid = threadIdx.x;
    const uint32_t lo = min(n, (blockIdx.x * 256 + tid) * RATIO_CH), hi = min(n, lo + RATIO_CH);
    fr_t run = f_one<FrP>();
    bool zero = false;
    for (uint32_t i = lo; i < hi; ++i) {
        st(scratch + i, run);   // product of this lane's denominators before i
        fr_t d = ld(den + i);
        if (f_is_zero<FrP>(d)) { zero = true; d = f_one<FrP>(); }
        run = fr_mul(run, d);
    }
    if (zero) atomicOr(flag, 1u);
    st(&pre[tid], run);
    st(&suf[tid], run);
    __syncthreads();
    fr_t p = run, q = run;
    for (uint32_t off = 1; off < 256; off <<= 1) {
        fr_t pv = f_one<FrP>(), qv = f_one<FrP>();
        if (tid >= off) pv = ld(&pre[tid - off]);
        if (tid + off < 256) qv = ld(&suf[tid + off]);
        __syncthreads();
        p = fr_mul(pv, p);
        q = fr_mul(q, qv);
        st(&pre[tid], p);
        st(&suf[tid], q);
        __syncthreads();
    }
    if (tid == 0) st(&inv_all, fr_inverse(ld(&pre[255])));
    __syncthreads();
    // 1 / (this lane's product) = 1 / (all) * (lanes before) * (lanes after)
    fr_t inv = ld(&inv_all);
    if (tid > 0) inv = fr_mul(inv, ld(&pre[tid - 1]));
    if (tid < 255) inv = fr_mul(inv, ld(&suf[tid + 1]));
    for (uint32_t i = hi; i-- > lo;) {
        fr_t d = ld(den + i);
        if (f_is_zero<FrP>(d)) d = f_one<FrP>();
        const fr_t dinv = fr_mul(inv, ld(scratch + i));
        inv = fr_mul(inv, d);
        st(den + i, fr_mul(ld(num + i), dinv));
    }
}

// ---- exclusive scans over Fr: OP 0 = add, 1 = mul; reverse: suffix instead of prefix ----
template <int OP>
__device__ __forceinline__ fr_t scan_id() { return OP ? f_one<FrP>() : f_zero<FrP>(); }
template <int OP>
__device__ __forceinline__ fr_t scan_op(const fr_t& a, const fr_t& b) { return OP ? fr_mul(a, b) : fr_add(a, b); }

template <int OP>
__global__ __launch_bounds__(256) void scan_totals_kernel(const fr_t* __restrict__ x, uint32_t n, int reverse,
                                                           fr_t* __restrict__ totals) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lo = t * CH;
    if (lo >= n) return;
    const uint32_t hi = min(n, lo + CH);
    fr_t run = scan_id<OP>();
    for (uint32_t k = lo; k < hi; ++k) run = scan_op<OP>(run, ld(x + (reverse ? n - 1 - k : k)));
    st(totals + t, run);
}

template <int OP>
__global__ __launch_bounds__(1024) void scan_middle_kernel(fr_t* totals, uint32_t m) {
    __shared__ fr_t part[1024];
    const uint32_t tid = threadIdx.x;
    const uint32_t per = (m + 1023) / 1024;
    const uint32_t lo = min(m, tid * per), hi = min(m, lo + per);
    fr_t run = scan_id<OP>();
    for (uint32_t k = lo; k < hi; ++k) {
        const fr_t v = ld(totals + k);
        st(totals + k, run);
        run = scan_op<OP>(run, v);
    }
    st(&part[tid], run);
    __syncthreads();
    fr_t incl = run;
    for (uint32_t off = 1; off < 1024; off <<= 1) {
        fr_t v = scan_id<OP>();
        if (tid >= off) v = ld(&part[tid - off]);
        __syncthreads();
        incl = scan_op<OP>(v, incl);
        st(&part[tid], incl);
        __syncthreads();
    }
    fr_t excl = scan_id<OP>();
    if (tid > 0) excl = ld(&part[tid - 1]);
    for (uint32_t k = lo; k < hi; ++k) st(totals + k, scan_op<OP>(excl, ld(totals + k)));
}

// x and out may be the same vector (no __restrict__: every element is read before it is written)
template <int OP>
__global__ __launch_bounds__(256) void scan_apply_kernel(const fr_t* x, uint32_t n, int reverse,
                                                          const fr_t* __restrict__ totals, fr_t* out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lo = t * CH;
    if (lo >= n) return;
    const uint32_t hi = min(n, lo + CH);
    fr_t run = ld(totals + t);
    for (uint32_t k = lo; k < hi; ++k) {
        const uint32_t idx = reverse ? n - 1 - k : k;
        const fr_t v = ld(x + idx);
        st(out + idx, run);
        run = scan_op<OP>(run, v);
    }
}

// `totals` is scratch for the per-chunk totals of every level: m + m/CH + ... elements.
template <int OP>
static void exclusive_scan_level(Ctx* ctx, const fr_t* x, fr_t* out, uint32_t n, bool reverse, fr_t* totals) {
    const uint32_t m = (n + CH - 1) / CH;
    const uint32_t blocks = (m + 255) / 256;
    hipLaunchKernelGGL(scan_totals_kernel<OP>, dim3(blocks), dim3(256), 0, ctx->stream, x, n, reverse ? 1 : 0, totals);
    // the chunk totals are scanned in the same direction by the level above (in place); only a few
    // thousand are left to the single-workgroup kernel, whose run time is a chain of dependent steps
    if (m > 4096) exclusive_scan_level<OP>(ctx, totals, totals, m, false, totals + m);
    else hipLaunchKernelGGL(scan_middle_kernel<OP>, dim3(1), dim3(1024), 0, ctx->stream, totals, m);
    hipLaunchKernelGGL(scan_apply_kernel<OP>, dim3(blocks), dim3(256), 0, ctx->stream, x, n, reverse ? 1 : 0, totals, out);
}

template <int OP>
static void exclusive_scan(Ctx* ctx, const fr_t* x, fr_t* out, uint32_t n, bool reverse, DevBuf<fr_t>& totals) {
    if (n == 0) return;
    size_t need = 0;
    for (uint32_t m = (n + CH - 1) / CH;; m = (m + CH - 1) / CH) {
        need += m;
        if (m <= 4096) break;
    }
    totals.ensure(need);
    exclusive_scan_level<OP>(ctx, x, out, n, reverse, totals.p);
    HIP_CHECK(hipGetLastError());
}

void scan_prefix_product(Ctx* ctx, const fr_t* x, fr_t* out, uint32_t n, DevBuf<fr_t>& totals) {
    exclusive_scan<1>(ctx, x, out, n, false, totals);
}

// ---- polynomial evaluation: out[q] = sum_i coeffs_q[i] * x_q^i ----
__global__ __launch_bounds__(256) void eval_partial_kernel(EvalArgs a, fr_t* __restrict__ partial, uint32_t blocks_per_poly) {
    __shared__ fr_t red[256];
    const uint32_t q = blockIdx.y, tid = threadIdx.x;
    const uint32_t t = blockIdx.x * blockDim.x + tid;
    const uint32_t len = a.len[q];
    const fr_t x = a.x[q];
    const uint32_t lo = t * CH;
    fr_t h = f_zero<FrP>();
    if (lo < len) {
        const uint32_t hi = min(len, lo + CH);
        for (uint32_t i = hi; i-- > lo;) h = fr_add(fr_mul(h, x), ld(a.coeffs[q] + i));
        if (lo) h = fr_mul(h, fr_pow_u64(x, lo));
    }
    st(&red[tid], h);
    __syncthreads();
    for (uint32_t s = 128; s > 0; s >>= 1) {
        if (tid < s) {
            h = fr_add(h, ld(&red[tid + s]));
            st(&red[tid], h);
        }
        __syncthreads();
    }
    if (tid == 0) st(partial + (size_t)q * blocks_per_poly + blockIdx.x, h);
}

__global__ __launch_bounds__(256) void eval_final_kernel(const fr_t* __restrict__ partial, uint32_t blocks_per_poly,
                                                          fr_t* __restrict__ out) {
    __shared__ fr_t red[256];
    const uint32_t q = blockIdx.x, tid = threadIdx.x;
    fr_t h = f_zero<FrP>();
    for (uint32_t k = tid; k < blocks_per_poly; k += 256) h = fr_add(h, ld(partial + (size_t)q * blocks_per_poly + k));
    st(&red[tid], h);
    __syncthreads();
    for (uint32_t s = 128; s > 0; s >>= 1) {
        if (tid < s) {
            h = fr_add(h, ld(&red[tid + s]));
            st(&red[tid], h);
        }
        __syncthreads();
    }
    if (tid == 0) st(out + q, h);
}

void poly_eval_batch(Ctx* ctx, const EvalArgs& a, uint32_t count, fr_t* out_dev, DevBuf<fr_t>& partial) {
    uint32_t max_len = 1;
    for (uint32_t q = 0; q < count; ++q) max_len = std::max(max_len, a.len[q]);
    const uint32_t lanes = (max_len + CH - 1) / CH;
    const uint32_t blocks = (lanes + 255) / 256;
    partial.ensure((size_t)blocks * count);
    hipLaunchKernelGGL(eval_partial_kernel, dim3(blocks, count), dim3(256), 0, ctx->stream, a, partial.p, blocks);
    hipLaunchKernelGGL(eval_final_kernel, dim3(count), dim3(256), 0, ctx->stream, partial.p, blocks, out_dev);
    HIP_CHECK(hipGetLastError());
}

// ---- out[i] = sum_k s_k * P_k[i] ----
__global__ __launch_bounds__(256) void lincomb_kernel(LincombArgs a, fr_t* __restrict__ out, uint32_t out_len) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= out_len) return;
    fr_t acc = f_zero<FrP>();
    for (uint32_t k = 0; k < a.count; ++k) {
        if (i < a.len[k]) acc = fr_add(acc, fr_mul(a.s[k], ld(a.p[k] + i)));
    }
    st(out + i, acc);
}

void poly_lincomb(Ctx* ctx, const LincombArgs& a, fr_t* out, uint32_t out_len) {
    hipLaunchKernelGGL(lincomb_kernel, dim3((out_len + 255) / 256), dim3(256), 0, ctx->stream, a, out, out_len);
    HIP_CHECK(hipGetLastError());
}

// ---- division by (X - z): q_i = z^-(i+1) * sum_{j>i} c_j z^j  (polynomial.rs:188-204) ----
__global__ __launch_bounds__(256) void scale_powers_kernel(const fr_t* __restrict__ in, fr_t* __restrict__ out, uint32_t n,
                                                            fr_t base, uint32_t exp_offset) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lo = t * CH;
    if (lo >= n) return;
    const uint32_t hi = min(n, lo + CH);
    fr_t p = fr_pow_u64(base, (uint64_t)lo + exp_offset);
    for (uint32_t i = lo; i < hi; ++i) {
        st(out + i, fr_mul(ld(in + i), p));
        p = fr_mul(p, base);
    }
}

void poly_div_linear(Ctx* ctx, const fr_t* c, uint32_t len, const host::fr_h& z, fr_t* tmp, fr_t* q_out,
                     DevBuf<fr_t>& totals) {
    // tmp[j] = c_j z^j ; q_out[i] = exclusive suffix sum ; q_out[i] *= z^-(i+1), i < len - 1
    if (len < 2) return;
    if (host::Fr::is_zero(z)) {  // division by X: drop the constant term
        HIP_CHECK(hipMemcpyAsync(q_out, c + 1, (size_t)(len - 1) * sizeof(fr_t), hipMemcpyDeviceToDevice, ctx->stream));
        return;
    }
    const uint32_t lanes = (len + CH - 1) / CH, blocks = (lanes + 255) / 256;
    hipLaunchKernelGGL(scale_powers_kernel, dim3(blocks), dim3(256), 0, ctx->stream, c, tmp, len, to_dev(z), 0u);
    exclusive_scan<0>(ctx, tmp, tmp, len, true, totals);
    const host::fr_h zinv = host::Fr::inv(z);
    hipLaunchKernelGGL(scale_powers_kernel, dim3(blocks), dim3(256), 0, ctx->stream, tmp, q_out, len - 1, to_dev(zinv), 1u);
    HIP_CHECK(hipGetLastError());
}

// ---- flag |= any(x[lo..hi) != 0) ----
__global__ __launch_bounds__(256) void nonzero_kernel(const fr_t* __restrict__ x, uint32_t lo, uint32_t hi,
                                                       uint32_t* __restrict__ flag) {
    const uint32_t i = lo + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= hi) return;
    if (!f_is_zero<FrP>(ld(x + i))) atomicOr(flag, 2u);
}

// ---- device-to-device copy as a kernel: hipMemcpyAsync may route a same-device copy through a DMA
//      engine at a few percent of the HBM rate (1.7 ms for 32 MB measured) ----
__global__ __launch_bounds__(256) void copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

void poly_copy(Ctx* ctx, fr_t* dst, const fr_t* src, size_t n) {
    if (n == 0) return;
    const size_t n16 = n * (sizeof(fr_t) / sizeof(uint4));
    const uint32_t blocks = (uint32_t)std::min<size_t>((n16 + 255) / 256, 8192);
    hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, ctx->stream, reinterpret_cast<const uint4*>(src),
                       reinterpret_cast<uint4*>(dst), n16);
    HIP_CHECK(hipGetLastError());
}

void poly_blind(Ctx* ctx, fr_t* coeffs, uint32_t n, const host::fr_h* bl, uint32_t k) {
    BlindArgs b;
    b.k = k;
    for (uint32_t j = 0; j < 3; ++j) b.bl[j] = j < k ? to_dev(bl[j]) : to_dev(host::Fr::zero());
    hipLaunchKernelGGL(blind_kernel, dim3(1), dim3(64), 0, ctx->stream, coeffs, n, b);
    HIP_CHECK(hipGetLastError());
}

void poly_poke(Ctx* ctx, fr_t* coeffs, uint32_t idx, const host::fr_h& v, bool subtract) {
    hipLaunchKernelGGL(poke_kernel, dim3(1), dim3(64), 0, ctx->stream, coeffs, idx, to_dev(v), subtract ? 1 : 0);
    HIP_CHECK(hipGetLastError());
}

void perm_terms(Ctx* ctx, const PermArgs& p) {
    hipLaunchKernelGGL(perm_terms_kernel, dim3((p.n + 255) / 256), dim3(256), 0, ctx->stream, p);
    HIP_CHECK(hipGetLastError());
}

void perm_ratio(Ctx* ctx, const fr_t* num, fr_t* den, uint32_t n, fr_t* scratch, uint32_t* flag) {
    const uint32_t lanes = (n + RATIO_CH - 1) / RATIO_CH;
    hipLaunchKernelGGL(ratio_kernel, dim3((lanes + 255) / 256), dim3(256), 0, ctx->stream, num, den, n, scratch, flag);
    HIP_CHECK(hipGetLastError());
}

void poly_nonzero_check(Ctx* ctx, const fr_t* x, uint32_t lo, uint32_t hi, uint32_t* flag) {
    if (hi <= lo) return;
    hipLaunchKernelGGL(nonzero_kernel, dim3((hi - lo + 255) / 256), dim3(256), 0, ctx->stream, x, lo, hi, flag);
    HIP_CHECK(hipGetLastError());
}

}  // namespace plonk
