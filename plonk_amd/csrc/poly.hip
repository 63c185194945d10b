// Device kernels for the O(n) polynomial work between the NTT / MSM phases of the prover
// (backend_new/src/plonk/prover.rs): blinding (:205-211,:318-320), the round-2 grand product
// (:265-302), Horner evaluations at zeta (:595-604; polynomial.rs:103-113), the round-5 linear
// combinations (:742-807) and division by (X - zeta) (polynomial.rs:188-204).  Field arithmetic is
// exact, so any association order gives the same canonical element as the reference's serial loops.
#include <algorithm>
#include <vector>

#include "common.hpp"
#include "fieldx.cuh"
#include "poly.hpp"

namespace plonk {

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

// ---- workgroup building blocks for the O(n) passes: a workgroup of 256 lanes covers PW = 2048 consecutive
//      elements, lane t owning the 8 consecutive ones from 8 t on (what a scan wants), while global memory
//      is touched in whole 32-byte elements by consecutive lanes (what HBM wants): the two views meet in LDS.
//      Slot of element u: one pad element per chunk, so the lanes' chunks start 9 elements apart. ----
static constexpr uint32_t PT = 256, PE = 8, PW = PT * PE;
static constexpr uint32_t STAGE_ELEMS = PT * (PE + 1);
__device__ __forceinline__ uint32_t stage_slot(uint32_t u) { return u + u / PE; }

// v[j] = g[index(base + 8 t + j)] for logical indices below len, `fill` above; index(r) = r, or len - 1 - r (reverse)
__device__ __forceinline__ void stage_in(const fr_t* __restrict__ g, uint32_t base, uint32_t len, bool reverse, fr_t* stage,
                                         fr_t (&v)[PE], const fr_t& fill) {
    const uint32_t t = threadIdx.x;
#pragma unroll
    for (uint32_t m = 0; m < PE; ++m) {
        const uint32_t u = t + m * PT, r = base + u;
        st(&stage[stage_slot(u)], r < len ? ld(g + (reverse ? len - 1 - r : r)) : fill);
    }
    __syncthreads();
#pragma unroll
    for (uint32_t j = 0; j < PE; ++j) v[j] = ld(&stage[stage_slot(t * PE + j)]);
    __syncthreads();
}

__device__ __forceinline__ void stage_out(fr_t* __restrict__ g, uint32_t base, uint32_t len, bool reverse, fr_t* stage,
                                          const fr_t (&v)[PE]) {
    const uint32_t t = threadIdx.x;
#pragma unroll
    for (uint32_t j = 0; j < PE; ++j) st(&stage[stage_slot(t * PE + j)], v[j]);
    __syncthreads();
#pragma unroll
    for (uint32_t m = 0; m < PE; ++m) {
        const uint32_t u = t + m * PT, r = base + u;
        if (r < len) st(g + (reverse ? len - 1 - r : r), ld(&stage[stage_slot(u)]));
    }
    __syncthreads();
}

template <int OP>
__device__ __forceinline__ fr_t scan_id();
template <int OP>
__device__ __forceinline__ fr_t scan_op(const fr_t& a, const fr_t& b);

// Exclusive scan over the lanes of a workgroup of `mine` (lane order); *total = the combination of all lanes.
// part: 256 elements of LDS.
template <int OP>
__device__ __forceinline__ fr_t wg_exclusive_scan(const fr_t& mine, fr_t* part, fr_t* total) {
    const uint32_t t = threadIdx.x;
    fr_t incl = mine;
    st(&part[t], incl);
    __syncthreads();
    for (uint32_t off = 1; off < PT; off <<= 1) {
        fr_t o = scan_id<OP>();
        if (t >= off) o = ld(&part[t - off]);
        __syncthreads();
        incl = scan_op<OP>(o, incl);
        st(&part[t], incl);
        __syncthreads();
    }
    *total = ld(&part[PT - 1]);
    fr_t excl = scan_id<OP>();
    if (t > 0) excl = ld(&part[t - 1]);
    __syncthreads();
    return excl;
}

// x^i from the two-level table of a point: lo[u] = x^u, u < 2048; hi[w] = x^(2048 w)
__device__ __forceinline__ fr_t tab_pow(const fr_t* __restrict__ lo, const fr_t* __restrict__ hi, uint32_t i) {
    const fr_t l = ld(lo + (i & (PW - 1)));
    return (i >> 11) ? fr_mul(l, ld(hi + (i >> 11))) : l;
}

__global__ __launch_bounds__(256) void pow_table_kernel(PowTabArgs a) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t per = PW + a.n_hi;
    if (idx >= per * a.count) return;
    const uint32_t which = idx / per, e = idx % per;
    // square-free: the powers x^(2^k) come with the launch
    uint32_t exp = e < PW ? e : (e - PW) << 11;
    fr_t r = f_one<FrP>();
#pragma unroll 1
    for (uint32_t k = 0; exp; ++k, exp >>= 1)
        if (exp & 1u) r = fr_mul(r, a.sq[which][k]);
    st(a.out + (size_t)which * per + e, r);
}

// ---- round 2 (prover.rs:265-302) in one pass per workgroup of 2048 gates: the factors num_i, den_i, the ratios
//      num_i / den_i by Montgomery's trick over the whole workgroup (ONE inversion per 2048 denominators, by the
//      binary almost-inverse above), and the exclusive prefix products of the ratios inside the workgroup.
//      local[i] = prod of the ratios before i in its workgroup; totals[wg] = prod of the workgroup's ratios.
//      A zero denominator sets *flag (prover.rs:290 panics "Denominator is zero"). ----
__global__ __launch_bounds__(256) void perm_fused_kernel(PermArgs p, fr_t* __restrict__ local, fr_t* __restrict__ totals,
                                                         uint32_t* __restrict__ flag) {
    extern __shared__ __attribute__((aligned(16))) unsigned char perm_lds[];
    fr_t* stage = reinterpret_cast<fr_t*>(perm_lds);
    fr_t* part = stage + STAGE_ELEMS;
    fr_t* part2 = part + PT;
    const uint32_t t = threadIdx.x, base = blockIdx.x * PW, i0 = base + t * PE;
    const fr_t zero = f_zero<FrP>(), one = f_one<FrP>();
    fr_t num[PE], den[PE];
    {
        fr_t a[PE], s[PE];
        fr_t w = one;
        if (i0 < p.n) {
            w = ld(p.w_lo + (i0 & ((1u << p.log_l) - 1u)));
            if (i0 >> p.log_l) w = fr_mul(w, ld(p.w_hi + (i0 >> p.log_l)));
        }
        fr_t bw[PE];
#pragma unroll
        for (uint32_t j = 0; j < PE; ++j) {
            bw[j] = fr_mul(p.beta, w);
            w = fr_mul(w, p.omega);
        }
        // wire a against the identity (beta w^i) and S_sigma1; then b (k1) and S_sigma2; then c (k2) and S_sigma3
        stage_in(p.a, base, p.n, false, stage, a, zero);
        stage_in(p.s1, base, p.n, false, stage, s, zero);
#pragma unroll
        for (uint32_t j = 0; j < PE; ++j) {
            const fr_t ag = fr_add(a[j], p.gamma);
            num[j] = fr_add(ag, bw[j]);
            den[j] = fr_add(ag, fr_mul(p.beta, s[j]));
        }
        stage_in(p.b, base, p.n, false, stage, a, zero);
        stage_in(p.s2, base, p.n, false, stage, s, zero);
#pragma unroll
        for (uint32_t j = 0; j < PE; ++j) {
            const fr_t bg = fr_add(a[j], p.gamma);
            num[j] = fr_mul(num[j], fr_add(bg, fr_mul(p.k1, bw[j])));
            den[j] = fr_mul(den[j], fr_add(bg, fr_mul(p.beta, s[j])));
        }
        stage_in(p.c, base, p.n, false, stage, a, zero);
        stage_in(p.s3, base, p.n, false, stage, s, zero);
#pragma unroll
        for (uint32_t j = 0; j < PE; ++j) {
            const fr_t cg = fr_add(a[j], p.gamma);
            num[j] = fr_mul(num[j], fr_add(cg, fr_mul(p.k2, bw[j])));
            den[j] = fr_mul(den[j], fr_add(cg, fr_mul(p.beta, s[j])));
        }
    }
    // only i < n - 1 enter the product (prover.rs:271); the rest count as the ratio 1
    bool zero_den = false;
#pragma unroll
    for (uint32_t j = 0; j < PE; ++j) {
        if (i0 + j + 1 >= p.n) {
            num[j] = one;
            den[j] = one;
        } else if (f_is_zero<FrP>(den[j])) {
            zero_den = true;
            den[j] = one;
        }
    }
    if (zero_den) atomicOr(flag, 1u);
    // batch inversion: products before each element inside the lane, lane products scanned from both sides
    fr_t pre[PE];
    fr_t run = one;
#pragma unroll
    for (uint32_t j = 0; j < PE; ++j) {
        pre[j] = run;
        run = fr_mul(run, den[j]);
    }
    fr_t all;
    const fr_t before = wg_exclusive_scan<1>(run, part, &all);
    // suffix products of the lanes after this one: the same scan on the mirrored lane order
    st(&part2[PT - 1 - t], run);
    __syncthreads();
    const fr_t mirrored = ld(&part2[t]);
    __syncthreads();
    fr_t all2;
    const fr_t after_m = wg_exclusive_scan<1>(mirrored, part, &all2);   // lane t holds the product of the lanes after lane 255 - t
    st(&part2[PT - 1 - t], after_m);
    __syncthreads();
    const fr_t after = ld(&part2[t]);
    __shared__ fr_t inv_all;
    if (t == 0) st(&inv_all, fr_inverse(all));
    __syncthreads();
    fr_t inv = fr_mul(fr_mul(ld(&inv_all), before), after);   // 1 / (this lane's product)
    fr_t ratio[PE];
#pragma unroll
    for (uint32_t jj = 0; jj < PE; ++jj) {
        const uint32_t j = PE - 1 - jj;
        ratio[j] = fr_mul(num[j], fr_mul(inv, pre[j]));
        inv = fr_mul(inv, den[j]);
    }
    // exclusive prefix products of the ratios inside the workgroup
    run = one;
#pragma unroll
    for (uint32_t j = 0; j < PE; ++j) {
        pre[j] = run;
        run = fr_mul(run, ratio[j]);
    }
    fr_t total;
    const fr_t off = wg_exclusive_scan<1>(run, part, &total);
#pragma unroll
    for (uint32_t j = 0; j < PE; ++j) pre[j] = fr_mul(off, pre[j]);
    stage_out(local, base, p.n, false, stage, pre);
    if (t == 0) st(totals + blockIdx.x, total);
}

// out[i] = offs[i / 2048] * local[i]   (OP 1)   or   offs[...] + local[i]   (OP 0)
template <int OP>
__global__ __launch_bounds__(256) void apply_offsets_kernel(const fr_t* __restrict__ local, const fr_t* __restrict__ offs, uint32_t n,
                                                            fr_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    st(out + i, scan_op<OP>(ld(offs + (i >> 11)), ld(local + i)));
}

// ---- exclusive scans over Fr: OP 0 = add, 1 = mul; reverse: suffix instead of prefix ----
template <int OP>
__device__ __forceinline__ fr_t scan_id() { return OP ? f_one<FrP>() : f_zero<FrP>(); }
template <int OP>
__device__ __forceinline__ fr_t scan_op(const fr_t& a, const fr_t& b) { return OP ? fr_mul(a, b) : fr_add(a, b); }

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

// exclusive scan (in place) of the per-workgroup totals
template <int OP>
static void scan_totals(Ctx* ctx, fr_t* totals, uint32_t m) {
    PLONK_REQUIRE(m <= (1u << 22), "scan: too many workgroup totals");
    // one workgroup of 1024 lanes, each taking a run of consecutive totals (a few at the sizes in use: 512 totals
    // at 2^20 elements, 8192 at 2^24)
    hipLaunchKernelGGL(scan_middle_kernel<OP>, dim3(1), dim3(1024), 0, ctx->stream, totals, m);
    HIP_CHECK(hipGetLastError());
}

void pow_tables(Ctx* ctx, const host::fr_h* points, uint32_t count, uint32_t max_len, DevBuf<fr_t>& storage, PowTab* out) {
    PLONK_REQUIRE(count >= 1 && count <= POWTAB_MAX, "pow tables: 1..4 points");
    PowTabArgs a;
    a.count = count;
    a.n_hi = (max_len + PW - 1) / PW + 1;
    const uint32_t per = PW + a.n_hi;
    storage.ensure((size_t)per * count);
    a.out = storage.p;
    for (uint32_t q = 0; q < count; ++q) {
        host::fr_h sq = points[q];
        for (uint32_t k = 0; k < 32; ++k) {
            a.sq[q][k] = to_dev(sq);
            sq = host::Fr::sqr(sq);
        }
        out[q].lo = storage.p + (size_t)q * per;
        out[q].hi = out[q].lo + PW;
    }
    hipLaunchKernelGGL(pow_table_kernel, dim3((per * count + 255) / 256), dim3(256), 0, ctx->stream, a);
    HIP_CHECK(hipGetLastError());
}

static constexpr size_t PERM_LDS_BYTES = (STAGE_ELEMS + 2 * PT) * sizeof(fr_t);

void perm_grand_product(Ctx* ctx, const PermArgs& p, fr_t* z_out, fr_t* local, uint32_t* flag, DevBuf<fr_t>& totals) {
    static bool attr_set[64] = {};
    if (!attr_set[ctx->device & 63]) {
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(perm_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)PERM_LDS_BYTES));
        attr_set[ctx->device & 63] = true;
    }
    const uint32_t wgs = (p.n + PW - 1) / PW;
    totals.ensure(wgs);
    hipLaunchKernelGGL(perm_fused_kernel, dim3(wgs), dim3(PT), PERM_LDS_BYTES, ctx->stream, p, local, totals.p, flag);
    HIP_CHECK(hipGetLastError());
    scan_totals<1>(ctx, totals.p, wgs);
    hipLaunchKernelGGL(apply_offsets_kernel<1>, dim3((p.n + 255) / 256), dim3(256), 0, ctx->stream, local, totals.p, p.n, z_out);
    HIP_CHECK(hipGetLastError());
}

// ---- polynomial evaluation: out[q] = sum_i coeffs_q[i] * x_q^i (prover.rs:595-604; polynomial.rs:103-113).
//      A lane takes the 8 coefficients base + t + 256 k: Horner in x^256, then one product with x^(base + t)
//      from the point's table; the lanes' terms are added up over the workgroup, the workgroups' by a second kernel. ----
__global__ __launch_bounds__(256) void eval_partial_kernel(EvalArgs a, fr_t* __restrict__ partial, uint32_t blocks_per_poly) {
    __shared__ fr_t red[256];
    const uint32_t q = blockIdx.y, tid = threadIdx.x;
    const uint32_t len = a.len[q];
    const uint32_t first = blockIdx.x * PW + tid;
    fr_t h = f_zero<FrP>();
    if (first < len) {
        const fr_t step = ld(a.lo[q] + PT);   // x^256
#pragma unroll
        for (uint32_t kk = 0; kk < PE; ++kk) {
            const uint32_t i = first + (PE - 1 - kk) * PT;
            h = fr_mul(h, step);
            if (i < len) h = fr_add(h, ld(a.coeffs[q] + i));
        }
        h = fr_mul(h, tab_pow(a.lo[q], a.hi[q], first));
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
    const uint32_t blocks = (max_len + PW - 1) / PW;
    partial.ensure((size_t)blocks * count);
    hipLaunchKernelGGL(eval_partial_kernel, dim3(blocks, count), dim3(256), 0, ctx->stream, a, partial.p, blocks);
    hipLaunchKernelGGL(eval_final_kernel, dim3(count), dim3(256), 0, ctx->stream, partial.p, blocks, out_dev);
    HIP_CHECK(hipGetLastError());
}

// ---- out[i] = sum_k s_k * P_k[i] (prover.rs:742-807), in the 29-bit-limb form: a stored word image regrouped into
//      29-bit limbs reads as the value / 32, a product with a scalar held as a true value of the form keeps that
//      scale, the at most 16 products (each below 2p) are added without carries (normalised every fourth), and the sum regroups back into
//      the saturated image after one product-free reduction. ----
__global__ __launch_bounds__(256) void lincomb_kernel(LincombArgs a, fr_t* __restrict__ out, uint32_t out_len) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= out_len) return;
    frx acc = x_zero<FrX>();
    for (uint32_t k = 0; k < a.count; ++k) {
        if (i < a.len[k]) {
            frx s;
#pragma unroll
            for (int w = 0; w < 9; ++w) s.l[w] = a.s[k].w[w];
            acc = x_add_lazy<FrX>(acc, x_mul<FrX>(x_repack_in<FrX>(ld(a.p[k] + i)), s));
        }
        if ((k & 3u) == 3u) acc = x_norm_u<FrX>(acc);   // limbs of a sum of up to five normalised values fit 32 bits
    }
    st(out + i, x_repack_out<FrX>(frx_canon64(x_norm_u<FrX>(acc))));   // the sum is below 16 * 2p < 64p
}

void poly_lincomb(Ctx* ctx, const LincombArgs& a, fr_t* out, uint32_t out_len) {
    hipLaunchKernelGGL(lincomb_kernel, dim3((out_len + 255) / 256), dim3(256), 0, ctx->stream, a, out, out_len);
    HIP_CHECK(hipGetLastError());
}

// a scalar (Montgomery, radix 2^256) as a true value of the 29-bit-limb form (radix 2^261): times 32, regrouped
LincombScalar lincomb_scalar(const host::fr_h& mont) {
    const host::fr_h t = host::Fr::mul(mont, host::Fr::from_u64(32));
    LincombScalar r;
    memset(&r, 0, sizeof(r));
    frx_words_from_u64(t.l, r.w);
    return r;
}

// ---- division by (X - z): q_i = z^-(i+1) * sum_{j>i} c_j z^j  (polynomial.rs:188-204).
//      Pass 1, a workgroup per 2048 coefficients from the top down: s_j = c_j z^j, the sums of the s_j above each j
//      inside the workgroup (local), the workgroup's sum (totals).  The totals are scanned; pass 2 adds the sum of
//      the workgroups above and multiplies by z^-(i+1).  Powers come from the tables of z and 1/z. ----
__global__ __launch_bounds__(256) void div_local_kernel(const fr_t* __restrict__ c, uint32_t len, PowTab z, fr_t z_inv,
                                                        fr_t* __restrict__ local, fr_t* __restrict__ totals) {
    extern __shared__ __attribute__((aligned(16))) unsigned char div_lds[];
    fr_t* stage = reinterpret_cast<fr_t*>(div_lds);
    fr_t* part = stage + STAGE_ELEMS;
    const uint32_t t = threadIdx.x, base = blockIdx.x * PW, r0 = base + t * PE;   // logical (top-down) positions
    fr_t v[PE];
    stage_in(c, base, len, true, stage, v, f_zero<FrP>());
    // logical r is coefficient i = len - 1 - r: the lane's first power is z^(len - 1 - r0), then one 1/z per step
    fr_t pw = f_one<FrP>();
    if (r0 < len) pw = tab_pow(z.lo, z.hi, len - 1 - r0);
    fr_t run = f_zero<FrP>();
#pragma unroll
    for (uint32_t j = 0; j < PE; ++j) {
        const fr_t s = fr_mul(v[j], pw);      // zero beyond the end
        pw = fr_mul(pw, z_inv);
        v[j] = run;                           // sum of the terms above this coefficient inside the lane
        run = fr_add(run, s);
    }
    fr_t total;
    const fr_t off = wg_exclusive_scan<0>(run, part, &total);
#pragma unroll
    for (uint32_t j = 0; j < PE; ++j) v[j] = fr_add(v[j], off);
    stage_out(local, base, len, true, stage, v);
    if (t == 0) st(totals + blockIdx.x, total);
}

__global__ __launch_bounds__(256) void div_apply_kernel(const fr_t* __restrict__ local, const fr_t* __restrict__ offs, uint32_t len,
                                                        PowTab z_inv, fr_t* __restrict__ q_out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i + 1 >= len) return;                 // q has len - 1 coefficients
    const uint32_t wg = (len - 1 - i) >> 11;
    const fr_t sum = fr_add(ld(local + i), ld(offs + wg));
    st(q_out + i, fr_mul(sum, tab_pow(z_inv.lo, z_inv.hi, i + 1)));
}

static constexpr size_t DIV_LDS_BYTES = (STAGE_ELEMS + PT) * sizeof(fr_t);

void poly_div_linear(Ctx* ctx, const fr_t* c, uint32_t len, const host::fr_h& z, const host::fr_h& z_inv, const PowTab& z_tab,
                     const PowTab& z_inv_tab, fr_t* tmp, fr_t* q_out, DevBuf<fr_t>& totals) {
    if (len < 2) return;
    if (host::Fr::is_zero(z)) {  // division by X: drop the constant term
        HIP_CHECK(hipMemcpyAsync(q_out, c + 1, (size_t)(len - 1) * sizeof(fr_t), hipMemcpyDeviceToDevice, ctx->stream));
        return;
    }
    static bool attr_set[64] = {};
    if (!attr_set[ctx->device & 63]) {
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(div_local_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)DIV_LDS_BYTES));
        attr_set[ctx->device & 63] = true;
    }
    const uint32_t wgs = (len + PW - 1) / PW;
    totals.ensure(wgs);
    hipLaunchKernelGGL(div_local_kernel, dim3(wgs), dim3(PT), DIV_LDS_BYTES, ctx->stream, c, len, z_tab, to_dev(z_inv), tmp, totals.p);
    HIP_CHECK(hipGetLastError());
    scan_totals<0>(ctx, totals.p, wgs);
    hipLaunchKernelGGL(div_apply_kernel, dim3((len + 255) / 256), dim3(256), 0, ctx->stream, tmp, totals.p, len, z_inv_tab, q_out);
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

void poly_nonzero_check(Ctx* ctx, const fr_t* x, uint32_t lo, uint32_t hi, uint32_t* flag) {
    if (hi <= lo) return;
    hipLaunchKernelGGL(nonzero_kernel, dim3((hi - lo + 255) / 256), dim3(256), 0, ctx->stream, x, lo, hi, flag);
    HIP_CHECK(hipGetLastError());
}

}  // namespace plonk
