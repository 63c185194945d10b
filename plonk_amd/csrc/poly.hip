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
// the same with E elements a lane (a workgroup then covers 256 E): E + 1 slots a lane
template <uint32_t E>
__device__ __forceinline__ uint32_t stage_slot(uint32_t u) { return u + u / E; }

// global -> LDS half of stage_in: stage[slot(u)] = g[index(base + u)] for logical indices below len, `fill` above;
// index(r) = r, or len - 1 - r (reverse)
template <uint32_t E>
__device__ __forceinline__ void stage_load(const fr_t* __restrict__ g, uint32_t base, uint32_t len, bool reverse, fr_t* stage,
                                           const fr_t& fill) {
    const uint32_t t = threadIdx.x;
#pragma clang loop unroll(full)
    for (uint32_t m = 0; m < E; ++m) {
        const uint32_t u = t + m * PT, r = base + u;
        st(&stage[stage_slot<E>(u)], r < len ? ld(g + (reverse ? len - 1 - r : r)) : fill);
    }
    __syncthreads();
}

// v[j] = g[index(base + E t + j)]
template <uint32_t E>
__device__ __forceinline__ void stage_in(const fr_t* __restrict__ g, uint32_t base, uint32_t len, bool reverse, fr_t* stage,
                                         fr_t (&v)[E], const fr_t& fill) {
    const uint32_t t = threadIdx.x;
    stage_load<E>(g, base, len, reverse, stage, fill);
#pragma clang loop unroll(full)
    for (uint32_t j = 0; j < E; ++j) v[j] = ld(&stage[stage_slot<E>(t * E + j)]);
    __syncthreads();
}

template <uint32_t E>
__device__ __forceinline__ void stage_out(fr_t* __restrict__ g, uint32_t base, uint32_t len, bool reverse, fr_t* stage,
                                          const fr_t (&v)[E]) {
    const uint32_t t = threadIdx.x;
#pragma clang loop unroll(full)
    for (uint32_t j = 0; j < E; ++j) st(&stage[stage_slot<E>(t * E + j)], v[j]);
    __syncthreads();
#pragma clang loop unroll(full)
    for (uint32_t m = 0; m < E; ++m) {
        const uint32_t u = t + m * PT, r = base + u;
        if (r < len) st(g + (reverse ? len - 1 - r : r), ld(&stage[stage_slot<E>(u)]));
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

// ---- round 2 (prover.rs:265-302): z_i = prod_{j<i} num_j / den_j without per-element division state.
//      With N(<i) the product of the numerators before i, D(>=i) the product of the denominators from i on:
//          z_i = N(<i) D(>=i) / D(all).
//      perm_tile_kernel, one workgroup per tile of 1024 gates: local[i] = N(<i) D(>=i) restricted to the tile, and the
//      tile's two products.  perm_middle_kernel, one workgroup: offs[w] = (numerators of the tiles before w)
//      (denominators of the tiles after w), and the product of all denominators, whose inverse -- the only inversion --
//      is taken on the host (a single lane of the device needs 0.2 ms for it, the round trip 0.05 ms).
//      z_i = offs[tile] local[i] / D(all).
//      A zero denominator sets *flag (prover.rs:290 panics "Denominator is zero").
//      The tile kernel computes in the 29-bit-limb form.  Stored vectors regrouped into 29-bit limbs read as
//      value / 32 ("scaled"); constants come as true values of the form.  scaled x true = scaled; scaled x scaled
//      loses a factor 32.  Each of num_j, den_j is a product of three scaled factors, i.e. the true-form image of
//      the value times 2^-15; every z_i holds as many numerators as denominators, so these factors cancel. ----
static constexpr uint32_t QE = 4, QW = PT * QE, QW_LOG = 10;   // gates a lane, gates a tile
static constexpr uint32_t Q_STAGE = PT * (QE + 1);

__device__ __forceinline__ frx ld_slot(const frx_slot* p) { return ld_tw(p); }
__device__ __forceinline__ void st_slot(frx_slot* p, const frx& v) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
    q[2] = make_uint4(v.l[8], 0u, 0u, 0u);
}
__device__ __forceinline__ frx frx_from(const PermScalar& c) {
    frx r;
#pragma clang loop unroll(full)
    for (int w = 0; w < 9; ++w) r.l[w] = c.w[w];
    return r;
}
// the value of the lane `delta` below (up) or above (down) in the wavefront
__device__ __forceinline__ frx x_shfl(const frx& v, uint32_t delta, bool up) {
    frx r;
#pragma clang loop unroll(full)
    for (int w = 0; w < 9; ++w) r.l[w] = up ? __shfl_up(v.l[w], delta, 64) : __shfl_down(v.l[w], delta, 64);
    return r;
}

// Lane totals of a workgroup -> for every lane the product of the lanes before it (forward) or after it (backward),
// done by ONE wavefront: its lane l owns the four totals 4 l .. 4 l + 3, multiplies through them, the 64 chunk
// products are scanned with wavefront shuffles, and the chunk's offset is multiplied back in.
// slots: 256 + 64 entries, entry u at u + u / 4; on return slots[..] hold the exclusive products, *total everything.
__device__ __forceinline__ uint32_t lane_slot(uint32_t u) { return u + (u >> 2); }
__device__ __forceinline__ void wave_scan_totals(frx_slot* slots, bool forward, frx_slot* total) {
    const uint32_t l = threadIdx.x & 63u;
    frx run = x_one<FrX>();
#pragma clang loop unroll(full)
    for (uint32_t kk = 0; kk < 4; ++kk) {
        const uint32_t u = 4 * l + (forward ? kk : 3 - kk);
        const frx v = ld_slot(&slots[lane_slot(u)]);
        st_slot(&slots[lane_slot(u)], run);
        run = x_mul<FrX>(run, v);
    }
    frx incl = run;
#pragma clang loop unroll(full)
    for (uint32_t off = 1; off < 64; off <<= 1) {
        const frx o = x_shfl(incl, off, forward);
        const bool has = forward ? l >= off : l + off < 64;
        const frx m = x_mul<FrX>(o, incl);
        if (has) incl = m;
    }
    frx excl = x_shfl(incl, 1, forward);
    if (forward ? l == 0 : l == 63) excl = x_one<FrX>();
    if (forward ? l == 63 : l == 0) st_slot(total, incl);
#pragma clang loop unroll(full)
    for (uint32_t kk = 0; kk < 4; ++kk) {
        const uint32_t u = 4 * l + kk;
        st_slot(&slots[lane_slot(u)], x_mul<FrX>(ld_slot(&slots[lane_slot(u)]), excl));
    }
}

__global__ __launch_bounds__(256, 2) void perm_tile_kernel(PermArgs p, fr_t* __restrict__ local, fr_t* __restrict__ tot_n,
                                                        fr_t* __restrict__ tot_d, uint32_t* __restrict__ flag) {
    extern __shared__ __attribute__((aligned(16))) unsigned char perm_lds[];
    fr_t* stage = reinterpret_cast<fr_t*>(perm_lds);
    // after the inputs are in registers the staging area holds the lane totals
    frx_slot* slots_n = reinterpret_cast<frx_slot*>(perm_lds);
    frx_slot* slots_d = slots_n + 320;
    frx_slot* tile_tot = slots_d + 320;   // [0] numerators, [1] denominators
    const uint32_t t = threadIdx.x, base = blockIdx.x * QW, i0 = base + t * QE;
    const fr_t zero = f_zero<FrP>();
    const frx one = x_one<FrX>();
    const frx beta = frx_from(p.beta_x), gamma = x_repack_in<FrX>(p.gamma), omega = frx_from(p.omega_x);   // true; scaled; true
    // beta w^i0, scaled: (32 beta) w_lo w_hi, or beta w_lo where there is no high factor
    frx bw0 = one;
    if (i0 < p.n) {
        const frx wl = x_repack_in<FrX>(ld(p.w_lo + (i0 & ((1u << p.log_l) - 1u))));
        if (i0 >> p.log_l)
            bw0 = x_mul<FrX>(x_mul<FrX>(wl, frx_from(p.beta32_x)), x_repack_in<FrX>(ld(p.w_hi + (i0 >> p.log_l))));
        else
            bw0 = x_mul<FrX>(wl, beta);
    }
    frx num[QE], den[QE];
    fr_t w[QE];
    // wire a against the identity beta w^i and S_sigma1
    stage_in<QE>(p.a, base, p.n, false, stage, w, zero);
    stage_load<QE>(p.s1, base, p.n, false, stage, zero);
    {
        frx id = bw0;
#pragma clang loop unroll(full)
        for (uint32_t j = 0; j < QE; ++j) {
            const frx wg = x_add_lazy<FrX>(x_repack_in<FrX>(w[j]), gamma);
            const frx sg = x_repack_in<FrX>(ld(&stage[stage_slot<QE>(t * QE + j)]));
            num[j] = x_norm_u<FrX>(x_add_lazy<FrX>(wg, id));                          // below 4p
            den[j] = x_norm_u<FrX>(x_add_lazy<FrX>(wg, x_mul<FrX>(sg, beta)));
            if (j + 1 < QE) id = x_mul<FrX>(id, omega);
        }
    }
    __syncthreads();
    // wires b and c against k1 beta w^i, S_sigma2 and k2 beta w^i, S_sigma3
#pragma unroll 1
    for (uint32_t k = 1; k < 3; ++k) {
        stage_in<QE>(k == 1 ? p.b : p.c, base, p.n, false, stage, w, zero);
        stage_load<QE>(k == 1 ? p.s2 : p.s3, base, p.n, false, stage, zero);
        frx id = x_mul<FrX>(bw0, frx_from(k == 1 ? p.k1_x : p.k2_x));
#pragma clang loop unroll(full)
        for (uint32_t j = 0; j < QE; ++j) {
            const frx wg = x_add_lazy<FrX>(x_repack_in<FrX>(w[j]), gamma);
            const frx sg = x_repack_in<FrX>(ld(&stage[stage_slot<QE>(t * QE + j)]));
            num[j] = x_mul<FrX>(x_add_lazy<FrX>(wg, id), num[j]);
            den[j] = x_mul<FrX>(x_add_lazy<FrX>(wg, x_mul<FrX>(sg, beta)), den[j]);
            if (j + 1 < QE) id = x_mul<FrX>(id, omega);
        }
        __syncthreads();
    }
    // only i < n - 1 enter the product (prover.rs:271); the rest count as the ratio 1
    bool zero_den = false;
#pragma clang loop unroll(full)
    for (uint32_t j = 0; j < QE; ++j) {
        if (i0 + j + 1 >= p.n) {
            num[j] = one;
            den[j] = one;
        } else if (x_is_zero16<FrX>(den[j])) {
            zero_den = true;
            den[j] = one;
        }
    }
    if (zero_den) atomicOr(flag, 1u);
    // inside the lane: num[j] <- product of the numerators before j; den[j] <- product of the denominators from j on
    frx run_n = one, run_d = one;
#pragma clang loop unroll(full)
    for (uint32_t j = 0; j < QE; ++j) {
        const frx v = num[j];
        num[j] = run_n;
        run_n = x_mul<FrX>(run_n, v);
    }
#pragma clang loop unroll(full)
    for (uint32_t jj = 0; jj < QE; ++jj) {
        const uint32_t j = QE - 1 - jj;
        run_d = x_mul<FrX>(run_d, den[j]);
        den[j] = run_d;
    }
    // over the lanes: wavefront 0 takes the numerators (lanes before), wavefront 1 the denominators (lanes after)
    st_slot(&slots_n[lane_slot(t)], run_n);
    st_slot(&slots_d[lane_slot(t)], run_d);
    __syncthreads();
    if (t < 64) wave_scan_totals(slots_n, true, &tile_tot[0]);
    else if (t < 128) wave_scan_totals(slots_d, false, &tile_tot[1]);
    __syncthreads();
    // lane factor, with the conversion to the stored form folded in
    frx kout;
#pragma clang loop unroll(full)
    for (int i = 0; i < 9; ++i) kout.l[i] = FrX::kout(i);
    const frx lane = x_mul<FrX>(x_mul<FrX>(ld_slot(&slots_n[lane_slot(t)]), ld_slot(&slots_d[lane_slot(t)])), kout);
    const frx tn = ld_slot(&tile_tot[0]), td = ld_slot(&tile_tot[1]);
    __syncthreads();
    fr_t out[QE];
#pragma clang loop unroll(full)
    for (uint32_t j = 0; j < QE; ++j)
        out[j] = x_repack_out<FrX>(x_canon2<FrX>(x_mul<FrX>(x_mul<FrX>(num[j], den[j]), lane)));
    stage_out<QE>(local, base, p.n, false, stage, out);
    if (t == 0) {
        st(tot_n + blockIdx.x, x_to_sat<FrX>(tn));
        st(tot_d + blockIdx.x, x_to_sat<FrX>(td));
    }
}

// Inclusive scans by products over the lanes of a wavefront, both at once: n over the lanes up to this one, d over
// the lanes from this one on (2^steps_log lanes take part; the lanes beyond must hold one in d).
__device__ __forceinline__ void wave_scan_pair(frx& n, frx& d, uint32_t steps_log) {
    const uint32_t l = threadIdx.x & 63u;
#pragma unroll 1
    for (uint32_t off = 1; off < (1u << steps_log); off <<= 1) {
        const frx o_n = x_shfl(n, off, true), o_d = x_shfl(d, off, false);
        const frx m_n = x_mul<FrX>(o_n, n), m_d = x_mul<FrX>(o_d, d);
        if (l >= off) n = m_n;
        if (l + off < 64) d = m_d;
    }
}

// offs[w] = (prod_{v<w} tot_n[v]) (prod_{v>w} tot_d[v]), *den_all = prod_v tot_d[v]; one workgroup of 1024 lanes, lane t
// taking a run of consecutive tiles; the lanes' products are scanned inside each wavefront by shuffles, the 16
// wavefronts' by the first one.  tot_n and tot_d are overwritten.
__global__ __launch_bounds__(1024) void perm_middle_kernel(fr_t* tot_n, fr_t* tot_d, uint32_t m, fr_t* __restrict__ offs,
                                                           fr_t* __restrict__ den_all) {
    __shared__ frx_slot wave_tot[2][16], wave_off[2][16];
    const uint32_t tid = threadIdx.x, l = tid & 63u, wave = tid >> 6;
    const uint32_t per = (m + 1023) / 1024;
    const uint32_t lo = min(m, tid * per), hi = min(m, lo + per);
    const frx one = x_one<FrX>();
    // inside the lane's run: tot_n[k] <- numerators before k, tot_d[k] <- denominators after k
    frx run_n = one, run_d = one;
#pragma unroll 1
    for (uint32_t k = lo; k < hi; ++k) {
        const frx v = x_from_sat<FrX>(ld(tot_n + k));
        st(tot_n + k, x_to_sat<FrX>(run_n));
        run_n = x_mul<FrX>(run_n, v);
    }
#pragma unroll 1
    for (uint32_t k = hi; k > lo; --k) {
        const frx v = x_from_sat<FrX>(ld(tot_d + k - 1));
        st(tot_d + k - 1, x_to_sat<FrX>(run_d));
        run_d = x_mul<FrX>(run_d, v);
    }
    frx incl_n = run_n, incl_d = run_d;
    wave_scan_pair(incl_n, incl_d, 6);
    frx ex_n = x_shfl(incl_n, 1, true), ex_d = x_shfl(incl_d, 1, false);
    if (l == 0) ex_n = one;
    if (l == 63) ex_d = one;
    if (l == 63) st_slot(&wave_tot[0][wave], incl_n);
    if (l == 0) st_slot(&wave_tot[1][wave], incl_d);
    __syncthreads();
    if (wave == 0) {
        frx a = l < 16 ? ld_slot(&wave_tot[0][l]) : one, b = l < 16 ? ld_slot(&wave_tot[1][l]) : one;
        wave_scan_pair(a, b, 4);
        frx ea = x_shfl(a, 1, true);
        const frx eb = x_shfl(b, 1, false);   // lane 15 reads the one of lane 16
        if (l == 0) ea = one;
        if (l < 16) {
            st_slot(&wave_off[0][l], ea);
            st_slot(&wave_off[1][l], eb);
        }
        if (l == 0) st(den_all, x_to_sat<FrX>(b));   // nonzero: zero denominators were replaced
    }
    __syncthreads();
    const frx before_n = x_mul<FrX>(ld_slot(&wave_off[0][wave]), ex_n), after_d = x_mul<FrX>(ld_slot(&wave_off[1][wave]), ex_d);
    const frx lane = x_mul<FrX>(before_n, after_d);
#pragma unroll 1
    for (uint32_t k = lo; k < hi; ++k)
        st(offs + k, x_to_sat<FrX>(x_mul<FrX>(x_mul<FrX>(x_from_sat<FrX>(ld(tot_n + k)), x_from_sat<FrX>(ld(tot_d + k))), lane)));
}

// out[i] = scale * offs[i >> tile_log] * local[i]
__global__ __launch_bounds__(256) void apply_offsets_kernel(const fr_t* __restrict__ local, const fr_t* __restrict__ offs, uint32_t n,
                                                            uint32_t tile_log, fr_t scale, fr_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    st(out + i, fr_mul(fr_mul(ld(offs + (i >> tile_log)), scale), ld(local + i)));
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

static constexpr size_t PERM_LDS_BYTES = Q_STAGE * sizeof(fr_t);
static_assert((2 * 320 + 2) * sizeof(frx_slot) <= PERM_LDS_BYTES, "the lane totals reuse the staging area");

static PermScalar perm_scalar(const host::fr_h& mont) {
    const host::fr_h t = host::Fr::mul(mont, host::Fr::from_u64(32));
    PermScalar r;
    memset(&r, 0, sizeof(r));
    frx_words_from_u64(t.l, r.w);
    return r;
}

void perm_grand_product(Ctx* ctx, const PermArgs& args, fr_t* z_out, fr_t* local, uint32_t* flag, DevBuf<fr_t>& totals) {
    PermArgs p = args;
    {
        host::fr_h beta, k1, k2, omega;
        memcpy(beta.l, p.beta.l, 32); memcpy(k1.l, p.k1.l, 32); memcpy(k2.l, p.k2.l, 32); memcpy(omega.l, p.omega.l, 32);
        p.beta_x = perm_scalar(beta);
        p.beta32_x = perm_scalar(host::Fr::mul(beta, host::Fr::from_u64(32)));
        p.k1_x = perm_scalar(k1);
        p.k2_x = perm_scalar(k2);
        p.omega_x = perm_scalar(omega);
    }
    const uint32_t wgs = (p.n + QW - 1) / QW;
    PLONK_REQUIRE(wgs <= (1u << 22), "grand product: too many tiles");
    totals.ensure(3 * (size_t)wgs + 1);
    fr_t *tot_n = totals.p, *tot_d = totals.p + wgs, *offs = totals.p + 2 * (size_t)wgs, *den_all = totals.p + 3 * (size_t)wgs;
    hipLaunchKernelGGL(perm_tile_kernel, dim3(wgs), dim3(PT), PERM_LDS_BYTES, ctx->stream, p, local, tot_n, tot_d, flag);
    HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(perm_middle_kernel, dim3(1), dim3(1024), 0, ctx->stream, tot_n, tot_d, wgs, offs, den_all);
    HIP_CHECK(hipGetLastError());
    host::fr_h den;
    HIP_CHECK(hipMemcpyAsync(&den, den_all, sizeof(fr_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    hipLaunchKernelGGL(apply_offsets_kernel, dim3((p.n + 255) / 256), dim3(256), 0, ctx->stream, local, offs, p.n, QW_LOG,
                       to_dev(host::Fr::inv(den)), z_out);
    HIP_CHECK(hipGetLastError());
}

// ---- polynomial evaluation: out[q] = sum_i coeffs_q[i] * x_q^(first_q + i) (prover.rs:595-604; polynomial.rs:103-113).
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
#pragma clang loop unroll(full)
        for (uint32_t kk = 0; kk < PE; ++kk) {
            const uint32_t i = first + (PE - 1 - kk) * PT;
            h = fr_mul(h, step);
            if (i < len) h = fr_add(h, ld(a.coeffs[q] + i));
        }
        h = fr_mul(h, tab_pow(a.lo[q], a.hi[q], first + a.first[q]));
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
#pragma clang loop unroll(full)
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
#pragma clang loop unroll(full)
    for (uint32_t j = 0; j < PE; ++j) {
        const fr_t s = fr_mul(v[j], pw);      // zero beyond the end
        pw = fr_mul(pw, z_inv);
        v[j] = run;                           // sum of the terms above this coefficient inside the lane
        run = fr_add(run, s);
    }
    fr_t total;
    const fr_t off = wg_exclusive_scan<0>(run, part, &total);
#pragma clang loop unroll(full)
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

// coeffs[idx] += *v (or -= *v), v in device memory: no round trip through the host for a value only the device needs
__global__ void poke_from_kernel(fr_t* coeffs, uint32_t idx, const fr_t* __restrict__ v, int subtract) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const fr_t c = ld(coeffs + idx), d = ld(v);
        st(coeffs + idx, subtract ? fr_sub(c, d) : fr_add(c, d));
    }
}
void poly_poke_from(Ctx* ctx, fr_t* coeffs, uint32_t idx, const fr_t* v_dev, bool subtract) {
    hipLaunchKernelGGL(poke_from_kernel, dim3(1), dim3(64), 0, ctx->stream, coeffs, idx, v_dev, subtract ? 1 : 0);
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
