// Proving key, fused round-3 quotient kernel and the 5-round prover shell.
//
// Replaces preprocess (backend_new/src/plonk/preprocess.rs:104-186), the round-3 body
// (plonk/prover.rs:364-457, :497) and Prover::prove (plonk/prover.rs:56-149).  Every bulk vector
// stays in HBM from the witness upload to the last commitment; the host sees only commitments,
// six evaluations and the transcript.
#include <algorithm>
#include <atomic>
#include <future>
#include <vector>

#include "common.hpp"
#include "fieldx.cuh"
#include "msm.hpp"
#include "poly.hpp"
#include "prover.hpp"
#include "transcript.hpp"

namespace plonk {

using host::Fr;
using host::fr_h;

// ---- fused quotient evaluation over the 4n coset points (prover.rs:415-457) ----
// Arithmetic in the 29-bit-limb form of fieldx.cuh.  Vectors stay saturated Montgomery words in
// memory; regrouped into 29-bit limbs a word image reads as the value divided by 32 ("scaled").
// Sums of scaled values are scaled values; a product of two scaled values comes out 32 times too
// small; a product with a constant held as a true value of the form keeps the scale.  Every term
// below is arranged to end up scaled exactly once, so the result regroups back into the saturated
// word image of t: q_M is stored times 2^10 and q_L, q_R, q_O times 2^5 in the key's coset tables,
// and the challenges that multiply the permutation and L1 terms carry 2^15 and 2^5.
struct QuotientArgs {
    const fr_t *a, *b, *c, *z, *pi;   // coset evaluations, 4n each (pi nullable)
    const fr_t* pk_ext;               // [9][4n]: 2^10 q_m, 2^5 q_l, 2^5 q_r, 2^5 q_o, q_c, s1, s2, s3, l1
    const fr_t* x_lo;                 // x_i = x_lo[i & mask] * x_hi[i >> log_l] = 7 * w_4n^i
    const frx_slot* x_hi;             //   (x_hi as true values)
    fr_t* t;
    frx alpha_32_3, alpha2_32, beta, beta_k1, beta_k2;   // true values: 2^15 alpha, 2^5 alpha^2, beta, beta k1, beta k2
    frx zh_inv[4];                    // true values; 1 / (x_i^n - 1) depends on i mod 4 only
    fr_t gamma, one;                  // saturated words (read as scaled values)
    uint32_t n4, log_l;
    // The points this launch covers: i = (j << log_g) + q for j < count.  One GPU: log_g = 0, q = 0, count = 4n.
    // A rank of a sharded proof holds the residue class q modulo G = 2^log_g of the coset: its vectors
    // (wires, z, key tables, t) are indexed by j; zw holds z on the class of i + 4, whose element for j sits
    // at (j + zw_off) mod count.
    const fr_t* zw;
    uint32_t log_g, q, count, zw_off;
};

__device__ __forceinline__ frx ld_scaled(const fr_t* p) { return x_repack_in<FrX>(ld(p)); }

__global__ __launch_bounds__(256) void quotient_kernel(QuotientArgs q) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= q.count) return;
    const size_t N = q.count;                    // row length of the key tables this launch reads
    const uint32_t i = (j << q.log_g) + q.q;     // index of the point on the whole 4n coset
    const frx a = ld_scaled(q.a + j), b = ld_scaled(q.b + j), c = ld_scaled(q.c + j), z = ld_scaled(q.z + j);
    const frx zw = ld_scaled(q.zw + ((j + q.zw_off) & (q.count - 1)));  // z(w X) on the 4n coset is a rotation by 4
    frx x = ld_scaled(q.x_lo + (i & ((1u << q.log_l) - 1u)));
    if (i >> q.log_l) x = x_mul<FrX>(x, ld_tw(q.x_hi + (i >> q.log_l)));
    // gate: q_M a b + q_L a + q_R b + q_O c + q_C + PI   (each product below 2p, limbs normalised)
    // (q_M b + q_L) a: with the tables' scaling (q_M by 2^10, q_L by 2^5) the inner sum is already on one scale
    frx gate = x_mul<FrX>(x_add_lazy<FrX>(x_mul<FrX>(ld_scaled(q.pk_ext + j), b), ld_scaled(q.pk_ext + N + j)), a);
    gate = x_add_lazy<FrX>(gate, x_mul<FrX>(ld_scaled(q.pk_ext + 2 * N + j), b));
    gate = x_add_lazy<FrX>(gate, x_mul<FrX>(ld_scaled(q.pk_ext + 3 * N + j), c));
    gate = x_add_lazy<FrX>(gate, ld_scaled(q.pk_ext + 4 * N + j));
    if (q.pi) gate = x_add_lazy<FrX>(gate, ld_scaled(q.pi + j));
    // permutation: the three-term factors are sums of at most 4p with limbs below 2^31
    const frx gamma = x_repack_in<FrX>(q.gamma);
    const frx ag = x_add_lazy<FrX>(a, gamma), bg = x_add_lazy<FrX>(b, gamma), cg = x_add_lazy<FrX>(c, gamma);
    frx num = x_mul<FrX>(z, x_add_lazy<FrX>(ag, x_mul<FrX>(x, q.beta)));
    num = x_mul<FrX>(num, x_add_lazy<FrX>(bg, x_mul<FrX>(x, q.beta_k1)));
    num = x_mul<FrX>(num, x_add_lazy<FrX>(cg, x_mul<FrX>(x, q.beta_k2)));
    frx den = x_mul<FrX>(zw, x_add_lazy<FrX>(ag, x_mul<FrX>(ld_scaled(q.pk_ext + 5 * N + j), q.beta)));
    den = x_mul<FrX>(den, x_add_lazy<FrX>(bg, x_mul<FrX>(ld_scaled(q.pk_ext + 6 * N + j), q.beta)));
    den = x_mul<FrX>(den, x_add_lazy<FrX>(cg, x_mul<FrX>(ld_scaled(q.pk_ext + 7 * N + j), q.beta)));
    const frx perm = x_mul<FrX>(x_sub_lazy<FrX, 3>(num, den), q.alpha_32_3);     // three products deep: 2^-15
    const frx zm1 = x_sub_lazy<FrX, 3>(z, x_repack_in<FrX>(q.one));
    const frx l1c = x_mul<FrX>(x_mul<FrX>(zm1, ld_scaled(q.pk_ext + 8 * N + j)), q.alpha2_32);
    const frx numer = x_norm_u<FrX>(x_add_lazy<FrX>(gate, x_add_lazy<FrX>(perm, l1c)));   // 8 terms: limbs below 2^32
    const uint32_t r = i & 3u;
    const frx zi = r == 0 ? q.zh_inv[0] : (r == 1 ? q.zh_inv[1] : (r == 2 ? q.zh_inv[2] : q.zh_inv[3]));
    st(q.t + j, x_repack_out<FrX>(x_canon2<FrX>(x_mul<FrX>(numer, zi))));
}

// rows of the coset table times a constant (the 2^10 / 2^5 of q_M / q_L, q_R, q_O)
__global__ __launch_bounds__(256) void scale_rows_kernel(fr_t* __restrict__ x, size_t count, fr_t k) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) st(x + i, fr_mul(ld(x + i), k));
}

static frx to_frx(const fr_h& mont) {
    const fr_h t = Fr::mul(mont, Fr::from_u64(32));
    frx r;
    frx_words_from_u64(t.l, r.l);
    return r;
}

static std::vector<fr_h> host_powers(const fr_h& first, const fr_h& base, size_t count) {
    std::vector<fr_h> v(count);
    fr_h acc = first;
    for (size_t i = 0; i < count; ++i) {
        v[i] = acc;
        acc = Fr::mul(acc, base);
    }
    return v;
}

static void upload_fr(DevBuf<fr_t>& buf, const std::vector<fr_h>& v) {
    buf.alloc(v.size());
    HIP_CHECK(hipMemcpy(buf.p, v.data(), v.size() * sizeof(fr_t), hipMemcpyHostToDevice));
}

Pk* pk_create(Ctx* ctx, uint32_t log_n, const plonk_fr* const sel[5], const plonk_fr* const sig[3], const fr_h& k1,
              const fr_h& k2, const uint32_t* pi_pos, size_t n_pi_pos) {
    PLONK_REQUIRE(log_n >= 2 && log_n <= 26, "pk: domain size must be 2^2 .. 2^26");
    std::unique_ptr<Pk> pk(new Pk());
    static std::atomic<uint64_t> next_id{1};
    pk->id = next_id.fetch_add(1);
    const size_t n = (size_t)1 << log_n, n4 = 4 * n;
    pk->device = ctx->device;
    pk->log_n = log_n;
    pk->n = n;
    pk->k1 = k1;
    pk->k2 = k2;
    pk->omega = host::fr_root_of_unity(log_n);
    for (size_t i = 0; i < n_pi_pos; ++i) {
        PLONK_REQUIRE(pi_pos[i] < n, "pk: public input position out of range");
        pk->pi_positions.push_back(pi_pos[i]);
    }
    pk->coef.alloc(9 * n);
    pk->ext.alloc(9 * n4);
    pk->sig_ev.alloc(3 * n);
    hipStream_t st = ctx->stream;
    for (int k = 0; k < 5; ++k)
        HIP_CHECK(hipMemcpyAsync(pk->coef.p + k * n, sel[k], n * sizeof(fr_t), hipMemcpyHostToDevice, st));
    for (int k = 0; k < 3; ++k) {
        HIP_CHECK(hipMemcpyAsync(pk->coef.p + (5 + k) * n, sig[k], n * sizeof(fr_t), hipMemcpyHostToDevice, st));
        HIP_CHECK(hipMemcpyAsync(pk->sig_ev.p + k * n, sig[k], n * sizeof(fr_t), hipMemcpyHostToDevice, st));
    }
    // L1 evaluations: e_0 (preprocess.rs:142)
    HIP_CHECK(hipMemsetAsync(pk->coef.p + 8 * n, 0, n * sizeof(fr_t), st));
    const fr_h one = Fr::one();
    HIP_CHECK(hipMemcpyAsync(pk->coef.p + 8 * n, &one, sizeof(fr_t), hipMemcpyHostToDevice, st));
    HIP_CHECK(hipStreamSynchronize(st));
    // evaluations -> coefficients (preprocess.rs:115-127)
    ntt_run(ctx, pk->coef.p, log_n, NTT_INVERSE, 9, n);
    // coefficients -> evaluations on the 4n coset 7*H (what prover.rs:380-392 does per proof)
    HIP_CHECK(hipMemcpy2DAsync(pk->ext.p, n4 * sizeof(fr_t), pk->coef.p, n * sizeof(fr_t), n * sizeof(fr_t), 9,
                               hipMemcpyDeviceToDevice, st));
    ntt_run(ctx, pk->ext.p, log_n + 2, NTT_COSET, 9, n4, n);   // rows hold n coefficients; the rest of a row is never read
    // the quotient kernel's scaling of the selector rows (see QuotientArgs)
    hipLaunchKernelGGL(scale_rows_kernel, dim3((uint32_t)((n4 + 255) / 256)), dim3(256), 0, st, pk->ext.p, n4,
                       to_dev(Fr::from_u64(1024)));
    hipLaunchKernelGGL(scale_rows_kernel, dim3((uint32_t)((3 * n4 + 255) / 256)), dim3(256), 0, st, pk->ext.p + n4, 3 * n4,
                       to_dev(Fr::from_u64(32)));
    HIP_CHECK(hipGetLastError());
    // two-level tables: w^i (domain n) and x_i = 7 w_4n^i
    pk->log_lw = (log_n + 1) / 2;
    upload_fr(pk->w_lo, host_powers(Fr::one(), pk->omega, (size_t)1 << pk->log_lw));
    upload_fr(pk->w_hi, host_powers(Fr::one(), Fr::pow_u64(pk->omega, (uint64_t)1 << pk->log_lw),
                                    (size_t)1 << (log_n - pk->log_lw)));
    const fr_h w4 = host::fr_root_of_unity(log_n + 2);
    const fr_h g = Fr::from_u64(7);
    pk->log_lx = (log_n + 3) / 2;
    upload_fr(pk->x_lo, host_powers(g, w4, (size_t)1 << pk->log_lx));
    {
        const std::vector<fr_h> hi = host_powers(Fr::one(), Fr::pow_u64(w4, (uint64_t)1 << pk->log_lx),
                                                 (size_t)1 << (log_n + 2 - pk->log_lx));
        std::vector<frx_slot> slots(hi.size());
        for (size_t i = 0; i < hi.size(); ++i) {
            memset(&slots[i], 0, sizeof(frx_slot));
            const frx v = to_frx(hi[i]);
            memcpy(slots[i].w, v.l, sizeof(v.l));
        }
        pk->x_hi.alloc(slots.size());
        HIP_CHECK(hipMemcpy(pk->x_hi.p, slots.data(), slots.size() * sizeof(frx_slot), hipMemcpyHostToDevice));
    }
    // Z_H(7 w_4n^i) = 7^n * w_4^(i mod 4) - 1   (prover.rs:400-405)
    const fr_h gn = Fr::pow_u64(g, n);
    const fr_h i4 = Fr::pow_u64(w4, n);  // primitive 4th root of unity
    fr_h cur = gn;
    for (int r = 0; r < 4; ++r) {
        const fr_h zh = Fr::sub(cur, Fr::one());
        PLONK_REQUIRE(!Fr::is_zero(zh), "pk: Z_H vanishes on the coset");
        pk->zh_inv[r] = Fr::inv(zh);
        cur = Fr::mul(cur, i4);
    }
    HIP_CHECK(hipStreamSynchronize(st));
    return pk.release();
}

// The residue class of the 4n coset one launch of the quotient kernel covers (see QuotientArgs).
struct QuotientSlice {
    uint32_t log_g = 0, q = 0;
    const fr_t* ext = nullptr;   // key tables restricted to the class, [9][4n >> log_g]; null = the key's own tables
    const fr_t* zw = nullptr;    // z on the class of i + 4; null = z itself
    uint32_t zw_off = 4;
};

void quotient_run(Ctx* ctx, const Pk* pk, fr_t* a, fr_t* b, fr_t* c, fr_t* z, fr_t* pi, const fr_h& alpha, const fr_h& beta,
                  const fr_h& gamma, fr_t* t_out, const QuotientSlice& sl = QuotientSlice()) {
    const uint32_t n4 = (uint32_t)(4 * pk->n);
    QuotientArgs q;
    q.a = a; q.b = b; q.c = c; q.z = z; q.pi = pi;
    q.pk_ext = sl.ext ? sl.ext : pk->ext.p;
    q.x_lo = pk->x_lo.p; q.x_hi = pk->x_hi.p;
    q.t = t_out;
    q.alpha_32_3 = to_frx(Fr::mul(alpha, Fr::from_u64(32768)));
    q.alpha2_32 = to_frx(Fr::mul(Fr::mul(alpha, alpha), Fr::from_u64(32)));
    q.beta = to_frx(beta);
    q.beta_k1 = to_frx(Fr::mul(beta, pk->k1));
    q.beta_k2 = to_frx(Fr::mul(beta, pk->k2));
    q.gamma = to_dev(gamma);
    q.one = to_dev(Fr::one());
    for (int r = 0; r < 4; ++r) q.zh_inv[r] = to_frx(pk->zh_inv[r]);
    q.n4 = n4;
    q.log_l = pk->log_lx;
    q.zw = sl.zw ? sl.zw : z;
    q.log_g = sl.log_g;
    q.q = sl.q;
    q.count = n4 >> sl.log_g;
    q.zw_off = sl.zw_off;
    ctx->q_timed = ctx->kernel_timing;
    if (ctx->q_timed) {
        if (!ctx->q_ev0) {
            HIP_CHECK(hipEventCreate(&ctx->q_ev0));
            HIP_CHECK(hipEventCreate(&ctx->q_ev1));
        }
        HIP_CHECK(hipEventRecord(ctx->q_ev0, ctx->stream));
    }
    hipLaunchKernelGGL(quotient_kernel, dim3((q.count + 255) / 256), dim3(256), 0, ctx->stream, q);
    HIP_CHECK(hipGetLastError());
    if (ctx->q_timed) HIP_CHECK(hipEventRecord(ctx->q_ev1, ctx->stream));
}

// ---- round 3 sharded over the G = 2^g ranks of a proof (prover.rs:364-457, :497 split by evaluation point) ----
// Rank q evaluates everything on the points x_i of the 4n coset with i = q mod G.  Those are the
// coset s_q * H_M, M = 4n / G, s_q = 7 w_4n^q, of the subgroup of order M, so the evaluations of a polynomial f
// there are ONE coset transform of size M of f mod (X^M - s_q^M) -- 1/G of the work, no exchange, every rank
// holding all coefficient vectors.  The quotient is point-wise.  Its evaluations on the class give, by the
// inverse coset transform of size M, F_q = t mod (X^M - c_q) with c_q = s_q^M = 7^M w_G^q:
//     F_q[k] = sum_m t[k + m M] 7^(M m) w_G^(q m),
// a DFT of length G across the ranks.  The F_q are all-gathered (4n field elements in all) and every rank
// inverts that DFT: t[k + m M] = 7^(-M m) / G * sum_q w_G^(-q m) F_q[k].

// out[k] = sum_m in[k + m M] c^m over the m with k + m M < len  (f mod (X^M - c))
__global__ __launch_bounds__(256) void fold_kernel(const fr_t* __restrict__ in, uint32_t len, uint32_t M, fr_t c,
                                                   fr_t* __restrict__ out) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= M) return;
    fr_t acc = k < len ? ld(in + k) : f_zero<FrP>();
    fr_t pw = c;
    for (uint32_t idx = k + M; idx < len; idx += M) {
        acc = fr_add(acc, fr_mul(ld(in + idx), pw));
        pw = fr_mul(pw, c);
    }
    st(out + k, acc);
}

// out[row][j] = in[row][(j << log_g) + q]
__global__ __launch_bounds__(256) void class_gather_kernel(const fr_t* __restrict__ in, size_t in_stride, uint32_t log_g,
                                                           uint32_t q, uint32_t M, fr_t* __restrict__ out) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= M) return;
    const size_t row = blockIdx.y;
    st(out + row * M + j, ld(in + row * in_stride + (((size_t)j << log_g) + q)));
}

// t[k + m M] = 7^(-M m) / G * sum_q w_G^(-q m) F[q][k], m < G = 2^LOG_G: the inverse DFT across the ranks as
// radix-2 butterflies in registers (decimation in time on the bit-reversed ranks), then one scaling per output --
// (G/2) log2 G + G products per k instead of G^2, in the 29-bit-limb form (the values are only regrouped: the
// transform is linear).  Only the multiplied operand of a butterfly is reduced; an element grows by at most 3p
// per stage.
static constexpr uint32_t SHARD_MAX_RANKS = 8;   // the GPUs of one node
struct RecombineConsts {
    frx tw[SHARD_MAX_RANKS / 2];   // w_G^(-j), j < G/2
    frx scale[SHARD_MAX_RANKS];    // 7^(-M m) / G
};
template <uint32_t LOG_G>
__global__ __launch_bounds__(256) void recombine_kernel(const fr_t* __restrict__ F, uint32_t M,
                                                        const RecombineConsts* __restrict__ consts, fr_t* __restrict__ t) {
    constexpr uint32_t G = 1u << LOG_G;
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= M) return;
    frx a[G];
#pragma clang loop unroll(full)
    for (uint32_t q = 0; q < G; ++q) a[__brev(q) >> (32 - LOG_G)] = x_repack_in<FrX>(ld(F + (size_t)q * M + k));   // canonical
#pragma clang loop unroll(full)
    for (uint32_t s = 1; s <= LOG_G; ++s) {
        const uint32_t m = 1u << s, half = m >> 1;
#pragma clang loop unroll(full)
        for (uint32_t base = 0; base < G; base += m) {
#pragma clang loop unroll(full)
            for (uint32_t j = 0; j < half; ++j) {
                const frx u = a[base + j];
                frx v = a[base + j + half];
                // w_m^(-j) = w_G^(-j G/m); the first stage's operands are canonical and need no product
                if (s > 1) {
                    frx w;
#pragma clang loop unroll(full)
                    for (int l = 0; l < 9; ++l) w.l[l] = consts->tw[j << (LOG_G - s)].l[l];
                    v = x_mul<FrX>(v, w);
                }
                a[base + j] = x_norm_u<FrX>(x_add_lazy<FrX>(u, v));
                a[base + j + half] = x_norm_u<FrX>(x_sub_lazy<FrX, 3>(u, v));
            }
        }
    }
#pragma clang loop unroll(full)
    for (uint32_t m = 0; m < G; ++m) {
        frx w;
#pragma clang loop unroll(full)
        for (int l = 0; l < 9; ++l) w.l[l] = consts->scale[m].l[l];
        st(t + (size_t)m * M + k, x_repack_out<FrX>(x_canon2<FrX>(x_mul<FrX>(a[m], w))));
    }
}

// plonk_quotient_3wire_dev: coset NTTs of the wire polynomials, fused evaluation, coset iNTT.
void quotient_3wire(Ctx* ctx, const Pk* pk, fr_t* a, fr_t* b, fr_t* c, fr_t* z, fr_t* pi, const fr_h& alpha,
                    const fr_h& beta, const fr_h& gamma, fr_t* t_out) {
    const uint32_t log4 = pk->log_n + 2;
    ntt_run(ctx, a, log4, NTT_COSET, 1, 0);
    ntt_run(ctx, b, log4, NTT_COSET, 1, 0);
    ntt_run(ctx, c, log4, NTT_COSET, 1, 0);
    ntt_run(ctx, z, log4, NTT_COSET, 1, 0);
    if (pi) ntt_run(ctx, pi, log4, NTT_COSET, 1, 0);
    quotient_run(ctx, pk, a, b, c, z, pi, alpha, beta, gamma, t_out);
    ntt_run(ctx, t_out, log4, NTT_COSET | NTT_INVERSE, 1, 0);
}

// ------------------------------------------------------------------------------------------
struct ProverWorkspace {
    DevBuf<fr_t> w4;      // [5][4n]  a b c z pi: coefficients, then coset evaluations
    DevBuf<fr_t> t;       // [4n]
    DevBuf<fr_t> ev;      // [3][n]   wire evaluations
    DevBuf<fr_t> co;      // [4][n+8] blinded coefficient vectors a b c z
    DevBuf<fr_t> scr;     // [n]
    DevBuf<fr_t> tp;      // [3][n+8] t_lo t_mid t_hi (blinded)
    DevBuf<fr_t> wnum, wq, wq2, tmp;  // [n+8]
    DevBuf<fr_t> evals;   // [8]
    DevBuf<fr_t> totals, partial;
    DevBuf<fr_t> powtab;  // power tables of zeta, 1/zeta, zeta omega, 1/(zeta omega) for rounds 4-5
    DevBuf<uint32_t> flag;
    DevBuf<uint64_t> xchg;   // sharded proving: (lo, count) of every rank's SRS slice
    DevBuf<fr_t> gath;       // sharded proving: the ranks' partial sums of rounds 4 and 5
    // round 3 sharded by evaluation point (see fold_kernel): this rank's residue class of the 4n coset
    DevBuf<fr_t> sl;         // [6][M] a b c z, z on the class of i + 4, pi: folded coefficients, then evaluations
    DevBuf<fr_t> tq;         // [M]    t on the class, then F_q
    DevBuf<fr_t> fall;       // [G][M] the F_q of all ranks
    DevBuf<fr_t> ext_slice;  // [9][M] the key's coset tables on the class
    DevBuf<RecombineConsts> coef;   // twiddles and scalings of the inverse DFT across the ranks
    unsigned char* stage = nullptr;   // pinned staging area for a host witness
    size_t stage_bytes = 0;
    ~ProverWorkspace() {
        if (stage) (void)hipHostFree(stage);
    }
    uint64_t slice_pk_id = 0;   // Pk::id the tables above were gathered from (0 = none)
    uint32_t slice_g = 0, slice_q = 0;
};

// Geometry of a sharded round 3 for this rank.
struct PointShard {
    bool on = false;
    uint32_t G = 1, log_g = 0, q = 0, q2 = 0, log_m = 0, zw_off = 4;
    size_t M = 0;
    fr_h s_q, c_q, s_q2, c_q2;   // class generators 7 w_4n^q and their M-th powers
};

static PointShard point_shard(const Pk* pk, const ShardInfo* shard) {
    PointShard ps;
    if (!shard || !shard->ex) return ps;
    const uint32_t G = shard->ex->world;
    const size_t n4 = 4 * pk->n;
    if (G < 2 || G > SHARD_MAX_RANKS || (G & (G - 1)) || n4 / G < 16) return ps;   // otherwise round 3 stays replicated
    ps.on = true;
    ps.G = G;
    while ((1u << ps.log_g) < G) ++ps.log_g;
    ps.q = shard->ex->rank;
    ps.M = n4 / G;
    ps.log_m = pk->log_n + 2 - ps.log_g;
    ps.q2 = (ps.q + 4) % G;
    ps.zw_off = (ps.q + 4) >> ps.log_g;
    const fr_h w4 = host::fr_root_of_unity(pk->log_n + 2), g = Fr::from_u64(7);
    ps.s_q = Fr::mul(g, Fr::pow_u64(w4, ps.q));
    ps.c_q = Fr::pow_u64(ps.s_q, ps.M);
    ps.s_q2 = Fr::mul(g, Fr::pow_u64(w4, ps.q2));
    ps.c_q2 = Fr::pow_u64(ps.s_q2, ps.M);
    return ps;
}

// Per (key, world, rank): the key's coset tables on this rank's class and the recombination matrix.
static void point_shard_tables(Ctx* ctx, ProverWorkspace* ws, const Pk* pk, const PointShard& ps) {
    const size_t M = ps.M, n4 = 4 * pk->n;
    ws->sl.ensure(6 * M);
    ws->tq.ensure(M);
    ws->fall.ensure(n4);
    if (ws->slice_pk_id == pk->id && ws->slice_g == ps.G && ws->slice_q == ps.q && ws->ext_slice.n >= 9 * M) return;
    ws->slice_pk_id = 0;
    ws->ext_slice.ensure(9 * M);
    hipLaunchKernelGGL(class_gather_kernel, dim3((uint32_t)((M + 255) / 256), 9), dim3(256), 0, ctx->stream, pk->ext.p, n4,
                       ps.log_g, ps.q, (uint32_t)M, ws->ext_slice.p);
    HIP_CHECK(hipGetLastError());
    // the inverse DFT across the ranks: w_G = w_4n^M; output m is scaled by 7^(-M m) / G
    const fr_h w4 = host::fr_root_of_unity(pk->log_n + 2);
    const fr_h wg_inv = Fr::inv(Fr::pow_u64(w4, M));
    const fr_h gm_inv = Fr::inv(Fr::pow_u64(Fr::from_u64(7), M));
    RecombineConsts rc;
    memset(&rc, 0, sizeof(rc));
    fr_h cur = Fr::one();
    for (uint32_t j = 0; j < ps.G / 2; ++j) {
        rc.tw[j] = to_frx(cur);
        cur = Fr::mul(cur, wg_inv);
    }
    fr_h scale = Fr::inv(Fr::from_u64(ps.G));
    for (uint32_t m = 0; m < ps.G; ++m) {
        rc.scale[m] = to_frx(scale);
        scale = Fr::mul(scale, gm_inv);
    }
    ws->coef.ensure(1);
    HIP_CHECK(hipMemcpyAsync(ws->coef.p, &rc, sizeof(rc), hipMemcpyHostToDevice, ctx->stream));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    ws->slice_pk_id = pk->id;
    ws->slice_g = ps.G;
    ws->slice_q = ps.q;
}

// evaluations of the polynomial `coeffs` (len coefficients) on the class generated by s: fold, then one coset transform
static void class_evaluate(Ctx* ctx, hipStream_t stream, const fr_t* coeffs, size_t len, const PointShard& ps, const fr_h& s,
                           const fr_h& c, fr_t* out) {
    hipLaunchKernelGGL(fold_kernel, dim3((uint32_t)((ps.M + 255) / 256)), dim3(256), 0, stream, coeffs, (uint32_t)len, (uint32_t)ps.M,
                       to_dev(c), out);
    HIP_CHECK(hipGetLastError());
    ntt_run_shift_on(ctx, stream, out, ps.log_m, NTT_COSET, s, 1, 0);
}

static ProverWorkspace* prover_workspace(Ctx* ctx) {
    if (!ctx->prover_ws)
        ctx->prover_ws = std::shared_ptr<void>(new ProverWorkspace(), [](void* p) { delete static_cast<ProverWorkspace*>(p); });
    return static_cast<ProverWorkspace*>(ctx->prover_ws.get());
}

host::G1AffineH g1_sum_affine(const plonk_g1* pts, size_t n) {
    host::G1XyzzH acc = host::g1_identity();
    for (size_t i = 0; i < n; ++i) {
        host::G1AffineH a;
        memcpy(a.x.l, pts[i].x, 48);
        memcpy(a.y.l, pts[i].y, 48);
        a.inf = pts[i].infinity != 0;
        acc = host::g1_add(acc, host::g1_from_affine(a));
    }
    return host::g1_to_affine(acc);
}

// a communicator of one rank still runs its (trivial) all-gathers: the same code path for every world size
static bool sharded(const ShardInfo* shard) { return shard && shard->ex; }

// This rank's part [lo, hi) of a polynomial of `len` coefficients: all of it (clipped to the SRS) on
// one GPU, the slice of the rank's SRS points otherwise.  Coefficients past the last SRS point must be
// zero: kzg.rs:232-238 strips trailing zeros before its length check and panics otherwise; here the
// flag is raised, identically on every rank (they all hold the whole polynomial).
// The slice of the SRS a rank uses for commitment `index` of a round of `batch` commitments, where in the
// coefficient vector it starts, and whether this rank takes part in that commitment at all.
struct CommitSlice {
    const Srs* srs;
    size_t lo;
    bool mine;
};
static CommitSlice commit_slice(const Srs* srs, const ShardInfo* shard, int batch, int index) {
    CommitSlice c{srs, sharded(shard) ? shard->lo : 0, true};
    if (!sharded(shard) || batch < 2 || batch > 3 || !shard->group_srs[batch - 2]) return c;
    c.srs = shard->group_srs[batch - 2];
    c.lo = shard->group_lo[batch - 2];
    c.mine = shard->ex->rank % (uint32_t)batch == (uint32_t)index;
    return c;
}

static void commit_range(Ctx* ctx, const CommitSlice& cs, const fr_t* coeffs, size_t len, uint32_t* flag_dev, const ShardInfo* shard,
                         size_t* lo, size_t* hi) {
    const size_t end = sharded(shard) ? shard->global_end : cs.srs->n;
    if (len > end) poly_nonzero_check(ctx, coeffs, (uint32_t)end, (uint32_t)len, flag_dev);
    if (sharded(shard)) {
        *lo = std::min(cs.lo, len);
        *hi = cs.mine ? std::min(cs.lo + cs.srs->n, len) : *lo;   // not this rank's commitment: an empty share
    } else {
        *lo = 0;
        *hi = std::min(len, cs.srs->n);
    }
}

static host::G1AffineH commit_dev(Ctx* ctx, const Srs* srs, const fr_t* coeffs, size_t len, uint32_t* flag_dev,
                                  const ShardInfo* shard = nullptr, const MsmFiller& filler = nullptr) {
    size_t lo, hi;
    const CommitSlice cs = commit_slice(srs, shard, 1, 0);
    commit_range(ctx, cs, coeffs, len, flag_dev, shard, &lo, &hi);
    return msm_run(ctx, msm_points(cs.srs), coeffs + lo, hi - lo, filler, sharded(shard) ? shard->ex : nullptr);
}

// Several commitments against the same SRS, overlapped on the device (independent MSMs on
// separate streams).  commit_add starts one as soon as its coefficients are ready on ctx->stream;
// commit_finish collects them.  Sharded runs all-gather the ranks' shares inside the MSM.
// `batch`: how many commitments the round will add (it selects the placement, see ShardInfo).
static void commit_add(Ctx* ctx, const Srs* srs, MsmBatch& b, const fr_t* coeffs, size_t len, uint32_t* flag_dev,
                       const ShardInfo* shard, int batch, bool hold_buckets = false) {
    const CommitSlice cs = commit_slice(srs, shard, batch, b.count);
    if (b.count == 0) {
        b.pts = msm_points(cs.srs);
        b.ex = sharded(shard) ? shard->ex : nullptr;
    }
    size_t lo, hi;
    commit_range(ctx, cs, coeffs, len, flag_dev, shard, &lo, &hi);
    msm_batch_add(ctx, b, coeffs + lo, hi - lo, hold_buckets);
}

static void commit_finish(Ctx* ctx, MsmBatch& b, const ShardInfo* shard, host::G1AffineH* out,
                          const MsmFiller& filler = nullptr) {
    (void)shard;
    msm_batch_finish(ctx, b, out, filler);
}

static void commit_batch(Ctx* ctx, const Srs* srs, const fr_t* const* coeffs, const size_t* lens, int count,
                         uint32_t* flag_dev, const ShardInfo* shard, host::G1AffineH* out, const MsmFiller& filler = nullptr) {
    MsmBatch b;
    for (int k = 0; k < count; ++k) commit_add(ctx, srs, b, coeffs[k], lens[k], flag_dev, shard, count);
    commit_finish(ctx, b, shard, out, filler);
}

// The ranks' SRS slices must tile [0, global_end) in rank order; every rank learns global_end.  With placement
// by groups all ranks or none bring the slices for a batch size b, and those of the ranks r = k mod b must tile
// the same range, for every k < b.
static void shard_check_coverage(Ctx* ctx, ProverWorkspace* ws, const Srs* srs, ShardInfo* shard) {
    Exchange* ex = shard->ex;
    constexpr size_t REC = 6;   // (first point, points) of the plain slice and of the slices for b = 2, 3 (points = 0: none)
    ws->xchg.ensure(REC + REC * (size_t)ex->world);
    uint64_t mine[REC] = {shard->lo, srs->n, 0, 0, 0, 0};
    for (int g = 0; g < 2; ++g) {
        if (!shard->group_srs[g]) continue;
        PLONK_REQUIRE(shard->group_srs[g]->n > 0, "prove: empty SRS slice for a group placement");
        mine[2 + 2 * g] = shard->group_lo[g];
        mine[3 + 2 * g] = shard->group_srs[g]->n;
    }
    HIP_CHECK(hipMemcpyAsync(ws->xchg.p, mine, sizeof(mine), hipMemcpyHostToDevice, ctx->stream));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    ex->allgather(ws->xchg.p, ws->xchg.p + REC, sizeof(mine), ctx->stream);
    std::vector<uint64_t> all(REC * (size_t)ex->world);
    HIP_CHECK(hipMemcpyAsync(all.data(), ws->xchg.p + REC, all.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    PLONK_REQUIRE(memcmp(&all[REC * ex->rank], mine, sizeof(mine)) == 0,
                  "prove: the all-gather does not return this rank's record at its rank index");
    uint64_t next = 0;
    for (uint32_t r = 0; r < ex->world; ++r) {
        PLONK_REQUIRE(all[REC * r] == next, "prove: the ranks' SRS slices do not tile the point range (gap or overlap)");
        next += all[REC * r + 1];
    }
    shard->global_end = next;
    shard->slices = all;
    for (uint32_t g = 0; g < 2; ++g) {
        const uint32_t b = g + 2;
        uint32_t have = 0;
        for (uint32_t r = 0; r < ex->world; ++r) have += all[REC * r + 3 + 2 * g] != 0;
        if (have == 0) continue;
        PLONK_REQUIRE(have == ex->world, "prove: some ranks bring group slices and some do not");
        PLONK_REQUIRE(ex->world >= b, "prove: fewer ranks than commitments in a group placement");
        for (uint32_t k = 0; k < b; ++k) {
            uint64_t at = 0;
            for (uint32_t r = k; r < ex->world; r += b) {
                PLONK_REQUIRE(all[REC * r + 2 + 2 * g] == at, "prove: the group slices of the ranks of one commitment do not tile the point range");
                at += all[REC * r + 3 + 2 * g];
            }
            PLONK_REQUIRE(at == next, "prove: the group slices of the ranks of one commitment do not cover the point range");
        }
    }
}

// ---- rounds 4-5 of a proof shared by several ranks, by coefficient range.  The evaluations at zeta are sums over the
//      ranks' SRS slices of partial sums sum_{i in slice} c_i zeta^i.  An opening quotient q = N / (X - z) is only needed
//      where this rank commits it, q_i = sum_{j > i} N_j z^(j - i - 1) for i in its slice [lo, hi) of that commitment: the
//      terms with j < hi come from the rank's own coefficients of N, the rest is z^(hi - i - 1) T with
//      T = z^(-hi) sum_{j >= hi} N_j z^j -- partial evaluations of N again, over the slices above.  Dividing the
//      polynomial (N_lo, ..., N_(hi-1), T) by X - z yields exactly q_lo .. q_(hi-1), so the kernels are those of one GPU;
//      what crosses the ranks is one record of a few field elements per round. ----
struct OpeningSlice {
    size_t lo, hi;     // quotient coefficients the rank commits (what commit_range hands its MSM)
    size_t own_hi;     // numerator coefficients [lo, own_hi) the rank works on: its quotient range, the top slice the leading one too
    bool any, top;
};
// the slice of rank r for commitment `index` of the round of two openings, from the table the coverage check gathered
static OpeningSlice opening_slice(const ShardInfo* shard, uint32_t r, int index, size_t q_len) {
    const uint64_t* rec = &shard->slices[6 * (size_t)r];
    const bool grouped = rec[3] != 0;
    const size_t first = grouped ? rec[2] : rec[0], points = grouped ? rec[3] : rec[1];
    const bool mine = !grouped || r % 2u == (uint32_t)index;
    OpeningSlice s;
    s.lo = std::min(first, q_len);
    s.hi = mine ? std::min(first + points, q_len) : s.lo;
    s.any = s.hi > s.lo;
    s.top = s.any && s.hi == q_len;
    s.own_hi = s.top ? q_len + 1 : s.hi;
    return s;
}

// T of the comment above for this rank's slice: `sums` holds every rank's sum_{j in its [lo, own_hi)} N_j z^j at
// element at + r * stride
static fr_h opening_carry(const ShardInfo* shard, int index, size_t q_len, const fr_h* sums, size_t at, size_t stride,
                          const OpeningSlice& mine, const fr_h& z_inv) {
    fr_h t = Fr::zero();
    for (uint32_t r = 0; r < shard->ex->world; ++r) {
        const OpeningSlice o = opening_slice(shard, r, index, q_len);
        if (!o.any || o.lo < mine.hi) continue;   // an empty slice, or one that is not above this one
        const fr_h& e = sums[at + (size_t)r * stride];
        PLONK_REQUIRE(!Fr::geq_p(e), "prove: a rank's partial sum is not a reduced field element");
        t = Fr::add(t, e);
    }
    return Fr::mul(t, Fr::pow_u64(z_inv, mine.hi));
}

// the numerator coefficients at hand in num[lo, own_hi) -> the slice's quotient coefficients in q[lo, hi)
static void divide_slice(Ctx* ctx, ProverWorkspace* ws, fr_t* num, const OpeningSlice& os, const fr_h& carry, const fr_h& z,
                         const fr_h& z_inv, const PowTab& z_tab, const PowTab& z_inv_tab, fr_t* q) {
    if (!os.any) return;
    size_t len = os.own_hi - os.lo;
    if (!os.top) {   // the sum of everything above, as one more coefficient on top
        HIP_CHECK(hipMemsetAsync(num + os.hi, 0, sizeof(fr_t), ctx->stream));
        poly_poke(ctx, num, (uint32_t)os.hi, carry, false);
        ++len;
    }
    poly_div_linear(ctx, num + os.lo, (uint32_t)len, z, z_inv, z_tab, z_inv_tab, ws->tmp.p, q + os.lo, ws->totals);
}

host::G1AffineH commit_checked(Ctx* ctx, const Srs* srs, const fr_t* coeffs_dev, size_t len) {
    ProverWorkspace* ws = prover_workspace(ctx);
    ws->flag.ensure(4);
    HIP_CHECK(hipMemsetAsync(ws->flag.p, 0, 16, ctx->stream));
    host::G1AffineH r = commit_dev(ctx, srs, coeffs_dev, len, ws->flag.p);
    uint32_t f = 0;
    HIP_CHECK(hipMemcpy(&f, ws->flag.p, 4, hipMemcpyDeviceToHost));
    PLONK_REQUIRE(f == 0, "commit: polynomial degree exceeds SRS max degree");
    return r;
}

// ---- the stand-alone polynomial / opening seam: Polynomial::evaluate (polynomial.rs:103-113),
//      Polynomial::div_by_linear (:188-204), OpeningProof::create (kzg.rs:262-276) and BatchOpening::create
//      (kzg.rs:322-341) on the kernels rounds 4 and 5 of the prover run ----
host::fr_h poly_evaluate(Ctx* ctx, const fr_t* coeffs_dev, size_t len, const fr_h& z) {
    if (len == 0) return Fr::zero();
    PLONK_REQUIRE(len < ((size_t)1 << 31), "evaluate: polynomial too long");
    ProverWorkspace* ws = prover_workspace(ctx);
    ws->evals.ensure(8);
    PowTab tab;
    pow_tables(ctx, &z, 1, (uint32_t)len, ws->powtab, &tab);
    EvalArgs ea;
    memset(&ea, 0, sizeof(ea));
    ea.coeffs[0] = coeffs_dev; ea.len[0] = (uint32_t)len; ea.lo[0] = tab.lo; ea.hi[0] = tab.hi;
    poly_eval_batch(ctx, ea, 1, ws->evals.p, ws->partial);
    fr_h y;
    HIP_CHECK(hipMemcpyAsync(&y, ws->evals.p, sizeof(fr_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return y;
}

// q_out_dev[0 .. len-1) = the quotient of the division by (X - z); the remainder f(z) is dropped (polynomial.rs:188-204).
void poly_div_by_linear(Ctx* ctx, const fr_t* coeffs_dev, size_t len, const fr_h& z, fr_t* q_out_dev) {
    if (len < 2) return;   // a constant (or the zero polynomial) has the empty quotient
    PLONK_REQUIRE(len < ((size_t)1 << 31), "div_by_linear: polynomial too long");
    ProverWorkspace* ws = prover_workspace(ctx);
    ws->tmp.ensure(len);
    if (Fr::is_zero(z)) {
        poly_div_linear(ctx, coeffs_dev, (uint32_t)len, z, z, PowTab{nullptr, nullptr}, PowTab{nullptr, nullptr}, ws->tmp.p, q_out_dev, ws->totals);
        return;
    }
    const fr_h z_inv = Fr::inv(z);
    const fr_h pts[2] = {z, z_inv};
    PowTab tabs[2];
    pow_tables(ctx, pts, 2, (uint32_t)len, ws->powtab, tabs);
    poly_div_linear(ctx, coeffs_dev, (uint32_t)len, z, z_inv, tabs[0], tabs[1], ws->tmp.p, q_out_dev, ws->totals);
}

// (y, [q]_1) with y = f(z), q = (f - y) / (X - z): subtracting y from the constant term changes the remainder of
// the division (to zero), not its quotient, so f itself is divided.
host::G1AffineH kzg_open(Ctx* ctx, const Srs* srs, const fr_t* coeffs_dev, size_t len, const fr_h& z, fr_h* y_out) {
    *y_out = poly_evaluate(ctx, coeffs_dev, len, z);
    if (len < 2) return commit_checked(ctx, srs, coeffs_dev, 0);
    ProverWorkspace* ws = prover_workspace(ctx);
    ws->wq.ensure(len);
    poly_div_by_linear(ctx, coeffs_dev, len, z, ws->wq.p);
    return commit_checked(ctx, srs, ws->wq.p, len - 1);
}

// evaluations y_i = f_i(z) and the opening of g = sum_i gamma^i f_i at z (kzg.rs:322-341)
host::G1AffineH kzg_open_batch(Ctx* ctx, const Srs* srs, const fr_t* const* polys_dev, const size_t* lens, size_t count,
                               const fr_h& z, const fr_h& gamma, fr_h* evals_out) {
    ProverWorkspace* ws = prover_workspace(ctx);
    size_t max_len = 0;
    for (size_t k = 0; k < count; ++k) {
        PLONK_REQUIRE(lens[k] < ((size_t)1 << 31), "open batch: polynomial too long");
        PLONK_REQUIRE(lens[k] == 0 || polys_dev[k] != nullptr, "open batch: null polynomial");
        max_len = std::max(max_len, lens[k]);
    }
    ws->evals.ensure(8);
    if (max_len) {
        PowTab tab;
        pow_tables(ctx, &z, 1, (uint32_t)max_len, ws->powtab, &tab);
        for (size_t base = 0; base < count; base += EVAL_MAX) {
            const uint32_t m = (uint32_t)std::min<size_t>(EVAL_MAX, count - base);
            EvalArgs ea;
            memset(&ea, 0, sizeof(ea));
            for (uint32_t k = 0; k < m; ++k) {
                ea.coeffs[k] = polys_dev[base + k]; ea.len[k] = (uint32_t)lens[base + k]; ea.lo[k] = tab.lo; ea.hi[k] = tab.hi;
            }
            poly_eval_batch(ctx, ea, m, ws->evals.p, ws->partial);
            HIP_CHECK(hipMemcpyAsync(evals_out + base, ws->evals.p, m * sizeof(fr_t), hipMemcpyDeviceToHost, ctx->stream));
            HIP_CHECK(hipStreamSynchronize(ctx->stream));
        }
    } else {
        for (size_t k = 0; k < count; ++k) evals_out[k] = Fr::zero();
    }
    // g = sum gamma^i f_i, at most LINCOMB_MAX - 1 new terms per pass on top of the running sum
    ws->wnum.ensure(std::max<size_t>(max_len, 1));
    fr_h gp = Fr::one();
    bool have = false;
    for (size_t base = 0; base < count || !have;) {
        LincombArgs la;
        memset(&la, 0, sizeof(la));
        uint32_t k = 0;
        if (have) {
            la.p[k] = ws->wnum.p; la.len[k] = (uint32_t)max_len; la.s[k] = lincomb_scalar(Fr::one()); ++k;
        }
        while (base < count && k < LINCOMB_MAX) {
            la.p[k] = polys_dev[base]; la.len[k] = (uint32_t)lens[base]; la.s[k] = lincomb_scalar(gp); ++k;
            gp = Fr::mul(gp, gamma);
            ++base;
        }
        la.count = k;
        if (max_len) poly_lincomb(ctx, la, ws->wnum.p, (uint32_t)max_len);
        have = true;
    }
    fr_h y;
    return kzg_open(ctx, srs, ws->wnum.p, max_len, z, &y);
}

// The kernels that detect a zero denominator (round 2) or an over-long polynomial (rounds 3, 5) raise a flag and
// carry on with harmless values; the flag is read once, at the end of the proof, instead of with a stream
// synchronisation after every round.
static void check_flag(Ctx* ctx, ProverWorkspace* ws) {
    uint32_t f = 0;
    HIP_CHECK(hipMemcpyAsync(&f, ws->flag.p, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    PLONK_REQUIRE(!(f & 1u), "prove: zero denominator in the permutation accumulator");
    PLONK_REQUIRE(!(f & 2u), "prove: polynomial degree exceeds its bound (t above 3n+5, or a commitment above the SRS)");
}

// Host (pageable) vector -> device through the workspace's pinned staging area: a few host threads copy pieces into
// pinned memory and every piece goes to the device as soon as it is there (56 GB/s instead of the 27 GB/s a
// pageable hipMemcpyAsync reaches on this host).  The call returns when the last piece is enqueued; `stage_off`
// keeps the vectors of one proof apart, so the host never rewrites a piece the DMA engine is still reading.
static void upload_staged(Ctx* ctx, ProverWorkspace* ws, fr_t* dst_dev, const void* src_host, size_t bytes, size_t stage_off) {
    constexpr size_t PIECE = (size_t)4 << 20;
    constexpr int THREADS = 4;
    if (bytes < 2 * PIECE || !ws->stage || stage_off + bytes > ws->stage_bytes) {   // small vectors: the plain copy is as fast; very large ones are not staged
        HIP_CHECK(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
        return;
    }
    const size_t pieces = (bytes + PIECE - 1) / PIECE;
    unsigned char* stage = ws->stage + stage_off;
    const unsigned char* src = static_cast<const unsigned char*>(src_host);
    std::vector<std::future<void>> done(pieces);
    size_t started = 0;
    auto start = [&](size_t j) {
        const size_t off = j * PIECE, len = std::min(PIECE, bytes - off);
        done[j] = std::async(std::launch::async, [=] { memcpy(stage + off, src + off, len); });
    };
    for (; started < pieces && started < (size_t)THREADS; ++started) start(started);
    for (size_t j = 0; j < pieces; ++j) {
        done[j].get();
        if (started < pieces) start(started++);
        const size_t off = j * PIECE, len = std::min(PIECE, bytes - off);
        HIP_CHECK(hipMemcpyAsync(reinterpret_cast<unsigned char*>(dst_dev) + off, stage + off, len, hipMemcpyHostToDevice, ctx->stream));
    }
}

void prove_3wire(Ctx* ctx, const Pk* pk, const Srs* srs, const plonk_fr* a_ev, const plonk_fr* b_ev, const plonk_fr* c_ev,
                 bool evals_on_device, const plonk_fr* pub, size_t n_pub, const plonk_fr* blinding, plonk_proof* out,
                 const ShardInfo* shard_in) {
    const size_t n = pk->n, n4 = 4 * n, ns = n + 8;
    const uint32_t log_n = pk->log_n;
    ProverWorkspace* ws = prover_workspace(ctx);
    ws->w4.ensure(5 * n4); ws->t.ensure(n4); ws->ev.ensure(3 * n); ws->co.ensure(4 * ns);
    ws->scr.ensure(n); ws->tp.ensure(3 * ns);
    ws->wnum.ensure(ns); ws->wq.ensure(ns); ws->wq2.ensure(ns); ws->tmp.ensure(ns); ws->evals.ensure(8); ws->flag.ensure(4);
    hipStream_t st = ctx->stream;
    // (up to 2^21 gates: 192 MiB of pinned host memory; larger witnesses go through the plain pageable copy)
    if (!evals_on_device && 3 * n * sizeof(fr_t) <= ((size_t)192 << 20) && ws->stage_bytes < 3 * n * sizeof(fr_t)) {
        // the previous call's copies out of the old area are complete: every call ends with a stream synchronisation
        if (ws->stage) (void)hipHostFree(ws->stage);
        ws->stage = nullptr;
        ws->stage_bytes = 0;
        HIP_CHECK(hipHostMalloc((void**)&ws->stage, 3 * n * sizeof(fr_t), hipHostMallocDefault));
        ws->stage_bytes = 3 * n * sizeof(fr_t);
    }
    ShardInfo shard_local;
    const ShardInfo* shard = nullptr;
    if (sharded(shard_in)) {
        PLONK_REQUIRE(shard_in->ex->device == ctx->device, "prove: the communicator belongs to another device");
        shard_local = *shard_in;
        shard_check_coverage(ctx, ws, srs, &shard_local);
        shard = &shard_local;
    }
    const PointShard ps = point_shard(pk, shard);
    fr_t* SL[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (ps.on) {
        point_shard_tables(ctx, ws, pk, ps);
        for (int k = 0; k < 6; ++k) SL[k] = ws->sl.p + k * ps.M;
    }
    fr_h bl[11];
    for (int i = 0; i < 11; ++i) {
        memcpy(bl[i].l, blinding[i].l, 32);
        PLONK_REQUIRE(!Fr::geq_p(bl[i]), "prove: blinding scalar not reduced");
    }
    std::vector<fr_h> pi(n_pub);
    for (size_t i = 0; i < n_pub; ++i) {
        memcpy(pi[i].l, pub[i].l, 32);
        PLONK_REQUIRE(!Fr::geq_p(pi[i]), "prove: public input not reduced");
    }
    fr_t* W[5];
    for (int k = 0; k < 5; ++k) W[k] = ws->w4.p + k * n4;
    fr_t* CO[4];
    for (int k = 0; k < 4; ++k) CO[k] = ws->co.p + k * ns;
    fr_t* TP[3];
    for (int k = 0; k < 3; ++k) TP[k] = ws->tp.p + k * ns;

    host::Transcript tr("PLONK-v1");  // prover.rs:64
    for (size_t i = 0; i < n_pub; ++i) tr.absorb_fr("pi_" + std::to_string(i), pi[i]);  // :69-71

    HIP_CHECK(hipMemsetAsync(ws->flag.p, 0, 16, st));
    // a, b, c, z: [0, n) is overwritten by the evaluations and the blinding is added onto [n, n + 3); the coset
    // transforms are told the coefficient count and never read past it, so nothing else has to be zero
    for (int k = 0; k < 4; ++k) HIP_CHECK(hipMemsetAsync(W[k] + n, 0, 4 * sizeof(fr_t), st));
    HIP_CHECK(hipMemsetAsync(ws->co.p, 0, 4 * ns * sizeof(fr_t), st));
    HIP_CHECK(hipMemsetAsync(ws->tp.p, 0, 3 * ns * sizeof(fr_t), st));

    // ---- round 1 (prover.rs:181-244) ----
    // wire evaluations: a resident witness is read where it lies, a host witness is staged in ws->ev
    const plonk_fr* evs[3] = {a_ev, b_ev, c_ev};
    const fr_t* EV[3];
    host::G1AffineH comm[9];
    // the wires' evaluations on the 4n coset (round 3 reads them; they depend on no challenge) are
    // computed while the last commitment's reduction drains
    const MsmFiller coset_abc = [&](hipStream_t s) {
        if (!ps.on) {
            ntt_run_on(ctx, s, W[0], log_n + 2, NTT_COSET, 3, n4, n + 2);
            return;
        }
        for (int k = 0; k < 3; ++k) class_evaluate(ctx, s, W[k], n + 2, ps, ps.s_q, ps.c_q, SL[k]);
    };
    // wire by wire: the device transforms and commits one wire while the next is prepared -- for a
    // host witness the pageable upload keeps the host busy staging meanwhile
    MsmBatch r1;
    for (int k = 0; k < 3; ++k) {
        if (evals_on_device) {
            EV[k] = reinterpret_cast<const fr_t*>(evs[k]);
        } else {
            upload_staged(ctx, ws, ws->ev.p + k * n, evs[k], n * sizeof(fr_t), (size_t)k * n * sizeof(fr_t));
            EV[k] = ws->ev.p + k * n;
        }
        poly_copy(ctx, W[k], EV[k], n);
        ntt_run(ctx, W[k], log_n, NTT_INVERSE, 1, 0);
        const fr_h b2[2] = {bl[2 * k + 1], bl[2 * k]};  // blind poly [b2, b1] (:205-207)
        poly_blind(ctx, W[k], (uint32_t)n, b2, 2);
        poly_copy(ctx, CO[k], W[k], n + 2);
        commit_add(ctx, srs, r1, CO[k], n + 2, ws->flag.p, shard, 3);
    }
    commit_finish(ctx, r1, shard, comm, coset_abc);
    tr.absorb_g1("a_comm", comm[0]);
    tr.absorb_g1("b_comm", comm[1]);
    tr.absorb_g1("c_comm", comm[2]);
    const fr_h beta = tr.squeeze_challenge("beta");
    const fr_h gamma = tr.squeeze_challenge("gamma");

    // ---- round 2 (prover.rs:247-336) ----
    {
        PermArgs p;
        p.a = EV[0]; p.b = EV[1]; p.c = EV[2];
        p.s1 = pk->sig_ev.p; p.s2 = pk->sig_ev.p + n; p.s3 = pk->sig_ev.p + 2 * n;
        p.w_lo = pk->w_lo.p; p.w_hi = pk->w_hi.p;
        p.beta = to_dev(beta); p.gamma = to_dev(gamma); p.k1 = to_dev(pk->k1); p.k2 = to_dev(pk->k2);
        p.omega = to_dev(pk->omega);
        p.n = (uint32_t)n; p.log_l = pk->log_lw;
        // only i < n-1 enter the product (:271); the last ratio is never read by the exclusive scan
        perm_grand_product(ctx, p, W[3], ws->scr.p, ws->flag.p, ws->totals);
    }
    ntt_run(ctx, W[3], log_n, NTT_INVERSE, 1, 0);
    {
        const fr_h b3[3] = {bl[8], bl[7], bl[6]};  // [b9, b8, b7] (:319)
        poly_blind(ctx, W[3], (uint32_t)n, b3, 3);
        poly_copy(ctx, CO[3], W[3], n + 3);
    }
    comm[3] = commit_dev(ctx, srs, CO[3], n + 3, ws->flag.p, shard, [&](hipStream_t s) {
        if (!ps.on) {
            ntt_run_on(ctx, s, W[3], log_n + 2, NTT_COSET, 1, 0, n + 3);
            return;
        }
        class_evaluate(ctx, s, W[3], n + 3, ps, ps.s_q, ps.c_q, SL[3]);
        if (ps.q2 != ps.q) class_evaluate(ctx, s, W[3], n + 3, ps, ps.s_q2, ps.c_q2, SL[4]);   // z(w X): the class of i + 4
    });
    tr.absorb_g1("z_comm", comm[3]);
    const fr_h alpha = tr.squeeze_challenge("alpha");

    // ---- round 3 (prover.rs:339-574) ----
    bool has_pi = false;
    if (n_pub && !pk->pi_positions.empty()) {  // prover.rs:840-857
        std::vector<fr_h> pe(n, Fr::zero());
        for (size_t i = 0; i < pk->pi_positions.size() && i < n_pub; ++i) pe[pk->pi_positions[i]] = Fr::neg(pi[i]);
        HIP_CHECK(hipMemcpyAsync(W[4], pe.data(), n * sizeof(fr_t), hipMemcpyHostToDevice, st));
        HIP_CHECK(hipStreamSynchronize(st));
        ntt_run(ctx, W[4], log_n, NTT_INVERSE, 1, 0);
        has_pi = true;
    }
    if (!ps.on) {
        if (has_pi) ntt_run(ctx, W[4], log_n + 2, NTT_COSET, 1, 0, n);
        quotient_run(ctx, pk, W[0], W[1], W[2], W[3], has_pi ? W[4] : nullptr, alpha, beta, gamma, ws->t.p);
        ntt_run(ctx, ws->t.p, log_n + 2, NTT_COSET | NTT_INVERSE, 1, 0);
    } else {
        if (has_pi) class_evaluate(ctx, st, W[4], n, ps, ps.s_q, ps.c_q, SL[5]);
        QuotientSlice qs;
        qs.log_g = ps.log_g;
        qs.q = ps.q;
        qs.ext = ws->ext_slice.p;
        qs.zw = ps.q2 != ps.q ? SL[4] : nullptr;
        qs.zw_off = ps.zw_off;
        quotient_run(ctx, pk, SL[0], SL[1], SL[2], SL[3], has_pi ? SL[5] : nullptr, alpha, beta, gamma, ws->tq.p, qs);
        ntt_run_shift(ctx, ws->tq.p, ps.log_m, NTT_COSET | NTT_INVERSE, ps.s_q, 1, 0);   // F_q = t mod (X^M - c_q)
        shard->ex->allgather(ws->tq.p, ws->fall.p, ps.M * sizeof(fr_t), st);
        {
            const dim3 grid((uint32_t)((ps.M + 255) / 256)), block(256);
            switch (ps.log_g) {
                case 1: hipLaunchKernelGGL(recombine_kernel<1>, grid, block, 0, st, ws->fall.p, (uint32_t)ps.M, ws->coef.p, ws->t.p); break;
                case 2: hipLaunchKernelGGL(recombine_kernel<2>, grid, block, 0, st, ws->fall.p, (uint32_t)ps.M, ws->coef.p, ws->t.p); break;
                default: hipLaunchKernelGGL(recombine_kernel<3>, grid, block, 0, st, ws->fall.p, (uint32_t)ps.M, ws->coef.p, ws->t.p); break;
            }
        }
        HIP_CHECK(hipGetLastError());
    }
    poly_nonzero_check(ctx, ws->t.p, (uint32_t)(3 * n + 6), (uint32_t)n4, ws->flag.p);  // :506-511
    poly_copy(ctx, TP[0], ws->t.p, n);
    poly_copy(ctx, TP[1], ws->t.p + n, n);
    poly_copy(ctx, TP[2], ws->t.p + 2 * n, std::min(n + 6, n4 - 2 * n));
    poly_poke(ctx, TP[0], (uint32_t)n, bl[9], false);   // t_lo  += b10 X^n   (:540-543)
    poly_poke(ctx, TP[1], 0, bl[9], true);              // t_mid -= b10
    poly_poke(ctx, TP[1], (uint32_t)n, bl[10], false);  // t_mid += b11 X^n
    poly_poke(ctx, TP[2], 0, bl[10], true);             // t_hi  -= b11
    {
        const fr_t* cp[3] = {TP[0], TP[1], TP[2]};
        const size_t cl[3] = {n + 1, n + 1, n + 6};
        commit_batch(ctx, srs, cp, cl, 3, ws->flag.p, shard, comm + 4);
    }
    tr.absorb_g1("t_lo_comm", comm[4]);
    tr.absorb_g1("t_mid_comm", comm[5]);
    tr.absorb_g1("t_hi_comm", comm[6]);
    const fr_h zeta = tr.squeeze_challenge("zeta");

    // ---- round 4 (prover.rs:577-641) ----
    const fr_h zeta_omega = Fr::mul(zeta, pk->omega);
    // W_{zeta omega} = (z - z(zeta omega)) / (X - zeta omega) (:815-816) needs zeta alone -- the value
    // subtracted from the constant term changes the remainder, not the quotient -- so its division and the
    // sort of its commitment start here and run beside the evaluations of round 4.  Its bucket kernel is held
    // back until round 5 has enqueued the division for W_zeta: a bucket kernel fills every wave slot of the
    // chip, and the short dependent kernels of rounds 4-5 (evaluations, linear combination, division) would
    // crawl beside it.  It enters the proof in the reference's order.
    MsmBatch openings;
    const fr_h zeta_inv = Fr::inv(zeta), zeta_omega_inv = Fr::inv(zeta_omega);
    PowTab tabs[4];   // zeta, 1/zeta, zeta omega, 1/(zeta omega)
    {
        const fr_h pts[4] = {zeta, zeta_inv, zeta_omega, zeta_omega_inv};
        pow_tables(ctx, pts, 4, (uint32_t)n + 8, ws->powtab, tabs);
    }
    HIP_CHECK(hipMemsetAsync(ws->wq2.p, 0, ns * sizeof(fr_t), st));
    const bool split = sharded(shard) && shard->ex->world > 1;   // rounds 4-5 by coefficient range (see opening_slice)
    fr_h e6[6];
    const fr_t* eval_polys[6] = {CO[0], CO[1], CO[2], pk->coef.p + 5 * n, pk->coef.p + 6 * n, CO[3]};
    const size_t eval_lens[6] = {n + 2, n + 2, n + 2, n, n, n + 3};
    if (!split) {
        poly_div_linear(ctx, CO[3], (uint32_t)n + 3, zeta_omega, zeta_omega_inv, tabs[2], tabs[3], ws->tmp.p, ws->wq2.p, ws->totals);
        commit_add(ctx, srs, openings, ws->wq2.p, n + 2, ws->flag.p, shard, 2, /*hold_buckets=*/true);
        EvalArgs ea;
        memset(&ea, 0, sizeof(ea));
        for (int k = 0; k < 6; ++k) {
            ea.coeffs[k] = eval_polys[k];
            ea.len[k] = (uint32_t)eval_lens[k];
            ea.lo[k] = tabs[k == 5 ? 2 : 0].lo;
            ea.hi[k] = tabs[k == 5 ? 2 : 0].hi;
        }
        poly_eval_batch(ctx, ea, 6, ws->evals.p, ws->partial);
        HIP_CHECK(hipMemcpyAsync(e6, ws->evals.p, 6 * sizeof(fr_t), hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
    } else {
        Exchange* ex = shard->ex;
        const OpeningSlice os = opening_slice(shard, ex->rank, 0, n + 2);
        // the six evaluations over this rank's SRS slice, and z's part of W_{zeta omega}'s numerator over its slice of that
        // commitment: sums of c_i x^i with the exponents of the whole polynomial, so the ranks' sums simply add up
        EvalArgs ea;
        memset(&ea, 0, sizeof(ea));
        for (int k = 0; k < 7; ++k) {
            const fr_t* poly = k < 6 ? eval_polys[k] : CO[3];
            const size_t len = k < 6 ? eval_lens[k] : n + 3;
            const size_t lo = k < 6 ? std::min(shard->lo, len) : os.lo;
            const size_t hi = k < 6 ? std::min(shard->lo + srs->n, len) : os.own_hi;
            ea.coeffs[k] = poly + lo;
            ea.first[k] = (uint32_t)lo;
            ea.len[k] = (uint32_t)(hi > lo ? hi - lo : 0);
            ea.lo[k] = tabs[k >= 5 ? 2 : 0].lo;
            ea.hi[k] = tabs[k >= 5 ? 2 : 0].hi;
        }
        poly_eval_batch(ctx, ea, 7, ws->evals.p, ws->partial);
        constexpr size_t REC = 8;   // seven sums and a spare element per rank
        ws->gath.ensure(REC * (size_t)ex->world);
        HIP_CHECK(hipMemsetAsync(ws->evals.p + 7, 0, sizeof(fr_t), st));
        ex->allgather(ws->evals.p, ws->gath.p, REC * sizeof(fr_t), st);
        std::vector<fr_h> all(REC * (size_t)ex->world);
        HIP_CHECK(hipMemcpyAsync(all.data(), ws->gath.p, all.size() * sizeof(fr_t), hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        for (int k = 0; k < 6; ++k) {
            fr_h sum = Fr::zero();
            for (uint32_t r = 0; r < ex->world; ++r) {
                PLONK_REQUIRE(!Fr::geq_p(all[REC * r + k]), "prove: a rank's partial evaluation is not a reduced field element");
                sum = Fr::add(sum, all[REC * r + k]);
            }
            e6[k] = sum;
        }
        const fr_h carry = os.any && !os.top ? opening_carry(shard, 0, n + 2, all.data(), 6, REC, os, zeta_omega_inv) : Fr::zero();
        if (os.any) poly_copy(ctx, ws->wnum.p + os.lo, CO[3] + os.lo, os.own_hi - os.lo);   // wnum: free until round 5 builds its numerator
        divide_slice(ctx, ws, ws->wnum.p, os, carry, zeta_omega, zeta_omega_inv, tabs[2], tabs[3], ws->wq2.p);
        commit_add(ctx, srs, openings, ws->wq2.p, n + 2, ws->flag.p, shard, 2, /*hold_buckets=*/true);
    }
    const fr_h a_bar = e6[0], b_bar = e6[1], c_bar = e6[2], s1_bar = e6[3], s2_bar = e6[4], zw_bar = e6[5];
    tr.absorb_fr("a_zeta", a_bar);
    tr.absorb_fr("b_zeta", b_bar);
    tr.absorb_fr("c_zeta", c_bar);
    tr.absorb_fr("s_sigma1_zeta", s1_bar);
    tr.absorb_fr("s_sigma2_zeta", s2_bar);
    tr.absorb_fr("z_zeta_omega", zw_bar);
    const fr_h v = tr.squeeze_challenge("v");

    // ---- round 5 (prover.rs:644-837) ----
    {
        const fr_h zeta_n = Fr::pow_u64(zeta, n);
        const fr_h zh_zeta = Fr::sub(zeta_n, Fr::one());                                                  // :697
        const fr_h l1_den = Fr::mul(Fr::from_u64(n), Fr::sub(zeta, Fr::one()));
        PLONK_REQUIRE(!Fr::is_zero(l1_den), "prove: zeta = 1");                                           // :700 unwrap
        const fr_h l1_zeta = Fr::mul(zh_zeta, Fr::inv(l1_den));
        const fr_h bz = Fr::mul(beta, zeta);
        fr_h pc1 = Fr::mul(alpha, Fr::add(Fr::add(a_bar, bz), gamma));                                   // :749-752
        pc1 = Fr::mul(pc1, Fr::add(Fr::add(b_bar, Fr::mul(pk->k1, bz)), gamma));
        pc1 = Fr::mul(pc1, Fr::add(Fr::add(c_bar, Fr::mul(pk->k2, bz)), gamma));
        fr_h pc2 = Fr::mul(alpha, Fr::add(Fr::add(a_bar, Fr::mul(beta, s1_bar)), gamma));                // :755-760
        pc2 = Fr::mul(pc2, Fr::add(Fr::add(b_bar, Fr::mul(beta, s2_bar)), gamma));
        pc2 = Fr::neg(Fr::mul(Fr::mul(pc2, beta), zw_bar));
        const fr_h z_coeff = Fr::add(pc1, Fr::mul(Fr::mul(alpha, alpha), l1_zeta));                      // :762,:766
        const fr_h neg_zh = Fr::neg(zh_zeta);
        const fr_h v2 = Fr::mul(v, v), v3 = Fr::mul(v2, v), v4 = Fr::mul(v3, v), v5 = Fr::mul(v4, v);
        LincombArgs la;
        memset(&la, 0, sizeof(la));
        uint32_t k = 0;
        auto term = [&](const fr_t* p, size_t len, const fr_h& s) {
            la.p[k] = p; la.len[k] = (uint32_t)len; la.s[k] = lincomb_scalar(s); ++k;
        };
        term(pk->coef.p + 0 * n, n, Fr::mul(a_bar, b_bar));  // q_M  (:742)
        term(pk->coef.p + 1 * n, n, a_bar);                  // q_L
        term(pk->coef.p + 2 * n, n, b_bar);                  // q_R
        term(pk->coef.p + 3 * n, n, c_bar);                  // q_O
        term(pk->coef.p + 4 * n, n, Fr::one());              // q_C
        term(CO[3], n + 3, z_coeff);                         // z
        term(pk->coef.p + 7 * n, n, pc2);                    // S_sigma3 (:763)
        term(TP[0], n + 1, neg_zh);                          // -Z_H(zeta) (t_lo + zeta^n t_mid + zeta^2n t_hi) (:776-784)
        term(TP[1], n + 1, Fr::mul(neg_zh, zeta_n));
        term(TP[2], n + 6, Fr::mul(neg_zh, Fr::mul(zeta_n, zeta_n)));
        term(CO[0], n + 2, v);                               // :800-807
        term(CO[1], n + 2, v2);
        term(CO[2], n + 2, v3);
        term(pk->coef.p + 5 * n, n, v4);
        term(pk->coef.p + 6 * n, n, v5);
        la.count = k;
        const uint32_t wlen = (uint32_t)n + 6;
        // (r + v (a - a_bar) + ...) / (X - zeta) (:800-809): the constants subtracted from the numerator (r(zeta), v a_bar,
        // ...) only change the remainder of the division, which is dropped -- the numerator is divided as it is
        HIP_CHECK(hipMemsetAsync(ws->wq.p, 0, ns * sizeof(fr_t), st));
        if (!split) {
            poly_lincomb(ctx, la, ws->wnum.p, wlen);
            poly_div_linear(ctx, ws->wnum.p, wlen, zeta, zeta_inv, tabs[0], tabs[1], ws->tmp.p, ws->wq.p, ws->totals);   // :809
        } else {
            Exchange* ex = shard->ex;
            const OpeningSlice os = opening_slice(shard, ex->rank, 1, wlen - 1);
            const size_t cnt = os.own_hi - os.lo;
            if (os.any) {
                for (uint32_t t = 0; t < la.count; ++t) {   // the numerator on [lo, own_hi) only
                    const size_t len = la.len[t];
                    la.p[t] += os.lo;
                    la.len[t] = (uint32_t)(len > os.lo ? std::min(len, os.own_hi) - os.lo : 0);
                }
                poly_lincomb(ctx, la, ws->wnum.p + os.lo, (uint32_t)cnt);
            }
            EvalArgs ea;
            memset(&ea, 0, sizeof(ea));
            ea.coeffs[0] = ws->wnum.p + os.lo;
            ea.first[0] = (uint32_t)os.lo;
            ea.len[0] = (uint32_t)(os.any ? cnt : 0);
            ea.lo[0] = tabs[0].lo;
            ea.hi[0] = tabs[0].hi;
            poly_eval_batch(ctx, ea, 1, ws->evals.p, ws->partial);
            ws->gath.ensure(ex->world);
            ex->allgather(ws->evals.p, ws->gath.p, sizeof(fr_t), st);
            std::vector<fr_h> all(ex->world);
            HIP_CHECK(hipMemcpyAsync(all.data(), ws->gath.p, all.size() * sizeof(fr_t), hipMemcpyDeviceToHost, st));
            HIP_CHECK(hipStreamSynchronize(st));
            const fr_h carry = os.any && !os.top ? opening_carry(shard, 1, wlen - 1, all.data(), 0, 1, os, zeta_inv) : Fr::zero();
            divide_slice(ctx, ws, ws->wnum.p, os, carry, zeta, zeta_inv, tabs[0], tabs[1], ws->wq.p);
        }
        msm_batch_release(ctx, openings);   // W_{zeta omega}'s bucket kernel; W_zeta's sort runs beside it
        commit_add(ctx, srs, openings, ws->wq.p, wlen - 1, ws->flag.p, shard, 2);
        host::G1AffineH w[2];
        commit_finish(ctx, openings, shard, w);
        comm[7] = w[1];   // W_zeta
        comm[8] = w[0];   // W_{zeta omega}
        check_flag(ctx, ws);
    }

    memset(out, 0, sizeof(*out));
    plonk_g1* g1s[9] = {&out->a_comm, &out->b_comm, &out->c_comm, &out->z_comm, &out->t_lo_comm,
                        &out->t_mid_comm, &out->t_hi_comm, &out->w_zeta_comm, &out->w_zeta_omega_comm};
    for (int k = 0; k < 9; ++k) g1_from_host(comm[k], g1s[k]);
    plonk_fr* frs[6] = {&out->a_eval, &out->b_eval, &out->c_eval, &out->s_sigma1_eval, &out->s_sigma2_eval, &out->z_omega_eval};
    for (int k = 0; k < 6; ++k) memcpy(frs[k]->l, e6[k].l, 32);
}

// Re-derives beta, gamma, alpha, zeta, v, u from a proof (the verifier's transcript replay,
// plonk/verifier.rs:51-92).
void proof_challenges(const plonk_proof* pr, const plonk_fr* pub, size_t n_pub, plonk_fr out[6]) {
    auto g = [](const plonk_g1& p) {
        host::G1AffineH a;
        memcpy(a.x.l, p.x, 48);
        memcpy(a.y.l, p.y, 48);
        a.inf = p.infinity != 0;
        return a;
    };
    auto f = [](const plonk_fr& x) {
        fr_h h;
        memcpy(h.l, x.l, 32);
        return h;
    };
    host::Transcript tr("PLONK-v1");
    for (size_t i = 0; i < n_pub; ++i) tr.absorb_fr("pi_" + std::to_string(i), f(pub[i]));
    tr.absorb_g1("a_comm", g(pr->a_comm)); tr.absorb_g1("b_comm", g(pr->b_comm)); tr.absorb_g1("c_comm", g(pr->c_comm));
    fr_h ch[6];
    ch[0] = tr.squeeze_challenge("beta");
    ch[1] = tr.squeeze_challenge("gamma");
    tr.absorb_g1("z_comm", g(pr->z_comm));
    ch[2] = tr.squeeze_challenge("alpha");
    tr.absorb_g1("t_lo_comm", g(pr->t_lo_comm)); tr.absorb_g1("t_mid_comm", g(pr->t_mid_comm));
    tr.absorb_g1("t_hi_comm", g(pr->t_hi_comm));
    ch[3] = tr.squeeze_challenge("zeta");
    tr.absorb_fr("a_zeta", f(pr->a_eval)); tr.absorb_fr("b_zeta", f(pr->b_eval)); tr.absorb_fr("c_zeta", f(pr->c_eval));
    tr.absorb_fr("s_sigma1_zeta", f(pr->s_sigma1_eval)); tr.absorb_fr("s_sigma2_zeta", f(pr->s_sigma2_eval));
    tr.absorb_fr("z_zeta_omega", f(pr->z_omega_eval));
    ch[4] = tr.squeeze_challenge("v");
    tr.absorb_g1("w_zeta_comm", g(pr->w_zeta_comm)); tr.absorb_g1("w_zeta_omega_comm", g(pr->w_zeta_omega_comm));
    ch[5] = tr.squeeze_challenge("u");
    for (int k = 0; k < 6; ++k) memcpy(out[k].l, ch[k].l, 32);
}

}  // namespace plonk
