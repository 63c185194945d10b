// Lazy Montgomery arithmetic in radix 2^29 for gfx950 ("x-form").
//
// On MI355X v_mad_u64_u32 issues at the same rate as a 32-bit add-with-carry or a 64-bit add
// (about 4.4 cycles per wave64 instruction per SIMD), so carry handling, not multiplication, is what
// a saturated 32-bit-limb Montgomery product pays for.  Here an element is L limbs of 29 bits
// (Fr: 9, Fq: 14) and the Montgomery radix is R' = 2^(29 L).  Limb products are below 2^58, so a
// whole column of a product plus its reduction terms (at most 2L <= 28 products) accumulates in one
// 64-bit register pair by v_mad_u64_u32 alone -- no carries, no register moves.  Values are kept
// "lazy": limbs normalised to 29 bits (the top limb holds the overflow), value below a small
// multiple of p; a product of inputs below 2^k p is below 2p whenever 2^(2k) p < R'.
// Same field elements as field.cuh; only the machine representation differs.  Conversions:
// x = xmul(repack(sat), KIN), sat = repack(canon(xmul(x, KOUT))).
#pragma once
#include "field.cuh"
#include "fieldx_consts.inc"

namespace plonk {

template <int L>
struct XE {
    uint32_t l[L];
};
template <class X>
using xe_t = XE<X::L>;

static constexpr uint32_t XMASK = (1u << 29) - 1u;

template <class X>
__device__ __forceinline__ xe_t<X> x_zero() {
    xe_t<X> r;
#pragma unroll
    for (int i = 0; i < X::L; ++i) r.l[i] = 0;
    return r;
}
template <class X>
__device__ __forceinline__ xe_t<X> x_one() {
    xe_t<X> r;
#pragma unroll
    for (int i = 0; i < X::L; ++i) r.l[i] = X::one(i);
    return r;
}
template <class X>
__device__ __forceinline__ bool x_limbs_zero(const xe_t<X>& a) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < X::L; ++i) o |= a.l[i];
    return o == 0;
}

// a*b/R' mod p, result < a*b/R' + p, limbs normalised.
template <class X>
__device__ __forceinline__ xe_t<X> x_mul(const xe_t<X>& a, const xe_t<X>& b) {
    constexpr int L = X::L;
    uint64_t c[2 * L];
#pragma unroll
    for (int k = 0; k < 2 * L; ++k) c[k] = 0;
#pragma unroll
    for (int i = 0; i < L; ++i)
#pragma unroll
        for (int j = 0; j < L; ++j) c[i + j] += (uint64_t)a.l[i] * b.l[j];
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

// (a b + c d)/R' mod p under ONE reduction (a sum of two products costs 3/4 of two x_mul):
// a, b, c normalised, d with limbs below 2^30 (an x_neg_lazy); result < (a b + c d)/R' + p, normalised.
// A column holds at most L (2^58 + 2^59) of products and L 2^58 of reduction terms: < 2^64 for L <= 14.
template <class X>
__device__ __forceinline__ xe_t<X> x_mul2(const xe_t<X>& a, const xe_t<X>& b, const xe_t<X>& c2, const xe_t<X>& d) {
    constexpr int L = X::L;
    uint64_t c[2 * L];
#pragma unroll
    for (int k = 0; k < 2 * L; ++k) c[k] = 0;
#pragma unroll
    for (int i = 0; i < L; ++i)
#pragma unroll
        for (int j = 0; j < L; ++j) c[i + j] += (uint64_t)a.l[i] * b.l[j];
#pragma unroll
    for (int i = 0; i < L; ++i)
#pragma unroll
        for (int j = 0; j < L; ++j) c[i + j] += (uint64_t)c2.l[i] * d.l[j];
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

// a^2/R': the cross products are taken once against the doubled operand.
template <class X>
__device__ __forceinline__ xe_t<X> x_sqr(const xe_t<X>& a) {
    constexpr int L = X::L;
    uint64_t c[2 * L];
#pragma unroll
    for (int k = 0; k < 2 * L; ++k) c[k] = 0;
    uint32_t a2[L];
#pragma unroll
    for (int i = 0; i < L; ++i) a2[i] = a.l[i] << 1;
#pragma unroll
    for (int i = 0; i < L; ++i) {
        c[2 * i] += (uint64_t)a.l[i] * a.l[i];
#pragma unroll
        for (int j = i + 1; j < L; ++j) c[i + j] += (uint64_t)a2[i] * a.l[j];
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

// carry-normalise limbs that may be out of [0, 2^29) (signed), value must be >= 0
template <class X>
__device__ __forceinline__ xe_t<X> x_norm(const xe_t<X>& a) {
    xe_t<X> r;
    int32_t carry = 0;
#pragma unroll
    for (int k = 0; k < X::L - 1; ++k) {
        const int32_t t = (int32_t)a.l[k] + carry;
        r.l[k] = (uint32_t)t & XMASK;
        carry = t >> 29;
    }
    r.l[X::L - 1] = (uint32_t)((int32_t)a.l[X::L - 1] + carry);
    return r;
}

template <class X>
__device__ __forceinline__ xe_t<X> x_add(const xe_t<X>& a, const xe_t<X>& b) {
    xe_t<X> t;
#pragma unroll
    for (int k = 0; k < X::L; ++k) t.l[k] = a.l[k] + b.l[k];
    return x_norm<X>(t);
}

// limb k of M p, M in {2, 4, 6, 8, 32}
template <class X, int M>
__device__ __forceinline__ constexpr uint32_t x_mp(int k) {
    return M == 2 ? X::p2(k) : (M == 4 ? X::p4(k) : (M == 6 ? X::p6(k) : (M == 8 ? X::p8(k) : X::p32(k))));
}

// a - b + M p: the caller picks M >= the bound of b in units of p.
template <class X, int M>
__device__ __forceinline__ xe_t<X> x_sub(const xe_t<X>& a, const xe_t<X>& b) {
    xe_t<X> t;
#pragma unroll
    for (int k = 0; k < X::L; ++k) t.l[k] = a.l[k] - b.l[k] + x_mp<X, M>(k);
    return x_norm<X>(t);
}
// a - b - 2c + M p under one normalisation: M >= the bound of b + 2c in units of p.
template <class X, int M>
__device__ __forceinline__ xe_t<X> x_sub_sub2(const xe_t<X>& a, const xe_t<X>& b, const xe_t<X>& c) {
    xe_t<X> t;
#pragma unroll
    for (int k = 0; k < X::L; ++k) t.l[k] = a.l[k] - b.l[k] - (c.l[k] << 1) + x_mp<X, M>(k);
    return x_norm<X>(t);
}

// ---- carry-free forms: limbs may run past 29 bits (an x_mul accepts a first operand with limbs
//      below 2^31 when the second is normalised); x_norm_u brings them back ----
template <class X>
__device__ __forceinline__ xe_t<X> x_norm_u(const xe_t<X>& a) {  // limbs are unsigned 32-bit quantities
    xe_t<X> r;
    uint32_t carry = 0;
#pragma unroll
    for (int k = 0; k < X::L - 1; ++k) {
        const uint32_t t = a.l[k] + carry;   // a.l[k] <= 2^32 - 8
        r.l[k] = t & XMASK;
        carry = t >> 29;
    }
    r.l[X::L - 1] = a.l[X::L - 1] + carry;
    return r;
}
template <class X>
__device__ __forceinline__ xe_t<X> x_add_lazy(const xe_t<X>& a, const xe_t<X>& b) {
    xe_t<X> t;
#pragma unroll
    for (int k = 0; k < X::L; ++k) t.l[k] = a.l[k] + b.l[k];
    return t;
}
// a - b + M p without carries, M in {3, 5}: b normalised and below (M - 1) p
template <class X, int M>
__device__ __forceinline__ xe_t<X> x_sub_lazy(const xe_t<X>& a, const xe_t<X>& b) {
    xe_t<X> t;
#pragma unroll
    for (int k = 0; k < X::L; ++k) t.l[k] = a.l[k] + ((M == 3 ? X::bp3(k) : X::bp5(k)) - b.l[k]);
    return t;
}

// M p - b without carries (limbs below 2^30), M in {3, 5}: b normalised and below (M - 1) p
template <class X, int M>
__device__ __forceinline__ xe_t<X> x_neg_lazy(const xe_t<X>& b) {
    xe_t<X> t;
#pragma unroll
    for (int k = 0; k < X::L; ++k) t.l[k] = (M == 3 ? X::bp3(k) : X::bp5(k)) - b.l[k];
    return t;
}

// a - K if a >= K else a, for K given limb-wise (normalised); a normalised
template <class X, int WHICH>
__device__ __forceinline__ xe_t<X> x_csub(const xe_t<X>& a) {
    xe_t<X> t;
    int32_t borrow = 0;
#pragma unroll
    for (int k = 0; k < X::L; ++k) {
        const uint32_t m = WHICH == 1 ? X::p(k) : (WHICH == 2 ? X::p2(k) : (WHICH == 4 ? X::p4(k) : X::p8(k)));
        const int32_t d = (int32_t)a.l[k] - (int32_t)m + borrow;
        if (k < X::L - 1) {
            t.l[k] = (uint32_t)d & XMASK;
            borrow = d >> 29;
        } else {
            t.l[k] = (uint32_t)d;
            borrow = d >> 31;
        }
    }
    xe_t<X> r;
#pragma unroll
    for (int k = 0; k < X::L; ++k) r.l[k] = borrow ? a.l[k] : t.l[k];
    return r;
}

// canonical representative in [0, p) of a normalised value below 16p
template <class X>
__device__ __forceinline__ xe_t<X> x_canon16(const xe_t<X>& a) {
    return x_csub<X, 1>(x_csub<X, 2>(x_csub<X, 4>(x_csub<X, 8>(a))));
}
// ... of a value below 2p (a product)
template <class X>
__device__ __forceinline__ xe_t<X> x_canon2(const xe_t<X>& a) {
    return x_csub<X, 1>(a);
}
// ... of a normalised Fr value below 64p, without a Montgomery product: the top limb (the value's bits from
// 232 up, below 2^29) estimates the quotient by p, one pass subtracts that multiple of p and one
// conditional subtraction finishes.  With p / 2^232 = 7597479.16 and M = floor(2^40 / 7597480) the estimate
// q = (top * M) >> 40 satisfies x / p - 1 < q <= x / p for every x below 64p, so x - q p lies in [0, 2p).
__device__ __forceinline__ xe_t<FrX> frx_canon64(const xe_t<FrX>& a) {
    constexpr uint64_t M = ((uint64_t)1 << 40) / 7597480u;
    const uint32_t q = (uint32_t)(((uint64_t)a.l[8] * M) >> 40);
    xe_t<FrX> r;
    int64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        acc += (int64_t)a.l[k] - (int64_t)((uint64_t)q * FrX::p(k));
        r.l[k] = (uint32_t)acc & XMASK;
        acc >>= 29;
    }
    acc += (int64_t)a.l[8] - (int64_t)((uint64_t)q * FrX::p(8));
    r.l[8] = (uint32_t)acc;
    return x_csub<FrX, 1>(r);
}

template <class X>
__device__ __forceinline__ bool x_is_zero16(const xe_t<X>& a) {
    // a normalised value below 16p is 0 mod p only if it is k p with k < 16, and then its low limb
    // times p^-1 mod 2^29 is k: all but 2^-25 of the nonzero values fail this one-multiply test
    const uint32_t k = (0u - a.l[0] * X::INV) & XMASK;
    if (k >= 16u) return false;
    return x_limbs_zero<X>(x_canon16<X>(a));
}

// ---- conversion to / from the saturated Montgomery form of field.cuh ----
template <class X>
__device__ __forceinline__ xe_t<X> x_repack_in(const Limbs<X::N32>& s) {
    xe_t<X> r;
#pragma unroll
    for (int k = 0; k < X::L; ++k) {
        const int bit = k * 29, limb = bit / 32, off = bit % 32;
        uint64_t v = limb < X::N32 ? s.l[limb] : 0u;
        if (limb + 1 < X::N32) v |= (uint64_t)s.l[limb + 1] << 32;
        r.l[k] = (uint32_t)(v >> off) & XMASK;
    }
    return r;
}
template <class X>
__device__ __forceinline__ Limbs<X::N32> x_repack_out(const xe_t<X>& x) {  // x canonical
    Limbs<X::N32> s;
#pragma unroll
    for (int k = 0; k < X::N32; ++k) s.l[k] = 0;
#pragma unroll
    for (int k = 0; k < X::L; ++k) {
        const int bit = k * 29, limb = bit / 32, off = bit % 32;
        const uint64_t v = (uint64_t)x.l[k] << off;
        if (limb < X::N32) s.l[limb] |= (uint32_t)v;
        if (limb + 1 < X::N32) s.l[limb + 1] |= (uint32_t)(v >> 32);
    }
    return s;
}
template <class X>
__device__ __forceinline__ xe_t<X> x_from_sat(const Limbs<X::N32>& s) {
    xe_t<X> k;
#pragma unroll
    for (int i = 0; i < X::L; ++i) k.l[i] = X::kin(i);
    return x_mul<X>(x_repack_in<X>(s), k);
}
template <class X>
__device__ __forceinline__ Limbs<X::N32> x_to_sat(const xe_t<X>& x) {  // x below 2^k p with x_mul's bound
    xe_t<X> k;
#pragma unroll
    for (int i = 0; i < X::L; ++i) k.l[i] = X::kout(i);
    return x_repack_out<X>(x_canon2<X>(x_mul<X>(x, k)));
}

// ---- Fr in this form: register type, a 48-byte table slot (9 limbs + 3 pad words), and the host-side
//      conversion of a saturated Montgomery value (radix 2^256) into a true value of this form
//      (radix 2^261: the value times 2^5, regrouped into 29-bit limbs) ----
typedef xe_t<FrX> frx;
struct alignas(16) frx_slot {
    uint32_t w[12];
};
__device__ __forceinline__ frx ld_tw(const frx_slot* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    const uint4 a = q[0], b = q[1], c = q[2];
    frx r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w; r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    r.l[8] = c.x;
    return r;
}
// `mont_times_32`: the four 64-bit words of (Montgomery value * 32 mod p)
inline void frx_words_from_u64(const uint64_t mont_times_32[4], uint32_t out[9]) {
    for (int k = 0; k < 9; ++k) {
        const int bit = k * 29, limb = bit / 64, off = bit % 64;
        uint64_t w = mont_times_32[limb] >> off;
        if (off > 64 - 29 && limb + 1 < 4) w |= mont_times_32[limb + 1] << (64 - off);
        out[k] = (uint32_t)w & ((1u << 29) - 1u);
    }
}

// 64-byte storage slot for an Fq x-form element (14 limbs + 2 pad words)
struct alignas(16) fqx_slot {
    uint32_t w[16];
};
__device__ __forceinline__ xe_t<FqX> ldx(const fqx_slot* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    const uint4 a = q[0], b = q[1], c = q[2], d = q[3];
    xe_t<FqX> r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w; r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    r.l[8] = c.x; r.l[9] = c.y; r.l[10] = c.z; r.l[11] = c.w; r.l[12] = d.x; r.l[13] = d.y;
    return r;
}
__device__ __forceinline__ void stx(fqx_slot* p, const xe_t<FqX>& a) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(a.l[0], a.l[1], a.l[2], a.l[3]);
    q[1] = make_uint4(a.l[4], a.l[5], a.l[6], a.l[7]);
    q[2] = make_uint4(a.l[8], a.l[9], a.l[10], a.l[11]);
    q[3] = make_uint4(a.l[12], a.l[13], 0u, 0u);
}

}  // namespace plonk
