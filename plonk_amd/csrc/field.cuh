// BLS12-381 field arithmetic for gfx950: Fr (8 x u32 limbs) and Fq (12 x u32 limbs),
// Montgomery form with R = 2^256 / 2^384 -- the same bytes as the 4 x u64 / 6 x u64
// little-endian Montgomery limbs of ark_ff::Fp256 / Fp384 that the reference's `Fr`
// (backend_new/src/field.rs:18) and G1 coordinates (curve.rs:33) hold in memory.
// Limbs live in VGPRs; every loop is fully unrolled so that no array is runtime-indexed.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace plonk {

template <int N>
struct alignas(16) Limbs {
    uint32_t l[N];
};
using fr_t = Limbs<8>;
using fq_t = Limbs<12>;

struct FrP {
    static constexpr int N = 8;
    static constexpr uint32_t INV = 0xffffffffu;  // -r^-1 mod 2^32
    __host__ __device__ static constexpr uint32_t p(int i) {
        constexpr uint32_t v[8] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u,
                                   0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
        return v[i];
    }
    // R mod r (Montgomery one)
    __host__ __device__ static constexpr uint32_t one(int i) {
        constexpr uint32_t v[8] = {0xfffffffeu, 0x00000001u, 0x00034802u, 0x5884b7fau,
                                   0xecbc4ff5u, 0x998c4fefu, 0xacc5056fu, 0x1824b159u};
        return v[i];
    }
    // R^2 mod r
    __host__ __device__ static constexpr uint32_t r2(int i) {
        constexpr uint32_t v[8] = {0xf3f29c6du, 0xc999e990u, 0x87925c23u, 0x2b6cedcbu,
                                   0x7254398fu, 0x05d31496u, 0x9f59ff11u, 0x0748d9d9u};
        return v[i];
    }
};

struct FqP {
    static constexpr int N = 12;
    static constexpr uint32_t INV = 0xfffcfffdu;  // -q^-1 mod 2^32
    __host__ __device__ static constexpr uint32_t p(int i) {
        constexpr uint32_t v[12] = {0xffffaaabu, 0xb9feffffu, 0xb153ffffu, 0x1eabfffeu,
                                    0xf6b0f624u, 0x6730d2a0u, 0xf38512bfu, 0x64774b84u,
                                    0x434bacd7u, 0x4b1ba7b6u, 0x397fe69au, 0x1a0111eau};
        return v[i];
    }
    // R mod q
    __host__ __device__ static constexpr uint32_t one(int i) {
        constexpr uint32_t v[12] = {0x0002fffdu, 0x76090000u, 0xc40c0002u, 0xebf4000bu,
                                    0x53c758bau, 0x5f489857u, 0x70525745u, 0x77ce5853u,
                                    0xa256ec6du, 0x5c071a97u, 0xfa80e493u, 0x15f65ec3u};
        return v[i];
    }
    // R^2 mod q
    __host__ __device__ static constexpr uint32_t r2(int i) {
        constexpr uint32_t v[12] = {0x1c341746u, 0xf4df1f34u, 0x09d104f1u, 0x0a76e6a6u,
                                    0x4c95b6d5u, 0x8de5476cu, 0x939d83c0u, 0x67eb88a9u,
                                    0xb519952du, 0x9a793e85u, 0x92cae3aau, 0x11988fe5u};
        return v[i];
    }
};

template <class F>
using elem_t = Limbs<F::N>;

template <class F>
__device__ __forceinline__ elem_t<F> f_zero() {
    elem_t<F> r;
#pragma unroll
    for (int i = 0; i < F::N; ++i) r.l[i] = 0;
    return r;
}

template <class F>
__device__ __forceinline__ elem_t<F> f_one() {
    elem_t<F> r;
#pragma unroll
    for (int i = 0; i < F::N; ++i) r.l[i] = F::one(i);
    return r;
}

template <class F>
__device__ __forceinline__ bool f_is_zero(const elem_t<F>& a) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < F::N; ++i) o |= a.l[i];
    return o == 0;
}

template <class F>
__device__ __forceinline__ bool f_eq(const elem_t<F>& a, const elem_t<F>& b) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < F::N; ++i) o |= a.l[i] ^ b.l[i];
    return o == 0;
}

// r = a - p if a >= p else a   (a < 2p)
template <class F>
__device__ __forceinline__ elem_t<F> f_reduce_once(const elem_t<F>& a) {
    elem_t<F> s;
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < F::N; ++i) s.l[i] = __builtin_subc(a.l[i], F::p(i), borrow, &borrow);
    elem_t<F> r;
#pragma unroll
    for (int i = 0; i < F::N; ++i) r.l[i] = borrow ? a.l[i] : s.l[i];
    return r;
}

template <class F>
__device__ __forceinline__ elem_t<F> f_add(const elem_t<F>& a, const elem_t<F>& b) {
    elem_t<F> t;
    uint32_t carry = 0;
#pragma unroll
    for (int i = 0; i < F::N; ++i) t.l[i] = __builtin_addc(a.l[i], b.l[i], carry, &carry);
    return f_reduce_once<F>(t);  // top limb of p has spare bits: a + b never overflows N limbs
}

template <class F>
__device__ __forceinline__ elem_t<F> f_sub(const elem_t<F>& a, const elem_t<F>& b) {
    elem_t<F> t;
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < F::N; ++i) t.l[i] = __builtin_subc(a.l[i], b.l[i], borrow, &borrow);
    const uint32_t mask = 0u - borrow;
    uint32_t carry = 0;
    elem_t<F> r;
#pragma unroll
    for (int i = 0; i < F::N; ++i) r.l[i] = __builtin_addc(t.l[i], F::p(i) & mask, carry, &carry);
    return r;
}

template <class F>
__device__ __forceinline__ elem_t<F> f_neg(const elem_t<F>& a) {
    return f_sub<F>(f_zero<F>(), a);
}

template <class F>
__device__ __forceinline__ elem_t<F> f_dbl(const elem_t<F>& a) {
    return f_add<F>(a, a);
}

// Montgomery product a*b*R^-1 mod p, fully reduced.  CIOS without the extra carry word:
// valid because the top limb of p is below 2^31 (Fr: 0x73eda753, Fq: 0x1a0111ea).
template <class F>
__device__ __forceinline__ elem_t<F> f_mul(const elem_t<F>& a, const elem_t<F>& b) {
    constexpr int N = F::N;
    uint32_t t[N];
#pragma unroll
    for (int i = 0; i < N; ++i) t[i] = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        uint64_t c = (uint64_t)a.l[0] * b.l[i] + t[0];
        const uint32_t t0 = (uint32_t)c;
        c >>= 32;
        const uint32_t m = t0 * F::INV;
        uint64_t c2 = (uint64_t)m * F::p(0) + t0;
        c2 >>= 32;
#pragma unroll
        for (int j = 1; j < N; ++j) {
            c += (uint64_t)a.l[j] * b.l[i] + t[j];
            c2 += (uint64_t)m * F::p(j) + (uint32_t)c;
            t[j - 1] = (uint32_t)c2;
            c2 >>= 32;
            c >>= 32;
        }
        t[N - 1] = (uint32_t)(c + c2);
    }
    elem_t<F> r;
#pragma unroll
    for (int i = 0; i < N; ++i) r.l[i] = t[i];
    return f_reduce_once<F>(r);
}

template <class F>
__device__ __forceinline__ elem_t<F> f_sqr(const elem_t<F>& a) {
    return f_mul<F>(a, a);
}

template <class F>
__device__ __forceinline__ elem_t<F> f_to_mont(const elem_t<F>& a) {
    elem_t<F> r2;
#pragma unroll
    for (int i = 0; i < F::N; ++i) r2.l[i] = F::r2(i);
    return f_mul<F>(a, r2);
}

template <class F>
__device__ __forceinline__ elem_t<F> f_from_mont(const elem_t<F>& a) {
    elem_t<F> one;
#pragma unroll
    for (int i = 0; i < F::N; ++i) one.l[i] = (i == 0) ? 1u : 0u;
    return f_mul<F>(a, one);
}

// 16-byte vector load/store of a whole element (elements are 16-byte aligned).
template <int N>
__device__ __forceinline__ Limbs<N> ld(const Limbs<N>* p) {
    Limbs<N> r;
    const uint4* q = reinterpret_cast<const uint4*>(p);
#pragma unroll
    for (int i = 0; i < N / 4; ++i) {
        uint4 v = q[i];
        r.l[4 * i] = v.x; r.l[4 * i + 1] = v.y; r.l[4 * i + 2] = v.z; r.l[4 * i + 3] = v.w;
    }
    return r;
}

template <int N>
__device__ __forceinline__ void st(Limbs<N>* p, const Limbs<N>& a) {
    uint4* q = reinterpret_cast<uint4*>(p);
#pragma unroll
    for (int i = 0; i < N / 4; ++i) q[i] = make_uint4(a.l[4 * i], a.l[4 * i + 1], a.l[4 * i + 2], a.l[4 * i + 3]);
}

__device__ __forceinline__ fr_t fr_mul(const fr_t& a, const fr_t& b) { return f_mul<FrP>(a, b); }
__device__ __forceinline__ fr_t fr_add(const fr_t& a, const fr_t& b) { return f_add<FrP>(a, b); }
__device__ __forceinline__ fr_t fr_sub(const fr_t& a, const fr_t& b) { return f_sub<FrP>(a, b); }
__device__ __forceinline__ fq_t fq_mul(const fq_t& a, const fq_t& b) { return f_mul<FqP>(a, b); }
__device__ __forceinline__ fq_t fq_add(const fq_t& a, const fq_t& b) { return f_add<FqP>(a, b); }
__device__ __forceinline__ fq_t fq_sub(const fq_t& a, const fq_t& b) { return f_sub<FqP>(a, b); }

}  // namespace plonk
