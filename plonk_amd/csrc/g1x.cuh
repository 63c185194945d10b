// G1 in XYZZ coordinates over the x-form (radix-2^29 lazy Montgomery) Fq of fieldx.cuh: the
// arithmetic of the MSM inner loops.  Stored coordinates are limb-normalised with X < 8p, Y < 4p,
// ZZ, ZZZ < 2p; an all-zero ZZ marks the identity; an affine point with x = y = 0 is the identity.
#pragma once
#include "fieldx.cuh"

namespace plonk {

typedef xe_t<FqX> fqx;

struct alignas(16) g1x_affine_t {
    fqx_slot x, y;  // 128 bytes: one cache line per point
};
struct alignas(16) g1x_xyzz_t {
    fqx_slot x, y, zz, zzz;  // 256 bytes
};
struct AffX {
    fqx x, y;
};
struct PtX {
    fqx x, y, zz, zzz;
};

__device__ __forceinline__ fqx qmul(const fqx& a, const fqx& b) { return x_mul<FqX>(a, b); }
__device__ __forceinline__ fqx qsqr(const fqx& a) { return x_sqr<FqX>(a); }

// How a point operation multiplies.  MulInline expands every product in place: ~3.5 KB of code per
// product, right for the bucket loop whose many waves share one instruction stream.  MulCall goes
// through one out-of-line routine (operands and result in registers), so a whole point addition
// stays a few KB: the reductions after the bucket phase run a handful of waves through long
// dependent chains, and there instruction fetch of the expanded form costs more than the
// arithmetic.
struct MulInline {
    static __device__ __forceinline__ fqx mul(const fqx& a, const fqx& b) { return x_mul<FqX>(a, b); }
    static __device__ __forceinline__ fqx sqr(const fqx& a) { return x_sqr<FqX>(a); }
};
struct MulCall {
    // operands as 14-lane vectors: those travel in registers, a struct argument would go through
    // scratch memory
    typedef uint32_t vec __attribute__((ext_vector_type(14)));
    static __device__ __forceinline__ vec pack(const fqx& a) {
        vec v;
#pragma unroll
        for (int i = 0; i < 14; ++i) v[i] = a.l[i];
        return v;
    }
    static __device__ __forceinline__ fqx unpack(const vec& v) {
        fqx a;
#pragma unroll
        for (int i = 0; i < 14; ++i) a.l[i] = v[i];
        return a;
    }
    static __device__ __noinline__ vec mul_fn(vec a, vec b) { return pack(x_mul<FqX>(unpack(a), unpack(b))); }
    static __device__ __noinline__ vec sqr_fn(vec a) { return pack(x_sqr<FqX>(unpack(a))); }
    static __device__ __forceinline__ fqx mul(const fqx& a, const fqx& b) { return unpack(mul_fn(pack(a), pack(b))); }
    static __device__ __forceinline__ fqx sqr(const fqx& a) { return unpack(sqr_fn(pack(a))); }
};

__device__ __forceinline__ bool ptx_is_inf(const PtX& p) { return x_limbs_zero<FqX>(p.zz); }
__device__ __forceinline__ bool affx_is_inf(const AffX& p) { return x_limbs_zero<FqX>(p.x) && x_limbs_zero<FqX>(p.y); }
__device__ __forceinline__ PtX ptx_identity() {
    PtX r;
    r.x = x_zero<FqX>(); r.y = x_zero<FqX>(); r.zz = x_zero<FqX>(); r.zzz = x_zero<FqX>();
    return r;
}

// 2 * P (dbl-2008-s-1, a = 0)
__device__ __forceinline__ PtX ptx_double(const PtX& p) {
    if (ptx_is_inf(p) || x_is_zero16<FqX>(p.y)) return ptx_identity();
    PtX r;
    const fqx U = x_add<FqX>(p.y, p.y);                       // < 8p
    const fqx V = qsqr(U);
    const fqx W = qmul(U, V);
    const fqx S = qmul(p.x, V);
    const fqx xx = qsqr(p.x);
    const fqx M = x_add<FqX>(x_add<FqX>(xx, xx), xx);         // < 6p
    r.x = x_sub<FqX, 4>(qsqr(M), x_add<FqX>(S, S));           // < 6p
    r.y = x_sub<FqX, 2>(qmul(M, x_sub<FqX, 8>(S, r.x)), qmul(W, p.y));  // < 4p
    r.zz = qmul(V, p.zz);
    r.zzz = qmul(W, p.zzz);
    return r;
}

__device__ __forceinline__ PtX ptx_double_affine(const AffX& q) {
    PtX p;
    p.x = q.x; p.y = q.y; p.zz = x_one<FqX>(); p.zzz = x_one<FqX>();
    return ptx_double(p);
}

// acc += q (madd-2008-s: 8M + 2S)
__device__ __forceinline__ void ptx_madd(PtX& acc, const AffX& q) {
    if (affx_is_inf(q)) return;
    if (ptx_is_inf(acc)) {
        acc.x = q.x; acc.y = q.y; acc.zz = x_one<FqX>(); acc.zzz = x_one<FqX>();
        return;
    }
    const fqx U2 = qmul(q.x, acc.zz);
    const fqx S2 = qmul(q.y, acc.zzz);
    const fqx P = x_sub<FqX, 8>(U2, acc.x);   // < 10p
    const fqx R = x_sub<FqX, 4>(S2, acc.y);   // < 6p
    if (x_is_zero16<FqX>(P)) {
        if (x_is_zero16<FqX>(R)) acc = ptx_double_affine(q);
        else acc = ptx_identity();
        return;
    }
    const fqx PP = qsqr(P);
    const fqx PPP = qmul(P, PP);
    const fqx Q = qmul(acc.x, PP);
    const fqx X3 = x_sub<FqX, 4>(x_sub<FqX, 2>(qsqr(R), PPP), x_add<FqX>(Q, Q));   // < 8p
    const fqx Y3 = x_sub<FqX, 2>(qmul(R, x_sub<FqX, 8>(Q, X3)), qmul(acc.y, PPP));  // < 4p
    acc.x = X3;
    acc.y = Y3;
    acc.zz = qmul(acc.zz, PP);
    acc.zzz = qmul(acc.zzz, PPP);
}

// acc += q (add-2008-s: 12M + 2S)
template <class M = MulInline>
__device__ __forceinline__ void ptx_add(PtX& acc, const PtX& q) {
    if (ptx_is_inf(q)) return;
    if (ptx_is_inf(acc)) {
        acc = q;
        return;
    }
    const fqx U1 = M::mul(acc.x, q.zz);
    const fqx U2 = M::mul(q.x, acc.zz);
    const fqx S1 = M::mul(acc.y, q.zzz);
    const fqx S2 = M::mul(q.y, acc.zzz);
    const fqx P = x_sub<FqX, 2>(U2, U1);  // < 4p
    const fqx R = x_sub<FqX, 2>(S2, S1);
    if (x_is_zero16<FqX>(P)) {
        if (x_is_zero16<FqX>(R)) acc = ptx_double(q);
        else acc = ptx_identity();
        return;
    }
    const fqx PP = M::sqr(P);
    const fqx PPP = M::mul(P, PP);
    const fqx Q = M::mul(U1, PP);
    const fqx X3 = x_sub<FqX, 4>(x_sub<FqX, 2>(M::sqr(R), PPP), x_add<FqX>(Q, Q));
    const fqx Y3 = x_sub<FqX, 2>(M::mul(R, x_sub<FqX, 8>(Q, X3)), M::mul(S1, PPP));
    acc.x = X3;
    acc.y = Y3;
    acc.zz = M::mul(M::mul(acc.zz, q.zz), PP);
    acc.zzz = M::mul(M::mul(acc.zzz, q.zzz), PPP);
}

__device__ __forceinline__ AffX ld_affx(const g1x_affine_t* p) {
    AffX r;
    r.x = ldx(&p->x);
    r.y = ldx(&p->y);
    return r;
}
__device__ __forceinline__ PtX ld_ptx(const g1x_xyzz_t* p) {
    PtX r;
    r.x = ldx(&p->x); r.y = ldx(&p->y); r.zz = ldx(&p->zz); r.zzz = ldx(&p->zzz);
    return r;
}
__device__ __forceinline__ void st_ptx(g1x_xyzz_t* p, const PtX& a) {
    stx(&p->x, a.x); stx(&p->y, a.y); stx(&p->zz, a.zz); stx(&p->zzz, a.zzz);
}
// -y for a y < 2p (identity marker (0,0) must stay (0,0))
__device__ __forceinline__ fqx fqx_neg_y(const fqx& y) {
    if (x_limbs_zero<FqX>(y)) return y;
    return x_sub<FqX, 2>(x_zero<FqX>(), y);
}

}  // namespace plonk
