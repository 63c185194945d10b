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

// How a point operation multiplies: every product expanded in place (~3.5 KB of code each).
struct MulInline {
    static __device__ __forceinline__ fqx mul(const fqx& a, const fqx& b) { return x_mul<FqX>(a, b); }
    static __device__ __forceinline__ fqx sqr(const fqx& a) { return x_sqr<FqX>(a); }
    // a b - c d for a d below 2p
    static __device__ __forceinline__ fqx mul_sub(const fqx& a, const fqx& b, const fqx& c, const fqx& d) {
        return x_mul2<FqX>(a, b, c, x_neg_lazy<FqX, 3>(d));
    }
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

// acc += q, or acc -= q when neg (madd-2008-s: 8M + 2S, the two products of Y3 under one reduction)
__device__ __forceinline__ void ptx_madd(PtX& acc, const AffX& q, bool neg = false) {
    if (affx_is_inf(q)) return;
    if (ptx_is_inf(acc)) {
        acc.x = q.x; acc.y = neg ? x_sub<FqX, 2>(x_zero<FqX>(), q.y) : q.y;
        acc.zz = x_one<FqX>(); acc.zzz = x_one<FqX>();
        return;
    }
    fqx ys;   // +-y; the negative one carry-free (limbs below 2^30), which a first operand of x_mul may be
#pragma unroll
    for (int k = 0; k < FqX::L; ++k) ys.l[k] = neg ? FqX::bp3(k) - q.y.l[k] : q.y.l[k];
    const fqx U2 = qmul(q.x, acc.zz);
    const fqx S2 = qmul(ys, acc.zzz);
    const fqx P = x_sub<FqX, 8>(U2, acc.x);   // < 10p
    const fqx R = x_sub<FqX, 4>(S2, acc.y);   // < 6p
    if (x_is_zero16<FqX>(P)) {
        if (x_is_zero16<FqX>(R)) {
            AffX s;
            s.x = q.x; s.y = neg ? x_sub<FqX, 2>(x_zero<FqX>(), q.y) : q.y;
            acc = ptx_double_affine(s);
        } else {
            acc = ptx_identity();
        }
        return;
    }
    const fqx PP = qsqr(P);
    const fqx PPP = qmul(P, PP);
    const fqx Q = qmul(acc.x, PP);
    const fqx X3 = x_sub_sub2<FqX, 6>(qsqr(R), PPP, Q);                                // R^2 - PPP - 2Q, < 8p
    const fqx Y3 = MulInline::mul_sub(R, x_sub<FqX, 8>(Q, X3), acc.y, PPP);             // < 2p
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
    const fqx X3 = x_sub_sub2<FqX, 6>(M::sqr(R), PPP, Q);
    const fqx Y3 = M::mul_sub(R, x_sub<FqX, 8>(Q, X3), S1, PPP);
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
}  // namespace plonk
