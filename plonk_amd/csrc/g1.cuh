// BLS12-381 G1 point arithmetic for gfx950 in XYZZ coordinates (x = X/ZZ, y = Y/ZZZ,
// ZZ^3 = ZZZ^2; ZZ == 0 is the identity), y^2 = x^3 + 4.  These are the group operations behind
// G1Projective::msm / scalar_mul of backend_new/src/curve.rs:163-173 (ark-ec there works in
// Jacobian coordinates; the group element is representation-independent).
// Affine points are (x, y) Montgomery Fq; the pair (0, 0) marks the identity.
#pragma once
#include "field.cuh"

namespace plonk {

struct alignas(16) g1_affine_t {
    fq_t x, y;
};
struct alignas(16) g1_xyzz_t {
    fq_t x, y, zz, zzz;
};

__device__ __forceinline__ bool affine_is_inf(const g1_affine_t& p) { return f_is_zero<FqP>(p.x) && f_is_zero<FqP>(p.y); }
__device__ __forceinline__ bool xyzz_is_inf(const g1_xyzz_t& p) { return f_is_zero<FqP>(p.zz); }

__device__ __forceinline__ g1_xyzz_t xyzz_identity() {
    g1_xyzz_t r;
    r.x = f_zero<FqP>(); r.y = f_zero<FqP>(); r.zz = f_zero<FqP>(); r.zzz = f_zero<FqP>();
    return r;
}

__device__ __forceinline__ g1_xyzz_t xyzz_from_affine(const g1_affine_t& p) {
    g1_xyzz_t r;
    if (affine_is_inf(p)) return xyzz_identity();
    r.x = p.x; r.y = p.y; r.zz = f_one<FqP>(); r.zzz = f_one<FqP>();
    return r;
}

// 2 * (x, y) for an affine point (mdbl-2008-s-1).
__device__ __noinline__ g1_xyzz_t xyzz_double_affine(const g1_affine_t& p) {
    if (affine_is_inf(p) || f_is_zero<FqP>(p.y)) return xyzz_identity();
    g1_xyzz_t r;
    fq_t U = f_dbl<FqP>(p.y);
    fq_t V = fq_mul(U, U);
    fq_t W = fq_mul(U, V);
    fq_t S = fq_mul(p.x, V);
    fq_t xx = fq_mul(p.x, p.x);
    fq_t M = fq_add(f_dbl<FqP>(xx), xx);
    r.x = fq_sub(fq_mul(M, M), f_dbl<FqP>(S));
    r.y = fq_sub(fq_mul(M, fq_sub(S, r.x)), fq_mul(W, p.y));
    r.zz = V;
    r.zzz = W;
    return r;
}

// 2 * P (dbl-2008-s-1, a = 0).
__device__ __noinline__ g1_xyzz_t xyzz_double(const g1_xyzz_t& p) {
    if (xyzz_is_inf(p) || f_is_zero<FqP>(p.y)) return xyzz_identity();
    g1_xyzz_t r;
    fq_t U = f_dbl<FqP>(p.y);
    fq_t V = fq_mul(U, U);
    fq_t W = fq_mul(U, V);
    fq_t S = fq_mul(p.x, V);
    fq_t xx = fq_mul(p.x, p.x);
    fq_t M = fq_add(f_dbl<FqP>(xx), xx);
    r.x = fq_sub(fq_mul(M, M), f_dbl<FqP>(S));
    r.y = fq_sub(fq_mul(M, fq_sub(S, r.x)), fq_mul(W, p.y));
    r.zz = fq_mul(V, p.zz);
    r.zzz = fq_mul(W, p.zzz);
    return r;
}

// acc += (x2, y2)  (madd-2008-s, 8M + 2S), all special cases handled.
__device__ __forceinline__ void xyzz_madd(g1_xyzz_t& acc, const g1_affine_t& q) {
    if (affine_is_inf(q)) return;
    if (xyzz_is_inf(acc)) {
        acc.x = q.x; acc.y = q.y; acc.zz = f_one<FqP>(); acc.zzz = f_one<FqP>();
        return;
    }
    fq_t U2 = fq_mul(q.x, acc.zz);
    fq_t S2 = fq_mul(q.y, acc.zzz);
    fq_t P = fq_sub(U2, acc.x);
    fq_t R = fq_sub(S2, acc.y);
    if (f_is_zero<FqP>(P)) {
        if (f_is_zero<FqP>(R)) acc = xyzz_double_affine(q);
        else acc = xyzz_identity();
        return;
    }
    fq_t PP = fq_mul(P, P);
    fq_t PPP = fq_mul(P, PP);
    fq_t Q = fq_mul(acc.x, PP);
    fq_t X3 = fq_sub(fq_sub(fq_mul(R, R), PPP), f_dbl<FqP>(Q));
    fq_t Y3 = fq_sub(fq_mul(R, fq_sub(Q, X3)), fq_mul(acc.y, PPP));
    acc.x = X3;
    acc.y = Y3;
    acc.zz = fq_mul(acc.zz, PP);
    acc.zzz = fq_mul(acc.zzz, PPP);
}

// acc += q  (add-2008-s, 12M + 2S), all special cases handled.
__device__ __forceinline__ void xyzz_add(g1_xyzz_t& acc, const g1_xyzz_t& q) {
    if (xyzz_is_inf(q)) return;
    if (xyzz_is_inf(acc)) {
        acc = q;
        return;
    }
    fq_t U1 = fq_mul(acc.x, q.zz);
    fq_t U2 = fq_mul(q.x, acc.zz);
    fq_t S1 = fq_mul(acc.y, q.zzz);
    fq_t S2 = fq_mul(q.y, acc.zzz);
    fq_t P = fq_sub(U2, U1);
    fq_t R = fq_sub(S2, S1);
    if (f_is_zero<FqP>(P)) {
        if (f_is_zero<FqP>(R)) acc = xyzz_double(q);
        else acc = xyzz_identity();
        return;
    }
    fq_t PP = fq_mul(P, P);
    fq_t PPP = fq_mul(P, PP);
    fq_t Q = fq_mul(U1, PP);
    fq_t X3 = fq_sub(fq_sub(fq_mul(R, R), PPP), f_dbl<FqP>(Q));
    fq_t Y3 = fq_sub(fq_mul(R, fq_sub(Q, X3)), fq_mul(S1, PPP));
    acc.x = X3;
    acc.y = Y3;
    acc.zz = fq_mul(fq_mul(acc.zz, q.zz), PP);
    acc.zzz = fq_mul(fq_mul(acc.zzz, q.zzz), PPP);
}

__device__ __forceinline__ g1_affine_t ld_affine(const g1_affine_t* p) {
    g1_affine_t r;
    r.x = ld(&p->x);
    r.y = ld(&p->y);
    return r;
}
__device__ __forceinline__ g1_xyzz_t ld_xyzz(const g1_xyzz_t* p) {
    g1_xyzz_t r;
    r.x = ld(&p->x); r.y = ld(&p->y); r.zz = ld(&p->zz); r.zzz = ld(&p->zzz);
    return r;
}
__device__ __forceinline__ void st_xyzz(g1_xyzz_t* p, const g1_xyzz_t& a) {
    st(&p->x, a.x); st(&p->y, a.y); st(&p->zz, a.zz); st(&p->zzz, a.zzz);
}

}  // namespace plonk
