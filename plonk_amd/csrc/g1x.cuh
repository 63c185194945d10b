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

// acc = +-a +- b for two affine points (mmadd-2008-s: 4M + 2S, the two products of Y3 under one reduction -- about
// half a mixed addition).  False, acc untouched, when one of them is the identity or they share x.
__device__ __forceinline__ bool ptx_from_two(PtX& acc, const AffX& a, bool neg_a, const AffX& b, bool neg_b) {
    if (affx_is_inf(a) || affx_is_inf(b)) return false;
    const fqx P = x_sub<FqX, 2>(b.x, a.x);                               // < 4p
    if (x_is_zero16<FqX>(P)) return false;
    const fqx y1 = neg_a ? x_sub<FqX, 2>(x_zero<FqX>(), a.y) : a.y;      // < 4p
    const fqx y2 = neg_b ? x_sub<FqX, 2>(x_zero<FqX>(), b.y) : b.y;
    const fqx R = x_sub<FqX, 4>(y2, y1);                                 // < 8p
    const fqx PP = qsqr(P);
    const fqx PPP = qmul(P, PP);
    const fqx Q = qmul(a.x, PP);
    const fqx X3 = x_sub_sub2<FqX, 6>(qsqr(R), PPP, Q);                  // R^2 - PPP - 2Q, < 8p
    acc.y = MulInline::mul_sub(R, x_sub<FqX, 8>(Q, X3), y1, PPP);        // < 2p
    acc.x = X3;
    acc.zz = PP;
    acc.zzz = PPP;
    return true;
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

// ---- one addition by a PAIR of neighbouring lanes (even, odd) ----
// The reductions after the bucket phase are chains of dependent additions run by a handful of waves,
// so what counts there is the length of one addition's instruction stream, not the work.  A pair
// holds a point as
//     even lane: p = X, s = ZZ,  t = ZZZ          odd lane: p = Y, s = ZZZ, t = ZZ
// and both lanes run ONE instruction stream whose products differ by lane:
//     m1 = (U1 | S1), m2 = (U2 | S2), m3 = (ZZZ1 ZZZ2 | ZZ1 ZZ2), D = (P | R), D^2 = (PP | RR),
//     (PPP | ZZ3), (Q | -), then (ZZZ3 | Y3) as one a b - c d (c = 0 on the even lane):
// 7 products deep instead of 14, with values handed across by quad-permute moves.
struct PairPt {
    fqx p, s, t;
};

template <int CTRL>
__device__ __forceinline__ fqx pair_bcast(const fqx& a) {   // CTRL 0xA0: from the even lane; 0xF5: from the odd lane
    fqx r;
#pragma unroll
    for (int k = 0; k < FqX::L; ++k) r.l[k] = (uint32_t)__builtin_amdgcn_mov_dpp((int)a.l[k], CTRL, 0xf, 0xf, true);
    return r;
}
__device__ __forceinline__ fqx pair_sel(bool odd, const fqx& if_odd, const fqx& if_even) {
    fqx r;
#pragma unroll
    for (int k = 0; k < FqX::L; ++k) r.l[k] = odd ? if_odd.l[k] : if_even.l[k];
    return r;
}
// ZZ of a pair-held point, on either lane (all-zero limbs mark the identity)
__device__ __forceinline__ bool pair_is_inf(const PairPt& a, bool odd) { return x_limbs_zero<FqX>(odd ? a.t : a.s); }

__device__ __forceinline__ PairPt pair_from_ptx(const PtX& r, bool odd) {
    PairPt a;
    a.p = pair_sel(odd, r.y, r.x);
    a.s = pair_sel(odd, r.zzz, r.zz);
    a.t = pair_sel(odd, r.zz, r.zzz);
    return a;
}
__device__ __forceinline__ PtX pair_to_ptx(const PairPt& a, bool odd) {   // the whole point, on both lanes
    PtX r;
    r.x = pair_bcast<0xA0>(a.p);
    r.y = pair_bcast<0xF5>(a.p);
    r.zz = pair_bcast<0xA0>(a.s);
    r.zzz = pair_bcast<0xA0>(a.t);
    return r;
}

// a += b; both lanes of a pair must call it together (same control flow on both)
__device__ __forceinline__ void pair_add(PairPt& a, const PairPt& b, bool odd) {
    if (pair_is_inf(b, odd)) return;
    if (pair_is_inf(a, odd)) {
        a = b;
        return;
    }
    const fqx m1 = qmul(a.p, b.s);                     // U1 = X1 ZZ2     | S1 = Y1 ZZZ2
    const fqx m2 = qmul(b.p, a.s);                     // U2 = X2 ZZ1     | S2 = Y2 ZZZ1
    const fqx m3 = qmul(a.t, b.t);                     // ZZZ1 ZZZ2       | ZZ1 ZZ2
    const fqx D = x_sub<FqX, 2>(m2, m1);               // P               | R        (< 4p)
    const int dz = x_is_zero16<FqX>(D) ? 1 : 0;
    const int pz = __builtin_amdgcn_mov_dpp(dz, 0xA0, 0xf, 0xf, true);
    if (pz) {                                          // P = 0: the same or opposite points -- whole-point routine
        PtX fa = pair_to_ptx(a, odd);
        const PtX fb = pair_to_ptx(b, odd);
        ptx_add<MulInline>(fa, fb);
        a = pair_from_ptx(fa, odd);
        return;
    }
    const fqx sq = qsqr(D);                            // PP              | RR
    const fqx PP = pair_bcast<0xA0>(sq);
    const fqx n1 = qmul(pair_sel(odd, m3, D), PP);     // PPP = P PP      | ZZ3 = ZZ1 ZZ2 PP
    const fqx n2 = qmul(m1, PP);                       // Q = U1 PP       | (unused)
    const fqx PPP = pair_bcast<0xA0>(n1);
    const fqx Q = pair_bcast<0xA0>(n2);
    const fqx X3 = x_sub_sub2<FqX, 6>(sq, PPP, Q);     // (unused)        | RR - PPP - 2Q  (< 8p)
    const fqx Dd = x_sub<FqX, 8>(Q, X3);               // (unused)        | Q - X3
    // ZZZ3 = ZZZ1 ZZZ2 PPP (c = 0)                    | Y3 = R (Q - X3) - S1 PPP
    const fqx v = MulInline::mul_sub(pair_sel(odd, D, m3), pair_sel(odd, Dd, n1), pair_sel(odd, m1, x_zero<FqX>()), PPP);
    const fqx X3e = pair_bcast<0xF5>(X3), ZZ3e = pair_bcast<0xF5>(n1), ZZZ3o = pair_bcast<0xA0>(v);
    a.p = pair_sel(odd, v, X3e);
    a.s = pair_sel(odd, ZZZ3o, ZZ3e);
    a.t = pair_sel(odd, n1, v);
}

// a += +-q for an affine q whose x the even lane holds in `c` and whose y the odd lane holds in `c`
// (madd-2008-s by a pair: 5.3 products deep instead of 9.8); both lanes of the pair call it together
__device__ __forceinline__ void pair_madd(PairPt& a, const fqx& c, bool neg, bool odd) {
    const int cz = x_limbs_zero<FqX>(c) ? 1 : 0;
    const int xz = __builtin_amdgcn_mov_dpp(cz, 0xA0, 0xf, 0xf, true), yz = __builtin_amdgcn_mov_dpp(cz, 0xF5, 0xf, 0xf, true);
    if (xz && yz) return;                              // (0, 0) marks the identity
    if (pair_is_inf(a, odd)) {
        a.p = (odd && neg) ? x_sub<FqX, 2>(x_zero<FqX>(), c) : c;
        a.s = x_one<FqX>();
        a.t = x_one<FqX>();
        return;
    }
    fqx cs;   // x | +-y, the negative one carry-free (limbs below 2^30), which a first operand of x_mul may be
#pragma unroll
    for (int k = 0; k < FqX::L; ++k) cs.l[k] = (odd && neg) ? FqX::bp3(k) - c.l[k] : c.l[k];
    const fqx m2 = qmul(cs, a.s);                      // U2 = X2 ZZ1     | S2 = +-Y2 ZZZ1
    const fqx D = x_sub<FqX, 8>(m2, a.p);              // P = U2 - X1     | R = S2 - Y1   (< 10p)
    const int dz = x_is_zero16<FqX>(D) ? 1 : 0;
    const int pz = __builtin_amdgcn_mov_dpp(dz, 0xA0, 0xf, 0xf, true);
    if (pz) {                                          // P = 0: the same or opposite points -- whole-point routine
        PtX fa = pair_to_ptx(a, odd);
        AffX q;
        q.x = pair_bcast<0xA0>(c);
        q.y = pair_bcast<0xF5>(c);
        ptx_madd(fa, q, neg);
        a = pair_from_ptx(fa, odd);
        return;
    }
    const fqx sq = qsqr(D);                            // PP              | RR
    const fqx PP = pair_bcast<0xA0>(sq);
    const fqx n1 = qmul(pair_sel(odd, a.t, D), PP);    // PPP = P PP      | ZZ3 = ZZ1 PP
    const fqx n2 = qmul(a.p, PP);                      // Q = X1 PP       | (unused)
    const fqx PPP = pair_bcast<0xA0>(n1);
    const fqx Q = pair_bcast<0xA0>(n2);
    const fqx X3 = x_sub_sub2<FqX, 6>(sq, PPP, Q);     // (unused)        | RR - PPP - 2Q  (< 8p)
    const fqx Dd = x_sub<FqX, 8>(Q, X3);               // (unused)        | Q - X3
    // ZZZ3 = ZZZ1 PPP (c = 0)                         | Y3 = R (Q - X3) - Y1 PPP
    const fqx v = MulInline::mul_sub(pair_sel(odd, D, a.t), pair_sel(odd, Dd, n1), pair_sel(odd, a.p, x_zero<FqX>()), PPP);
    const fqx X3e = pair_bcast<0xF5>(X3), ZZ3e = pair_bcast<0xF5>(n1), ZZZ3o = pair_bcast<0xA0>(v);
    a.p = pair_sel(odd, v, X3e);
    a.s = pair_sel(odd, ZZZ3o, ZZ3e);
    a.t = pair_sel(odd, n1, v);
}

// a point in memory as four coordinate slots
__device__ __forceinline__ PairPt ld_pair(const g1x_xyzz_t* q, bool odd) {
    PairPt a;
    a.p = ldx(odd ? &q->y : &q->x);
    a.s = ldx(odd ? &q->zzz : &q->zz);
    a.t = ldx(odd ? &q->zz : &q->zzz);
    return a;
}
__device__ __forceinline__ void st_pair(g1x_xyzz_t* q, const PairPt& a, bool odd) {
    if (odd) {
        stx(&q->y, a.p);
    } else {
        stx(&q->x, a.p);
        stx(&q->zz, a.s);
        stx(&q->zzz, a.t);
    }
}
__device__ __forceinline__ PairPt pair_identity() {
    PairPt a;
    a.p = x_zero<FqX>(); a.s = x_zero<FqX>(); a.t = x_zero<FqX>();
    return a;
}

// ---- one addition by a QUAD of neighbouring lanes: lane j of the quad holds coordinate j of X, Y, ZZ, ZZZ ----
// add-2008-s laid out four products deep (a pair needs seven, a lane fourteen):
//     level 1:  U1 = X1 ZZ2   | S1 = Y1 ZZZ2    | U2 = X2 ZZ1 | S2 = Y2 ZZZ1
//               P = U2 - U1 on lane 2, R = S2 - S1 on lane 3
//     level 2:  ZZ1 ZZ2       | ZZZ1 ZZZ2       | PP = P^2    | RR = R^2
//     level 3:  Q = U1 PP     | -               | PPP = P PP  | ZZ3 = ZZ1 ZZ2 PP
//               X3 = RR - PPP - 2Q on lane 3
//     level 4:  ZZZ3 = ZZZ1 ZZZ2 PPP | S1 PPP   | -           | R (Q - X3)
//               Y3 = R (Q - X3) - S1 PPP on lane 1
// Values cross lanes by quad-permute moves.  For the reduction trees, where the length of one addition's
// instruction stream is what counts.
constexpr int quad_ctrl(int s0, int s1, int s2, int s3) { return s0 | (s1 << 2) | (s2 << 4) | (s3 << 6); }
template <int CTRL>
__device__ __forceinline__ fqx quad_perm(const fqx& a) { return pair_bcast<CTRL>(a); }
template <int J>
__device__ __forceinline__ fqx quad_from(const fqx& a) { return pair_bcast<quad_ctrl(J, J, J, J)>(a); }
__device__ __forceinline__ bool quad_is_inf(const fqx& v) {   // ZZ = 0 marks the identity; lane 2 holds ZZ
    const int z = x_limbs_zero<FqX>(v) ? 1 : 0;
    return __builtin_amdgcn_mov_dpp(z, quad_ctrl(2, 2, 2, 2), 0xf, 0xf, true) != 0;
}
__device__ __forceinline__ fqx quad_identity() { return x_zero<FqX>(); }

// a += b; the four lanes of a quad call it together; j = lane & 3
__device__ __forceinline__ void quad_add(fqx& a, const fqx& b, uint32_t j) {
    if (quad_is_inf(b)) return;
    if (quad_is_inf(a)) {
        a = b;
        return;
    }
    const bool upper = j >= 2;
    const fqx m1 = qmul(a, quad_perm<quad_ctrl(2, 3, 0, 1)>(b));
    const fqx D = x_sub<FqX, 2>(m1, quad_perm<quad_ctrl(0, 1, 0, 1)>(m1));   // lane 2: P, lane 3: R (< 4p)
    const int dz = x_is_zero16<FqX>(D) ? 1 : 0;
    if (__builtin_amdgcn_mov_dpp(dz, quad_ctrl(2, 2, 2, 2), 0xf, 0xf, true)) {   // P = 0: the same or opposite points
        PtX fa, fb;
        fa.x = quad_from<0>(a); fa.y = quad_from<1>(a); fa.zz = quad_from<2>(a); fa.zzz = quad_from<3>(a);
        fb.x = quad_from<0>(b); fb.y = quad_from<1>(b); fb.zz = quad_from<2>(b); fb.zzz = quad_from<3>(b);
        ptx_add<MulInline>(fa, fb);
        a = pair_sel(upper, pair_sel(j == 3, fa.zzz, fa.zz), pair_sel(j == 1, fa.y, fa.x));
        return;
    }
    const fqx m2 = qmul(pair_sel(upper, D, quad_perm<quad_ctrl(2, 3, 2, 3)>(a)),
                        pair_sel(upper, D, quad_perm<quad_ctrl(2, 3, 2, 3)>(b)));
    const fqx PP = quad_from<2>(m2);
    const fqx m3 = qmul(pair_sel(j == 2, D, pair_sel(j == 3, quad_from<0>(m2), m1)), PP);
    const fqx PPP = quad_from<2>(m3), Q = quad_from<0>(m3);
    const fqx X3 = x_sub_sub2<FqX, 6>(m2, PPP, Q);        // lane 3: RR - PPP - 2Q (< 8p)
    const fqx Dd = x_sub<FqX, 8>(Q, X3);                  // lane 3: Q - X3
    const fqx m4 = qmul(pair_sel(j == 0, quad_from<1>(m2), pair_sel(j == 3, D, m1)), pair_sel(j == 3, Dd, PPP));
    const fqx Y3 = x_sub<FqX, 2>(quad_from<3>(m4), m4);   // lane 1: R (Q - X3) - S1 PPP (< 4p)
    a = pair_sel(upper, pair_sel(j == 3, quad_from<0>(m4), quad_from<3>(m3)), pair_sel(j == 1, Y3, quad_from<3>(X3)));
}

__device__ __forceinline__ fqx ld_quad(const g1x_xyzz_t* q, uint32_t j) { return ldx(&q->x + j); }
__device__ __forceinline__ void st_quad(g1x_xyzz_t* q, const fqx& a, uint32_t j) { stx(&q->x + j, a); }

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
