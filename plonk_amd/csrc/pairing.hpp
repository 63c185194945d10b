// BLS12-381 G2 and the optimal-ate pairing on the host, for the verifier's one pairing check
// (backend_new/src/curve.rs:392-412 pairing / multi_pairing / pairing_check, which call
// ark-bls12-381 0.4.0; restated here from the curve's public definition, not from that crate).
//
// Tower: Fq2 = Fq[u]/(u^2 + 1); Fq6 = Fq2[v]/(v^3 - xi), xi = 1 + u; Fq12 = Fq6[w]/(w^2 - v), so
// w^6 = xi.  Twist E'(Fq2): y^2 = x^3 + 4 xi; the untwist (x, y) -> (x / w^2, y / w^3) lands on
// E(Fq12): y^2 = x^3 + 4.  A line through untwisted points with slope l' on E' has slope l' / w on
// E; evaluated at P = (xP, yP) in G1 and scaled by w^3 it is
//     (l' xR - yR) + (-l' xP) v + (yP) v w,
// three Fq2 coefficients of an Fq12 element.  The scale factor w^3 (like the omitted vertical lines)
// satisfies c^(q^6 - 1) = +-1 and disappears in the final exponentiation, whose remaining exponent
// (q^6 + 1) / r is even.  The Miller loop runs over |z| = 0xd201000000010000 with the running point in
// affine coordinates (one Fq2 inversion per step: this is O(1) work per proof, clarity over speed);
// the final exponentiation is f -> (conj(f) / f)^((q^6 + 1) / r) with the exponent as a constant.
// Only equalities of pairing products are exposed, which no sign or scaling convention can change.
#pragma once
#include "host_field.hpp"

namespace plonk {
namespace host {

struct Fq2 {
    fq_h c0, c1;
};
inline Fq2 f2_zero() { return Fq2{Fq::zero(), Fq::zero()}; }
inline Fq2 f2_one() { return Fq2{Fq::one(), Fq::zero()}; }
inline Fq2 f2_from_fq(const fq_h& a) { return Fq2{a, Fq::zero()}; }
inline bool f2_is_zero(const Fq2& a) { return Fq::is_zero(a.c0) && Fq::is_zero(a.c1); }
inline bool f2_eq(const Fq2& a, const Fq2& b) { return a.c0 == b.c0 && a.c1 == b.c1; }
inline Fq2 f2_add(const Fq2& a, const Fq2& b) { return Fq2{Fq::add(a.c0, b.c0), Fq::add(a.c1, b.c1)}; }
inline Fq2 f2_sub(const Fq2& a, const Fq2& b) { return Fq2{Fq::sub(a.c0, b.c0), Fq::sub(a.c1, b.c1)}; }
inline Fq2 f2_neg(const Fq2& a) { return Fq2{Fq::neg(a.c0), Fq::neg(a.c1)}; }
inline Fq2 f2_mul(const Fq2& a, const Fq2& b) {
    const fq_h t0 = Fq::mul(a.c0, b.c0), t1 = Fq::mul(a.c1, b.c1);
    const fq_h cross = Fq::mul(Fq::add(a.c0, a.c1), Fq::add(b.c0, b.c1));
    return Fq2{Fq::sub(t0, t1), Fq::sub(Fq::sub(cross, t0), t1)};
}
inline Fq2 f2_sqr(const Fq2& a) { return f2_mul(a, a); }
inline Fq2 f2_mul_fq(const Fq2& a, const fq_h& k) { return Fq2{Fq::mul(a.c0, k), Fq::mul(a.c1, k)}; }
inline Fq2 f2_mul_xi(const Fq2& a) { return Fq2{Fq::sub(a.c0, a.c1), Fq::add(a.c0, a.c1)}; }   // (1 + u) a
inline Fq2 f2_inv(const Fq2& a) {   // conj(a) / (c0^2 + c1^2)
    const fq_h d = Fq::inv(Fq::add(Fq::sqr(a.c0), Fq::sqr(a.c1)));
    return Fq2{Fq::mul(a.c0, d), Fq::neg(Fq::mul(a.c1, d))};
}

struct Fq6 {
    Fq2 c0, c1, c2;
};
inline Fq6 f6_zero() { return Fq6{f2_zero(), f2_zero(), f2_zero()}; }
inline Fq6 f6_one() { return Fq6{f2_one(), f2_zero(), f2_zero()}; }
inline bool f6_eq(const Fq6& a, const Fq6& b) { return f2_eq(a.c0, b.c0) && f2_eq(a.c1, b.c1) && f2_eq(a.c2, b.c2); }
inline Fq6 f6_add(const Fq6& a, const Fq6& b) { return Fq6{f2_add(a.c0, b.c0), f2_add(a.c1, b.c1), f2_add(a.c2, b.c2)}; }
inline Fq6 f6_sub(const Fq6& a, const Fq6& b) { return Fq6{f2_sub(a.c0, b.c0), f2_sub(a.c1, b.c1), f2_sub(a.c2, b.c2)}; }
inline Fq6 f6_neg(const Fq6& a) { return Fq6{f2_neg(a.c0), f2_neg(a.c1), f2_neg(a.c2)}; }
inline Fq6 f6_mul(const Fq6& a, const Fq6& b) {   // v^3 = xi
    const Fq2 a0b0 = f2_mul(a.c0, b.c0), a1b1 = f2_mul(a.c1, b.c1), a2b2 = f2_mul(a.c2, b.c2);
    Fq6 r;
    r.c0 = f2_add(a0b0, f2_mul_xi(f2_add(f2_mul(a.c1, b.c2), f2_mul(a.c2, b.c1))));
    r.c1 = f2_add(f2_add(f2_mul(a.c0, b.c1), f2_mul(a.c1, b.c0)), f2_mul_xi(a2b2));
    r.c2 = f2_add(f2_add(f2_mul(a.c0, b.c2), f2_mul(a.c2, b.c0)), a1b1);
    return r;
}
inline Fq6 f6_mul_v(const Fq6& a) { return Fq6{f2_mul_xi(a.c2), a.c0, a.c1}; }
inline Fq6 f6_inv(const Fq6& a) {
    const Fq2 t0 = f2_sub(f2_sqr(a.c0), f2_mul_xi(f2_mul(a.c1, a.c2)));
    const Fq2 t1 = f2_sub(f2_mul_xi(f2_sqr(a.c2)), f2_mul(a.c0, a.c1));
    const Fq2 t2 = f2_sub(f2_sqr(a.c1), f2_mul(a.c0, a.c2));
    const Fq2 d = f2_inv(f2_add(f2_mul(a.c0, t0), f2_mul_xi(f2_add(f2_mul(a.c2, t1), f2_mul(a.c1, t2)))));
    return Fq6{f2_mul(t0, d), f2_mul(t1, d), f2_mul(t2, d)};
}

struct Fq12 {
    Fq6 c0, c1;
};
inline Fq12 f12_one() { return Fq12{f6_one(), f6_zero()}; }
inline bool f12_eq(const Fq12& a, const Fq12& b) { return f6_eq(a.c0, b.c0) && f6_eq(a.c1, b.c1); }
inline Fq12 f12_mul(const Fq12& a, const Fq12& b) {   // w^2 = v
    const Fq6 t0 = f6_mul(a.c0, b.c0), t1 = f6_mul(a.c1, b.c1);
    return Fq12{f6_add(t0, f6_mul_v(t1)), f6_add(f6_mul(a.c0, b.c1), f6_mul(a.c1, b.c0))};
}
inline Fq12 f12_conj(const Fq12& a) { return Fq12{a.c0, f6_neg(a.c1)}; }   // the q^6-power map
inline Fq12 f12_inv(const Fq12& a) {
    const Fq6 d = f6_inv(f6_sub(f6_mul(a.c0, a.c0), f6_mul_v(f6_mul(a.c1, a.c1))));
    return Fq12{f6_mul(a.c0, d), f6_neg(f6_mul(a.c1, d))};
}

// ---- G2: affine points of E'(Fq2) ----
struct G2AffineH {
    Fq2 x, y;
    bool inf;
};
inline G2AffineH g2_infinity() { return G2AffineH{f2_zero(), f2_zero(), true}; }
inline G2AffineH g2_generator() {   // canonical coordinates of the standard generator (curve.rs:250-254)
    static const uint64_t X0[6] = {0xd48056c8c121bdb8ull, 0x0bac0326a805bbefull, 0xb4510b647ae3d177ull,
                                   0xc6e47ad4fa403b02ull, 0x260805272dc51051ull, 0x024aa2b2f08f0a91ull};
    static const uint64_t X1[6] = {0xe5ac7d055d042b7eull, 0x334cf11213945d57ull, 0xb5da61bbdc7f5049ull,
                                   0x596bd0d09920b61aull, 0x7dacd3a088274f65ull, 0x13e02b6052719f60ull};
    static const uint64_t Y0[6] = {0xe193548608b82801ull, 0x923ac9cc3baca289ull, 0x6d429a695160d12cull,
                                   0xadfd9baa8cbdd3a7ull, 0x8cc9cdc6da2e351aull, 0x0ce5d527727d6e11ull};
    static const uint64_t Y1[6] = {0xaaa9075ff05f79beull, 0x3f370d275cec1da1ull, 0x267492ab572e99abull,
                                   0xcb3e287e85a763afull, 0x32acd2b02bc28b99ull, 0x0606c4a02ea734ccull};
    auto mont = [](const uint64_t* v) {
        fq_h c;
        memcpy(c.l, v, 48);
        return Fq::to_mont(c);
    };
    return G2AffineH{Fq2{mont(X0), mont(X1)}, Fq2{mont(Y0), mont(Y1)}, false};
}
inline bool g2_on_curve(const G2AffineH& p) {
    if (p.inf) return true;
    const Fq2 b = f2_mul_xi(f2_from_fq(Fq::from_u64(4)));
    return f2_eq(f2_sqr(p.y), f2_add(f2_mul(f2_sqr(p.x), p.x), b));
}
inline G2AffineH g2_add(const G2AffineH& p, const G2AffineH& q) {
    if (p.inf) return q;
    if (q.inf) return p;
    Fq2 m;
    if (f2_eq(p.x, q.x)) {
        if (f2_is_zero(f2_add(p.y, q.y))) return g2_infinity();
        const Fq2 xx = f2_sqr(p.x);
        m = f2_mul(f2_add(f2_add(xx, xx), xx), f2_inv(f2_add(p.y, p.y)));
    } else {
        m = f2_mul(f2_sub(q.y, p.y), f2_inv(f2_sub(q.x, p.x)));
    }
    G2AffineH r;
    r.x = f2_sub(f2_sub(f2_sqr(m), p.x), q.x);
    r.y = f2_sub(f2_mul(m, f2_sub(p.x, r.x)), p.y);
    r.inf = false;
    return r;
}
// scalar given as canonical little-endian limbs (kzg.rs:86: g2_tau = g2_generator * tau)
inline G2AffineH g2_mul_limbs(const G2AffineH& p, const uint64_t* e, int nl) {
    G2AffineH r = g2_infinity();
    for (int i = nl * 64 - 1; i >= 0; --i) {
        r = g2_add(r, r);
        if ((e[i / 64] >> (i % 64)) & 1) r = g2_add(r, p);
    }
    return r;
}
inline G2AffineH g2_neg(const G2AffineH& p) {
    if (p.inf) return p;
    return G2AffineH{p.x, f2_neg(p.y), false};
}
// membership in the order-r subgroup: [r] Q = O, with r the modulus of Fr
inline bool g2_in_subgroup(const G2AffineH& p) {
    const uint64_t r[4] = {0xffffffff00000001ull, 0x53bda402fffe5bfeull, 0x3339d80809a1d805ull, 0x73eda753299d7d48ull};
    return g2_mul_limbs(p, r, 4).inf;
}

// ---- pairing ----
inline Fq12 line_value(const Fq2& slope, const G2AffineH& r, const G1AffineH& p) {
    Fq12 l;
    l.c0 = Fq6{f2_sub(f2_mul(slope, r.x), r.y), f2_neg(f2_mul_fq(slope, p.x)), f2_zero()};
    l.c1 = Fq6{f2_zero(), f2_from_fq(p.y), f2_zero()};
    return l;
}

inline Fq12 miller_loop(const G1AffineH& p, const G2AffineH& q) {
    if (p.inf || q.inf) return f12_one();
    const uint64_t z = 0xd201000000010000ull;
    G2AffineH r = q;
    Fq12 f = f12_one();
    for (int i = 62; i >= 0; --i) {   // bit 63 is the leading one
        const Fq2 xx = f2_sqr(r.x);
        const Fq2 tangent = f2_mul(f2_add(f2_add(xx, xx), xx), f2_inv(f2_add(r.y, r.y)));
        f = f12_mul(f12_mul(f, f), line_value(tangent, r, p));
        r = g2_add(r, r);
        if ((z >> i) & 1) {
            const Fq2 chord = f2_mul(f2_sub(q.y, r.y), f2_inv(f2_sub(q.x, r.x)));
            f = f12_mul(f, line_value(chord, r, p));
            r = g2_add(r, q);
        }
    }
    return f12_conj(f);   // z is negative
}

inline Fq12 final_exponentiation(const Fq12& f) {
    // (q^6 + 1) / r, little-endian 64-bit limbs
    static const uint64_t E[32] = {
        0x8739e1cdc0705d6aull, 0x09a5256de0381a16ull, 0x9cf0f70a61c791e2ull, 0x3a09c4497903f76eull,
        0x2d7271563890f133ull, 0x224741b36fec7760ull, 0x338259c22a12bd40ull, 0x38ee1cd4778e0de7ull,
        0xc3b5ef4b188a20b0ull, 0x1d615d49e2764d7bull, 0x816101ddd076117dull, 0xf007c01e7ebe3afcull,
        0x27d7bd90935021c3ull, 0xc3b5e2f557c0b15full, 0x5e886c94c4f82384ull, 0xee6a95db11e63f56ull,
        0x2b822f514a9c4f6full, 0x12d6a874d21b73daull, 0x1304275ef499dffbull, 0x967878febcb95d1full,
        0x4744497f8b2f2922ull, 0x85a2e707f0841855ull, 0x9f0c50126c802eecull, 0xfb46e197bd2fa489ull,
        0x548ce0809bc5f61aull, 0xcf56fb1573beaa8cull, 0xad7375a3763bdf7cull, 0xe0ec9031179bdeccull,
        0x6579aea83c48c1daull, 0xdbf85ae664cf5bb3ull, 0x7b6f235c55ca7566ull, 0x000028b314877503ull};
    const Fq12 g = f12_mul(f12_conj(f), f12_inv(f));   // f^(q^6 - 1)
    Fq12 r = f12_one();
    for (int i = 32 * 64 - 1; i >= 0; --i) {
        r = f12_mul(r, r);
        if ((E[i / 64] >> (i % 64)) & 1) r = f12_mul(r, g);
    }
    return r;
}

// curve.rs:408-412: e(P1, Q1) == e(P2, Q2), as e(P1, Q1) * e(-P2, Q2) == 1
inline bool pairing_check(const G1AffineH& p1, const G2AffineH& q1, const G1AffineH& p2, const G2AffineH& q2) {
    G1AffineH np2 = p2;
    if (!np2.inf) np2.y = Fq::neg(np2.y);
    const Fq12 f = f12_mul(miller_loop(p1, q1), miller_loop(np2, q2));
    return f12_eq(final_exponentiation(f), f12_one());
}

}  // namespace host
}  // namespace plonk
