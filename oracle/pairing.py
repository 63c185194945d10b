"""BLS12-381 G2 and the pairing on Python ints (oracle; test infrastructure).

Follows backend_new/src/curve.rs: G2Affine :40-48, generator :250-254, pairing :392-394,
multi_pairing :398-404, pairing_check :408-412, and kzg.rs:26-35,86 for the G2 half of the SRS
(g2_generator, g2_tau = tau * [1]_2).  The arithmetic behind those calls is ark-ec 0.4.2 /
ark-bls12-381 0.4.0 (not in the reference tree); it is restated here from the curve's public
definition: Fq2 = Fq[u]/(u^2 + 1); E'(Fq2): y^2 = x^3 + 4(u + 1); Fq12 = Fq[w]/(w^12 - 2 w^6 + 2)
with u = w^6 - 1; the untwist (x, y) -> (x / w^2, y / w^3) into E(Fq12): y^2 = x^3 + 4; the ate
Miller loop over |z| = 0xd201000000010000 and the final exponentiation (q^12 - 1) / r.

The reference holds no GT known-answer vector, so GT values themselves are "parity unpinned"
(this loop yields the inverse of the value an implementation that applies the sign of z would
give); only equalities of pairing products are used, which do not depend on that convention, and
they are pinned by bilinearity, non-degeneracy and e(P, Q)^r = 1 in tests/test_oracle.py, the same
properties curve.rs:535-544 tests.

Fq12 elements are lists of 12 ints (coefficients of w^0..w^11); Fq2 elements are (c0, c1).
"""
from .curve import G1_GEN, affine_neg
from .field import Q_MOD, R_MOD

P = Q_MOD
ATE_LOOP = 0xD201000000010000

# ---------------------------------------------------------------------------------- Fq2, G2
G2_GEN = (
    (0x024AA2B2F08F0A91260805272DC51051C6E47AD4FA403B02B4510B647AE3D1770BAC0326A805BBEFD48056C8C121BDB8,
     0x13E02B6052719F607DACD3A088274F65596BD0D09920B61AB5DA61BBDC7F5049334CF11213945D57E5AC7D055D042B7E),
    (0x0CE5D527727D6E118CC9CDC6DA2E351AADFD9BAA8CBDD3A76D429A695160D12C923AC9CC3BACA289E193548608B82801,
     0x0606C4A02EA734CC32ACD2B02BC28B99CB3E287E85A763AF267492AB572E99AB3F370D275CEC1DA1AAA9075FF05F79BE),
)
B2 = (4, 4)


def f2_add(a, b):
    return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)


def f2_sub(a, b):
    return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)


def f2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def f2_inv(a):
    d = pow(a[0] * a[0] + a[1] * a[1], P - 2, P)
    return (a[0] * d % P, -a[1] * d % P)


def g2_is_on_curve(pt):
    if pt is None:
        return True
    x, y = pt
    return f2_mul(y, y) == f2_add(f2_mul(f2_mul(x, x), x), B2)


def g2_add(p, q):
    """Affine addition on E'(Fq2); None is the identity."""
    if p is None:
        return q
    if q is None:
        return p
    (x1, y1), (x2, y2) = p, q
    if x1 == x2:
        if f2_add(y1, y2) == (0, 0):
            return None
        m = f2_mul(f2_mul((3, 0), f2_mul(x1, x1)), f2_inv(f2_add(y1, y1)))
    else:
        m = f2_mul(f2_sub(y2, y1), f2_inv(f2_sub(x2, x1)))
    x3 = f2_sub(f2_sub(f2_mul(m, m), x1), x2)
    return (x3, f2_sub(f2_mul(m, f2_sub(x1, x3)), y1))


def g2_mul(pt, k):
    """k * pt by double-and-add; k is taken as given (not reduced mod r)."""
    acc, base = None, pt
    while k:
        if k & 1:
            acc = g2_add(acc, base)
        base = g2_add(base, base)
        k >>= 1
    return acc


def g2_neg(pt):
    return None if pt is None else (pt[0], ((-pt[1][0]) % P, (-pt[1][1]) % P))


assert g2_is_on_curve(G2_GEN)


# ---------------------------------------------------------------------------------- Fq12
F12_ONE = [1] + [0] * 11


def f12_mul(a, b):
    t = [0] * 23
    for i, ai in enumerate(a):
        if ai:
            for j, bj in enumerate(b):
                t[i + j] += ai * bj
    for k in range(22, 11, -1):     # w^12 = 2 w^6 - 2
        c = t[k]
        if c:
            t[k - 6] += 2 * c
            t[k - 12] -= 2 * c
    return [x % P for x in t[:12]]


def f12_sub(a, b):
    return [(x - y) % P for x, y in zip(a, b)]


def f12_scalar(a, k):
    return [x * k % P for x in a]


def _deg(p):
    d = len(p) - 1
    while d and p[d] == 0:
        d -= 1
    return d


def f12_inv(a):
    """Extended Euclid on polynomials over Fq against w^12 - 2 w^6 + 2."""
    lm, hm = [1] + [0] * 12, [0] * 13
    low, high = list(a) + [0], [2, 0, 0, 0, 0, 0, P - 2, 0, 0, 0, 0, 0, 1]
    while _deg(low):
        dl, dh = _deg(low), _deg(high)
        inv_lead = pow(low[dl], P - 2, P)
        quo = [0] * 13
        rem = list(high)
        for i in range(dh - dl, -1, -1):
            q = rem[dl + i] * inv_lead % P
            quo[i] = q
            if q:
                for c in range(dl + 1):
                    rem[c + i] = (rem[c + i] - q * low[c]) % P
        nm = list(hm)
        for i in range(13):
            if lm[i]:
                for j in range(13 - i):
                    if quo[j]:
                        nm[i + j] = (nm[i + j] - lm[i] * quo[j]) % P
        lm, low, hm, high = nm, rem, lm, low
    d = pow(low[0], P - 2, P)
    return [x * d % P for x in lm[:12]]


def f12_pow(a, e):
    r, base = F12_ONE, a
    while e:
        if e & 1:
            r = f12_mul(r, base)
        base = f12_mul(base, base)
        e >>= 1
    return r


# ---------------------------------------------------------------------------------- pairing
_W2_INV = f12_inv([0, 0, 1] + [0] * 9)
_W3_INV = f12_inv([0, 0, 0, 1] + [0] * 8)


def _untwist(q):
    """E'(Fq2) -> E(Fq12): a + b u = (a - b) + b w^6, then x / w^2, y / w^3."""
    (x0, x1), (y0, y1) = q
    nx = [(x0 - x1) % P, 0, 0, 0, 0, 0, x1, 0, 0, 0, 0, 0]
    ny = [(y0 - y1) % P, 0, 0, 0, 0, 0, y1, 0, 0, 0, 0, 0]
    return f12_mul(nx, _W2_INV), f12_mul(ny, _W3_INV)


def _line(p1, p2, t):
    """The line through p1 and p2 (tangent if equal) on E(Fq12), evaluated at t; and p1 + p2."""
    (x1, y1), (x2, y2), (xt, yt) = p1, p2, t
    if x1 != x2:
        m = f12_mul(f12_sub(y2, y1), f12_inv(f12_sub(x2, x1)))
    elif y1 == y2:
        m = f12_mul(f12_scalar(f12_mul(x1, x1), 3), f12_inv(f12_scalar(y1, 2)))
    else:
        return f12_sub(xt, x1), None
    val = f12_sub(f12_mul(m, f12_sub(xt, x1)), f12_sub(yt, y1))
    x3 = f12_sub(f12_sub(f12_mul(m, m), x1), x2)
    y3 = f12_sub(f12_mul(m, f12_sub(x1, x3)), y1)
    return val, (x3, y3)


def miller_loop(p, q):
    """p in G1 (affine ints), q in G2 (affine Fq2); either identity gives one."""
    if p is None or q is None:
        return F12_ONE
    Q = _untwist(q)
    T = ([p[0]] + [0] * 11, [p[1]] + [0] * 11)
    R, f = Q, F12_ONE
    for i in range(ATE_LOOP.bit_length() - 2, -1, -1):
        val, R = _line(R, R, T)
        f = f12_mul(f12_mul(f, f), val)
        if (ATE_LOOP >> i) & 1:
            val, R = _line(R, Q, T)
            f = f12_mul(f, val)
    return f


def final_exponentiation(f):
    return f12_pow(f, (P ** 12 - 1) // R_MOD)


def pairing(p, q):
    """curve.rs:392-394."""
    return final_exponentiation(miller_loop(p, q))


def multi_pairing(pairs):
    """curve.rs:398-404: the product of e(P_i, Q_i) under one final exponentiation."""
    f = F12_ONE
    for p, q in pairs:
        f = f12_mul(f, miller_loop(p, q))
    return final_exponentiation(f)


def pairing_check(p1, q1, p2, q2):
    """curve.rs:408-412: e(P1, Q1) == e(P2, Q2), as e(P1, Q1) * e(-P2, Q2) == 1."""
    return multi_pairing([(p1, q1), (affine_neg(p2), q2)]) == F12_ONE
