"""BLS12-381 G1 on Python ints (oracle; test infrastructure).

Follows backend_new/src/curve.rs: G1Affine :33, generator :64, to_compressed_bytes
:84-88, x_hex/y_hex :111-133, G1Projective::scalar_mul :163, msm :169-173.
The arithmetic behind those calls lives in ark-ec 0.4.2 / ark-bls12-381 0.4.0
(not in the reference tree); it is restated here from the curve equation
y^2 = x^3 + 4 over Fq and the ZCash/IETF point encoding ark-bls12-381 uses.

Affine points are (x, y) int tuples, or None for the identity.
Jacobian points are (X, Y, Z) with x = X/Z^2, y = Y/Z^3; Z == 0 is the identity.
"""
from .field import Q_MOD, R_MOD

P = Q_MOD
B_COEFF = 4

G1_GEN = (
    0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB,
    0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1,
)


def is_on_curve(pt):
    if pt is None:
        return True
    x, y = pt
    return (y * y - x * x * x - B_COEFF) % P == 0


assert is_on_curve(G1_GEN)

JAC_ID = (1, 1, 0)


def to_jac(pt):
    return JAC_ID if pt is None else (pt[0], pt[1], 1)


def jac_is_id(p):
    return p[2] == 0


def jac_double(p):
    X, Y, Z = p
    if Z == 0 or Y == 0:
        return JAC_ID
    A = X * X % P
    Bq = Y * Y % P
    C = Bq * Bq % P
    D = 2 * ((X + Bq) * (X + Bq) - A - C) % P
    E = 3 * A % P
    F = E * E % P
    X3 = (F - 2 * D) % P
    Y3 = (E * (D - X3) - 8 * C) % P
    Z3 = 2 * Y * Z % P
    return (X3, Y3, Z3)


def jac_add(p, q):
    X1, Y1, Z1 = p
    X2, Y2, Z2 = q
    if Z1 == 0:
        return q
    if Z2 == 0:
        return p
    Z1Z1 = Z1 * Z1 % P
    Z2Z2 = Z2 * Z2 % P
    U1 = X1 * Z2Z2 % P
    U2 = X2 * Z1Z1 % P
    S1 = Y1 * Z2 * Z2Z2 % P
    S2 = Y2 * Z1 * Z1Z1 % P
    if U1 == U2:
        if S1 == S2:
            return jac_double(p)
        return JAC_ID
    H = (U2 - U1) % P
    Rr = (S2 - S1) % P
    HH = H * H % P
    HHH = H * HH % P
    V = U1 * HH % P
    X3 = (Rr * Rr - HHH - 2 * V) % P
    Y3 = (Rr * (V - X3) - S1 * HHH) % P
    Z3 = Z1 * Z2 * H % P
    return (X3, Y3, Z3)


def jac_add_affine(p, a):
    return jac_add(p, to_jac(a))


def jac_neg(p):
    return (p[0], (-p[1]) % P, p[2])


def to_affine(p):
    """curve.rs:158 — into_affine()."""
    X, Y, Z = p
    if Z == 0:
        return None
    zi = pow(Z, P - 2, P)
    zi2 = zi * zi % P
    return (X * zi2 % P, Y * zi2 * zi % P)


def affine_neg(a):
    return None if a is None else (a[0], (-a[1]) % P)


def scalar_mul(pt, k):
    """curve.rs:163 — scalar * point (double-and-add over the canonical scalar)."""
    k %= R_MOD
    acc = JAC_ID
    base = to_jac(pt) if (pt is None or len(pt) == 2) else pt
    while k:
        if k & 1:
            acc = jac_add(acc, base)
        base = jac_double(base)
        k >>= 1
    return acc


def msm_naive(points, scalars):
    """Definition of curve.rs:169-173: sum of scalars[i] * points[i]."""
    acc = JAC_ID
    for pt, s in zip(points, scalars):
        acc = jac_add(acc, scalar_mul(pt, s))
    return acc


def msm(points, scalars):
    """curve.rs:169-173 — G1Projective::msm: canonical scalars (`into_bigint`), then
    ark-ec's `msm_bigint`, a Pippenger bucket method (window c = 3 below 32 points,
    else ln(n) + 2).  The result is a group element, independent of the window."""
    n = min(len(points), len(scalars))
    if n == 0:
        return JAC_ID
    import math
    c = 3 if n < 32 else int(math.log(n)) + 2
    nbits = 255
    window_sums = []
    for w0 in range(0, nbits, c):
        buckets = [JAC_ID] * ((1 << c) - 1)
        for pt, s in zip(points[:n], scalars[:n]):
            if pt is None:
                continue
            d = ((s % R_MOD) >> w0) & ((1 << c) - 1)
            if d:
                buckets[d - 1] = jac_add_affine(buckets[d - 1], pt)
        running = JAC_ID
        total = JAC_ID
        for b in reversed(buckets):
            running = jac_add(running, b)
            total = jac_add(total, running)
        window_sums.append(total)
    acc = window_sums[-1]
    for ws in reversed(window_sums[:-1]):
        for _ in range(c):
            acc = jac_double(acc)
        acc = jac_add(acc, ws)
    return acc


def compress(pt):
    """curve.rs:84-88 — 48-byte compressed encoding (ZCash/IETF flags as used by
    ark-bls12-381 0.4.0): big-endian x; byte 0 bit 7 = compressed, bit 6 = infinity,
    bit 5 = y is the lexicographically larger root (y > (q-1)/2)."""
    if pt is None:
        return bytes([0xC0] + [0] * 47)
    x, y = pt
    b = bytearray(x.to_bytes(48, "big"))
    b[0] |= 0x80
    if y > (P - 1) // 2:
        b[0] |= 0x20
    return bytes(b)


def decompress(b):
    assert len(b) == 48 and b[0] & 0x80
    if b[0] & 0x40:
        return None
    sign = bool(b[0] & 0x20)
    x = int.from_bytes(bytes([b[0] & 0x1F]) + bytes(b[1:]), "big")
    y2 = (x * x * x + B_COEFF) % P
    y = pow(y2, (P + 1) // 4, P)
    if y * y % P != y2:
        return None
    if (y > (P - 1) // 2) != sign:
        y = P - y
    return (x, y)


def uncompressed(pt):
    """curve.rs:111-133 — serialize_uncompressed: x || y big-endian, 96 B."""
    if pt is None:
        return bytes([0x40] + [0] * 95)
    return pt[0].to_bytes(48, "big") + pt[1].to_bytes(48, "big")
