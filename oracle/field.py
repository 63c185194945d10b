"""BLS12-381 scalar field Fr and base field Fq on Python ints (oracle; test infrastructure).

Elements are plain ints in [0, p).  Follows backend_new/src/field.rs (a newtype
over ark_bls12_381::Fr): ops :212-263, inverse :60, pow :65, to_bytes :106-114,
from_bytes :117-125, get_root_of_unity :167-182.
"""

# field.rs:22-23
R_MOD = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
Q_MOD = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB

TWO_ADICITY = 32
# arkworks FrConfig::TWO_ADIC_ROOT_OF_UNITY (= 7^((r-1)/2^32)); used by
# ArkFr::get_root_of_unity, field.rs:180.
TWO_ADIC_ROOT = 10238227357739495823651030575849232062558860180284477541189508159991286009131
assert TWO_ADIC_ROOT == pow(7, (R_MOD - 1) >> 32, R_MOD)


def fr(x):
    return x % R_MOD


def fr_add(a, b):
    return (a + b) % R_MOD


def fr_sub(a, b):
    return (a - b) % R_MOD


def fr_mul(a, b):
    return (a * b) % R_MOD


def fr_neg(a):
    return (-a) % R_MOD


def fr_inv(a):
    """field.rs:60 — inverse(); None for zero in the reference, error here."""
    if a % R_MOD == 0:
        raise ZeroDivisionError("Fr inverse of zero")
    return pow(a, R_MOD - 2, R_MOD)


def fr_pow(a, e):
    """field.rs:65 — pow with a u64 exponent."""
    return pow(a, e, R_MOD)


def fr_to_bytes(a):
    """field.rs:106-114 — canonical (non-Montgomery) value, 32 B little-endian."""
    return int(a % R_MOD).to_bytes(32, "little")


def fr_from_bytes(b):
    """field.rs:117-125 — None unless the LE integer is < r."""
    v = int.from_bytes(b, "little")
    return v if v < R_MOD else None


def fr_to_hex(a):
    """field.rs to_hex: big-endian display of the canonical value, 64 hex chars."""
    return "%064x" % (a % R_MOD)


def root_of_unity(n):
    """field.rs:167-182 — primitive n-th root of unity, n a power of two <= 2^32."""
    if n <= 0 or n & (n - 1):
        return None
    log_n = n.bit_length() - 1
    if log_n > TWO_ADICITY:
        return None
    return pow(TWO_ADIC_ROOT, 1 << (TWO_ADICITY - log_n), R_MOD)


def batch_inverse(xs):
    """Montgomery's trick; mathematically identical to per-element inverse()."""
    n = len(xs)
    pref = [1] * (n + 1)
    for i, x in enumerate(xs):
        pref[i + 1] = pref[i] * x % R_MOD
    inv = fr_inv(pref[n])
    out = [0] * n
    for i in range(n - 1, -1, -1):
        out[i] = inv * pref[i] % R_MOD
        inv = inv * xs[i] % R_MOD
    return out


# ---- Montgomery-form limb helpers (the device ABI carries R = 2^256 Montgomery limbs,
# the in-memory form of ark_ff::Fp256<MontBackend<_,4>>) ----
MONT_R = (1 << 256) % R_MOD
MONT_R_INV = pow(MONT_R, R_MOD - 2, R_MOD)
FQ_MONT_R = (1 << 384) % Q_MOD
FQ_MONT_R_INV = pow(FQ_MONT_R, Q_MOD - 2, Q_MOD)


def fr_to_mont(a):
    return a * MONT_R % R_MOD


def fr_from_mont(a):
    return a * MONT_R_INV % R_MOD


def fq_to_mont(a):
    return a * FQ_MONT_R % Q_MOD


def fq_from_mont(a):
    return a * FQ_MONT_R_INV % Q_MOD
