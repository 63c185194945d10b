"""ctypes wrapper of oracle/cref/libplonk_ref.so, the C restatement of the reference's hot loops
(test infrastructure; CPU baseline).  Arrays are numpy uint64 Montgomery limbs: Fr (n, 4),
affine points (n, 12) as x || y with (0, 0) for the identity."""
import ctypes as C
import os
import subprocess

import numpy as np

from .field import Q_MOD, R_MOD, FQ_MONT_R_INV, fr_to_mont, fr_from_mont, fq_to_mont

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cref")
_SO = os.path.join(_DIR, "libplonk_ref.so")


def build():
    subprocess.check_call(["make", "-s", "-C", _DIR])


def _load():
    if not os.path.exists(_SO):
        build()
    lib = C.CDLL(_SO)
    vp = C.c_void_p
    lib.ref_ntt.argtypes = [vp, C.c_uint, C.c_uint, vp]
    lib.ref_ntt.restype = None
    lib.ref_quotient.argtypes = [vp, C.c_uint64, vp, vp, vp, vp, vp, C.c_uint64, C.c_uint64, vp]
    lib.ref_quotient.restype = None
    lib.ref_msm.argtypes = [vp, vp, C.c_uint64, vp]
    lib.ref_msm.restype = None
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


def fr_array(values):
    buf = b"".join(fr_to_mont(v % R_MOD).to_bytes(32, "little") for v in values)
    return np.frombuffer(buf, dtype=np.uint64).reshape(-1, 4).copy()


def fr_list(arr):
    b = np.ascontiguousarray(arr, dtype=np.uint64).tobytes()
    return [fr_from_mont(int.from_bytes(b[i:i + 32], "little")) for i in range(0, len(b), 32)]


def points_array(points):
    out = bytearray()
    for p in points:
        if p is None:
            out += bytes(96)
        else:
            out += fq_to_mont(p[0]).to_bytes(48, "little") + fq_to_mont(p[1]).to_bytes(48, "little")
    return np.frombuffer(bytes(out), dtype=np.uint64).reshape(-1, 12).copy()


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def ntt(arr, flags=0, shift=None):
    """In place over an (n, 4) array; flags 1 = inverse, 2 = coset."""
    n = arr.shape[0]
    log_n = n.bit_length() - 1
    assert 1 << log_n == n and arr.flags.c_contiguous
    sh = fr_array([shift]) if shift is not None else None
    lib().ref_ntt(_p(arr), log_n, flags, _p(sh) if sh is not None else None)
    return arr


def quotient(ext, n, alpha, beta, gamma, k1, k2, first=0, count=None):
    """ext: (15, 4n, 4) array a b c z z_omega q_m q_l q_r q_o q_c s1 s2 s3 l1 pi."""
    count = 4 * n - first if count is None else count
    out = np.zeros((count, 4), dtype=np.uint64)
    sc = [fr_array([v]) for v in (alpha, beta, gamma, k1, k2)]
    lib().ref_quotient(_p(np.ascontiguousarray(ext)), n, *[_p(s) for s in sc], first, count, _p(out))
    return out


def msm(points_arr, scalars_arr):
    """Returns the affine point (x, y) as canonical ints, or None."""
    n = min(points_arr.shape[0], scalars_arr.shape[0])
    out = np.zeros(18, dtype=np.uint64)
    lib().ref_msm(_p(np.ascontiguousarray(points_arr)), _p(np.ascontiguousarray(scalars_arr)), n, _p(out))
    b = out.tobytes()
    X, Y, Z = [int.from_bytes(b[i * 48:(i + 1) * 48], "little") * FQ_MONT_R_INV % Q_MOD for i in range(3)]
    if Z == 0:
        return None
    zi = pow(Z, Q_MOD - 2, Q_MOD)
    return (X * zi * zi % Q_MOD, Y * zi * zi % Q_MOD * zi % Q_MOD)


class Engine:
    """The transforms and the round-3 loop of oracle.protocol.prove / preprocess run in the C
    restatement (ref_ntt = fft.rs:23-157, ref_quotient = prover.rs:400-457) so that the oracle reaches
    2^18 .. 2^20 gates; everything else of the protocol stays in oracle/protocol.py.  Independent
    transforms and ranges of the 4n-point loop run on `threads` host threads (the C calls release
    the interpreter lock)."""

    def __init__(self, threads=8):
        self.threads = threads

    def _map(self, fn, items):
        if self.threads <= 1 or len(items) <= 1:
            return [fn(x) for x in items]
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(self.threads) as ex:
            return list(ex.map(fn, items))

    def from_evaluations(self, evals, domain):
        """polynomial.rs:56-65 -> fft.rs:42-59 (normalised coefficients)."""
        from .fft import strip
        if len(evals) == 0:
            return []
        assert len(evals) == domain.n
        return strip(fr_list(ntt(fr_array(evals), 1)))

    def quotient_poly(self, n, pk, a, b, c, z, pi, alpha, beta, gamma):
        """prover.rs:364-457 and :497-498: the 15 coset transforms over 4n points (shift 7), the fused
        loop (one pow and one inverse per point), the inverse coset transform; returns the normalised
        coefficients of t."""
        from .fft import strip
        from .protocol import poly_shift
        n4 = 4 * n
        polys = [a, b, c, z, poly_shift(z, pk.omega), pk.q_m, pk.q_l, pk.q_r, pk.q_o, pk.q_c,
                 pk.s_sigma1, pk.s_sigma2, pk.s_sigma3, pk.l1, pi]
        ext = np.zeros((15, n4, 4), dtype=np.uint64)

        def coset(i):
            if polys[i]:
                ext[i, :len(polys[i])] = fr_array(polys[i])
                ntt(ext[i], 2)

        self._map(coset, list(range(15)))
        parts = max(1, self.threads)
        step = -(-n4 // parts)
        ranges = [(f, min(step, n4 - f)) for f in range(0, n4, step)]
        outs = self._map(lambda r: quotient(ext, n, alpha, beta, gamma, pk.k1, pk.k2, r[0], r[1]), ranges)
        t = np.concatenate(outs)
        del ext, outs
        ntt(t, 3)
        return strip(fr_list(t))
