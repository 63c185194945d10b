"""Host-side mirror of the reference's prover interface over the C ABI.

Names and argument meaning follow backend_new/src: fft.rs (fft / ifft / coset_fft / coset_ifft,
Domain), kzg.rs (Srs, Commitment::commit), curve.rs (G1Projective::msm), plonk/preprocess.rs
(preprocess -> ProvingKey) and plonk/prover.rs (Prover::new / prove).  Field elements cross this
layer as Python ints (canonical value); bulk vectors as numpy uint64 arrays of shape (n, 4)
holding Montgomery limbs, the in-memory image the C ABI uses.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import lib, PlonkError

R_MOD = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
Q_MOD = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
_R = (1 << 256) % R_MOD
_R_INV = pow(_R, R_MOD - 2, R_MOD)
_RQ = (1 << 384) % Q_MOD
_RQ_INV = pow(_RQ, Q_MOD - 2, Q_MOD)


def fr_to_array(values):
    """ints (canonical) -> (n, 4) uint64 Montgomery limbs."""
    buf = b"".join(((v % R_MOD) * _R % R_MOD).to_bytes(32, "little") for v in values)
    return np.frombuffer(buf, dtype=np.uint64).reshape(-1, 4).copy()


def fr_from_array(arr):
    """(n, 4) uint64 Montgomery limbs -> list of canonical ints."""
    b = np.ascontiguousarray(arr, dtype=np.uint64).tobytes()
    return [int.from_bytes(b[i:i + 32], "little") * _R_INV % R_MOD for i in range(0, len(b), 32)]


def _fr_struct(v):
    s = _lib.Fr()
    m = (v % R_MOD) * _R % R_MOD
    for i in range(4):
        s.l[i] = (m >> (64 * i)) & 0xFFFFFFFFFFFFFFFF
    return s


def _fr_value(s):
    return sum(int(s.l[i]) << (64 * i) for i in range(4)) * _R_INV % R_MOD


def _g1_struct(pt):
    g = _lib.G1()
    if pt is None:
        g.infinity = 1
        return g
    x = pt[0] * _RQ % Q_MOD
    y = pt[1] * _RQ % Q_MOD
    for i in range(6):
        g.x[i] = (x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF
        g.y[i] = (y >> (64 * i)) & 0xFFFFFFFFFFFFFFFF
    return g


def _g1_value(g):
    """plonk_g1 -> (x, y) canonical ints, or None for the identity."""
    if g.infinity:
        return None
    x = sum(int(g.x[i]) << (64 * i) for i in range(6)) * _RQ_INV % Q_MOD
    y = sum(int(g.y[i]) << (64 * i) for i in range(6)) * _RQ_INV % Q_MOD
    return (x, y)


def _ptr(arr):
    return arr.ctypes.data_as(C.c_void_p)


def _as_limbs(x, n=None):
    """Accept a list of ints or an (n, 4) uint64 array; return a contiguous (n, 4) uint64 array."""
    if isinstance(x, np.ndarray):
        a = np.ascontiguousarray(x, dtype=np.uint64)
        assert a.ndim == 2 and a.shape[1] == 4
    else:
        a = fr_to_array(list(x))
    if n is not None and a.shape[0] != n:
        raise PlonkError(_lib.PLONK_E_INVALID, "expected %d field elements, got %d" % (n, a.shape[0]))
    return a


class Context:
    """One HIP device + stream (plonk_ctx)."""

    def __init__(self, device=0):
        h = C.c_void_p()
        rc = lib.plonk_ctx_create(device, C.byref(h))
        if rc != 0:
            raise PlonkError(rc, "plonk_ctx_create(device=%d) failed (no usable HIP device?)" % device)
        self.h = h
        self.device = device

    def check(self, rc):
        if rc != 0:
            raise PlonkError(rc, (lib.plonk_last_error(self.h) or b"").decode())

    def close(self):
        if getattr(self, "h", None):
            lib.plonk_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        self.check(lib.plonk_sync(self.h))

    def stream(self):
        return lib.plonk_ctx_stream(self.h)

    # raw device memory
    def alloc(self, nbytes):
        p = C.c_void_p()
        self.check(lib.plonk_dev_alloc(self.h, nbytes, C.byref(p)))
        return p

    def free(self, p):
        self.check(lib.plonk_dev_free(self.h, p))

    def upload(self, dev, arr):
        arr = np.ascontiguousarray(arr)
        self.check(lib.plonk_dev_upload(self.h, dev, _ptr(arr), arr.nbytes))

    def download(self, dev, nbytes):
        out = np.empty(nbytes // 8, dtype=np.uint64)
        self.check(lib.plonk_dev_download(self.h, _ptr(out), dev, nbytes))
        return out

    def ntt(self, data, flags=0, shift=None, batch=1):
        """In-place NTT of a host (batch*n, 4) uint64 array."""
        n = data.shape[0] // batch
        log_n = n.bit_length() - 1
        if n <= 0 or (1 << log_n) != n:
            raise PlonkError(_lib.PLONK_E_INVALID, "FFT size must be a power of 2")
        sh = C.byref(_fr_struct(shift)) if shift is not None else None
        self.check(lib.plonk_ntt(self.h, _ptr(data), log_n, flags, sh, batch, n))
        return data

    def ntt_dev(self, dev, log_n, flags=0, shift=None, batch=1, stride=None):
        sh = C.byref(_fr_struct(shift)) if shift is not None else None
        self.check(lib.plonk_ntt_dev(self.h, dev, log_n, flags, sh, batch, stride or (1 << log_n)))

    def timer_start(self):
        self.check(lib.plonk_timer_start(self.h))

    def timer_stop(self):
        v = C.c_double()
        self.check(lib.plonk_timer_stop(self.h, C.byref(v)))
        return v.value

    def msm_set_timing(self, on):
        self.check(lib.plonk_msm_set_timing(self.h, 1 if on else 0))

    def msm_last_timing(self):
        """Device-time breakdown (ms) of the last MSM on this context and its mixed-addition count."""
        t = _lib.MsmTiming()
        self.check(lib.plonk_msm_last_timing(self.h, C.byref(t)))
        return {"sort_ms": t.sort_ms, "big_bucket_ms": t.big_bucket_ms, "bucket_ms": t.bucket_ms, "reduce_ms": t.reduce_ms,
                "entries": t.entries, "window_bits": t.window_bits, "bucket_sets": t.bucket_sets}

    def quotient_last_timing(self):
        v = C.c_double()
        self.check(lib.plonk_quotient_last_timing(self.h, C.byref(v)))
        return v.value

    def bench_field_mul(self, which):
        v = C.c_double()
        self.check(lib.plonk_bench_field_mul(self.h, which, C.byref(v)))
        return v.value


_default_ctx = None


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


def _pad(coeffs, n):
    c = list(coeffs)
    if len(c) > n:
        raise PlonkError(_lib.PLONK_E_INVALID, "polynomial longer than the domain")
    return c + [0] * (n - len(c))


def _strip(c):
    while c and c[-1] == 0:
        c.pop()
    return c


def fft(poly, omega, n, ctx=None):
    """fft.rs:23 -- evaluations of `poly` (coefficient list) on {omega^i}; omega must be the
    library's primitive n-th root (Fr::get_root_of_unity(n))."""
    ctx = ctx or default_context()
    _check_omega(omega, n)
    return fr_from_array(ctx.ntt(fr_to_array(_pad(poly, n)), 0))


def ifft(evals, omega, ctx=None):
    """fft.rs:42 -- normalised coefficient list."""
    ctx = ctx or default_context()
    _check_omega(omega, len(evals))
    return _strip(fr_from_array(ctx.ntt(fr_to_array(evals), _lib.NTT_INVERSE)))


def coset_fft(poly, omega, k, n, ctx=None):
    """fft.rs:121."""
    ctx = ctx or default_context()
    _check_omega(omega, n)
    return fr_from_array(ctx.ntt(fr_to_array(_pad(poly, n)), _lib.NTT_COSET, shift=k))


def coset_ifft(evals, omega, k, ctx=None):
    """fft.rs:140 -- n coefficients (zero padded)."""
    ctx = ctx or default_context()
    _check_omega(omega, len(evals))
    return fr_from_array(ctx.ntt(fr_to_array(evals), _lib.NTT_COSET | _lib.NTT_INVERSE, shift=k))


_TWO_ADIC_ROOT = pow(7, (R_MOD - 1) >> 32, R_MOD)


def root_of_unity(n):
    """field.rs:167-182."""
    if n <= 0 or n & (n - 1) or n > (1 << 32):
        return None
    return pow(_TWO_ADIC_ROOT, (1 << 32) // n, R_MOD)


def _check_omega(omega, n):
    if n <= 0 or n & (n - 1):
        raise PlonkError(_lib.PLONK_E_INVALID, "FFT size must be a power of 2")
    if omega % R_MOD != root_of_unity(n):
        raise PlonkError(_lib.PLONK_E_INVALID, "omega is not the canonical primitive n-th root of unity")


class Domain:
    """fft.rs:175-249."""

    def __init__(self, n, ctx=None):
        if n <= 0 or n & (n - 1):
            raise PlonkError(_lib.PLONK_E_INVALID, "domain size must be a power of 2")
        self.n = n
        self.ctx = ctx or default_context()
        self.omega = root_of_unity(n)
        self.omega_inv = pow(self.omega, R_MOD - 2, R_MOD)
        self.n_inv = pow(n, R_MOD - 2, R_MOD)

    def fft(self, poly):
        return fft(poly, self.omega, self.n, self.ctx)

    def ifft(self, evals):
        return ifft(evals, self.omega, self.ctx)

    def coset_fft(self, poly, k):
        return coset_fft(poly, self.omega, k, self.n, self.ctx)

    def coset_ifft(self, evals, k):
        return coset_ifft(evals, self.omega, k, self.ctx)


class Srs:
    """kzg.rs:26-35 -- g1_powers resident on the device."""

    def __init__(self, handle, ctx):
        self.h = handle
        self.ctx = ctx

    @classmethod
    def from_points(cls, points, ctx=None):
        """points: list of affine (x, y) ints or None."""
        ctx = ctx or default_context()
        arr = (_lib.G1 * len(points))(*[_g1_struct(p) for p in points])
        h = C.c_void_p()
        ctx.check(lib.plonk_srs_upload(ctx.h, arr, len(points), C.byref(h)))
        return cls(h, ctx)

    @classmethod
    def from_tau(cls, tau, max_degree, ctx=None):
        """kzg.rs:71-94 -- [tau^i]_1 for i = 0..=max_degree, generated on the device."""
        ctx = ctx or default_context()
        h = C.c_void_p()
        t = _fr_struct(tau)
        ctx.check(lib.plonk_srs_from_tau(ctx.h, C.byref(t), max_degree + 1, C.byref(h)))
        return cls(h, ctx)

    @classmethod
    def from_tau_range(cls, tau, first, end, ctx=None):
        """[tau^i]_1 for i in [first, end): one rank's slice of the point range."""
        ctx = ctx or default_context()
        h = C.c_void_p()
        t = _fr_struct(tau)
        ctx.check(lib.plonk_srs_from_tau_range(ctx.h, C.byref(t), first, end - first, C.byref(h)))
        return cls(h, ctx)

    def commit_dev(self, coeffs_dev, length):
        """Commitment::commit over a coefficient vector already on the device."""
        out = _lib.G1()
        self.ctx.check(lib.plonk_commit_dev(self.ctx.h, self.h, coeffs_dev, length, C.byref(out)))
        return _g1_value(out)

    def __len__(self):
        return lib.plonk_srs_len(self.h)

    @property
    def max_degree(self):
        return len(self) - 1

    def g1_powers(self, first=0, count=None):
        count = len(self) - first if count is None else count
        arr = (_lib.G1 * count)()
        self.ctx.check(lib.plonk_srs_download(self.ctx.h, self.h, first, count, arr))
        return [_g1_value(g) for g in arr]

    def close(self):
        if getattr(self, "h", None):
            lib.plonk_srs_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Commitment:
    """kzg.rs:229-245."""

    def __init__(self, point):
        self.point = point

    @staticmethod
    def commit(poly, srs):
        a = _as_limbs(poly)
        out = _lib.G1()
        srs.ctx.check(lib.plonk_commit(srs.ctx.h, srs.h, _ptr(a), a.shape[0], C.byref(out)))
        return Commitment(_g1_value(out))


class Polynomial:
    """polynomial.rs:13-287 for a coefficient vector resident on the device (low degree first): evaluate (:103-113)
    and div_by_linear (:188-204) run on the GPU; the vector comes back to the host only on request."""

    def __init__(self, coeffs, ctx=None):
        self.ctx = ctx or default_context()
        a = _as_limbs(coeffs) if len(coeffs) else np.zeros((0, 4), dtype=np.uint64)
        self.len = a.shape[0]
        self.dev = self.ctx.alloc(max(self.len, 1) * 32)
        if self.len:
            self.ctx.upload(self.dev, a)

    @classmethod
    def _adopt(cls, ctx, dev, length):
        p = cls.__new__(cls)
        p.ctx, p.dev, p.len = ctx, dev, length
        return p

    def coeffs(self):
        """Normalised coefficient list (polynomial.rs:91-99)."""
        if not self.len:
            return []
        return _strip(fr_from_array(self.ctx.download(self.dev, self.len * 32).reshape(-1, 4)))

    def evaluate(self, z):
        y = _lib.Fr()
        t = _fr_struct(z)
        self.ctx.check(lib.plonk_poly_eval_dev(self.ctx.h, self.dev, self.len, C.byref(t), C.byref(y)))
        return _fr_value(y)

    def div_by_linear(self, z):
        qlen = max(self.len - 1, 0)
        q = self.ctx.alloc(max(qlen, 1) * 32)
        t = _fr_struct(z)
        self.ctx.check(lib.plonk_div_by_linear_dev(self.ctx.h, self.dev, self.len, C.byref(t), q))
        return Polynomial._adopt(self.ctx, q, qlen)

    def close(self):
        if getattr(self, "dev", None):
            self.ctx.free(self.dev)
            self.dev = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class OpeningProof:
    """kzg.rs:258-304."""

    def __init__(self, point):
        self.point = point

    @staticmethod
    def create(poly, z, srs):
        """(y, proof) with y = f(z) and proof = [(f - y) / (X - z)]_1 (kzg.rs:262-276); `poly` a Polynomial."""
        y, out = _lib.Fr(), _lib.G1()
        t = _fr_struct(z)
        srs.ctx.check(lib.plonk_open_dev(srs.ctx.h, srs.h, poly.dev, poly.len, C.byref(t), C.byref(y), C.byref(out)))
        return _fr_value(y), OpeningProof(_g1_value(out))

    @staticmethod
    def verify(commitment, z, y, proof, g2_tau):
        """kzg.rs:282-303 against [tau]_2 (a plonk_g2, e.g. from g2_mul(g2_generator(), tau))."""
        ok = C.c_int32()
        c, w = _g1_struct(commitment.point), _g1_struct(proof.point)
        zz, yy = _fr_struct(z), _fr_struct(y)
        rc = lib.plonk_open_verify(C.byref(c), C.byref(zz), C.byref(yy), C.byref(w), C.byref(g2_tau), C.byref(ok))
        if rc != 0:
            raise PlonkError(rc, "plonk_open_verify: invalid input")
        return bool(ok.value)


class BatchOpening:
    """kzg.rs:310-379."""

    def __init__(self, proof, evaluations):
        self.proof, self.evaluations = proof, evaluations

    @staticmethod
    def create(polys, z, gamma, srs):
        """kzg.rs:322-341: y_i = f_i(z) and one opening of sum_i gamma^i f_i at z."""
        k = len(polys)
        ptrs = (C.c_void_p * max(k, 1))(*[p.dev for p in polys])
        lens = (C.c_size_t * max(k, 1))(*[p.len for p in polys])
        ys = (_lib.Fr * max(k, 1))()
        out = _lib.G1()
        zz, gg = _fr_struct(z), _fr_struct(gamma)
        srs.ctx.check(lib.plonk_open_batch_dev(srs.ctx.h, srs.h, ptrs, lens, k, C.byref(zz), C.byref(gg), ys, C.byref(out)))
        return BatchOpening(OpeningProof(_g1_value(out)), [_fr_value(ys[i]) for i in range(k)])

    def verify(self, commitments, z, gamma, g2_tau):
        """kzg.rs:344-378: the combined commitment sum gamma^i [f_i]_1 (an MSM on the device) and combined evaluation,
        then OpeningProof::verify."""
        if len(commitments) != len(self.evaluations):
            return False
        gp, powers = 1, []
        for _ in commitments:
            powers.append(gp)
            gp = gp * gamma % R_MOD
        combined = Commitment(msm([c.point for c in commitments], powers))
        y = sum(g * e for g, e in zip(powers, self.evaluations)) % R_MOD
        return OpeningProof.verify(combined, z, y, self.proof, g2_tau)


def msm_window_bits(n):
    return lib.plonk_msm_window_bits(n)


def msm(points, scalars, ctx=None):
    """curve.rs:169-173 -- sum scalars[i] * points[i]; returns the affine point (or None)."""
    ctx = ctx or default_context()
    n = min(len(points), len(scalars))
    arr = (_lib.G1 * max(n, 1))(*[_g1_struct(p) for p in points[:n]])
    sc = _as_limbs(scalars[:n]) if n else np.zeros((1, 4), dtype=np.uint64)
    out = _lib.G1()
    ctx.check(lib.plonk_msm(ctx.h, arr, _ptr(sc), n, C.byref(out)))
    return _g1_value(out)


PROOF_G1 = ("a_comm", "b_comm", "c_comm", "z_comm", "t_lo_comm", "t_mid_comm", "t_hi_comm",
            "w_zeta_comm", "w_zeta_omega_comm")
PROOF_FR = ("a_eval", "b_eval", "c_eval", "s_sigma1_eval", "s_sigma2_eval", "z_omega_eval")


class Proof:
    """plonk/types.rs:74-114."""

    def __init__(self, raw):
        self.raw = raw

    def to_dict(self):
        d = {k: _g1_value(getattr(self.raw, k)) for k in PROOF_G1}
        d.update({k: _fr_value(getattr(self.raw, k)) for k in PROOF_FR})
        return d

    def to_bytes(self):
        out = (C.c_uint8 * _lib.PROOF_WIRE_BYTES)()
        rc = lib.plonk_proof_serialize(C.byref(self.raw), out)
        if rc != 0:
            raise PlonkError(rc, "plonk_proof_serialize")
        return bytes(out)

    @classmethod
    def from_bytes(cls, data):
        """Inverse of to_bytes; every point is validated as G1Affine::from_compressed_bytes does
        (curve.rs:90-93)."""
        if len(data) != _lib.PROOF_WIRE_BYTES:
            raise ValueError("proof: expected %d bytes" % _lib.PROOF_WIRE_BYTES)
        raw = _lib.Proof()
        buf = (C.c_uint8 * _lib.PROOF_WIRE_BYTES).from_buffer_copy(data)
        rc = lib.plonk_proof_deserialize(buf, C.byref(raw))
        if rc != 0:
            raise PlonkError(rc, "plonk_proof_deserialize")
        return cls(raw)

    def to_json(self):
        """serde_json's text of CompactProof (trace.rs:255-332), produced by the library."""
        cap = 8192
        buf = C.create_string_buffer(cap)
        ln = C.c_size_t()
        rc = lib.plonk_proof_to_json(C.byref(self.raw), buf, cap, C.byref(ln))
        if rc != 0:
            raise PlonkError(rc, "plonk_proof_to_json")
        return buf.raw[:ln.value].decode()

    @classmethod
    def from_json(cls, text):
        """Inverse of to_json (reads `compressed` of every point and `full` of every scalar)."""
        data = text.encode() if isinstance(text, str) else bytes(text)
        raw = _lib.Proof()
        rc = lib.plonk_proof_from_json(data, len(data), C.byref(raw))
        if rc != 0:
            raise PlonkError(rc, "plonk_proof_from_json: malformed CompactProof")
        return cls(raw)

    def to_compact(self):
        """The CompactProof JSON object of trace.rs:255-332: points as {compressed, x, y}
        (curve.rs:418-434: 0x-hex of the 48-byte compressed form and of the two big-endian
        coordinates, "infinity" for the identity), scalars as {full, short} (field.rs:294-308:
        0x-hex of the canonical value, big-endian, and its first 10 / last 8 characters)."""
        d = self.to_dict()

        def g1(key):
            pt = d[key]
            comp = (C.c_uint8 * 48)()
            rc = lib.plonk_g1_compress(C.byref(getattr(self.raw, key)), comp)
            if rc != 0:
                raise PlonkError(rc, "plonk_g1_compress")
            xy = ("infinity", "infinity") if pt is None else tuple("0x%096x" % v for v in pt)
            return {"compressed": "0x" + bytes(comp).hex(), "x": xy[0], "y": xy[1]}

        def fr(key):
            full = "0x%064x" % d[key]
            return {"full": full, "short": full[:10] + "..." + full[-8:]}

        return {
            "round1": {"a": g1("a_comm"), "b": g1("b_comm"), "c": g1("c_comm")},
            "round2": {"z": g1("z_comm")},
            "round3": {"t_lo": g1("t_lo_comm"), "t_mid": g1("t_mid_comm"), "t_hi": g1("t_hi_comm")},
            "round4": {"a_zeta": fr("a_eval"), "b_zeta": fr("b_eval"), "c_zeta": fr("c_eval"),
                       "s_sigma1_zeta": fr("s_sigma1_eval"), "s_sigma2_zeta": fr("s_sigma2_eval"),
                       "z_omega_zeta": fr("z_omega_eval")},
            "round5": {"w_zeta": g1("w_zeta_comm"), "w_zeta_omega": g1("w_zeta_omega_comm")},
        }

    @classmethod
    def from_compact(cls, obj):
        """What serde's Deserialize of CompactProof reads: `compressed` of every point
        (curve.rs:436-450) and `full` of every scalar (field.rs:310-320)."""
        def g1(o):
            h = o["compressed"]
            return bytes.fromhex(h[2:] if h.startswith("0x") else h)

        def fr(o):
            h = o["full"]
            return int(h[2:] if h.startswith("0x") else h, 16).to_bytes(32, "little")

        r1, r3, r4, r5 = obj["round1"], obj["round3"], obj["round4"], obj["round5"]
        wire = b"".join([g1(r1["a"]), g1(r1["b"]), g1(r1["c"]), g1(obj["round2"]["z"]),
                         g1(r3["t_lo"]), g1(r3["t_mid"]), g1(r3["t_hi"]),
                         fr(r4["a_zeta"]), fr(r4["b_zeta"]), fr(r4["c_zeta"]), fr(r4["s_sigma1_zeta"]),
                         fr(r4["s_sigma2_zeta"]), fr(r4["z_omega_zeta"]),
                         g1(r5["w_zeta"]), g1(r5["w_zeta_omega"])])
        return cls.from_bytes(wire)

    def challenges(self, public_inputs=()):
        """beta, gamma, alpha, zeta, v, u re-derived from the proof (verifier.rs:51-92)."""
        pis = (_lib.Fr * max(len(public_inputs), 1))(*[_fr_struct(p) for p in public_inputs])
        out = (_lib.Fr * 6)()
        rc = lib.plonk_proof_challenges(C.byref(self.raw), pis, len(public_inputs), out)
        if rc != 0:
            raise PlonkError(rc, "plonk_proof_challenges")
        return [_fr_value(x) for x in out]


class ProvingKey:
    """plonk/preprocess.rs:104-186 + plonk/types.rs:12 -- built on the device from the selector and
    S_sigma evaluation vectors."""

    def __init__(self, n, selectors, sigmas, k1, k2, public_input_positions=(), ctx=None):
        ctx = ctx or default_context()
        log_n = n.bit_length() - 1
        if n <= 0 or (1 << log_n) != n:
            raise PlonkError(_lib.PLONK_E_INVALID, "n must be a power of 2")
        arrs = [_as_limbs(v, n) for v in list(selectors) + list(sigmas)]
        pos = np.asarray(list(public_input_positions), dtype=np.uint32)
        h = C.c_void_p()
        k1s, k2s = _fr_struct(k1), _fr_struct(k2)
        ctx.check(lib.plonk_pk_create(ctx.h, log_n, *[_ptr(a) for a in arrs], C.byref(k1s), C.byref(k2s),
                                      _ptr(pos) if len(pos) else None, len(pos), C.byref(h)))
        self.h, self.ctx, self.n = h, ctx, n
        self.k1, self.k2 = k1, k2
        self.omega = root_of_unity(n)

    def commitments(self, srs):
        """The eight verification-key commitments (preprocess.rs:130-138)."""
        out = (_lib.G1 * 8)()
        self.ctx.check(lib.plonk_pk_commitments(self.ctx.h, self.h, srs.h, out))
        return [_g1_value(g) for g in out]

    def poly(self, which):
        out = np.empty((self.n, 4), dtype=np.uint64)
        ln = C.c_size_t()
        self.ctx.check(lib.plonk_pk_poly(self.ctx.h, self.h, which, _ptr(out), C.byref(ln)))
        return fr_from_array(out[:ln.value])

    def close(self):
        if getattr(self, "h", None):
            lib.plonk_pk_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Prover:
    """plonk/prover.rs:24-62 -- Prover::new(cs, pk, permutation, domain, srs) / prove(public_inputs).
    `wires` = ConstraintSystem::get_wire_values(): three length-n vectors.  The reference draws its
    eleven blinding scalars from thread_rng (prover.rs:189-194, 310-312, 537-538); here the caller
    passes them (`blinding`), or they come from os.urandom."""

    def __init__(self, wires, pk, srs, resident=False, comm=None, shard=None, groups=None):
        """comm/shard: multi-GPU proving (plonk_amd.sharded): `srs` is then this rank's slice
        [shard[0], shard[1]) of g1_powers and `comm` a sharded.Communicator.
        groups: optional {2: (srs_slice, (lo, hi)), 3: (srs_slice, (lo, hi))} -- the rank's slices for the rounds
        that commit 2 / 3 polynomials at once, whose commitments are then placed on groups of ranks
        (sharded.group_range gives the ranges; plonk_prove_3wire_grouped)."""
        if comm is not None:
            if shard is None or len(shard) != 2 or shard[1] - shard[0] != len(srs):
                raise PlonkError(_lib.PLONK_E_INVALID, "sharded proving needs shard=(lo, hi) with hi - lo = len(srs)")
            for b, (g_srs, g_range) in (groups or {}).items():
                if b not in (2, 3) or len(g_range) != 2 or g_range[1] - g_range[0] != len(g_srs):
                    raise PlonkError(_lib.PLONK_E_INVALID, "groups needs {2 or 3: (srs slice, (lo, hi))} with hi - lo = len(slice)")
        elif groups:
            raise PlonkError(_lib.PLONK_E_INVALID, "groups without a communicator")
        self.pk, self.srs = pk, srs
        self.wires = [_as_limbs(w, pk.n) for w in wires]
        self.dev = None
        self.comm, self.shard, self.groups = comm, shard, dict(groups or {})
        if resident or comm is not None:  # keep the witness in HBM: prove() then starts from device buffers
            ctx = pk.ctx
            self.dev = [ctx.alloc(w.nbytes) for w in self.wires]
            for d, w in zip(self.dev, self.wires):
                ctx.upload(d, w)

    def close(self):
        if self.dev:
            for d in self.dev:
                self.pk.ctx.free(d)
            self.dev = None

    def prove(self, public_inputs=(), blinding=None):
        import os
        if blinding is None:
            blinding = [int.from_bytes(os.urandom(40), "little") % R_MOD for _ in range(11)]
        if len(blinding) != 11:
            raise PlonkError(_lib.PLONK_E_INVALID, "blinding must hold 11 scalars")
        bl = (_lib.Fr * 11)(*[_fr_struct(b) for b in blinding])
        pis = (_lib.Fr * max(len(public_inputs), 1))(*[_fr_struct(p) for p in public_inputs])
        raw = _lib.Proof()
        ctx = self.pk.ctx
        if self.comm is not None and self.groups:
            sl = _lib.ShardSlices()
            sl.srs[0], sl.lo[0] = self.srs.h.value, self.shard[0]
            for b, (g_srs, g_range) in self.groups.items():
                sl.srs[b - 1], sl.lo[b - 1] = g_srs.h.value, g_range[0]
            ctx.check(lib.plonk_prove_3wire_grouped(ctx.h, self.pk.h, C.byref(sl), self.comm.h,
                                                    self.dev[0], self.dev[1], self.dev[2],
                                                    pis, len(public_inputs), bl, C.byref(raw)))
        elif self.comm is not None:
            ctx.check(lib.plonk_prove_3wire_sharded(ctx.h, self.pk.h, self.srs.h, self.shard[0], self.comm.h,
                                                    self.dev[0], self.dev[1], self.dev[2],
                                                    pis, len(public_inputs), bl, C.byref(raw)))
        elif self.dev:
            ctx.check(lib.plonk_prove_3wire_dev(ctx.h, self.pk.h, self.srs.h, self.dev[0], self.dev[1], self.dev[2],
                                                pis, len(public_inputs), bl, C.byref(raw)))
        else:
            ctx.check(lib.plonk_prove_3wire(ctx.h, self.pk.h, self.srs.h, _ptr(self.wires[0]), _ptr(self.wires[1]),
                                            _ptr(self.wires[2]), pis, len(public_inputs), bl, C.byref(raw)))
        return Proof(raw)


def _g2_struct(pt):
    g = _lib.G2()
    if pt is None:
        g.infinity = 1
        return g
    (x0, x1), (y0, y1) = pt
    for name, v in (("x_c0", x0), ("x_c1", x1), ("y_c0", y0), ("y_c1", y1)):
        m = v * _RQ % Q_MOD
        arr = getattr(g, name)
        for i in range(6):
            arr[i] = (m >> (64 * i)) & 0xFFFFFFFFFFFFFFFF
    return g


def _g2_value(g):
    if g.infinity:
        return None
    val = lambda arr: sum(int(arr[i]) << (64 * i) for i in range(6)) * _RQ_INV % Q_MOD
    return ((val(g.x_c0), val(g.x_c1)), (val(g.y_c0), val(g.y_c1)))


def g2_generator():
    """curve.rs:250-254."""
    g = _lib.G2()
    lib.plonk_g2_generator(C.byref(g))
    return g


def g2_mul(point, k):
    """k * point in G2 (kzg.rs:86: g2_tau = g2_generator * tau); `point` a _lib.G2."""
    out = _lib.G2()
    ks = _fr_struct(k)
    rc = lib.plonk_g2_mul(C.byref(point), C.byref(ks), C.byref(out))
    if rc != 0:
        raise PlonkError(rc, "plonk_g2_mul")
    return out


def pairing_check(p1, q1, p2, q2):
    """curve.rs:408-412; p1, p2 affine (x, y) or None, q1, q2 _lib.G2."""
    eq = C.c_int32()
    a, b = _g1_struct(p1), _g1_struct(p2)
    rc = lib.plonk_pairing_check(C.byref(a), C.byref(q1), C.byref(b), C.byref(q2), C.byref(eq))
    if rc != 0:
        raise PlonkError(rc, "plonk_pairing_check: invalid point")
    return bool(eq.value)


class VerificationKey:
    """plonk/types.rs:42-69."""

    def __init__(self, raw, public_input_positions=()):
        self.raw = raw
        self.public_input_positions = list(public_input_positions)

    @classmethod
    def from_proving_key(cls, pk, srs, public_input_positions=()):
        """preprocess.rs:130-178: the eight commitments are computed on the device."""
        raw = _lib.Vk()
        pk.ctx.check(lib.plonk_vk_from_pk(pk.ctx.h, pk.h, srs.h, C.byref(raw)))
        return cls(raw, public_input_positions)

    @classmethod
    def from_values(cls, n, commitments, omega, k1, k2, public_input_positions=()):
        """commitments: q_m, q_l, q_r, q_o, q_c, s_sigma1, s_sigma2, s_sigma3 as affine (x, y) or None."""
        raw = _lib.Vk()
        raw.n = n
        for name, pt in zip(("q_m_comm", "q_l_comm", "q_r_comm", "q_o_comm", "q_c_comm", "s_sigma1_comm", "s_sigma2_comm",
                             "s_sigma3_comm"), commitments):
            setattr(raw, name, _g1_struct(pt))
        raw.omega, raw.k1, raw.k2 = _fr_struct(omega), _fr_struct(k1), _fr_struct(k2)
        return cls(raw, public_input_positions)


class Verifier:
    """plonk/verifier.rs:24-40 -- Verifier::new(vk, srs) / verify(proof, public_inputs); of the SRS the
    verifier needs g2_tau only (kzg.rs:34), a _lib.G2."""

    def __init__(self, vk, g2_tau):
        self.vk, self.g2_tau = vk, g2_tau

    def verify(self, proof, public_inputs=()):
        pos = (C.c_uint64 * max(len(self.vk.public_input_positions), 1))(*self.vk.public_input_positions)
        pis = (_lib.Fr * max(len(public_inputs), 1))(*[_fr_struct(p) for p in public_inputs])
        ok = C.c_int32()
        rc = lib.plonk_verify_3wire(C.byref(self.vk.raw), pos, len(self.vk.public_input_positions), C.byref(self.g2_tau),
                                    C.byref(proof.raw), pis, len(public_inputs), C.byref(ok))
        if rc != 0:
            raise PlonkError(rc, "plonk_verify_3wire: malformed verification key, proof or public input")
        return bool(ok.value)
