"""Polynomials, transcript, circuits, permutation, KZG, preprocess, prover and verifier
(oracle; test infrastructure).  Restates backend_new/src/{polynomial,transcript,kzg}.rs,
circuit/{constraint,range_proof}.rs and plonk/{permutation,preprocess,prover,verifier}.rs.
Field elements are ints mod r; polynomials are coefficient lists, low degree first.
"""
import hashlib

from . import curve
from .curve import JAC_ID, jac_add, scalar_mul, to_affine, to_jac, jac_neg
from .fft import Domain, strip, ifft
from .field import R_MOD, fr_inv, fr_to_bytes, fr_from_bytes

WIRE_A, WIRE_B, WIRE_C = 0, 1, 2


# ----------------------------------------------------------------------------- polynomial.rs
def poly_eval(c, x):
    """polynomial.rs:103-113 — Horner."""
    r = 0
    for coeff in reversed(c):
        r = (r * x + coeff) % R_MOD
    return r


def poly_add(a, b):
    """polynomial.rs:352-368 — result normalised."""
    m = max(len(a), len(b))
    out = [0] * m
    for i, c in enumerate(a):
        out[i] = c
    for i, c in enumerate(b):
        out[i] = (out[i] + c) % R_MOD
    return strip(out)


def poly_sub(a, b):
    """polynomial.rs:370-386 — result normalised."""
    m = max(len(a), len(b))
    out = [0] * m
    for i, c in enumerate(a):
        out[i] = c
    for i, c in enumerate(b):
        out[i] = (out[i] - c) % R_MOD
    return strip(out)


def poly_scale(a, s):
    """polynomial.rs:217-221 — not normalised."""
    return [c * s % R_MOD for c in a]


def poly_mul(a, b):
    """polynomial.rs:136-151 — schoolbook."""
    if not strip(a) or not strip(b):
        return []
    out = [0] * (len(a) + len(b) - 1)
    for i, x in enumerate(a):
        for j, y in enumerate(b):
            out[i + j] = (out[i + j] + x * y) % R_MOD
    return strip(out)


def poly_vanishing(n):
    """polynomial.rs:244-249 — X^n - 1."""
    return [R_MOD - 1] + [0] * (n - 1) + [1]


def poly_div_by_linear(c, a):
    """polynomial.rs:188-204 — synthetic division by (X - a)."""
    if not strip(c):
        return []
    n = len(c)
    q = [0] * (n - 1)
    carry = 0
    for i in range(n - 2, -1, -1):
        q[i] = (c[i + 1] + carry) % R_MOD
        carry = q[i] * a % R_MOD
    return strip(q)


def poly_shift(c, omega):
    """polynomial.rs:268-276 — f(X) -> f(omega X)."""
    out, w = [], 1
    for x in c:
        out.append(x * w % R_MOD)
        w = w * omega % R_MOD
    return out


def from_evaluations(evals, domain):
    """polynomial.rs:56-65."""
    if len(evals) == 0:
        return []
    return ifft(evals, domain.omega)


def evaluate_coset(c, domain, k):
    """polynomial.rs:279-281."""
    return domain.coset_fft(c, k)


# ----------------------------------------------------------------------------- transcript.rs
class Transcript:
    """transcript.rs:39-131 — streaming SHA-256 chained across squeezes."""

    def __init__(self, domain_separator):
        self.h = hashlib.sha256()
        self.h.update(domain_separator)  # :41-43

    def absorb_bytes(self, label, data):  # :57-61
        self.h.update(label.encode())
        self.h.update(len(data).to_bytes(8, "little"))
        self.h.update(data)

    def absorb_fr(self, label, x):  # :71-74
        self.absorb_bytes(label, fr_to_bytes(x))

    def absorb_g1(self, label, pt):  # :77-80 — no length prefix
        self.h.update(label.encode())
        self.h.update(curve.compress(pt))

    def squeeze_challenge(self, label):  # :110-131
        self.h.update(label.encode())
        digest = self.h.digest()
        self.h = hashlib.sha256()
        self.h.update(digest)
        return hash_to_fr(digest)


def hash_to_fr(digest):
    """transcript.rs:157-173 — re-hash until the LE integer is < r."""
    b = bytes(digest[:32])
    while True:
        v = fr_from_bytes(b)
        if v is not None:
            return v
        b = hashlib.sha256(b).digest()


# ----------------------------------------------------------------------------- circuit/constraint.rs
class Gate:
    """constraint.rs:72-99."""

    def __init__(self, q_m=0, q_l=0, q_r=0, q_o=0, q_c=0, a=None, b=None, c=None):
        self.q_m, self.q_l, self.q_r, self.q_o, self.q_c = q_m, q_l, q_r, q_o, q_c
        self.a, self.b, self.c = a, b, c

    @staticmethod
    def mul():  # :129-139
        return Gate(q_m=1, q_o=R_MOD - 1)

    @staticmethod
    def add():  # :143-153
        return Gate(q_l=1, q_r=1, q_o=R_MOD - 1)

    @staticmethod
    def add_scaled(k):  # :157-167
        return Gate(q_l=1, q_r=k % R_MOD, q_o=R_MOD - 1)

    @staticmethod
    def boolean():  # :172-182
        return Gate(q_m=1, q_l=R_MOD - 1)

    @staticmethod
    def dummy():  # :213-226
        return Gate(a=0, b=0, c=0)

    def is_satisfied(self):  # :236-248
        return (self.q_m * self.a * self.b + self.q_l * self.a + self.q_r * self.b
                + self.q_o * self.c + self.q_c) % R_MOD == 0


class ConstraintSystem:
    """constraint.rs:311-411."""

    def __init__(self):
        self.gates = []
        self.copy_constraints = []  # ((wire_type, gate), (wire_type, gate))
        self.public_input_positions = []
        self.num_gates = 0
        self.n = 0

    def add_gate(self, g):
        self.gates.append(g)
        self.num_gates = len(self.gates)

    def add_copy_constraint(self, w1, w2):
        self.copy_constraints.append((w1, w2))

    def finalize(self):  # :365-382
        n = 1
        while n < self.num_gates:
            n *= 2
        if n < 4:
            n = 4
        while len(self.gates) < n:
            self.gates.append(Gate.dummy())
        self.n = n

    def selectors(self):  # :385-393
        g = self.gates
        return ([x.q_m for x in g], [x.q_l for x in g], [x.q_r for x in g],
                [x.q_o for x in g], [x.q_c for x in g])

    def wire_values(self):  # :401-410
        g = self.gates
        return [x.a for x in g], [x.b for x in g], [x.c for x in g]

    def is_satisfied(self):
        return all(g.is_satisfied() for g in self.gates)


def range_proof_circuit(n_bits, x):
    """range_proof.rs:52-199 — RangeProofCircuit::new(n_bits, x).constraint_system."""
    v = x & ((1 << n_bits) - 1)
    bits = [(v >> i) & 1 for i in range(n_bits)]
    cs = ConstraintSystem()
    for i in range(n_bits):  # :99-109
        g = Gate.boolean()
        g.a, g.b, g.c = bits[i], bits[i], 0
        cs.add_gate(g)
    for i in range(n_bits):  # :112-118
        cs.add_copy_constraint((WIRE_A, i), (WIRE_B, i))
    acc, p2 = 0, 1
    for i in range(n_bits):  # :136-186
        gi = n_bits + i
        nxt = (acc + p2 * bits[i]) % R_MOD
        g = Gate.add_scaled(p2)
        g.a, g.b, g.c = acc, bits[i], nxt
        cs.add_gate(g)
        cs.add_copy_constraint((WIRE_B, gi), (WIRE_A, i))
        if i > 0:
            cs.add_copy_constraint((WIRE_C, gi - 1), (WIRE_A, gi))
        acc = nxt
        p2 = p2 * 2 % R_MOD
    cs.finalize()
    return cs


def square_add_circuit(w=3, x=3, min_gates=16):
    """The w^2 + x = y demo statement (backend/src/circuit.rs:78-113) as a 3-wire
    constraint system: gate 0 mul (w*w = t), gate 1 add (t + x = y), copies
    a[0]=b[0] and c[0]=a[1], padded with dummy gates to n = 16, the size the demo runs at
    (backend/src/srs.rs:17).  dusk-plonk's own 4-wire gates are not in the tree.  (At n = 4 the
    reference's 4n-point quotient domain cannot hold t of degree 3n+5, so its own proofs would
    not verify there.)"""
    cs = ConstraintSystem()
    t = w * w % R_MOD
    g = Gate.mul(); g.a, g.b, g.c = w, w, t; cs.add_gate(g)
    g = Gate.add(); g.a, g.b, g.c = t, x, (t + x) % R_MOD; cs.add_gate(g)
    cs.add_copy_constraint((WIRE_A, 0), (WIRE_B, 0))
    cs.add_copy_constraint((WIRE_C, 0), (WIRE_A, 1))
    while len(cs.gates) < min_gates:
        cs.add_gate(Gate.dummy())
    cs.finalize()
    return cs


# splitmix64 stream used for the synthetic 2^k-gate circuits and SRS trapdoor.
class SplitMix64:
    def __init__(self, seed):
        self.s = seed & 0xFFFFFFFFFFFFFFFF

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)

    def fr_rand(self):
        """Four outputs as little-endian limbs, top limb masked to 62 bits (< 2^254 < r)."""
        l = [self.next() for _ in range(4)]
        l[3] &= (1 << 62) - 1
        return l[0] | (l[1] << 64) | (l[2] << 128) | (l[3] << 192)


SYNTH_SEED = 0x504C4F4E4B000000
SRS_SEED = 0x5352530000000000


def synthetic_circuit(k):
    """2^k-gate alternating mul/add chain: gate i even a*b=c, odd a+b=c, a[i+1] = c[i]
    (copy constraint), a[0] and every b[i] from splitmix64(SYNTH_SEED + k).  Returns
    (constraint system, blinding b1..b11 drawn next from the same stream)."""
    rng = SplitMix64(SYNTH_SEED + k)
    n = 1 << k
    cs = ConstraintSystem()
    a = rng.fr_rand()
    for i in range(n):
        b = rng.fr_rand()
        if i % 2 == 0:
            g = Gate.mul(); c = a * b % R_MOD
        else:
            g = Gate.add(); c = (a + b) % R_MOD
        g.a, g.b, g.c = a, b, c
        cs.add_gate(g)
        if i + 1 < n:
            cs.add_copy_constraint((WIRE_C, i), (WIRE_A, i + 1))
        a = c
    cs.finalize()
    blinding = [rng.fr_rand() for _ in range(11)]
    return cs, blinding


def synthetic_tau(k):
    return SplitMix64(SRS_SEED + k).fr_rand()


# ----------------------------------------------------------------------------- plonk/permutation.rs
def find_coset_generator(n, index):
    """permutation.rs:307-326."""
    k = (5 + index) % R_MOD
    while pow(k, n, R_MOD) == 1:
        k = (k + 1) % R_MOD
    return k


class Permutation:
    """permutation.rs:55-140."""

    def __init__(self, cs, domain):
        n = domain.n
        self.n = n
        self.k1 = find_coset_generator(n, 1)
        self.k2 = find_coset_generator(n, 2)
        parent = list(range(3 * n))
        rank = [0] * (3 * n)

        def find(x):
            root = x
            while parent[root] != root:
                root = parent[root]
            while parent[x] != root:
                parent[x], x = root, parent[x]
            return root

        def union(x, y):  # :349-365
            px, py = find(x), find(y)
            if px == py:
                return
            if rank[px] < rank[py]:
                parent[px] = py
            elif rank[px] > rank[py]:
                parent[py] = px
            else:
                parent[py] = px
                rank[px] += 1

        def widx(w):  # :266-272
            return w[0] * n + w[1]

        for w1, w2 in cs.copy_constraints:  # :74-78
            union(widx(w1), widx(w2))
        classes = {}
        for i in range(3 * n):  # :81-85
            classes.setdefault(find(i), []).append(i)
        sigma = list(range(3 * n))
        for members in classes.values():  # :88-105
            if len(members) > 1:
                for i, cur in enumerate(members):
                    sigma[cur] = members[(i + 1) % len(members)]
        self.sigma = sigma
        powers = domain.elements()
        ks = (1, self.k1, self.k2)

        def encode(idx):  # :289-296
            return ks[idx // n] * powers[idx % n] % R_MOD

        self.s_sigma1_evals = [encode(sigma[i]) for i in range(n)]
        self.s_sigma2_evals = [encode(sigma[n + i]) for i in range(n)]
        self.s_sigma3_evals = [encode(sigma[2 * n + i]) for i in range(n)]


# ----------------------------------------------------------------------------- kzg.rs
class Srs:
    """kzg.rs:26-35, from_tau :71-94.  g1_powers[i] = [tau^i]_1, affine."""

    def __init__(self, tau, max_degree):
        self.max_degree = max_degree
        self.tau = tau % R_MOD  # kept: the oracle's trapdoor checks need it
        self._g2_tau = None
        self.g1_powers = []
        tp = 1
        for _ in range(max_degree + 1):
            self.g1_powers.append(to_affine(scalar_mul(curve.G1_GEN, tp)))
            tp = tp * tau % R_MOD


    @property
    def g2_generator(self):
        """kzg.rs:32."""
        from . import pairing
        return pairing.G2_GEN

    @property
    def g2_tau(self):
        """kzg.rs:34,86: [tau]_2, what the verifier's pairing check is taken against."""
        from . import pairing
        if self._g2_tau is None:
            self._g2_tau = pairing.g2_mul(pairing.G2_GEN, self.tau)
        return self._g2_tau


def commit(coeffs, srs, use_trapdoor=False):
    """kzg.rs:231-245 — MSM over srs.g1_powers[..len], then to_affine().
    `use_trapdoor` evaluates the same group element as f(tau)*G (cross-check path)."""
    if len(coeffs) > len(srs.g1_powers):
        raise ValueError("Polynomial degree %d exceeds SRS max degree %d"
                         % (len(coeffs) - 1, srs.max_degree))
    if use_trapdoor:
        return to_affine(scalar_mul(curve.G1_GEN, poly_eval(coeffs, srs.tau)))
    return to_affine(curve.msm(srs.g1_powers[:len(coeffs)], coeffs))


# ----------------------------------------------------------------------------- plonk/preprocess.rs
class ProvingKey:
    pass


def preprocess(cs, srs, use_trapdoor=False, engine=None):
    """preprocess.rs:104-186.  Returns (pk, vk dict, permutation, domain).
    `engine` (oracle.cref.Engine) runs the transforms in the C restatement of fft.rs instead of
    the Python one; the values are the same."""
    n = cs.n
    domain = Domain(n)
    from_evaluations = engine.from_evaluations if engine is not None else globals()["from_evaluations"]
    q_m, q_l, q_r, q_o, q_c = [from_evaluations(s, domain) for s in cs.selectors()]
    perm = Permutation(cs, domain)
    s1 = from_evaluations(perm.s_sigma1_evals, domain)
    s2 = from_evaluations(perm.s_sigma2_evals, domain)
    s3 = from_evaluations(perm.s_sigma3_evals, domain)
    l1 = from_evaluations([1] + [0] * (n - 1), domain)  # :142
    pk = ProvingKey()
    pk.n, pk.omega, pk.k1, pk.k2 = n, domain.omega, perm.k1, perm.k2
    pk.q_m, pk.q_l, pk.q_r, pk.q_o, pk.q_c = q_m, q_l, q_r, q_o, q_c
    pk.s_sigma1, pk.s_sigma2, pk.s_sigma3, pk.l1 = s1, s2, s3, l1
    c = lambda p: commit(p, srs, use_trapdoor)
    vk = dict(n=n, omega=domain.omega, k1=perm.k1, k2=perm.k2,
              public_input_positions=list(cs.public_input_positions),
              q_m=c(q_m), q_l=c(q_l), q_r=c(q_r), q_o=c(q_o), q_c=c(q_c),
              s_sigma1=c(s1), s_sigma2=c(s2), s_sigma3=c(s3))
    return pk, vk, perm, domain


# ----------------------------------------------------------------------------- plonk/prover.rs
PROOF_G1 = ("a_comm", "b_comm", "c_comm", "z_comm", "t_lo_comm", "t_mid_comm", "t_hi_comm",
            "w_zeta_comm", "w_zeta_omega_comm")
PROOF_FR = ("a_eval", "b_eval", "c_eval", "s_sigma1_eval", "s_sigma2_eval", "z_omega_eval")


def public_input_poly(cs, domain, public_inputs, from_evaluations=from_evaluations):
    """prover.rs:840-857."""
    ev = [0] * cs.n
    for i, pos in enumerate(cs.public_input_positions):
        if i < len(public_inputs):
            ev[pos] = (-public_inputs[i]) % R_MOD
    return from_evaluations(ev, domain)


def quotient_evals(n, pk, a, b, c, z, pi, alpha, beta, gamma):
    """prover.rs:364-457 — the round-3 body: 15 coset NTTs over 4n points (shift 7) and the
    fused per-point loop.  Returns t on the coset, natural order."""
    big = Domain(4 * n)
    g = 7
    ec = lambda p: evaluate_coset(p, big, g)
    A, B, C, Z = ec(a), ec(b), ec(c), ec(z)
    ZW = ec(poly_shift(z, pk.omega))  # :376-377
    QM, QL, QR, QO, QC = ec(pk.q_m), ec(pk.q_l), ec(pk.q_r), ec(pk.q_o), ec(pk.q_c)
    S1, S2, S3 = ec(pk.s_sigma1), ec(pk.s_sigma2), ec(pk.s_sigma3)
    L1 = ec(pk.l1)
    PI = ec(pi)
    k1, k2 = pk.k1, pk.k2
    t = [0] * (4 * n)
    x = g
    for i in range(4 * n):  # :415-457
        zh = (pow(x, n, R_MOD) - 1) % R_MOD  # :400-405
        gate = (QM[i] * A[i] * B[i] + QL[i] * A[i] + QR[i] * B[i] + QO[i] * C[i] + QC[i] + PI[i]) % R_MOD
        num = Z[i] * (A[i] + beta * x + gamma) % R_MOD * (B[i] + beta * k1 * x + gamma) % R_MOD \
            * (C[i] + beta * k2 * x + gamma) % R_MOD
        den = ZW[i] * (A[i] + beta * S1[i] + gamma) % R_MOD * (B[i] + beta * S2[i] + gamma) % R_MOD \
            * (C[i] + beta * S3[i] + gamma) % R_MOD
        perm = (num - den) % R_MOD
        l1c = L1[i] * (Z[i] - 1) % R_MOD
        numerator = (gate + alpha * perm + alpha * alpha * l1c) % R_MOD
        t[i] = numerator * fr_inv(zh) % R_MOD
        x = x * big.omega % R_MOD
    return t


def prove(cs, pk, perm, domain, srs, public_inputs, blinding, use_trapdoor=False, trace=None, engine=None):
    """prover.rs:62-149 with the RNG draws (b1..b6 :189-194, b7..b9 :310-312, b10,b11 :537-538)
    injected as `blinding[0..11]`.  Returns a dict of 9 affine G1 points and 6 Fr values.
    `engine` (oracle.cref.Engine) runs the transforms (fft.rs) and the round-3 loop
    (prover.rs:400-457) in their C restatement, for sizes the Python loops cannot reach; every
    other step stays as written here."""
    n = pk.n
    from_evaluations = engine.from_evaluations if engine is not None else globals()["from_evaluations"]
    omega, k1, k2 = pk.omega, pk.k1, pk.k2
    b1, b2, b3, b4, b5, b6, b7, b8, b9, b10, b11 = [x % R_MOD for x in blinding]
    cm = lambda p: commit(p, srs, use_trapdoor)
    tr = Transcript(b"PLONK-v1")
    for i, pi in enumerate(public_inputs):  # :69-71
        tr.absorb_fr("pi_%d" % i, pi)

    # Round 1 — :181-244
    a_ev, b_ev, c_ev = cs.wire_values()
    zh = poly_vanishing(n)
    a = poly_add(from_evaluations(a_ev, domain), poly_mul(strip([b2, b1]), zh))
    b = poly_add(from_evaluations(b_ev, domain), poly_mul(strip([b4, b3]), zh))
    c = poly_add(from_evaluations(c_ev, domain), poly_mul(strip([b6, b5]), zh))
    a_comm, b_comm, c_comm = cm(a), cm(b), cm(c)
    tr.absorb_g1("a_comm", a_comm); tr.absorb_g1("b_comm", b_comm); tr.absorb_g1("c_comm", c_comm)
    beta = tr.squeeze_challenge("beta")
    gamma = tr.squeeze_challenge("gamma")

    # Round 2 — :247-336
    z_ev = [0] * n
    z_ev[0] = 1
    w = 1
    for i in range(n - 1):
        num = (a_ev[i] + beta * w + gamma) * (b_ev[i] + beta * k1 * w + gamma) % R_MOD \
            * (c_ev[i] + beta * k2 * w + gamma) % R_MOD
        den = (a_ev[i] + beta * perm.s_sigma1_evals[i] + gamma) * (b_ev[i] + beta * perm.s_sigma2_evals[i] + gamma) \
            % R_MOD * (c_ev[i] + beta * perm.s_sigma3_evals[i] + gamma) % R_MOD
        z_ev[i + 1] = z_ev[i] * num % R_MOD * fr_inv(den) % R_MOD
        w = w * omega % R_MOD
    z = poly_add(from_evaluations(z_ev, domain), poly_mul(strip([b9, b8, b7]), zh))
    z_comm = cm(z)
    tr.absorb_g1("z_comm", z_comm)
    alpha = tr.squeeze_challenge("alpha")

    # Round 3 — :339-574
    pi_poly = public_input_poly(cs, domain, public_inputs, from_evaluations)
    if engine is None:
        t_coset = quotient_evals(n, pk, a, b, c, z, pi_poly, alpha, beta, gamma)
        t = strip(Domain(4 * n).coset_ifft(t_coset, 7))  # :497-498
    else:
        t_coset = None
        t = engine.quotient_poly(n, pk, a, b, c, z, pi_poly, alpha, beta, gamma)
    assert len(t) <= 3 * n + 6, "t_poly degree exceeds paper bound"  # :506-511
    t_lo = [0] * (n + 1); t_mid = [0] * (n + 1); t_hi = [0] * (n + 6)
    for i, coeff in enumerate(t):  # :521-529
        if i < n:
            t_lo[i] = coeff
        elif i < 2 * n:
            t_mid[i - n] = coeff
        else:
            t_hi[i - 2 * n] = coeff
    t_lo[n] = (t_lo[n] + b10) % R_MOD  # :540-543
    t_mid[0] = (t_mid[0] - b10) % R_MOD
    t_mid[n] = (t_mid[n] + b11) % R_MOD
    t_hi[0] = (t_hi[0] - b11) % R_MOD
    t_lo, t_mid, t_hi = strip(t_lo), strip(t_mid), strip(t_hi)
    t_lo_comm, t_mid_comm, t_hi_comm = cm(t_lo), cm(t_mid), cm(t_hi)
    tr.absorb_g1("t_lo_comm", t_lo_comm); tr.absorb_g1("t_mid_comm", t_mid_comm)
    tr.absorb_g1("t_hi_comm", t_hi_comm)
    zeta = tr.squeeze_challenge("zeta")

    # Round 4 — :577-641
    zeta_omega = zeta * omega % R_MOD
    a_bar, b_bar, c_bar = poly_eval(a, zeta), poly_eval(b, zeta), poly_eval(c, zeta)
    s1_bar, s2_bar = poly_eval(pk.s_sigma1, zeta), poly_eval(pk.s_sigma2, zeta)
    zw_bar = poly_eval(z, zeta_omega)
    for label, val in (("a_zeta", a_bar), ("b_zeta", b_bar), ("c_zeta", c_bar),
                       ("s_sigma1_zeta", s1_bar), ("s_sigma2_zeta", s2_bar), ("z_zeta_omega", zw_bar)):
        tr.absorb_fr(label, val)
    v = tr.squeeze_challenge("v")

    # Round 5 — :644-837
    zh_zeta = (pow(zeta, n, R_MOD) - 1) % R_MOD  # :697
    l1_zeta = zh_zeta * fr_inv(n % R_MOD * (zeta - 1) % R_MOD) % R_MOD  # :700
    r = poly_scale(pk.q_m, a_bar * b_bar % R_MOD)  # :742-746
    r = poly_add(r, poly_scale(pk.q_l, a_bar))
    r = poly_add(r, poly_scale(pk.q_r, b_bar))
    r = poly_add(r, poly_scale(pk.q_o, c_bar))
    r = poly_add(r, pk.q_c)
    pc1 = alpha * (a_bar + beta * zeta + gamma) % R_MOD * (b_bar + beta * k1 * zeta + gamma) % R_MOD \
        * (c_bar + beta * k2 * zeta + gamma) % R_MOD  # :749-752
    pc2 = (-(alpha * (a_bar + beta * s1_bar + gamma) % R_MOD * (b_bar + beta * s2_bar + gamma) % R_MOD
             * beta % R_MOD * zw_bar)) % R_MOD  # :755-760
    r = poly_add(r, poly_scale(z, pc1))
    r = poly_add(r, poly_scale(pk.s_sigma3, pc2))
    r = poly_add(r, poly_scale(z, alpha * alpha % R_MOD * l1_zeta % R_MOD))  # :766
    zeta_n = pow(zeta, n, R_MOD)
    zeta_2n = zeta_n * zeta_n % R_MOD
    t_comb = poly_add(poly_add(t_lo, poly_scale(t_mid, zeta_n)), poly_scale(t_hi, zeta_2n))  # :776-778
    r = poly_sub(r, poly_scale(t_comb, zh_zeta))  # :784
    r_zeta = poly_eval(r, zeta)
    v2 = v * v % R_MOD; v3 = v2 * v % R_MOD; v4 = v3 * v % R_MOD; v5 = v4 * v % R_MOD
    wz = poly_sub(r, strip([r_zeta]))  # :800-807
    wz = poly_add(wz, poly_scale(poly_sub(a, strip([a_bar])), v))
    wz = poly_add(wz, poly_scale(poly_sub(b, strip([b_bar])), v2))
    wz = poly_add(wz, poly_scale(poly_sub(c, strip([c_bar])), v3))
    wz = poly_add(wz, poly_scale(poly_sub(pk.s_sigma1, strip([s1_bar])), v4))
    wz = poly_add(wz, poly_scale(poly_sub(pk.s_sigma2, strip([s2_bar])), v5))
    w_zeta = poly_div_by_linear(wz, zeta)  # :809
    w_zeta_omega = poly_div_by_linear(poly_sub(z, strip([zw_bar])), zeta_omega)  # :815-816
    w_zeta_comm, w_zeta_omega_comm = cm(w_zeta), cm(w_zeta_omega)
    tr.absorb_g1("w_zeta_comm", w_zeta_comm); tr.absorb_g1("w_zeta_omega_comm", w_zeta_omega_comm)
    u = tr.squeeze_challenge("u")

    if trace is not None:
        trace.update(a=a, b=b, c=c, z=z, z_evals=z_ev, t=t, t_coset=t_coset, t_lo=t_lo, t_mid=t_mid,
                     t_hi=t_hi, r=r, w_zeta=w_zeta, w_zeta_omega=w_zeta_omega, beta=beta, gamma=gamma,
                     alpha=alpha, zeta=zeta, v=v, u=u)
    return dict(a_comm=a_comm, b_comm=b_comm, c_comm=c_comm, z_comm=z_comm, t_lo_comm=t_lo_comm,
                t_mid_comm=t_mid_comm, t_hi_comm=t_hi_comm, a_eval=a_bar, b_eval=b_bar, c_eval=c_bar,
                s_sigma1_eval=s1_bar, s_sigma2_eval=s2_bar, z_omega_eval=zw_bar,
                w_zeta_comm=w_zeta_comm, w_zeta_omega_comm=w_zeta_omega_comm)


def proof_to_bytes(proof):
    """Wire image of types.rs:74-114 `Proof`, in CompactProof order (trace.rs:255-332):
    a,b,c,z,t_lo,t_mid,t_hi (48 B compressed each), the six evaluations (32 B canonical LE
    each), w_zeta, w_zeta_omega (48 B each) = 624 bytes."""
    out = b""
    for k in PROOF_G1[:7]:
        out += curve.compress(proof[k])
    for k in PROOF_FR:
        out += fr_to_bytes(proof[k])
    for k in PROOF_G1[7:]:
        out += curve.compress(proof[k])
    return out


# ----------------------------------------------------------------------------- plonk/verifier.rs
def verify(vk, srs, proof, public_inputs):
    """verifier.rs:42-314.  The final check e(W, [tau]_2) == e(Rhs, [1]_2) (:309-314) is
    evaluated with the SRS trapdoor as tau*W == Rhs in G1, which is the same statement for
    [tau]_2 = tau*[1]_2 (kzg.rs:86).  verify_pairing evaluates it as the reference does."""
    w_comb, right = verifier_points(vk, proof, public_inputs)
    return to_affine(scalar_mul(w_comb, srs.tau)) == right


def verify_pairing(vk, g2_tau, proof, public_inputs):
    """verifier.rs:42-314 with the pairing check of :309-314, pairing_check(w_combined, [tau]_2,
    right_inner, [1]_2): needs only the public half of the SRS."""
    from . import pairing
    w_comb, right = verifier_points(vk, proof, public_inputs)
    return pairing.pairing_check(w_comb, g2_tau, right, pairing.G2_GEN)


def verifier_points(vk, proof, public_inputs):
    """verifier.rs:42-300: transcript replay and the two G1 sides of the final check,
    (w_combined, right_inner), affine."""
    n, omega, k1, k2 = vk["n"], vk["omega"], vk["k1"], vk["k2"]
    tr = Transcript(b"PLONK-v1")
    for i, pi in enumerate(public_inputs):
        tr.absorb_fr("pi_%d" % i, pi)
    tr.absorb_g1("a_comm", proof["a_comm"]); tr.absorb_g1("b_comm", proof["b_comm"])
    tr.absorb_g1("c_comm", proof["c_comm"])
    beta = tr.squeeze_challenge("beta"); gamma = tr.squeeze_challenge("gamma")
    tr.absorb_g1("z_comm", proof["z_comm"])
    alpha = tr.squeeze_challenge("alpha")
    tr.absorb_g1("t_lo_comm", proof["t_lo_comm"]); tr.absorb_g1("t_mid_comm", proof["t_mid_comm"])
    tr.absorb_g1("t_hi_comm", proof["t_hi_comm"])
    zeta = tr.squeeze_challenge("zeta")
    a_bar, b_bar, c_bar = proof["a_eval"], proof["b_eval"], proof["c_eval"]
    s1_bar, s2_bar, zw_bar = proof["s_sigma1_eval"], proof["s_sigma2_eval"], proof["z_omega_eval"]
    for label, val in (("a_zeta", a_bar), ("b_zeta", b_bar), ("c_zeta", c_bar),
                       ("s_sigma1_zeta", s1_bar), ("s_sigma2_zeta", s2_bar), ("z_zeta_omega", zw_bar)):
        tr.absorb_fr(label, val)
    v = tr.squeeze_challenge("v")
    tr.absorb_g1("w_zeta_comm", proof["w_zeta_comm"])
    tr.absorb_g1("w_zeta_omega_comm", proof["w_zeta_omega_comm"])
    u = tr.squeeze_challenge("u")

    zh_zeta = (pow(zeta, n, R_MOD) - 1) % R_MOD  # :106
    l1_zeta = 1 if zeta == 1 else zh_zeta * fr_inv(n * (zeta - 1) % R_MOD) % R_MOD  # :112-116
    pi_zeta = 0
    for i, pi in enumerate(public_inputs):  # :124-143
        if i >= len(vk["public_input_positions"]):
            continue
        wp = pow(omega, vk["public_input_positions"][i], R_MOD)
        if zeta != wp:
            lp = wp * zh_zeta % R_MOD * fr_inv(n * (zeta - wp) % R_MOD) % R_MOD
            pi_zeta = (pi_zeta - pi * lp) % R_MOD
    pc1 = (a_bar + beta * zeta + gamma) * (b_bar + beta * k1 * zeta + gamma) % R_MOD \
        * (c_bar + beta * k2 * zeta + gamma) % R_MOD  # :167-169
    pc2 = (-((a_bar + beta * s1_bar + gamma) * (b_bar + beta * s2_bar + gamma) % R_MOD * beta % R_MOD
             * zw_bar)) % R_MOD  # :171-175
    zeta_n = pow(zeta, n, R_MOD); zeta_2n = zeta_n * zeta_n % R_MOD
    sm = lambda pt, k: scalar_mul(pt, k)
    r_comm = JAC_ID  # :182-202
    r_comm = jac_add(r_comm, sm(vk["q_m"], a_bar * b_bar % R_MOD))
    r_comm = jac_add(r_comm, sm(vk["q_l"], a_bar))
    r_comm = jac_add(r_comm, sm(vk["q_r"], b_bar))
    r_comm = jac_add(r_comm, sm(vk["q_o"], c_bar))
    r_comm = jac_add(r_comm, to_jac(vk["q_c"]))
    r_comm = jac_add(r_comm, sm(proof["z_comm"], (alpha * pc1 + alpha * alpha % R_MOD * l1_zeta) % R_MOD))
    r_comm = jac_add(r_comm, sm(vk["s_sigma3"], alpha * pc2 % R_MOD))
    t_coeff = (-zh_zeta) % R_MOD
    r_comm = jac_add(r_comm, sm(proof["t_lo_comm"], t_coeff))
    r_comm = jac_add(r_comm, sm(proof["t_mid_comm"], t_coeff * zeta_n % R_MOD))
    r_comm = jac_add(r_comm, sm(proof["t_hi_comm"], t_coeff * zeta_2n % R_MOD))
    v2 = v * v % R_MOD; v3 = v2 * v % R_MOD; v4 = v3 * v % R_MOD; v5 = v4 * v % R_MOD
    f_comm = r_comm  # :244-250
    f_comm = jac_add(f_comm, sm(proof["a_comm"], v))
    f_comm = jac_add(f_comm, sm(proof["b_comm"], v2))
    f_comm = jac_add(f_comm, sm(proof["c_comm"], v3))
    f_comm = jac_add(f_comm, sm(vk["s_sigma1"], v4))
    f_comm = jac_add(f_comm, sm(vk["s_sigma2"], v5))
    f_comm = jac_add(f_comm, sm(proof["z_comm"], u))
    perm_term = alpha * zw_bar % R_MOD * (a_bar + beta * s1_bar + gamma) % R_MOD \
        * (b_bar + beta * s2_bar + gamma) % R_MOD * (c_bar + gamma) % R_MOD  # :266-270
    r0 = (pi_zeta - alpha * alpha % R_MOD * l1_zeta - perm_term) % R_MOD  # :272
    e_eval = (-r0 + v * a_bar + v2 * b_bar + v3 * c_bar + v4 * s1_bar + v5 * s2_bar + u * zw_bar) % R_MOD
    e_comm = sm(curve.G1_GEN, e_eval)  # :286
    w_comb = jac_add(to_jac(proof["w_zeta_comm"]), sm(proof["w_zeta_omega_comm"], u))  # :290-291
    zeta_omega = zeta * omega % R_MOD
    right = jac_add(f_comm, sm(proof["w_zeta_comm"], zeta))  # :296-299
    right = jac_add(right, sm(proof["w_zeta_omega_comm"], u * zeta_omega % R_MOD))
    right = jac_add(right, jac_neg(e_comm))
    return to_affine(w_comb), to_affine(right)
