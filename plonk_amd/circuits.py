"""Host-side circuit front end: the input formats on the caller's side of the hot path.

Mirrors backend_new/src/circuit/constraint.rs (Gate, ConstraintSystem), circuit/range_proof.rs
(RangeProofCircuit) and plonk/permutation.rs (Permutation::compute), plus the synthetic 2^k-gate
benchmark circuit.  One-time host work (graph/union-find); nothing here is on the timed path.
Field values are Python ints mod r.
"""
R_MOD = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
_TWO_ADIC_ROOT = pow(7, (R_MOD - 1) >> 32, R_MOD)
WIRE_A, WIRE_B, WIRE_C = 0, 1, 2


def root_of_unity(n):
    """field.rs:167-182."""
    return pow(_TWO_ADIC_ROOT, (1 << 32) // n, R_MOD)


class ConstraintSystem:
    """constraint.rs:311-411, column-wise: selector and wire vectors instead of Gate objects."""

    def __init__(self):
        self.q_m, self.q_l, self.q_r, self.q_o, self.q_c = [], [], [], [], []
        self.a, self.b, self.c = [], [], []
        self.copy_constraints = []
        self.public_input_positions = []
        self.num_gates = 0
        self.n = 0

    def add_gate(self, q_m, q_l, q_r, q_o, q_c, a, b, c):
        """constraint.rs:103-125,340-345; returns the gate index."""
        for col, v in zip((self.q_m, self.q_l, self.q_r, self.q_o, self.q_c, self.a, self.b, self.c),
                          (q_m, q_l, q_r, q_o, q_c, a, b, c)):
            col.append(v % R_MOD)
        self.num_gates = len(self.a)
        return self.num_gates - 1

    def mul(self, a, b):  # constraint.rs:129-139
        return self.add_gate(1, 0, 0, -1, 0, a, b, a * b)

    def add(self, a, b):  # :143-153
        return self.add_gate(0, 1, 1, -1, 0, a, b, a + b)

    def add_scaled(self, k, a, b):  # :157-167
        return self.add_gate(0, 1, k, -1, 0, a, b, a + k * b)

    def boolean(self, bit):  # :172-182
        return self.add_gate(1, -1, 0, 0, 0, bit, bit, 0)

    def add_copy_constraint(self, w1, w2):  # :348-350
        self.copy_constraints.append((w1, w2))

    def mark_public_input(self, gate):  # :353-357
        if gate not in self.public_input_positions:
            self.public_input_positions.append(gate)

    def finalize(self):
        """constraint.rs:365-382: pad with all-zero gates to max(4, next power of two)."""
        n = 1
        while n < self.num_gates:
            n *= 2
        n = max(n, 4)
        while len(self.a) < n:
            for col in (self.q_m, self.q_l, self.q_r, self.q_o, self.q_c, self.a, self.b, self.c):
                col.append(0)
        self.n = n

    def selectors(self):  # :385-393
        return self.q_m, self.q_l, self.q_r, self.q_o, self.q_c

    def wire_values(self):  # :401-410
        return self.a, self.b, self.c

    def is_satisfied(self):  # :396-398
        return all((m * a * b + l * a + r * b + o * c + k) % R_MOD == 0 for m, l, r, o, k, a, b, c in
                   zip(self.q_m, self.q_l, self.q_r, self.q_o, self.q_c, self.a, self.b, self.c))


def range_proof_circuit(n_bits, x):
    """range_proof.rs:52-199 -- RangeProofCircuit::new(n_bits, x).constraint_system."""
    v = x & ((1 << n_bits) - 1)
    bits = [(v >> i) & 1 for i in range(n_bits)]
    cs = ConstraintSystem()
    for i in range(n_bits):
        cs.boolean(bits[i])
    for i in range(n_bits):
        cs.add_copy_constraint((WIRE_A, i), (WIRE_B, i))
    acc, p2 = 0, 1
    for i in range(n_bits):
        gi = cs.add_scaled(p2, acc, bits[i])
        cs.add_copy_constraint((WIRE_B, gi), (WIRE_A, i))
        if i > 0:
            cs.add_copy_constraint((WIRE_C, gi - 1), (WIRE_A, gi))
        acc = (acc + p2 * bits[i]) % R_MOD
        p2 = p2 * 2 % R_MOD
    cs.finalize()
    return cs


def square_add_circuit(w=3, x=3, min_gates=16):
    """The w^2 + x = y demo statement of backend/src/circuit.rs:78-113 on three wires, padded to
    the n = 16 the demo runs at (backend/src/srs.rs:17)."""
    cs = ConstraintSystem()
    g0 = cs.mul(w, w)
    g1 = cs.add(cs.c[g0], x)
    cs.add_copy_constraint((WIRE_A, g0), (WIRE_B, g0))
    cs.add_copy_constraint((WIRE_C, g0), (WIRE_A, g1))
    while cs.num_gates < min_gates:
        cs.add_gate(0, 0, 0, 0, 0, 0, 0, 0)
    cs.finalize()
    return cs


def find_coset_generator(n, index):
    """permutation.rs:307-326."""
    k = (5 + index) % R_MOD
    while pow(k, n, R_MOD) == 1:
        k = (k + 1) % R_MOD
    return k


class Permutation:
    """permutation.rs:55-140 -- union-find over the copy constraints, one ascending cycle per class,
    S_sigma evaluations with wire A at omega^i, B at k1 omega^i, C at k2 omega^i."""

    def __init__(self, cs):
        n = cs.n
        self.n = n
        self.k1, self.k2 = find_coset_generator(n, 1), find_coset_generator(n, 2)
        parent = list(range(3 * n))
        rank = [0] * (3 * n)

        def find(x):
            root = x
            while parent[root] != root:
                root = parent[root]
            while parent[x] != root:
                parent[x], x = root, parent[x]
            return root

        for w1, w2 in cs.copy_constraints:
            px, py = find(w1[0] * n + w1[1]), find(w2[0] * n + w2[1])
            if px == py:
                continue
            if rank[px] < rank[py]:
                parent[px] = py
            elif rank[px] > rank[py]:
                parent[py] = px
            else:
                parent[py] = px
                rank[px] += 1
        classes = {}
        for i in range(3 * n):
            classes.setdefault(find(i), []).append(i)
        sigma = list(range(3 * n))
        for members in classes.values():
            if len(members) > 1:
                for i, cur in enumerate(members):
                    sigma[cur] = members[(i + 1) % len(members)]
        self.sigma = sigma
        omega = root_of_unity(n)
        powers, w = [], 1
        for _ in range(n):
            powers.append(w)
            w = w * omega % R_MOD
        ks = (1, self.k1, self.k2)
        enc = lambda idx: ks[idx // n] * powers[idx % n] % R_MOD
        self.s_sigma1_evals = [enc(sigma[i]) for i in range(n)]
        self.s_sigma2_evals = [enc(sigma[n + i]) for i in range(n)]
        self.s_sigma3_evals = [enc(sigma[2 * n + i]) for i in range(n)]


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
        l = [self.next() for _ in range(4)]
        l[3] &= (1 << 62) - 1
        return l[0] | (l[1] << 64) | (l[2] << 128) | (l[3] << 192)


SYNTH_SEED = 0x504C4F4E4B000000
SRS_SEED = 0x5352530000000000


class SyntheticCircuit:
    """2^k-gate alternating mul/add chain: gate i even a*b = c, odd a+b = c, a[i+1] = c[i] by copy
    constraint; a[0], every b[i] and then the eleven blinding scalars from splitmix64(SYNTH_SEED+k);
    the SRS trapdoor from splitmix64(SRS_SEED+k).  sigma is written in closed form: the classes
    {a[i+1], c[i]} become the 2-cycles a[i+1] <-> c[i] of Permutation::compute."""

    def __init__(self, k, native=None):
        """native: generate with the library's host routine (plonk_synthetic_circuit) -- vectors are then
        (n, 4) uint64 arrays of Montgomery limbs instead of lists of ints; the default picks it from 2^17
        gates up, where the Python loop below takes long."""
        n = 1 << k
        self.k, self.n = k, n
        self.public_input_positions = []
        if native is None:
            native = k >= 17
        if native:
            self._native(k)
            return
        rng = SplitMix64(SYNTH_SEED + k)
        a = [0] * n
        b = [0] * n
        c = [0] * n
        cur = rng.fr_rand()
        for i in range(n):
            bi = rng.fr_rand()
            a[i], b[i] = cur, bi
            cur = cur * bi % R_MOD if i % 2 == 0 else (cur + bi) % R_MOD
            c[i] = cur
        self.wires = (a, b, c)
        self.blinding = [rng.fr_rand() for _ in range(11)]
        self.tau = SplitMix64(SRS_SEED + k).fr_rand()
        one, neg1 = 1, R_MOD - 1
        self.selectors = ([one if i % 2 == 0 else 0 for i in range(n)], [0 if i % 2 == 0 else one for i in range(n)],
                          [0 if i % 2 == 0 else one for i in range(n)], [neg1] * n, [0] * n)
        self.k1, self.k2 = find_coset_generator(n, 1), find_coset_generator(n, 2)
        omega = root_of_unity(n)
        powers, w = [], 1
        for _ in range(n):
            powers.append(w)
            w = w * omega % R_MOD
        # sigma(a[i]) = c[i-1] (i >= 1), sigma(c[i]) = a[i+1] (i <= n-2), identity elsewhere
        self.sigmas = ([powers[0]] + [self.k2 * powers[i - 1] % R_MOD for i in range(1, n)],
                       [self.k1 * powers[i] % R_MOD for i in range(n)],
                       [powers[i + 1] for i in range(n - 1)] + [self.k2 * powers[n - 1] % R_MOD])

    def _native(self, k):
        import ctypes as C
        import numpy as np
        from . import _lib
        from .api import _fr_value
        n = self.n
        vecs = [np.empty((n, 4), dtype=np.uint64) for _ in range(11)]
        bl = (_lib.Fr * 11)()
        tau, k1, k2 = _lib.Fr(), _lib.Fr(), _lib.Fr()
        rc = _lib.lib.plonk_synthetic_circuit(k, *[v.ctypes.data_as(C.c_void_p) for v in vecs], bl, C.byref(tau),
                                              C.byref(k1), C.byref(k2))
        if rc != 0:
            raise _lib.PlonkError(rc, "plonk_synthetic_circuit(%d)" % k)
        self.wires = tuple(vecs[0:3])
        self.selectors = tuple(vecs[3:8])
        self.sigmas = tuple(vecs[8:11])
        self.blinding = [_fr_value(x) for x in bl]
        self.tau, self.k1, self.k2 = _fr_value(tau), _fr_value(k1), _fr_value(k2)
