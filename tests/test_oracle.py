"""The CPU oracle against every known-answer test the reference's own unit tests hold for the
path (file:line of each reference test in the docstrings), against published BLS12-381 constants,
against the committed golden fixtures, and the C restatement against the Python one."""
import hashlib
import random

import pytest

from conftest import golden
from oracle import cref, curve, fft as offt, protocol as P
from oracle.field import (R_MOD, Q_MOD, fr_add, fr_sub, fr_mul, fr_inv, fr_to_bytes, fr_from_bytes, fr_to_hex,
                          root_of_unity)


def test_basic_arithmetic_and_inverse():
    """field.rs:328-342."""
    assert fr_add(5, 3) == 8 and fr_sub(5, 3) == 2 and fr_mul(5, 3) == 15
    assert fr_mul(12345, fr_inv(12345)) == 1


def test_root_of_unity():
    """field.rs:345-354 and the constants of the two-adic subgroup."""
    w = root_of_unity(16)
    assert pow(w, 16, R_MOD) == 1 and pow(w, 8, R_MOD) != 1
    assert w == 0x20B1CE9140267AF9DD1C0AF834CEC32C17BEB312F20B6F7653EA61D87742BCCE
    assert root_of_unity(1 << 20) == 0x03E1C54BCB947035A57A6E07CB98DE4A2F69E02D265E09D9FECE7E0E39898D4B
    assert root_of_unity(3) is None and root_of_unity(1 << 33) is None


def test_hex_and_bytes_roundtrip():
    """field.rs:357-362."""
    v = 12345678901234567890
    assert fr_from_bytes(fr_to_bytes(v)) == v
    assert int(fr_to_hex(v), 16) == v
    assert fr_from_bytes(R_MOD.to_bytes(32, "little")) is None


def test_fft_ifft_roundtrip():
    """fft.rs:278-297."""
    w = root_of_unity(8)
    assert offt.ifft(offt.fft([1, 2, 3, 4], w, 8), w) == [1, 2, 3, 4]


def test_fft_evaluates_correctly():
    """fft.rs:300-318 -- output order is natural."""
    w = root_of_unity(4)
    ev = offt.fft([1, 2, 3], w, 4)
    for i in range(4):
        assert ev[i] == P.poly_eval([1, 2, 3], pow(w, i, R_MOD))


def test_domain_and_vanishing():
    """fft.rs:321-347."""
    d = offt.Domain(8)
    el = d.elements()
    assert el[0] == 1 and el[1] == d.omega
    assert all(d.vanishing_eval(x) == 0 for x in el) and d.vanishing_eval(123) != 0


def test_polynomial_kats():
    """polynomial.rs:437-498."""
    assert P.poly_eval([1, 2, 3], 2) == 17
    assert P.poly_div_by_linear([R_MOD - 1, 0, 1], 1) == [1, 1]
    assert P.poly_mul([1, 1], [1, 1]) == [1, 2, 1]
    assert P.poly_eval(P.poly_vanishing(4), root_of_unity(4)) == 0


def test_coset_roundtrip_and_shift():
    rng = random.Random(5)
    v = [rng.randrange(R_MOD) for _ in range(16)]
    w = root_of_unity(16)
    ev = offt.coset_fft(v, w, 7, 16)
    assert ev[3] == P.poly_eval(v, 7 * pow(w, 3, R_MOD) % R_MOD)
    assert offt.coset_ifft(ev, w, 7) == v
    # z(omega X) on the 4n coset is z's coset evaluations rotated by 4 (what the HIP path uses)
    n = 4
    z = [rng.randrange(R_MOD) for _ in range(n + 3)]
    big = offt.Domain(4 * n)
    zs = P.evaluate_coset(P.poly_shift(z, root_of_unity(n)), big, 7)
    zc = P.evaluate_coset(z, big, 7)
    assert zs == [zc[(i + 4) % (4 * n)] for i in range(4 * n)]


def test_g1_kats():
    """curve.rs:493-544: G+G = 2G; MSM([G,G,G],[1,2,3]) = 6G."""
    G = curve.G1_GEN
    assert curve.to_affine(curve.jac_add(curve.to_jac(G), curve.to_jac(G))) == curve.to_affine(curve.scalar_mul(G, 2))
    assert curve.to_affine(curve.msm([G, G, G], [1, 2, 3])) == curve.to_affine(curve.scalar_mul(G, 6))
    assert curve.to_affine(curve.scalar_mul(G, R_MOD)) is None  # group order


def test_g1_encoding_constants():
    """ZCash/IETF encoding of the generator and of the identity (public BLS12-381 constants)."""
    G = curve.G1_GEN
    assert curve.compress(G).hex() == ("97f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac58"
                                       "6c55e83ff97a1aeffb3af00adb22c6bb")
    assert curve.compress(None).hex() == "c0" + "00" * 47
    assert curve.decompress(curve.compress(G)) == G
    negG = curve.affine_neg(G)
    assert curve.compress(negG)[0] & 0x20 and curve.decompress(curve.compress(negG)) == negG
    assert len(curve.uncompressed(G)) == 96


def test_commit_and_open():
    """kzg.rs:442-469: f = 1 + 2X + 3X^2, f(5) = 86; the opening identity holds, y = 87 does not."""
    srs = P.Srs(0xC0FFEE, 10)
    f = [1, 2, 3]
    assert P.poly_eval(f, 5) == 86
    comm = P.commit(f, srs)
    assert comm == P.commit(f, srs, use_trapdoor=True)
    q = P.poly_div_by_linear([(1 - 86) % R_MOD, 2, 3], 5)
    proof = P.commit(q, srs)
    G = curve.G1_GEN
    # e([f] - y[1], [1]) = e([pi], [tau] - z[1])  <=>  [f] - yG == (tau - z) * [pi]
    def holds(y):
        lhs = curve.jac_add(curve.to_jac(comm), curve.jac_neg(curve.scalar_mul(G, y)))
        return curve.to_affine(lhs) == curve.to_affine(curve.scalar_mul(proof, (srs.tau - 5) % R_MOD))
    assert holds(86) and not holds(87)
    with pytest.raises(ValueError):
        P.commit([1] * 12, srs)


def test_transcript():
    """transcript.rs:234-281 (determinism, sensitivity) + the byte layout of absorb/squeeze."""
    def run(extra):
        t = P.Transcript(b"PLONK-v1")
        t.absorb_fr("x", 5)
        t.absorb_g1("g", curve.G1_GEN)
        t.absorb_bytes("e", extra)
        return t.squeeze_challenge("c"), t.squeeze_challenge("c")
    assert run(b"a") == run(b"a") and run(b"a") != run(b"b")
    c1, c2 = run(b"a")
    assert c1 != c2 and c1 < R_MOD
    h = hashlib.sha256(b"PLONK-v1" + b"x" + (32).to_bytes(8, "little") + fr_to_bytes(5) + b"g"
                       + curve.compress(curve.G1_GEN) + b"e" + (1).to_bytes(8, "little") + b"a" + b"c").digest()
    assert c1 == P.hash_to_fr(h)
    big = (R_MOD + 5).to_bytes(32, "little")
    assert P.hash_to_fr(big) == P.hash_to_fr(hashlib.sha256(big).digest())


def test_permutation():
    """permutation.rs:374-419."""
    cs = P.ConstraintSystem()
    for _ in range(4):
        cs.add_gate(P.Gate.dummy())
    cs.finalize()
    d = offt.Domain(cs.n)
    perm = P.Permutation(cs, d)
    assert perm.sigma == list(range(3 * cs.n)) and (perm.k1, perm.k2) == (6, 7)
    assert perm.s_sigma2_evals[1] == 6 * d.omega % R_MOD
    cs = P.range_proof_circuit(8, 42)
    perm = P.Permutation(cs, offt.Domain(16))
    n = 16
    for w1, w2 in cs.copy_constraints:
        i1, i2 = w1[0] * n + w1[1], w2[0] * n + w2[1]
        cur, seen = i1, False
        for _ in range(3 * n):
            if cur == i2:
                seen = True
                break
            cur = perm.sigma[cur]
        assert seen
    assert pow(6, 8, R_MOD) != 1 and pow(7, 8, R_MOD) != 1
    assert P.find_coset_generator(1 << 20, 1) == 6 and P.find_coset_generator(1 << 20, 2) == 7


def test_range_proof_circuit():
    """range_proof.rs:534-650."""
    cs = P.range_proof_circuit(8, 42)
    assert cs.num_gates == 16 and cs.n == 16 and cs.public_input_positions == [] and cs.is_satisfied()
    assert [g.a for g in cs.gates[:8]] == [0, 1, 0, 1, 0, 1, 0, 0]
    assert cs.gates[15].c == 42
    assert P.range_proof_circuit(8, 300).gates[15].c == 44
    for x in (0, 255):
        assert P.range_proof_circuit(8, x).is_satisfied()


def test_preprocess_and_prove_verify():
    """preprocess.rs:320-357, prover.rs:937-972, verifier.rs:366-482."""
    cs = P.range_proof_circuit(8, 42)
    srs = P.Srs(0xABCDEF123456789, cs.n + 10)
    pk, vk, perm, dom = P.preprocess(cs, srs)
    assert pk.n == 16 and len(pk.q_m) <= 16 and len(pk.l1) == 16
    rng = random.Random(11)
    bl = [rng.randrange(R_MOD) for _ in range(11)]
    proof = P.prove(cs, pk, perm, dom, srs, [42], bl)
    assert all(proof[k] is not None for k in ("a_comm", "z_comm", "t_lo_comm", "w_zeta_comm"))
    assert P.verify(vk, srs, proof, [42])
    assert not P.verify(vk, srs, proof, [43])
    assert P.prove(cs, pk, perm, dom, srs, [42], bl, use_trapdoor=True) == proof
    assert len(P.proof_to_bytes(proof)) == 624


def test_pairing_properties():
    """What curve.rs:535-544 checks of the pairing: bilinearity and non-degeneracy; plus the
    order of GT elements and of the G2 generator."""
    from oracle import pairing as pr
    G, H = curve.G1_GEN, pr.G2_GEN
    assert pr.g2_is_on_curve(H) and pr.g2_mul(H, R_MOD) is None
    e = pr.pairing(G, H)
    assert e != pr.F12_ONE and pr.f12_pow(e, R_MOD) == pr.F12_ONE
    a, b = 0x1234567, R_MOD - 5
    aG = curve.to_affine(curve.scalar_mul(G, a))
    bH = pr.g2_mul(H, b)
    assert pr.pairing(aG, bH) == pr.f12_pow(e, a * b % R_MOD)
    assert pr.pairing(aG, H) == pr.pairing(G, pr.g2_mul(H, a))
    assert pr.pairing(None, H) == pr.F12_ONE and pr.pairing(G, None) == pr.F12_ONE
    assert pr.pairing_check(aG, bH, curve.to_affine(curve.scalar_mul(G, a * b % R_MOD)), H)
    assert not pr.pairing_check(aG, bH, curve.to_affine(curve.scalar_mul(G, a * b % R_MOD + 1)), H)
    assert pr.multi_pairing([(G, H), (curve.affine_neg(G), H)]) == pr.F12_ONE


def test_verify_with_the_pairing_agrees_with_the_trapdoor_form():
    """verifier.rs:309-314 evaluated as the reference does, against [tau]_2 (kzg.rs:86) alone."""
    cs = P.range_proof_circuit(8, 42)
    srs = P.Srs(0xABCDEF123456789, cs.n + 10)
    pk, vk, perm, dom = P.preprocess(cs, srs)
    rng = random.Random(11)
    proof = P.prove(cs, pk, perm, dom, srs, [42], [rng.randrange(R_MOD) for _ in range(11)])
    assert P.verify_pairing(vk, srs.g2_tau, proof, [42])
    assert not P.verify_pairing(vk, srs.g2_tau, proof, [43])
    bad = dict(proof, z_omega_eval=(proof["z_omega_eval"] + 1) % R_MOD)
    assert not P.verify_pairing(vk, srs.g2_tau, bad, [42])
    from oracle import pairing as pr
    assert not P.verify_pairing(vk, pr.g2_mul(pr.G2_GEN, srs.tau + 1), proof, [42])   # another SRS


def test_golden_ntt():
    for e in golden("ntt.json"):
        n = e["n"]
        rng = random.Random(e["seed"])
        v = [rng.randrange(R_MOD) for _ in range(n)]
        w = root_of_unity(n)
        res = {"fft": offt.fft(v, w, n), "ifft": (offt.ifft(v, w) + [0] * n)[:n],
               "coset_fft": offt.coset_fft(v, w, 7, n), "coset_ifft": offt.coset_ifft(v, w, 7)}
        for k, r in res.items():
            assert hashlib.sha256(b"".join(fr_to_bytes(x) for x in r)).hexdigest() == e["sha256"][k]
            if "output" in e:
                assert [int(x, 16) for x in e["output"][k]] == r
        if "input" in e:
            assert [int(x, 16) for x in e["input"]] == v


def test_golden_msm_and_proofs():
    for e in golden("msm.json"):
        if "points" not in e:
            continue
        pts = [curve.decompress(bytes.fromhex(p)) for p in e["points"]]
        sc = [int(s, 16) for s in e["scalars"]]
        assert curve.compress(curve.to_affine(curve.msm(pts, sc))).hex() == e["commit"]
    for e in golden("proofs.json"):
        if e["n"] > 64:
            continue
        cs = {"c1_square_add": lambda: P.square_add_circuit(), "c2_range_proof_255": lambda: P.range_proof_circuit(8, 255),
              "range_proof_42_pi": lambda: P.range_proof_circuit(8, 42),
              "synthetic_2^6": lambda: P.synthetic_circuit(6)[0]}[e["name"]]()
        srs = P.Srs(int(e["tau"], 16), cs.n + 10)
        pk, vk, perm, dom = P.preprocess(cs, srs)  # MSM path, not the trapdoor path that made the fixture
        pis = [int(x, 16) for x in e["public_inputs"]]
        proof = P.prove(cs, pk, perm, dom, srs, pis, [int(b, 16) for b in e["blinding"]])
        assert P.proof_to_bytes(proof).hex() == e["proof"]
        assert [curve.compress(vk[k]).hex() for k in ("q_m", "q_l", "q_r", "q_o", "q_c", "s_sigma1", "s_sigma2", "s_sigma3")] == e["vk"]


def test_synthetic_circuit():
    cs, bl = P.synthetic_circuit(4)
    assert cs.n == 16 and cs.is_satisfied() and len(bl) == 11 and len(cs.copy_constraints) == 15
    assert cs.gates[0].q_m == 1 and cs.gates[1].q_l == 1 and cs.gates[1].a == cs.gates[0].c
    assert P.synthetic_tau(4) < R_MOD


def test_c_restatement_matches_python():
    rng = random.Random(3)
    for log_n in (0, 1, 5, 9):
        n = 1 << log_n
        v = [rng.randrange(R_MOD) for _ in range(n)]
        w = root_of_unity(n)
        exp = [offt.fft(v, w, n), (offt.ifft(v, w) + [0] * n)[:n], offt.coset_fft(v, w, 7, n), offt.coset_ifft(v, w, 7)]
        for fl in range(4):
            assert cref.fr_list(cref.ntt(cref.fr_array(v), fl)) == exp[fl]
    srs = P.Srs(424242, 39)
    sc = [rng.randrange(R_MOD) for _ in range(40)]
    sc[3], sc[4], sc[5] = 0, 1, R_MOD - 1
    pts = list(srs.g1_powers)
    pts[7] = None
    assert cref.msm(cref.points_array(pts), cref.fr_array(sc)) == curve.to_affine(curve.msm(pts, sc))
    assert cref.msm(cref.points_array(pts[:2]), cref.fr_array([0, 0])) is None
    # quotient loop
    import numpy as np
    cs = P.range_proof_circuit(8, 200)
    srs = P.Srs(99, 26)
    pk, vk, perm, dom = P.preprocess(cs, srs, use_trapdoor=True)
    tr = {}
    P.prove(cs, pk, perm, dom, srs, [], [rng.randrange(R_MOD) for _ in range(11)], use_trapdoor=True, trace=tr)
    big = offt.Domain(64)
    ec = lambda p: P.evaluate_coset(p, big, 7)
    polys = [tr["a"], tr["b"], tr["c"], tr["z"], P.poly_shift(tr["z"], pk.omega), pk.q_m, pk.q_l, pk.q_r, pk.q_o, pk.q_c,
             pk.s_sigma1, pk.s_sigma2, pk.s_sigma3, pk.l1, []]
    ext = np.stack([cref.fr_array(ec(p)) for p in polys])
    t = cref.fr_list(cref.quotient(ext, 16, tr["alpha"], tr["beta"], tr["gamma"], pk.k1, pk.k2))
    assert t == tr["t_coset"]
    assert cref.fr_list(cref.quotient(ext, 16, tr["alpha"], tr["beta"], tr["gamma"], pk.k1, pk.k2, 5, 9)) == tr["t_coset"][5:14]


def test_prover_with_the_c_engine_matches_the_python_oracle():
    """oracle.cref.Engine (the transforms and the round-3 loop run by the C restatement, what generates the
    2^18 / 2^20-gate fixtures) gives the proofs the all-Python oracle gave: the golden 2^6 / 2^10 bytes, and a
    circuit with public inputs against a fresh Python run."""
    class _S:
        pass

    gold = {e["name"]: e for e in golden("proofs.json")}
    eng = cref.Engine(4)
    for k in (6, 10):
        cs, bl = P.synthetic_circuit(k)
        srs = _S()
        srs.tau, srs.max_degree, srs.g1_powers = P.synthetic_tau(k), cs.n + 10, [None] * (cs.n + 11)
        pk, vk, perm, dom = P.preprocess(cs, srs, use_trapdoor=True, engine=eng)
        tr = {}
        proof = P.prove(cs, pk, perm, dom, srs, [], bl, use_trapdoor=True, trace=tr, engine=eng)
        e = gold["synthetic_2^%d" % k]
        assert P.proof_to_bytes(proof).hex() == e["proof"]
        assert [curve.compress(vk[key]).hex() for key in ("q_m", "q_l", "q_r", "q_o", "q_c", "s_sigma1", "s_sigma2", "s_sigma3")] == e["vk"]
        assert ["%064x" % tr[key] for key in ("beta", "gamma", "alpha", "zeta", "v", "u")] == e["challenges"]
    cs = P.range_proof_circuit(8, 42)
    srs = _S()
    srs.tau, srs.max_degree, srs.g1_powers = 424242, cs.n + 10, [None] * (cs.n + 11)
    bl = list(range(3, 14))
    pk, vk, perm, dom = P.preprocess(cs, srs, use_trapdoor=True)
    want = P.prove(cs, pk, perm, dom, srs, [42], bl, use_trapdoor=True)
    pk2, vk2, perm2, dom2 = P.preprocess(cs, srs, use_trapdoor=True, engine=eng)
    assert vk2 == vk
    assert P.prove(cs, pk2, perm2, dom2, srs, [42], bl, use_trapdoor=True, engine=eng) == want
