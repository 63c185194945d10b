"""Development check: MSM / SRS / commit / prove against the oracle."""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plonk_amd as pa
from oracle import protocol as P, curve, field
from oracle.field import R_MOD

ok = True
def check(name, cond):
    global ok
    ok &= bool(cond)
    print(("OK   " if cond else "FAIL ") + name, flush=True)

t0 = time.time()
ctx = pa.api.default_context()
G = curve.G1_GEN
# curve.rs:535-544 KAT
r = pa.msm([G, G, G], [1, 2, 3])
check("msm([G,G,G],[1,2,3]) == 6G", r == curve.to_affine(curve.scalar_mul(G, 6)))
check("msm([], []) identity", pa.msm([], []) is None)
check("msm zero scalars", pa.msm([G, G], [0, 0]) is None)
check("msm with identity point", pa.msm([None, G], [5, 7]) == curve.to_affine(curve.scalar_mul(G, 7)))
check("msm P + (-P)", pa.msm([G, curve.affine_neg(G)], [9, 9]) is None)
check("msm r-1", pa.msm([G], [R_MOD - 1]) == curve.affine_neg(G))
rng = random.Random(7)
tau = rng.randrange(R_MOD)
for n in (1, 2, 18, 22, 100, 700, 3000):
    srs = pa.Srs.from_tau(tau, n - 1)
    pts = srs.g1_powers()
    if n <= 100:
        exp_pts = P.Srs(tau, n - 1).g1_powers
        check("srs_from_tau n=%d" % n, pts == exp_pts)
    sc = [rng.randrange(R_MOD) for _ in range(n)]
    got = pa.Commitment.commit(sc, srs).point
    exp = curve.to_affine(curve.scalar_mul(G, P.poly_eval(sc, tau)))
    check("commit n=%d (uniform scalars)" % n, got == exp)
    sc2 = [rng.choice([0, 1, R_MOD - 1, 2, 255]) for _ in range(n)]
    got = pa.Commitment.commit(sc2, srs).point
    exp = curve.to_affine(curve.scalar_mul(G, P.poly_eval(sc2, tau)))
    check("commit n=%d (skewed scalars)" % n, got == exp)
    if n == 100:
        check("msm n=100 vs oracle msm", pa.msm(pts, sc) == curve.to_affine(curve.msm(pts, sc)))
print("msm phase %.1fs" % (time.time() - t0), flush=True)

# C2: range proof 255
def run_case(name, cs, tau, blinding, pis):
    t = time.time()
    n = cs.n
    osrs = P.Srs(tau, n + 10) if n <= 64 else None
    class _S: pass
    if osrs is None:
        osrs = _S(); osrs.tau = tau; osrs.max_degree = n + 10; osrs.g1_powers = [None] * (n + 11)
    pk_o, vk_o, perm, dom = P.preprocess(cs, osrs, use_trapdoor=True)
    proof_o = P.prove(cs, pk_o, perm, dom, osrs, pis, blinding, use_trapdoor=True)
    print("  oracle %.1fs" % (time.time() - t), flush=True); t = time.time()
    srs = pa.Srs.from_tau(tau, n + 10)
    pk = pa.ProvingKey(n, cs.selectors(), [perm.s_sigma1_evals, perm.s_sigma2_evals, perm.s_sigma3_evals],
                       perm.k1, perm.k2, cs.public_input_positions)
    vkc = pk.commitments(srs)
    check(name + " vk commitments", vkc == [vk_o[k] for k in ("q_m", "q_l", "q_r", "q_o", "q_c", "s_sigma1", "s_sigma2", "s_sigma3")])
    prover = pa.Prover(cs.wire_values(), pk, srs)
    proof = prover.prove(pis, blinding)
    print("  gpu %.1fs" % (time.time() - t), flush=True)
    d = proof.to_dict()
    for k in P.PROOF_G1 + P.PROOF_FR:
        if d[k] != proof_o[k]:
            print("   mismatch", k)
    check(name + " proof fields", d == proof_o)
    check(name + " proof bytes", proof.to_bytes() == P.proof_to_bytes(proof_o))
    check(name + " oracle verifier accepts", P.verify(vk_o, osrs, d, pis))

bl = [rng.randrange(R_MOD) for _ in range(11)]
run_case("C2 range_proof(8,255)", P.range_proof_circuit(8, 255), tau, bl, [])
run_case("range_proof(8,42) pi=[42]", P.range_proof_circuit(8, 42), tau, bl, [42])
run_case("C1 square_add", P.square_add_circuit(), tau, bl, [])
for k in (6, 10):
    cs, blk = P.synthetic_circuit(k)
    run_case("synthetic 2^%d" % k, cs, P.synthetic_tau(k), blk, [])
print("ALL OK" if ok else "FAILED", "%.1fs" % (time.time() - t0))
sys.exit(0 if ok else 1)
