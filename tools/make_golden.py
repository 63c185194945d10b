"""Generates tests/golden/*.json from the CPU oracle (oracle/).  The reference ships no golden
vectors and cannot be run here, so these pin the oracle's own outputs: any later change to the
oracle or the HIP path that alters a byte shows up against them.
Usage: python tools/make_golden.py [--big] [--huge] [--only-huge]
  --big   adds the 2^14 and 2^16 synthetic proofs (Python oracle throughout, about an hour);
  --huge  adds 2^18 and 2^20 (the benchmark size): oracle/protocol.py with its transforms and its
          round-3 loop run by the C restatement (oracle.cref.Engine), commitments as f(tau)*G;
          minutes on 8 cores, ~6 GiB of memory;
  --only-huge  leaves every other entry of proofs.json as it is."""
import hashlib, json, os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import protocol as P, curve, cref, fft as offt
from oracle.field import R_MOD, root_of_unity, fr_to_bytes

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
os.makedirs(OUT, exist_ok=True)
hx = lambda v: "%064x" % v


def seeded(seed, n):
    rng = random.Random(seed)
    return [rng.randrange(R_MOD) for _ in range(n)]


def digest(vals):
    return hashlib.sha256(b"".join(fr_to_bytes(v) for v in vals)).hexdigest()


def ntt_vectors():
    out = []
    for n in (4, 8, 16, 64, 1024):
        v = seeded(1000 + n, n)
        w = root_of_unity(n)
        res = {"fft": offt.fft(v, w, n), "ifft": (offt.ifft(v, w) + [0] * n)[:n],
               "coset_fft": offt.coset_fft(v, w, 7, n), "coset_ifft": offt.coset_ifft(v, w, 7)}
        e = {"n": n, "seed": 1000 + n, "shift": 7}
        if n <= 64:
            e["input"] = [hx(x) for x in v]
            e["output"] = {k: [hx(x) for x in r] for k, r in res.items()}
        e["sha256"] = {k: digest(r) for k, r in res.items()}
        out.append(e)
    json.dump(out, open(os.path.join(OUT, "ntt.json"), "w"), indent=0)


def msm_vectors():
    out = []
    tau = seeded(77, 1)[0]
    for n in (1, 3, 18, 1024):
        sc = seeded(2000 + n, n)
        expected = curve.to_affine(curve.scalar_mul(curve.G1_GEN, P.poly_eval(sc, tau)))
        e = {"n": n, "tau": hx(tau), "scalar_seed": 2000 + n, "commit": curve.compress(expected).hex()}
        if n <= 18:
            srs = P.Srs(tau, n - 1)
            e["points"] = [curve.compress(p).hex() for p in srs.g1_powers]
            e["scalars"] = [hx(x) for x in sc]
            assert curve.to_affine(curve.msm(srs.g1_powers, sc)) == expected
        out.append(e)
    json.dump(out, open(os.path.join(OUT, "msm.json"), "w"), indent=0)


class _TrapSrs:
    def __init__(self, tau, max_degree):
        self.tau, self.max_degree, self.g1_powers = tau, max_degree, [None] * (max_degree + 1)


def proof_case(name, cs, tau, blinding, pis, engine=None):
    srs = _TrapSrs(tau, cs.n + 10)
    pk, vk, perm, dom = P.preprocess(cs, srs, use_trapdoor=True, engine=engine)
    tr = {}
    proof = P.prove(cs, pk, perm, dom, srs, pis, blinding, use_trapdoor=True, trace=tr, engine=engine)
    assert P.verify(vk, srs, proof, pis)
    return {"name": name, "n": cs.n, "tau": hx(tau), "blinding": [hx(b) for b in blinding],
            "public_inputs": [hx(p) for p in pis], "proof": P.proof_to_bytes(proof).hex(),
            "vk": [curve.compress(vk[k]).hex() for k in ("q_m", "q_l", "q_r", "q_o", "q_c", "s_sigma1", "s_sigma2", "s_sigma3")],
            "challenges": [hx(tr[k]) for k in ("beta", "gamma", "alpha", "zeta", "v", "u")]}


def proof_vectors(big, huge, only_huge):
    path = os.path.join(OUT, "proofs.json")
    old = {e["name"]: e for e in json.load(open(path))} if os.path.exists(path) else {}
    out = []
    if not only_huge:
        tau = seeded(99, 1)[0]
        bl = seeded(98, 11)
        out = [proof_case("c1_square_add", P.square_add_circuit(), tau, bl, []),
               proof_case("c2_range_proof_255", P.range_proof_circuit(8, 255), tau, bl, []),
               proof_case("range_proof_42_pi", P.range_proof_circuit(8, 42), tau, bl, [42])]
    ks = ([] if only_huge else [6, 10] + ([14, 16] if big else [])) + (huge if huge else [])
    for k in ks:
        cs, blk = P.synthetic_circuit(k)
        engine = cref.Engine(os.cpu_count() or 1) if k >= 17 else None
        out.append(proof_case("synthetic_2^%d" % k, cs, P.synthetic_tau(k), blk, [], engine))
        print("synthetic", k, "done", flush=True)
    fresh = {e["name"]: e for e in out}
    merged = [fresh.pop(name, e) for name, e in old.items()]   # earlier cases keep their place (and their value unless regenerated)
    merged += list(fresh.values())
    merged.sort(key=lambda e: e["n"] if e["name"].startswith("synthetic") else 0)   # stable: named cases first, then by size
    json.dump(merged, open(path, "w"), indent=0)


if __name__ == "__main__":
    only_huge = "--only-huge" in sys.argv
    if not only_huge:
        ntt_vectors()
        msm_vectors()
    huge = [int(a.split("=")[1]) for a in sys.argv if a.startswith("--huge=")] or ([18, 20] if "--huge" in sys.argv or only_huge else [])
    proof_vectors("--big" in sys.argv, huge, only_huge)
    print("golden written to", OUT)
