"""Times proofs of small circuits (development aid)."""
import sys
import time

import plonk_amd as pa
from plonk_amd import circuits

for k in (4, 8, 10, 12, 14):
    if k == 4:
        cs = circuits.range_proof_circuit(8, 255)
        perm = circuits.Permutation(cs)
        wires, sel, sig, k1, k2, n = cs.wire_values(), cs.selectors(), [perm.s_sigma1_evals, perm.s_sigma2_evals, perm.s_sigma3_evals], perm.k1, perm.k2, cs.n
        tau, bl = 12345, list(range(1, 12))
    else:
        c = circuits.SyntheticCircuit(k)
        wires, sel, sig, k1, k2, n, tau, bl = c.wires, c.selectors, c.sigmas, c.k1, c.k2, c.n, c.tau, c.blinding
    srs = pa.Srs.from_tau(tau, n + 10)
    pk = pa.ProvingKey(n, sel, sig, k1, k2, ())
    pr = pa.Prover(wires, pk, srs, resident=True)
    pr.prove([], bl)
    t = time.perf_counter()
    for _ in range(5):
        pr.prove([], bl)
    print("n = 2^%d: %.3f ms per proof" % (k, (time.perf_counter() - t) / 5 * 1e3), flush=True)
