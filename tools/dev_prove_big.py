"""Timing and digest of one proof of the synthetic 2^k-gate circuit (default k = 24) on one GPU."""
import hashlib
import sys
import time

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import plonk_amd as pa
from plonk_amd import circuits

k = int(sys.argv[1]) if len(sys.argv) > 1 else 24
t = time.time(); circ = circuits.SyntheticCircuit(k); print("circuit %.1fs" % (time.time() - t), flush=True)
n = circ.n
t = time.time(); srs = pa.Srs.from_tau(circ.tau, n + 10); print("srs %.1fs" % (time.time() - t), flush=True)
t = time.time(); pk = pa.ProvingKey(n, circ.selectors, circ.sigmas, circ.k1, circ.k2, ()); print("pk %.1fs" % (time.time() - t), flush=True)
prover = pa.Prover(circ.wires, pk, srs, resident=True)
for i in range(3):
    t = time.time(); proof = prover.prove([], circ.blinding); print("prove %.1f ms" % ((time.time() - t) * 1e3), flush=True)
print("digest", hashlib.sha256(proof.to_bytes()).hexdigest()[:16])
