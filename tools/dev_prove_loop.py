"""Proves the synthetic 2^k-gate circuit a few times (development aid; run under rocprofv3)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plonk_amd as pa
from plonk_amd import circuits

k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
circ = circuits.SyntheticCircuit(k)
n = circ.n
srs = pa.Srs.from_tau(circ.tau, n + 10)
pk = pa.ProvingKey(n, circ.selectors, circ.sigmas, circ.k1, circ.k2, ())
host = len(sys.argv) > 3 and sys.argv[3] == "host"   # witness in (pageable) host memory: plonk_prove_3wire
prover = pa.Prover(circ.wires, pk, srs, resident=not host)
prover.prove([], circ.blinding)
t = time.perf_counter()
for _ in range(reps):
    prover.prove([], circ.blinding)
print("prove 2^%d%s: %.3f ms" % (k, " (host witness)" if host else "", (time.perf_counter() - t) / reps * 1e3))
