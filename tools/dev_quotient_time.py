"""Device time of quotient_kernel inside a proof of 2^k gates (development aid): the event marks bench.py reads."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plonk_amd as pa
from plonk_amd import circuits

k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
circ = circuits.SyntheticCircuit(k)
ctx = pa.Context()
srs = pa.Srs.from_tau(circ.tau, circ.n + 10, ctx=ctx)
pk = pa.ProvingKey(circ.n, circ.selectors, circ.sigmas, circ.k1, circ.k2, (), ctx=ctx)
prover = pa.Prover(circ.wires, pk, srs, resident=True)
prover.prove([], circ.blinding)
ctx.msm_set_timing(True)
ts = []
for _ in range(8):
    prover.prove([], circ.blinding)
    ts.append(ctx.quotient_last_timing())
print("quotient_kernel 2^%d: min %.4f ms, median %.4f ms" % (k + 2, min(ts), sorted(ts)[len(ts) // 2]))
