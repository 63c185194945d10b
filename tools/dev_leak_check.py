"""Proves the synthetic circuit many times and prints the free device memory before and after (development aid:
the library allocates its workspaces and transform plans on first use and must not grow afterwards)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import plonk_amd as pa
from plonk_amd import circuits
def free_mib():
    return torch.cuda.mem_get_info()[0] / 2**20
for k, reps in ((14, 300), (20, 40)):
    circ = circuits.SyntheticCircuit(k)
    srs = pa.Srs.from_tau(circ.tau, circ.n + 10)
    pk = pa.ProvingKey(circ.n, circ.selectors, circ.sigmas, circ.k1, circ.k2, ())
    prover = pa.Prover(circ.wires, pk, srs, resident=True)
    prover.prove([], circ.blinding)
    a = free_mib()
    t = time.perf_counter()
    for _ in range(reps):
        p = prover.prove([], circ.blinding)
    dt = (time.perf_counter() - t) / reps * 1e3
    b = free_mib()
    print("2^%d: %d proofs, %.3f ms each, free memory %.0f -> %.0f MiB (delta %.1f)" % (k, reps, dt, a, b, a - b), flush=True)
