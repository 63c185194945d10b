"""Throughput with several proofs in flight on one GPU (development aid): T host threads, a context and a resident
witness each, one shared proving key and SRS; prints proofs per second against one prover alone."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plonk_amd as pa
from plonk_amd import circuits

k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
circ = circuits.SyntheticCircuit(k)
n = circ.n
ctx0 = pa.api.default_context()
srs = pa.Srs.from_tau(circ.tau, n + 10)
pk = pa.ProvingKey(n, circ.selectors, circ.sigmas, circ.k1, circ.k2, ())


class Worker:
    def __init__(self, ctx):
        self.ctx = ctx
        # a prover bound to this worker's context: same key and SRS handles, its own streams and workspaces
        self.pk = pa.ProvingKey.__new__(pa.ProvingKey)
        self.pk.__dict__.update(pk.__dict__)
        self.pk.ctx = ctx
        self.srs = pa.Srs(srs.h, ctx)
        self.prover = pa.Prover(circ.wires, self.pk, self.srs, resident=True)
        self.proof = self.prover.prove([], circ.blinding)

    def run(self, count):
        for _ in range(count):
            self.proof = self.prover.prove([], circ.blinding)


for threads in (1, 2, 3):
    workers = [Worker(ctx0 if i == 0 else pa.Context(0)) for i in range(threads)]
    th = [threading.Thread(target=w.run, args=(reps,)) for w in workers]
    t = time.perf_counter()
    for x in th:
        x.start()
    for x in th:
        x.join()
    dt = time.perf_counter() - t
    assert all(w.proof.to_bytes() == workers[0].proof.to_bytes() for w in workers)
    print("%d prover(s) in flight: %.2f proofs/s, %.3f ms per proof" % (threads, threads * reps / dt, dt / (threads * reps) * 1e3), flush=True)
    for w in workers:
        w.srs.h = None          # the handles belong to the first objects
        w.pk.h = None
        w.prover.close()
