"""Per-rank cost of a sharded proof without the other ranks: one rank of a `world`-rank proof runs alone
on the GPU behind the library's loopback communicator (plonk_comm_create_loopback).  The proof that
comes out is wrong; the time the rank spends on its shares of the MSMs, transforms and quotient is what
a real rank spends, minus the exchange itself.  Development aid for the scaling table.
usage: dev_shard_timing.py <log_n> <world> [reps] [rank] [grouped]
(grouped = 1: the rounds with two / three commitments place them on groups of ranks)"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plonk_amd as pa
from plonk_amd import _lib, circuits, sharded

k, world = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
circ = circuits.SyntheticCircuit(k)
n = circ.n
ctx = pa.Context(0)
ranges = [sharded.point_range(n + 11, r, world) for r in range(world)]


h = C.c_void_p()
rank = int(sys.argv[4]) if len(sys.argv) > 4 else 0
ctx.check(_lib.lib.plonk_comm_create_loopback(ctx.h, rank, world, n + 11, C.byref(h)))


class Comm:
    pass


comm = Comm()
comm.h, comm.world = h, world
lo, hi = ranges[rank]
srs = pa.Srs.from_tau_range(circ.tau, lo, hi, ctx=ctx)
pk = pa.ProvingKey(n, circ.selectors, circ.sigmas, circ.k1, circ.k2, (), ctx=ctx)
groups = {}
if len(sys.argv) > 5 and sys.argv[5] == "1":
    for b in (2, 3):
        if world >= b:
            glo, ghi = sharded.group_range(n + 11, rank, world, b)
            groups[b] = (pa.Srs.from_tau_range(circ.tau, glo, ghi, ctx=ctx), (glo, ghi))
prover = pa.Prover(circ.wires, pk, srs, resident=True, comm=comm, shard=(lo, hi), groups=groups)
prover.prove([], circ.blinding)
ctx.sync()
t = time.perf_counter()
for _ in range(reps):
    prover.prove([], circ.blinding)
ctx.sync()
print("2^%d gates, rank %d of %d%s: %.2f ms per proof without the exchange (which moves %.1f MiB per rank and proof)" %
      (k, rank, world, " (grouped)" if groups else "", (time.perf_counter() - t) / reps * 1e3, (9 * 49216 * (world - 1) + 4 * n * 32 * (world - 1) // world) / 2**20))
