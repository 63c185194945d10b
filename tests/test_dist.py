"""world_size-2 rehearsal of the sharded-commit exchange.  On CPU (gloo): point-range split,
all-gather of per-rank records through the same callback object the C library calls, and the
host-side combine (plonk_g1_sum); the per-rank partial MSMs come from the oracle there (no GPU).
On MI355X: plonk_prove_3wire_sharded itself, two ranks on one GPU with the callback communicator,
one rank with the RCCL communicator, and bench.py started with --gpus 2."""
import os
import subprocess
import sys

from conftest import ROOT

WORKER = r'''
import ctypes as C, os, sys
sys.path.insert(0, %(root)r)
import torch.distributed as dist
from plonk_amd import sharded, _lib
from plonk_amd.api import _g1_struct, _g1_value
from oracle import curve, protocol as P
import random
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
comm = sharded.HostAllGather(dist, rank, world)
n_points = 37
srs = P.Srs(0xD15EA5E, n_points - 1)
rng = random.Random(5)
coeffs = [rng.randrange(1 << 250) for _ in range(n_points - 4)]   # polynomial shorter than the SRS
lo, hi = sharded.point_range(n_points, rank, world)
lo_c, hi_c = min(lo, len(coeffs)), min(hi, len(coeffs))
part = curve.to_affine(curve.msm(srs.g1_powers[lo_c:hi_c], coeffs[lo_c:hi_c]))
# go through the C-callable callback exactly as the library does
mine = _g1_struct(part)
recv = (_lib.G1 * world)()
rc = comm.callback(None, C.cast(C.byref(mine), C.c_void_p), C.sizeof(_lib.G1), C.cast(recv, C.c_void_p))
assert rc == 0 and comm.calls == 1
total = sharded.combine_partials([_g1_value(g) for g in recv])
assert total == P.commit(coeffs, srs), "combined partial commitments differ from the full commitment"
# every rank must hold the same bytes (same transcript on all ranks)
blobs = comm.allgather_bytes(curve.compress(total))
assert blobs[:48] == blobs[48:96]
# an empty slice contributes the identity
assert sharded.combine_partials([None, part, None]) == part
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_point_range_partition():
    from plonk_amd import sharded
    for n in (0, 1, 7, 8, 1048587):
        for world in (1, 2, 3, 8):
            rs = [sharded.point_range(n, r, world) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in rs]
            assert max(sizes) - min(sizes) <= 1


def test_sharded_commit_exchange_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out.decode())
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, out
        assert "rank %d ok" % rank in out


def test_group_ranges_tile_the_points():
    """sharded.group_range: for every batch size b the ranks r = k mod b split [0, n_points) among them in
    rank order, for every k < b (what plonk_prove_3wire_grouped checks on the device side)."""
    from plonk_amd import sharded
    for n_points in (27, 75, 1035, (1 << 20) + 11):
        for world in (2, 3, 4, 5, 8, 16):
            for b in (2, 3):
                if world < b:
                    with pytest.raises(ValueError):
                        sharded.group_range(n_points, 0, world, b)
                    continue
                for k in range(b):
                    at = 0
                    for r in range(k, world, b):
                        lo, hi = sharded.group_range(n_points, r, world, b)
                        assert lo == at and hi >= lo
                        at = hi
                    assert at == n_points
            tiles = [sharded.point_range(n_points, r, world) for r in range(world)]
            assert tiles[0][0] == 0 and tiles[-1][1] == n_points
            assert all(tiles[i][1] == tiles[i + 1][0] for i in range(world - 1))


GPU_WORKER = r'''
import hashlib, json, os, random, sys
sys.path.insert(0, %(root)r)
sys.path.insert(0, os.path.join(%(root)r, "tests"))
import torch.distributed as dist
import plonk_amd as pa
from plonk_amd import circuits, sharded
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
case = os.environ["PLONK_TEST_CASE"]
kind = os.environ.get("PLONK_TEST_KIND", "callback")
# callback: all ranks share the one GPU of the test box; rccl: a GPU per rank, all-gathers over xGMI
ctx = pa.Context(rank if kind == "rccl" else 0)
comm = sharded.Communicator(ctx, dist, rank, world, kind=kind)
assert comm.kind == kind
pis = []
if case.startswith("pi"):
    # public-input gates feeding a mul/add chain: the PI polynomial takes part in the sharded round 3
    from oracle import protocol as P, fft as offt
    from test_gpu_parity import _public_input_circuit, _oracle_case
    cs, pis = _public_input_circuit(int(case[2:]))
    n = cs.n
    rng = random.Random(7)
    blinding = [rng.randrange(pa.api.R_MOD) for _ in range(11)]
    tau = 0xC0FFEE1234567
    _, _, _, perm, proof_o, _ = _oracle_case(cs, tau, blinding, pis)
    want = P.proof_to_bytes(proof_o)
    wires, selectors = cs.wire_values(), cs.selectors()
    sigmas, k1, k2, positions = [perm.s_sigma1_evals, perm.s_sigma2_evals, perm.s_sigma3_evals], perm.k1, perm.k2, cs.public_input_positions
else:
    k = int(case)
    circ = circuits.SyntheticCircuit(k)
    n, tau, blinding = circ.n, circ.tau, circ.blinding
    wires, selectors, sigmas, k1, k2, positions = circ.wires, circ.selectors, circ.sigmas, circ.k1, circ.k2, ()
    es = [e for e in json.load(open(os.path.join(%(root)r, "tests", "golden", "proofs.json"))) if e["name"] == "synthetic_2^%%d" %% k]
    want = bytes.fromhex(es[0]["proof"])   # the oracle's bytes
lo, hi = sharded.point_range(n + 11, rank, world)
srs = pa.Srs.from_tau_range(tau, lo, hi, ctx=ctx)
assert len(srs) == hi - lo
pk = pa.ProvingKey(n, selectors, sigmas, k1, k2, positions, ctx=ctx)
groups = {}
if os.environ.get("PLONK_TEST_GROUPED") == "1":
    # the commitments of the rounds with two / three polynomials placed on groups of ranks
    for b in (2, 3):
        if world >= b:
            glo, ghi = sharded.group_range(n + 11, rank, world, b)
            groups[b] = (pa.Srs.from_tau_range(tau, glo, ghi, ctx=ctx), (glo, ghi))
    assert groups
prover = pa.Prover(wires, pk, srs, resident=True, comm=comm, shard=(lo, hi), groups=groups)
proof = prover.prove(pis, blinding)
calls, nbytes = comm.stats()
# the coverage check, one all-gather per commitment MSM, the partial sums of rounds 4 and 5 (256 + 32 bytes) and --
# when round 3 is sharded by evaluation point (a power-of-two number of ranks) -- the slices of the quotient polynomial
point_sharded = world & (world - 1) == 0
assert calls == 1 + 9 + 2 + (1 if point_sharded else 0), calls
assert comm.host is None or comm.host.calls == calls
if point_sharded:
    assert nbytes == 48 + 9 * (64 + 256 * 192) + 256 + 32 + (4 * n // world) * 32, nbytes
assert proof.to_bytes() == want, "sharded proof differs from the oracle's"
assert prover.prove(pis, blinding).to_bytes() == proof.to_bytes()   # and again on warm buffers
if case == "pi9" and world == 2:
    # a key destroyed and another made for a different circuit of the same size (it may well get the old key's
    # address): the rank's cached slices of the key's coset tables must not outlive the key
    from oracle import protocol as P
    from test_gpu_parity import _oracle_case
    prover.close(); pk.close()
    cs2, bl2 = P.synthetic_circuit(9)
    assert cs2.n == n
    _, _, _, perm2, proof2_o, _ = _oracle_case(cs2, tau, bl2, [])
    pk2 = pa.ProvingKey(n, cs2.selectors(), [perm2.s_sigma1_evals, perm2.s_sigma2_evals, perm2.s_sigma3_evals], perm2.k1, perm2.k2, (), ctx=ctx)
    got2 = pa.Prover(cs2.wire_values(), pk2, srs, resident=True, comm=comm, shard=(lo, hi)).prove([], bl2)
    assert got2.to_bytes() == P.proof_to_bytes(proof2_o), "proof with the second key of a context differs from the oracle's"
if case == "16" and world == 2:
    # slices that do not tile the point range are refused on every rank (the single-device prover
    # returns PLONK_E_INVALID for a polynomial past the SRS; a gap would silently drop coefficients)
    bad = pa.Srs.from_tau_range(tau, lo + (1 if rank else 0), hi, ctx=ctx)
    try:
        pa.Prover(wires, pk, bad, resident=True, comm=comm, shard=(lo + (1 if rank else 0), hi)).prove([], blinding)
        raise SystemExit("a gap in the point range was accepted")
    except pa.PlonkError as e:
        assert e.code == pa._lib.PLONK_E_INVALID and "tile" in str(e), e
    # ... and so is an SRS whose ranks together hold fewer points than the blinded polynomials need
    lo2, hi2 = sharded.point_range(n + 1, rank, world)
    short = pa.Srs.from_tau_range(tau, lo2, hi2, ctx=ctx)
    try:
        pa.Prover(wires, pk, short, resident=True, comm=comm, shard=(lo2, hi2)).prove([], blinding)
        raise SystemExit("a too-short sharded SRS was accepted")
    except pa.PlonkError as e:
        assert e.code == pa._lib.PLONK_E_INVALID, e
if groups and case == "14" and world == 4:
    # group slices that leave a gap among the ranks of one commitment are refused
    b = 3
    glo, ghi = sharded.group_range(n + 11, rank, world, b)
    shift = 1 if rank == 3 else 0   # rank 3 is the second member of commitment 0's group
    bad = dict(groups)
    bad[b] = (pa.Srs.from_tau_range(tau, glo + shift, ghi, ctx=ctx), (glo + shift, ghi))
    try:
        pa.Prover(wires, pk, srs, resident=True, comm=comm, shard=(lo, hi), groups=bad).prove(pis, blinding)
        raise SystemExit("a gap in a group's point range was accepted")
    except pa.PlonkError as e:
        assert e.code == pa._lib.PLONK_E_INVALID and "group" in str(e), e
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
'''


import pytest


@pytest.mark.gpu
@pytest.mark.parametrize("world,case", [(2, "16"), (2, "20"), (4, "10"), (8, "14"), (3, "10"), (2, "pi9"), (8, "pi6"),
                                        (2, "16g"), (3, "10g"), (4, "14g"), (8, "16g"), (8, "pi6g")])
def test_sharded_prove_ranks_on_one_gpu(tmp_path, world, case):
    """plonk_prove_3wire_sharded with several ranks on cuda:0 (gloo + the callback communicator for the
    exchange): every rank holds an SRS slice with its own window table; the ranks' shares of every MSM
    are all-gathered; with a power-of-two number of ranks round 3 is sharded by evaluation point
    (transforms of size 4n / world on the rank's residue class of the coset, quotient slices
    all-gathered and recombined -- with 8 ranks z(wX) lives on another rank's class).  All ranks emit
    the oracle's proof: golden bytes (2^10 / 2^14 / 2^16 and the 2^20-gate benchmark configuration),
    the oracle's bytes for circuits with public inputs.  Cases ending in "g": the rounds
    that commit two / three polynomials place each commitment on the ranks r = k mod 2 / 3
    (plonk_prove_3wire_grouped), every rank bringing a slice per placement."""
    grouped = case.endswith("g")
    case = case.rstrip("g")
    script = tmp_path / "gpu_worker.py"
    script.write_text(GPU_WORKER % {"root": ROOT})
    procs = []
    port = 29600 + world * 32 + sum(map(ord, case)) % 16 + (16 if grouped else 0)
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   PLONK_TEST_CASE=case, PLONK_TEST_GROUPED="1" if grouped else "0")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=280)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out.decode())
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, out[-3000:]
        assert "rank %d ok" % rank in out


@pytest.mark.gpu
def test_rccl_communicator_single_rank(ctx):
    """The RCCL communicator with one rank (all this box can host: RCCL wants a GPU per rank):
    librccl is bound at run time, ncclCommInitRank / ncclAllGather run on the context's stream, and the
    sharded prover -- same code path as with more ranks -- returns the golden 2^16 proof."""
    import json
    import plonk_amd as pa
    from plonk_amd import circuits, sharded
    k = 16
    circ = circuits.SyntheticCircuit(k)
    n = circ.n
    comm = sharded.Communicator(ctx, None, 0, 1, kind="rccl")
    srs = pa.Srs.from_tau_range(circ.tau, 0, n + 11, ctx=ctx)
    pk = pa.ProvingKey(n, circ.selectors, circ.sigmas, circ.k1, circ.k2, (), ctx=ctx)
    proof = pa.Prover(circ.wires, pk, srs, resident=True, comm=comm, shard=(0, n + 11)).prove([], circ.blinding)
    want = [e for e in json.load(open(os.path.join(ROOT, "tests", "golden", "proofs.json"))) if e["name"] == "synthetic_2^16"][0]
    assert proof.to_bytes().hex() == want["proof"]
    calls, nbytes = comm.stats()
    assert calls == 10 and nbytes == 48 + 9 * (64 + 256 * 192)   # one rank: round 3 is not sharded
    comm.close()


@pytest.mark.gpu
def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` starts two ranks by itself (both on this box's one GPU, exchange staged
    through gloo) and reports n_gpus = 2 and the same proof as the one-rank run."""
    import json

    def run(gpus):
        env = dict(os.environ, PLONK_BENCH_BACKEND="gloo")
        env.pop("WORLD_SIZE", None)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--log-n", "14", "--steps", "2",
                              "--warmup", "1", "--no-sweep", "--no-cpu-baseline"], env=env, stdout=subprocess.PIPE,
                             stderr=subprocess.PIPE, timeout=280)
        assert out.returncode == 0, out.stderr.decode()[-2000:]
        lines = [ln for ln in out.stdout.decode().splitlines() if ln.startswith("{")]
        assert len(lines) == 1
        return json.loads(lines[0])

    one, two = run(1), run(2)
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert two["config"]["proof_sha256_16"] == one["config"]["proof_sha256_16"]
    assert two["steps"] == 2 and two["value"] > 0
    ex = two["exchange"]   # event marks around the all-gathers: 9 MSM shares + coverage check + quotient slices + the sums of rounds 4, 5
    assert ex["kind"] == "callback" and ex["all_gathers_per_proof"] == 13 and ex["stream_ms_per_proof_rank0"] > 0
    assert "exchange" not in one
    # asking for more ranks than WORLD_SIZE provides is an error, not a silent one-rank run
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, timeout=120)
    assert bad.returncode != 0 and b"WORLD_SIZE" in bad.stderr
