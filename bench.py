"""bench.py -- prover wall time at 2^k gates on N MI355X, with NTT / MSM rates and rooflines.

One step = one proof (plonk_prove_3wire_dev) of the synthetic 2^k-gate circuit, witness resident
in HBM.  N > 1: one process per GPU; every rank holds the full key, the nine commitment MSMs are
sharded by point range and the ranks' shares all-gathered on the device by the library's RCCL
communicator (torch.distributed/gloo only carries the RCCL id, the barriers and the max over ranks).
`python bench.py --gpus N` starts the N ranks itself; under torch.distributed.run (WORLD_SIZE set)
the process is one of the ranks.  Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12


def pmc_traffic(kernel, run_name):
    """HBM bytes per launch of `kernel` from this round's committed rocprofv3 counter passes (FETCH_SIZE
    and WRITE_SIZE collected in separate runs and corrected as profiles/README.md states), with the
    file it came from; (None, None) when there is no measurement of this workload.  Not measured by
    this run: counters need the profiler."""
    for name in ("r3_pmc_traffic.json", "r2_pmc_traffic.json"):   # the newest round that measured this kernel
        try:
            table = json.load(open(os.path.join(ROOT, "profiles", name)))
            row = table[kernel]
            if row["run"] == run_name:
                return row["hbm_bytes_per_launch"], "profiles/" + name
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def cpu_baseline(log_n, srs_points):
    """The C restatement of the reference's loops (oracle/cref) timed on the host cores over the prover's
    per-proof operation list -- 5 iNTT(n) + 16 NTT(4n) + quotient(4n) + 9 MSM(n) (prover.rs:181-837).
    (i) All cores, at the full size n = 2^log_n: the list pulled from a queue by one thread per core, timed
    directly (the stand-in for a rayon-parallel CPU prover).  (ii) One thread, as backend_new runs: a bounded
    sample at m = 2^16 scaled to n by operation counts (N log N per transform, Pippenger's additions per MSM,
    linear for the quotient loop) -- the full size on one thread would take minutes.
    `srs_points`: (n, 12) uint64 affine points (Montgomery x | y)."""
    import math
    import queue
    import threading
    import numpy as np
    from oracle import cref
    try:
        cores = min(16, len(os.sched_getaffinity(0)))
    except AttributeError:
        cores = min(16, os.cpu_count() or 1)

    def op_list(lg, pts, quot_slices):
        m = 1 << lg
        rng = np.random.default_rng(1)
        small = [rng.integers(0, 1 << 62, size=(m, 4), dtype=np.uint64) for _ in range(5)]
        big = [rng.integers(0, 1 << 62, size=(4 * m, 4), dtype=np.uint64) for _ in range(16)]
        ext = np.stack(big[:15])
        scal = [rng.integers(0, 1 << 62, size=(m, 4), dtype=np.uint64) for _ in range(9)]
        qs = 4 * m // quot_slices
        return [("msm", (lambda s=s: cref.msm(pts, s))) for s in scal] + \
               [("quot", (lambda i=i: cref.quotient(ext, m, 3, 5, 7, 6, 7, i * qs, qs))) for i in range(quot_slices)] + \
               [("ntt4", (lambda a=a: cref.ntt(a, 2))) for a in big] + \
               [("ntt1", (lambda a=a: cref.ntt(a, 1))) for a in small]

    # (ii) one thread, bounded sample
    lm = min(log_n, 16)
    m = 1 << lm
    ops = op_list(lm, np.ascontiguousarray(srs_points[:m]), 16)
    per_class = {}
    t_all = time.perf_counter()
    for name, fn in ops:
        t = time.perf_counter()
        fn()
        per_class[name] = per_class.get(name, 0.0) + time.perf_counter() - t
    serial_s = time.perf_counter() - t_all
    n = 1 << log_n
    c_of = lambda k: 3 if k < 32 else int(math.log(k)) + 2
    msm_work = lambda k: k * math.ceil(255 / c_of(k)) + 2 * (1 << c_of(k)) * math.ceil(255 / c_of(k))
    ntt_work = lambda lg: (1 << lg) * lg
    total = (per_class["msm"] * msm_work(n) / msm_work(m) + per_class["ntt4"] * ntt_work(log_n + 2) / ntt_work(lm + 2) +
             per_class["ntt1"] * ntt_work(log_n) / ntt_work(lm) + per_class["quot"] * n / m)

    # (i) all cores, full size, timed directly (the longest operations first so the tail is short)
    full_ops = op_list(log_n, np.ascontiguousarray(srs_points[:n]), 4 * cores)
    work = queue.Queue()
    for item in full_ops:
        work.put(item)

    def pull():
        while True:
            try:
                _, fn = work.get_nowait()
            except queue.Empty:
                return
            fn()   # ctypes releases the GIL for the duration of the C call

    th = [threading.Thread(target=pull) for _ in range(cores)]
    t = time.perf_counter()
    for x in th:
        x.start()
    for x in th:
        x.join()
    threaded_s = time.perf_counter() - t
    return {"value": round(threaded_s * 1e3, 1), "unit": "ms", "cores": cores, "kind": "port", "extrapolated": False,
            "sample": "oracle/cref (C restatement of backend_new fft.rs / prover.rs:415-457 / ark msm_bigint): the whole per-proof "
                      "operation list at n=2^%d -- 9 MSM(n) on the SRS points, 16 coset NTT(4n), 5 iNTT(n), quotient(4n) in %d slices -- "
                      "pulled from a queue by %d threads, timed directly" % (log_n, 4 * cores, cores),
            "one_thread": {"value": round(total * 1e3, 1), "unit": "ms", "cores": 1, "extrapolated": True,
                           "how": "the same list at m=2^%d on one thread (9 MSM %.2fs, 16 coset NTT(4m) %.2fs, 5 iNTT(m) %.2fs, "
                                  "quotient(4m) %.2fs = %.2fs) scaled to n=2^%d by operation counts" %
                                  (lm, per_class["msm"], per_class["ntt4"], per_class["ntt1"], per_class["quot"], serial_s, log_n)},
            "ntt_gbps": round(64.0 * 4 * m * 16 / per_class["ntt4"] / 1e9, 3),
            "msm_mpoints_per_s": round(9 * m / per_class["msm"] / 1e6, 4),
            "note": "the restatement is plain C, several times slower than tuned arkworks assembly would be: a stated baseline, not a target"}


def launch_ranks(n_ranks):
    """Start one fresh child process per rank (this process has not touched a GPU and does not), wait
    for all of them, print rank 0's JSON line.  Any failing rank fails the run."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for rank in range(n_ranks):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if rank == 0 else subprocess.DEVNULL))
    import threading
    box = {}
    reader = threading.Thread(target=lambda: box.setdefault("out", procs[0].stdout.read()))
    reader.start()
    while any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs):   # a rank died: the others would wait for it in a collective
            time.sleep(2.0)
            for p in procs:
                if p.poll() is None:
                    p.kill()
            break
        time.sleep(0.05)
    codes = [p.wait() for p in procs]
    reader.join()
    out0 = box.get("out", b"")
    lines = [ln for ln in out0.decode().splitlines() if ln.startswith("{")]
    if any(codes) or not lines:
        sys.stderr.write("bench.py: rank exit codes %r; rank 0 printed %d JSON line(s)\n" % (codes, len(lines)))
        sys.exit(1)
    print(lines[-1], flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log-n", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sweep", action="store_true", help="skip the NTT / MSM rates at the other sizes")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args.gpus)   # before anything in this process touches a GPU

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d: start it as `python bench.py --gpus N`, or under "
                 "torch.distributed.run with --nproc-per-node equal to --gpus" % (args.gpus, world))
    import numpy as np
    import torch
    dist = None
    # PLONK_BENCH_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks: the ranks
    # share the devices and the library stages its all-gathers through the host (RCCL wants a GPU per rank)
    backend = os.environ.get("PLONK_BENCH_BACKEND", "nccl")
    if backend != "nccl" or os.environ.get("PLONK_BENCH_SHARE_GPUS"):
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo")   # control plane only: RCCL id, barriers, max over ranks
    import plonk_amd as pa
    from plonk_amd import circuits, sharded

    ctx = pa.Context(local_rank)
    k = args.log_n
    n = 1 << k
    t0 = time.time()
    circ = circuits.SyntheticCircuit(k)
    comm, exchange_note = None, None
    if world > 1:
        kind = "rccl" if backend == "nccl" else "callback"
        if kind == "rccl":
            # RCCL over xGMI; if it cannot be brought up on ANY rank, all ranks agree (over gloo) to stage the
            # all-gathers through the host instead, and the JSON line says so
            try:
                comm = sharded.Communicator(ctx, dist, rank, world, kind="rccl")
                failure = ""
            except Exception as e:   # noqa: BLE001 -- reported below
                failure = "%s: %s" % (type(e).__name__, e)
            ok = torch.tensor([0 if failure else 1], dtype=torch.int32)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                notes = [None] * world
                dist.all_gather_object(notes, failure)
                exchange_note = "RCCL unavailable (%s): all-gathers staged through the host over gloo" % \
                    "; ".join(sorted({x for x in notes if x}))
                if comm is not None:
                    comm.close()
                comm = None
                kind = "callback"
        if comm is None:
            comm = sharded.Communicator(ctx, dist, rank, world, kind=kind)
    lo, hi = sharded.point_range(n + 11, rank, world)
    srs = pa.Srs.from_tau(circ.tau, n + 10, ctx=ctx) if world == 1 else pa.Srs.from_tau_range(circ.tau, lo, hi, ctx=ctx)
    pk = pa.ProvingKey(n, circ.selectors, circ.sigmas, circ.k1, circ.k2, (), ctx=ctx)
    # up to 2^21 gates a rank's share of a commitment is latency-bound: the rounds that commit two / three
    # polynomials then place each commitment on a group of ranks (one larger MSM per rank instead of two / three small ones)
    groups = {}
    if world > 1 and k <= 21:
        for b in (2, 3):
            if world >= b:
                glo, ghi = sharded.group_range(n + 11, rank, world, b)
                groups[b] = (pa.Srs.from_tau_range(circ.tau, glo, ghi, ctx=ctx), (glo, ghi))
    prover = pa.Prover(circ.wires, pk, srs, resident=True, comm=comm, shard=(lo, hi), groups=groups)
    setup_s = time.time() - t0

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.sync()

    proof = None
    for _ in range(args.warmup):
        proof = prover.prove([], circ.blinding)
    barrier()
    if comm is not None:
        comm.timing(True)   # event marks around this rank's all-gathers from here on
    t = time.perf_counter()
    for _ in range(args.steps):
        proof = prover.prove([], circ.blinding)
    barrier()
    dt = time.perf_counter() - t
    exchange = None
    if comm is not None:
        calls, nbytes = comm.stats()
        per_proof = args.steps + args.warmup
        exchange = {"kind": comm.kind, "all_gathers_per_proof": calls // per_proof,
                    "mib_contributed_per_proof": round(nbytes / per_proof / 2**20, 2),
                    "stream_ms_per_proof_rank0": round(comm.timing(False) / args.steps, 3),
                    "note": "time between event marks around each all-gather on rank 0's streams: the transfer and the wait for the slowest rank"}
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms = dt / args.steps * 1e3
    digest = __import__("hashlib").sha256(proof.to_bytes()).hexdigest()[:16]
    # the same proof from host witness vectors (3 n 32 bytes of pageable memory cross PCIe inside the call), over
    # the same number of steps: reported beside `value`, never as `value`
    host_ms = None
    if world == 1:
        hp = pa.Prover(circ.wires, pk, srs, resident=False)
        for _ in range(max(1, min(args.warmup, 2))):
            hp.prove([], circ.blinding)
        barrier()
        t = time.perf_counter()
        for _ in range(args.steps):
            hproof = hp.prove([], circ.blinding)
        barrier()
        host_ms = (time.perf_counter() - t) / args.steps * 1e3
        assert hproof.to_bytes() == proof.to_bytes()

    # one more proof with the event marks on, on every rank (a sharded proof is a collective): the quotient kernel alone
    ctx.msm_set_timing(True)
    prover.prove([], circ.blinding)
    quot_ms = ctx.quotient_last_timing()
    ctx.msm_set_timing(False)

    out = None
    if rank == 0:
        # ---- stand-alone kernel rates on this GPU (after the timed region) ----
        log4 = k + 2
        n4 = 1 << log4
        buf = ctx.alloc(n4 * 32)
        ctx.upload(buf, np.random.default_rng(0).integers(0, 1 << 62, size=(n4, 4), dtype=np.uint64))
        reps = 10
        ctx.ntt_dev(buf, log4, pa._lib.NTT_COSET)
        ctx.timer_start()
        for _ in range(reps):
            ctx.ntt_dev(buf, log4, pa._lib.NTT_COSET)
        ntt_ms = ctx.timer_stop() / reps
        ctx.free(buf)
        sweeps = pa._lib.lib.plonk_ntt_sweeps(log4)
        ntt_bytes = 64.0 * n4
        ntt_rate = ntt_bytes / (ntt_ms * 1e-3)
        full = pa.Srs.from_tau(circ.tau, n - 1, ctx=ctx) if world > 1 else srs
        sc = ctx.alloc(n * 32)
        ctx.upload(sc, np.random.default_rng(2).integers(0, 1 << 62, size=(n, 4), dtype=np.uint64))
        full.commit_dev(sc, n)
        t1 = time.perf_counter()
        phases = []
        ctx.msm_set_timing(True)
        for _ in range(5):
            full.commit_dev(sc, n)
            phases.append(ctx.msm_last_timing())   # HIP events on the streams the MSM kernels run on
        msm_s = (time.perf_counter() - t1) / 5
        ctx.free(sc)
        ctx.msm_set_timing(False)
        fq_peak = ctx.bench_field_mul(1)
        fr_peak = ctx.bench_field_mul(0)
        mad_rate = ctx.bench_field_mul(4)          # v_mad_u64_u32 lane-operations per second
        tm = {key: sum(p[key] for p in phases) / len(phases) for key in ("sort_ms", "big_bucket_ms", "bucket_ms", "reduce_ms")}
        entries, c, sets = phases[-1]["entries"], phases[-1]["window_bits"], phases[-1]["bucket_sets"]
        buckets = sets << (c - 1)
        # work model: one mixed XYZZ addition (8M + 2S) per entry, two full additions (12M + 2S) per bucket
        fq_muls = entries * 10 + 2 * buckets * 14
        bucket_muls = entries * 10
        # msm_bucket_kernel per launch: 128-byte point + 4-byte entry read per addition, 256-byte sum written per bucket
        bucket_bytes = entries * 132.0 + buckets * 256.0 + buckets * 12.0
        bucket_rate = bucket_bytes / (tm["bucket_ms"] * 1e-3)
        bucket_mul_rate = bucket_muls / (tm["bucket_ms"] * 1e-3)
        # the instruction-rate bound: every issue slot a v_mad_u64_u32, a 29-bit-limb product of 14 (9) limbs
        # being 2 * 14^2 = 392 (2 * 9^2 = 162) of them
        fq_instr_bound, fr_instr_bound = mad_rate / 392.0, mad_rate / 162.0
        traffic, traffic_src = pmc_traffic("msm_bucket_kernel", "msm") if (k == 20 and c == 20) else (None, None)
        ntt_traffic, ntt_traffic_src = pmc_traffic("ntt_pass_kernel", "ntt") if log4 == 22 else (None, None)
        butterflies = (1 << (log4 - 1)) * log4
        quot_bytes = 480.0 * n4                    # 14 input streams + 1 output per point (z(wX) is a rotated read of z)
        # products per point in quotient_kernel: x 1, gate 4, beta x / beta k x 3, beta sigma 3, the two triple products 6,
        # alpha 1, L1 term 2, 1/Z_H 1
        quot_muls = 21.0 * n4
        # multiply-add slots: a mixed addition issues 3 542 v_mad_u64_u32 per lane (7 products of 2 x 196, two squarings of
        # 105 + 196, one two-product reduction of 3 x 196), the kernel's first addition of a bucket (affine + affine) 2 366
        mads = entries * 3542.0
        # whole proof: what its kernels' field products would take at the measured product ceilings, with nothing else
        # (no sort, no latency, no launch gaps) -- 9 MSM bucket passes and reductions (Fq), 4 iNTT(n) + 4 NTT(4n) +
        # 1 iNTT(4n) butterflies, the quotient pass, the grand product (17 per gate) and the round 4-5 passes (~30 per gate)
        floor_fq = 9.0 * fq_muls / fq_peak
        floor_fr = (4.0 * (n // 2) * k + 5.0 * (n4 // 2) * log4 + quot_muls + 47.0 * n) / fr_peak
        out = {
            "metric": "prover_wall_time_ms_2^%d_gates" % k, "value": round(ms, 3), "unit": "ms",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": False, "scaling": "strong", "vs_baseline": None,
            "dtype": "u32 (Fr / Fq Montgomery residues; multiplied as 9 / 14 limbs of 29 bits in 64-bit accumulators)",
            "data": "synthetic",
            "config": {"workload": "synthetic 2^%d-gate mul/add chain, 3-wire PLONK prove, witness "
                                   "resident in HBM, SRS/proving key preloaded" % k,
                       "parallelism": "1 GPU" if world == 1 else "MSMs sharded by point range%s and round 3 by evaluation point over %d GPUs, shares all-gathered %s" % (
                                          " (commitments of a round placed on groups of ranks)" if groups else "",
                                          world, "on the device by RCCL" if comm.kind == "rccl" else "through the host (gloo)"),
                       "setup_seconds": round(setup_s, 1), "proof_sha256_16": digest,
                       **({"exchange_note": exchange_note} if exchange_note else {})},
            **({"exchange": exchange} if exchange else {}),
            "ntt_gbps": round(ntt_rate / 1e9, 2), "ntt_ms_coset_2^%d" % log4: round(ntt_ms, 4),
            "msm_mpoints_per_s": round(n / msm_s / 1e6, 3), "msm_ms_2^%d" % k: round(msm_s * 1e3, 3),
            # dominant kernel of the proof: the MSM bucket accumulation (over half of the GPU time).  Its bound is the
            # integer VALU: field products per second against the product ceiling measured on this device.
            "roofline": {"kernel": "msm_bucket_kernel (2^%d points, c=%d, %d bucket set(s))" % (k, c, sets), "bound": "valu",
                         "achieved": round(bucket_mul_rate / 1e9, 2), "peak": round(fq_peak / 1e9, 2), "unit": "G Fq-mul/s",
                         "frac": round(bucket_mul_rate / fq_peak, 4),
                         "peak_is": "plonk_bench_field_mul: dependency-free 381-bit Montgomery products of this library's own multiplier, measured in this run",
                         "instruction_rate_bound": round(fq_instr_bound / 1e9, 2),
                         "frac_of_instruction_rate_bound": round(bucket_mul_rate / fq_instr_bound, 4),
                         "instruction_rate_bound_is": "v_mad_u64_u32 issue rate measured in this run (%.2f T lane-op/s) / 392 per product" % (mad_rate / 1e12),
                         "mad_slot_frac": round(mads / (tm["bucket_ms"] * 1e-3) / mad_rate, 4),
                         "mad_slot_frac_is": "v_mad_u64_u32 lane-operations the kernel must issue (3 542 per mixed addition x additions) / "
                                             "launch time / the measured v_mad_u64_u32 issue rate: no model of a product's cost involved",
                         "work_per_launch": {"mixed_additions": entries, "fq_mul": bucket_muls, "v_mad_u64_u32": mads},
                         "avg_launch_ms": round(tm["bucket_ms"], 4),
                         "traffic": traffic, "traffic_source": traffic_src,
                         "hbm": {"bound": "hbm", "achieved": round(bucket_rate / 1e9, 2), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                                 "frac": round(bucket_rate / HBM_PEAK, 4), "algorithmic_bytes_per_launch": bucket_bytes}},
            "roofline_msm": {"kernel": "whole MSM: sort + bucket accumulation + reduction", "bound": "valu",
                             "achieved": round(fq_muls / msm_s / 1e9, 2), "peak": round(fq_peak / 1e9, 2),
                             "unit": "G Fq-mul/s", "frac": round(fq_muls / msm_s / fq_peak, 4),
                             "phases_ms": {key: round(v, 4) for key, v in tm.items()}},
            "roofline_ntt": {"kernel": "ntt_pass_kernel (coset NTT 2^%d = %d sweeps)" % (log4, sweeps), "bound": "valu",
                             "achieved": round(butterflies / (ntt_ms * 1e-3) / 1e9, 2), "peak": round(fr_peak / 1e9, 2),
                             "unit": "G Fr-mul/s (butterfly products, (N/2) log2 N per transform)",
                             "frac": round(butterflies / (ntt_ms * 1e-3) / fr_peak, 4),
                             "instruction_rate_bound": round(fr_instr_bound / 1e9, 2),
                             "avg_launch_ms": round(ntt_ms / sweeps, 4),
                             "traffic": ntt_traffic, "traffic_source": ntt_traffic_src,
                             # a 255-bit butterfly is one Fr product per 64 bytes moved: at the measured product ceiling
                             # the transform cannot pass ~11 % of the HBM rate, so the product rate is the bound that bites
                             "hbm": {"bound": "hbm", "achieved": round(ntt_rate / 1e9, 2), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                                     "frac": round(ntt_rate / HBM_PEAK, 4), "algorithmic_bytes_per_launch": ntt_bytes / sweeps}},
            "roofline_quotient": {"kernel": "quotient_kernel (2^%d coset points)" % log4, "bound": "valu",
                                  "achieved": round(quot_muls / (quot_ms * 1e-3) / 1e9, 2), "peak": round(fr_peak / 1e9, 2),
                                  "unit": "G Fr-mul/s (21 per point)", "frac": round(quot_muls / (quot_ms * 1e-3) / fr_peak, 4),
                                  "avg_launch_ms": round(quot_ms, 4),
                                  "hbm": {"bound": "hbm", "achieved": round(quot_bytes / (quot_ms * 1e-3) / 1e9, 2),
                                          "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": round(quot_bytes / (quot_ms * 1e-3) / HBM_PEAK, 4),
                                          "algorithmic_bytes_per_launch": quot_bytes}},
            "roofline_proof": {"kernel": "whole proof (all kernels of one plonk_prove_3wire_dev call)", "bound": "valu",
                               "achieved": round((floor_fq + floor_fr) * 1e3, 3), "peak": round(ms, 3), "unit": "ms",
                               "frac": round((floor_fq + floor_fr) * 1e3 / ms, 4),
                               "is": "time the proof's field products would take at the product ceilings measured in this run "
                                     "(Fq %.2f ms: 9 x (bucket additions + bucket-set reduction); Fr %.2f ms: transforms, quotient, "
                                     "rounds 2, 4, 5) over the measured time per proof" % (floor_fq * 1e3, floor_fr * 1e3)},
            "field_mul_peaks": {"fr_G_per_s": round(fr_peak / 1e9, 2), "fq_G_per_s": round(fq_peak / 1e9, 2),
                                "v_mad_u64_u32_T_lane_op_per_s": round(mad_rate / 1e12, 2)},
        }
        out["prover_ms_resident_witness"] = round(ms, 3)
        out["value_is"] = ("plonk_prove_3wire_dev: witness vectors resident in HBM when the timed region starts, proof bytes on the host when "
                           "it ends; prover_ms_host_witness is plonk_prove_3wire on the same input from pageable host memory, same step count")
        if host_ms is not None:
            out["prover_ms_host_witness"] = round(host_ms, 3)
        if not args.no_sweep:
            # stand-alone rates at the other sizes (forward NTT in place; commit of uniform scalars
            # against a resident SRS of the same size)
            ntt_sweep, msm_sweep = {}, {}
            for lg in (16, 20, 22, 24, 26):
                try:
                    b = ctx.alloc((1 << lg) * 32)
                    ctx.upload(b, np.random.default_rng(lg).integers(0, 1 << 62, size=(1 << lg, 4), dtype=np.uint64))
                    ctx.ntt_dev(b, lg, 0)
                    ctx.timer_start()
                    for _ in range(10):
                        ctx.ntt_dev(b, lg, 0)
                    ntt_sweep["2^%d" % lg] = round(64.0 * (1 << lg) / (ctx.timer_stop() / 10 * 1e-3) / 1e9, 1)
                    ctx.free(b)
                except pa.PlonkError as e:
                    ntt_sweep["2^%d" % lg] = "failed: %s" % e
            for lg in (16, 18, 20, 22, 24):
                try:
                    m = 1 << lg
                    s2 = srs if (world == 1 and lg == k) else pa.Srs.from_tau(circ.tau, m - 1, ctx=ctx)
                    b = ctx.alloc(m * 32)
                    ctx.upload(b, np.random.default_rng(lg).integers(0, 1 << 62, size=(m, 4), dtype=np.uint64))
                    s2.commit_dev(b, m)
                    t1 = time.perf_counter()
                    for _ in range(3):
                        s2.commit_dev(b, m)
                    msm_sweep["2^%d" % lg] = round(m / ((time.perf_counter() - t1) / 3) / 1e6, 1)
                    ctx.free(b)
                    if s2 is not srs:
                        s2.close()
                except pa.PlonkError as e:
                    msm_sweep["2^%d" % lg] = "failed: %s" % e
            out["ntt_gbps_by_size"] = ntt_sweep
            out["msm_mpoints_per_s_by_size"] = msm_sweep
        if not args.no_cpu_baseline and world == 1:
            raw = (pa._lib.G1 * n)()
            ctx.check(pa._lib.lib.plonk_srs_download(ctx.h, srs.h, 0, n, raw))
            pts = np.frombuffer(raw, dtype=np.uint64).reshape(n, 13)[:, :12].copy()
            del raw
            out["cpu_baseline"] = cpu_baseline(k, pts)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
