"""Per-phase device time of one MSM of 2^k points (development aid): the event marks bench.py reads."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plonk_amd as pa

k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << k
ctx = pa.Context()
srs = pa.Srs.from_tau(0x1234567, n - 1, ctx=ctx)
sc = ctx.alloc(n * 32)
ctx.upload(sc, np.random.default_rng(2).integers(0, 1 << 62, size=(n, 4), dtype=np.uint64))
srs.commit_dev(sc, n)
ctx.msm_set_timing(True)
rows = []
for _ in range(6):
    srs.commit_dev(sc, n)
    rows.append(ctx.msm_last_timing())
t = rows[-1]
keys = ("sort_ms", "big_bucket_ms", "bucket_ms", "reduce_ms")
print("msm 2^%d: c=%d sets=%d entries=%d  " % (k, t["window_bits"], t["bucket_sets"], t["entries"]) +
      "  ".join("%s %.3f" % (key, sum(r[key] for r in rows) / len(rows)) for key in keys))
