"""Times commit() of 2^k random scalars against a resident SRS (development aid; run under rocprofv3)."""
import ctypes as C
import sys
import time

import numpy as np

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plonk_amd as pa

k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
n = 1 << k
ctx = pa.Context()
srs = pa.Srs.from_tau(0x1234567, n - 1)
rng = np.random.default_rng(k)
sc = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
if len(sys.argv) > 3 and sys.argv[3] == "skew":   # selector-like vector: 0 / 1 / -1 / 2
    R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
    sc = pa.fr_to_array([0, 1, R - 1, 2])[rng.integers(0, 4, size=n)]
dev = C.c_void_p()
ctx.check(pa._lib.lib.plonk_dev_alloc(ctx.h, n * 32, C.byref(dev)))
ctx.check(pa._lib.lib.plonk_dev_upload(ctx.h, dev, sc.ctypes.data_as(C.c_void_p), n * 32))
out = pa._lib.G1()
for i in range(reps + 2):
    if i == 2:
        t0 = time.perf_counter()
    ctx.check(pa._lib.lib.plonk_commit_dev(ctx.h, srs.h, dev, n, C.byref(out)))
dt = (time.perf_counter() - t0) / reps
print("msm 2^%d: %.3f ms  (%.1f Mpoints/s)" % (k, dt * 1e3, n / dt / 1e6))
