"""Times the in-place coset NTT of 2^k resident elements (development aid; run under rocprofv3)."""
import ctypes as C
import sys
import time

import numpy as np

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plonk_amd as pa

k = int(sys.argv[1]) if len(sys.argv) > 1 else 22
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
flags = int(sys.argv[3]) if len(sys.argv) > 3 else 2
n = 1 << k
ctx = pa.Context()
rng = np.random.default_rng(k)
x = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
dev = C.c_void_p()
L = pa._lib.lib
ctx.check(L.plonk_dev_alloc(ctx.h, n * 32, C.byref(dev)))
ctx.check(L.plonk_dev_upload(ctx.h, dev, x.ctypes.data_as(C.c_void_p), n * 32))
ctx.check(L.plonk_ntt_dev(ctx.h, dev, k, flags, None, 1, n))
y = np.empty_like(x)
ctx.check(L.plonk_dev_download(ctx.h, y.ctypes.data_as(C.c_void_p), dev, n * 32))
import hashlib
digest = hashlib.sha256(y.tobytes()).hexdigest()[:12]
for i in range(reps + 3):
    if i == 3:
        ctx.check(L.plonk_sync(ctx.h))
        t0 = time.perf_counter()
    ctx.check(L.plonk_ntt_dev(ctx.h, dev, k, flags, None, 1, n))
ctx.check(L.plonk_sync(ctx.h))
dt = (time.perf_counter() - t0) / reps
print("ntt 2^%d flags=%d: %.4f ms  (%.1f GB/s algorithmic)  first output sha %s" % (k, flags, dt * 1e3, 2 * n * 32 / dt / 1e9, digest))
