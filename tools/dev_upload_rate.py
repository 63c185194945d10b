"""Host-to-device copy rates on this box (development aid): pageable numpy memory through plonk_dev_upload, and the
same bytes from a pinned buffer."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import plonk_amd as pa
ctx = pa.Context()
n = 1 << 20
a = np.random.default_rng(0).integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
dev = ctx.alloc(n * 32)
for rep in range(5):
    t = time.perf_counter(); ctx.upload(dev, a); dt = time.perf_counter() - t
    print("pageable 32 MiB: %.3f ms  %.1f GB/s" % (dt * 1e3, n * 32 / dt / 1e9))
hip = C.CDLL("libamdhip64.so")
p = C.c_void_p()
assert hip.hipHostMalloc(C.byref(p), C.c_size_t(n * 32), 0) == 0
C.memmove(p, a.ctypes.data, n * 32)
for rep in range(5):
    t = time.perf_counter()
    assert hip.hipMemcpy(dev, p, C.c_size_t(n * 32), 1) == 0
    dt = time.perf_counter() - t
    print("pinned   32 MiB: %.3f ms  %.1f GB/s" % (dt * 1e3, n * 32 / dt / 1e9))
for rep in range(3):
    t = time.perf_counter(); C.memmove(p, a.ctypes.data, n * 32); dt = time.perf_counter() - t
    print("host memcpy 32 MiB (one thread): %.3f ms  %.1f GB/s" % (dt * 1e3, n * 32 / dt / 1e9))
t = time.perf_counter(); rc = hip.hipHostRegister(C.c_void_p(a.ctypes.data), C.c_size_t(n * 32), 0); dt = time.perf_counter() - t
print("hipHostRegister 32 MiB: rc=%d %.3f ms" % (rc, dt * 1e3))
if rc == 0:
    for rep in range(3):
        t = time.perf_counter(); assert hip.hipMemcpy(dev, C.c_void_p(a.ctypes.data), C.c_size_t(n * 32), 1) == 0; dt = time.perf_counter() - t
        print("registered 32 MiB: %.3f ms  %.1f GB/s" % (dt * 1e3, n * 32 / dt / 1e9))
    t = time.perf_counter(); hip.hipHostUnregister(C.c_void_p(a.ctypes.data)); dt = time.perf_counter() - t
    print("hipHostUnregister: %.3f ms" % (dt * 1e3))
