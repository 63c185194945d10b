"""Development check: NTT parity against the oracle + field-mul microbench (direct ctypes)."""
import ctypes as C, os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import fft as offt
from oracle.field import R_MOD, fr_to_mont, fr_from_mont, root_of_unity

lib = C.CDLL(os.path.join(os.path.dirname(__file__), "..", "plonk_amd", "libplonk_hip.so"))
ctx = C.c_void_p()
rc = lib.plonk_ctx_create(0, C.byref(ctx)); assert rc == 0, rc
lib.plonk_last_error.restype = C.c_char_p

def to_arr(vals):
    return np.frombuffer(b"".join(fr_to_mont(v).to_bytes(32, "little") for v in vals), dtype=np.uint64).reshape(-1, 4).copy()
def from_arr(a):
    b = a.tobytes()
    return [fr_from_mont(int.from_bytes(b[i*32:(i+1)*32], "little")) for i in range(len(b)//32)]

rng = random.Random(1)
ok = True
for log_n in [0, 1, 2, 3, 4, 6, 8, 10, 11, 12, 13, 14, 16, 17, 18]:
    n = 1 << log_n
    vals = [rng.randrange(R_MOD) for _ in range(n)]
    w = root_of_unity(n)
    for flags in (0, 1, 2, 3):
        if flags == 0: exp = offt.fft(vals, w, n)
        elif flags == 1: exp = offt.ifft(vals, w); exp = exp + [0]*(n-len(exp))
        elif flags == 2: exp = offt.coset_fft(vals, w, 7, n)
        else: exp = offt.coset_ifft(vals, w, 7)
        if log_n >= 17 and flags in (1,):  # keep runtime down
            continue
        a = to_arr(vals)
        t = time.time()
        rc = lib.plonk_ntt(ctx, a.ctypes.data_as(C.c_void_p), log_n, flags, None, 1, n)
        if rc != 0:
            print("ERR", log_n, flags, lib.plonk_last_error(ctx)); ok = False; continue
        got = from_arr(a)
        good = got == exp
        ok &= good
        print("log_n=%d flags=%d %s (%.3fs)" % (log_n, flags, "OK" if good else "MISMATCH", time.time()-t), flush=True)
        if not good:
            bad = [i for i in range(n) if got[i] != exp[i]]
            print("  first bad", bad[:8], "count", len(bad))
for which in (0, 1):
    v = C.c_double()
    rc = lib.plonk_bench_field_mul(ctx, which, C.byref(v))
    print("field mul bench", "Fr" if which == 0 else "Fq", rc, "%.3e mul/s" % v.value)
print("ALL OK" if ok else "FAILED")
sys.exit(0 if ok else 1)
