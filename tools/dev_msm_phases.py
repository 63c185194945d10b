"""Per-phase device times of commit() by size, with the window width chosen (development aid)."""
import sys, ctypes as C, numpy as np, time
import plonk_amd as pa
for k in (17,18,19,20):
    n=1<<k
    ctx=pa.api.default_context()
    srs=pa.Srs.from_tau(0x1234567, n-1)
    sc=np.random.default_rng(k).integers(0,1<<62,size=(n,4),dtype=np.uint64)
    dev=ctx.alloc(n*32); ctx.upload(dev, sc)
    srs.commit_dev(dev,n)
    ctx.msm_set_timing(True)
    srs.commit_dev(dev,n); t=ctx.msm_last_timing()
    ctx.msm_set_timing(False)
    t0=time.perf_counter()
    for _ in range(20): srs.commit_dev(dev,n)
    dt=(time.perf_counter()-t0)/20
    print(k, "c=%d"%t["window_bits"], "%.3f ms"%(dt*1e3), {a:round(t[a],3) for a in ("sort_ms","bucket_ms","big_bucket_ms","reduce_ms")})
    ctx.free(dev); srs.close()
