"""plonk_ntt_dev_noalloc (the transform that calls no hipMalloc: its intermediate image and twiddle
tables live in a static device array) against plonk_ntt_dev on the same inputs, bit for bit, through
the C ABI.  Consecutive calls share the static array, so the sequence below changes size, direction
and coset generator from one call to the next, the second call smaller than the first."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _both(ctx, pa, log_n, flags, shift, seed):
    n = 1 << log_n
    a = np.random.default_rng(seed).integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)  # Montgomery images below r
    sh = C.byref(pa.api._fr_struct(shift)) if shift is not None else None
    d_ref, d_new = ctx.alloc(n * 32), ctx.alloc(n * 32)
    try:
        ctx.upload(d_ref, a)
        ctx.upload(d_new, a)
        ctx.check(pa._lib.lib.plonk_ntt_dev(ctx.h, d_ref, log_n, flags, sh, 1, n))
        ctx.check(pa._lib.lib.plonk_ntt_dev_noalloc(ctx.h, d_new, log_n, flags, sh))
        ref = ctx.download(d_ref, n * 32)
        new = ctx.download(d_new, n * 32)
    finally:
        ctx.free(d_ref)
        ctx.free(d_new)
    return a, ref.reshape(n, 4), new.reshape(n, 4)


def test_noalloc_ntt_matches_ntt_dev(ctx):
    import plonk_amd as pa
    assert pa._lib.lib.plonk_ntt_noalloc_max_log_n() == 18
    # (log_n, flags, shift): two sweeps, then a smaller two-sweep one, a single-sweep one, the
    # same plan twice in a row (tables reused), another generator, the largest size, sizes 1 and 2
    seq = [(17, 2, None), (13, 3, None), (9, 0, None), (9, 0, None), (13, 2, 5), (13, 3, 5), (18, 2, None), (18, 3, None),
           (16, 1, None), (11, 0, None), (1, 1, None), (0, 2, None), (12, 2, None)]
    for step, (log_n, flags, shift) in enumerate(seq):
        a, ref, new = _both(ctx, pa, log_n, flags, shift, 1000 + step)
        assert np.array_equal(ref, new), (step, log_n, flags, shift)
        if log_n >= 1:
            assert not np.array_equal(new, a)   # the transform did run


def test_noalloc_ntt_against_the_oracle_and_round_trip(ctx):
    import random
    import plonk_amd as pa
    from oracle import fft as offt
    from oracle.field import R_MOD, root_of_unity
    log_n = 12
    n = 1 << log_n
    rng = random.Random(5)
    v = [rng.randrange(R_MOD) for _ in range(n)]
    dev = ctx.alloc(n * 32)
    try:
        ctx.upload(dev, pa.fr_to_array(v))
        ctx.check(pa._lib.lib.plonk_ntt_dev_noalloc(ctx.h, dev, log_n, 2, None))
        assert pa.fr_from_array(ctx.download(dev, n * 32).reshape(n, 4)) == offt.coset_fft(v, root_of_unity(n), 7, n)
        ctx.check(pa._lib.lib.plonk_ntt_dev_noalloc(ctx.h, dev, log_n, 3, None))
        assert pa.fr_from_array(ctx.download(dev, n * 32).reshape(n, 4)) == v
    finally:
        ctx.free(dev)


def test_noalloc_ntt_rejects_what_does_not_fit(ctx):
    import plonk_amd as pa
    dev = ctx.alloc(32)
    try:
        rc = pa._lib.lib.plonk_ntt_dev_noalloc(ctx.h, dev, 19, 0, None)
        assert rc == pa._lib.PLONK_E_INVALID
        assert b"static workspace" in pa._lib.lib.plonk_last_error(ctx.h)
        assert pa._lib.lib.plonk_ntt_dev_noalloc(ctx.h, dev, 4, 8, None) == pa._lib.PLONK_E_INVALID   # unknown flag
        assert pa._lib.lib.plonk_ntt_dev_noalloc(ctx.h, None, 4, 0, None) == pa._lib.PLONK_E_INVALID
    finally:
        ctx.free(dev)


def test_noalloc_ntt_from_two_contexts_back_to_back(ctx):
    """Two contexts (two streams) on the same GPU call one after the other with no synchronisation in between: the
    second call's stream is ordered behind the first call's kernels by the library (the first call regenerates the
    tables the second reads, and both use the one intermediate image)."""
    import plonk_amd as pa
    other = pa.Context(0)
    log_n = 16
    n = 1 << log_n
    lib = pa._lib.lib
    bufs = []
    try:
        for which, c in enumerate((ctx, other, ctx, other)):
            a = np.random.default_rng(70 + which).integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
            d = c.alloc(n * 32)
            c.upload(d, a)
            bufs.append((c, d, a))
        # a first call with other tables, so that the sequence below starts by regenerating them
        ctx.check(lib.plonk_ntt_dev_noalloc(ctx.h, bufs[0][1], 10, 0, None))
        ctx.upload(bufs[0][1], bufs[0][2])
        for c, d, _ in bufs:   # enqueued back to back, alternating streams, nothing waited for
            c.check(lib.plonk_ntt_dev_noalloc(c.h, d, log_n, 2, None))
        for c, d, a in bufs:
            got = c.download(d, n * 32).reshape(n, 4)
            ref = ctx.alloc(n * 32)
            ctx.upload(ref, a)
            ctx.check(lib.plonk_ntt_dev(ctx.h, ref, log_n, 2, None, 1, n))
            want = ctx.download(ref, n * 32).reshape(n, 4)
            ctx.free(ref)
            assert np.array_equal(got, want)
    finally:
        for c, d, _ in bufs:
            c.free(d)
        other.close()
