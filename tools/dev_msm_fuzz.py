"""One-off check of commit() at many SRS / polynomial sizes against the trapdoor identity
commit(f) = f(tau) * G (development aid; every size picks its own window width, bucket kernel and
lanes-per-bucket)."""
import random
import sys

import plonk_amd as pa
from oracle import curve, protocol as P
from oracle.field import R_MOD

rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
tau = 0x1F2E3D4C5B6A79887766554433221100FFEEDDCCBBAA9988 % R_MOD
sizes = [1, 2, 3, 15, 16, 17, 255, 256, 257, 511, 512, 513, 4095, 4096, 8191, 8192, 8193] + \
        [rng.randrange(1, 1 << rng.randrange(4, 17)) for _ in range(30)] + [(1 << 17) - 5, (1 << 18) + 3]
bad = 0
for n in sizes:
    srs = pa.Srs.from_tau(tau, n - 1)
    for m in sorted({n, max(1, n // 3), max(1, n - 1)}):
        kind = rng.randrange(3)
        if kind == 0:
            f = [rng.randrange(R_MOD) for _ in range(m)]
        elif kind == 1:
            f = [rng.choice([0, 1, R_MOD - 1, 2, 5]) for _ in range(m)]
        else:
            f = [rng.randrange(R_MOD) if rng.random() < 0.3 else 0 for _ in range(m)]
        got = pa.Commitment.commit(f, srs).point
        want = curve.to_affine(curve.scalar_mul(curve.G1_GEN, P.poly_eval(f, tau)))
        if got != want:
            bad += 1
            print("MISMATCH n=%d m=%d kind=%d" % (n, m, kind), flush=True)
    srs.close()
print("checked %d SRS sizes, %d mismatches" % (len(sizes), bad))
sys.exit(1 if bad else 0)
