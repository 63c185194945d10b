"""Condensed timeline of one proof from a rocprofv3 kernel trace (development aid):
python tools/timeline.py <kernel_trace.csv> [proof_index]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']),
             r['Kernel_Name'].split('(')[0].replace('plonk::', '').replace('void ', '')) for r in rows)
q = [i for i, e in enumerate(ev) if e[2].startswith('quotient')]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(q) // 2
a, b = q[k], q[k + 1]
t0 = ev[a][0]
print("proof period %.2f ms" % ((ev[b][0] - t0) / 1e6))
# busy intervals of bucket kernels; everything else listed when no bucket kernel is running
bucket = [(s, e) for s, e, n in ev[a:b] if n.startswith('msm_bucket')]
def in_bucket(t):
    return any(s <= t < e for s, e in bucket)
gap_start = t0
for s, e in bucket + [(ev[b][0], ev[b][0])]:
    if s > gap_start:
        names = {}
        for s2, e2, n in ev[a:b]:
            lo, hi = max(s2, gap_start), min(e2, s)
            if hi > lo:
                names[n] = names.get(n, 0) + (hi - lo)
        top = sorted(names.items(), key=lambda x: -x[1])[:8]
        print("gap %8.1f -> %8.1f us (%7.1f): %s" % ((gap_start - t0) / 1e3, (s - t0) / 1e3, (s - gap_start) / 1e3,
              ", ".join("%s %.0f" % (n[:22], d / 1e3) for n, d in top)))
    print("    bucket %8.1f -> %8.1f (%7.1f)" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3)) if e > s else None
    gap_start = max(gap_start, e)
