"""Totals per kernel over ONE proof period of a rocprofv3 kernel trace, plus the union of busy time
(development aid).  usage: python tools/kernel_sums.py <kernel_trace.csv> [marker_kernel_prefix]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
mark = sys.argv[2] if len(sys.argv) > 2 else "quotient"
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']),
             r['Kernel_Name'].split('(')[0].replace('plonk::', '').replace('void ', '')) for r in rows)
q = [i for i, e in enumerate(ev) if e[2].startswith(mark)]
k = len(q) // 2
a, b = q[k], q[k + 1]
period = ev[b][0] - ev[a][0]
tot = {}
for s, e, n in ev[a:b]:
    c = tot.setdefault(n, [0, 0])
    c[0] += 1
    c[1] += e - s
busy, cur_s, cur_e = 0, None, None
for s, e, n in ev[a:b]:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("period %.3f ms, some kernel running %.3f ms, idle %.3f ms" % (period / 1e6, busy / 1e6, (period - busy) / 1e6))
for n, (c, d) in sorted(tot.items(), key=lambda x: -x[1][1]):
    print("%-44s %4d  %9.1f us" % (n[:44], c, d / 1e3))
