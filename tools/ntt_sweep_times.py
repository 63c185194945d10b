"""Average duration of each sweep of a multi-sweep transform from a rocprofv3 kernel-trace CSV.

usage: python tools/ntt_sweep_times.py <kernel_trace.csv> <sweeps>"""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "ntt_pass_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
P = int(sys.argv[2])
rows = rows[-(len(rows) // P // 2) * P:]   # the second half: steady state
for p in range(P):
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[p::P]]
    r0 = rows[p]
    print("sweep %d: %.1f us avg over %d (grid %s, lds %s, vgpr %s)" % (
        p, sum(d) / len(d) / 1e3, len(d), r0.get("Grid_Size_X", r0.get("Grid_Size")), r0.get("LDS_Block_Size"), r0.get("VGPR_Count")))
