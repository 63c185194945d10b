"""Condenses rocprofv3 SQ counter passes (one --pmc run per program, --output-format csv) into one line
per kernel: the counters of its last launch, wave-level VALU instructions per unit of work and the
share of VALU issue slots in use.

usage: sq_summary.py <out.txt> <label>=<counter_collection.csv>:<units per launch of the kernels named after>=<kernel,kernel> ...
e.g.   sq_summary.py out.txt "msm 2^20=.../610_counter_collection.csv:13631456=msm_bucket_kernel"

valu_issue_utilisation = SQ_INSTS_VALU / 1024 SIMDs * 4 cycles / (GRBM_GUI_ACTIVE / 8 XCDs): a wave64 VALU
instruction occupies its SIMD's issue port for 4 cycles; GRBM_GUI_ACTIVE is summed over the 8 XCDs."""
import collections
import re
import csv
import sys


def main():
    out, specs = sys.argv[1], sys.argv[2:]
    lines = ["SQ counters per launch (last launch of each kernel); utilisation = SQ_INSTS_VALU / 1024 SIMDs * 4 cycles / (GRBM_GUI_ACTIVE / 8 XCDs)"]
    for spec in specs:
        label, rest = spec.split("=", 1)
        path, units, kernels = rest.split(":")[0], None, None
        tail = rest.split(":", 1)[1]
        units, kernels = tail.split("=")
        units, kernels = float(units), kernels.split(",")
        per = collections.defaultdict(dict)   # (kernel, dispatch) -> counter -> value
        for r in csv.DictReader(open(path)):
            name = re.sub(r"<.*", "", r["Kernel_Name"].split("(")[0]).replace("void ", "").replace("plonk::", "").strip()   # compile-time variants count as one kernel
            per[(name, int(r["Dispatch_Id"]))][r["Counter_Name"]] = float(r["Counter_Value"])
        for k in kernels:
            last = 1
            if "*" in k:   # kernel*K: the sum over its last K launches (the sweeps of one transform)
                k, last = k.split("*")[0], int(k.split("*")[1])
            ids = sorted(d for (n, d) in per if n == k)
            if not ids:
                continue
            c = collections.defaultdict(float)
            for d in ids[-last:]:
                for name, v in per[(k, d)].items():
                    c[name] += v
            if last > 1:
                k = "%s (last %d launches)" % (k, last)
            util = c["SQ_INSTS_VALU"] / 1024 * 4 / (c["GRBM_GUI_ACTIVE"] / 8)
            lines.append("%-15s %-20s %s  valu_wave_instr_per_unit=%.1f (x64 lanes: %.0f lane-instr)  valu_issue_utilisation=%.2f" % (
                label, k, " ".join("%s=%d" % (n, v) for n, v in sorted(c.items())), c["SQ_INSTS_VALU"] / units,
                c["SQ_INSTS_VALU"] * 64 / units, util))
    open(out, "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
