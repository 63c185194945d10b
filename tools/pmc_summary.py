"""Condenses two rocprofv3 counter passes (--pmc FETCH_SIZE and --pmc WRITE_SIZE, separate runs of
the same command, --output-format csv) into per-kernel HBM traffic per launch.

usage: pmc_summary.py <dir with *_fetch/ and *_write/ sub-directories> <out.txt> <out.json>

HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in KiB, and on gfx950 FETCH_SIZE
tallies a 128-byte request as 64 bytes."""
import collections
import csv
import glob
import re
import json
import os
import sys


def kernel_name(full):
    """plonk::ntt_pass_kernel<true, false>(plonk::NttPassArgs) -> ntt_pass_kernel: the compile-time variants of a kernel count as one"""
    return re.sub(r"<.*", "", full.split("(")[0]).replace("void ", "").replace("plonk::", "").strip()


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[kernel_name(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return acc


def main():
    root, out_txt, out_json = sys.argv[1:4]
    lines = ["HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024; averages over the later half of the launches"]
    table = {}
    for fdir in sorted(glob.glob(os.path.join(root, "*_fetch"))):
        tag = os.path.basename(fdir)[:-6]
        f = per_kernel(glob.glob(os.path.join(fdir, "*counter_collection.csv"))[0], "FETCH_SIZE")
        w = per_kernel(glob.glob(os.path.join(root, tag + "_write", "*counter_collection.csv"))[0], "WRITE_SIZE")
        for k in f:
            if not (k.startswith("msm_") or k.startswith("ntt_") or k.startswith("quotient")):
                continue
            fv = f[k][-(len(f[k]) // 2 or 1):]
            wv = w.get(k, [0.0])[-(len(w.get(k, [0.0])) // 2 or 1):]
            fa, wa = sum(fv) / len(fv), sum(wv) / len(wv)
            traffic = (2 * fa + wa) * 1024
            table[k] = {"run": tag, "launches": len(f[k]), "fetch_size_kib": round(fa, 1), "write_size_kib": round(wa, 1),
                        "hbm_bytes_per_launch": round(traffic)}
            lines.append("%-4s %-28s launches=%3d FETCH_SIZE=%11.1f KiB WRITE_SIZE=%11.1f KiB traffic=%8.1f MB"
                         % (tag, k, len(f[k]), fa, wa, traffic / 1e6))
    open(out_txt, "w").write("\n".join(lines) + "\n")
    json.dump(table, open(out_json, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
