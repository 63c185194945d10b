"""Per-kernel statistics from a rocprofv3 results database (the .db written by --kernel-trace).

usage: python tools/db_kernels.py results.db [substring ...]
With substrings: the dispatch timeline around the LAST dispatch whose kernel name holds the first one."""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    if len(sys.argv) == 2:
        q = (f"select s.kernel_name, count(*), avg(d.end-d.start)/1000.0, sum(d.end-d.start)/1000.0, min(d.end-d.start)/1000.0 "
             f"from {kd} d join {ks} s on d.kernel_id=s.id group by 1 order by 4 desc")
        rows = list(cur.execute(q))
        tot = sum(r[3] for r in rows)
        print(f"{'kernel':70s} {'calls':>6s} {'avg us':>10s} {'min us':>10s} {'%':>6s}")
        for r in rows:
            print(f"{r[0][:70]:70s} {r[1]:6d} {r[2]:10.1f} {r[4]:10.1f} {r[3] / tot * 100:6.2f}")
        print(f"total {tot / 1000:.2f} ms")
        return
    rows = list(cur.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
    idx = [i for i, r in enumerate(rows) if sys.argv[2] in r[0]]
    if not idx:
        print("no such kernel")
        return
    i0 = idx[-1]
    t0 = rows[i0][1]
    for r in rows[max(0, i0 - 6):i0 + 10]:
        print(f"{(r[1] - t0) / 1000:9.1f} {(r[2] - t0) / 1000:9.1f} {(r[2] - r[1]) / 1000:8.1f} {r[0][:80]}")


if __name__ == "__main__":
    main()
