#!/usr/bin/env python3
"""Per-kernel summary (calls, total, average, share) of a rocprofv3 --kernel-trace run.

rocprofv3 (ROCm 7.2) writes a rocpd SQLite database by default; this prints the same table
`--stats` would, so that the summary can be committed under profiles/ as plain text.

    python tools/rocprof_summary.py gpurun_out/prof/frozen/r1_results.db > profiles/r01_frozen.txt
"""
import sqlite3
import sys


def main(path, top=40):
    db = sqlite3.connect(path)
    rows = list(db.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, "
                           "max(end-start)/1e3 from kernels group by name order by 3 desc"))
    total = sum(r[2] for r in rows)
    span = db.execute("select (max(end)-min(start))/1e3 from kernels").fetchone()[0]
    print("# %s" % path)
    print("# kernels: %d distinct, %d dispatches, busy %.1f us over a %.1f us span" %
          (len(rows), sum(r[1] for r in rows), total, span))
    print("%-100s %7s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "%"))
    for r in rows[:top]:
        print("%-100s %7d %12.1f %10.2f %10.2f %10.2f %6.2f" % (r[0][:100], r[1], r[2], r[3], r[4], r[5], 100 * r[2] / total))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
