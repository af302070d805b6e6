#!/usr/bin/env python3
"""Per-kernel summary (calls, total, average, share) of a rocprofv3 --kernel-trace run, plus the
stream-level picture (busy time per HIP stream, concurrency) that matters for the look-ahead pipeline.

Accepts the rocpd SQLite database rocprofv3 (ROCm 7.2) writes by default, or the *_kernel_trace.csv of
`--output-format csv`; prints the table `--stats` would, so that it can be committed under profiles/.

    python tools/rocprof_summary.py gpurun_out/prof/final_frozen/f_kernel_trace.csv > profiles/r01_b_frozen.txt
"""
import collections
import csv
import sqlite3
import sys


BY_SHAPE = False      # --by-shape: one row per (kernel, grid size) = per distinct launch shape


def load(path):
    """-> list of (start_ns, end_ns, kernel name, stream id)"""
    if path.endswith(".csv"):
        rows = csv.DictReader(open(path))
        out = []
        for r in rows:
            name = r["Kernel_Name"]
            if BY_SHAPE:
                name = "%s  grid=%sx%sx%s wg=%s" % (name[:70], r.get("Grid_Size_X", "?"), r.get("Grid_Size_Y", "?"),
                                                     r.get("Grid_Size_Z", "?"), r.get("Workgroup_Size_X", "?"))
            out.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Stream_Id", "0")))
        return out
    db = sqlite3.connect(path)
    return [(r[0], r[1], r[2], "0") for r in db.execute("select start, end, name from kernels")]


def main(path, top=40):
    ev = load(path)
    ev.sort()
    agg = collections.defaultdict(list)
    for s, e, name, _ in ev:
        agg[name].append((e - s) / 1e3)
    rows = sorted(((n, len(v), sum(v), sum(v) / len(v), min(v), max(v)) for n, v in agg.items()), key=lambda r: -r[2])
    total = sum(r[2] for r in rows)
    span = (max(e for _, e, _, _ in ev) - min(s for s, _, _, _ in ev)) / 1e3
    print("# %s" % path)
    print("# kernels: %d distinct, %d dispatches, busy %.1f us over a %.1f us span" % (len(rows), len(ev), total, span))
    streams = collections.defaultdict(float)
    for s, e, _, st in ev:
        streams[st] += (e - s) / 1e3
    print("# busy us per HIP stream: " + ", ".join("%s: %.0f" % kv for kv in sorted(streams.items())))
    print("%-100s %7s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "%"))
    for r in rows[:top]:
        print("%-100s %7d %12.1f %10.2f %10.2f %10.2f %6.2f" % (r[0][:100], r[1], r[2], r[3], r[4], r[5], 100 * r[2] / total))


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a != "--by-shape"]
    BY_SHAPE = "--by-shape" in sys.argv
    main(args[0], int(args[1]) if len(args) > 1 else 40)
