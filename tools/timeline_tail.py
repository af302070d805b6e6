"""Timeline of the LAST n kernel dispatches of a rocprofv3 kernel trace (start relative to the first of them, duration,
queue, name): what the timed region of a short bench.py run looks like on the device.
usage: python tools/timeline_tail.py <kernel_trace.csv> [n=400]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 400
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = {}
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    q = r.get("Queue_Id", "?")
    name = r["Kernel_Name"].replace("void slu::", "").replace("slu::", "")[:60]
    gap = (s - prev_end[q]) / 1e3 if q in prev_end else 0.0
    prev_end[q] = e
    print("%9.1f us  +%8.1f us  q%-3s gap %7.1f  grid %-8s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, gap, r.get("Grid_Size", "?"), name))
