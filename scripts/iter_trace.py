#!/usr/bin/env python3
"""From a rocprofv3 kernel trace CSV: the kernels of a few consecutive ADMM iterations of one layer, per queue, with
start offsets, durations and the gap to the previous kernel of the same queue.
    python scripts/iter_trace.py <kernel_trace.csv> <anchor kernel substring> [occurrence=1000] [iterations=3]"""
import csv, sys
path, anchor = sys.argv[1], sys.argv[2]
occ = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
nit = int(sys.argv[4]) if len(sys.argv) > 4 else 3
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
hits = [i for i, r in enumerate(rows) if anchor in r[2]]
if len(hits) <= occ + nit:
    sys.exit(f"only {len(hits)} occurrences of {anchor}")
t0, t1 = rows[hits[occ]][0], rows[hits[occ + nit]][0]
last_end = {}
for s, e, name, q in rows:
    if s < t0 - 50000 or s >= t1:
        if s < t0:
            last_end[q] = e
        continue
    gap = (s - last_end[q]) / 1e3 if q in last_end else float("nan")
    last_end[q] = e
    nm = name.replace("effq::", "").replace("void ", "")
    nm = nm[:nm.find("(")] if "(" in nm else nm
    print(f"q{q} +{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f} us  gap {gap:8.1f} us  {nm[:60]}")
