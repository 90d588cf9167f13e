#!/usr/bin/env python3
"""Where does the main queue wait for an inverse?  From a rocprofv3 kernel trace of the bench
(`rocprofv3 --kernel-trace --output-format csv -- python3 bench.py --steps 1 --warmup 1 ...`):

    python scripts/inv_timeline.py <..._kernel_trace.csv> [min_gap_ms=0.5]

For the LAST calibration in the trace: every inverse (k_build_a64 ... k_a64_to_f32 on one queue) with its queue, start, end
and duration, and every gap of the busiest (main) queue longer than min_gap_ms, all on one time axis (ms from the start of
the calibration), so that a gap can be matched with the inverse whose end closes it."""
import csv
import sys


def main():
    path = sys.argv[1]
    min_gap = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 0.5e6
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"),
                         int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)))
    rows.sort()
    cut, end = 0, rows[0][1]
    for i, (s, e, *_r) in enumerate(rows):
        if s - end > 50e6 and i < len(rows) - 1000:
            cut = i
        end = max(end, e)
    rows = rows[cut:]
    t0 = rows[0][0]
    per_q = {}
    for r in rows:
        per_q.setdefault(r[3], []).append(r)
    main_q = max(per_q, key=lambda q: len(per_q[q]))
    events = []
    for q, rs in per_q.items():
        open_inv = None
        for s, e, name, _q, grid in rs:
            if "k_build_a64" in name:
                open_inv = (s, grid)
            elif "k_a64_to_f32" in name and open_inv is not None:
                events.append((open_inv[0], f"inverse on queue {q}{' (main)' if q == main_q else ''}: start {(open_inv[0]-t0)/1e6:9.3f}  "
                                            f"end {(e-t0)/1e6:9.3f}  took {(e-open_inv[0])/1e6:7.3f} ms"))
                open_inv = None
    prev_e, prev_n = None, None
    for s, e, name, _q, _g in per_q[main_q]:
        if prev_e is not None and s - prev_e >= min_gap:
            events.append((prev_e, f"MAIN QUEUE IDLE {(prev_e-t0)/1e6:9.3f} -> {(s-t0)/1e6:9.3f}  ({(s-prev_e)/1e6:7.3f} ms)  "
                                   f"{prev_n[:40]} -> {name.replace('effq::','')[:50]}"))
        if prev_e is None or e > prev_e:
            prev_e, prev_n = e, name.replace("effq::", "")
    span = (max(r[1] for r in rows) - t0) / 1e6
    print(f"calibration span {span:.1f} ms, main queue {main_q}")
    for _t, line in sorted(events):
        print(line)


if __name__ == "__main__":
    main()
