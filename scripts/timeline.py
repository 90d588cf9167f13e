#!/usr/bin/env python3
"""Idle-gap analysis of a rocprofv3 kernel trace (`rocprofv3 --kernel-trace --output-format csv -- python3 bench.py ...`).

    python scripts/timeline.py <..._kernel_trace.csv> [min_gap_us=15] [top=40]

Prints, for the LAST calibration in the trace (kernels after the last long quiet period that follows the warm-up):
the span, the time at least one kernel was running on the device (union over all streams), the idle time, the time
per queue, and the idle gaps grouped by (kernel that ended before the gap -> kernel that started after it): a host
synchronisation shows up as a repeated gap between the same pair of kernels.
"""
import csv
import sys
from collections import defaultdict


def short(name):
    name = name.replace("effq::", "").replace("void ", "")
    cut = name.find("(")
    return (name if cut < 0 else name[:cut])[:60]


def main():
    path = sys.argv[1]
    min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 15.0
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    rows = []
    with open(path) as f:
        rd = csv.DictReader(f)
        for r in rd:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                         r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
    rows.sort()
    # the timed calibration = everything after the last gap > 50 ms (state-dict reload + sync between warm-up and step)
    cut = 0
    end = rows[0][1]
    for i, (s, e, *_r) in enumerate(rows):
        if s - end > 50e6 and i < len(rows) - 1000:
            cut = i
        end = max(end, e)
    # refine: the calibration starts at the last gap > 2 ms in the first part of the remaining trace
    rows = rows[cut:]
    t0 = rows[0][0]
    end = rows[0][1]
    start_i = 0
    for i, (s, e, *_r) in enumerate(rows):
        if s - end > 2e6 and (s - t0) < 0.5 * (rows[-1][1] - t0):
            start_i = i
        end = max(end, e)
    rows = rows[start_i:]
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    span = (t1 - t0) / 1e6
    busy = 0.0
    gaps = defaultdict(lambda: [0, 0.0])
    cur_end = rows[0][0]
    last_name = "(start)"
    per_q = defaultdict(float)
    for s, e, name, q, st in rows:
        per_q[(q, st)] += (e - s) / 1e6
        if s > cur_end:
            g = (s - cur_end) / 1e3
            if g >= min_gap:
                k = (short(last_name), short(name))
                gaps[k][0] += 1
                gaps[k][1] += g
            busy += 0  # idle
        if e > cur_end:
            busy += (e - max(s, cur_end)) / 1e6
            cur_end = e
            last_name = name
    print(f"kernels {len(rows)}  span {span:.1f} ms  device busy (union) {busy:.1f} ms  idle {span - busy:.1f} ms "
          f"({100 * (span - busy) / span:.1f} %)")
    for (q, st), ms in sorted(per_q.items(), key=lambda kv: -kv[1]):
        print(f"  queue {q} stream {st}: {ms:.1f} ms of kernels")
    # the same per queue: where does the busiest queue (the ADMM chain) wait?  (a wait for another stream's event or for
    # the host shows up as a gap on this queue while the device is busy elsewhere)
    main_q = max(per_q.items(), key=lambda kv: kv[1])[0]
    qrows = [r for r in rows if (r[3], r[4]) == main_q]
    qgaps = defaultdict(lambda: [0, 0.0])
    qidle = 0.0
    for (s0, e0, n0, *_a), (s1, e1, n1, *_b) in zip(qrows, qrows[1:]):
        g = (s1 - e0) / 1e3
        if g > 0:
            qidle += g
        if g >= min_gap:
            k = (short(n0), short(n1))
            qgaps[k][0] += 1
            qgaps[k][1] += g
    print(f"busiest queue {main_q}: {len(qrows)} kernels, idle between its kernels {qidle / 1e3:.1f} ms; gaps >= "
          f"{min_gap:.0f} us: {sum(v[1] for v in qgaps.values()) / 1e3:.1f} ms")
    for (a, b), (n, us) in sorted(qgaps.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"  [main] {us / 1e3:8.2f} ms {n:5d} x {us / n:8.1f} us   {a}  ->  {b}")
    tot = sum(v[1] for v in gaps.values()) / 1e3
    print(f"idle gaps >= {min_gap:.0f} us: {tot:.1f} ms in {sum(v[0] for v in gaps.values())} gaps")
    for (a, b), (n, us) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"  {us / 1e3:8.2f} ms {n:5d} x {us / n:8.1f} us   {a}  ->  {b}")


if __name__ == "__main__":
    main()
