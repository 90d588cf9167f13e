"""Kernel-trace excerpt around the launches of one kernel (diagnostic): python scripts/trace_excerpt.py trace.csv [pattern] [first] [count]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
pat = sys.argv[2] if len(sys.argv) > 2 else "k_fpt"
first = int(sys.argv[3]) if len(sys.argv) > 3 else 2
count = int(sys.argv[4]) if len(sys.argv) > 4 else 4
idx = [i for i, r in enumerate(rows) if pat in r["Kernel_Name"]]
print(len(rows), "kernels,", len(idx), "match", pat)
if len(idx) > first + count:
    t0 = int(rows[idx[first]]["Start_Timestamp"])
    for r in rows[idx[first] - 5: idx[first + count] + 3]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"{(s - t0) / 1000:9.1f} us  +{(e - s) / 1000:8.1f}  q{r.get('Queue_Id', '?')}  {r['Kernel_Name'][:80]}")
