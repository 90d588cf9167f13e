"""Where the main queue spends a layer (diagnostic): python scripts/layer_breakdown.py kernel_trace.csv [layer indices ...]
Layers are cut at k_select_best (one per calibrated layer); per layer: span, busy time of the main queue, idle gaps, and
the kernels by total time."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
q_time = collections.Counter()
for r in rows:
    q_time[r["Queue_Id"]] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
main_q = max(q_time, key=lambda q: sum(1 for r in rows if r["Queue_Id"] == q))
main = [r for r in rows if r["Queue_Id"] == main_q]
cuts = [i for i, r in enumerate(main) if "k_select_best" in r["Kernel_Name"]]
want = [int(a) for a in sys.argv[2:]] or list(range(len(cuts) - 22, len(cuts)))
def short(n):
    n = n.replace("void ", "").replace("effq::", "")
    return n[:n.index("(")] if "(" in n else n[:60]
for li in want:
    if li <= 0 or li >= len(cuts):
        continue
    seg = main[cuts[li - 1] + 1: cuts[li] + 1]
    t0, t1 = int(seg[0]["Start_Timestamp"]), int(seg[-1]["End_Timestamp"])
    by = collections.Counter(); cnt = collections.Counter()
    busy = 0
    for r in seg:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        by[short(r["Kernel_Name"])] += d; cnt[short(r["Kernel_Name"])] += 1; busy += d
    # the ADMM chain: from the first prox GEMM to the last projection
    chain = [i for i, r in enumerate(seg) if "k_prox" in r["Kernel_Name"] or "k_project_dual" in r["Kernel_Name"]]
    cs = (int(seg[chain[-1]]["End_Timestamp"]) - int(seg[chain[0]]["Start_Timestamp"])) / 1e6 if chain else 0.0
    pre = (int(seg[chain[0]]["Start_Timestamp"]) - t0) / 1e6 if chain else 0.0
    post = (t1 - int(seg[chain[-1]]["End_Timestamp"])) / 1e6 if chain else 0.0
    print(f"layer {li}: span {(t1 - t0) / 1e6:.2f} ms = before the chain {pre:.2f} + chain {cs:.2f} + after {post:.2f};  main-queue busy {busy / 1e6:.2f} ms, {len(seg)} kernels")
    for k, v in by.most_common(10):
        print(f"      {v / 1e6:7.3f} ms  {cnt[k]:5d} x  {k}")

# optional: LIST=<layer index> prints every main-queue kernel of that layer before its chain starts, with the idle gap in front
import os
if os.environ.get("LIST"):
    li = int(os.environ["LIST"])
    seg = main[cuts[li - 1] + 1: cuts[li] + 1]
    prev_end = int(main[cuts[li - 1]]["End_Timestamp"])
    t0 = prev_end
    for r in seg:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"  {(s - t0) / 1000:9.1f} us  gap {(s - prev_end) / 1000:7.1f}  +{(e - s) / 1000:8.1f}  {short(r['Kernel_Name'])}")
        prev_end = e
        if "k_prox" in r["Kernel_Name"]:
            break
