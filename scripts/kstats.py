"""Condense a rocprofv3 kernel_stats.csv: per-step seconds for the top kernels."""
import csv, sys
path, steps = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = list(csv.DictReader(open(path)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:18]:
    print("%-64s calls %7d  %7.3f s/step  avg %9.1f us  %5.1f%%" % (
        r["Name"][:64], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e9 / steps, float(r["AverageNs"]) / 1e3,
        float(r["Percentage"])))
print("total kernel time per step: %.3f s" % (tot / 1e9 / steps))
