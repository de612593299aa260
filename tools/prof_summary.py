#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats CSV: per-kernel ms per DDIM step."""
import csv, glob, sys
d, steps = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 7.0
f = glob.glob(d + "/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:18]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print(f"{n[:66]:66s} calls={r['Calls']:>5s} ms/step={float(r['TotalDurationNs']) / steps / 1e6:7.2f} "
          f"avg_us={float(r['AverageNs']) / 1e3:8.1f} pct={float(r['Percentage']):6.2f}")
print("total ms per step", tot / steps / 1e6)
