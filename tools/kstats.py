#!/usr/bin/env python3
"""Summarise a rocprofv3 *_kernel_stats.csv: total kernel time, launches, top kernels."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 12
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / 1e6:.2f} ms, {sum(int(r['Calls']) for r in rows)} launches")
for r in rows[:top]:
    n = r["Name"]
    n = n[n.find("k_"):][:46] if "k_" in n else n[:46]
    print(f"{n:48s} {r['Calls']:>6s} {float(r['TotalDurationNs']) / 1e6:8.2f} ms  avg {float(r['AverageNs']) / 1e3:8.2f} us")
