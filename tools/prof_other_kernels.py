#!/usr/bin/env python3
"""From a rocprofv3 kernel_stats CSV: the kernels that are NOT the GEMM / depthwise families, per step.
usage: tools/prof_other_kernels.py <kernel_stats.csv> <steps>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"kernel time {tot / steps / 1e3:.1f} us per step over {steps:g} steps")
for r in rows:
    n = r["Name"]
    if not any(k in n for k in ("pw_gemm", "pw_wgrad", "dwconv")):
        print(f"{float(r['TotalDurationNs']) / steps / 1e3:8.1f} us/step {int(r['Calls']) / steps:6.1f} x  avg {float(r['AverageNs']) / 1e3:8.1f} us  {n[:100]}")
