#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats CSV: ms per step and calls per step per kernel.
usage: tools/prof_summary.py <kernel_stats.csv> <n_steps_in_run> [top]"""
import csv
import sys

f, n = sys.argv[1], float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / n / 1e6:.3f} ms/step over {n:g} steps")
for r in rows[:top]:
    print(f"{float(r['TotalDurationNs']) / n / 1e6:8.3f} ms/step {int(r['Calls']) / n:6.1f} calls/step  avg {float(r['AverageNs']) / 1e3:8.1f} us  {r['Name'][:100]}")
