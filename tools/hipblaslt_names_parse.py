#!/usr/bin/env python3
"""kernel trace csv + the order printed by tools/hipblaslt_names.py -> one line per yardstick GEMM: the vendor kernel's name and its
average duration in that run (3 launches, warm)."""
import csv, sys
trace, order = sys.argv[1], [l.strip() for l in open(sys.argv[2]) if l.startswith("C=")]
rows = sorted(csv.DictReader(open(trace)), key=lambda r: int(r["Start_Timestamp"]))
groups, cur = [], []
for r in rows:
    n = r["Kernel_Name"]
    if "elementwise" in n or "fill" in n.lower() or "vectorized" in n:
        if cur: groups.append(cur); cur = []
        continue
    if n.startswith("Cijk") or "gemm" in n.lower() or "Cijk" in n:
        cur.append((n, int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
if cur: groups.append(cur)
# every case: 1 warm-up group + 1 timed group of three
timed = [g for g in groups if len(g) >= 3]
for name, g in zip(order, timed):
    k = g[-1][0]
    us = sum(d for _, d in g[-3:]) / 3e3
    print(f"{name:70s} {us:7.1f} us  {k}")
