#!/usr/bin/env python3
"""The dependent launch chain of ONE iteration from a rocprofv3 --kernel-trace CSV (tools/prof_eval.py workloads): the last
<n_iter>-th of the dispatches, each with its start offset, duration and the gap to its predecessor.
usage: tools/trace_chain.py <kernel_trace.csv> <iterations_in_trace>"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
iters = int(sys.argv[2])
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
per = len(rows) // iters
last = rows[-per:]
t0 = int(last[0]["Start_Timestamp"])
prev_end = t0
tot = 0.0
print(f"{len(rows)} dispatches, {per} per iteration; last iteration:")
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = re.sub(r"\(.*", "", r["Kernel_Name"])[:90]
    print(f"{(s - t0) / 1e3:8.1f} us  +{(s - prev_end) / 1e3:6.1f} gap  {(e - s) / 1e3:7.1f} us  grid {r.get('Grid_Size', '?'):>8} wg {r.get('Workgroup_Size', '?'):>5}  {n}")
    tot += (e - s) / 1e3
    prev_end = e
print(f"iteration: {(prev_end - t0) / 1e3:.1f} us wall on the device, {tot:.1f} us inside kernels")
