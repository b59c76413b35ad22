#!/usr/bin/env python3
"""Per-kernel time per training step from a rocprofv3 --kernel-trace CSV of bench.py: steps are delimited by adam_step_kernel.
usage: tools/trace_step.py <kernel_trace.csv> [top_n]"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 45
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
adam = [i for i, n in enumerate(names) if n.startswith("adam_step")]
agg = collections.defaultdict(lambda: [0, 0.0])
nst = 0
for s in range(4, len(adam) - 1):          # skip the warm-up steps
    a, b = adam[s], adam[s + 1]
    nst += 1
    for i in range(a, b):
        n = re.sub(r"\(.*", "", names[i])
        if "dwconv" in n:
            n = re.sub(r"<.*", "", n)
        agg[n][0] += 1
        agg[n][1] += (int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in agg.values()) / max(nst, 1)
nl = sum(v[0] for v in agg.values()) / max(nst, 1)
print(f"{nst} steps, {nl:.0f} launches/step, kernel time {tot:.0f} us/step")
for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{d / nst:8.1f} us {c / nst:5.1f} x {n[:120]}")
