#!/usr/bin/env python3
"""Per-kernel table from a rocprofv3 --kernel-trace results .db (rocpd): ms/step, calls/step, avg us.
usage: tools/prof_db.py <results.db> <n_steps_in_run> [top] [--csv out.csv]"""
import csv
import sqlite3
import sys

db, n = sys.argv[1], float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 and not sys.argv[3].startswith("--") else 40
out = sys.argv[sys.argv.index("--csv") + 1] if "--csv" in sys.argv else None
c = sqlite3.connect(db).cursor()
rows = list(c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
if out:
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([r[0], r[1], r[2], r[3], 100 * r[2] / tot, r[4], r[5]])
print(f"total kernel time {tot / n / 1e6:.3f} ms/step over {n:g} steps, {sum(r[1] for r in rows) / n:.0f} launches/step")
for r in rows[:top]:
    print(f"{r[2] / n / 1e6:8.3f} ms/step {r[1] / n:6.1f} calls/step  avg {r[3] / 1e3:8.1f} us  {r[0][:110]}")
