#!/usr/bin/env python3
"""Neighbours of a kernel in a rocprofv3 --kernel-trace results .db: for every dispatch whose name contains <pattern>, the
names of the `ctx` dispatches before and after it (to find which host call issues it).
usage: tools/prof_seq.py <results.db> <pattern> [ctx] [max_hits]"""
import sqlite3
import sys

db, pat = sys.argv[1], sys.argv[2]
ctx = int(sys.argv[3]) if len(sys.argv) > 3 else 2
max_hits = int(sys.argv[4]) if len(sys.argv) > 4 else 12
c = sqlite3.connect(db).cursor()
rows = list(c.execute("select name, start, end from kernels order by start"))
hits = [i for i, r in enumerate(rows) if pat in r[0]]
print(f"{len(hits)} dispatches match {pat!r}")
for i in hits[len(hits) // 2: len(hits) // 2 + max_hits]:
    print("---")
    for j in range(max(0, i - ctx), min(len(rows), i + ctx + 1)):
        n, s, e = rows[j]
        print(f"{'>>' if j == i else '  '} {(e - s) / 1e3:8.1f} us  {n[:120]}")
