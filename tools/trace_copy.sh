#!/bin/bash
# kernel trace of a short bench run + the neighbours of every __amd_rocclr_copyBuffer dispatch (which host call issues them)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/tr && mkdir -p gpurun_out/tr
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr -o t -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-other-configs > gpurun_out/tr/log.txt 2>&1
f=$(find gpurun_out/tr -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
hits = [i for i, n in enumerate(names) if "copyBuffer" in n or "fillBuffer" in n]
print(len(rows), "dispatches,", len(hits), "copy/fill")
# the last full step: show every hit in the last 300 dispatches with 2 neighbours each side
for i in hits:
    if i < len(rows) - 260:
        continue
    print("---", i)
    for j in range(max(0, i - 2), min(len(rows), i + 3)):
        r = rows[j]
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        print(("  >> " if j == i else "     ") + f"{d:8.1f} us grid={r.get('Grid_Size_X','?'):>9}  {names[j][:110]}")
PY
