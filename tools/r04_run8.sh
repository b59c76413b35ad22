#!/bin/bash
# kernel tables of the inference configs (rocprofv3 --kernel-trace --stats), channel-major path on / off
mkdir -p gpurun_out/r04h
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
prof() {  # tag workload precision
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04h/p_$1 -o w -- python3 tools/prof_eval.py $2 $3 --iters 10 > gpurun_out/r04h/$1.log 2>&1
  f=$(find gpurun_out/r04h/p_$1 -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" "$1" >> gpurun_out/r04h/kernel_tables.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"## {sys.argv[2]}: kernel time {tot/13/1e3:.1f} us per call (13 calls: 3 warm-up + 10)")
for r in rows[:16]:
    print(f"  {float(r['TotalDurationNs'])/13/1e3:8.1f} us/call {int(r['Calls'])/13:6.1f} x  avg {float(r['AverageNs'])/1e3:7.1f} us  {r['Name'][:110]}")
PY
  rm -rf gpurun_out/r04h/p_$1
}
prof asr32_bf16_cm asr32 bf16
VOICE100_EVAL_CM=0 prof asr32_bf16_bm asr32 bf16
prof stream256_fp16_cm stream256 fp16
VOICE100_EVAL_CM=0 prof stream256_fp16_bm stream256 fp16
prof predict16_bf16 predict16 bf16
prof chainwave_bf16 chainwave bf16
grep "ms per call" gpurun_out/r04h/*.log; cat gpurun_out/r04h/kernel_tables.txt | head -150
