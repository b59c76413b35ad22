#!/bin/bash
# Launch chains of the small inference configs (configs[4] at B = 32, configs[0]) -> <outdir>/chain_<workload>.txt
out=${1:-gpurun_out/infer_small}; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for w in stream32nd asr2; do
  rm -rf "$out/raw"
  rocprofv3 --kernel-trace --output-format csv -d "$out/raw" -o p -- python3 tools/prof_eval.py $w ${2:-bf16} --iters 10 > "$out/time_$w.txt" 2>/dev/null
  python3 tools/trace_chain.py "$(find $out/raw -name '*kernel_trace.csv' | head -1)" 13 > "$out/chain_$w.txt"
  rm -rf "$out/raw"
  python3 tools/prof_eval.py $w ${2:-bf16} --iters 200 | tail -1 >> "$out/time_$w.txt"
done
tail -3 "$out"/time_*.txt
