#!/bin/bash
# Per-kernel time per training step (rocprofv3 --kernel-trace over bench.py, tools/trace_step.py) for the default library and every
# build/variants/lib_*.so:   tools/trace_ab.sh <outdir> [top_n]
out=$1; top=${2:-40}
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for l in "" build/variants/lib_*.so; do
  name=$(basename "${l:-default}" .so)
  export VOICE100_LIB=$l
  rm -rf "$out/raw"
  rocprofv3 --kernel-trace --output-format csv -d "$out/raw" -o p -- python3 bench.py --steps 12 --warmup 3 --windows 0 --host-contention 0 --sustained-seconds 0 --no-extras --no-cpu-baseline --no-other-configs --no-kernel-timing --diag-no-timestretch > /dev/null 2>&1
  python3 tools/trace_step.py "$(find $out/raw -name '*kernel_trace.csv' | head -1)" $top > "$out/steps_$name.txt"
  rm -rf "$out/raw"
  echo "== $name"; head -$((top + 1)) "$out/steps_$name.txt" | cut -c1-150
done
