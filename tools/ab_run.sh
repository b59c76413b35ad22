#!/bin/bash
# Run a micro-benchmark over every build/variants/lib_*.so, twice, interleaved (same box, same clocks).
#   tools/ab_run.sh dw|pw
what=${1:-dw}
for rep in 1 2; do
  for lib in build/variants/lib_*.so; do
    echo "== $lib rep $rep"
    python tools/bench_kernels.py --what $what --lib $lib 2>&1 | grep -E "dw_fwd|dw_bwd|dw_wgrad|TOTAL|bf16"
  done
done
