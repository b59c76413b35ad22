#!/bin/bash
# Run the depthwise micro-benchmark over every build/variants/lib_*.so, twice, interleaved (same box, same clocks).
for rep in 1 2; do
  for lib in build/variants/lib_*.so; do
    echo "== $lib rep $rep"
    python tools/bench_kernels.py --what dw --lib $lib 2>&1 | grep -E "dw_fwd|dw_bwd|TOTAL"
  done
done
