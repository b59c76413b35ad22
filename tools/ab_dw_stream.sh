#!/bin/bash
# A/B of the streaming depthwise kernels (depthwise_stream16.h) against the general MFMA kernels in the three cache regimes
# (tools/bench_dw_regimes.py), one box, interleaved: rows in flight per wave (D) and cache policy of the row traffic (NT).
# Run on the GPU box from the repo root.
run() { echo "== $*"; env "$@" python tools/bench_dw_regimes.py --iters 40 --bwd 2>&1 | grep TOTAL; }
for rep in 1 2; do
  run V100_DW_STREAM=0 V100_DW_STREAM_BWD=0
  for nt in 0 1; do
    run V100_DW_STREAM_D=1 V100_DW_STREAM_BWD_D=1 V100_DW_STREAM_NT=$nt
    run V100_DW_STREAM_D=2 V100_DW_STREAM_BWD_D=2 V100_DW_STREAM_NT=$nt
    run V100_DW_STREAM_D=4 V100_DW_STREAM_BWD_D=3 V100_DW_STREAM_NT=$nt
  done
done
