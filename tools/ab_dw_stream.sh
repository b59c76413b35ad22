#!/bin/bash
# A/B of the streaming depthwise forward kernel (depthwise_stream16.h) against the general MFMA kernel in the three cache regimes
# (tools/bench_dw_regimes.py), same box, same process order; run on the GPU box from the repo root.
for v in 1 0 1 0; do
  echo "== V100_DW_STREAM=$v"
  V100_DW_STREAM=$v python tools/bench_dw_regimes.py --iters 40 | tail -4
done
