#!/bin/bash
# A/B of the streaming depthwise kernels (depthwise_stream16.h) against the general MFMA kernels in the three cache regimes
# (tools/bench_dw_regimes.py), one box, interleaved: rows in flight per wave (D) and cache policy of the row traffic (NT).
# Depth and cache policy are COMPILE-TIME macros of the library (DWS_DEPTH / DWS_BWD_DEPTH / DWS_NT in depthwise_fwd_train16.hip,
# depthwise_fwd_eval16.hip, depthwise_bwd_fused16g.hip): each point of the matrix is its own build (tools/ab_variants.sh) selected with
# VOICE100_LIB; only V100_DW_STREAM / V100_DW_STREAM_BWD (streaming kernels on / off) are read at run time.
# Build here (CPU container, hipcc cross-compiles), run on the GPU box from the repo root:  tools/ab_dw_stream.sh [build|run]
set -e
FILES_DW="depthwise_fwd_train16 depthwise_fwd_eval16 depthwise_bwd_fused16g"
variants() { for nt in 0 1; do for d in 1 2 4; do echo "d${d}nt${nt}"; done; done; }
if [ "${1:-build}" = build ]; then
  for nt in 0 1; do for d in 1 2 4; do
    bd=$d; [ $d = 4 ] && bd=3                       # the fused backward holds three streams per row: depth 3 is its deepest
    FILES="$FILES_DW" VEXTRA="-DDWS_BWD_DEPTH=$bd -DDWS_NT=$nt" tools/ab_variants.sh DWS_DEPTH $d >/dev/null
    mv build/variants/lib_$d.so build/variants/lib_dws_d${d}nt${nt}.so
  done; done
  ls -la build/variants/lib_dws_*.so
  exit 0
fi
run() { echo "== $*"; env "$@" python tools/bench_dw_regimes.py --iters 40 --bwd 2>&1 | grep TOTAL; }
for rep in 1 2; do
  run V100_DW_STREAM=0 V100_DW_STREAM_BWD=0
  for v in $(variants); do run VOICE100_LIB=$PWD/build/variants/lib_dws_$v.so; done
done
mkdir -p gpurun_out
tools/ab_bench_libs.sh build/variants/lib_dws_d1nt0.so build/variants/lib_dws_d1nt1.so
V100_DW_STREAM=0 V100_DW_STREAM_BWD=0 python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('general', d['ms_per_step'],d['host_enqueue_ms_per_step'],d['windows_ms_per_step'],d['roofline']['frac'],d['roofline']['avg_launch_us'],d['kernel_ms_per_step'])"
