#!/bin/bash
# Whole-step A/B of library variants (build/variants/lib_*.so, tools/ab_variants.sh) on one box, interleaved, two rounds:
#   tools/ab_bench_libs.sh [lib ...]      (default: every build/variants/lib_*.so) -- "base" = the in-tree library
mkdir -p gpurun_out
libs=${@:-$(ls build/variants/lib_*.so)}
for rep in 1 2; do for lib in base $libs; do
  if [ "$lib" = base ]; then unset VOICE100_LIB; else export VOICE100_LIB=$PWD/$lib; fi
  python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$lib', d['ms_per_step'],d['windows_ms_per_step'],d['roofline']['frac'],d['kernel_ms_per_step'])"
done; done
