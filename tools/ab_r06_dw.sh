#!/bin/bash
# Round-6 A/B of the depthwise kernels: micro-benchmark on a rotating working set (tools/bench_dw_regimes.py) for the named variant
# libraries (build/variants/lib_<v>.so, "base" = the in-tree library), two interleaved repetitions, then the whole step.
#   MICRO="base k1 k3" MICRO563="base bd2 fd2" STEP="base k1 fp3" tools/ab_r06_dw.sh
lib() { if [ "$1" = base ]; then echo ""; else echo "--lib build/variants/lib_$1.so"; fi; }
for rep in 1 2; do
  for v in $MICRO; do
    echo "== T=512 $v rep $rep"
    python tools/bench_dw_regimes.py --iters 30 --bwd --da1 --only "rotating,bwd rotating,da1 rotating" $(lib $v) 2>&1 | grep -E "TOTAL|k= 83|k= 35"
  done
  for v in $MICRO563; do
    echo "== T=563 $v rep $rep"
    python tools/bench_dw_regimes.py --iters 30 --bwd --T 563 --only "rotating,bwd rotating" $(lib $v) 2>&1 | grep -E "TOTAL|k= 83|k= 35"
  done
done
for rep in 1 2; do for v in $STEP; do
  if [ "$v" = base ]; then unset VOICE100_LIB; else export VOICE100_LIB=$PWD/build/variants/lib_$v.so; fi
  python bench.py --no-cpu-baseline --no-other-configs --no-extras --sustained-seconds 3 --host-contention 0 --windows 2 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());r=d['roofline'];print('STEP $v', d['ms_per_step'],'sust',d['sustained']['ms_per_step'],'frac',r['frac'],r.get('frac_nominal_step'),'fam',r.get('frac_family'),r.get('frac_family_nominal_step'),'bwd',r.get('frac_bwd_nominal_step'),'launches',d['launches_per_step'],d['kernel_ms_per_step'],'nominal',d['roofline_step']['families_ms'])"
done; done
unset VOICE100_LIB
