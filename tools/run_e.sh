mkdir -p gpurun_out
(timeout 1200 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -8) > gpurun_out/r06e_tests.txt; cat gpurun_out/r06e_tests.txt
for rep in 1 2; do for v in base oldpitch; do
  if [ "$v" = base ]; then L=""; unset VOICE100_LIB; else L="--lib build/variants/lib_$v.so"; export VOICE100_LIB=$PWD/build/variants/lib_$v.so; fi
  for T in 563 460 640; do echo "== micro $v T=$T rep $rep"; python tools/bench_dw_regimes.py --iters 20 --bwd --T $T --only "rotating,bwd rotating" $L 2>&1 | grep TOTAL; done
  for r in 110 90 130; do
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-extras --sustained-seconds 0 --host-contention 0 --windows 0 --diag-stretch-rate $r 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());r=d['roofline'];print('STRETCH $r $v', d['ms_per_step'],'frac',r['frac'],'fam',r.get('frac_family'),d['kernel_ms_per_step'])"
  done
  python bench.py --no-cpu-baseline --no-other-configs --no-extras --sustained-seconds 3 --host-contention 0 --windows 2 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());r=d['roofline'];print('STEP $v', d['ms_per_step'],'sust',d['sustained']['ms_per_step'],'frac',r['frac'],r.get('frac_nominal_step'),'fam',r.get('frac_family'),r.get('frac_family_nominal_step'),d['kernel_ms_per_step'])"
done; done > gpurun_out/r06e_pitch_ab.txt 2>&1
unset VOICE100_LIB
cat gpurun_out/r06e_pitch_ab.txt
