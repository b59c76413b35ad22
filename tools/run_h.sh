mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_io_oracle.py tests/test_gpu_act16.py tests/test_gpu_stack.py tests/test_gpu_kernels.py tests/test_gpu_models.py tests/test_gpu_fuzz.py -x -q -p no:cacheprovider 2>&1 | tail -5) > gpurun_out/r06h_tests.txt; cat gpurun_out/r06h_tests.txt
MICRO="base g4 g8 g32 g64" MICRO563="" STEP="base bnx" tools/ab_r06_dw.sh > gpurun_out/r06h_group_ab.txt 2>&1
for rep in 1 2; do for tm in 768 512; do for r in 110 130; do
  V100_IR_DA1_TMAX=$tm python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-extras --sustained-seconds 0 --host-contention 0 --windows 0 --diag-stretch-rate $r 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());r=d['roofline'];print('STRETCH $r da1_tmax=$tm', d['ms_per_step'],'frac',r['frac'],'fam',r.get('frac_family'),d['kernel_ms_per_step'])"
done; done; done >> gpurun_out/r06h_group_ab.txt 2>&1
grep -E "^==|TOTAL|STEP|STRETCH" gpurun_out/r06h_group_ab.txt
