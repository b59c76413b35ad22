mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -k "ctc" -q -p no:cacheprovider 2>&1 | tail -25 > gpurun_out/r06k_ctc_tests.txt; cat gpurun_out/r06k_ctc_tests.txt
timeout 600 python -m pytest tests/test_gpu_models.py tests/test_gpu_edge_cases.py -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|Error" | tail -5 >> gpurun_out/r06k_ctc_tests.txt
python tools/micro/ctc_time.py 2>&1 | tail -12 > gpurun_out/r06k_ctc_time.txt; V100_CTC_LIN=0 python tools/micro/ctc_time.py 2>&1 | tail -12 >> gpurun_out/r06k_ctc_time.txt; cat gpurun_out/r06k_ctc_time.txt
for rep in 1 2; do for v in 1 0; do
  V100_CTC_LIN=$v python bench.py --no-cpu-baseline --no-other-configs --no-extras --sustained-seconds 3 --host-contention 0 --windows 2 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('STEP ctc_lin=$v', d['ms_per_step'],'sust',d['sustained']['ms_per_step'],'loss',d['loss'],'launches',d['launches_per_step'],'nominal',d['roofline_step']['families_ms'])"
done; done > gpurun_out/r06k_ctc_step.txt 2>&1; cat gpurun_out/r06k_ctc_step.txt
