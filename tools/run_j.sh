mkdir -p gpurun_out
V100_SLAB_SIDE=1 timeout 900 python -m pytest tests/test_gpu_act16.py tests/test_gpu_stack.py tests/test_gpu_models.py tests/test_gpu_fuzz.py tests/test_gpu_dist2.py -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|error|Error" | tail -8 > gpurun_out/r06j_tests.txt; cat gpurun_out/r06j_tests.txt
for rep in 1 2 3; do for v in 0 1; do
  V100_SLAB_SIDE=$v python bench.py --no-cpu-baseline --no-other-configs --no-extras --sustained-seconds 3 --host-contention 0 --windows 2 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());r=d['roofline'];print('STEP slab_side=$v', d['ms_per_step'],'sust',d['sustained']['ms_per_step'],'host',d['host_enqueue_ms_per_step'],d['kernel_ms_per_step'],'nominal',d['roofline_step']['families_ms'])"
done; done > gpurun_out/r06j_slab_side.txt 2>&1; cat gpurun_out/r06j_slab_side.txt
