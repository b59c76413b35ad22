mkdir -p gpurun_out
for rep in 1 2; do for nt in 1 0; do
  V100_DW_STREAM_NT=$nt python bench.py --no-cpu-baseline --no-other-configs > gpurun_out/r03e_bench_nt${nt}_$rep.json 2>/dev/null
  python -c "
import json;d=json.load(open('gpurun_out/r03e_bench_nt${nt}_$rep.json'));print('nt=$nt', d['ms_per_step'],d['host_enqueue_ms_per_step'],d['windows_ms_per_step'],d['roofline']['frac'],d['roofline']['avg_launch_us'],d['kernel_ms_per_step'])"
done; done
V100_DW_STREAM=0 V100_DW_STREAM_BWD=0 python bench.py --no-cpu-baseline --no-other-configs > gpurun_out/r03e_bench_general.json 2>/dev/null
python -c "
import json;d=json.load(open('gpurun_out/r03e_bench_general.json'));print('general', d['ms_per_step'],d['host_enqueue_ms_per_step'],d['windows_ms_per_step'],d['roofline']['frac'],d['roofline']['avg_launch_us'],d['kernel_ms_per_step'])"
