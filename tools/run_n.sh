mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py tests/test_gpu_edge_cases.py tests/test_gpu_dist2.py -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|Error|assert" | tail -6 > gpurun_out/r06n_tests.txt; cat gpurun_out/r06n_tests.txt
for v in 1 0; do
  export V100_CTC_LIN=$v
  rm -rf gpurun_out/ctcprof
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ctcprof -o p -- python3 tools/micro/ctc_time.py > gpurun_out/ctc_n_$v.log 2>&1
  echo "== V100_CTC_LIN=$v"; python3 tools/prof_summary.py "$(find gpurun_out/ctcprof -name '*kernel_stats.csv' | head -1)" 205 6
  grep ctc_loss gpurun_out/ctc_n_$v.log
done > gpurun_out/r06n_ctc_kernels.txt 2>&1
rm -rf gpurun_out/ctcprof; unset V100_CTC_LIN
cat gpurun_out/r06n_ctc_kernels.txt
for rep in 1 2 3; do for v in 1 0; do
  V100_CTC_LIN=$v python bench.py --no-cpu-baseline --no-other-configs --no-extras --sustained-seconds 3 --host-contention 0 --windows 2 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('STEP ctc_lin=$v', d['ms_per_step'],'sust',d['sustained']['ms_per_step'],'loss',d['loss'],'launches',d['launches_per_step'],'other',d['roofline_step']['families_ms']['other'])"
done; done > gpurun_out/r06n_ctc_step.txt 2>&1; cat gpurun_out/r06n_ctc_step.txt
