mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_kernels.py -k "ctc" -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|Error|assert" | tail -6 > gpurun_out/r06m_ctc_tests.txt; cat gpurun_out/r06m_ctc_tests.txt
for v in base ctcabl1 ctcabl2 ctcabl4 ctcabl7; do
  if [ $v = base ]; then unset VOICE100_LIB; else export VOICE100_LIB=$PWD/build/variants/lib_$v.so; fi
  rm -rf gpurun_out/ctcprof
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ctcprof -o p -- python3 tools/micro/ctc_time.py > gpurun_out/ctc_l_$v.log 2>&1
  echo "== $v"; python3 tools/prof_summary.py "$(find gpurun_out/ctcprof -name '*kernel_stats.csv' | head -1)" 205 5 | grep -E "lin|total"
done > gpurun_out/r06m_ctc_kernels.txt 2>&1
rm -rf gpurun_out/ctcprof; unset VOICE100_LIB
cat gpurun_out/r06m_ctc_kernels.txt
for rep in 1 2; do for v in 1 0; do
  V100_CTC_LIN=$v python bench.py --no-cpu-baseline --no-other-configs --no-extras --sustained-seconds 3 --host-contention 0 --windows 2 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('STEP ctc_lin=$v', d['ms_per_step'],'sust',d['sustained']['ms_per_step'],'loss',d['loss'],'launches',d['launches_per_step'],'other',d['roofline_step']['families_ms']['other'])"
done; done > gpurun_out/r06m_ctc_step.txt 2>&1; cat gpurun_out/r06m_ctc_step.txt
