mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_kernels.py -k "ctc" -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|Error|assert|^E " | tail -8 > gpurun_out/r06o_ctc_tests.txt; cat gpurun_out/r06o_ctc_tests.txt
python tools/micro/ctc_flag_probe.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06o_ctc_flags.txt; cat gpurun_out/r06o_ctc_flags.txt
for v in 1 0; do
  export V100_CTC_LIN=$v
  rm -rf gpurun_out/ctcprof
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ctcprof -o p -- python3 tools/micro/ctc_time.py > gpurun_out/ctc_o_$v.log 2>&1
  echo "== V100_CTC_LIN=$v"; python3 tools/prof_summary.py "$(find gpurun_out/ctcprof -name '*kernel_stats.csv' | head -1)" 205 6
  grep ctc_loss gpurun_out/ctc_o_$v.log
done > gpurun_out/r06o_ctc_kernels.txt 2>&1
rm -rf gpurun_out/ctcprof; unset V100_CTC_LIN
cat gpurun_out/r06o_ctc_kernels.txt
