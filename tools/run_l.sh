mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in base ctcabl1 ctcabl2 ctcabl4 ctcabl7; do
  if [ $v = base ]; then unset VOICE100_LIB; else export VOICE100_LIB=$PWD/build/variants/lib_$v.so; fi
  rm -rf gpurun_out/ctcprof
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ctcprof -o p -- python3 tools/micro/ctc_time.py > gpurun_out/ctc_l_$v.log 2>&1
  echo "== $v"; python3 tools/prof_summary.py "$(find gpurun_out/ctcprof -name '*kernel_stats.csv' | head -1)" 205 3
done > gpurun_out/r06l_ctc_kernels.txt 2>&1
rm -rf gpurun_out/ctcprof
cat gpurun_out/r06l_ctc_kernels.txt
