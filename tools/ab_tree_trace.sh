#!/bin/bash
# per-kernel step breakdown (rocprofv3 kernel trace, nominal steps only) of this tree and of build/<name>/ on one box
other=$1; here=$PWD; out=$here/gpurun_out/abtrace; rm -rf $out; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
for t in this $other; do
  if [ $t = this ]; then d=$here; else d=$here/build/$other; fi
  cd $d
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/$t -o p -- python3 bench.py --steps 30 --warmup 5 --windows 0 --host-contention 0 --sustained-seconds 0 --no-extras --no-cpu-baseline --no-other-configs --diag-no-timestretch > $out/$t.json 2> $out/$t.err
  python3 $here/tools/trace_step.py "$(find $out/$t -name '*kernel_trace.csv' | head -1)" 70 > $out/${t}_breakdown.txt
  rm -rf $out/$t
done
