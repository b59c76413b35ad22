#!/bin/bash
# per-kernel breakdown of this tree's NOMINAL step (rocprofv3 kernel trace) -> gpurun_out/trace_this_breakdown.txt
here=$PWD; out=$here/gpurun_out/tt; rm -rf $out; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd $here
rocprofv3 --kernel-trace --stats --output-format csv -d $out/p -o p -- python3 bench.py --steps 30 --warmup 5 --windows 0 --host-contention 0 --sustained-seconds 0 --no-extras --no-cpu-baseline --no-other-configs --diag-no-timestretch > $out/line.json 2> $out/err.txt
python3 tools/trace_step.py "$(find $out/p -name '*kernel_trace.csv' | head -1)" 70 > gpurun_out/trace_this_breakdown.txt
rm -rf $out
