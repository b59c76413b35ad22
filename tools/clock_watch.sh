#!/bin/bash
# sample the GPU's clocks and power for as long as the bench runs (sustained window of 12 s): is the step power- / clock-managed?
out=${1:-gpurun_out/clock_watch.txt}
python bench.py --no-cpu-baseline --no-other-configs --no-extras --sustained-seconds 12 --host-contention 0 --windows 0 --no-kernel-timing > /dev/null 2>&1 &
pid=$!
while kill -0 $pid 2>/dev/null; do
  rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|Power \(W\)|junction" | sed -e 's/GPU\[0\]\s*: //' | tr -s ' \t' ' ' | tr '\n' ';'
  echo
done > $out
wait $pid
