#!/bin/bash
# A/B of one environment knob over the whole step, interleaved repetitions on one box:
#   tools/ab_env.sh V100_WG_TARGET "512 256 384 768 1024" [reps]
name=$1; vals=$2; reps=${3:-2}
for rep in $(seq $reps); do for v in $vals; do
  env $name=$v python bench.py --no-cpu-baseline --no-other-configs --no-extras --sustained-seconds 3 --host-contention 0 --windows 2 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('STEP $name=$v rep $rep', d['ms_per_step'],'sust',d['sustained']['ms_per_step'],d['kernel_ms_per_step'],'nominal',d['roofline_step']['families_ms'])"
done; done
