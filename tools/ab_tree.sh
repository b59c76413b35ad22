#!/bin/bash
# A/B of two whole trees on one box (python-side changes cannot be switched by VOICE100_LIB): this tree against a built copy of
# another commit under build/<name>/ (git archive <commit> | tar -x -C build/<name>; make there).   tools/ab_tree.sh base [reps]
other=$1; reps=${2:-2}; here=$PWD
line() { python -c "
import json,sys;d=json.loads(sys.stdin.read());print('STEP $1 rep $2', d['ms_per_step'],'sust',d['sustained']['ms_per_step'],'launches',d['launches_per_step'],'host',d['host_enqueue_ms_per_step'],d['kernel_ms_per_step'],'nominal',d['roofline_step']['families_ms'])"; }
opts="--no-cpu-baseline --no-other-configs --no-extras --sustained-seconds 3 --host-contention 0 --windows 2"
for rep in $(seq $reps); do
  (cd $here && python bench.py $opts 2>/dev/null | line this $rep)
  (cd $here/build/$other && python bench.py $opts 2>/dev/null | line $other $rep)
done
