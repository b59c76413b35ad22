#!/bin/bash
# Diagnostic: ms/step and ms per 1024 input frames when EVERY step is time-stretched by a fixed rate (bench.py --diag-stretch-rate)
for r in "$@"; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --diag-stretch-rate $r 2>/dev/null | R=$r python -c "
import sys, json, os
r = int(os.environ['R'])
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print(r, d['ms_per_step'], 'per-100%:', round(d['ms_per_step'] * 100 / r, 3), d['kernel_ms_per_step'])"
done
