#!/bin/bash
# In-step A/B of library builds: the bench's contract window + a short sustained window + the per-family kernel times, for the default
# library and every build/variants/lib_*.so, twice, interleaved.   tools/ab_bench.sh [extra bench.py args]
for r in 1 2; do for l in "" build/variants/lib_*.so; do
  VOICE100_LIB=$l python3 bench.py --no-cpu-baseline --no-other-configs --no-extras --sustained-seconds 4 --host-contention 0 --windows 2 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('${l:-default}'.ljust(32), 'ms', d['ms_per_step'], 'sust', d['sustained']['ms_per_step'], 'dwfrac', d['roofline']['frac'], d['kernel_ms_per_step'])"
done; done
