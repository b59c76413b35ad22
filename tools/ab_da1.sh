for r in 1 2; do for v in 1 0; do
  V100_IR_DA1=$v python3 bench.py --no-cpu-baseline --no-other-configs --no-extras --sustained-seconds 4 --host-contention 0 --windows 2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('DA1=$v', 'ms', d['ms_per_step'], 'sust', d['sustained']['ms_per_step'], 'nominal kernel_ms', d['roofline_step']['kernel_ms'], d['roofline_step']['families_ms'])"
done; done
