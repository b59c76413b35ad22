mkdir -p gpurun_out
for rep in 1 2 3; do for v in 1 0; do
  V100_CTC_LIN=$v python bench.py --no-cpu-baseline --no-other-configs --no-extras --sustained-seconds 4 --host-contention 0 --windows 2 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('STEP ctc_lin=$v', d['ms_per_step'],'sust',d['sustained']['ms_per_step'],'loss',d['loss'],'launches',d['launches_per_step'],'other',d['roofline_step']['families_ms']['other'])"
done; done > gpurun_out/r06p_ctc_step.txt 2>&1; cat gpurun_out/r06p_ctc_step.txt
