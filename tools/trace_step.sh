#!/bin/bash
# kernel trace of a short bench run -> gpurun_out/tr/t_kernel_trace.csv + per-kernel us/step of the timed steps (tools/trace_step.py)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/tr && mkdir -p gpurun_out/tr
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr -o t -- python3 bench.py --steps ${STEPS:-12} --warmup 3 --no-cpu-baseline --no-other-configs > gpurun_out/tr/log.txt 2>&1
python3 tools/trace_step.py "$(find gpurun_out/tr -name '*kernel_trace.csv' | head -1)" ${TOP:-45}
