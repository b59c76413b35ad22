#!/bin/bash
mkdir -p gpurun_out/r04g
python -m pytest tests/test_gpu_world.py -q > gpurun_out/r04g/pytest_world.log 2>&1; echo "pytest world rc=$?" > gpurun_out/r04g/status.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04g/prof_world -o w -- python3 tools/bench_world.py --iters 5 > gpurun_out/r04g/world_prof.log 2>&1
f=$(find gpurun_out/r04g/prof_world -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -6 "$f" | cut -c1-200 > gpurun_out/r04g/world_kernel_stats.csv; rm -rf gpurun_out/r04g/prof_world
python tools/bench_world.py > gpurun_out/r04g/world.txt 2>&1
tail -5 gpurun_out/r04g/pytest_world.log; cat gpurun_out/r04g/status.txt gpurun_out/r04g/world.txt gpurun_out/r04g/world_kernel_stats.csv
