#!/bin/bash
mkdir -p gpurun_out/r04f
python -m pytest tests/test_gpu_world.py -q > gpurun_out/r04f/pytest_world.log 2>&1; echo "pytest world rc=$?" > gpurun_out/r04f/status.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04f/prof_world -o w -- python3 tools/bench_world.py --iters 5 > gpurun_out/r04f/world_prof.log 2>&1
f=$(find gpurun_out/r04f/prof_world -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -6 "$f" | cut -c1-200 > gpurun_out/r04f/world_kernel_stats.csv; rm -rf gpurun_out/r04f/prof_world
python -m pytest tests -m gpu -q -x > gpurun_out/r04f/pytest_all.log 2>&1; echo "pytest all rc=$?" >> gpurun_out/r04f/status.txt
tail -8 gpurun_out/r04f/pytest_world.log; tail -4 gpurun_out/r04f/pytest_all.log; cat gpurun_out/r04f/status.txt; tail -1 gpurun_out/r04f/world_prof.log; cat gpurun_out/r04f/world_kernel_stats.csv
