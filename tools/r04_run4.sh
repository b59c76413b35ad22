#!/bin/bash
mkdir -p gpurun_out/r04d
python -m pytest tests/test_gpu_eval_cm.py tests/test_gpu_world.py -q -x > gpurun_out/r04d/pytest_new.log 2>&1; echo "pytest new rc=$?" > gpurun_out/r04d/status.txt
python -m pytest tests/test_gpu_models.py -q -x > gpurun_out/r04d/pytest_models.log 2>&1; echo "pytest models rc=$?" >> gpurun_out/r04d/status.txt
python tools/bench_world.py > gpurun_out/r04d/world.txt 2>&1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04d/prof_world -o w -- python3 tools/bench_world.py --iters 5 > gpurun_out/r04d/world_prof.log 2>&1
f=$(find gpurun_out/r04d/prof_world -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f" > gpurun_out/r04d/world_kernel_stats.csv; rm -rf gpurun_out/r04d/prof_world
python bench.py --graph-diag --no-cpu-baseline --no-other-configs --sustained-seconds 0 > gpurun_out/r04d/bench_graph.json 2> gpurun_out/r04d/bench_graph.err
tail -12 gpurun_out/r04d/pytest_new.log; tail -5 gpurun_out/r04d/pytest_models.log; cat gpurun_out/r04d/status.txt gpurun_out/r04d/world.txt gpurun_out/r04d/world_kernel_stats.csv
python -c "
import json;d=json.loads(open('gpurun_out/r04d/bench_graph.json').readline());print(d['ms_per_step'], d['launch_graph_diagnostic'])"
