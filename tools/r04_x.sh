#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_world_analysis.py -q -x -p no:cacheprovider 2>&1 | grep -E "^FAILED|^ERROR|passed|failed|^E  " | head
python tools/bench_world_analysis.py --iters 10 --no-oracle 2>&1 | grep -v amdgpu
python tools/fuzz_world_analysis.py --n 24 2>&1 | tail -2
