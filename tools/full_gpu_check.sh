#!/bin/bash
# full GPU suite + round-end evidence (tools/final_profile.sh) in one box
tag=${1:-run}
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|error" | tail -5 > gpurun_out/${tag}_pytest_gpu.txt
cat gpurun_out/${tag}_pytest_gpu.txt
bash tools/final_profile.sh $tag
