#!/bin/bash
# round-4 second GPU pass: WORLD synthesis tests, new stack tests, PMC of the depthwise forward (fixed), bench line
mkdir -p gpurun_out/r04b
python -m pytest tests/test_gpu_world.py tests/test_gpu_stack.py -q -x > gpurun_out/r04b/pytest_world.log 2>&1; echo "pytest rc=$?" > gpurun_out/r04b/status.txt
tools/pmc_bench_dw.sh gpurun_out/r04b/pmc_dw > gpurun_out/r04b/pmc_dw.log 2>&1
python tools/pmc_dw_json.py gpurun_out/r04b/pmc_dw --json gpurun_out/r04b/dw_fwd_pmc.json > gpurun_out/r04b/pmc_dw_table.txt 2>&1
cp gpurun_out/r04b/dw_fwd_pmc.json profiles/dw_fwd_pmc.json
python bench.py > gpurun_out/r04b/bench.json 2> gpurun_out/r04b/bench.err; echo "bench rc=$?" >> gpurun_out/r04b/status.txt
tail -25 gpurun_out/r04b/pytest_world.log; cat gpurun_out/r04b/status.txt; cat gpurun_out/r04b/pmc_dw_table.txt | tail -30; wc -l gpurun_out/r04b/bench.json
