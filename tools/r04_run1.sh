#!/bin/bash
# round-4 first GPU pass: tests, GEMM yardstick, bench line, learnable loss curves, per-launch PMC of the depthwise forward
mkdir -p gpurun_out/r04a
python -m pytest tests -m gpu -x -q > gpurun_out/r04a/pytest.log 2>&1; echo "pytest rc=$?" > gpurun_out/r04a/status.txt
python tools/gemm_yardstick.py > gpurun_out/r04a/gemm_yardstick.txt 2> gpurun_out/r04a/gemm_yardstick.err
python bench.py > gpurun_out/r04a/bench.json 2> gpurun_out/r04a/bench.err; echo "bench rc=$?" >> gpurun_out/r04a/status.txt
python tools/loss_curve.py --task learnable --steps 600 > gpurun_out/r04a/loss_learnable.txt 2>&1
tools/pmc_bench_dw.sh gpurun_out/r04a/pmc_dw > gpurun_out/r04a/pmc_dw.log 2>&1
python tools/pmc_dw_json.py gpurun_out/r04a/pmc_dw --json gpurun_out/r04a/dw_fwd_pmc.json > gpurun_out/r04a/pmc_dw_table.txt 2>&1
tail -3 gpurun_out/r04a/pytest.log; cat gpurun_out/r04a/status.txt; head -c 600 gpurun_out/r04a/bench.json
