#!/bin/bash
mkdir -p gpurun_out/r04e
python -m pytest tests/test_gpu_world.py tests/test_gpu_eval_cm.py -q -x > gpurun_out/r04e/pytest_new.log 2>&1; echo "pytest new rc=$?" > gpurun_out/r04e/status.txt
python tools/bench_world.py > gpurun_out/r04e/world.txt 2>&1
python tools/bench_dw_regimes.py --bwd > gpurun_out/r04e/dw_regimes_bm.txt 2>&1
python tools/bench_dw_regimes.py --bwd --cm > gpurun_out/r04e/dw_regimes_cm.txt 2>&1
python tools/bench_dw_regimes.py --bwd > gpurun_out/r04e/dw_regimes_bm2.txt 2>&1
python tools/bench_dw_regimes.py --bwd --cm > gpurun_out/r04e/dw_regimes_cm2.txt 2>&1
python tools/gemm_yardstick.py > gpurun_out/r04e/gemm_yardstick.txt 2> gpurun_out/r04e/gemm_yardstick.err
tail -6 gpurun_out/r04e/pytest_new.log; cat gpurun_out/r04e/status.txt gpurun_out/r04e/world.txt; grep TOTAL gpurun_out/r04e/dw_regimes_*.txt; grep -A6 "layout A/B" gpurun_out/r04e/gemm_yardstick.txt
