#!/bin/bash
# round-4 third GPU pass: channel-major inference tests, WORLD tests, the eval / streaming suites that the new path touches, inference bench
mkdir -p gpurun_out/r04c
python -m pytest tests/test_gpu_eval_cm.py tests/test_gpu_world.py -q -x > gpurun_out/r04c/pytest_new.log 2>&1; echo "pytest new rc=$?" > gpurun_out/r04c/status.txt
python -m pytest tests/test_gpu_models.py tests/test_gpu_act16.py tests/test_gpu_edge_cases.py -q -x > gpurun_out/r04c/pytest_models.log 2>&1; echo "pytest models rc=$?" >> gpurun_out/r04c/status.txt
for p in bf16 fp16; do
  python tools/bench_infer.py --precision $p > gpurun_out/r04c/infer_$p.txt 2>&1
  VOICE100_EVAL_CM=0 python tools/bench_infer.py --precision $p > gpurun_out/r04c/infer_${p}_nocm.txt 2>&1
done
tail -15 gpurun_out/r04c/pytest_new.log; tail -5 gpurun_out/r04c/pytest_models.log; cat gpurun_out/r04c/status.txt; tail -4 gpurun_out/r04c/infer_fp16.txt; tail -4 gpurun_out/r04c/infer_fp16_nocm.txt
