mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_io_oracle.py tests/test_gpu_act16.py tests/test_gpu_stack.py tests/test_gpu_kernels.py tests/test_gpu_models.py tests/test_gpu_fuzz.py tests/test_gpu_edge_cases.py -x -q -p no:cacheprovider 2>&1 | tail -5) > gpurun_out/r06d_tests.txt; cat gpurun_out/r06d_tests.txt
for T in 512 560 563 568 576 640 704 763 768 460 448 384; do echo "== T=$T"; python tools/bench_dw_regimes.py --iters 20 --bwd --T $T --only "rotating,bwd rotating" 2>&1 | grep -E "TOTAL"; done > gpurun_out/r06d_Tsweep.txt 2>&1; cat gpurun_out/r06d_Tsweep.txt
MICRO="" MICRO563="" STEP="base fp0" tools/ab_r06_dw.sh > gpurun_out/r06d_step_ab.txt 2>&1; cat gpurun_out/r06d_step_ab.txt
