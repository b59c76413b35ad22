mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -15 > gpurun_out/r06i_tests.txt
V100_IR_DA1_TMAX=768 timeout 600 python -m pytest tests/test_gpu_act16.py tests/test_gpu_stack.py tests/test_gpu_fuzz.py tests/test_gpu_models.py -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -15 >> gpurun_out/r06i_tests.txt
cat gpurun_out/r06i_tests.txt
