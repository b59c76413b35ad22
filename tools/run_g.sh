mkdir -p gpurun_out
MICRO="base x2" MICRO563="base x2" STEP="base x2" tools/ab_r06_dw.sh > gpurun_out/r06g_x2_ab.txt 2>&1
export VOICE100_LIB=$PWD/build/variants/lib_x2.so
tools/pmc_bench_dw.sh gpurun_out/pmc_dw_x2 > /dev/null 2>&1
python3 tools/pmc_dw_json.py gpurun_out/pmc_dw_x2 --json gpurun_out/dw_fwd_pmc_x2.json > gpurun_out/r06g_x2_dw_fwd_pmc_per_launch.txt 2>&1
rm -rf gpurun_out/pmc_dw_x2
unset VOICE100_LIB
grep -E "^==|TOTAL|STEP" gpurun_out/r06g_x2_ab.txt; cat gpurun_out/r06g_x2_dw_fwd_pmc_per_launch.txt
