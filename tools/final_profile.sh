#!/bin/bash
# Round-end evidence -> gpurun_out/final/ (copy what is to be judged into profiles/):
#   rocprofv3 --kernel-trace --stats of the bench command (per-kernel table + per-step breakdown), the PMC passes over bench.py's
#   nominal step for EVERY kernel (tools/pmc_bench_step.sh -> the 1x1-GEMM counter table, the step-level roofline table and
#   profiles/step_pmc.json, which bench.py's roofline_step.bytes_measured reads), the per-launch HBM traffic of the depthwise forward
#   launches (nominal + stretched; profiles/dw_fwd_pmc.json for roofline.traffic), the GEMM yardstick, kernel tables of config 3 / 5,
#   and the unprofiled bench line.
tag=${1:-r05}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/final; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o p -- python3 bench.py --steps 30 --warmup 5 --windows 0 --host-contention 0 --sustained-seconds 0 --no-extras --no-cpu-baseline --no-other-configs > $out/${tag}_bench_line_under_rocprof.json 2> $out/prof_err.txt
f=$(find $out/prof -name "*kernel_stats.csv" | head -1)
cp "$f" $out/${tag}_bench_bf16_kernel_stats.csv
python3 tools/prof_summary.py "$f" 36 60 > $out/${tag}_bench_bf16_summary.txt
python3 tools/trace_step.py "$(find $out/prof -name '*kernel_trace.csv' | head -1)" 70 > $out/${tag}_step_breakdown.txt
rm -rf $out/prof
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o p -- python3 tools/bench_infer.py --iters 20 > $out/${tag}_infer_line_under_rocprof.json 2>> $out/prof_err.txt
f=$(find $out/prof -name "*kernel_stats.csv" | head -1)
python3 tools/prof_summary.py "$f" 1 40 > $out/${tag}_infer_configs_kernel_summary.txt
rm -rf $out/prof
# every kernel of the nominal step: SQ / TCC / GRBM counters (separate passes) -> GEMM counter table, step roofline table, step_pmc.json
tools/pmc_bench_step.sh $out/pmc_step > $out/pmc_step.log 2>&1
python3 tools/pmc_step_table.py $out/pmc_step --gemm-out $out/${tag}_gemm_pmc.txt --step-out $out/${tag}_step_roofline.txt --json $out/step_pmc.json > /dev/null 2>> $out/prof_err.txt
cp $out/step_pmc.json profiles/step_pmc.json
# HBM traffic of the depthwise forward launches AS bench.py DISPATCHES THEM (one row per launch) -> the file bench.py reads for
# `roofline.traffic` (tagged with the depthwise sources' hash), refreshed before the unprofiled bench line below
tools/pmc_bench_dw.sh $out/pmc_dw > /dev/null 2>&1
python3 tools/pmc_dw_json.py $out/pmc_dw --json $out/dw_fwd_pmc.json > $out/${tag}_dw_fwd_pmc_per_launch.txt 2>&1
cp $out/dw_fwd_pmc.json profiles/dw_fwd_pmc.json
rm -rf $out/pmc_dw/*.csv
python3 tools/gemm_yardstick.py > $out/${tag}_gemm_yardstick.txt 2>> $out/prof_err.txt
# per-kernel table of a step with EVERY batch stretched to 110 % (T' = 563: the 768-position depthwise forms, five 128-column GEMM tiles)
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o p -- python3 bench.py --steps 20 --warmup 5 --windows 0 --host-contention 0 --sustained-seconds 0 --no-extras --no-cpu-baseline --no-other-configs --diag-stretch-rate 110 > $out/${tag}_stretch110_line_under_rocprof.json 2>> $out/prof_err.txt
python3 tools/prof_summary.py "$(find $out/prof -name '*kernel_stats.csv' | head -1)" 26 40 > $out/${tag}_stretch110_kernel_summary.txt
rm -rf $out/prof
python3 bench.py > $out/${tag}_bench_line.json 2> $out/bench_err.txt
cp gpurun_out/bench_details.json $out/${tag}_bench_details.json 2>/dev/null
tail -c 600 $out/${tag}_bench_line.json
