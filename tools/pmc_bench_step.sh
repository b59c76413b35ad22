#!/bin/bash
# Hardware counters of EVERY kernel of the training step exactly as bench.py dispatches it (round-4 review, items 1a and 2):
# rocprofv3 PMC passes over bench.py itself on the nominal step (time-stretch off, so every dispatch of a kernel is the same
# problem), one counter set per run (--pmc never together with a sys / hip trace; FETCH_SIZE and WRITE_SIZE in separate runs: TCC
# slots), plus one plain --kernel-trace run for the un-perturbed durations.  tools/pmc_step_table.py turns the CSVs into
# profiles/r05_gemm_pmc.txt (the 1x1-GEMM family) and profiles/step_pmc.json (HBM bytes per kernel and per step, for bench.py's
# `roofline_step.bytes_measured`).
#   tools/pmc_bench_step.sh <outdir>          (GPU box, repo root; the program itself follows `--`)
out=$1
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
COMMON="--no-presize --steps 4 --warmup 2 --windows 0 --host-contention 0 --sustained-seconds 0 --no-extras --no-cpu-baseline --no-other-configs --no-kernel-timing --diag-no-timestretch"
run() {  # name counters...
  name=$1; shift
  rm -rf "$out/raw_$name"
  if [ "$1" = "-" ]; then
    rocprofv3 --kernel-trace --output-format csv -d "$out/raw_$name" -o p -- python3 bench.py $COMMON > "$out/$name.log" 2>&1 || true
    f=$(find "$out/raw_$name" -name "*kernel_trace.csv" | head -1)
  else
    rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$out/raw_$name" -o p -- python3 bench.py $COMMON > "$out/$name.log" 2>&1 || true
    f=$(find "$out/raw_$name" -name "*counter_collection.csv" | head -1)
  fi
  [ -n "$f" ] && cp "$f" "$out/$name.csv"
  rm -rf "$out/raw_$name"
}
run trace -
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_BF16
# (an unknown counter name fails the whole pass: fall back to the set every earlier round used)
[ -f "$out/sq2.csv" ] || { run sq2b SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVES; [ -f "$out/sq2b.csv" ] && mv "$out/sq2b.csv" "$out/sq2.csv"; }
run fetch FETCH_SIZE
run write WRITE_SIZE
run grbm GRBM_GUI_ACTIVE
rocprofv3 -L > "$out/counters_available.txt" 2>&1 || true
ls -la "$out"
