#!/bin/bash
# rocprofv3 PMC passes over the kernel micro-benchmark (one counter set per run: --pmc never together with sys/hip traces).
#   tools/pmc_passes.sh <outdir> [bench_kernels.py args...]     ->  <outdir>/{sq,fetch,write}.csv
#   BENCH_SCRIPT=tools/bench_gemm_io.py tools/pmc_passes.sh <outdir>    (another micro-benchmark with an --iters flag)
# Run on the GPU box from the repo root; put the program itself after `--` (no env / bash -c hop).
set -e
out=$1; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() {  # name, counters...
  name=$1; shift
  rm -rf "$out/raw_$name"
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$out/raw_$name" -o p -- python3 ${BENCH_SCRIPT:-tools/bench_kernels.py} --iters 2 "${ARGS[@]}" > "$out/$name.log" 2>&1 || true
  f=$(find "$out/raw_$name" -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$out/$name.csv"
  rm -rf "$out/raw_$name"
}
ARGS=("$@")
run sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS
run fetch FETCH_SIZE
run write WRITE_SIZE
ls -la "$out"
