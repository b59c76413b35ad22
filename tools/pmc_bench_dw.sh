#!/bin/bash
# HBM traffic of the depthwise FORWARD launches exactly as bench.py dispatches them (review round 3, item 2a): rocprofv3 PMC passes
# over bench.py itself -- FETCH_SIZE and WRITE_SIZE in separate runs (TCC slots), never together with a sys / hip trace -- once on
# the nominal step (no time-stretch: the nine launches of a 1024-frame batch) and once with every step stretched to 110 % (the
# variants the ~30 % time-stretched steps of the seeded sequence take).  tools/pmc_dw_json.py turns the CSVs into
# profiles/dw_fwd_pmc.json (one row per launch) and a table.
#   tools/pmc_bench_dw.sh <outdir>          (GPU box, repo root; the program itself follows `--`)
set -e
out=$1
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
COMMON="--no-presize --steps 4 --warmup 2 --windows 0 --host-contention 0 --sustained-seconds 0 --no-extras --no-cpu-baseline --no-other-configs --no-kernel-timing"
run() {  # name counter bench-args...
  name=$1; ctr=$2; shift 2
  rm -rf "$out/raw_$name"
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$out/raw_$name" -o p -- python3 bench.py $COMMON "$@" > "$out/$name.log" 2>&1 || true
  f=$(find "$out/raw_$name" -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$out/$name.csv"
  rm -rf "$out/raw_$name"
}
run nominal_fetch FETCH_SIZE --diag-no-timestretch
run nominal_write WRITE_SIZE --diag-no-timestretch
run stretch110_fetch FETCH_SIZE --diag-stretch-rate 110
run stretch110_write WRITE_SIZE --diag-stretch-rate 110
ls -la "$out"
