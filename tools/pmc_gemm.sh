#!/bin/bash
# LDS / issue counters of the 1x1 GEMM kernels in tools/bench_gemm_io.py for one library:  tools/pmc_gemm.sh <tag> [lib.so]
tag=$1; lib=$2
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc_$tag
args="tools/bench_gemm_io.py --iters 2"
[ -n "$lib" ] && args="$args --lib $lib"
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
  n=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc_$tag/$n -o p -- python3 $args > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_$tag/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "pw_gemm" not in k and "pw_wgrad" not in k: continue
        acc[(k[:90], r.get("Grid_Size"))][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("gpurun_out/pmc_$tag/summary.txt", "w") as out:
    for k, d in sorted(acc.items()):
        out.write(f"{k[0]} grid={k[1]}\n")
        for c, v in sorted(d.items()):
            out.write(f"    {c:28s} {sum(v)/len(v):16.0f}  (n={len(v)})\n")
PY
cat gpurun_out/pmc_$tag/summary.txt | head -150
