#!/bin/bash
# ctc_grad / transpose times of the three routes on one box (kernel trace of the nominal step)
for v in "VOICE100_CTC_BVT=0" "V100_CTC_BVT_NW=4" "V100_CTC_BVT_NW=16" "VOICE100_CTC_BVT=0" "V100_CTC_BVT_NW=4" "V100_CTC_BVT_NW=16"; do
  export $v
  bash tools/trace_this.sh
  echo "== $v: $(head -1 gpurun_out/trace_this_breakdown.txt)"; grep -E "ctc_grad|transpose" gpurun_out/trace_this_breakdown.txt
  unset VOICE100_CTC_BVT V100_CTC_BVT_NW
done
