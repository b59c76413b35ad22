mkdir -p gpurun_out
for rep in 1 2 3; do for v in base l4 l5 w8; do
  if [ "$v" = base ]; then L=""; else L="--lib build/variants/lib_$v.so"; fi
  echo "== $v rep $rep"; python tools/bench_dw_regimes.py --iters 30 --only "rotating,produced" $L 2>&1 | grep -E "TOTAL|k= 83|k= 59|k= 35"
done; done > gpurun_out/r06f_fwd_occ.txt 2>&1
cat gpurun_out/r06f_fwd_occ.txt
