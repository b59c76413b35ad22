#!/bin/bash
# PW_ABLATE table of the act16 GEMMs (timing-only builds from tools/ab_variants.sh PW_ABLATE 8 16 32 64 128 160 with FILES=pointwise_bf16):
#   8 no epilogue | 16 no main loop | 32 no staging transforms / LDS stores | 64 no LDS fragment reads | 128 no barrier | 160 = 32 + 128
echo "== full kernels"; python tools/bench_gemm_io.py 2>&1 | grep -E "C=|TOTAL"
for v in 8 16 32 64 128 160; do echo "== PW_ABLATE=$v"; python tools/bench_gemm_io.py --lib build/variants/lib_$v.so 2>&1 | grep -E "C= 512" | cut -c1-92; done
