#!/bin/bash
# Build A/B variants of the library that differ in one -D define of some kernels (FILES="pointwise_bf16 ...", default: depthwise):
#   tools/ab_variants.sh DW_TAPS_RESIDENT 0 48 60 84   ->  build/variants/lib_<value>.so      (VEXTRA="-DX=1 ..." adds fixed defines)
set -e
name=$1; shift
mkdir -p build/variants
make -j8 >/dev/null
for v in "$@"; do
  d=build/variants/obj_$v; mkdir -p $d
  for f in ${FILES:-depthwise depthwise_fwd_train depthwise_fwd_eval depthwise_bwd_data}; do
    /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Iinclude -Ivoice100_amd/csrc -Wno-unused-result \
      -mllvm -pragma-unroll-threshold=1000000 -fno-slp-vectorize -D$name=$v $VEXTRA -c voice100_amd/csrc/$f.hip -o $d/$f.o &
  done
  wait
  pat=$(echo ${FILES:-depthwise depthwise_fwd_train depthwise_fwd_eval depthwise_bwd_data} | sed 's/ /|/g')
  others=$(ls build/obj/*.o | grep -v -E "/($pat)\.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/lib_$v.so $others $d/*.o
done
ls -la build/variants/*.so
