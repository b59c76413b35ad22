#!/bin/bash
# One A/B variant of the library: tools/ab_build.sh <tag> "<-D flags>" file1 [file2 ...]   ->  build/variants/lib_<tag>.so
# (the named csrc/*.hip files rebuilt with the flags, every other object taken from build/obj; run `make` first)
set -e
tag=$1; flags=$2; shift 2
d=build/variants/obj_$tag; mkdir -p $d
for f in "$@"; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Iinclude -Ivoice100_amd/csrc -Wno-unused-result \
    -mllvm -pragma-unroll-threshold=1000000 -fno-slp-vectorize $flags -c voice100_amd/csrc/$f.hip -o $d/$f.o &
done
wait
pat=$(echo "$@" | sed 's/ /|/g')
others=$(ls build/obj/*.o | grep -v -E "/($pat)\.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/lib_$tag.so $others $d/*.o
ls -la build/variants/lib_$tag.so
