#!/bin/bash
# Builds variants of the decoder library that differ in one macro of pqd_kernels.hip, into parquet_amd/lib/exp/ (for timing
# experiments on the GPU box: PQD_LIB=parquet_amd/lib/exp/libpqdecode_<tag>.so).
# usage: bash tools/build_variants.sh tag1=-DNAME=1 tag2="-DNAME=2 -DOTHER=3" ...
set -e
cd "$(dirname "$0")/../parquet_amd"
python build.py
mkdir -p lib/exp
for spec in "$@"; do
  tag=${spec%%=*}; flags=${spec#*=}
  ( cd csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function $flags -c pqd_kernels.hip -o ../lib/exp/k_$tag.o ) &
done
wait
for spec in "$@"; do
  tag=${spec%%=*}
  objs=$(ls lib/obj/*.o | grep -v pqd_kernels.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o lib/exp/libpqdecode_$tag.so lib/exp/k_$tag.o $objs -ldl -lpthread
  rm lib/exp/k_$tag.o
done
ls -la lib/exp
