#!/bin/bash
# Builds variants of the decoder library that differ in compile-time macros, into parquet_amd/lib/exp/ (for timing
# experiments on the GPU box: PQD_LIB=parquet_amd/lib/exp/libpqdecode_<tag>.so).
# usage: bash tools/build_variants.sh tag1=-DNAME=1 tag2="-DNAME=2 -DOTHER=3" ...
set -e
cd "$(dirname "$0")/../parquet_amd"
python build.py
mkdir -p lib/exp
SRCS="pqd_kernels.hip pqd_levels.hip pqd_levels_4k.hip pqd_encode.hip pqd_api.cpp pqd_file.cpp pqd_multi.cpp"
for spec in "$@"; do
  tag=${spec%%=*}; flags=${spec#*=}
  mkdir -p lib/exp/$tag
  for src in $SRCS; do
    ( cd csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function $flags -c $src -o ../lib/exp/$tag/${src%.*}.o ) &
    while [ "$(jobs -r | wc -l)" -ge 8 ]; do sleep 0.5; done
  done
done
wait
for spec in "$@"; do
  tag=${spec%%=*}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o lib/exp/libpqdecode_$tag.so lib/exp/$tag/*.o -ldl -lpthread
  rm -r lib/exp/$tag
done
ls -la lib/exp
