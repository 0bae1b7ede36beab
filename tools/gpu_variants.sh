#!/bin/bash
# Runs on the GPU box: the string A/B benches with the stock library and with each variant under parquet_amd/lib/exp/.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
echo "== stock"; PQD_OVERLAP=${PQD_OVERLAP:-1} bash tools/gpu_ab_str.sh || exit 1
for tag in "$@"; do
  echo "== $tag"
  PQD_LIB="$GRAFT_REPO_ROOT/parquet_amd/lib/exp/libpqdecode_$tag.so" bash tools/gpu_ab_str.sh || exit 1
done
