#!/bin/bash
# Runs on the GPU box: given workloads (bench arguments in quotes) with the stock library and each variant under
# parquet_amd/lib/exp/, twice.  usage: bash tools/gpu_variants_wl.sh "tag1 tag2" "optint --pages 32" "nested --pages 16" ...
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
TAGS=$1; shift
for rep in 1 2; do
  for tag in stock $TAGS; do
    if [ $tag = stock ]; then unset PQD_LIB; else export PQD_LIB="$GRAFT_REPO_ROOT/parquet_amd/lib/exp/libpqdecode_$tag.so"; fi
    for spec in "$@"; do
      set -- "$@"
      timeout -k 10 300 python -u bench.py --workload $spec --steps 100 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/vwl.log 2>&1 || { tail -20 gpurun_out/vwl.log; exit 1; }
      grep "^{" gpurun_out/vwl.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$tag | $spec |', d['value'], 'GB/s', d['ms_per_step'], 'ms')"
    done
  done
done
