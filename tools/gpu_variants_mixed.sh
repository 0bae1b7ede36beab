#!/bin/bash
# Runs on the GPU box: the mixed and the flat bench with the stock library and with each variant under parquet_amd/lib/exp/.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
run() {
  for wl in mixed flat; do
    timeout -k 10 300 python -u bench.py --workload $wl --steps 50 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/var.log 2>&1 || { tail -20 gpurun_out/var.log; exit 1; }
    grep "^{" gpurun_out/var.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('  $wl', d['value'], 'GB/s', d['ms_per_step'], 'ms')"
  done
}
echo "== stock"; run
for tag in "$@"; do
  echo "== $tag"
  export PQD_LIB="$GRAFT_REPO_ROOT/parquet_amd/lib/exp/libpqdecode_$tag.so"
  run
done
