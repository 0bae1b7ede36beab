#!/bin/bash
# Runs on the GPU box: the wide workload at several batch sizes and with both level tile sizes.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
for tile in 0 4096 8192; do
  for shape in "2 2" "2 8" "4 8" "2 32"; do
    set -- $shape
    if [ $tile = 0 ]; then unset PQD_LV_TILE; else export PQD_LV_TILE=$tile; fi
    timeout -k 10 300 python -u bench.py --workload wide --pages $1 --row-groups $2 --steps 50 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/wide.log 2>&1 || { tail -20 gpurun_out/wide.log; exit 1; }
    grep "^{" gpurun_out/wide.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('tile $tile | pages $1 rgs $2 |', d['value'], 'GB/s', d['ms_per_step'], 'ms')"
  done
done
