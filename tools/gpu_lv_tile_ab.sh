#!/bin/bash
# Runs on the GPU box: the level-heavy workloads with either build of k_levels forced, and with the planner's choice.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
for spec in "optint --pages 32" "nested --pages 16" "optint --pages 8" "nested --pages 8"; do
  for t in auto 4096 8192; do
    set -- $spec
    if [ $t = auto ]; then unset PQD_LV_TILE; else export PQD_LV_TILE=$t; fi
    timeout -k 10 300 python -u bench.py --workload "$@" --steps 100 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/lvab.log 2>&1 || { tail -20 gpurun_out/lvab.log; exit 1; }
    grep "^{" gpurun_out/lvab.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$spec tile=$t', d['value'], 'GB/s', d['ms_per_step'], 'ms')"
  done
done
