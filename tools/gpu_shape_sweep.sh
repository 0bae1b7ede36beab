#!/bin/bash
# Runs on the GPU box: single-config workloads at growing batch sizes ("workload pages row-groups" per argument).
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
for spec in "$@"; do
  set -- $spec
  timeout -k 10 300 python -u bench.py --workload $1 --pages $2 --row-groups $3 --steps 50 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/shape.log 2>&1 || { tail -20 gpurun_out/shape.log; exit 1; }
  grep "^{" gpurun_out/shape.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 | pages $2 rgs $3 |', d['value'], 'GB/s', d['ms_per_step'], 'ms')"
done
