#!/bin/bash
# Runs on the GPU box: the mixed bench under several stream priority assignments (main,strings,gather) and orders.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
for spec in "$@"; do
  set -- ${spec//:/ }
  echo "== PQD_PRIO=$1 PQD_ROWS_FIRST=$2"
  PQD_PRIO=$1 PQD_ROWS_FIRST=$2 timeout -k 10 300 python -u bench.py --workload mixed --steps 50 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/prio.log 2>&1 || { tail -20 gpurun_out/prio.log; exit 1; }
  grep "^{" gpurun_out/prio.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'GB/s', d['ms_per_step'], 'ms')"
done
