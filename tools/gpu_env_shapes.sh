#!/bin/bash
# Runs on the GPU box: bench shapes ("workload pages row-groups") under each value of one environment variable.
# usage: bash tools/gpu_env_shapes.sh VAR "v1 v2" "optstr 16 8" ...
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
VAR=$1; VALS=$2; shift 2
for rep in 1 2; do
  for v in $VALS; do
    for spec in "$@"; do
      set -- "$@"
      a=($spec)
      env "$VAR=$v" timeout -k 10 300 python -u bench.py --workload ${a[0]} --pages ${a[1]} --row-groups ${a[2]} --steps 50 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/es.log 2>&1 || { tail -20 gpurun_out/es.log; exit 1; }
      grep "^{" gpurun_out/es.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$VAR=$v | $spec |', d['value'], 'GB/s', d['ms_per_step'], 'ms')"
    done
  done
done
