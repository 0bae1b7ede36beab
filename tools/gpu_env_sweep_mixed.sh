#!/bin/bash
# Runs on the GPU box: the mixed bench under each value of one environment variable.
# usage: bash tools/gpu_env_sweep_mixed.sh VAR v1 v2 ...
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
VAR=$1; shift
for v in "$@"; do
  env "$VAR=$v" timeout -k 10 300 python -u bench.py --workload mixed --steps 50 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/sweep.log 2>&1 || { tail -20 gpurun_out/sweep.log; exit 1; }
  grep "^{" gpurun_out/sweep.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$VAR=$v', d['value'], 'GB/s', d['ms_per_step'], 'ms')"
done
