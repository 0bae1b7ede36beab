#!/bin/bash
# Runs on the GPU box: short benches of the string-heavy workloads (per-step wall time, string chunks that fell back).
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for spec in "optstr --pages 16" "mixed --row-groups ${MIXED_RG:-32}"; do
  set -- $spec
  timeout -k 10 300 python -u bench.py --workload "$@" --steps 50 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/ab_$1.log 2>&1 || { tail -20 gpurun_out/ab_$1.log; exit 1; }
  grep "^{" gpurun_out/ab_$1.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], 'GB/s', d['ms_per_step'], 'ms  fallbacks', d['string_chunks_through_two_passes'])"
done
