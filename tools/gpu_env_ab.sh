#!/bin/bash
# Runs on the GPU box: the bench under several environment settings, one line of result each.
# usage: [BENCH_ARGS="--workload optint --pages 32"] bash tools/gpu_env_ab.sh "VAR=1 VAR2=x" "" "VAR=2" ...   ("" = defaults)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
i=0
for setting in "$@"; do
  i=$((i+1))
  log=gpurun_out/ab_$i.log
  env $setting timeout -k 10 300 python -u bench.py ${BENCH_ARGS:-} --steps ${STEPS:-100} --warmup 3 --no-cpu-baseline --no-extras > $log 2>&1 || { tail -20 $log; exit 1; }
  python - "[$setting]" $log <<'PY'
import json, sys
l=[x for x in open(sys.argv[2]) if x.startswith("{")][-1]
d=json.loads(l)
pg=d["roofline"].get("plain_gather") or {}
print(sys.argv[1], "ms/step", d["ms_per_step"], "value", d["value"], "copy launch ms", pg.get("launch_ms_avg"))
PY
done
