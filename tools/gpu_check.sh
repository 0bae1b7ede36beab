#!/bin/bash
# Runs on the GPU box: parity tests, then short benches of the three workloads. Stops at the first failure.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gputests.log 2>&1
rc=$?; tail -4 gpurun_out/gputests.log
[ $rc -ne 0 ] && { grep -E "^E|Error|FAILED" gpurun_out/gputests.log | head -30; exit $rc; }
for w in flat optint optstr nested mixed; do
  timeout -k 10 300 python -u bench.py --workload $w $( [ $w = mixed ] && echo "--row-groups ${MIXED_RG:-16}" || echo "--pages ${PAGES:-8}" ) --steps 20 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/bench_$w.log 2>&1 || { tail -20 gpurun_out/bench_$w.log; exit 1; }
  python - <<PY
import json
l=[x for x in open("gpurun_out/bench_$w.log") if x.startswith("{")][-1]
d=json.loads(l)
print("$w", "value", d["value"], "GB/s  ms/step", d["ms_per_step"], " copy", d["roofline"]["achieved"])
PY
done
