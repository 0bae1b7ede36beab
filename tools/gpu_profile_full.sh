#!/bin/bash
# Runs on the GPU box: three rocprofv3 passes of one bench workload (kernel trace + stats, FETCH_SIZE, WRITE_SIZE;
# counters each in a pass of their own) into gpurun_out/prof_<workload>/{kt,fetch,write}.
# usage: bash tools/gpu_profile_full.sh <workload> [bench.py arguments that set the shape, e.g. --pages 32]
set -o pipefail
R="$GRAFT_REPO_ROOT"; W=${1:-mixed}; shift
D="$R/gpurun_out/prof_$W"
cd /tmp && export TMPDIR=/tmp
rm -rf "$D"; mkdir -p "$D"
ARGS="--workload $W $* --steps 10 --warmup 2 --no-cpu-baseline --no-extras"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$D/kt" -- python3 "$R/bench.py" $ARGS > "$D/kt.log" 2>&1 || { tail -20 "$D/kt.log"; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$D/fetch" -- python3 "$R/bench.py" $ARGS > "$D/fetch.log" 2>&1 || { tail -20 "$D/fetch.log"; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$D/write" -- python3 "$R/bench.py" $ARGS > "$D/write.log" 2>&1 || { tail -20 "$D/write.log"; exit 1; }
grep '^{' "$D/kt.log" | tail -1 | cut -c1-300
