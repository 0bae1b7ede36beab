#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel trace of a short mixed bench, then the timeline of its last step.
# usage: bash tools/gpu_timeline.sh <tag> [bench.py args...]
set -o pipefail
R="$GRAFT_REPO_ROOT"; TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf "$R/gpurun_out/tl_$TAG"; mkdir -p "$R/gpurun_out/tl_$TAG"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/tl_$TAG" -- python3 "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-extras "$@" > "$R/gpurun_out/tl_$TAG/run.log" 2>&1 || { tail -20 "$R/gpurun_out/tl_$TAG/run.log"; exit 1; }
f=$(ls "$R"/gpurun_out/tl_$TAG/*/*_kernel_trace.csv | head -1)
python3 "$R/tools/step_timeline.py" "$f" | tee "$R/gpurun_out/tl_$TAG/timeline.txt"
rm -f "$f"
