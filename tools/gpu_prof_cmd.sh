#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel trace of an arbitrary python tool; prints the per-kernel table.
# usage: bash tools/gpu_prof_cmd.sh <tag> <script.py> [args...]
set -o pipefail
R="$GRAFT_REPO_ROOT"; TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf "$R/gpurun_out/prof_$TAG"; mkdir -p "$R/gpurun_out/prof_$TAG"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_$TAG" -- python3 "$R/$1" "${@:2}" > "$R/gpurun_out/prof_$TAG/run.log" 2>&1 || { tail -20 "$R/gpurun_out/prof_$TAG/run.log"; exit 1; }
python3 - "$R/gpurun_out/prof_$TAG" <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+"/*/*_kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    n=r["Name"].split("(")[0].replace("void ","").replace("pqd::","")
    print("%-40s calls %4s avg %10.1f us  %6s%%" % (n[:40], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
tail -3 "$R/gpurun_out/prof_$TAG/run.log"
