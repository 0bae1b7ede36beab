#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel trace of one bench workload; prints the per-kernel table.
# usage: bash tools/gpu_prof.sh <workload> [pages]
set -o pipefail
R="$GRAFT_REPO_ROOT"; W=${1:-optional}; P=${2:-8}
cd /tmp && export TMPDIR=/tmp
rm -rf "$R/gpurun_out/prof_$W"; mkdir -p "$R/gpurun_out/prof_$W"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_$W" -- python3 "$R/bench.py" --workload $W --pages $P --steps 10 --warmup 2 --no-cpu-baseline --no-extras > "$R/gpurun_out/prof_$W/run.log" 2>&1 || { tail -20 "$R/gpurun_out/prof_$W/run.log"; exit 1; }
python3 - "$R/gpurun_out/prof_$W" <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+"/*/*_kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    n=r["Name"].split("(")[0].replace("void ","").replace("pqd::","")
    print("%-40s calls %4s avg %10.1f us  %6s%%" % (n[:40], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
grep '^{' "$R/gpurun_out/prof_$W/run.log" | tail -1 | cut -c1-260
