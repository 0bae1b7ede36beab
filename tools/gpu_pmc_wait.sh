#!/bin/bash
# Runs on the GPU box: one rocprofv3 --pmc pass of SQ busy / wait counters over a bench workload; prints per-kernel shares.
# usage: bash tools/gpu_pmc_wait.sh <workload> [pages|row groups]
set -o pipefail
R="$GRAFT_REPO_ROOT"; W=${1:-mixed}; P=${2:-8}
D="$R/gpurun_out/pmcw_$W"
cd /tmp && export TMPDIR=/tmp
rm -rf "$D"; mkdir -p "$D"
ARG="--pages $P"; [ "$W" = mixed ] && ARG="--row-groups $P"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d "$D" -- python3 "$R/bench.py" --workload $W $ARG --steps 3 --warmup 1 --no-cpu-baseline --no-extras > "$D/run.log" 2>&1 || { tail -20 "$D/run.log"; exit 1; }
python3 - "$D" <<'PY'
import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+"/*/*_counter_collection.csv")[0]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    agg[r["Kernel_Name"].split("(")[0].replace("void ","").replace("pqd::","")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,d in agg.items():
    g=lambda c: sum(d[c])/max(len(d[c]),1)
    wc=g("SQ_WAVE_CYCLES") or 1
    print("%-26s waves %8d wave_cyc/wave %9.0f  wait_any %4.0f%% wait_inst %4.0f%% active %4.0f%%  valu/wave %6.0f lds/wave %5.0f busy_cyc %10.0f" % (
        k[:26], g("SQ_WAVES"), wc/max(g("SQ_WAVES"),1), 100*g("SQ_WAIT_ANY")/wc, 100*g("SQ_WAIT_INST_ANY")/wc, 100*g("SQ_ACTIVE_INST_ANY")/wc,
        g("SQ_INSTS_VALU")/max(g("SQ_WAVES"),1), g("SQ_INSTS_LDS")/max(g("SQ_WAVES"),1), g("SQ_BUSY_CYCLES")))
PY
