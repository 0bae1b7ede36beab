#!/bin/bash
# Runs on the GPU box: one rocprofv3 --pmc pass of SQ instruction counters over a bench workload; prints per-kernel averages.
# usage: bash tools/gpu_pmc_sq.sh <workload> [pages]
set -o pipefail
R="$GRAFT_REPO_ROOT"; W=${1:-optstr}; P=${2:-8}
D="$R/gpurun_out/pmc_$W"
cd /tmp && export TMPDIR=/tmp
rm -rf "$D"; mkdir -p "$D"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d "$D" -- python3 "$R/bench.py" --workload $W --pages $P --steps 3 --warmup 1 --no-cpu-baseline --no-extras > "$D/run.log" 2>&1 || { tail -20 "$D/run.log"; exit 1; }
python3 - "$D" <<'PY'
import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+"/*/*_counter_collection.csv")[0]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    agg[r["Kernel_Name"].split("(")[0].replace("void ","").replace("pqd::","")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,d in agg.items():
    w=sum(d["SQ_WAVES"])/len(d["SQ_WAVES"]) or 1
    print("%-28s waves %8d  per wave: valu %7.0f salu %7.0f lds %6.0f smem %5.0f vrd %5.0f vwr %5.0f" % (k[:28], w, *[sum(d[c])/len(d[c])/w for c in ("SQ_INSTS_VALU","SQ_INSTS_SALU","SQ_INSTS_LDS","SQ_INSTS_SMEM","SQ_INSTS_VMEM_RD","SQ_INSTS_VMEM_WR")]))
PY
