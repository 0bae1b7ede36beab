#!/bin/bash
# Runs on the GPU box: kernel times of pqd_snappy_decode on the three synthetic payload kinds (64 pages x 4 MiB).
set -o pipefail
for k in text levels ints; do
  echo "== $k"
  bash "$GRAFT_REPO_ROOT/tools/gpu_prof_cmd.sh" sn_$k tools/snappy_bench.py $k 64 4 > "$GRAFT_REPO_ROOT/gpurun_out/sn_$k.txt" 2>&1 || { tail -5 "$GRAFT_REPO_ROOT/gpurun_out/sn_$k.txt"; exit 1; }
  grep -E "k_snappy|k_chain<2" "$GRAFT_REPO_ROOT/gpurun_out/sn_$k.txt"
done
