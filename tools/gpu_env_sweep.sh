#!/bin/bash
# Runs on the GPU box: the string A/B benches under each value of one environment variable.
# usage: bash tools/gpu_env_sweep.sh VAR v1 v2 ...
set -o pipefail
VAR=$1; shift
for v in "$@"; do
  echo "== $VAR=$v"
  env "$VAR=$v" bash "$GRAFT_REPO_ROOT/tools/gpu_ab_str.sh" || exit 1
done
