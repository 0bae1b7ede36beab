#!/bin/bash
# Runs on the GPU box: the three rocprofv3 passes (tools/gpu_profile_full.sh) for each workload named, one after another,
# stopping at the first failure.  usage: bash tools/gpu_profile_round.sh "mixed" "optint --pages 32" "bools" ...
set -o pipefail
for spec in "$@"; do
  set -- $spec
  echo "== $spec"
  bash "$GRAFT_REPO_ROOT/tools/gpu_profile_full.sh" "$@" || exit 1
done
