#!/bin/bash
# Times the reference Go reader beside the GPU reader on this machine's host cores, when a Go toolchain and the
# reference module's dependencies are available; otherwise says so and exits 0 (bench.py then reports the C
# restatement of the reader as the CPU baseline, labelled "port").
set -e
if ! command -v go >/dev/null 2>&1; then
  echo "go toolchain not found: the Go baseline cannot be measured here"; exit 0
fi
REF=${PARSYL_PARQUET:-$HOME/go/src/github.com/parsyl/parquet}
if [ ! -d "$REF" ] || ! (cd "$REF" && go list ./... >/dev/null 2>&1); then
  echo "parsyl/parquet module (with its dependencies) not found at $REF: the Go baseline cannot be measured here"; exit 0
fi
HERE=$(cd "$(dirname "$0")" && pwd)
mkdir -p "$REF/gpu" "$REF/go/bench"
cp "$HERE/gpu/pqdecode.go" "$REF/gpu/"
cp "$HERE/gen/template_gpu.go" "$REF/cmd/parquetgen/gen/"
cp "$HERE/bench/read_bench_test.go" "$REF/go/bench/"
echo "copied the shim into $REF; apply the template changes described at the top of gen/template_gpu.go, run"
echo "  go generate ./go/bench/... && PQ_BENCH_FILE=<file> go test -bench Read -benchtime 5x ./go/bench/"
echo "GOMAXPROCS=$(go env GOMAXPROCS 2>/dev/null || echo default) cores=$(nproc)"
