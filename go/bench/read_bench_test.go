// BenchmarkRead in the shape of the reference's own (parquet_test.go:922-947): rows through Next + Scan of a
// generated reader, once with the stock generated code (CPU) and once with the GPU templates of ../gen
// applied.  The `Person` type and its two generated readers come from running parquetgen over person.go with and
// without the GPU templates (see ../run_bench.sh); this file is the harness only.
// Written without a Go toolchain at hand; not compiled.
package bench

import (
	"bytes"
	"os"
	"runtime"
	"testing"

	cpu "github.com/parsyl/parquet/go/bench/person_cpu"
	gpuread "github.com/parsyl/parquet/go/bench/person_gpu"
)

func load(b *testing.B) []byte {
	path := os.Getenv("PQ_BENCH_FILE") // an UNCOMPRESSED Person file written by the reference writer or by `python tools/make_bench_file.py out.parquet`
	buf, err := os.ReadFile(path)
	if err != nil {
		b.Skipf("PQ_BENCH_FILE not readable: %v", err)
	}
	return buf
}

func BenchmarkReadCPU(b *testing.B) {
	buf := load(b)
	b.Logf("GOMAXPROCS=%d NumCPU=%d", runtime.GOMAXPROCS(0), runtime.NumCPU())
	b.SetBytes(int64(len(buf)))
	b.ResetTimer()
	for i := 0; i < b.N; i++ {
		r, err := cpu.NewParquetReader(bytes.NewReader(buf))
		if err != nil {
			b.Fatal(err)
		}
		var p cpu.Person
		for r.Next() {
			r.Scan(&p)
		}
		if err := r.Error(); err != nil {
			b.Fatal(err)
		}
	}
}

func BenchmarkReadGPU(b *testing.B) {
	buf := load(b)
	b.SetBytes(int64(len(buf)))
	b.ResetTimer()
	for i := 0; i < b.N; i++ {
		r, err := gpuread.NewParquetReader(bytes.NewReader(buf), gpuread.WithDevice(0))
		if err != nil {
			b.Fatal(err)
		}
		var p gpuread.Person
		for r.Next() {
			r.Scan(&p)
		}
		if err := r.Error(); err != nil {
			b.Fatal(err)
		}
		r.Close()
	}
}
