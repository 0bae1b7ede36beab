// Package person holds the struct the read benchmark decodes: the nested example of the reference's README
// (README.md:146-159).  parquetgen runs over this file twice, once with the stock templates and once with the GPU
// templates of ../../gen applied, into the packages person_cpu and person_gpu that ../read_bench_test.go imports:
//
//	parquetgen -input person.go -type Person -package person_cpu -output ../person_cpu/parquet.go
//	parquetgen -input person.go -type Person -package person_gpu -output ../person_gpu/parquet.go   (GPU templates)
//
// tools/make_bench_file.py writes a file of this schema (uncompressed pages) without a Go toolchain.
// Written without a Go toolchain at hand; not compiled.
package person

type Being struct {
	ID  int32  `parquet:"id"`
	Age *int32 `parquet:"age"`
}

type Person struct {
	Being    Being   `parquet:"being"`
	Username string  `parquet:"username"`
	Friends  []Being `parquet:"friends"`
}
