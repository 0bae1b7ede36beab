#!/usr/bin/env python3
"""Writes a Parquet file of the Person schema of go/bench/person/person.go (README.md:146-159 of the reference:
Person{Being{ID int32; Age *int32}; Username string; Friends []Being}) with uncompressed DATA_PAGE v1 pages, laid out
as the reference writer lays files out, for go/bench/read_bench_test.go (PQ_BENCH_FILE) where no Go writer is at hand.
usage: python tools/make_bench_file.py out.parquet [rows] [rows per row group] [rows per page]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402
from parquet_amd import _native as N  # noqa: E402
from parquet_amd.writer import Column, ParquetWriter  # noqa: E402


def main():
    out = sys.argv[1]
    rows = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
    per_group = int(sys.argv[3]) if len(sys.argv) > 3 else 100_000  # BenchmarkRead of the reference: parquet_test.go:922-947
    per_page = int(sys.argv[4]) if len(sys.argv) > 4 else 10_000
    cols = [Column("being.id", N.INT32, [0, 0]), Column("being.age", N.INT32, [0, 1]), Column("username", N.BYTE_ARRAY, [0]),
            Column("friends.id", N.INT32, [2, 0]), Column("friends.age", N.INT32, [2, 1])]
    w = ParquetWriter(cols, stats=True)
    for g0 in range(0, rows, per_group):
        g1 = min(rows, g0 + per_group)
        w.begin_row_group()
        for ci, c in enumerate(cols):
            for p0 in range(g0, g1, per_page):
                n = min(g1, p0 + per_page) - p0
                # friends.id and friends.age of a row hold the same friends: the same draw of their number for both
                v, d, r, o = bench._gen_page(N, c.path, np.random.default_rng([7, p0, ci]), n, np.random.default_rng([7, p0, 99]))
                w.write_page(ci, v, defs=d, reps=r, offsets=o)
        w.end_row_group(g1 - g0)
    buf = w.finish()
    with open(out, "wb") as f:
        f.write(bytes(buf))
    w.close()
    print("%s: %d bytes, %d rows, row groups of %d, pages of %d" % (out, len(buf), rows, per_group, per_page))


if __name__ == "__main__":
    main()
