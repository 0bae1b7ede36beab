"""CPU-only tests: host logic of the library (file walk, descriptors), the synthetic writer, the oracle
on whole files, and that the C-ABI library loads and exports every declared symbol."""
import io
import os
import re

import numpy as np
import pytest

import helpers as H
import oracle as O
from parquet_amd import _native as N
from parquet_amd import reader as R
from parquet_amd.writer import Column, ParquetWriter, encode_levels, write_flat_table, REQUIRED, OPTIONAL, REPEATED

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "pqdecode.h")).read()
    declared = set(re.findall(r"\b(pqd_[a-z_0-9]+)\s*\(", header))
    assert declared, "no declarations found"
    for name in declared:
        assert hasattr(N.decode, name), name
    assert declared == set(N.DECODE_SYMBOLS), declared ^ set(N.DECODE_SYMBOLS)


def test_header_is_plain_c_and_a_c_program_links(tmp_path):
    """The boundary is a C ABI: the header compiles as C99 with no C++ or torch types in it, and a C program
    linked against the library walks a file with the host-only entry points (no device needed)."""
    import subprocess
    buf = bytes(_flat_file(np.random.default_rng(11), n=500, page=100, group=250)[2])
    (tmp_path / "f.parquet").write_bytes(buf)
    (tmp_path / "t.c").write_text(r'''
#include <stdio.h>
#include <stdlib.h>
#include "pqdecode.h"
int main(int argc, char** argv) {
    FILE* fp = fopen(argv[1], "rb");
    if (!fp) return 2;
    fseek(fp, 0, SEEK_END); long n = ftell(fp); fseek(fp, 0, SEEK_SET);
    unsigned char* b = malloc((size_t)n);
    if (fread(b, 1, (size_t)n, fp) != (size_t)n) return 3;
    pqd_file* f = NULL; char err[256];
    if (pqd_file_open(b, (size_t)n, &f, err, sizeof err) != 0) { puts(err); return 4; }
    printf("%lld %lld\n", (long long)pqd_file_num_rows(f), (long long)pqd_file_num_pages(f));
    pqd_file_close(f);
    return 0;
}
''')
    lib = os.path.join(ROOT, "parquet_amd", "lib")
    exe = str(tmp_path / "t")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), str(tmp_path / "t.c"),
                    "-o", exe, "-L", lib, "-lpqdecode", "-Wl,-rpath," + lib], check=True)
    out = subprocess.run([exe, str(tmp_path / "f.parquet")], check=True, capture_output=True, text=True).stdout.split()
    assert out[0] == "500" and int(out[1]) == 8 * 2 * 3  # 8 columns, 2 row groups of 250 rows, 3 pages of <= 100 rows each


def test_cpp_reader_mirror_builds_and_fails_loudly_without_a_device(tmp_path):
    """include/pqreader.hpp (NewParquetReader / Next / Scan / Rows / Error over the C ABI) compiles warning-free,
    rejects a file that is not Parquet, and reports the missing device instead of falling back to anything."""
    import subprocess
    lib = os.path.join(ROOT, "parquet_amd", "lib")
    exe = str(tmp_path / "read_rows")
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "read_rows.cpp"), "-o", exe, "-L", lib, "-lpqdecode", "-Wl,-rpath," + lib],
                   check=True)
    bad = tmp_path / "bad.parquet"
    bad.write_bytes(b"PAR1 not a parquet file PAR1")
    p = subprocess.run([exe, str(bad)], capture_output=True, text=True)
    assert p.returncode == 1 and p.stdout.startswith("error")
    import torch
    if not torch.cuda.is_available():
        good = tmp_path / "good.parquet"
        good.write_bytes(bytes(_flat_file(np.random.default_rng(12), n=200, page=50, group=100)[2]))
        p = subprocess.run([exe, str(good)], capture_output=True, text=True)
        assert p.returncode == 1 and "device" in p.stdout


def test_no_device_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(N.PqdError):
        R.Context(0)


def test_writer_encoder_matches_reference_encoder_restatement():
    rng = np.random.default_rng(5)
    for w in range(0, 5):
        for n in (0, 1, 5, 8, 9, 16, 63, 64, 65, 503, 504, 505, 512, 1000, 5003):
            for p in (0.0, 0.5, 0.9, 0.99):
                v = rng.integers(0, max(1 << w, 1), n, dtype=np.uint8)
                keep = rng.random(n) < p
                for i in range(1, n):
                    if keep[i]:
                        v[i] = v[i - 1]
                assert encode_levels(w, v) == O.rle_encode(w, v), (w, n, p)


def _flat_file(rng, n=3000, page=257, group=1100):
    cols = [Column("id", N.INT32, [0]), Column("age", N.INT64, [1]), Column("name", N.BYTE_ARRAY, [1]),
            Column("ok", N.BOOLEAN, [0]), Column("x", N.DOUBLE, [0]), Column("f", N.FLOAT, [1]),
            Column("flag", N.BOOLEAN, [1]), Column("s", N.BYTE_ARRAY, [0])]
    ids = rng.integers(-1000, 1000, n).astype(np.int32)
    age = rng.integers(0, 100, n).astype(np.int64)
    am = rng.random(n) > 0.1
    names = [bytes(rng.integers(97, 123, rng.integers(0, 20)).astype(np.uint8)) for _ in range(n)]
    nm = rng.random(n) > 0.3
    ok = rng.random(n) > 0.5
    x = rng.random(n)
    f = rng.random(n).astype(np.float32)
    fm = rng.random(n) > 0.5
    flag = rng.random(n) > 0.5
    flm = rng.random(n) > 0.2
    s = [bytes(rng.integers(32, 127, rng.integers(0, 9)).astype(np.uint8)) for _ in range(n)]
    data = [ids, (age, am), (names, nm), ok, x, (f, fm), (flag, flm), s]
    return cols, data, write_flat_table(cols, data, group, page)


def test_written_file_is_read_by_pyarrow():
    import pyarrow.parquet as pq
    rng = np.random.default_rng(0)
    cols, data, buf = _flat_file(rng)
    t = pq.read_table(io.BytesIO(buf))
    assert np.array_equal(t["id"].to_numpy(), data[0])
    assert t["age"].to_pylist() == [int(v) if m else None for v, m in zip(*data[1])]
    assert t["name"].to_pylist() == [v if m else None for v, m in zip(*data[2])]
    assert t["ok"].to_pylist() == [bool(v) for v in data[3]]
    assert np.array_equal(t["x"].to_numpy(), data[4])
    assert t["flag"].to_pylist() == [bool(v) if m else None for v, m in zip(*data[6])]
    assert t["s"].to_pylist() == data[7]


def test_oracle_reads_written_file():
    rng = np.random.default_rng(1)
    cols, data, buf = _flat_file(rng)
    pf = R.ParquetFile(buf)
    assert pf.num_rows == 3000 and pf.num_row_groups == 3
    o = H.oracle_file(buf, pf)
    got_ids = np.concatenate([o[(g, "id")].values.view(np.int32) for g in range(3)])
    assert np.array_equal(got_ids, data[0])
    age, am = data[1]
    got_age = np.concatenate([o[(g, "age")].values.view(np.int64) for g in range(3)])
    assert np.array_equal(got_age, age[am])
    defs = np.concatenate([o[(g, "age")].defs for g in range(3)])
    assert np.array_equal(defs, am.astype(np.uint8))
    ok = np.concatenate([o[(g, "ok")].values for g in range(3)])
    assert np.array_equal(ok, data[3].astype(np.uint8))
    names, nm = data[2]
    want = [v for v, m in zip(names, nm) if m]
    got = []
    for g in range(3):
        c = o[(g, "name")]
        got += [c.values[c.offsets[i]:c.offsets[i + 1]].tobytes() for i in range(c.n_values)]
    assert got == want
    # every chunk consumed exactly its bytes: the next chunk starts where the previous ended
    for g in range(3):
        for c in range(pf.num_columns):
            m = pf.chunk_meta(g, c)
            assert o[(g, pf.columns[c][0])].consumed == m["total_compressed_size"]


def test_file_walk_reads_pyarrow_files():
    """pyarrow as an independent writer: uncompressed, PLAIN, data page v1."""
    import pyarrow as pa
    import pyarrow.parquet as pq
    rng = np.random.default_rng(2)
    n = 5000
    a = rng.integers(-10**9, 10**9, n).astype(np.int32)
    b = [None if rng.random() < 0.1 else int(v) for v in rng.integers(0, 10**12, n)]
    s = [None if rng.random() < 0.2 else "s%d" % v for v in rng.integers(0, 10**6, n)]
    fl = [None if rng.random() < 0.5 else bool(v) for v in rng.integers(0, 2, n)]
    lst = [[int(x) for x in rng.integers(0, 100, rng.integers(0, 4))] if rng.random() > 0.1 else None for _ in range(n)]
    t = pa.table({"a": pa.array(a), "b": pa.array(b, pa.int64()), "s": pa.array(s, pa.string()),
                  "fl": pa.array(fl, pa.bool_()), "lst": pa.array(lst, pa.list_(pa.int64()))},
                 schema=pa.schema([pa.field("a", pa.int32(), nullable=False), pa.field("b", pa.int64()),
                                   pa.field("s", pa.string()), pa.field("fl", pa.bool_()),
                                   pa.field("lst", pa.list_(pa.int64()))]))
    sink = io.BytesIO()
    pq.write_table(t, sink, use_dictionary=False, compression="NONE", data_page_version="1.0",
                   row_group_size=2000, data_page_size=4096, write_statistics=True)
    buf = sink.getvalue()
    pf = R.ParquetFile(buf)
    assert pf.num_rows == n and pf.num_row_groups == 3
    names = [c[0] for c in pf.columns]
    assert names[:4] == ["a", "b", "s", "fl"] and names[4].startswith("lst.")
    lci = pf.columns[4][1]
    assert (lci.max_def, lci.max_rep) == (3, 1)
    o = H.oracle_file(buf, pf)
    for k, ch in o.items():
        assert ch.error == 0, k
    got_a = np.concatenate([o[(g, "a")].values.view(np.int32) for g in range(3)])
    assert np.array_equal(got_a, a)
    got_b = np.concatenate([o[(g, "b")].values.view(np.int64) for g in range(3)])
    assert got_b.tolist() == [v for v in b if v is not None]
    defs_b = np.concatenate([o[(g, "b")].defs for g in range(3)])
    assert defs_b.tolist() == [0 if v is None else 1 for v in b]
    got_fl = np.concatenate([o[(g, "fl")].values for g in range(3)])
    assert got_fl.tolist() == [int(v) for v in fl if v is not None]
    got_s = []
    for g in range(3):
        c = o[(g, "s")]
        got_s += [c.values[c.offsets[i]:c.offsets[i + 1]].tobytes().decode() for i in range(c.n_values)]
    assert got_s == [v for v in s if v is not None]
    # list column: levels reassemble to the original lists
    lname = names[4]
    defs = np.concatenate([o[(g, lname)].defs for g in range(3)])
    reps = np.concatenate([o[(g, lname)].reps for g in range(3)])
    vals = np.concatenate([o[(g, lname)].values.view(np.int64) for g in range(3)])
    lo, vo = O.row_offsets(defs, reps, 3, 1)
    assert len(lo) - 1 == n
    for r in range(0, n, 37):
        want = lst[r]
        d = defs[lo[r]:lo[r + 1]]
        if want is None:
            assert d.tolist() == [0]
        elif want == []:
            assert d.tolist() == [1]
        else:
            assert vals[vo[r]:vo[r + 1]].tolist() == want


def test_descriptors_and_page_counts():
    rng = np.random.default_rng(3)
    cols = [Column("a", N.INT32, [0]), Column("b.c", N.INT64, [1, 1]), Column("b.d", N.BYTE_ARRAY, [1, 2])]
    buf = H.build_file(cols, 1000, 400, 90, rng)
    pf = R.ParquetFile(buf)
    assert [(n, c.max_def, c.max_rep) for n, c in pf.columns] == [("a", 0, 0), ("b.c", 2, 0), ("b.d", 2, 1)]
    # 3 row groups x 3 columns x ceil(rows/90) pages
    assert pf.num_pages == 3 * (5 + 5 + 3)
    chunks, nc, pages, npg = pf.descriptors(0, 3)
    assert nc == 9 and npg == pf.num_pages
    assert [chunks[i].column for i in range(3)] == [0, 1, 2]
    assert sum(pages[i].num_values for i in range(chunks[0].first_page, chunks[0].first_page + chunks[0].n_pages)) == 400
    h = pf.page_header(0)
    assert h["type"] == 0 and h["num_values"] == 90 and h["pos"] == 4
    assert h["uncompressed_page_size"] == 360 == h["compressed_page_size"]


def test_not_a_parquet_file():
    with pytest.raises(N.PqdError) as e:
        R.ParquetFile(b"PAR1 nothing to see here")
    assert e.value.code == N.E_FORMAT


def test_reference_fixture_headers():
    """_examples/via_parquet/people.parquet (snappy): the walk needs no decompression."""
    p = os.path.join(ROOT, "tests", "golden", "people_fixture_headers.json")
    if not os.path.exists(p):
        pytest.skip("golden not generated")
    import json
    g = json.load(open(p))
    assert g["first_header_hex"].startswith("150015a00615aa062c15c8011500150615060000")


def test_c_partitioner_matches_shard_py():
    """pqd_shard_assign (the partitioner below the C ABI) against parquet_amd.shard.assign_row_groups."""
    import ctypes as C
    from parquet_amd import shard
    rng = np.random.default_rng(3)
    for _ in range(300):
        n = int(rng.integers(0, 40))
        world = int(rng.integers(1, 10))
        kind = int(rng.integers(0, 3))
        w = [1] * n if kind == 0 else [int(x) for x in rng.integers(0, 1000, n)] if kind == 1 else [int(x) for x in rng.integers(10**6, 10**9, n)]
        arr = (C.c_int64 * max(n, 1))(*w)
        first = (C.c_int32 * world)()
        count = (C.c_int32 * world)()
        assert N.decode.pqd_shard_assign(arr, n, world, first, count) == 0
        got = [(first[r], count[r]) for r in range(world)]
        assert got == shard.assign_row_groups(w, world), (w, world)
        assert sum(c for _, c in got) == n and all(got[r][0] + got[r][1] == got[r + 1][0] for r in range(world - 1))
    assert N.decode.pqd_shard_assign(None, 3, 2, (C.c_int32 * 2)(), (C.c_int32 * 2)()) == N.E_ARG
