"""The C++ mirror of the generated reader (include/pqreader.hpp: NewParquetReader / Next / Scan / Rows / Error over
the C ABI) against the oracle: tests/cpp/read_rows.cpp scans a file row by row and hashes everything Scan hands
out; the same hash is computed here from the oracle's column chunks."""
import os
import subprocess

import numpy as np
import pytest

import helpers as H
from parquet_amd import _native as N
from parquet_amd import reader as R
from parquet_amd.writer import Column

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "parquet_amd", "lib", "pq_read_rows")
MASK = (1 << 64) - 1


def _fnv(h, data):
    for b in bytes(data):
        h = ((h ^ b) * 1099511628211) & MASK
    return h


def _expected(buf):
    """Per column: (values, levels, hash) over the rows of all row groups, in the order Scan yields them."""
    pf = R.ParquetFile(buf)
    o = H.oracle_file(buf, pf)
    out = {}
    for name, ci in pf.columns:
        h, n_values, n_levels = 1469598103934665603, 0, 0
        for rg in range(pf.num_row_groups):
            ch = o[(rg, name)]
            assert ch.error == 0
            width = {N.INT32: 4, N.FLOAT: 4, N.INT64: 8, N.DOUBLE: 8, N.BOOLEAN: 1}.get(ci.phys_type, 0)
            vals = np.asarray(ch.values, np.uint8)
            if ci.max_def == 0:
                n_rows = len(ch.offsets) - 1 if ci.phys_type == N.BYTE_ARRAY else len(vals) // width
                lo = np.zeros(n_rows + 1, np.int64)
                vo = np.arange(n_rows + 1, dtype=np.int64)
            else:
                lo, vo = H.O.row_offsets(ch.defs, ch.reps, ci.max_def, ci.max_rep)
            for r in range(len(lo) - 1):
                v0, v1, l0, l1 = int(vo[r]), int(vo[r + 1]), int(lo[r]), int(lo[r + 1])
                if ci.phys_type == N.BYTE_ARRAY:
                    for k in range(v0, v1):
                        a, b = int(ch.offsets[k]), int(ch.offsets[k + 1])
                        h = _fnv(h, (b - a).to_bytes(4, "little"))
                        h = _fnv(h, vals[a:b])
                else:
                    h = _fnv(h, vals[v0 * width:v1 * width])
                if ci.max_def > 0:
                    h = _fnv(h, ch.defs[l0:l1])
                if ci.max_rep > 0:
                    h = _fnv(h, ch.reps[l0:l1])
                n_values += v1 - v0
                n_levels += l1 - l0
        out[name] = (n_values, n_levels, "%016x" % h)
    return pf.num_rows, out


def _run(path):
    p = subprocess.run([EXE, path], capture_output=True, text=True, timeout=120)
    return p.returncode, p.stdout.strip().splitlines()


def test_scan_rows_like_the_generated_reader(tmp_path):
    assert os.path.exists(EXE), "build first: python __graft_entry__.py"
    rng = np.random.default_rng(77)
    cols = [Column("id", N.INT32, [0]), Column("age", N.INT64, [1]), Column("name", N.BYTE_ARRAY, [1]),
            Column("ok", N.BOOLEAN, [0]), Column("score", N.DOUBLE, [1]), Column("tags", N.BYTE_ARRAY, [2]),
            Column("friends.id", N.INT32, [2, 0]), Column("friends.age", N.FLOAT, [2, 1]), Column("flag", N.BOOLEAN, [1])]
    for codec in (0, 1):
        buf = H.build_file(cols, 1500, 600, 250, rng, p_null=0.3, str_len=(0, 90), alphabet=(97, 123), codec=codec)
        path = str(tmp_path / ("f%d.parquet" % codec))
        with open(path, "wb") as f:
            f.write(bytes(buf))
        rows, want = _expected(buf)
        rc, lines = _run(path)
        assert rc == 0, lines
        assert lines[0] == "rows %d of %d" % (rows, rows)
        got = {}
        for ln in lines[1:]:
            name, nv, nl, hh = ln.split()
            got[name] = (int(nv), int(nl), hh)
        assert got == want


def test_errors_come_back_as_in_the_reference(tmp_path):
    """NewParquetReader fails on a file that is not Parquet; a page that the reference would fail on ends the
    scan with Error() set instead of crashing."""
    bad = tmp_path / "bad.parquet"
    bad.write_bytes(b"PAR1 this is not a parquet file PAR1")
    rc, lines = _run(str(bad))
    assert rc == 1 and lines and lines[0].startswith("error")
    rng = np.random.default_rng(78)
    cols = [Column("name", N.BYTE_ARRAY, [0])]
    buf = bytearray(H.build_file(cols, 300, 300, 300, rng, str_len=(5, 9), alphabet=(97, 123), stats=False))
    pf = R.ParquetFile(bytes(buf))
    h = pf.page_header(0)
    first = h["pos"] + h["header_len"]
    buf[first:first + 4] = (0x7FFFFFF0).to_bytes(4, "little")  # a string length far beyond the page
    path = tmp_path / "corrupt.parquet"
    path.write_bytes(bytes(buf))
    rc, lines = _run(str(path))
    assert rc == 1 and lines[0].startswith("error")
