#!/usr/bin/env python3
"""Generates the committed golden fixtures from the reference's own binary fixture
_examples/via_parquet/people.parquet (SNAPPY, 4000 rows, 2 row groups, 10 flat columns).

Run in the build container only (needs /root/reference and pyarrow; neither exists on the GPU box):
    python tests/golden/make_golden.py

Outputs (data only — page payload bytes and expected column values, no reference source text):
  people_uncompressed.parquet  the same pages, each payload decompressed with pyarrow's snappy codec and
                               re-framed as an UNCOMPRESSED page (same level blocks, same PLAIN bytes)
  people_expected.npz          the columns as pyarrow (an independent reader) decodes the ORIGINAL file
  people_fixture_headers.json  literal header bytes / counts of the original file
"""
import io
import json
import os
import sys

import numpy as np
import pyarrow as pa
import pyarrow.parquet as pq

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from parquet_amd import _native as N  # noqa: E402
from parquet_amd import reader as R  # noqa: E402
from parquet_amd.writer import Column, ParquetWriter  # noqa: E402

SRC = "/root/reference/_examples/via_parquet/people.parquet"


def main():
    buf = open(SRC, "rb").read()
    pf = R.ParquetFile(buf)
    snappy = pa.Codec("snappy")
    cols = [Column(name, ci.phys_type, [ci.leaf_repetition], ci.converted_type) for name, ci in pf.columns]
    w = ParquetWriter(cols, stats=False)
    headers = []
    for rg in range(pf.num_row_groups):
        w.begin_row_group()
        for c in range(pf.num_columns):
            m = pf.chunk_meta(rg, c)
            assert pf.chunk_codec(rg, c) == 1  # SNAPPY
            for p in range(m["first_page"], m["first_page"] + m["n_pages"]):
                h = pf.page_header(p)
                a = h["pos"] + h["header_len"]
                comp = buf[a:a + h["compressed_page_size"]]
                raw = snappy.decompress(comp, decompressed_size=h["uncompressed_page_size"]).to_pybytes()
                assert len(raw) == h["uncompressed_page_size"]
                w.write_raw_page(m["column"], raw, h["num_values"])
                headers.append(h)
        w.end_row_group(pf.row_group_rows(rg))
    out = w.finish()
    open(os.path.join(HERE, "people_uncompressed.parquet"), "wb").write(out)

    t = pq.read_table(io.BytesIO(buf))
    exp = {}
    for name in t.column_names:
        col = t[name].combine_chunks()
        valid = np.array([v is not None for v in col.to_pylist()], bool)
        exp[name + ".valid"] = valid
        if pa.types.is_binary(col.type) or pa.types.is_string(col.type):
            items = [v for v in col.to_pylist() if v is not None]
            offs = np.zeros(len(items) + 1, np.int32)
            np.cumsum([len(s) for s in items], out=offs[1:])
            exp[name + ".offsets"] = offs
            exp[name + ".chars"] = np.frombuffer(b"".join(items), np.uint8)
        else:
            dense = col.drop_null()
            exp[name + ".values"] = dense.to_numpy(zero_copy_only=False)
    exp["row_group_rows"] = np.array([pf.row_group_rows(g) for g in range(pf.num_row_groups)], np.int64)
    np.savez_compressed(os.path.join(HERE, "people_expected.npz"), **exp)

    h0 = headers[0]
    json.dump({
        "source": "_examples/via_parquet/people.parquet",
        "file_bytes": len(buf), "num_rows": pf.num_rows, "row_groups": pf.num_row_groups,
        "num_pages": pf.num_pages,
        "first_header_hex": buf[4:4 + h0["header_len"]].hex(),
        "first_header": h0,
        "columns": [[n, ci.phys_type, ci.max_def, ci.max_rep, ci.converted_type] for n, ci in pf.columns],
    }, open(os.path.join(HERE, "people_fixture_headers.json"), "w"), indent=1)
    print("pages", pf.num_pages, "uncompressed file", len(out))


if __name__ == "__main__":
    main()
