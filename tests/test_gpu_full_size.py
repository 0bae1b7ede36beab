"""BASELINE sizes: pages of 1M rows for the string and the nested workloads (configs[2] and configs[3]), two pages per
chunk.  Every decoded array is rebuilt from the generator's inputs (levels, dense values, string offsets and
chars); one chunk of each file also goes through the CPU oracle (template_string_optional.go:62-81,
cmd/parquetgen/dremel/write_repeated.go:62-80)."""
import os
import sys

import numpy as np
import pytest

import helpers as H
from parquet_amd import _native as N
from parquet_amd import reader as R

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (the synthetic workloads of the benchmark, with what they must decode to)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = R.Context(0)
    yield c
    c.close()


def _decode_and_verify(ctx, workload, oracle_columns, pages=2):
    buf, expect = bench.build_file(workload, [0], pages, 1_000_000)
    pf = R.ParquetFile(buf)
    chunks, nc, pages, npg = pf.descriptors(0, 1)
    plan = R.decode_host(ctx, chunks, nc, pages, npg, pf.address, pf.size)
    try:
        bench.verify(plan, pf, chunks, nc, expect)  # row group 0 is kept in full by the generator
        o = H.oracle_file(buf, pf)
        for k in range(nc):
            name = pf.columns[chunks[k].column][0]
            if name in oracle_columns:
                H.assert_chunk_equal(o[(0, name)], plan.column(k), name)
    finally:
        plan.close()


def test_optional_strings_1m_slot_pages(ctx):
    """Optional string column, 8-32 bytes, 10 % nulls: 2 pages of 1M slots (about 24 MB of values each)."""
    _decode_and_verify(ctx, "optstr", {"ostr"})


def test_person_friends_1m_row_pages(ctx):
    """Person{Being, Friends []Being}: friends.id (def/rep width 1/1) and friends.age (2/1), 2 pages of 1M rows."""
    _decode_and_verify(ctx, "nested", {"friends.age"})


def test_document_and_deep_chain_1m_row_pages(ctx):
    """The rest of the type list at full size: bools, float32, uint32 / uint64, and the level widths the Person columns
    do not reach: the Dremel Document (doc.go:5-24: names.languages.country with definition levels up to 3 and
    repetition levels up to 2, links.backward) and the five-deep chain a.b.c.d.e (definition levels up to 5: width 3),
    one page of 1M rows each (2.9M and 2.2M levels), against the generator's inputs and the oracle."""
    _decode_and_verify(ctx, "wide", {"names.languages.country", "a.b.c.d.e", "links.backward", "ob", "ou64"}, pages=1)


def test_mixed_schema_row_group(ctx):
    """One row group of the headline workload: every column family side by side in one batch."""
    buf, expect = bench.build_file("mixed", [0, 1], 1, 200_000)
    pf = R.ParquetFile(buf)
    chunks, nc, pages, npg = pf.descriptors(0, 2)
    plan = R.decode_host(ctx, chunks, nc, pages, npg, pf.address, pf.size)
    try:
        bench.verify(plan, pf, chunks, nc, expect)
        o = H.oracle_file(buf, pf)
        for k in range(nc):
            H.assert_chunk_equal(o[(k // pf.num_columns, pf.columns[chunks[k].column][0])], plan.column(k), k)
    finally:
        plan.close()
