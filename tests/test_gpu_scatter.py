"""GPU tests of pqd_plan_scatter (null-mask scatter): the values of a column with definition levels at their level
slots, against what numpy derives from the CPU oracle's dense values and levels (fields.go:216-228 per level)."""
import numpy as np
import pytest

import helpers as H
from parquet_amd import _native as N
from parquet_amd import reader as R
from parquet_amd.writer import Column

pytestmark = pytest.mark.gpu

_DT = {N.INT32: np.int32, N.INT64: np.int64, N.FLOAT: np.float32, N.DOUBLE: np.float64}


@pytest.fixture(scope="module")
def ctx():
    c = R.Context(0)
    yield c
    c.close()


def _expected(o):
    """Slot arrays from an oracle chunk: (array, dtype)."""
    valid = o.defs == o.max_def
    vi = np.cumsum(valid) - valid  # value index of every level (of the next value for a null)
    if o.phys_type == N.BYTE_ARRAY:
        offs = o.offsets.astype(np.int64)
        return np.concatenate([offs[vi], offs[-1:]]).astype(np.int32)
    if o.phys_type == N.BOOLEAN:
        dense = o.values.view(np.uint8)
    else:
        dense = o.values.view(_DT[o.phys_type])
    out = np.zeros(o.defs.size, dense.dtype)
    out[valid] = dense[: int(valid.sum())]
    return out


def _check(buf, ctx):
    pf = R.ParquetFile(buf)
    orc = H.oracle_file(buf, pf)
    chunks, nc, pages, npg = pf.descriptors(0, pf.num_row_groups)
    plan = R.decode_host(ctx, chunks, nc, pages, npg, pf.address, pf.size)
    n_checked = 0
    try:
        per_rg = nc // pf.num_row_groups
        for k in range(nc):
            name = pf.columns[chunks[k].column][0]
            o = orc[(k // per_rg, name)]
            if o.max_def == 0:
                with pytest.raises(N.PqdError) as e:
                    plan.scatter(k)
                assert e.value.code == N.E_ARG
                continue
            got = plan.scatter(k)
            want = _expected(o)
            assert got.dtype == want.dtype and got.size == want.size, (name, got.dtype, got.size, want.size)
            assert np.array_equal(got.view(np.uint8), want.view(np.uint8)), name
            n_checked += 1
        # the decoded arrays are untouched by the scatter
        for k in range(nc):
            name = pf.columns[chunks[k].column][0]
            H.assert_chunk_equal(orc[(k // per_rg, name)], plan.column(k), name)
    finally:
        plan.close()
    return n_checked


@pytest.mark.parametrize("p_null", [0.0, 0.1, 0.5, 1.0])
@pytest.mark.parametrize("rows,page", [(5000, 1700), (4096, 1024), (64, 64), (17, 5), (1, 1)])
def test_scatter_optional_columns(ctx, p_null, rows, page):
    rng = np.random.default_rng(int(p_null * 10) * 100 + rows)
    cols = [Column("a", N.INT32, [1]), Column("b", N.INT64, [1]), Column("f", N.FLOAT, [1]), Column("d", N.DOUBLE, [1]),
            Column("k", N.BOOLEAN, [1]), Column("s", N.BYTE_ARRAY, [1]), Column("req", N.INT32, [0])]
    buf = H.build_file(cols, rows, rows, page, rng, p_null=p_null, str_len=(0, 12), alphabet=(97, 123))
    assert _check(buf, ctx) == 6


def test_scatter_nested_columns_and_several_row_groups(ctx):
    """Repeated columns: one slot per level (list element or empty / null list), several row groups in one plan."""
    rng = np.random.default_rng(77)
    cols = [Column("p.id", N.INT64, [1, 0]), Column("f.id", N.INT32, [2, 0]), Column("f.age", N.INT32, [2, 1]),
            Column("f.name", N.BYTE_ARRAY, [2, 1]), Column("deep.a.b", N.DOUBLE, [1, 2, 1])]
    buf = H.build_file(cols, 9000, 3000, 1100, rng, p_null=0.3, str_len=(1, 20), alphabet=(65, 91))
    assert _check(buf, ctx) == 15


def test_scatter_full_size_page(ctx):
    """One page of 1M optional int64 slots with 10 % nulls."""
    rng = np.random.default_rng(5)
    cols = [Column("x", N.INT64, [1]), Column("s", N.BYTE_ARRAY, [1])]
    buf = H.build_file(cols, 1_000_000, 1_000_000, 1_000_000, rng, p_null=0.1, str_len=(8, 32), alphabet=(97, 123))
    assert _check(buf, ctx) == 2


def test_scatter_argument_checks(ctx):
    rng = np.random.default_rng(6)
    buf = H.build_file([Column("a", N.INT32, [1])], 100, 100, 100, rng, p_null=0.2)
    pf = R.ParquetFile(buf)
    chunks, nc, pages, npg = pf.descriptors(0, 1)
    plan = R.decode_host(ctx, chunks, nc, pages, npg, pf.address, pf.size)
    try:
        dev = N.decode.pqd_device_alloc(ctx.handle, 1024)
        assert N.decode.pqd_plan_scatter(plan._h, 0, dev, 399) == N.E_ARG       # too small for 100 int32
        assert N.decode.pqd_plan_scatter(plan._h, 0, dev + 4, 1000) == N.E_ARG  # not 16-byte aligned
        assert N.decode.pqd_plan_scatter(plan._h, 1, dev, 1024) == N.E_ARG      # no such chunk
        assert N.decode.pqd_plan_scatter(plan._h, 0, None, 1024) == N.E_ARG
        assert N.decode.pqd_plan_scatter(plan._h, 0, dev, 400) == N.OK
        plan.sync()
        N.decode.pqd_device_free(ctx.handle, dev)
    finally:
        plan.close()
