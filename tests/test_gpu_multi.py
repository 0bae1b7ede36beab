"""Multi-GPU entry points of the C ABI on the one GPU there is: two shards on two contexts of device 0, and the
RCCL row-count gather with a world of one."""
import ctypes as C

import numpy as np
import pytest

import helpers as H
from parquet_amd import _native as N
from parquet_amd import reader as R
from parquet_amd import shard
from parquet_amd.writer import Column

pytestmark = pytest.mark.gpu


def _plan_view(handle, ctx, chunks, nc):
    pl = R.Plan.__new__(R.Plan)
    pl.ctx, pl._chunks, pl.n_chunks, pl._h = ctx, chunks, nc, C.c_void_p(handle)
    return pl


def test_multi_decode_two_shards_on_one_device():
    rng = np.random.default_rng(31)
    cols = [Column("a", N.INT32, [1]), Column("s", N.BYTE_ARRAY, [1]), Column("f.x", N.INT64, [2, 1])]
    buf = H.build_file(cols, 9000, 1500, 700, rng, p_null=0.2, str_len=(0, 20))   # 6 row groups
    pf = R.ParquetFile(buf)
    o = H.oracle_file(buf, pf)
    devs = (C.c_int * 3)(0, 0, 0)
    m = C.c_void_p()
    assert N.decode.pqd_multi_create(devs, 3, C.byref(m)) == 0
    ctx = R.Context(0)
    try:
        rc = N.decode.pqd_multi_decode_file(m, pf.address, pf.size)
        assert rc == 0, N.decode.pqd_multi_last_error(m)
        assert N.decode.pqd_multi_shards(m) == 3
        want = shard.assign_row_groups(shard.row_group_weights(pf), 3)
        seen_rows = 0
        for s in range(3):
            first, count, rows, off, plan = C.c_int32(), C.c_int32(), C.c_int64(), C.c_int64(), C.c_void_p()
            assert N.decode.pqd_multi_shard(m, s, C.byref(first), C.byref(count), C.byref(rows), C.byref(off), C.byref(plan)) == 0
            assert (first.value, count.value) == want[s]
            assert off.value == seen_rows
            assert rows.value == sum(pf.row_group_rows(g) for g in range(first.value, first.value + count.value))
            seen_rows += rows.value
            chunks, nc, pages, npg = pf.descriptors(first.value, count.value)
            view = _plan_view(plan.value, ctx, chunks, nc)
            per_rg = nc // count.value
            for k in range(nc):
                name = pf.columns[chunks[k].column][0]
                H.assert_chunk_equal(o[(first.value + k // per_rg, name)], view.column(k), (s, k))
            view._h = C.c_void_p()  # the plan belongs to the multi object
        assert seen_rows == pf.num_rows
    finally:
        ctx.close()
        N.decode.pqd_multi_destroy(m)


def test_rccl_gather_rows_world_of_one():
    ident = (C.c_char * 128)()
    rc = N.decode.pqd_rccl_unique_id(ident)
    if rc == N.E_UNSUPPORTED:
        pytest.skip("librccl not found")
    assert rc == 0
    comm = C.c_void_p()
    assert N.decode.pqd_rccl_comm_create(0, 0, 1, ident, C.byref(comm)) == 0
    try:
        allr = (C.c_int64 * 1)()
        off = C.c_int64(-1)
        assert N.decode.pqd_rccl_gather_rows(0, comm, 0, 1, 123456789012, allr, C.byref(off)) == 0
        assert allr[0] == 123456789012 and off.value == 0
    finally:
        N.decode.pqd_rccl_comm_destroy(comm)
