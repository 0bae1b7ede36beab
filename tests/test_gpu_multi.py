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
        seen_rows = uploaded = 0
        for s in range(3):
            first, count, rows, off, plan = C.c_int32(), C.c_int32(), C.c_int64(), C.c_int64(), C.c_void_p()
            assert N.decode.pqd_multi_shard(m, s, C.byref(first), C.byref(count), C.byref(rows), C.byref(off), C.byref(plan)) == 0
            assert (first.value, count.value) == want[s]
            assert off.value == seen_rows
            assert rows.value == sum(pf.row_group_rows(g) for g in range(first.value, first.value + count.value))
            seen_rows += rows.value
            chunks, nc, pages, npg = pf.descriptors(first.value, count.value)
            # a shard's device holds only the bytes of its own row groups
            lo = min(pages[i].src_off for i in range(npg))
            hi = max(pages[i].src_off + pages[i].payload_len for i in range(npg))
            assert N.decode.pqd_plan_h2d_bytes(plan) == hi - lo
            uploaded += hi - lo
            view = _plan_view(plan.value, ctx, chunks, nc)
            per_rg = nc // count.value
            for k in range(nc):
                name = pf.columns[chunks[k].column][0]
                H.assert_chunk_equal(o[(first.value + k // per_rg, name)], view.column(k), (s, k))
            view._h = C.c_void_p()  # the plan belongs to the multi object
        assert seen_rows == pf.num_rows
        assert uploaded <= pf.size
    finally:
        ctx.close()
        N.decode.pqd_multi_destroy(m)


def test_decode_row_group_by_row_group_uploads_every_byte_once():
    """pqd_decode copies only the range its pages cover: a reader that goes through a file of six row groups one at a time
    (readRowGroup, template.go:320-351) moves no more than the file over the host link, SNAPPY chunks included."""
    rng = np.random.default_rng(33)
    cols = [Column("a", N.INT32, [1]), Column("s", N.BYTE_ARRAY, [1]), Column("f.x", N.INT64, [2, 1]), Column("b", N.BOOLEAN, [0])]
    for codec in (0, 1):
        buf = H.build_file(cols, 9000, 1500, 700, rng, p_null=0.2, str_len=(0, 20), codec=codec)
        pf = R.ParquetFile(buf)
        assert pf.num_row_groups == 6
        o = H.oracle_file(buf, pf)
        ctx = R.Context(0)
        try:
            total = 0
            for g in range(pf.num_row_groups):
                chunks, nc, pages, npg = pf.descriptors(g, 1)
                plan = R.decode_host(ctx, chunks, nc, pages, npg, pf.address, pf.size)
                try:
                    lo = min(pages[i].src_off for i in range(npg))
                    hi = max(pages[i].src_off + pages[i].payload_len for i in range(npg))
                    assert N.decode.pqd_plan_h2d_bytes(plan._h) == hi - lo
                    total += hi - lo
                    for k in range(nc):
                        H.assert_chunk_equal(o[(g, pf.columns[chunks[k].column][0])], plan.column(k), (codec, g, k))
                finally:
                    plan.close()
            assert total <= pf.size
        finally:
            ctx.close()


def test_bench_spawns_its_ranks():
    """`python bench.py --gpus 2` without a launcher around it starts two ranks itself (here both on the one GPU, over
    gloo) and rank 0 prints the one line."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--steps", "3", "--warmup", "1",
                          "--row-groups", "2", "--rows-per-page", "100000", "--no-extras", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=280, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [x for x in out.stdout.splitlines() if x.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["collective"]["world"] == 2
    assert len(d["collective"]["rows_per_rank"]) == 2 and sum(d["collective"]["rows_per_rank"]) == 2 * 2 * 100000


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
        for rows in (123456789012, 7):  # a gather allocates nothing: the same communicator serves any number of them
            assert N.decode.pqd_rccl_gather_rows(comm, rows, allr, C.byref(off)) == 0
            assert allr[0] == rows and off.value == 0
    finally:
        N.decode.pqd_rccl_comm_destroy(comm)
