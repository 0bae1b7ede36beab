"""The level decoder is built twice: 8 KiB super-tiles for small batches, 4 KiB super-tiles for large ones (the
planner picks by the batch's super-tiles against those of its longest block).  The files of the other parity tests hold few level blocks, so most of them run the 8 KiB
build; here the same kinds of input go through the 4 KiB build (forced with PQD_LV_TILE, which the planner reads
when a plan is created), and one batch large enough to select it by itself."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (the synthetic workloads of the benchmark, with what they must decode to)
import helpers as H
import oracle as O
import test_gpu_parity as P
from parquet_amd import _native as N
from parquet_amd import reader as R
from parquet_amd.writer import Column

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = R.Context(0)
    yield c
    c.close()


@pytest.fixture()
def small_tiles(monkeypatch):
    monkeypatch.setenv("PQD_LV_TILE", "4096")


def test_optional_and_nested_columns(ctx, small_tiles):
    rng = np.random.default_rng(41)
    cols = [Column("a", N.INT32, [1]), Column("s", N.BYTE_ARRAY, [1]), Column("b", N.BOOLEAN, [1]),
            Column("friends.id", N.INT32, [2, 0]), Column("friends.age", N.INT32, [2, 1]),
            Column("a.b.c.d.e", N.INT32, [1, 2, 1, 2, 1])]
    for p_null in (0.0, 0.1, 0.5, 1.0):
        buf = H.build_file(cols, 60000, 30000, 17000, rng, p_null=p_null, str_len=(0, 12))
        H.check_file(buf, ctx)


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_fuzzed_level_blocks(ctx, small_tiles, seed):
    P.test_fuzz_level_blocks_through_pages(ctx, seed)


@pytest.mark.parametrize("seed", [0, 5, 11, 17])
def test_fuzzed_files(ctx, small_tiles, seed):
    P.test_fuzz_whole_files(ctx, seed)


def test_errors_and_surplus_levels(ctx, small_tiles):
    P.test_corrupt_pages_return_errors_not_crashes(ctx)
    P.test_surplus_levels_equal_to_max_def_are_refused(ctx)
    P.test_short_def_block_on_multi_page_string_and_bool_columns(ctx)


def test_long_runs_across_super_tiles(ctx, small_tiles):
    """RLE runs of millions of values and bit-packed runs of thousands of groups: runs that start in one super-tile
    and end many super-tiles later (the 16-bit exit tables cannot hold them)."""
    rng = np.random.default_rng(43)
    n = 3_000_000
    defs = np.ones(n, np.uint8)
    defs[1_000_000:1_000_300] = 0
    defs[2_500_000:] = rng.integers(0, 2, n - 2_500_000)
    vals = rng.integers(-2**31, 2**31 - 1, int(defs.sum()), dtype=np.int64).astype(np.int32)
    from parquet_amd.writer import ParquetWriter
    w = ParquetWriter([Column("x", N.INT32, [1])], stats=False)
    w.begin_row_group()
    w.write_page(0, vals, defs=defs)
    w.end_row_group(n)
    buf = w.finish()
    w.close()
    H.check_file(buf, ctx)


def test_a_large_batch_selects_the_small_tiles_by_itself(ctx):
    """More level blocks than the planner gives to the 8 KiB build (the batch's super-tiles are more than 28 times those of its longest
    block): 112 pages of 1M optional levels (128 KB of level bytes each)."""
    buf, ex = bench.build_file("optint", list(range(7)), 16, 1_000_000)
    pf = R.ParquetFile(buf)
    chunks, nc, pgs, npg = pf.descriptors(0, pf.num_row_groups)
    plan = R.Plan(ctx, chunks, nc, pgs, npg, pf.address, pf.size)
    d_src = N.decode.pqd_device_alloc(ctx.handle, pf.size)
    try:
        assert N.decode.pqd_copy_to_device(ctx.handle, d_src, pf.address, pf.size) == 0
        plan.run(d_src)
        plan.sync()
        bench.verify(plan, pf, chunks, nc, ex)
        m = pf.chunk_meta(3, 0)
        o = O.read_chunk(buf, m["start_pos"], N.INT32, 1, 0, m["num_values"], m["total_compressed_size"])
        assert o.error == 0
        assert np.array_equal(plan.fetch(3, N.OUT_DEFS), o.defs)
        assert np.array_equal(plan.fetch(3, N.OUT_VALUES), o.values.view(np.uint8))
    finally:
        N.decode.pqd_device_free(ctx.handle, d_src)
        plan.close()


@pytest.mark.parametrize("tile", ["4096", "8192"])
def test_level_blocks_shared_out_over_the_xcds(ctx, monkeypatch, tile):
    """The launch order of a large batch gives every level block to one XCD and leaves positions without a super-tile
    (PQD_LV_BALANCE=2 asks for that order whatever the number of blocks): same results on optional, nested and fuzzed
    inputs, blocks of one and of many super-tiles, both builds."""
    monkeypatch.setenv("PQD_LV_TILE", tile)
    monkeypatch.setenv("PQD_LV_BALANCE", "2")
    rng = np.random.default_rng(47)
    cols = [Column("a", N.INT32, [1]), Column("s", N.BYTE_ARRAY, [1]), Column("b", N.BOOLEAN, [1]),
            Column("friends.id", N.INT32, [2, 0]), Column("friends.age", N.INT32, [2, 1]),
            Column("a.b.c.d.e", N.INT32, [1, 2, 1, 2, 1])]
    for p_null in (0.1, 0.5):
        H.check_file(H.build_file(cols, 90000, 30000, 17000, rng, p_null=p_null, str_len=(0, 12)), ctx)
    for seed in (2, 3):
        P.test_fuzz_level_blocks_through_pages(ctx, seed)
        P.test_fuzz_whole_files(ctx, seed)
    P.test_corrupt_pages_return_errors_not_crashes(ctx)
    P.test_surplus_levels_equal_to_max_def_are_refused(ctx)
