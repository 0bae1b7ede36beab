"""The N>1 path on CPU: 2 ranks over gloo shard a file's row groups, each works only on its shard, one
all_gather of row counts gives the global row offsets."""
import os
import socket

import numpy as np
import pytest

import helpers as H
from parquet_amd import _native as N
from parquet_amd import reader as R
from parquet_amd import shard
from parquet_amd.writer import Column


def test_assign_row_groups_balanced_and_contiguous():
    for weights, world in (([1] * 8, 2), ([1] * 8, 8), ([5, 1, 1, 1, 1, 1], 2), ([1, 2, 3, 4, 5, 6, 7], 3), ([3], 4), ([], 2)):
        parts = shard.assign_row_groups(weights, world)
        assert len(parts) == world
        nxt = 0
        for first, count in parts:
            assert first == nxt and count >= 0
            nxt += count
        assert nxt == len(weights)
    assert shard.assign_row_groups([1] * 8, 4) == [(0, 2), (2, 2), (4, 2), (6, 2)]
    assert shard.assign_row_groups([5, 1, 1, 1, 1, 1], 2) == [(0, 1), (1, 5)]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmpdir):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(42)  # same file on every rank
        cols = [Column("a", N.INT32, [0]), Column("b", N.INT64, [1]), Column("r", N.INT32, [2])]
        buf = H.build_file(cols, 2300, 400, 150, rng)
        pf = R.ParquetFile(buf)
        parts = shard.assign_row_groups(shard.row_group_weights(pf), world)
        first, count = parts[rank]
        local_rows = sum(pf.row_group_rows(g) for g in range(first, first + count))
        # the shard's descriptors only reference the shard's pages
        chunks, nc, pages, npg = pf.descriptors(first, count)
        assert nc == count * pf.num_columns
        lo = min(pages[i].src_off for i in range(npg))
        hi = max(pages[i].src_off + pages[i].payload_len for i in range(npg))
        counts, offset = shard.gather_row_counts(local_rows)
        np.save(os.path.join(tmpdir, "r%d.npy" % rank),
                np.array([first, count, local_rows, offset, lo, hi] + counts.tolist(), np.int64))
    finally:
        dist.destroy_process_group()


def test_two_ranks_gloo(tmp_path):
    import torch.multiprocessing as mp
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    res = [np.load(tmp_path / ("r%d.npy" % r)) for r in range(world)]
    # shards are disjoint, ordered, and cover the 6 row groups
    assert res[0][0] == 0 and res[0][0] + res[0][1] == res[1][0] and res[1][0] + res[1][1] == 6
    # byte ranges of the shards do not overlap
    assert res[0][5] <= res[1][4]
    # gathered counts agree on both ranks and give the global offsets
    assert res[0][6:].tolist() == res[1][6:].tolist() == [res[0][2], res[1][2]]
    assert res[0][3] == 0 and res[1][3] == res[0][2]
    assert res[0][2] + res[1][2] == 2300
