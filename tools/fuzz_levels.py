#!/usr/bin/env python3
"""Long fuzz run on the GPU box: hand-built level blocks (random runs, padded headers, widths 1..7, surplus levels)
through the page decoder and the oracle: the generator of tests/test_gpu_parity.py::test_fuzz_level_blocks_through_pages
with more seeds and, with a scale > 1, blocks of hundreds of thousands of values (many super-tiles per block).
usage: python tools/fuzz_levels.py [first_seed] [count] [scale]    (PQD_LV_TILE=4096 selects the other build)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import helpers as H  # noqa: E402
import oracle as O  # noqa: E402
import test_gpu_parity as P  # noqa: E402
from parquet_amd import _native as N  # noqa: E402
from parquet_amd import reader as R  # noqa: E402
from parquet_amd.writer import OPTIONAL, REPEATED, Column, ParquetWriter  # noqa: E402


def one(ctx, seed, scale):
    rng = np.random.default_rng(100 + seed)
    shapes = [[1], [1, 1, 1], [1] * 7, [2], [2, 1, 2], [1, 2, 1, 2, 1]]
    cols = [Column(".".join("c%dn%d" % (ci, i) for i in range(len(t))), N.INT32, t) for ci, t in enumerate(shapes)]
    w = ParquetWriter(cols, stats=False)
    w.begin_row_group()
    for ci, t in enumerate(shapes):
        max_def = sum(1 for x in t if x in (OPTIONAL, REPEATED))
        max_rep = sum(1 for x in t if x == REPEATED)
        for _ in range(int(rng.integers(1, 4))):
            want = int(rng.choice([1, 4096, 20000, int(rng.integers(1, 60000))])) * scale
            dwidth = int(max_def).bit_length()
            dblock, dtotal = P._random_level_block(rng, dwidth, want)
            dblock = (len(dblock) - 4 + 2).to_bytes(4, "little") + dblock[4:] + bytes([8 << 1, 0])
            n = max(dtotal + int(rng.integers(0, 9)), 1)
            payload = b""
            if max_rep:
                rblock, _ = P._random_level_block(rng, int(max_rep).bit_length(), dtotal + 8)
                payload += rblock
            payload += dblock
            levels, _ = O.rle_read(dwidth, dblock)
            nn = int(np.count_nonzero(np.asarray(levels[:n]) == max_def))
            payload += rng.integers(-2**31, 2**31 - 1, nn, dtype=np.int64).astype(np.int32).tobytes()
            w.write_raw_page(ci, payload, n)
    w.end_row_group(1)
    buf = w.finish()
    w.close()
    H.check_file(buf, ctx)


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    scale = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    ctx = R.Context(0)
    for s in range(first, first + count):
        one(ctx, s, scale)
        if (s - first) % 10 == 9:
            print("seeds %d..%d ok" % (first, s), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
