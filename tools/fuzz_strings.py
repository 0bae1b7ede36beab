#!/usr/bin/env python3
"""Long fuzz run on the GPU box: string pages of a few MB (hundreds of 2 KiB tiles, several hop groups and
super-groups per page) with mixed length distributions and byte alphabets, against the oracle.
usage: python tools/fuzz_strings.py [first_seed] [count]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import helpers as H  # noqa: E402
from parquet_amd import _native as N  # noqa: E402
from parquet_amd import reader as R  # noqa: E402
from parquet_amd.writer import Column, ParquetWriter  # noqa: E402

RANGES = [(0, 0), (0, 3), (1, 12), (8, 32), (40, 60), (61, 130), (100, 400), (500, 530), (1000, 3000), (2040, 2060), (5000, 40000)]
ALPHABETS = [(97, 123), (0, 256), (0, 3), (0, 1), (0, 40)]


def one(ctx, seed):
    rng = np.random.default_rng(9000 + seed)
    optional = bool(rng.integers(0, 2))
    cols = [Column("s", N.BYTE_ARRAY, [1 if optional else 0])]
    w = ParquetWriter(cols, stats=False)
    w.begin_row_group()
    n_rows = 0
    for _ in range(int(rng.integers(1, 4))):
        target = int(rng.choice([3000, 200_000, 1_200_000, 3_000_000]))
        kinds = [RANGES[int(i)] for i in rng.integers(0, len(RANGES), int(rng.integers(1, 4)))]
        lo_a, hi_a = ALPHABETS[int(rng.integers(0, len(ALPHABETS)))]
        lens = []
        total = 0
        while total < target:
            lo, hi = kinds[int(rng.integers(0, len(kinds)))]
            k = int(rng.integers(1, 400)) if hi < 1000 else int(rng.integers(1, 6))
            part = rng.integers(lo, hi + 1, k)
            lens.append(part)
            total += int(part.sum()) + 4 * k
        lens = np.concatenate(lens)
        offs = np.zeros(len(lens) + 1, np.int32)
        np.cumsum(lens, out=offs[1:])
        chars = rng.integers(lo_a, hi_a, int(offs[-1]), dtype=np.uint8) if offs[-1] else np.zeros(0, np.uint8)
        if optional:
            defs = np.ones(len(lens) + int(rng.integers(0, len(lens) // 5 + 2)), np.uint8)
            defs[rng.choice(len(defs), len(defs) - len(lens), replace=False)] = 0
            w.write_page(0, chars, defs=defs, offsets=offs)
            n_rows += len(defs)
        else:
            w.write_page(0, chars, offsets=offs)
            n_rows += len(lens)
    w.end_row_group(n_rows)
    buf = w.finish()
    w.close()
    H.check_file(buf, ctx)


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    ctx = R.Context(0)
    for s in range(first, first + count):
        one(ctx, s)
        if (s - first) % 10 == 9:
            print("seeds %d..%d ok" % (first, s), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
