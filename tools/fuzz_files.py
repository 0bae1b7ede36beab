#!/usr/bin/env python3
"""Long fuzz run on the GPU box: random files through the decoder and the oracle (the same generator as
tests/test_gpu_parity.py::test_fuzz_whole_files, more seeds, larger files).
usage: python tools/fuzz_files.py [first_seed] [count] [max_rows]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import helpers as H  # noqa: E402
from parquet_amd import _native as N  # noqa: E402
from parquet_amd import reader as R  # noqa: E402
from parquet_amd.writer import Column  # noqa: E402


def one(seed, ctx, max_rows):
    rng = np.random.default_rng(4000 + seed)
    phys = [N.INT32, N.INT64, N.FLOAT, N.DOUBLE, N.BOOLEAN, N.BYTE_ARRAY, N.BYTE_ARRAY]
    shapes = [[0], [1], [1, 1], [2], [2, 1], [1, 2, 0], [2, 2, 1]]
    cols = []
    for ci in range(int(rng.integers(2, 7))):
        t = shapes[int(rng.integers(0, len(shapes)))]
        cols.append(Column(".".join("f%dl%d" % (ci, i) for i in range(len(t))), phys[int(rng.integers(0, len(phys)))], t))
    n_rows = int(rng.integers(1, max_rows))
    group = int(rng.integers(1, n_rows + 1))
    page = int(rng.integers(1, group + 1))
    lo = int(rng.choice([0, 0, 1, 8, 50, 300]))
    hi = lo + int(rng.choice([0, 3, 30, 200, 3000]))
    alphabet = [(97, 123), (0, 256), (0, 3)][int(rng.integers(0, 3))]
    buf = H.build_file(cols, n_rows, group, page, rng, p_null=float(rng.choice([0.0, 0.05, 0.5, 1.0])), str_len=(lo, hi),
                       alphabet=alphabet, stats=bool(rng.integers(0, 2)), codec=int(rng.integers(0, 2)))
    H.check_file(buf, ctx)


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    max_rows = int(sys.argv[3]) if len(sys.argv) > 3 else 6000
    ctx = R.Context(0)
    for s in range(first, first + count):
        one(s, ctx, max_rows)
        if (s - first) % 10 == 9:
            print("seeds %d..%d ok" % (first, s), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
