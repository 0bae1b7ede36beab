#!/usr/bin/env python3
"""Times pqd_encode_pages on pages shaped like the optional int32 workload (GPU box).
usage: encode_bench.py [pages] [rows_per_page] [p_null]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from parquet_amd import _native as N, reader as R  # noqa: E402
from parquet_amd.writer import encode_pages  # noqa: E402


def main():
    pages = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    rows = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
    p_null = float(sys.argv[3]) if len(sys.argv) > 3 else 0.1
    rng = np.random.default_rng(3)
    base = []
    for _ in range(min(pages, 4)):
        defs = (rng.random(rows) >= p_null).astype(np.uint8)
        vals = rng.integers(-2**31, 2**31 - 1, int(defs.sum()), dtype=np.int64).astype(np.int32)
        base.append({"phys_type": N.INT32, "max_def": 1, "max_rep": 0, "defs": defs, "reps": None, "values": vals})
    pg = [base[i % len(base)] for i in range(pages)]
    ctx = R.Context(0)
    ts = []
    for _ in range(4):
        t = time.perf_counter()
        out = encode_pages(ctx, pg)
        ts.append(time.perf_counter() - t)
    nbytes_in = sum(p["defs"].size + p["values"].nbytes for p in pg)
    nbytes_out = sum(len(o[0]) for o in out)
    print("pages %d x %d rows, %.0f%% nulls: %.1f MB of arrays -> %.1f MB of payloads; best call %.1f ms wall "
          "(host staging + H2D + kernels + D2H)" % (pages, rows, 100 * p_null, nbytes_in / 1e6, nbytes_out / 1e6, min(ts) * 1e3))


if __name__ == "__main__":
    main()
