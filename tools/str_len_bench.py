#!/usr/bin/env python3
"""Decode time of one optional string column as a function of the string length (runs on the GPU box).
usage: python tools/str_len_bench.py [rows] [lo:hi ...]      (ALPHABET=lo:hi in the environment sets the byte range of the chars,
ALPHABET=utf16 puts a zero byte behind every letter, PAGES=n cuts the chunk into n pages; bytes near zero make most positions look like string lengths)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import helpers as H  # noqa: E402
from parquet_amd import _native as N  # noqa: E402
from parquet_amd import reader as R  # noqa: E402
from parquet_amd.writer import Column  # noqa: E402


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
    ranges = [tuple(int(x) for x in a.split(":")) for a in sys.argv[2:]] or [(8, 32), (40, 60), (61, 100), (100, 400), (1000, 3000)]
    alphabet = os.environ.get("ALPHABET", "97:123")
    alphabet = alphabet if alphabet == "utf16" else tuple(int(x) for x in alphabet.split(":"))
    pages = int(os.environ.get("PAGES", "4"))  # pages of the column chunk
    ctx = R.Context(0)
    for lo, hi in ranges:
        rng = np.random.default_rng(lo)
        n = max(1000, int(rows * 20 / ((lo + hi) / 2)))  # about the same number of bytes for every range
        buf = H.build_file([Column("s", N.BYTE_ARRAY, [1])], n, n, (n + pages - 1) // pages, rng, p_null=0.1, str_len=(lo, hi), alphabet=alphabet)
        pf = R.ParquetFile(buf)
        chunks, nc, pgs, npg = pf.descriptors(0, pf.num_row_groups)
        plan = R.Plan(ctx, chunks, nc, pgs, npg, pf.address, pf.size)
        d_src = N.decode.pqd_device_alloc(ctx.handle, pf.size)
        assert N.decode.pqd_copy_to_device(ctx.handle, d_src, pf.address, pf.size) == 0
        for _ in range(2):
            plan.run(d_src)
        plan.sync()
        t0 = time.perf_counter()
        for _ in range(5):
            plan.run(d_src)
        plan.sync()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        bi, bo, _ = plan.bytes()
        print("len %d..%d: %d strings, %.1f MB file, %.3f ms per run, %.1f GB/s, chunks through the two passes: %d" %
              (lo, hi, n, pf.size / 1e6, ms, (bi + bo) / ms / 1e6, plan.string_fallbacks()), flush=True)
        plan.close()
        N.decode.pqd_device_free(ctx.handle, d_src)
    ctx.close()


if __name__ == "__main__":
    main()
