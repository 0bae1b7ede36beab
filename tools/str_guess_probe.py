#!/usr/bin/env python3
"""Diagnostics (GPU box): does the single string pass keep a page of given contents, or hand it to the two passes?
usage: python tools/str_guess_probe.py"""
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


def main():
    ctx = R.Context(0)
    for alphabet in ("utf16", (0, 256), (97, 123)):
        for n in (50, 200, 1000, 5000, 50000):
            for lo, hi in ((8, 32), (100, 400)):
                rng = np.random.default_rng(n + lo)
                buf = H.build_file([Column("s", N.BYTE_ARRAY, [0])], n, n, n, rng, p_null=0.0, str_len=(lo, hi), alphabet=alphabet)
                pf = R.ParquetFile(buf)
                chunks, nc, pgs, npg = pf.descriptors(0, pf.num_row_groups)
                plan = R.decode_host(ctx, chunks, nc, pgs, npg, pf.address, pf.size)
                print(alphabet, "n", n, "len", lo, hi, "bytes", pf.size, "fallbacks", plan.string_fallbacks(), flush=True)
                plan.close()
    ctx.close()


if __name__ == "__main__":
    main()
