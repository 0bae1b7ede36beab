#!/usr/bin/env python3
"""Times pqd_snappy_decode on synthetic page payloads (GPU box).  usage: snappy_bench.py [kind] [pages] [page_mb]
kinds: ints (incompressible), text (strings of a small alphabet), levels (runs), mixed."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C  # noqa: E402
from parquet_amd import _native as N, reader as R  # noqa: E402
from parquet_amd.writer import snappy_compress  # noqa: E402


def payload(kind, nbytes, rng):
    if kind == "ints":
        return rng.integers(0, 256, nbytes, dtype=np.uint8).tobytes()
    if kind == "text":
        words = [bytes(rng.integers(97, 123, int(rng.integers(3, 12)), dtype=np.uint8)) for _ in range(2000)]
        out = bytearray()
        while len(out) < nbytes:
            w = words[int(rng.integers(0, len(words)))]
            out += len(w).to_bytes(4, "little") + w
        return bytes(out[:nbytes])
    if kind == "levels":
        return np.repeat(rng.integers(0, 4, nbytes // 16 + 1, dtype=np.uint8), rng.integers(1, 40, nbytes // 16 + 1))[:nbytes].tobytes()
    a = payload("ints", nbytes // 3, rng) + payload("text", nbytes // 3, rng)
    return a + payload("levels", nbytes - len(a), rng)


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "text"
    pages = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    page_mb = float(sys.argv[3]) if len(sys.argv) > 3 else 4.0
    rng = np.random.default_rng(7)
    base = [payload(kind, int(page_mb * (1 << 20)), rng) for _ in range(min(pages, 4))]
    comp = [snappy_compress(b) for b in base]
    blocks = [comp[i % len(comp)] for i in range(pages)]
    src = b"".join(blocks)
    ctx = R.Context(0)
    arr = (N.SnappyBlock * pages)()
    pos = 0
    for i, b in enumerate(blocks):
        arr[i].src_off, arr[i].comp_len = pos, len(b)
        pos += len(b)
    host = np.frombuffer(src, np.uint8)
    assert N.decode.pqd_snappy_sizes(host.ctypes.data, len(src), arr, pages) == 0
    total = 0
    for i in range(pages):
        arr[i].dst_off = total
        total += (arr[i].decoded_len + 16 + 255) // 256 * 256
    dsrc = N.decode.pqd_device_alloc(ctx.handle, len(src))
    ddst = N.decode.pqd_device_alloc(ctx.handle, total)
    assert N.decode.pqd_copy_to_device(ctx.handle, dsrc, host.ctypes.data, len(src)) == 0
    ts = []
    for _ in range(5):
        t = time.perf_counter()
        rc = N.decode.pqd_snappy_decode(ctx.handle, arr, pages, dsrc, len(src), ddst, total)
        ts.append(time.perf_counter() - t)
        assert rc == 0, rc
    out = np.zeros(arr[0].decoded_len, np.uint8)
    N.decode.pqd_copy_to_host(ctx.handle, out.ctypes.data, ddst, out.size)
    assert out.tobytes() == base[0]
    dec = sum(arr[i].decoded_len for i in range(pages))
    best = min(ts)
    print("kind %s pages %d: compressed %.1f MB -> %.1f MB (ratio %.2f); best call %.2f ms = %.1f GB/s of output (wall, incl. scratch alloc)"
          % (kind, pages, len(src) / 1e6, dec / 1e6, dec / len(src), best * 1e3, dec / best / 1e9))


if __name__ == "__main__":
    main()
