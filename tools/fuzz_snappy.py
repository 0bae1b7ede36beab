#!/usr/bin/env python3
"""Long fuzz run on the GPU box: snappy blocks through pqd_snappy_decode and the oracle.  Payloads of several kinds
(random bytes, words of a small dictionary, runs, their mixtures) and sizes from empty to a few MB, compressed by the
synthetic writer's encoder and, when pyarrow is importable, by C++ snappy; plus hand-built element streams whose copies
reach back as far as the format allows.  usage: python tools/fuzz_snappy.py [first_seed] [count]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import oracle as O  # noqa: E402
from parquet_amd import reader as R  # noqa: E402
from parquet_amd.writer import snappy_compress  # noqa: E402
from test_gpu_snappy import _apply, _copy, _lit, _uvarint  # noqa: E402

try:
    import pyarrow as pa
    CODEC = pa.Codec("snappy")
except ImportError:
    CODEC = None


def payload(rng, n):
    kind = int(rng.integers(0, 5))
    if kind == 0:
        return rng.integers(0, 256, n, dtype=np.uint8).tobytes()
    if kind == 1:
        words = [bytes(rng.integers(97, 123, int(rng.integers(1, 14)), dtype=np.uint8)) for _ in range(int(rng.integers(2, 3000)))]
        out = bytearray()
        while len(out) < n:
            w = words[int(rng.integers(0, len(words)))]
            out += len(w).to_bytes(4, "little") + w
        return bytes(out[:n])
    if kind == 2:
        m = n // 8 + 1
        return np.repeat(rng.integers(0, int(rng.integers(2, 200)), m, dtype=np.uint8), rng.integers(1, int(rng.integers(2, 300)), m))[:n].tobytes()
    if kind == 3:
        return bytes(n)
    parts = bytearray()
    while len(parts) < n:
        parts += payload(rng, int(rng.integers(1, max(2, n // 3))))
    return bytes(parts[:n])


def element_stream(rng, target):
    out, body = bytearray(), bytearray()
    far = rng.random() < 0.5
    while len(out) < target:
        if not out or rng.random() < 0.25:
            x = bytes(rng.integers(0, 256, int(rng.integers(1, 200)), dtype=np.uint8))
            body += _lit(x)
            out += x
        else:
            length = int(rng.integers(1, 65))
            hi = len(out) if far and rng.random() < 0.6 else min(len(out), int(rng.integers(1, 70000)))
            offset = int(rng.integers(1, hi + 1))
            body += _copy(length, offset, None if rng.random() < 0.8 else 4)
            _apply(out, length, offset)
    return _uvarint(len(out)) + bytes(body), bytes(out)


def one(ctx, seed):
    rng = np.random.default_rng(31000 + seed)
    blocks, wants = [], []
    for _ in range(int(rng.integers(1, 9))):
        r = rng.random()
        if r < 0.25:
            b, w = element_stream(rng, int(rng.integers(1, 300_000)))
        else:
            size = int(rng.choice([0, 1, 5, 63, 64, 65, 4096, 32767, 32768, 32769, 65536, 65537, int(rng.integers(1, 3_000_000))]))
            w = payload(rng, size)
            b = CODEC.compress(w).to_pybytes() if CODEC is not None and size and rng.random() < 0.5 else snappy_compress(w)
        blocks.append(b)
        wants.append(w)
    for b, w in zip(blocks[:2], wants[:2]):  # the oracle on a sample: the expected bytes are known by construction
        assert O.snappy_decode(b) == (w, 0), seed
    got = R.snappy_decode_blocks(ctx, blocks)
    for i, (g, w) in enumerate(zip(got, wants)):
        assert g == w, (seed, i, len(w))


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    ctx = R.Context(0)
    for s in range(first, first + count):
        one(ctx, s)
        if (s - first) % 10 == 9 or s == first + count - 1:
            print("seeds %d..%d ok" % (first, s), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
