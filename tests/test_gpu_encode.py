"""GPU parity tests of the level ENCODER (the first piece of the write path): pqd_rle_encode against the
oracle's restatement of rle.(*RLE).Write/Bytes (internal/rle/rle.go:48-156), byte for byte."""
import numpy as np
import pytest

import oracle as O
from parquet_amd import reader as R

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = R.Context(0)
    yield c
    c.close()


def _mod(m, c):
    return [i % m for i in range(c)]


# internal/rle/rle_test.go:21-71
REF_CASES = [(3, [4] * 100 + [5] * 100), (0, [0] * 10), (1, [0] * 17), (1, [1] * 17), (3, _mod(3, 100)),
             (3, _mod(3, 103)), (1, [1]), (1, [1, 0, 1, 1, 0]), (2, [1, 2, 3]), (3, [1, 2, 7]), (1, [])]


@pytest.mark.parametrize("width,vals", REF_CASES)
def test_reference_cases(ctx, width, vals):
    want = O.rle_encode(width, vals)
    got = R.rle_encode(ctx, width, vals)
    assert got == bytes(want)


def test_known_bytes(ctx):
    # seven ones then a zero: one bit-packed group; 8 ones: one RLE run; 63-group limit: two headers
    assert R.rle_encode(ctx, 1, [1] * 7 + [0]) == bytes([2, 0, 0, 0, 0x03, 0x7F])
    assert R.rle_encode(ctx, 1, [1] * 8) == bytes([2, 0, 0, 0, 0x10, 0x01])
    alt = [0, 1] * (4 * 64)  # 64 groups, never 8 equal
    enc = R.rle_encode(ctx, 1, alt)
    assert enc == bytes(O.rle_encode(1, alt))
    assert enc[4] == (63 << 1) | 1 and enc[4 + 1 + 63] == (1 << 1) | 1


def _random_levels(rng, n, width, p_run):
    """Values with runs: after every value, keep it with probability p_run."""
    hi = max(1, 1 << width)
    out = np.empty(n, np.uint8)
    v = int(rng.integers(0, hi))
    keep = rng.random(n) < p_run
    fresh = rng.integers(0, hi, n)
    for i in range(n):
        if not keep[i]:
            v = int(fresh[i])
        out[i] = v
    return out


@pytest.mark.parametrize("width", [0, 1, 2, 3, 4])
def test_random_levels_all_reference_widths(ctx, width):
    rng = np.random.default_rng(500 + width)
    for n in (1, 7, 8, 9, 15, 16, 17, 63 * 8, 63 * 8 + 1, 2047, 2048, 2049, 2055, 2056, 4096, 4103, 10000, 70001):
        for p_run in (0.0, 0.5, 0.9, 0.99, 0.9999):
            lv = _random_levels(rng, n, width, p_run)
            want = bytes(O.rle_encode(width, lv))
            got = R.rle_encode(ctx, width, lv)
            assert got == want, (width, n, p_run, len(got), len(want))


def test_runs_across_many_tiles(ctx):
    rng = np.random.default_rng(9)
    parts = [np.full(5, 1), np.full(30000, 2), np.array([0, 1, 0, 1, 1]), np.full(2048 * 3, 3), np.full(7, 1),
             np.full(2048, 0), rng.integers(0, 4, 5000), np.full(9000, 1)]
    lv = np.concatenate(parts).astype(np.uint8)
    for cut in (len(lv), len(lv) - 1, 2048 * 17, 2048 * 17 + 3):
        assert R.rle_encode(ctx, 2, lv[:cut]) == bytes(O.rle_encode(2, lv[:cut]))


@pytest.mark.parametrize("width", [5, 6, 7, 8])
def test_wide_levels_round_trip(ctx, width):
    """Widths the reference cannot pack (bitpack.Pack stops at 4): checked through the decoder."""
    rng = np.random.default_rng(600 + width)
    for n in (1, 100, 5000, 33333):
        lv = _random_levels(rng, n, width, 0.8)
        enc = R.rle_encode(ctx, width, lv)
        dec, total = R.rle_decode(ctx, width, enc, n + 8)
        assert total >= n and np.array_equal(dec[:n], lv) and not dec[n:total].any()
