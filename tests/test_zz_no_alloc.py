"""The entry point that allocates nothing (pqd_rle_decode_noalloc) against the existing standalone level decode
(pqd_rle_decode), through the C ABI: bit-exact outputs, on two calls in a row of which the second is smaller, so
that whatever the first call left in the static scratch array must not leak into the second."""
import ctypes as C

import numpy as np
import pytest

import oracle as O
from parquet_amd import _native as N
from parquet_amd import reader as R
from parquet_amd.writer import encode_levels

pytestmark = pytest.mark.gpu


def _levels(rng, n, width, p_run):
    """Mixed runs: long equal stretches (RLE) and noisy ones (bit-packed)."""
    out = np.zeros(n, np.uint8)
    i = 0
    while i < n:
        k = int(rng.integers(1, 200))
        if rng.random() < p_run:
            out[i:i + k] = rng.integers(0, 1 << width)
        else:
            out[i:i + k] = rng.integers(0, 1 << width, min(k, n - i))
        i += k
    return out


def _noalloc(ctx, width, block, max_values):
    runs = np.frombuffer(block, np.uint8)[4:].copy()
    d_runs = N.decode.pqd_device_alloc(ctx.handle, max(runs.size, 1))
    d_out = N.decode.pqd_device_alloc(ctx.handle, max(max_values, 1))
    d_res = N.decode.pqd_device_alloc(ctx.handle, 16)
    assert d_runs and d_out and d_res
    try:
        if runs.size:
            assert N.decode.pqd_copy_to_device(ctx.handle, d_runs, runs.ctypes.data, runs.size) == 0
        poison = np.full(max(max_values, 1), 0xAB, np.uint8)
        assert N.decode.pqd_copy_to_device(ctx.handle, d_out, poison.ctypes.data, poison.size) == 0
        rc = N.decode.pqd_rle_decode_noalloc(ctx.handle, width, d_runs, runs.size, d_out, max_values, d_res)
        assert rc == 0, (rc, N.decode.pqd_last_error(ctx.handle))
        res = np.zeros(2, np.int64)
        assert N.decode.pqd_copy_to_host(ctx.handle, res.ctypes.data, d_res, 16) == 0  # waits for the stream
        out = np.zeros(max(max_values, 1), np.uint8)
        assert N.decode.pqd_copy_to_host(ctx.handle, out.ctypes.data, d_out, out.size) == 0
        return out[:max_values], int(res[0]), int(res[1])
    finally:
        for p in (d_runs, d_out, d_res):
            N.decode.pqd_device_free(ctx.handle, p)


def _compare(ctx, width, levels, max_values):
    block = encode_levels(width, levels)
    want, n_want = R.rle_decode(ctx, width, block, max_values)
    got, n_got, st = _noalloc(ctx, width, block, max_values)
    assert st == 0
    assert n_got == n_want
    n = min(n_want, max_values)
    assert np.array_equal(got[:n], want[:n])
    assert np.array_equal(want[:min(n, levels.size)], levels[:min(n, levels.size)])
    assert (got[n:] == 0xAB).all()  # nothing written past what the block holds
    # and against the oracle's rle.Read (internal/rle/rle.go:159-195): the values the block holds (bit-packed tails
    # count in full) and the first of them
    ovals, oerr = _oracle_read(width, block)
    assert oerr == 0 and n_got == ovals.size
    assert np.array_equal(got[:n], ovals[:n])


def _oracle_read(width, block):
    """(values, error) of the oracle's rle.Read over a whole block (length prefix included)."""
    try:
        vals, consumed = O.rle_read(width, block)
    except O.OracleError as e:
        return np.zeros(0, np.uint8), e.code or 1
    assert consumed == len(block)
    return vals, 0


def test_noalloc_matches_rle_decode_two_calls_second_smaller():
    rng = np.random.default_rng(77)
    ctx = R.Context(0)
    try:
        big = _levels(rng, 1_500_000, 2, 0.5)      # > 20 super-tiles of runs: the hand-off chain is exercised
        small = _levels(rng, 3_001, 2, 0.5)
        _compare(ctx, 2, big, big.size)
        _compare(ctx, 2, small, small.size)        # second call, smaller input, same static scratch
        _compare(ctx, 1, _levels(rng, 70_000, 1, 0.3), 50_000)  # truncated by max_values
        _compare(ctx, 4, _levels(rng, 10, 4, 0.0), 64)          # fewer values than max_values
    finally:
        ctx.close()


def test_noalloc_rejects_oversized_and_reports_corrupt():
    ctx = R.Context(0)
    try:
        d = N.decode.pqd_device_alloc(ctx.handle, 64)
        assert N.decode.pqd_rle_decode_noalloc(ctx.handle, 1, d, (16 << 20) + 1, d, 1, d) == N.E_ARG
        assert N.decode.pqd_rle_decode_noalloc(ctx.handle, 9, d, 4, d, 1, d) == N.E_ARG
        N.decode.pqd_device_free(ctx.handle, d)
        # a header cut off by the block end: corrupt, reported through result[1]
        bad = np.array([4, 0, 0, 0, 0x03, 0xFF, 0x80, 0x80], np.uint8).tobytes()
        _, n, st = _noalloc(ctx, 1, bad, 100)
        assert st == N.E_CORRUPT and n == 0
        assert _oracle_read(1, bad)[1] != 0  # io.EOF in the header for the reference too
        # io.EOF on the value byte of an RLE run, and on the packed bytes of a bit-packed run
        for blk in (np.array([1, 0, 0, 0, 0x10], np.uint8).tobytes(), np.array([1, 0, 0, 0, 0x03], np.uint8).tobytes()):
            assert _oracle_read(1, blk)[1] != 0
            _, n, st = _noalloc(ctx, 1, blk, 100)
            assert st == N.E_CORRUPT and n == 0
        # width 0: headers without payload bytes, values all zero (rle.go:198-201, 232-244)
        blk = np.array([3, 0, 0, 0, 0x14, 0x05, 0x08], np.uint8).tobytes()  # RLE x10, bit-packed 2 groups, RLE x4
        ovals, oerr = _oracle_read(0, blk)
        got, n, st = _noalloc(ctx, 0, blk, 64)
        assert oerr == 0 and st == 0 and n == ovals.size == 30 and not got[:n].any() and (got[n:] == 0xAB).all()
    finally:
        ctx.close()


def test_noalloc_calls_from_two_contexts_take_turns():
    """Two contexts of one GPU share the static scratch: calls queued on both without waiting in between are serialised
    by the library (each is queued behind the one before it), so both decode their own block."""
    rng = np.random.default_rng(79)
    ca, cb = R.Context(0), R.Context(0)
    try:
        jobs = []
        for ctx, n, width in ((ca, 900_000, 1), (cb, 700_000, 2), (ca, 5_000, 3), (cb, 1_200_000, 1)):
            levels = _levels(rng, n, width, 0.4)
            block = encode_levels(width, levels)
            runs = np.frombuffer(block, np.uint8)[4:].copy()
            d_runs = N.decode.pqd_device_alloc(ctx.handle, runs.size)
            d_out = N.decode.pqd_device_alloc(ctx.handle, n)
            d_res = N.decode.pqd_device_alloc(ctx.handle, 16)
            assert N.decode.pqd_copy_to_device(ctx.handle, d_runs, runs.ctypes.data, runs.size) == 0
            jobs.append((ctx, levels, block, width, d_runs, runs.size, d_out, d_res))
        for ctx, levels, block, width, d_runs, nr, d_out, d_res in jobs:  # all queued before any result is read
            assert N.decode.pqd_rle_decode_noalloc(ctx.handle, width, d_runs, nr, d_out, levels.size, d_res) == 0
        for ctx, levels, block, width, d_runs, nr, d_out, d_res in jobs:
            res = np.zeros(2, np.int64)
            out = np.zeros(levels.size, np.uint8)
            assert N.decode.pqd_copy_to_host(ctx.handle, res.ctypes.data, d_res, 16) == 0
            assert N.decode.pqd_copy_to_host(ctx.handle, out.ctypes.data, d_out, out.size) == 0
            ovals, oerr = _oracle_read(width, block)
            assert oerr == 0 and res[1] == 0 and res[0] == ovals.size
            assert np.array_equal(out, ovals[:levels.size]) and np.array_equal(out, levels)
            for p in (d_runs, d_out, d_res):
                N.decode.pqd_device_free(ctx.handle, p)
    finally:
        ca.close()
        cb.close()
