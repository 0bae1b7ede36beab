"""The three ways a BYTE_ARRAY chunk can be decoded give the same arrays as the oracle: the single pass over the page
bytes (the default), the exact two passes alone (PQD_STR_PATH=twopass), and the two passes behind a single pass that
gave every chunk up (PQD_STR_PATH=giveup).  Every string test of test_gpu_parity.py runs again on the other two; the
last tests pin which contents the single pass keeps and which it hands over."""
import os

import numpy as np
import pytest

import helpers as H
import test_gpu_parity as P
from parquet_amd import _native as N
from parquet_amd import reader as R
from parquet_amd.writer import Column

pytestmark = pytest.mark.gpu

STRING_TESTS = sorted(n for n in dir(P) if n.startswith("test_strings_")) + ["test_short_def_block_on_multi_page_string_and_bool_columns"]


@pytest.fixture(scope="module", params=["twopass", "giveup"])
def path_ctx(request):
    old = os.environ.get("PQD_STR_PATH")
    os.environ["PQD_STR_PATH"] = request.param  # read by pqd_create
    c = R.Context(0)
    if old is None:
        del os.environ["PQD_STR_PATH"]
    else:
        os.environ["PQD_STR_PATH"] = old
    yield c
    c.close()


@pytest.mark.parametrize("name", STRING_TESTS)
def test_string_tests_on_the_other_paths(path_ctx, name):
    getattr(P, name)(path_ctx)


@pytest.mark.parametrize("p_null", [0.0, 0.1, 0.5])
def test_optional_strings_on_the_other_paths(path_ctx, p_null):
    P.test_optional_numeric_and_strings(path_ctx, p_null)


def _fallbacks(ctx, buf):
    pf = R.ParquetFile(buf)
    chunks, nc, pages, npg = pf.descriptors(0, pf.num_row_groups)
    plan = R.Plan(ctx, chunks, nc, pages, npg, pf.address, pf.size)
    d_src = N.decode.pqd_device_alloc(ctx.handle, pf.size)
    try:
        assert N.decode.pqd_copy_to_device(ctx.handle, d_src, pf.address, pf.size) == 0
        plan.run(d_src)
        plan.sync()
        n_str = sum(1 for k in range(nc) if chunks[k].phys_type == N.BYTE_ARRAY)
        return plan.string_fallbacks(), n_str
    finally:
        N.decode.pqd_device_free(ctx.handle, d_src)
        plan.close()


def test_single_pass_keeps_text_and_hands_over_the_rest():
    """Text of any length (empty strings, a few bytes, hundreds, longer than a tile) stays on the single pass; contents
    whose bytes look like length prefixes (a zero behind every letter, bytes from {0, 1, 2}) are given up and decoded
    by the two passes.  Both are checked against the oracle."""
    ctx = R.Context(0)
    try:
        rng = np.random.default_rng(41)
        cols = [Column("s", N.BYTE_ARRAY, [0]), Column("o", N.BYTE_ARRAY, [1])]
        for lo, hi, n in ((8, 32, 60000), (1, 12, 60000), (100, 400, 8000), (1500, 5000, 600), (1, 1, 50000)):
            buf = H.build_file(cols, n, n, n // 3, rng, p_null=0.1, str_len=(lo, hi), alphabet=(97, 123))
            H.check_file(buf, ctx)
            assert _fallbacks(ctx, buf) == (0, 2), (lo, hi)
        # two empty strings in a row are eight zero bytes, every one of which reads as a length: such runs fail the guess
        # of their tile, and a column full of them is handed over
        buf = H.build_file(cols, 60000, 60000, 20000, rng, p_null=0.1, str_len=(0, 3), alphabet=(97, 123))
        H.check_file(buf, ctx)
        assert _fallbacks(ctx, buf)[0] in (0, 1, 2)
        for alphabet in ("utf16", (0, 3)):
            buf = H.build_file(cols, 60000, 60000, 20000, rng, p_null=0.1, str_len=(8, 40), alphabet=alphabet)
            H.check_file(buf, ctx)
            assert _fallbacks(ctx, buf) == (2, 2), alphabet
    finally:
        ctx.close()


def test_single_pass_repairs_a_lone_bad_tile():
    """One string of random bytes among text: the tiles it covers fail their guess, are walked string by string inside
    the kernel, and the chunk stays on the single pass."""
    ctx = R.Context(0)
    try:
        rng = np.random.default_rng(43)
        from parquet_amd.writer import ParquetWriter
        cols = [Column("s", N.BYTE_ARRAY, [0])]
        parts = [bytes(rng.integers(97, 123, int(rng.integers(5, 40)), dtype=np.uint8)) for _ in range(40000)]
        parts[20000] = bytes([0, 0, 1, 0, 0, 2, 0, 0, 0, 3, 0, 0]) * 3  # looks like prefixes; shorter than a tile
        payload = b"".join(len(x).to_bytes(4, "little") + x for x in parts)
        w = ParquetWriter(cols, stats=False)
        w.begin_row_group()
        w.write_raw_page(0, payload, len(parts))
        w.end_row_group(len(parts))
        buf = w.finish()
        o, g = H.check_file(buf, ctx)
        assert bytes(g[(0, "s")]["chars"]) == b"".join(parts)
        assert _fallbacks(ctx, buf) == (0, 1)
    finally:
        ctx.close()


def _payload_with_prefixes_at(rng, targets, tail):
    """Text strings whose length prefixes start exactly at the page positions `targets` (ascending), with strings of
    8-32 bytes in between and `tail` more strings behind the last target."""
    parts, pos = [], 0
    for t in targets:
        while t - pos > 120:
            n = int(rng.integers(8, 33))
            parts.append(bytes(rng.integers(97, 123, n, dtype=np.uint8)))
            pos += 4 + n
        n = t - pos - 4  # this string ends where the next prefix must start
        assert n >= 0
        parts.append(bytes(rng.integers(97, 123, n, dtype=np.uint8)))
        pos = t
    for _ in range(tail):
        n = int(rng.integers(8, 33))
        parts.append(bytes(rng.integers(97, 123, n, dtype=np.uint8)))
    return parts


@pytest.mark.parametrize("path", ["single", "twopass"])
def test_length_prefixes_across_tile_and_unit_boundaries(path):
    """A length prefix with 1, 2 or 3 of its bytes behind the end of a 4 KiB tile, of a 16 KiB unit and of the page's last
    full tile; pages of odd sizes in one chunk, so that the chars of the later pages start at every alignment; a page
    that ends a few bytes into a tile.  Against the oracle, on the single pass (which must keep the chunk) and on the two."""
    from parquet_amd.writer import ParquetWriter
    old = os.environ.get("PQD_STR_PATH")
    if path == "twopass":
        os.environ["PQD_STR_PATH"] = "twopass"
    ctx = R.Context(0)
    if path == "twopass":
        if old is None:
            del os.environ["PQD_STR_PATH"]
        else:
            os.environ["PQD_STR_PATH"] = old
    try:
        rng = np.random.default_rng(47)
        cols = [Column("s", N.BYTE_ARRAY, [0])]
        w = ParquetWriter(cols, stats=False)
        w.begin_row_group()
        all_parts, n_rows = [], 0
        for page in range(4):
            targets = []
            for k, b in enumerate((4096, 8192, 12288, 16384, 20480, 32768, 49152, 65536 + 4096, 98304)):
                targets.append(b - 1 - (k + page) % 3)        # 3, 2 or 1 bytes of the prefix lie behind the boundary
            targets.append(98304 + 4096 * 3 + 4 + page)        # a prefix just behind a boundary
            parts = _payload_with_prefixes_at(rng, targets, tail=1 + page)  # the page ends a few strings into a tile
            parts.append(bytes(rng.integers(97, 123, page, dtype=np.uint8)))  # odd page sizes: every alignment of the next page's chars
            payload = b"".join(len(x).to_bytes(4, "little") + x for x in parts)
            w.write_raw_page(0, payload, len(parts))
            all_parts += parts
            n_rows += len(parts)
        w.end_row_group(n_rows)
        buf = w.finish()
        o, g = H.check_file(buf, ctx)
        assert bytes(g[(0, "s")]["chars"]) == b"".join(all_parts)
        offs = np.asarray(g[(0, "s")]["offsets"])
        assert offs[-1] == sum(len(x) for x in all_parts) and offs.size == n_rows + 1
        if path == "single":
            assert _fallbacks(ctx, buf) == (0, 1)
    finally:
        ctx.close()
