"""GPU parity tests of the SNAPPY page path: pqd_snappy_decode on raw blocks and whole SNAPPY files through
pqd_decode, against the CPU oracle (snappy.Decode of golang/snappy v0.0.2 restated), bit exact."""
import numpy as np
import pytest

import helpers as H
import oracle as O
from parquet_amd import _native as N
from parquet_amd import reader as R
from parquet_amd.writer import Column, snappy_compress
from test_snappy_cpu import _snappy_samples

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = R.Context(0)
    yield c
    c.close()


def test_hand_derived_blocks(ctx):
    blocks = [b"\x05" + b"\x10hello",
              b"\x0a" + b"\x00a" + bytes([((9 - 4) << 2) | 1, 1]),                       # overlapping copy: a run
              b"\x0c" + b"\x0cabcd" + bytes([((8 - 1) << 2) | 2, 4, 0]),
              b"\x08" + b"\x0cabcd" + bytes([((4 - 1) << 2) | 3, 4, 0, 0, 0]),            # 4-byte offset
              b"\x00"]
    want = [b"hello", b"a" * 10, b"abcd" * 3, b"abcd" * 2, b""]
    assert [O.snappy_decode(b)[0] for b in blocks] == want
    assert R.snappy_decode_blocks(ctx, blocks) == want


def test_blocks_from_two_encoders(ctx):
    samples = _snappy_samples()
    blocks = [snappy_compress(d) for d in samples]
    try:
        import pyarrow as pa
        codec = pa.Codec("snappy")
        blocks += [codec.compress(d).to_pybytes() for d in samples if d]
        samples = samples + [d for d in samples if d]
    except ImportError:
        pass
    for b, d in zip(blocks, samples):
        assert O.snappy_decode(b) == (d, 0)
    got = R.snappy_decode_blocks(ctx, blocks)
    for i, (g, d) in enumerate(zip(got, samples)):
        assert g == d, (i, len(d))


def test_long_dependency_chains(ctx):
    """Records that each copy the one before them: every copy depends on the previous element."""
    rec = bytes(range(40))
    body = bytearray(b"\x9c" + rec)  # literal of 40
    n = 40
    for _ in range(1500):
        body += bytes([((40 - 1) << 2) | 2, 40, 0])  # copy2 length 40 offset 40
        n += 40
    pre = []
    v = n
    while v >= 0x80:
        pre.append((v & 0x7F) | 0x80)
        v >>= 7
    pre.append(v)
    block = bytes(pre) + bytes(body)
    want, err = O.snappy_decode(block)
    assert err == 0 and want == rec * 1501
    assert R.snappy_decode_blocks(ctx, [block, block]) == [want, want]


def test_corrupt_blocks_are_errors(ctx):
    bad = [b"\x05" + b"\x10hell",                                # literal runs past the input
           b"\x04" + b"\x10hello",                               # output overflow
           b"\x06" + b"\x10hello",                               # d != len(dst)
           b"\x08" + b"\x0cabcd" + bytes([(3 << 2) | 2, 0, 0]),  # offset 0
           b"\x08" + b"\x0cabcd" + bytes([(3 << 2) | 2, 5, 0]),  # offset beyond what was written
           b"\x08" + b"\x0cabcd" + bytes([(3 << 2) | 2, 4])]     # copy tag cut off
    for b in bad:
        assert O.snappy_decode(b)[0] is None
        with pytest.raises(N.PqdError) as e:
            R.snappy_decode_blocks(ctx, [b"\x05" + b"\x10hello", b])
        assert e.value.code == N.E_CORRUPT, b
    with pytest.raises(N.PqdError):
        R.snappy_decode_blocks(ctx, [b"\x80\x80\x80\x80\x10"])  # length that does not fit 32 bits


@pytest.mark.parametrize("p_null", [0.0, 0.2, 1.0])
def test_snappy_files(ctx, p_null):
    rng = np.random.default_rng(int(p_null * 10) + 40)
    cols = [Column("a", N.INT32, [1]), Column("b", N.INT64, [0]), Column("s", N.BYTE_ARRAY, [1]),
            Column("t", N.BYTE_ARRAY, [0]), Column("d", N.DOUBLE, [1]), Column("k", N.BOOLEAN, [1]),
            Column("r.x", N.INT32, [2, 1]), Column("r.s", N.BYTE_ARRAY, [2, 0])]
    buf = H.build_file(cols, 30011, 12000, 5000, rng, p_null=p_null, str_len=(0, 20), alphabet=(97, 103), codec=1)
    pf = R.ParquetFile(buf)
    assert pf.chunk_codec(0, 0) == 1
    H.check_file(buf, ctx)
    H.check_file(buf, ctx, rg_batch=1)


def test_snappy_pages_larger_than_a_tile(ctx):
    """Pages of several hundred KiB: many output tiles per page, literals that span tiles."""
    rng = np.random.default_rng(44)
    cols = [Column("i", N.INT64, [0]), Column("s", N.BYTE_ARRAY, [1]), Column("z", N.INT32, [1])]
    buf = H.build_file(cols, 120000, 120000, 60000, rng, p_null=0.1, str_len=(4, 30), alphabet=(48, 58), codec=1)
    H.check_file(buf, ctx)


def test_reader_api_on_a_snappy_file(ctx):
    rng = np.random.default_rng(45)
    cols = [Column("id", N.INT32, [0]), Column("name", N.BYTE_ARRAY, [1])]
    buf = H.build_file(cols, 3000, 1000, 400, rng, p_null=0.3, str_len=(1, 9), alphabet=(97, 123), codec=1)
    rd = R.ParquetReader(buf, ctx=ctx)
    n = 0
    while rd.next():
        assert rd.scan() is not None
        n += 1
    assert rd.error() is None and n == 3000 == rd.rows()


def _uvarint(v):
    out = []
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def test_bad_copy_with_dependents_is_an_error_not_a_hang(ctx):
    """A copy that reaches before the start of the output, followed by copies of its bytes."""
    body = b"\x0cabcd" + bytes([(3 << 2) | 2, 5, 0])   # literal 'abcd', copy2 length 4 offset 5: d < offset
    n = 8
    for _ in range(300):
        body += bytes([(3 << 2) | 2, 4, 0])           # each copies the 4 bytes before it
        n += 4
    block = _uvarint(n) + body
    assert O.snappy_decode(block)[0] is None
    with pytest.raises(N.PqdError) as e:
        R.snappy_decode_blocks(ctx, [block])
    assert e.value.code == N.E_CORRUPT


def _lit(x):
    n = len(x) - 1
    if n < 60:
        return bytes([n << 2]) + x
    if n < 256:
        return bytes([60 << 2, n]) + x
    if n < 65536:
        return bytes([61 << 2]) + n.to_bytes(2, "little") + x
    return bytes([62 << 2]) + n.to_bytes(3, "little") + x


def _copy(length, offset, kind=None):
    """One copy element; kind 1 / 2 / 4 = bytes of the offset field (1: 11-bit offset, lengths 4..11)."""
    if kind is None:
        kind = 1 if (4 <= length <= 11 and offset < 2048) else 2 if offset < 65536 else 4
    if kind == 1:
        return bytes([((offset >> 8) << 5) | ((length - 4) << 2) | 1, offset & 0xFF])
    if kind == 2:
        return bytes([((length - 1) << 2) | 2]) + offset.to_bytes(2, "little")
    return bytes([((length - 1) << 2) | 3]) + offset.to_bytes(4, "little")


def _apply(out, length, offset):
    for _ in range(length):
        out.append(out[-offset])


@pytest.mark.parametrize("first_literal", [32762, 65530])
def test_copies_across_output_tiles(ctx, first_literal):
    """Copies whose source lies in an earlier output tile of the decoder (32 KiB each), copies that run over a tile
    boundary, far 4-byte offsets; with the longer literal, what no encoder of 64 KiB blocks writes.  snappy.Decode
    reads them all."""
    rng = np.random.default_rng(3)
    a = bytes(rng.integers(0, 256, first_literal, dtype=np.uint8))
    b = bytes(rng.integers(0, 256, 10, dtype=np.uint8))
    blocks, wants = [], []
    # source in the previous tile
    blocks.append(_uvarint(first_literal + 10 + 8) + _lit(a) + _lit(b) + _copy(8, 12))
    wants.append(a + b + (a + b)[-12:-4])
    # a copy that starts 6 bytes before the boundary and ends behind it; then copies of its bytes
    out = bytearray(a)
    body = _lit(a)
    for length, offset, kind in [(40, 1000, 2), (8, 8, 1), (30, 3, 2), (64, first_literal - 100, 2), (64, first_literal, 4), (11, 2047, 1)]:
        body += _copy(length, offset, kind)
        _apply(out, length, offset)
    blocks.append(_uvarint(len(out)) + body)
    wants.append(bytes(out))
    # a run (offset 1) over two tile boundaries, written as copies of 64
    out = bytearray(b"ab")
    body = _lit(b"ab")
    while len(out) < 3 * 65536 + 100:
        body += _copy(64, 1, 2)
        _apply(out, 64, 1)
    body += _copy(64, 140000, 4)
    _apply(out, 64, 140000)
    blocks.append(_uvarint(len(out)) + body)
    wants.append(bytes(out))
    for blk, want in zip(blocks, wants):
        assert O.snappy_decode(blk) == (want, 0)
    got = R.snappy_decode_blocks(ctx, blocks)
    for i, (g, w) in enumerate(zip(got, wants)):
        assert g == w, i
    # one at a time too (a batch of one block)
    assert R.snappy_decode_blocks(ctx, blocks[1:2]) == wants[1:2]


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_random_element_streams_with_a_wide_window(ctx, seed):
    """Random literals and copies with offsets up to everything written so far (all three copy tags), about 400 KB of
    output per block: most copies reach into earlier tiles, and copies of copies form chains across them."""
    rng = np.random.default_rng(100 + seed)
    blocks, wants = [], []
    for target in (200_000, 400_000, 70_000):
        out = bytearray()
        body = bytearray()
        while len(out) < target:
            if not out or rng.random() < 0.3:
                x = bytes(rng.integers(0, 256, int(rng.integers(1, 120)), dtype=np.uint8))
                body += _lit(x)
                out += x
            else:
                length = int(rng.integers(1, 65))
                r = rng.random()
                hi = len(out) if r < 0.5 else min(len(out), 70_000) if r < 0.8 else min(len(out), 40)
                offset = int(rng.integers(1, hi + 1))
                kind = None if rng.random() < 0.8 else 4
                body += _copy(length, offset, kind)
                _apply(out, length, offset)
        blocks.append(_uvarint(len(out)) + bytes(body))
        wants.append(bytes(out))
    for blk, want in zip(blocks, wants):
        assert O.snappy_decode(blk) == (want, 0)
    got = R.snappy_decode_blocks(ctx, blocks)
    for i, (g, w) in enumerate(zip(got, wants)):
        assert g == w, (seed, i)


def test_fuzz_blocks_of_every_kind():
    """A short run of tools/fuzz_snappy.py in the suite (the long runs stay under tools/): payloads of several kinds from
    two encoders and hand-built element streams with copies across tiles, against the oracle."""
    import importlib
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    fuzz = importlib.import_module("fuzz_snappy")
    ctx = R.Context(0)
    try:
        for seed in range(12):
            fuzz.one(ctx, seed)
    finally:
        ctx.close()


def test_fuzz_blocks_with_release_acquire_flags():
    """The same with PQD_SNAPPY_STRICT=1: tiles hand their bytes over with a release store and acquire loads of the flag
    instead of device-scope stores and a wait for them.  Both orderings must decode the same bytes (the switch is read once
    per process, hence the child process)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_snappy.py"), "100", "6"], env=dict(os.environ, PQD_SNAPPY_STRICT="1"),
                         capture_output=True, text=True, timeout=250)
    assert out.returncode == 0 and "seeds 100..105 ok" in out.stdout, out.stderr[-1500:]
