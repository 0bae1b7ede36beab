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
    """Pages of several hundred KiB: many 64 KiB output tiles per page, literals that span tiles."""
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


def test_copy_across_64k_output_blocks_is_reported_unsupported(ctx):
    """Valid snappy that no block-wise encoder writes: a copy whose source lies in the previous 64 KiB of
    output.  The oracle (like snappy.Decode) reads it; the device decoder says so instead of guessing."""
    rng = np.random.default_rng(3)
    a = bytes(rng.integers(0, 256, 65530, dtype=np.uint8))
    b = bytes(rng.integers(0, 256, 10, dtype=np.uint8))
    lit = lambda x: (bytes([61 << 2]) + (len(x) - 1).to_bytes(2, "little") if len(x) > 60 else bytes([(len(x) - 1) << 2])) + x
    block = _uvarint(65530 + 10 + 8) + lit(a) + lit(b) + bytes([(7 << 2) | 2, 12, 0])
    want, err = O.snappy_decode(block)
    assert err == 0 and want == a + b + (a + b)[-12:-4]
    with pytest.raises(N.PqdError) as e:
        R.snappy_decode_blocks(ctx, [block])
    assert e.value.code == N.E_UNSUPPORTED
