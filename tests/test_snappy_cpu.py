"""CPU tests of the snappy restatement in the oracle (pageData's SNAPPY arm, fields.go:361-371 →
snappy.Decode of golang/snappy v0.0.2).  The decoder is pinned three ways: byte streams derived by hand
from the format, round trips through an independent encoder (pyarrow's C++ snappy), and — in the build
container, where the reference tree is mounted — the reference's own SNAPPY fixture read in place."""
import os

import numpy as np
import pytest

import oracle as O

REF_FIXTURE = "/root/reference/_examples/via_parquet/people.parquet"


def test_hand_derived_streams():
    # preamble 5, literal of 5
    assert O.snappy_decode(b"\x05" + b"\x10hello") == (b"hello", 0)
    # literal 'a', then copy1 length 9 offset 1: the copy overlaps its own output (run of 10)
    assert O.snappy_decode(b"\x0a" + b"\x00a" + bytes([((9 - 4) << 2) | 1, 1])) == (b"a" * 10, 0)
    # literal 'abcd', copy2 length 8 offset 4
    assert O.snappy_decode(b"\x0c" + b"\x0cabcd" + bytes([((8 - 1) << 2) | 2, 4, 0])) == (b"abcd" * 3, 0)
    # copy4 with a small offset
    assert O.snappy_decode(b"\x08" + b"\x0cabcd" + bytes([((4 - 1) << 2) | 3, 4, 0, 0, 0])) == (b"abcd" * 2, 0)
    # copy1 with offset bits in the tag: 300 = 0b1_0010_1100 → tag bits 7..5 = 1, low byte 0x2C
    body = bytes(range(256)) + bytes(range(44))
    lit = bytes([61 << 2]) + (len(body) - 1).to_bytes(2, "little") + body
    assert O.snappy_decode(b"\xb0\x02" + lit + bytes([(1 << 5) | ((4 - 4) << 2) | 1, 0x2C])) == (body + body[:4], 0)
    # literal lengths in 1, 2, 3 extra bytes
    for n, tag in ((61, 60), (300, 61), (70000, 62)):
        data = bytes(np.random.default_rng(n).integers(0, 256, n, dtype=np.uint8))
        nb = tag - 59
        pre = []
        v = n
        while v >= 0x80:
            pre.append((v & 0x7F) | 0x80)
            v >>= 7
        pre.append(v)
        stream = bytes(pre) + bytes([tag << 2]) + (n - 1).to_bytes(nb, "little") + data
        assert O.snappy_decode(stream) == (data, 0)
    # empty block
    assert O.snappy_decode(b"\x00") == (b"", 0)


def test_corrupt_streams():
    bad = [
        b"",                                                  # no preamble
        b"\x05" + b"\x10hell",                                # literal runs past the input
        b"\x04" + b"\x10hello",                               # output overflow
        b"\x06" + b"\x10hello",                               # d != len(dst) at the end
        b"\x08" + b"\x0cabcd" + bytes([(3 << 2) | 2, 0, 0]),  # offset 0
        b"\x08" + b"\x0cabcd" + bytes([(3 << 2) | 2, 5, 0]),  # offset beyond what was written
        b"\x08" + b"\x0cabcd" + bytes([(3 << 2) | 2, 4]),     # copy tag cut off
        b"\x08" + b"\x0cabcd" + bytes([(3 << 2) | 3, 4, 0, 0, 0x80]),  # 4-byte offset that is negative as an int32
        b"\x80\x80\x80\x80\x10",                              # length that does not fit 32 bits
    ]
    for s in bad:
        out, err = O.snappy_decode(s)
        assert out is None and err < 0, s


def test_round_trip_through_independent_encoder():
    pa = pytest.importorskip("pyarrow")
    codec = pa.Codec("snappy")
    rng = np.random.default_rng(5)
    words = [b"alpha", b"beta", b"gamma", b"delta", b"", b"x" * 70]
    cases = [b"", b"a", bytes(100000), bytes(rng.integers(0, 256, 200000, dtype=np.uint8)),
             b"".join(words[i] for i in rng.integers(0, len(words), 50000)),
             np.arange(100000, dtype=np.int32).tobytes(), np.repeat(rng.integers(0, 9, 3000), 37).astype(np.int64).tobytes()]
    for data in cases:
        comp = codec.compress(data).to_pybytes()
        out, err = O.snappy_decode(comp)
        assert err == 0 and out == data, (len(data), err)


@pytest.mark.skipif(not os.path.exists(REF_FIXTURE), reason="reference tree not mounted (GPU box)")
def test_oracle_reads_the_reference_snappy_fixture_in_place():
    """The fixture written by the reference library with its default SNAPPY codec, read where it lies
    (it is not copied into this repository); expected values from pyarrow's decode of the same file."""
    import helpers as H
    from parquet_amd import reader as R
    from test_oracle_golden import GOLDEN, expected_people_chunk
    buf = open(REF_FIXTURE, "rb").read()
    exp = np.load(os.path.join(GOLDEN, "people_expected.npz"), allow_pickle=False)
    pf = R.ParquetFile(buf)
    assert all(pf.chunk_codec(g, c) == 1 for g in range(pf.num_row_groups) for c in range(pf.num_columns))
    o = H.oracle_file(buf, pf)
    rows = exp["row_group_rows"]
    r0 = 0
    for g in range(2):
        r1 = r0 + int(rows[g])
        for name, ci in pf.columns:
            ch = o[(g, name)]
            assert ch.error == 0, (g, name, ch.error)
            e = expected_people_chunk(exp, name, r0, r1)
            if ci.max_def:
                assert np.array_equal(ch.defs, e["valid"].astype(np.uint8)), (g, name)
            if "offsets" in e:
                assert np.array_equal(ch.offsets, e["offsets"]) and np.array_equal(ch.values, e["chars"]), (g, name)
            elif ci.phys_type == 0:
                assert np.array_equal(ch.values, e["values"].astype(np.uint8)), (g, name)
            else:
                assert ch.values.tobytes() == np.ascontiguousarray(e["values"]).tobytes(), (g, name)
        r0 = r1


def _snappy_samples():
    rng = np.random.default_rng(11)
    words = [b"alpha", b"beta", b"gamma", b"delta", b"", b"x" * 70, b"0123456789abcdef" * 9]
    return [b"", b"a", b"abc", bytes(100000), bytes(rng.integers(0, 256, 200001, dtype=np.uint8)),
            b"".join(words[i] for i in rng.integers(0, len(words), 60000)),
            np.arange(100000, dtype=np.int32).tobytes(), np.repeat(rng.integers(0, 9, 3000), 37).astype(np.int64).tobytes(),
            bytes(rng.integers(0, 2, 300000, dtype=np.uint8)), b"ab" * 40000 + bytes(rng.integers(0, 256, 70000, dtype=np.uint8))]


def test_synthetic_writer_compressor_is_valid_snappy():
    """The test files' own compressor against the oracle decoder and, where available, C++ snappy."""
    from parquet_amd.writer import snappy_compress
    try:
        import pyarrow as pa
        codec = pa.Codec("snappy")
    except Exception:
        codec = None
    for data in _snappy_samples():
        comp = snappy_compress(data)
        out, err = O.snappy_decode(comp)
        assert err == 0 and out == data, len(data)
        if len(data) > 1000 and data != bytes(np.random.default_rng(11).integers(0, 256, 1, dtype=np.uint8)):
            assert len(comp) <= len(data) + len(data) // 6 + 32
        if codec is not None and data:
            assert codec.decompress(comp, decompressed_size=len(data)).to_pybytes() == data


def test_oracle_reads_snappy_files_of_the_synthetic_writer():
    import helpers as H
    from parquet_amd import _native as N
    from parquet_amd.writer import Column
    rng = np.random.default_rng(12)
    cols = [Column("a", N.INT32, [1]), Column("s", N.BYTE_ARRAY, [1]), Column("b", N.BOOLEAN, [0]),
            Column("r.x", N.INT64, [2, 1])]
    state = rng.bit_generator.state
    plain = H.build_file(cols, 5000, 2500, 700, rng, p_null=0.25, str_len=(0, 12), alphabet=(97, 101))
    rng.bit_generator.state = state
    comp = H.build_file(cols, 5000, 2500, 700, rng, p_null=0.25, str_len=(0, 12), alphabet=(97, 101), codec=1)
    assert len(comp) < len(plain)
    a, b = H.oracle_file(plain), H.oracle_file(comp)
    assert a.keys() == b.keys()
    for k in a:
        assert a[k].error == 0 and b[k].error == 0
        assert np.array_equal(a[k].defs, b[k].defs) and np.array_equal(a[k].reps, b[k].reps)
        assert np.array_equal(a[k].values, b[k].values) and np.array_equal(a[k].sizes, b[k].sizes)
