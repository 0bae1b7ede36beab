"""Pins the CPU oracle to the reference's own known-answer vectors.

Sources (paths inside parsyl/parquet): internal/bitpack/bitpack_test.go:21-43,
internal/rle/rle_test.go:21-71, and the encoder/decoder text in internal/rle/rle.go.
"""
import numpy as np
import pytest

import oracle as O


# bitpack_test.go:21-43 — bytes are written there as MSB-left binary strings.
BITPACK_CASES = [
    (1, [0, 1, 1, 0, 0, 1, 1, 1], bytes([0b11100110])),
    (2, [0, 1, 2, 0, 0, 1, 2, 2], bytes([0b00100100, 0b10100100])),
    (3, [0, 1, 2, 3, 4, 5, 6, 7], bytes([0b10001000, 0b11000110, 0b11111010])),
    (4, [0, 2, 4, 7, 14, 15, 1, 0], None),  # round-trip only in the reference
]


@pytest.mark.parametrize("width,ints,data", BITPACK_CASES)
def test_bitpack_known_answers(width, ints, data):
    packed = O.bitpack_pack(width, ints)
    if data is not None:
        assert packed == data
    assert O.bitpack_unpack(width, packed) == ints
    assert O.bitpack_unpack_spec(width, packed) == ints


def test_bitpack_width4_bytes_from_pack4():
    # bitpack.go:63-74 gives these bytes for the width-4 case
    assert O.bitpack_pack(4, [0, 2, 4, 7, 14, 15, 1, 0]) == bytes([0x20, 0x74, 0xFE, 0x01])


def test_bitpack_unsupported_width_is_empty():
    # bitpack.go:86-87 / :17-18
    assert O.bitpack_unpack(5, b"\xff" * 5) == []
    assert O.bitpack_pack(0, [0] * 8) == b""


def test_unpack_all_bytes_match_spec_generalisation():
    rng = np.random.default_rng(0)
    for w in (1, 2, 3, 4):
        for _ in range(200):
            b = rng.integers(0, 256, w, dtype=np.uint8).tobytes()
            assert O.bitpack_unpack(w, b) == O.bitpack_unpack_spec(w, b)


def _mod(m, c):
    return [i % m for i in range(c)]


# rle_test.go:21-71 (round trip: decode(encode(x))[:len(x)] == x)
RLE_CASES = [
    ("rle only", 3, [4] * 100 + [5] * 100),
    ("repeated zeros", 0, [0] * 10),
    ("odd number of repeated zeros", 1, [0] * 17),
    ("odd number of repeated ones", 1, [1] * 17),
    ("bitpacking only", 3, _mod(3, 100)),
    ("more bitpacking only", 3, _mod(3, 103)),
    ("single value", 1, [1]),
    ("odd number of non-repeated values", 1, [1, 0, 1, 1, 0]),
    ("width 2", 2, [1, 2, 3]),
    ("width 3", 3, [1, 2, 7]),
]


@pytest.mark.parametrize("name,width,vals", RLE_CASES)
def test_rle_round_trip(name, width, vals):
    enc = O.rle_encode(width, vals)
    out, consumed = O.rle_read(width, enc)
    assert consumed == len(enc)
    assert out[:len(vals)].tolist() == vals
    # the bit-packed tail is padded to a multiple of 8, never truncated (rle.go:211)
    assert len(out) >= len(vals)
    assert all(v == 0 for v in out[len(vals):])


def test_rle_only_bytes():
    # rle.go:98-109,130-137,140-156: two RLE runs of 100 at width 3
    enc = O.rle_encode(3, [4] * 100 + [5] * 100)
    assert enc == bytes([6, 0, 0, 0, 0xC8, 0x01, 0x04, 0xC8, 0x01, 0x05])


def test_encoder_bitpacked_run_layout():
    # 5 values at width 1 → one bit-packed group, header (1<<1)|1 = 3, zero padded (rle.go:143-148)
    assert O.rle_encode(1, [1, 0, 1, 1, 0]) == bytes([2, 0, 0, 0, 0x03, 0b00001101])
    # width 0: ten equal values become one RLE run with a zero-byte value (rle.go:111-116)
    assert O.rle_encode(0, [0] * 10) == bytes([1, 0, 0, 0, 0x14])


def test_encoder_63_group_limit():
    # rle.go:70 — a bit-packed run holds at most 63 groups (504 values)
    vals = _mod(2, 8 * 64)
    enc = O.rle_encode(1, vals)
    body = enc[4:]
    assert body[0] == (63 << 1) | 1
    assert body[1 + 63] == (1 << 1) | 1
    out, _ = O.rle_read(1, enc)
    assert out.tolist() == vals


def test_decoder_accepts_runs_of_one():
    # the older writer behind _examples/via_parquet/people.parquet emitted RLE runs of length 1
    block = bytes([8, 0, 0, 0, 0x02, 0x01, 0x02, 0x00, 0x02, 0x01, 0x08, 0x00])
    out, consumed = O.rle_read(1, block)
    assert out.tolist() == [1, 0, 1, 0, 0, 0, 0]
    assert consumed == 12


def test_decoder_zero_pads_short_bitpacked_run():
    # rle.go:204-207: rawBytes is pre-zeroed and a short Read is not an error
    block = bytes([2, 0, 0, 0, (2 << 1) | 1, 0xFF])
    out, _ = O.rle_read(1, block)
    assert out.tolist() == [1] * 8 + [0] * 8


def test_decoder_eof_cases():
    # bit-packed header as the very last byte: bytes.Reader.Read on an exhausted reader is io.EOF
    with pytest.raises(O.OracleError) as e:
        O.rle_read(1, bytes([1, 0, 0, 0, 0x03]))
    assert e.value.code == O.EOF
    # RLE run without its value byte
    with pytest.raises(O.OracleError) as e:
        O.rle_read(1, bytes([1, 0, 0, 0, 0x10]))
    assert e.value.code == O.EOF
    # truncated LEB128
    with pytest.raises(O.OracleError):
        O.rle_read(1, bytes([1, 0, 0, 0, 0x80]))
    # missing length prefix
    with pytest.raises(O.OracleError):
        O.rle_read(1, b"\x01\x00")


def test_rle_value_byte_is_not_masked():
    # readRLE (rle.go:218-230) copies the byte as is
    out, _ = O.rle_read(1, bytes([2, 0, 0, 0, 0x06, 0xFF]))
    assert out.tolist() == [255, 255, 255]


def test_multibyte_leb128_header():
    out, _ = O.rle_read(2, bytes([3, 0, 0, 0, 0xD0, 0x0F, 0x03]))  # count 1000
    assert out.tolist() == [3] * 1000


def test_random_round_trips_all_widths():
    rng = np.random.default_rng(1)
    for w in range(0, 5):
        for n in (1, 7, 8, 9, 63, 64, 503, 504, 505, 513, 4000):
            for p_run in (0.0, 0.5, 0.95):
                hi = (1 << w) if w else 1
                vals = rng.integers(0, hi, n, dtype=np.uint8)
                # long runs with probability p_run
                keep = rng.random(n) < p_run
                for i in range(1, n):
                    if keep[i]:
                        vals[i] = vals[i - 1]
                enc = O.rle_encode(w, vals)
                out, consumed = O.rle_read(w, enc)
                assert consumed == len(enc)
                assert np.array_equal(out[:n], vals)


def test_wide_decode_matches_narrow():
    rng = np.random.default_rng(2)
    vals = rng.integers(0, 8, 3000, dtype=np.uint8)
    vals[100:400] = 5
    enc = O.rle_encode(3, vals)
    a, _ = O.rle_read(3, enc)
    b, _ = O.rle_read_u32(3, enc, 4096)
    assert np.array_equal(a.astype(np.uint32), b)


def test_row_offsets_dremel_paper():
    # cmd/parquetgen/dremel/dremel_test.go:77-84, column names.languages.country (max def 3, max rep 2)
    defs = [3, 2, 1, 3, 1]
    reps = [0, 2, 1, 1, 0]
    lo, vo = O.row_offsets(defs, reps, 3, 2)
    assert lo.tolist() == [0, 4, 5]
    assert vo.tolist() == [0, 2, 2]
    lo, vo = O.row_offsets([1, 2, 2], [0, 0, 1], 2, 1)  # link.backward
    assert lo.tolist() == [0, 1, 3]
    assert vo.tolist() == [0, 0, 2]


# ---------------------------------------------------------------- the reference's binary fixture
import json
import os

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load_people():
    buf = open(os.path.join(GOLDEN, "people_uncompressed.parquet"), "rb").read()
    exp = np.load(os.path.join(GOLDEN, "people_expected.npz"), allow_pickle=False)
    return buf, exp


def expected_people_chunk(exp, name, r0, r1):
    """Expected arrays of one column over rows [r0, r1) from the pyarrow-decoded golden."""
    valid = exp[name + ".valid"]
    v0, v1 = int(valid[:r0].sum()), int(valid[:r1].sum())
    out = {"valid": valid[r0:r1]}
    if name + ".offsets" in exp:
        offs = exp[name + ".offsets"]
        out["offsets"] = (offs[v0:v1 + 1] - offs[v0]).astype(np.int32)
        out["chars"] = exp[name + ".chars"][offs[v0]:offs[v1]]
    else:
        out["values"] = exp[name + ".values"][v0:v1]
    return out


def test_fixture_page_header_bytes():
    # the first PageHeader of _examples/via_parquet/people.parquet, byte for byte
    h = json.load(open(os.path.join(GOLDEN, "people_fixture_headers.json")))
    raw = bytes.fromhex(h["first_header_hex"])
    assert raw == bytes.fromhex("150015a00615aa062c15c8011500150615060000")
    ph = O.page_header(raw + b"\0" * 8)
    assert (ph.type, ph.uncompressed_page_size, ph.compressed_page_size, ph.num_values) == (0, 400, 405, 100)
    assert (ph.encoding, ph.def_encoding, ph.rep_encoding, ph.header_len) == (0, 3, 3, 20)
    assert h["num_pages"] == 400 and h["num_rows"] == 4000


def test_oracle_decodes_reference_fixture_pages():
    """Real page bytes written by the reference library (older encoder: RLE runs of length 1), expected
    values from pyarrow reading the original snappy file."""
    import helpers as H
    from parquet_amd import reader as R
    buf, exp = _load_people()
    pf = R.ParquetFile(buf)
    assert pf.num_rows == 4000 and pf.num_row_groups == 2 and pf.num_pages == 400
    o = H.oracle_file(buf, pf)
    rows = exp["row_group_rows"]
    r0 = 0
    for g in range(2):
        r1 = r0 + int(rows[g])
        for name, ci in pf.columns:
            ch = o[(g, name)]
            assert ch.error == 0, (g, name)
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
    # spot values quoted from the file: row 0 = {id 0, age 20, sadness 5, keen true, anniversary 2^64-1}
    assert o[(0, "age")].values.view(np.int32)[0] == 20
    assert o[(0, "anniversary")].values.view(np.uint64)[0] == 18446744073709551615
