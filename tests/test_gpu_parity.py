"""GPU parity tests: the HIP decoder (through the C ABI) against the CPU oracle, bit exact."""
import numpy as np
import pytest

import helpers as H
import oracle as O
from parquet_amd import _native as N
from parquet_amd import reader as R
from parquet_amd.writer import Column, ParquetWriter, encode_levels, write_flat_table, REQUIRED, OPTIONAL, REPEATED

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = R.Context(0)
    yield c
    c.close()


# ---------------------------------------------------------------- levels on their own (rle.Read)
def _bitpacked_block(width, groups_bytes, n_groups):
    body = bytes([(n_groups << 1) | 1]) + bytes(groups_bytes)
    return len(body).to_bytes(4, "little") + body


def test_reference_bitpack_vectors(ctx):
    # internal/bitpack/bitpack_test.go:21-43 as single-group bit-packed runs
    cases = [(1, [0, 1, 1, 0, 0, 1, 1, 1], [0b11100110]),
             (2, [0, 1, 2, 0, 0, 1, 2, 2], [0b00100100, 0b10100100]),
             (3, [0, 1, 2, 3, 4, 5, 6, 7], [0b10001000, 0b11000110, 0b11111010]),
             (4, [0, 2, 4, 7, 14, 15, 1, 0], [0x20, 0x74, 0xFE, 0x01])]
    for w, ints, data in cases:
        vals, n = R.rle_decode(ctx, w, _bitpacked_block(w, data, 1), 8)
        assert n == 8 and vals.tolist() == ints


def _mod(m, c):
    return [i % m for i in range(c)]


RLE_CASES = [(3, [4] * 100 + [5] * 100), (0, [0] * 10), (1, [0] * 17), (1, [1] * 17), (3, _mod(3, 100)),
             (3, _mod(3, 103)), (1, [1]), (1, [1, 0, 1, 1, 0]), (2, [1, 2, 3]), (3, [1, 2, 7])]


@pytest.mark.parametrize("width,vals", RLE_CASES)
def test_reference_rle_round_trips(ctx, width, vals):
    # internal/rle/rle_test.go:21-71
    enc = O.rle_encode(width, vals)
    want, _ = O.rle_read(width, enc)
    got, n = R.rle_decode(ctx, width, enc, len(want))
    assert n == len(want)
    assert np.array_equal(got, want)
    assert got[:len(vals)].tolist() == vals


def test_old_writer_runs_of_one(ctx):
    block = bytes([8, 0, 0, 0, 0x02, 0x01, 0x02, 0x00, 0x02, 0x01, 0x08, 0x00])
    got, n = R.rle_decode(ctx, 1, block, 16)
    assert n == 7 and got.tolist() == [1, 0, 1, 0, 0, 0, 0]


def test_short_bitpacked_run_is_zero_padded(ctx):
    got, n = R.rle_decode(ctx, 1, bytes([2, 0, 0, 0, (2 << 1) | 1, 0xFF]), 16)
    assert n == 16 and got.tolist() == [1] * 8 + [0] * 8


def test_malformed_blocks_return_errors(ctx):
    for block in (bytes([1, 0, 0, 0, 0x03]), bytes([1, 0, 0, 0, 0x10]), bytes([1, 0, 0, 0, 0x80]), b"\x01\x00"):
        with pytest.raises(O.OracleError):
            O.rle_read(1, block)
        with pytest.raises(N.PqdError) as e:
            R.rle_decode(ctx, 1, block, 64)
        assert e.value.code == N.E_CORRUPT


def test_rle_value_byte_not_masked_and_multibyte_header(ctx):
    got, n = R.rle_decode(ctx, 1, bytes([2, 0, 0, 0, 0x06, 0xFF]), 8)
    assert got.tolist() == [255, 255, 255]
    got, n = R.rle_decode(ctx, 2, bytes([3, 0, 0, 0, 0xD0, 0x0F, 0x03]), 2000)
    assert n == 1000 and got.tolist() == [3] * 1000


@pytest.mark.parametrize("width", [0, 1, 2, 3, 4])
def test_random_level_blocks_reference_widths(ctx, width):
    rng = np.random.default_rng(100 + width)
    for n in (1, 7, 8, 9, 63, 64, 503, 504, 505, 513, 4000, 4096, 4097, 20000, 300000):
        for p_run in (0.0, 0.5, 0.9, 0.999):
            vals = rng.integers(0, max(1 << width, 1), n, dtype=np.uint8)
            keep = rng.random(n) < p_run
            idx = np.where(~keep)[0]
            if len(idx):
                # forward-fill to make runs
                last = np.maximum.accumulate(np.where(~keep, np.arange(n), 0))
                vals = vals[last]
            enc = O.rle_encode(width, vals)
            want, _ = O.rle_read(width, enc)
            got, cnt = R.rle_decode(ctx, width, enc, len(want) + 5)
            assert cnt == len(want), (n, p_run)
            assert np.array_equal(got, want), (width, n, p_run)
            # truncated request (fields.go:300,308 take [:NumValues])
            got2, _ = R.rle_decode(ctx, width, enc, n)
            assert np.array_equal(got2, vals)


def _spec_encode(width, vals, rng):
    """Hybrid block for any width 1..32 (Parquet format text): random mix of RLE and bit-packed runs."""
    vals = np.asarray(vals, np.uint64)
    out = bytearray()
    i, n = 0, len(vals)
    bw = (width + 7) // 8
    while i < n:
        j = i
        while j < n and vals[j] == vals[i]:
            j += 1
        if j - i >= 8 or (rng.random() < 0.2):
            h = (j - i) << 1
            while h >= 0x80:
                out.append((h & 0x7F) | 0x80)
                h >>= 7
            out.append(h)
            out += int(vals[i]).to_bytes(bw, "little")
            i = j
        else:
            g = int(rng.integers(1, 70))
            cnt = min(g * 8, ((n - i) + 7) // 8 * 8)
            g = cnt // 8
            chunk = np.zeros(cnt, np.uint64)
            m = min(cnt, n - i)
            chunk[:m] = vals[i:i + m]
            h = (g << 1) | 1
            while h >= 0x80:
                out.append((h & 0x7F) | 0x80)
                h >>= 7
            out.append(h)
            acc = 0
            for k, v in enumerate(chunk):
                acc |= int(v) << (k * width)
            out += acc.to_bytes(cnt * width // 8, "little")
            i += m
    return len(out).to_bytes(4, "little") + bytes(out)


@pytest.mark.parametrize("width", list(range(1, 33)))
def test_level_widths_1_to_32_spec_parity(ctx, width):
    """Widths 5..32 have no reference behaviour (rle.New rejects > 4): parity here is against the
    Parquet format text as restated by the oracle's pqo_rle_read_u32."""
    rng = np.random.default_rng(200 + width)
    n = 5000
    vals = rng.integers(0, 1 << width, n, dtype=np.uint64)
    vals[100:400] = vals[100]
    vals[2000:2009] = vals[2000]
    block = _spec_encode(width, vals, rng)
    want, _ = O.rle_read_u32(width, block, n + 600)
    assert np.array_equal(want[:n], vals.astype(np.uint32))
    esz = 1 if width <= 8 else 2 if width <= 16 else 4
    got, cnt = R.rle_decode(ctx, width, block, len(want), elem_size=esz)
    assert cnt == len(want)
    assert np.array_equal(got.astype(np.uint32), want)
    if width <= 8:
        want8, _ = O.rle_read(width, block)
        assert np.array_equal(got, want8)


# ---------------------------------------------------------------- whole files
def test_flat_required_numeric(ctx):
    # config "flat required int32/int64/float64", small pages + ragged tail
    rng = np.random.default_rng(1)
    cols = [Column("i32", N.INT32, [0]), Column("i64", N.INT64, [0]), Column("f64", N.DOUBLE, [0]),
            Column("f32", N.FLOAT, [0]), Column("u32", N.INT32, [0], 13), Column("u64", N.INT64, [0], 14)]
    for n, group, page in ((1, 10, 10), (1000, 1000, 1000), (10007, 4000, 999), (70000, 70000, 33333)):
        buf = H.build_file(cols, n, group, page, rng)
        H.check_file(buf, ctx)
        H.check_file(buf, ctx, rg_batch=1)


@pytest.mark.parametrize("p_null", [0.0, 0.1, 0.5, 1.0])
def test_optional_numeric_and_strings(ctx, p_null):
    rng = np.random.default_rng(int(p_null * 100) + 7)
    cols = [Column("a", N.INT32, [1]), Column("b", N.INT64, [1]), Column("s", N.BYTE_ARRAY, [1]),
            Column("t", N.BYTE_ARRAY, [0]), Column("d", N.DOUBLE, [1])]
    buf = H.build_file(cols, 30011, 12000, 5000, rng, p_null=p_null, str_len=(8, 32), alphabet=(97, 123))
    H.check_file(buf, ctx)


def test_bools_across_page_boundaries(ctx):
    # parquet_test.go:219-376: 3001 values at page sizes 2 / 3 / 3000
    rng = np.random.default_rng(3)
    cols = [Column("b", N.BOOLEAN, [0]), Column("ob", N.BOOLEAN, [1])]
    for page in (2, 3, 7, 8, 9, 3000, 4096):
        buf = H.build_file(cols, 3001, 3001, page, rng, p_null=0.4)
        H.check_file(buf, ctx)
    buf = H.build_file(cols, 100003, 100003, 50001, rng, p_null=0.1)
    H.check_file(buf, ctx)


def test_nested_repeated_person(ctx):
    # README Person{Being{ID, Age*}, Username, Friends []Being}
    rng = np.random.default_rng(4)
    cols = [Column("being.id", N.INT32, [0, 0]), Column("being.age", N.INT32, [0, 1]),
            Column("username", N.BYTE_ARRAY, [0]),
            Column("friends.id", N.INT32, [2, 0]), Column("friends.age", N.INT32, [2, 1])]
    buf = H.build_file(cols, 20000, 8000, 3000, rng, p_null=0.5)
    o, g = H.check_file(buf, ctx)
    assert o[(0, "friends.age")].max_def == 2 and o[(0, "friends.age")].max_rep == 1


def test_dremel_document_and_deep_chain(ctx):
    # cmd/parquetgen/dremel/testcases/doc schema (def <= 3, rep <= 2) and a 5-deep chain (width 3)
    rng = np.random.default_rng(5)
    cols = [Column("docid", N.INT64, [0]),
            Column("links.backward", N.INT64, [1, 2]), Column("links.forward", N.INT64, [1, 2]),
            Column("names.languages.code", N.BYTE_ARRAY, [2, 2, 0]),
            Column("names.languages.country", N.BYTE_ARRAY, [2, 2, 1]),
            Column("names.url", N.BYTE_ARRAY, [2, 1]),
            Column("a.b.c.d.e", N.INT32, [1, 2, 1, 2, 1]),
            Column("x.y.z.w.v.u.t", N.INT64, [2, 1, 2, 1, 2, 1, 1])]
    buf = H.build_file(cols, 6000, 2500, 700, rng, p_null=0.25, str_len=(0, 12))
    o, g = H.check_file(buf, ctx)
    assert o[(0, "a.b.c.d.e")].max_def == 5 and o[(0, "x.y.z.w.v.u.t")].max_def == 7


def test_dremel_paper_known_levels(ctx):
    """cmd/parquetgen/dremel/dremel_test.go:77-84: the levels of the two Dremel-paper documents,
    written with the reference encoder layout and read back through the GPU path."""
    kat = [("links.backward", [1, 2], N.INT64, [1, 2, 2], [0, 0, 1]),
           ("links.forward", [1, 2], N.INT64, [2, 2, 2, 2], [0, 1, 1, 0]),
           ("names.languages.code", [2, 2, 0], N.BYTE_ARRAY, [2, 2, 1, 2, 1], [0, 2, 1, 1, 0]),
           ("names.languages.country", [2, 2, 1], N.BYTE_ARRAY, [3, 2, 1, 3, 1], [0, 2, 1, 1, 0]),
           ("names.url", [2, 1], N.BYTE_ARRAY, [2, 2, 1, 2], [0, 1, 1, 0])]
    cols = [Column(p, t, types) for p, types, t, _, _ in kat]
    w = ParquetWriter(cols)
    w.begin_row_group()
    for ci, (p, types, t, d, r) in enumerate(kat):
        md = cols[ci].max_def
        nv = sum(1 for x in d if x == md)
        if t == N.INT64:
            w.write_page(ci, np.arange(nv, dtype=np.int64) * 10, defs=d, reps=r)
        else:
            chars = np.frombuffer(b"ab" * nv, np.uint8)
            w.write_page(ci, chars, defs=d, reps=r, offsets=np.arange(nv + 1, dtype=np.int32) * 2)
    w.end_row_group(2)
    buf = w.finish()
    o, g = H.check_file(buf, ctx)
    for p, types, t, d, r in kat:
        assert g[(0, p)]["defs"].tolist() == d
        assert g[(0, p)]["reps"].tolist() == r
        assert len(g[(0, p)]["row_level_off"]) == 3  # two documents


def test_strings_edge_cases(ctx):
    rng = np.random.default_rng(6)
    cols = [Column("s", N.BYTE_ARRAY, [0]), Column("o", N.BYTE_ARRAY, [1])]
    # empty strings only, long strings (beyond the tabulated entry window), binary bytes, mixed
    for lo, hi, n, page in ((0, 0, 5000, 1000), (0, 3, 20000, 7000), (100, 400, 3000, 1000),
                            (1000, 6000, 200, 77), (0, 300, 9000, 2048)):
        buf = H.build_file(cols, n, n, page, rng, p_null=0.2, str_len=(lo, hi))
        H.check_file(buf, ctx)


def test_strings_fixed_strides_and_giants(ctx):
    """Fixed lengths walk the 4 length bytes over every offset of the 2 KiB chain tiles and 32-byte
    sub-tiles; the giant strings cover tiles that no string starts in."""
    rng = np.random.default_rng(16)
    cols = [Column("s", N.BYTE_ARRAY, [0])]
    for ln in (1, 13, 27, 28, 29, 61, 2044, 2045):
        buf = H.build_file(cols, 3000, 3000, 1500, rng, str_len=(ln, ln))
        H.check_file(buf, ctx)
    w = ParquetWriter(cols, stats=False)
    w.begin_row_group()
    parts = [b"ab", bytes(rng.integers(0, 256, 100001, dtype=np.uint8)), b"", b"xyz",
             bytes(rng.integers(0, 256, 4096, dtype=np.uint8)), b"q" * 2047, b"tail"]
    payload = b"".join(len(x).to_bytes(4, "little") + x for x in parts)
    w.write_raw_page(0, payload, len(parts))
    w.end_row_group(len(parts))
    buf = w.finish()
    o, g = H.check_file(buf, ctx)
    assert bytes(g[(0, "s")]["chars"]) == b"".join(parts)


def test_strings_of_mixed_lengths_text(ctx):
    """Text strings from 0 to a few tiles long in one column: short ones give a tile more plausible starts than
    its exit table has rows, long ones make the chain enter tiles far from their first bytes, and the mix
    puts both next to each other (entries tabulated, linked, walked and, past the table, hopped serially)."""
    rng = np.random.default_rng(18)
    cols = [Column("s", N.BYTE_ARRAY, [1])]
    for ranges in (((61, 100),), ((100, 400),), ((0, 3), (61, 100), (100, 400), (0, 3), (2000, 5000), (8, 32))):
        w = ParquetWriter(cols, stats=False)
        w.begin_row_group()
        n_rows = 0
        for page in range(3):
            lens = np.concatenate([rng.integers(lo, hi + 1, int(rng.integers(50, 4000) if hi < 1000 else rng.integers(5, 40))) for lo, hi in ranges])
            if len(ranges) > 1:
                rng.shuffle(lens[: len(lens) // 2])  # first half mixed at random, second half in blocks of one kind
            offs = np.zeros(len(lens) + 1, np.int32)
            np.cumsum(lens, out=offs[1:])
            chars = rng.integers(97, 123, int(offs[-1]), dtype=np.uint8)
            defs = (rng.random(int(len(lens) * 1.1) + 1) > 0.09).astype(np.uint8)
            while int(defs.sum()) < len(lens):
                defs = np.concatenate([defs, np.ones(len(lens) - int(defs.sum()), np.uint8)])
            nz = np.flatnonzero(defs)
            defs = defs[: nz[len(lens) - 1] + 1]  # exactly len(lens) non-null slots
            w.write_page(0, chars, defs=defs, offsets=offs)
            n_rows += len(defs)
        w.end_row_group(n_rows)
        buf = w.finish()
        w.close()
        H.check_file(buf, ctx)


def test_strings_longer_than_tiles_across_groups_and_super_groups(ctx):
    """Hundreds of strings of several KiB: nearly every 2 KiB tile is entered far from its start or not at all,
    and the chain enters most groups of 16 tiles and super-groups of 256 beyond their first tile."""
    rng = np.random.default_rng(19)
    cols = [Column("s", N.BYTE_ARRAY, [0])]
    for lo, hi, n in ((3000, 9000, 700), (2049, 2049, 1500), (40000, 70000, 40)):
        w = ParquetWriter(cols, stats=False)
        w.begin_row_group()
        lens = rng.integers(lo, hi + 1, n)
        offs = np.zeros(n + 1, np.int32)
        np.cumsum(lens, out=offs[1:])
        chars = rng.integers(97, 123, int(offs[-1]), dtype=np.uint8)
        w.write_page(0, chars, offsets=offs)
        w.end_row_group(n)
        buf = w.finish()
        w.close()
        o, g = H.check_file(buf, ctx)
        assert bytes(g[(0, "s")]["chars"]) == chars.tobytes()


def test_strings_that_look_like_length_prefixes(ctx):
    """Chars drawn from {0, 1, 2}: almost every byte of the page is a plausible length prefix, so the false
    candidate chains of the chain kernels are as long as they can get."""
    rng = np.random.default_rng(17)
    cols = [Column("s", N.BYTE_ARRAY, [0]), Column("o", N.BYTE_ARRAY, [1])]
    for lo, hi in ((0, 40), (0, 5), (30, 300)):
        buf = H.build_file(cols, 20000, 20000, 6000, rng, p_null=0.3, str_len=(lo, hi), alphabet=(0, 3))
        H.check_file(buf, ctx)


def test_strings_around_quarter_and_tile_boundaries(ctx):
    """Pass 1 walks a tile in quarters of 512 bytes with rows of their own and hops from quarter to quarter: strings
    whose 4 + len is a divisor or a near-divisor of 512 / 2048 put every string start on, just before or just behind
    a quarter or tile boundary; all-zero chars make every position a plausible start of an empty string (chains of
    4-byte steps that no quarter row catches); lengths of a few hundred bytes land in quarters whose leading
    positions hold no row at all."""
    rng = np.random.default_rng(23)
    cols = [Column("s", N.BYTE_ARRAY, [0])]
    for ln in (0, 4, 12, 60, 124, 252, 507, 508, 509, 1020, 1021, 2043):
        for alphabet in ((97, 123), (0, 1), (0, 3)):
            n = max(40, 30000 // (ln + 4))
            buf = H.build_file(cols, n, n, max(1, n // 2), rng, str_len=(ln, ln), alphabet=alphabet)
            H.check_file(buf, ctx)
    cols = [Column("o", N.BYTE_ARRAY, [1])]
    for lo, hi in ((500, 520), (250, 260), (120, 135), (2040, 2056)):
        for alphabet in ((97, 123), (0, 3), (0, 256)):
            buf = H.build_file(cols, 400, 400, 130, rng, p_null=0.15, str_len=(lo, hi), alphabet=alphabet)
            H.check_file(buf, ctx)


def test_strings_with_a_zero_behind_every_letter_in_a_large_page(ctx):
    """UTF-16-like chars in one page of 16 MB: four bytes `a 0 b 0` read as a length of a few megabytes, which fits
    into the rest of such a page; without an upper bound on the lengths a candidate may have, half of all positions
    are candidates and no tile's rows hold its true entry."""
    rng = np.random.default_rng(29)
    cols = [Column("o", N.BYTE_ARRAY, [1])]
    n = 520000
    buf = H.build_file(cols, n, n, n, rng, p_null=0.1, str_len=(8, 52), alphabet="utf16")
    o, g = H.check_file(buf, ctx)
    assert len(g[(0, "o")]["chars"]) > 12 << 20


def test_strings_longer_than_the_candidate_bound(ctx):
    """Strings of 1 MiB and more are true starts that pass 1 does not take for candidates: the hop kernels find no
    table row for them and step over them one by one.  Short strings between them, and one as the page's last value."""
    rng = np.random.default_rng(31)
    cols = [Column("s", N.BYTE_ARRAY, [0])]
    lens = []
    for big in ((1 << 20) - 1, 1 << 20, (1 << 20) + 1, 1500000, 2200000):
        lens += [int(x) for x in rng.integers(0, 40, int(rng.integers(1, 200)))] + [big]
    lens += [5, 0, 1 << 20]
    offs = np.zeros(len(lens) + 1, np.int32)
    np.cumsum(lens, out=offs[1:])
    chars = rng.integers(0, 256, int(offs[-1]), dtype=np.uint8)
    w = ParquetWriter(cols, stats=False)
    w.begin_row_group()
    w.write_page(0, chars, offsets=offs)
    w.end_row_group(len(lens))
    buf = w.finish()
    w.close()
    o, g = H.check_file(buf, ctx)
    assert bytes(g[(0, "s")]["chars"]) == chars.tobytes()


def test_tiny_and_empty_pages(ctx):
    rng = np.random.default_rng(8)
    cols = [Column("a", N.INT32, [1]), Column("s", N.BYTE_ARRAY, [1]), Column("b", N.BOOLEAN, [1]),
            Column("r", N.INT64, [2])]
    for page in (1, 2, 3, 5):
        buf = H.build_file(cols, 200, 90, page, rng, p_null=0.5)
        H.check_file(buf, ctx)


def test_statistics_free_headers_and_all_null_pages(ctx):
    rng = np.random.default_rng(9)
    cols = [Column("a", N.INT32, [1]), Column("s", N.BYTE_ARRAY, [1])]
    buf = H.build_file(cols, 5000, 5000, 1000, rng, p_null=1.0, stats=False)
    o, g = H.check_file(buf, ctx)
    assert g[(0, "a")]["values"].size == 0 and g[(0, "s")]["offsets"].tolist() == [0]


def test_hand_built_old_format_level_block(ctx):
    """A page whose def levels are RLE runs of length 1, as in _examples/via_parquet/people.parquet."""
    cols = [Column("age", N.INT32, [1])]
    w = ParquetWriter(cols, stats=False)
    w.begin_row_group()
    defs = [1, 0, 1, 1, 0, 0, 1, 0, 1, 1]
    body = b"".join(bytes([0x02, d]) for d in defs)
    vals = np.arange(sum(defs), dtype=np.int32) + 20
    payload = len(body).to_bytes(4, "little") + body + vals.tobytes()
    w.write_raw_page(0, payload, len(defs))
    w.end_row_group(len(defs))
    buf = w.finish()
    o, g = H.check_file(buf, ctx)
    assert g[(0, "age")]["defs"].tolist() == defs
    assert g[(0, "age")]["values"].tolist() == vals.tolist()


def test_corrupt_pages_return_errors_not_crashes(ctx):
    cols = [Column("age", N.INT32, [1]), Column("s", N.BYTE_ARRAY, [0])]

    def one(payload0, nv0, payload1, nv1):
        w = ParquetWriter(cols, stats=False)
        w.begin_row_group()
        w.write_raw_page(0, payload0, nv0)
        w.write_raw_page(1, payload1, nv1)
        w.end_row_group(nv0)
        return w.finish()

    good_s = (3).to_bytes(4, "little") + b"abc"
    good_a = bytes([2, 0, 0, 0, 0x02, 0x01]) + (7).to_bytes(4, "little")
    buf = one(good_a, 1, good_s, 1)
    H.check_file(buf, ctx)
    cases = {
        "levels shorter than num_values": one(bytes([2, 0, 0, 0, 0x02, 0x01]) + (7).to_bytes(4, "little"), 5, good_s, 1),
        "level block longer than the page": one(bytes([200, 0, 0, 0, 0x02, 0x01]), 1, good_s, 1),
        "truncated run header": one(bytes([1, 0, 0, 0, 0x80]) + b"\0" * 4, 1, good_s, 1),
        "values shorter than the non-null count": one(bytes([2, 0, 0, 0, 0x06, 0x01]) + (7).to_bytes(4, "little"), 3, good_s, 1),
        "negative string length": one(good_a, 1, (0xFFFFFFFF).to_bytes(4, "little") + b"abc", 1),
        "string runs past the page": one(good_a, 1, (300).to_bytes(4, "little") + b"abc", 1),
        "fewer strings than values": one(good_a, 1, good_s, 2),
    }
    # The reference does not fail on a final string whose declared length exceeds the bytes left: the
    # short Read leaves the tail of the string zero (template_string.go:52-55).  The decoder reports it
    # as corrupt instead of returning padding.
    lenient = {"string runs past the page"}
    for name, buf in cases.items():
        pf = R.ParquetFile(buf)
        o = H.oracle_file(buf, pf)
        if name in lenient:
            assert all(ch.error == 0 for ch in o.values()), name
            assert o[(0, "s")].values.tobytes() == b"abc" + b"\0" * 297
        else:
            assert any(ch.error != 0 for ch in o.values()), name
        with pytest.raises(N.PqdError) as e:
            H.gpu_file(buf, ctx, pf=pf)
        assert e.value.code in (N.E_CORRUPT, N.E_NONCANONICAL), (name, e.value.code)


def test_reader_api_next_scan(ctx):
    """NewParquetReader / Next / Scan / Rows / Levels through the Python mirror."""
    rng = np.random.default_rng(10)
    n = 500
    ids = rng.integers(0, 1000, n).astype(np.int32)
    age = rng.integers(0, 100, n).astype(np.int32)
    am = rng.random(n) > 0.5
    names = [b"user%d" % i for i in range(n)]
    cols = [Column("id", N.INT32, [0]), Column("age", N.INT32, [1]), Column("name", N.BYTE_ARRAY, [0])]
    buf = write_flat_table(cols, [ids, (age, am), names], 200, 60)
    pr = R.ParquetReader(buf, ctx=ctx)
    assert pr.rows() == n
    lv = pr.levels()
    assert [l.name for l in lv] == ["id", "age", "name"] and lv[0].defs is None
    assert lv[1].defs.tolist() == am[:200].astype(np.uint8).tolist()
    i = 0
    while pr.next():
        row = pr.scan()
        assert row["id"][0].tolist() == [ids[i]]
        assert row["age"][0].tolist() == ([age[i]] if am[i] else [])
        assert row["name"][0] == [names[i]]
        i += 1
    assert i == n and pr.error() is None


def test_full_size_pages_property(ctx):
    """BASELINE size: 1M-row pages.  Size-independent property: a required fixed-width chunk equals the
    concatenation of its pages' payload bytes; the def levels of an optional chunk expand back to the
    validity mask they were encoded from; the dense values equal the non-null inputs."""
    n, page = 3_000_000, 1_000_000
    i32 = H.splitmix64(2, n).astype(np.uint32).view(np.int32)
    i64 = H.splitmix64(3, n).view(np.int64)
    f64 = (H.splitmix64(4, n) & np.uint64(0x7FEFFFFFFFFFFFFF)).view(np.float64)
    mask = (H.splitmix64(5, n) % np.uint64(10)) != 0
    opt = H.splitmix64(6, n).astype(np.uint32).view(np.int32)
    cols = [Column("i32", N.INT32, [0]), Column("i64", N.INT64, [0]), Column("f64", N.DOUBLE, [0]),
            Column("opt", N.INT32, [1])]
    buf = write_flat_table(cols, [i32, i64, f64, (opt, mask)], n, page)
    g = H.gpu_file(buf, ctx)
    assert np.array_equal(g[(0, "i32")]["values"], i32)
    assert np.array_equal(g[(0, "i64")]["values"], i64)
    assert np.array_equal(g[(0, "f64")]["values"].view(np.uint64), f64.view(np.uint64))
    assert np.array_equal(g[(0, "opt")]["defs"], mask.astype(np.uint8))
    assert np.array_equal(g[(0, "opt")]["values"], opt[mask])
    assert np.array_equal(g[(0, "opt")]["validity"], np.packbits(mask, bitorder="little"))
    assert g[(0, "opt")]["sizes"].tolist() == [int(mask[i:i + page].sum()) for i in range(0, n, page)]
    # and against the oracle on the optional chunk (seconds)
    o = H.oracle_file(buf)
    H.assert_chunk_equal(o[(0, "opt")], g[(0, "opt")], "opt")


def test_reference_fixture_people(ctx):
    """Pages of the reference's own people.parquet fixture (decompressed): GPU == oracle == pyarrow."""
    from test_oracle_golden import _load_people, expected_people_chunk
    buf, exp = _load_people()
    o, g = H.check_file(buf, ctx)
    pf = R.ParquetFile(buf)
    rows = exp["row_group_rows"]
    r0 = 0
    for rg in range(2):
        r1 = r0 + int(rows[rg])
        for name, ci in pf.columns:
            e = expected_people_chunk(exp, name, r0, r1)
            got = g[(rg, name)]
            if ci.max_def:
                assert np.array_equal(got["defs"], e["valid"].astype(np.uint8))
                assert np.array_equal(got["validity"], np.packbits(e["valid"], bitorder="little"))
            if "offsets" in e:
                assert np.array_equal(got["offsets"], e["offsets"]) and np.array_equal(got["chars"], e["chars"])
            elif ci.phys_type == N.BOOLEAN:
                assert np.array_equal(got["values"], e["values"].astype(np.uint8))
            else:
                assert got["values"].tobytes() == np.ascontiguousarray(e["values"]).tobytes()
        r0 = r1


def _message_file(n_rows, rng):
    """configs[0]: the 40-column schema of performance/message/message.go:7-53, written like
    performance/parquet_performance_test.go:47-55 (Write() when i % 5000 == 0, so the first row group
    holds one row; default 1000-row pages) but uncompressed.  Pointer columns `*_0` are nil with p = 0.5,
    the other pointer columns are never nil (:29-36)."""
    spec = []
    strs = {0: (3, 10), 1: (0, 5), 2: (100, 400), 3: (100, 400), 4: (20, 80), 5: (20, 80), 6: (20, 80),
            7: (3, 10), 8: (3, 10), 9: (3, 10)}
    for i in range(10):
        spec.append(("col_str_%d" % i, N.BYTE_ARRAY, i % 2 == 0, strs[i]))
    for i in range(5):
        spec.append(("col_int_%d" % i, N.INT64, i % 2 == 0, None))
    for i in range(5):
        spec.append(("col_int_32_%d" % i, N.INT32, i % 2 == 0, None))
    for i in range(5):
        spec.append(("col_float_%d" % i, N.DOUBLE, i % 2 == 0, None))
    for i in range(5):
        spec.append(("col_float_32_%d" % i, N.FLOAT, i % 2 == 0, None))
    for i in range(10):
        spec.append(("col_bool_%d" % i, N.BOOLEAN, i % 2 == 0, None))
    cols = [Column(name, t, [OPTIONAL if opt else REQUIRED]) for name, t, opt, _ in spec]
    groups = [1] + [5000] * ((n_rows - 1) // 5000) + ([(n_rows - 1) % 5000] if (n_rows - 1) % 5000 else [])
    w = ParquetWriter(cols)
    for rows in groups:
        w.begin_row_group()
        for ci, (name, t, opt, sl) in enumerate(spec):
            for p0 in range(0, rows, 1000):
                nr = min(1000, rows - p0)
                defs = None
                nv = nr
                if opt:
                    p_nil = 0.5 if name.endswith("_0") else 0.0
                    defs = (rng.random(nr) >= p_nil).astype(np.uint8)
                    nv = int(defs.sum())
                if t == N.BYTE_ARRAY:
                    chars, offs = H.random_strings(nv, rng, sl[0], sl[1], (97, 123))
                    w.write_page(ci, chars, defs=defs, offsets=offs)
                else:
                    w.write_page(ci, H.random_values(t, nv, rng), defs=defs)
        w.end_row_group(rows)
    buf = w.finish()
    w.close()
    return buf, len(groups)


def test_performance_message_schema(ctx):
    rng = np.random.default_rng(11)
    buf, n_groups = _message_file(20001, rng)
    pf = R.ParquetFile(buf)
    assert pf.num_columns == 40 and pf.num_row_groups == n_groups and pf.row_group_rows(0) == 1
    H.check_file(buf, ctx)            # all row groups in one batch
    H.check_file(buf, ctx, rg_batch=1)  # one readRowGroup per call, as the reference reader does


# ---------------------------------------------------------------- fuzz: hand-built level blocks through the page path
def _leb(h, rng, pad_ok=True):
    out = []
    while True:
        b = h & 0x7F
        h >>= 7
        if h:
            out.append(b | 0x80)
        else:
            out.append(b)
            break
    if pad_ok and rng.random() < 0.1:  # non-canonical: continuation bytes that add nothing (readLEB128 accepts them)
        extra = int(rng.integers(1, 9 - len(out))) if len(out) < 8 else 0
        if extra:
            out[-1] |= 0x80
            out += [0x80] * (extra - 1) + [0x00]
    return bytes(out)


def _random_level_block(rng, width, min_values):
    """A hybrid RLE/bit-packed block no encoder would write: empty runs, runs of 1, long runs, zero-group
    bit-packed headers, random packed bytes, padded headers."""
    body = bytearray()
    total = 0
    while total < min_values:
        if rng.random() < 0.5:
            count = int(rng.choice([0, 1, 2, 7, 8, 9, 100, int(rng.integers(0, 6000))]))
            body += _leb(count << 1, rng) + bytes([int(rng.integers(0, 1 << width))])
            total += count
        else:
            groups = int(rng.choice([0, 1, 2, 3, 63, 64, int(rng.integers(0, 300))]))
            body += _leb((groups << 1) | 1, rng) + bytes(rng.integers(0, 256, groups * width, dtype=np.uint8))
            total += 8 * groups
    return len(body).to_bytes(4, "little") + bytes(body), total


@pytest.mark.parametrize("seed", list(range(24)))
def test_fuzz_whole_files(ctx, seed):
    """Random schemas (every physical type, required / optional / repeated paths), page and row-group sizes, null
    rates, string lengths from empty to several tiles, text and binary alphabets, both codecs."""
    rng = np.random.default_rng(4000 + seed)
    phys = [N.INT32, N.INT64, N.FLOAT, N.DOUBLE, N.BOOLEAN, N.BYTE_ARRAY, N.BYTE_ARRAY]
    shapes = [[0], [1], [1, 1], [2], [2, 1], [1, 2, 0], [2, 2, 1]]
    cols = []
    for ci in range(int(rng.integers(2, 7))):
        t = shapes[int(rng.integers(0, len(shapes)))]
        cols.append(Column(".".join("f%dl%d" % (ci, i) for i in range(len(t))), phys[int(rng.integers(0, len(phys)))], t))
    n_rows = int(rng.integers(1, 6000))
    group = int(rng.integers(1, n_rows + 1))
    page = int(rng.integers(1, group + 1))
    lo = int(rng.choice([0, 0, 1, 8, 50, 300]))
    hi = lo + int(rng.choice([0, 3, 30, 200, 3000]))
    alphabet = [(97, 123), (0, 256), (0, 3)][int(rng.integers(0, 3))]
    buf = H.build_file(cols, n_rows, group, page, rng, p_null=float(rng.choice([0.0, 0.05, 0.5, 1.0])), str_len=(lo, hi),
                       alphabet=alphabet, stats=bool(rng.integers(0, 2)), codec=int(rng.integers(0, 2)))
    H.check_file(buf, ctx)


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6, 7, 8])
def test_fuzz_level_blocks_through_pages(ctx, seed):
    rng = np.random.default_rng(100 + seed)
    shapes = [[1], [1, 1, 1], [1] * 7, [1] * 15, [2], [2, 1, 2], [1, 2, 1, 2, 1]]
    cols = [Column(".".join("c%dn%d" % (ci, i) for i in range(len(t))), N.INT32, t) for ci, t in enumerate(shapes)]
    w = ParquetWriter(cols, stats=False)
    w.begin_row_group()
    for ci, t in enumerate(shapes):
        max_def = sum(1 for x in t if x in (OPTIONAL, REPEATED))
        max_rep = sum(1 for x in t if x == REPEATED)
        n_total = 0
        for _ in range(int(rng.integers(1, 5))):
            want = int(rng.choice([1, 5, 4095, 4096, 4097, 20000, int(rng.integers(1, 60000))]))
            # def block: random runs, then a run of 8 zeros that DataPageHeader.num_values may cut anywhere
            # (cut-off levels equal to max_def would make the reference read more values than the page has)
            dwidth = int(max_def).bit_length()
            dblock, dtotal = _random_level_block(rng, dwidth, want)
            dblock = (len(dblock) - 4 + 2).to_bytes(4, "little") + dblock[4:] + bytes([8 << 1, 0])
            n = dtotal + int(rng.integers(0, 9))
            payload = b""
            if max_rep:
                rblock, _ = _random_level_block(rng, int(max_rep).bit_length(), dtotal + 8)
                payload += rblock
            payload += dblock
            levels, _ = O.rle_read(dwidth, dblock)
            n = max(n, 1)
            nn = int(np.count_nonzero(np.asarray(levels[:n]) == max_def))
            payload += rng.integers(-2**31, 2**31 - 1, nn, dtype=np.int64).astype(np.int32).tobytes()
            w.write_raw_page(ci, payload, n)
            n_total += n
    w.end_row_group(1)
    buf = w.finish()
    w.close()
    H.check_file(buf, ctx)


_GRAPH_REPLAY_CHILD = r"""
import os, sys
root = sys.argv[1]
sys.path[:0] = [root, os.path.join(root, "tests")]
os.environ["PQD_GRAPH"] = "1"
import numpy as np
import helpers as H
from parquet_amd import _native as N
from parquet_amd import reader as R
from parquet_amd.writer import Column

c = R.Context(0)
rng = np.random.default_rng(77)
cols = [Column("a", N.INT32, [1]), Column("s", N.BYTE_ARRAY, [1]), Column("r.x", N.INT64, [2, 1])]
buf = H.build_file(cols, 20000, 20000, 6000, rng, p_null=0.2, str_len=(0, 20))
H.check_file(buf, c)
pf = R.ParquetFile(buf)
chunks, nc, pages, npg = pf.descriptors(0, pf.num_row_groups)
plan = R.Plan(c, chunks, nc, pages, npg, pf.address, pf.size)
d_src = N.decode.pqd_device_alloc(c.handle, pf.size)
assert N.decode.pqd_copy_to_device(c.handle, d_src, pf.address, pf.size) == 0
for _ in range(3):  # the second and third run replay the captured graph
    plan.run(d_src)
plan.sync()
o = H.oracle_file(buf, pf)
for k in range(nc):
    H.assert_chunk_equal(o[(0, pf.columns[chunks[k].column][0])], plan.column(k), key=k)
N.decode.pqd_device_free(c.handle, d_src)
plan.close()
c.close()
print("graph replay ok")
"""


def test_graph_replay_mode():
    """PQD_GRAPH=1: chain batches are captured once into a HIP graph and replayed; same results.  The mode is opt-in (it is
    slower than stream launches on ROCm 7.2).  The graph is linear: captured with the three streams' fork and join in it,
    its replays took seconds now and then (the runtime runs parallel branches on streams of its own, and more streams than
    hardware queues share queues).  The test runs in a process of its own with a time limit, so that a replay that does
    not come back fails this test and not the ones behind it."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        r = subprocess.run([sys.executable, "-c", _GRAPH_REPLAY_CHILD, root], capture_output=True, text=True, timeout=150)
    except subprocess.TimeoutExpired:  # subprocess.run has killed the child it started
        pytest.fail("HIP graph replay did not return within 150 s")
    assert r.returncode == 0 and "graph replay ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_short_def_block_on_multi_page_string_and_bool_columns(ctx):
    """A definition-level block that decodes cleanly but holds far fewer values than the page's num_values, on
    pages of more than one expand tile, after an earlier decode has left the device arena full of stale counts:
    the error comes back as a status and nothing is written through stale counts (the process survives and the
    next decode is correct)."""
    rng = np.random.default_rng(5)
    cols = [Column("s", N.BYTE_ARRAY, [1]), Column("b", N.BOOLEAN, [1])]
    good = H.build_file(cols, 60000, 60000, 20000, rng, p_null=0.1, str_len=(8, 32))
    H.check_file(good, ctx)  # dirties the arena

    def short_page(n_claimed, n_real, string):
        defs = np.ones(n_real, np.uint8)
        block = encode_levels(1, defs)
        if string:
            vals = b"".join((3).to_bytes(4, "little") + b"abc" for _ in range(n_real))
        else:
            vals = bytes([0xFF] * ((n_real + 7) // 8))
        return block + vals, n_claimed

    w = ParquetWriter(cols, stats=False)
    w.begin_row_group()
    for string, ci in ((True, 0), (False, 1)):
        for _ in range(3):
            payload, n = short_page(3 * 4096 + 17, 100, string)
            w.write_raw_page(ci, payload, n)
    w.end_row_group(3 * (3 * 4096 + 17))
    bad = w.finish()
    w.close()
    with pytest.raises(N.PqdError) as e:
        H.gpu_file(bad, ctx)
    assert e.value.code == N.E_CORRUPT
    H.check_file(good, ctx)


def test_surplus_levels_equal_to_max_def_are_refused(ctx):
    """Levels past num_values are dropped from Defs (fields.go:300,308) but counted by valsFromDefs (fields.go:311).
    A zero surplus (what every encoder pads with) decodes; a surplus that holds max_def would make the reference
    read values the page does not have: refused as non-canonical."""
    cols = [Column("a", N.INT32, [1])]

    def one(block_levels, n):
        w = ParquetWriter(cols, stats=False)
        w.begin_row_group()
        nn = int(np.count_nonzero(np.asarray(block_levels[:n]) == 1))
        w.write_raw_page(0, encode_levels(1, np.asarray(block_levels, np.uint8)) + np.arange(nn, dtype=np.int32).tobytes(), n)
        w.end_row_group(n)
        b = w.finish()
        w.close()
        return b

    H.check_file(one([1, 0, 1, 1, 0, 0, 0, 0], 5), ctx)                 # bit-packed group, zero surplus
    H.check_file(one([1] * 20 + [0] * 20, 25), ctx)                     # RLE surplus of zeros
    for lv, n in (([1, 0, 1, 1, 0, 0, 1, 0], 5), ([0] * 8 + [1] * 30, 10), ([1] * 9000, 4100)):
        with pytest.raises(N.PqdError) as e:
            H.gpu_file(one(lv, n), ctx)
        assert e.value.code == N.E_NONCANONICAL, (lv[:10], n, e.value.code)


def test_decode_rejects_uncompressed_page_outside_the_host_buffer_in_a_mixed_codec_batch(ctx):
    """pqd_decode moves SNAPPY pages behind the file bytes, which makes the device range longer than the host
    buffer; the descriptors of uncompressed chunks must still be checked against the host length."""
    rng = np.random.default_rng(9)
    cols = [Column("a", N.INT32, [1]), Column("b", N.INT32, [1])]
    buf = H.build_file(cols, 3000, 3000, 1000, rng, p_null=0.2, codec=1)
    pf = R.ParquetFile(buf)
    chunks, nc, pages, npg = pf.descriptors(0, 1)
    cv = (N.ChunkDesc * nc)(*[chunks[i] for i in range(nc)])
    pv = (N.PageDesc * npg)(*[pages[i] for i in range(npg)])
    assert cv[0].codec == 1 and cv[1].codec == 1
    cv[1].codec = 0                                     # pretend chunk 1 is uncompressed ...
    for k in range(cv[1].first_page, cv[1].first_page + cv[1].n_pages):
        pv[k].src_off = pf.size + 16                    # ... with pages just past the end of the host buffer
        pv[k].payload_len = 32
        pv[k].rep_bytes, pv[k].def_bytes = 0, -1        # level block length to be read from the host copy
    with pytest.raises(N.PqdError) as e:
        R.decode_host(ctx, cv, nc, pv, npg, pf.address, pf.size)
    assert e.value.code == N.E_CORRUPT


def test_value_index_scatters_dense_values_to_their_rows(ctx):
    """PQD_OUT_VALUE_INDEX + the validity bitmap give every non-null level its dense value in O(1): the per-row
    consumption of the generated write functions (performance/parquet.go:104-113).  Pages of sizes that are no
    multiple of 64, so 64-level words straddle pages."""
    rng = np.random.default_rng(21)
    n = 10007
    vals = rng.integers(-2**31, 2**31 - 1, n, dtype=np.int64).astype(np.int32)
    mask = rng.random(n) >= 0.3
    buf = write_flat_table([Column("a", N.INT32, [1])], [(vals, mask)], n, 999)
    pf = R.ParquetFile(buf)
    chunks, nc, pages, npg = pf.descriptors(0, 1)
    plan = R.decode_host(ctx, chunks, nc, pages, npg, pf.address, pf.size)
    try:
        vi = plan.fetch(0, N.OUT_VALUE_INDEX, np.uint32)
        bits = np.unpackbits(plan.fetch(0, N.OUT_VALIDITY), bitorder="little")[:n].astype(bool)
        dense = plan.fetch(0, N.OUT_VALUES, np.int32)
        assert vi.size == (n + 63) // 64 + 1 and vi[-1] == mask.sum() == dense.size
        assert np.array_equal(vi[:-1], np.concatenate(([0], np.cumsum(mask)))[0:n:64])
        rows = np.nonzero(bits)[0]
        below = np.array([bin(int.from_bytes(np.packbits(bits[(i // 64) * 64:i], bitorder="little").tobytes(), "little")).count("1") for i in rows[:500]])
        assert np.array_equal(dense[vi[rows[:500] // 64] + below], vals[rows[:500]])
    finally:
        plan.close()


@pytest.mark.parametrize("with_strings", [True, False])
@pytest.mark.parametrize("env", [{}, {"PQD_OVERLAP": "0"}, {"PQD_STR_AFTER_LEVELS": "1"}, {"PQD_GRAPH": "1"}, {"PQD_OVERLAP": "0", "PQD_GRAPH": "1"}])
def test_stream_schedules_give_the_same_results(monkeypatch, env, with_strings):
    """The decoder's launches on its three streams, on one stream (PQD_OVERLAP=0), with the strings' marking pass behind the
    level decode and replayed from a captured graph: every schedule of the same kernels decodes the same arrays,
    repeated runs included."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    c = R.Context(0)
    try:
        rng = np.random.default_rng(78)
        cols = [Column("a", N.INT32, [1]), Column("s", N.BYTE_ARRAY, [1]), Column("t", N.BYTE_ARRAY, [0]), Column("b", N.BOOLEAN, [1]),
                Column("d", N.DOUBLE, [0]), Column("r.x", N.INT64, [2, 1]), Column("r.s", N.BYTE_ARRAY, [2, 1])]
        if not with_strings:  # a batch without strings runs the value index on their stream
            cols = [c for c in cols if c.phys_type != N.BYTE_ARRAY]
        buf = H.build_file(cols, 30000, 15000, 4000, rng, p_null=0.2, str_len=(0, 40), alphabet=(97, 123))
        H.check_file(buf, c)
        pf = R.ParquetFile(buf)
        chunks, nc, pages, npg = pf.descriptors(0, pf.num_row_groups)
        plan = R.Plan(c, chunks, nc, pages, npg, pf.address, pf.size)
        d_src = N.decode.pqd_device_alloc(c.handle, pf.size)
        try:
            assert N.decode.pqd_copy_to_device(c.handle, d_src, pf.address, pf.size) == 0
            for _ in range(3):
                plan.run(d_src)
            plan.sync()
            o = H.oracle_file(buf, pf)
            per_rg = nc // pf.num_row_groups
            for k in range(nc):
                H.assert_chunk_equal(o[(k // per_rg, pf.columns[chunks[k].column][0])], plan.column(k), key=k)
        finally:
            N.decode.pqd_device_free(c.handle, d_src)
            plan.close()
    finally:
        c.close()


@pytest.mark.parametrize("share", ["1", "0"])
def test_sibling_leaves_share_their_repetition_levels(monkeypatch, share):
    """Friends.ID and Friends.Age (README.md:146-159) have one level per friend each: byte-identical repetition level
    blocks.  The decoder decodes them once and both chunks hand out the same REPS and ROW_LEVEL_OFF; with PQD_SHARE_REPS=0
    each chunk gets its own.  Either way every array equals the oracle's.  A third repeated column with other levels, and a
    sibling whose pages are cut differently, share nothing."""
    monkeypatch.setenv("PQD_SHARE_REPS", share)
    rng = np.random.default_rng(91)
    cols = [Column("friends.id", N.INT32, [2, 0]), Column("friends.age", N.INT32, [2, 1]), Column("friends.nick", N.BYTE_ARRAY, [2, 1]),
            Column("tags", N.BYTE_ARRAY, [2])]
    w = ParquetWriter(cols, stats=False)
    n_rows = 30000
    for g in range(2):
        w.begin_row_group()
        k = rng.integers(0, 4, n_rows)
        nl = np.maximum(k, 1)
        starts = np.zeros(n_rows + 1, np.int64)
        np.cumsum(nl, out=starts[1:])
        reps = np.ones(int(starts[-1]), np.uint8)
        reps[starts[:-1]] = 0
        present = np.repeat(k > 0, nl)
        cut = int(starts[n_rows // 2])  # a page boundary on a row boundary
        for ci in range(4):
            if ci == 3:  # other levels altogether
                d2, r2, nv = H.gen_levels([2], 3000, rng)
                lens = rng.integers(0, 9, nv)
                offs = np.zeros(nv + 1, np.int32)
                np.cumsum(lens, out=offs[1:])
                w.write_page(ci, rng.integers(97, 123, int(offs[-1]), dtype=np.uint8), defs=d2, reps=r2, offsets=offs)
                continue
            if ci == 0:
                defs = present.astype(np.uint8)
            else:
                defs = present.astype(np.uint8) + (present & (rng.random(reps.size) >= 0.4)).astype(np.uint8)
            md = cols[ci].max_def
            bounds = [0, cut, reps.size] if ci != 2 else [0, int(starts[n_rows // 3]), reps.size]  # friends.nick: other pages
            for a, b in zip(bounds[:-1], bounds[1:]):
                nv = int((defs[a:b] == md).sum())
                if ci == 2:
                    lens = rng.integers(1, 12, nv)
                    offs = np.zeros(nv + 1, np.int32)
                    np.cumsum(lens, out=offs[1:])
                    w.write_page(ci, rng.integers(97, 123, int(offs[-1]), dtype=np.uint8), defs=defs[a:b], reps=reps[a:b], offsets=offs)
                else:
                    w.write_page(ci, rng.integers(0, 100, nv, dtype=np.int64).astype(np.int32), defs=defs[a:b], reps=reps[a:b])
        w.end_row_group(n_rows)
    buf = w.finish()
    w.close()
    c = R.Context(0)
    try:
        H.check_file(buf, c)
        pf = R.ParquetFile(buf)
        chunks, nc, pages, npg = pf.descriptors(0, pf.num_row_groups)
        plan = R.decode_host(c, chunks, nc, pages, npg, pf.address, pf.size)
        try:
            assert N.decode.pqd_plan_shared_reps(plan._h) == (2 if share == "1" else 0)  # friends.age of both row groups
            same = plan.device_ptr(0, N.OUT_REPS) == plan.device_ptr(1, N.OUT_REPS)
            assert same == (share == "1")
            assert plan.device_ptr(0, N.OUT_REPS) != plan.device_ptr(3, N.OUT_REPS)
            assert plan.device_ptr(0, N.OUT_ROW_VALUE_OFF) != plan.device_ptr(1, N.OUT_ROW_VALUE_OFF)
        finally:
            plan.close()
    finally:
        c.close()
