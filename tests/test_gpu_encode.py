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


def test_runs_around_layout_chunk_boundaries(ctx):
    """Bit-packed runs (headers every 63 groups, counted from the last RLE run) that start in one chunk of 2048
    elements and go on through the next ones; the RLE run that restarts the count sits on either side of a
    chunk boundary."""
    alt = np.tile(np.array([0, 1], np.uint8), 4)  # one group that never opens an RLE run
    for width in (1, 3):
        for groups in (2046, 2047, 2048, 2049, 4095, 4096, 4097):
            for run in (8, 16, 24, 8 * 2048):
                lv = np.concatenate([np.tile(alt, groups), np.full(run, 1, np.uint8), np.tile(alt, 5000), np.full(9, 0, np.uint8)])
                assert R.rle_encode(ctx, width, lv) == bytes(O.rle_encode(width, lv)), (width, groups, run)
    # no RLE run at all: every chunk's count comes from the block start
    lv = np.tile(alt, 3 * 2048 + 5)
    assert R.rle_encode(ctx, 2, lv) == bytes(O.rle_encode(2, lv))


@pytest.mark.parametrize("width", [5, 6, 7, 8])
def test_wide_levels_round_trip(ctx, width):
    """Widths the reference cannot pack (bitpack.Pack stops at 4): checked through the decoder."""
    rng = np.random.default_rng(600 + width)
    for n in (1, 100, 5000, 33333):
        lv = _random_levels(rng, n, width, 0.8)
        enc = R.rle_encode(ctx, width, lv)
        dec, total = R.rle_decode(ctx, width, enc, n + 8)
        assert total >= n and np.array_equal(dec[:n], lv) and not dec[n:total].any()


# ---------------------------------------------------------------- page payloads
import helpers as H  # noqa: E402
from parquet_amd import _native as N  # noqa: E402
from parquet_amd.writer import Column, ParquetWriter, encode_pages  # noqa: E402


def _file_payloads(buf):
    """The page payloads of an uncompressed file, in file order."""
    pf = R.ParquetFile(buf)
    out = []
    for i in range(pf.num_pages):
        h = pf.page_header(i)
        a = h["pos"] + h["header_len"]
        out.append(bytes(buf[a:a + h["uncompressed_page_size"]]))
    return out


def _ref_minmax(values, np_type):
    """min starts at the type's maximum, max starts at zero (template_required.go:77-96)."""
    v = np.asarray(values, np_type)
    if np.issubdtype(np_type, np.floating):
        mn, mx = np.finfo(np_type).max, np_type(0)
    else:
        mn, mx = np.iinfo(np_type).max, np_type(0)
    for x in v:  # NaN never wins a Go comparison
        if x < mn:
            mn = x
        if x > mx:
            mx = x
    return np.array([mn], np_type).tobytes(), np.array([mx], np_type).tobytes()


def test_page_payloads_match_the_writer(ctx):
    """The same pages through the synthetic writer (whose level blocks are pinned to the reference encoder
    and whose PLAIN layout the oracle reads back) and through the device encoder."""
    rng = np.random.default_rng(70)
    cols = [Column("a", N.INT32, [0]), Column("b", N.INT64, [1]), Column("f", N.FLOAT, [1]), Column("d", N.DOUBLE, [0]),
            Column("s", N.BYTE_ARRAY, [1]), Column("t", N.BYTE_ARRAY, [0]), Column("k", N.BOOLEAN, [1]),
            Column("r.x", N.INT32, [2, 1]), Column("r.s", N.BYTE_ARRAY, [2, 0])]
    w = ParquetWriter(cols, stats=False)
    w.begin_row_group()
    pages = []
    for ci, c in enumerate(cols):
        for nr in (1, 7, 1000, 5000, 20011):
            if c.max_def > 0:
                defs, reps, nv = H.gen_levels(c.types, nr, rng, 0.3)
            else:
                defs, reps, nv = None, None, nr
            pg = {"phys_type": c.phys_type, "max_def": c.max_def, "max_rep": c.max_rep, "defs": defs, "reps": reps}
            if c.phys_type == N.BYTE_ARRAY:
                chars, offs = H.random_strings(nv, rng, 0, 24, (0, 256))
                w.write_page(ci, chars, defs=defs, reps=reps, offsets=offs)
                pg["values"] = (chars, offs)
            else:
                v = H.random_values(c.phys_type, nv, rng)
                w.write_page(ci, v, defs=defs, reps=reps)
                pg["values"] = v
            pages.append(pg)
    w.end_row_group(1)
    buf = w.finish()
    w.close()
    want = _file_payloads(buf)
    got = encode_pages(ctx, pages)
    assert len(want) == len(got)
    for i, ((payload, info), pg) in enumerate(zip(got, pages)):
        assert payload == want[i], (i, pg["phys_type"], len(payload), len(want[i]))
        n_levels = len(pg["defs"]) if pg["defs"] is not None else 0
        if pg["max_def"] > 0:
            nv = len(pg["values"][1]) - 1 if pg["phys_type"] == N.BYTE_ARRAY else len(pg["values"])
            assert info["null_count"] == n_levels - nv
        np_type = {N.INT32: np.int32, N.INT64: np.int64, N.FLOAT: np.float32, N.DOUBLE: np.float64}.get(pg["phys_type"])
        if np_type is None or (pg["max_def"] > 0 and len(pg["values"]) == 0):
            assert info["min"] is None and info["max"] is None
        else:
            assert (info["min"], info["max"]) == _ref_minmax(pg["values"], np_type), i


def test_page_payloads_round_trip_through_the_oracle(ctx):
    """Device-encoded pages, framed by the writer as raw pages, read back by the oracle reader."""
    rng = np.random.default_rng(71)
    cols = [Column("a", N.INT32, [1]), Column("s", N.BYTE_ARRAY, [1]), Column("r.x", N.INT64, [2, 1])]
    pages, per_col = [], []
    for c in cols:
        mine = []
        for nr in (3000, 4096, 9001):
            defs, reps, nv = H.gen_levels(c.types, nr, rng, 0.2)
            pg = {"phys_type": c.phys_type, "max_def": c.max_def, "max_rep": c.max_rep, "defs": defs, "reps": reps}
            pg["values"] = H.random_strings(nv, rng, 0, 16, (97, 123)) if c.phys_type == N.BYTE_ARRAY else H.random_values(c.phys_type, nv, rng)
            mine.append(len(pages))
            pages.append(pg)
        per_col.append(mine)
    enc = encode_pages(ctx, pages)
    w = ParquetWriter(cols, stats=False)
    w.begin_row_group()
    for ci, mine in enumerate(per_col):
        for i in mine:
            w.write_raw_page(ci, enc[i][0], len(pages[i]["defs"]))
    w.end_row_group(1)
    buf = w.finish()
    w.close()
    o = H.oracle_file(buf)
    for ci, (c, mine) in enumerate(zip(cols, per_col)):
        ch = o[(0, c.path)]
        assert ch.error == 0
        assert np.array_equal(ch.defs, np.concatenate([pages[i]["defs"] for i in mine]))
        if c.max_rep:
            assert np.array_equal(ch.reps, np.concatenate([pages[i]["reps"] for i in mine]))
        if c.phys_type == N.BYTE_ARRAY:
            assert ch.values.tobytes() == b"".join(np.asarray(pages[i]["values"][0]).tobytes() for i in mine)
        else:
            assert ch.values.tobytes() == b"".join(np.asarray(pages[i]["values"]).tobytes() for i in mine)
    # and the decoder reads them too
    H.check_file(buf, ctx)


def test_page_payload_edge_cases(ctx):
    pages = [
        {"phys_type": N.INT32, "max_def": 1, "max_rep": 0, "defs": np.zeros(100, np.uint8), "reps": None,
         "values": np.zeros(0, np.int32)},                                             # all null
        {"phys_type": N.INT32, "max_def": 0, "max_rep": 0, "defs": None, "reps": None, "values": np.zeros(0, np.int32)},  # empty
        {"phys_type": N.INT32, "max_def": 0, "max_rep": 0, "defs": None, "reps": None,
         "values": np.array([-5, -1, -7], np.int32)},                                  # all negative: max stays 0
        {"phys_type": N.INT32, "max_def": 0, "max_rep": 0, "defs": None, "reps": None, "is_unsigned": True,
         "values": np.array([0xFFFFFFF0, 3, 7], np.uint32).view(np.int32)},            # UINT_32 compares unsigned
        {"phys_type": N.INT64, "max_def": 0, "max_rep": 0, "defs": None, "reps": None, "is_unsigned": True,
         "values": np.array([2**63 + 5, 9], np.uint64).view(np.int64)},
        {"phys_type": N.BYTE_ARRAY, "max_def": 0, "max_rep": 0, "defs": None, "reps": None,
         "values": (np.zeros(0, np.uint8), np.zeros(4, np.int32))},                    # three empty strings
        {"phys_type": N.BOOLEAN, "max_def": 0, "max_rep": 0, "defs": None, "reps": None,
         "values": np.array([1, 0, 1, 1, 0, 0, 0, 0, 1], np.uint8)},                   # 9 bools: 2 bytes
    ]
    got = encode_pages(ctx, pages)
    assert got[0][0] == bytes(O.rle_encode(1, np.zeros(100, np.uint8))) and got[0][1]["null_count"] == 100
    assert got[0][1]["min"] is None and got[0][1]["def_bytes"] == len(got[0][0])
    assert got[1][0] == b"" and got[1][1]["min"] == np.array([2**31 - 1], np.int32).tobytes() and got[1][1]["max"] == bytes(4)
    assert got[2][0] == np.array([-5, -1, -7], np.int32).tobytes()
    assert got[2][1]["min"] == np.array([-7], np.int32).tobytes() and got[2][1]["max"] == bytes(4)
    assert got[3][1]["min"] == np.array([3], np.uint32).tobytes() and got[3][1]["max"] == np.array([0xFFFFFFF0], np.uint32).tobytes()
    assert got[4][1]["min"] == np.array([9], np.uint64).tobytes() and got[4][1]["max"] == np.array([2**63 + 5], np.uint64).tobytes()
    assert got[5][0] == bytes(12) and got[5][1]["min"] is None
    assert got[6][0] == bytes([0b00001101, 0b00000001])


def test_blocks_of_more_than_1024_tiles(ctx):
    """The per-block scans work in batches of 1024 tiles / elements with carries."""
    rng = np.random.default_rng(12)
    n = 2048 * 1100 + 5
    lv = rng.integers(0, 2, n).astype(np.uint8)
    lv[2048 * 10 + 7:2048 * 1090 + 3] = 1      # one run over 1080 tiles
    lv[2048 * 1095:2048 * 1097] = 0            # and one over exactly two
    assert R.rle_encode(ctx, 1, lv) == bytes(O.rle_encode(1, lv))
    lv2 = rng.integers(0, 4, n).astype(np.uint8)  # no long runs: ~275k bit-packed groups
    assert R.rle_encode(ctx, 2, lv2) == bytes(O.rle_encode(2, lv2))
    assert R.rle_encode(ctx, 1, np.ones(n, np.uint8)) == bytes(O.rle_encode(1, np.ones(n, np.uint8)))
