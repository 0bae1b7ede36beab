"""Shared test helpers: synthetic inputs, the oracle run over a whole file, the GPU run over a whole
file, and the comparison between the two."""
import numpy as np

import oracle as O
from parquet_amd import _native as N
from parquet_amd import reader as R
from parquet_amd.writer import Column, ParquetWriter, REQUIRED, OPTIONAL, REPEATED


def splitmix64(seed, n, start=0):
    """splitmix64(seed + i) for i in [start, start+n) as uint64 (vectorised)."""
    with np.errstate(over="ignore"):
        z = (np.arange(start, start + n, dtype=np.uint64) + np.uint64(seed)) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


# ------------------------------------------------------------------ nested level generator
def gen_levels(types, n_rows, rng, p_null=0.3, max_rep_count=3):
    """Random Dremel levels for one leaf whose path has the given repetition types.
    Returns (defs, reps, n_values, rows) where rows = number of records."""
    defs, reps = [], []
    max_def = sum(1 for t in types if t in (OPTIONAL, REPEATED))
    rep_level_at = []
    r = 0
    for t in types:
        if t == REPEATED:
            r += 1
        rep_level_at.append(r)

    def emit(i, cur_def, rep):
        if i == len(types):
            defs.append(cur_def)
            reps.append(rep)
            return
        t = types[i]
        if t == REQUIRED:
            emit(i + 1, cur_def, rep)
        elif t == OPTIONAL:
            if rng.random() < p_null:
                defs.append(cur_def)
                reps.append(rep)
            else:
                emit(i + 1, cur_def + 1, rep)
        else:
            k = int(rng.integers(0, max_rep_count + 1))
            if k == 0:
                defs.append(cur_def)
                reps.append(rep)
            for j in range(k):
                emit(i + 1, cur_def + 1, rep if j == 0 else rep_level_at[i])

    for _ in range(n_rows):
        emit(0, 0, 0)
    d = np.array(defs, np.uint8)
    rp = np.array(reps, np.uint8)
    return d, rp, int((d == max_def).sum())


def random_values(phys_type, n, rng, str_len=(0, 24)):
    if phys_type == N.INT32:
        return rng.integers(-2**31, 2**31 - 1, n, dtype=np.int64).astype(np.int32)
    if phys_type == N.INT64:
        return rng.integers(-2**63, 2**63 - 1, n, dtype=np.int64)
    if phys_type == N.FLOAT:
        return rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32).view(np.float32)  # any bit pattern
    if phys_type == N.DOUBLE:
        return rng.integers(0, 2**63, n, dtype=np.uint64).view(np.float64)
    if phys_type == N.BOOLEAN:
        return (rng.random(n) < 0.5).astype(np.uint8)
    raise ValueError(phys_type)


def random_strings(n, rng, lo=0, hi=24, alphabet=(0, 256)):
    lens = rng.integers(lo, hi + 1, n)
    offsets = np.zeros(n + 1, np.int32)
    np.cumsum(lens, out=offsets[1:])
    if alphabet == "utf16":  # letters with a zero byte behind each: most positions then look like a length prefix
        chars = rng.integers(97, 123, int(offsets[-1]), dtype=np.uint8)
        chars[1::2] = 0
    else:
        chars = rng.integers(alphabet[0], alphabet[1], int(offsets[-1]), dtype=np.uint8)
    return chars, offsets


def build_file(columns, n_rows, rows_per_group, rows_per_page, rng, p_null=0.3, str_len=(0, 24), stats=True,
               alphabet=(0, 256), codec=0):
    """Random file for arbitrary (nested) columns.  Pages are cut on record boundaries every
    rows_per_page rows, like the reference's chained writers."""
    w = ParquetWriter(columns, stats=stats, codec=codec)
    for g0 in range(0, n_rows, rows_per_group):
        g1 = min(n_rows, g0 + rows_per_group)
        w.begin_row_group()
        for ci, c in enumerate(columns):
            for p0 in range(g0, g1, rows_per_page):
                p1 = min(g1, p0 + rows_per_page)
                nr = p1 - p0
                if c.max_def > 0:
                    defs, reps, nv = gen_levels(c.types, nr, rng, p_null)
                else:
                    defs, reps, nv = None, None, nr
                if c.phys_type == N.BYTE_ARRAY:
                    chars, offs = random_strings(nv, rng, str_len[0], str_len[1], alphabet)
                    w.write_page(ci, chars, defs=defs, reps=reps, offsets=offs)
                else:
                    w.write_page(ci, random_values(c.phys_type, nv, rng), defs=defs, reps=reps)
        w.end_row_group(g1 - g0)
    buf = w.finish()
    w.close()
    return buf


# ------------------------------------------------------------------ oracle / GPU over a whole file
def oracle_file(buf, pf=None):
    """Runs the CPU oracle over every column chunk, reading the file sequentially like the reference
    reader.  Returns {(rg, column_name): oracle Chunk}."""
    pf = pf or R.ParquetFile(buf)
    out = {}
    for rg in range(pf.num_row_groups):
        for c in range(pf.num_columns):
            try:
                m = pf.chunk_meta(rg, c)
            except N.PqdError:
                break
            name, ci = pf.columns[m["column"]]
            ch = O.read_chunk(buf, m["start_pos"], ci.phys_type, ci.max_def, ci.max_rep, m["num_values"],
                              m["total_compressed_size"], codec=pf.chunk_codec(rg, c))
            ch.max_def, ch.max_rep, ch.phys_type = ci.max_def, ci.max_rep, ci.phys_type
            out[(rg, name)] = ch
    return out


def gpu_file(buf, ctx, rg_batch=None, pf=None):
    """Decodes the file on the GPU, rg_batch row groups per pqd_decode call (None = all at once).
    Returns {(rg, column_name): dict of numpy arrays}."""
    pf = pf or R.ParquetFile(buf)
    out = {}
    step = rg_batch or max(pf.num_row_groups, 1)
    for g0 in range(0, pf.num_row_groups, step):
        cnt = min(step, pf.num_row_groups - g0)
        chunks, nc, pages, npg = pf.descriptors(g0, cnt)
        plan = R.decode_host(ctx, chunks, nc, pages, npg, pf.address, pf.size)
        try:
            per_rg = nc // cnt
            for k in range(nc):
                name = pf.columns[chunks[k].column][0]
                out[(g0 + k // per_rg, name)] = plan.column(k)
        finally:
            plan.close()
    return out


def assert_chunk_equal(o, g, key=""):
    """Oracle chunk vs GPU chunk, bit exact."""
    assert o.error == 0, (key, "oracle error", o.error)
    if o.max_def > 0:
        assert np.array_equal(o.defs, g["defs"]), (key, "defs")
        assert np.array_equal(O.validity(o.defs, o.max_def), g["validity"]), (key, "validity")
        lo, vo = O.row_offsets(o.defs, o.reps, o.max_def, o.max_rep)
        assert np.array_equal(lo, g["row_level_off"]), (key, "row_level_off")
        assert np.array_equal(vo, g["row_value_off"]), (key, "row_value_off")
    if o.max_rep > 0:
        assert np.array_equal(o.reps, g["reps"]), (key, "reps")
    assert np.array_equal(o.sizes, g["sizes"]), (key, "sizes", o.sizes[:8], g["sizes"][:8])
    if o.phys_type == N.BYTE_ARRAY:
        assert np.array_equal(o.offsets, g["offsets"]), (key, "offsets")
        assert np.array_equal(o.values, g["chars"]), (key, "chars")
    else:
        assert np.array_equal(o.values, g["values"].view(np.uint8)), (key, "values")


def check_file(buf, ctx, rg_batch=None):
    pf = R.ParquetFile(buf)
    o = oracle_file(buf, pf)
    g = gpu_file(buf, ctx, rg_batch, pf)
    assert set(o) == set(g) and len(o) > 0
    for k in sorted(o):
        assert_chunk_equal(o[k], g[k], k)
    return o, g
