"""Synthetic-file writer (libpqsynth.so): files laid out exactly as parsyl/parquet writes them with the
`Uncompressed` option.  Support code for tests and benchmarks, not part of the decode path.

Mirrors the write side of the generated API loosely: columns are declared like the generated
`Fields()` (path + repetition `Types`, cmd/parquetgen/gen/template_optional.go:15-22), a row group
corresponds to one `ParquetWriter.Write()` call and pages to the chained writers of `MaxPageSize`
rows (template.go:164-215).
"""
import ctypes as C

import numpy as np

from . import _native as N

REQUIRED, OPTIONAL, REPEATED = 0, 1, 2
UINT_32, UINT_64, UTF8 = 13, 14, 0


def bits_len(v):
    return int(v).bit_length()


class Column:
    def __init__(self, path, phys_type, types, converted_type=-1):
        self.path = path
        self.phys_type = phys_type
        self.types = list(types)
        self.converted_type = converted_type
        self.max_def = sum(1 for t in self.types if t in (OPTIONAL, REPEATED))
        self.max_rep = sum(1 for t in self.types if t == REPEATED)


class ParquetWriter:
    def __init__(self, columns, stats=True, codec=0):
        """codec: 0 UNCOMPRESSED, 1 SNAPPY (every page payload goes through the writer's own snappy compressor)."""
        self.columns = list(columns)
        self._w = N.synth.pqw_create(1 if stats else 0)
        self._open = False
        if codec:
            assert N.synth.pqw_set_codec(self._w, codec) == 0
        for c in self.columns:
            t = (C.c_int32 * len(c.types))(*c.types)
            rc = N.synth.pqw_add_column(self._w, c.path.encode(), c.phys_type, c.converted_type, t, len(c.types))
            if rc < 0:
                raise ValueError("bad column %r" % c.path)

    def reserve(self, nbytes):
        N.synth.pqw_reserve(self._w, int(nbytes))

    def begin_row_group(self):
        assert N.synth.pqw_begin_row_group(self._w) == 0
        self._open = True

    def end_row_group(self, num_rows):
        assert N.synth.pqw_end_row_group(self._w, num_rows) == 0
        self._open = False

    def write_page(self, col, values, defs=None, reps=None, offsets=None):
        """One data page.  values: numpy array of the non-null values (fixed width), uint8 0/1 (bool),
        or the concatenated chars with `offsets` (strings)."""
        c = self.columns[col]
        values = np.ascontiguousarray(values)
        if c.phys_type == N.BYTE_ARRAY:
            offsets = np.ascontiguousarray(offsets, np.int32)
            n_values = offsets.size - 1
            values = values.view(np.uint8) if values.size else np.zeros(1, np.uint8)
        elif c.phys_type == N.BOOLEAN:
            values = values.astype(np.uint8)
            n_values = values.size
        else:
            want = 4 if c.phys_type in (N.INT32, N.FLOAT) else 8
            assert values.dtype.itemsize == want, (values.dtype, c.path)
            n_values = values.size
        if c.max_def > 0:
            defs = np.ascontiguousarray(defs, np.uint8)
            n_levels = defs.size
            dp = defs.ctypes.data
        else:
            n_levels, dp = n_values, None
        rp = None
        if c.max_rep > 0:
            reps = np.ascontiguousarray(reps, np.uint8)
            assert reps.size == n_levels
            rp = reps.ctypes.data
        op = offsets.ctypes.data if offsets is not None else None
        vp = values.ctypes.data if values.size else None
        rc = N.synth.pqw_write_page(self._w, col, dp, rp, n_levels, vp, n_values, op)
        if rc != 0:
            raise ValueError("pqw_write_page failed for %s" % c.path)

    def write_raw_page(self, col, payload, num_values):
        payload = bytes(payload)
        buf = (C.c_uint8 * max(len(payload), 1)).from_buffer_copy(payload or b"\0")
        rc = N.synth.pqw_write_raw_page(self._w, col, buf, len(payload), num_values)
        if rc != 0:
            raise ValueError("pqw_write_raw_page failed")

    def finish(self):
        assert N.synth.pqw_finish(self._w) == 0
        n = C.c_int64()
        p = N.synth.pqw_data(self._w, C.byref(n))
        if n.value < (1 << 31):
            return C.string_at(p, n.value)
        # ctypes.string_at takes a C int: copy larger files into a bytearray
        out = bytearray(n.value)
        C.memmove((C.c_char * n.value).from_buffer(out), p, n.value)
        return out

    def finish_into(self, alloc):
        """Like finish(), but copies the file into a buffer obtained from alloc(nbytes) → address."""
        assert N.synth.pqw_finish(self._w) == 0
        n = C.c_int64()
        p = N.synth.pqw_data(self._w, C.byref(n))
        dst = alloc(n.value)
        C.memmove(dst, p, n.value)
        return dst, n.value

    def close(self):
        if self._w:
            N.synth.pqw_destroy(self._w)
            self._w = None

    def __del__(self):
        self.close()


def encode_levels(width, levels):
    levels = np.ascontiguousarray(levels, np.uint8)
    cap = levels.size * 2 + 64
    out = np.zeros(cap, np.uint8)
    n = N.synth.pqw_encode_levels(width, levels.ctypes.data if levels.size else None, levels.size, out.ctypes.data, cap)
    assert n >= 0
    return out[:n].tobytes()


def write_flat_table(columns, data, rows_per_group, rows_per_page, stats=True):
    """Writes flat (non-repeated) columns.  data[i] is either a numpy array (required column), or a
    pair (values_of_all_rows, valid_mask) for an optional column; strings are lists of bytes/None."""
    w = ParquetWriter(columns, stats=stats)
    n_rows = None
    for c, d in zip(columns, data):
        n = len(d[0]) if isinstance(d, tuple) else len(d)
        n_rows = n if n_rows is None else n_rows
        assert n == n_rows
    for g0 in range(0, max(n_rows, 1), rows_per_group):
        g1 = min(n_rows, g0 + rows_per_group)
        if g1 <= g0:
            break
        w.begin_row_group()
        for ci, (c, d) in enumerate(zip(columns, data)):
            for p0 in range(g0, g1, rows_per_page):
                p1 = min(g1, p0 + rows_per_page)
                if isinstance(d, tuple):
                    vals, mask = d
                    m = np.asarray(mask[p0:p1], bool)
                    defs = m.astype(np.uint8) * c.max_def
                else:
                    vals, m, defs = d, None, None
                if c.phys_type == N.BYTE_ARRAY:
                    items = [vals[i] for i in range(p0, p1) if m is None or m[i - p0]]
                    lens = np.array([len(s) for s in items], np.int64)
                    offsets = np.zeros(len(items) + 1, np.int32)
                    np.cumsum(lens, out=offsets[1:])
                    chars = np.frombuffer(b"".join(items), np.uint8)
                    w.write_page(ci, chars, defs=defs, offsets=offsets)
                else:
                    v = np.asarray(vals[p0:p1])
                    if m is not None:
                        v = v[m]
                    w.write_page(ci, v, defs=defs)
        w.end_row_group(g1 - g0)
    out = w.finish()
    w.close()
    return out


def snappy_compress(data):
    """The synthetic writer's snappy compressor on its own (raw block format)."""
    data = bytes(data)
    cap = 32 + len(data) + len(data) // 6
    out = (C.c_uint8 * cap)()
    src = (C.c_uint8 * max(len(data), 1)).from_buffer_copy(data or b"\0")
    n = N.synth.pqw_snappy_compress(src, len(data), out, cap)
    assert n >= 0
    return bytes(out[:n])


def encode_pages(ctx, pages):
    """pqd_encode_pages: page payloads of the write path on the device.

    pages: list of dicts with phys_type, max_def, max_rep, values (numpy array; for strings a
    (chars uint8 array, offsets int32 array) pair), optional defs / reps (uint8) and is_unsigned.
    Returns a list of (payload bytes, result) where result has rep_bytes, def_bytes, null_count, min, max
    (bytes or None)."""
    import numpy as np
    parts, pos = [], 0

    def put(arr, align):
        nonlocal pos
        pad = (-pos) % align
        if pad:
            parts.append(b"\0" * pad)
            pos += pad
        off = pos
        b = np.ascontiguousarray(arr).tobytes()
        parts.append(b)
        pos += len(b)
        return off

    descs = (N.EncPage * max(len(pages), 1))()
    for d, pg in zip(descs, pages):
        d.phys_type, d.max_def, d.max_rep = pg["phys_type"], pg["max_def"], pg["max_rep"]
        d.is_unsigned = 1 if pg.get("is_unsigned") else 0
        d.defs_off = d.reps_off = d.offsets_off = -1
        if pg["phys_type"] == N.BYTE_ARRAY:
            chars, offs = pg["values"]
            offs = np.ascontiguousarray(offs, np.int32)
            d.n_values = offs.size - 1
            d.values_off = put(np.ascontiguousarray(chars, np.uint8), 8)
            d.offsets_off = put(offs, 8)
        else:
            v = np.ascontiguousarray(pg["values"])
            if pg["phys_type"] == N.BOOLEAN:
                v = v.astype(np.uint8)
            d.n_values = v.size
            d.values_off = put(v, 8)
        if pg["max_def"] > 0:
            defs = np.ascontiguousarray(pg["defs"], np.uint8)
            d.n_levels = defs.size
            d.defs_off = put(defs, 8)
            if pg["max_rep"] > 0:
                d.reps_off = put(np.ascontiguousarray(pg["reps"], np.uint8), 8)
        else:
            d.n_levels = d.n_values
    src = b"".join(parts) or b"\0"
    host = np.frombuffer(src, np.uint8)
    cap = 64 + len(src) * 2 + 64 * len(pages)
    out = np.zeros(cap, np.uint8)
    res = (N.EncResult * max(len(pages), 1))()
    n = C.c_int64()
    rc = N.decode.pqd_encode_pages(ctx.handle, descs, len(pages), host.ctypes.data, len(src), out.ctypes.data, cap, res, C.byref(n))
    if rc != 0:
        m = N.decode.pqd_last_error(ctx.handle)
        raise N.PqdError(rc, m.decode() if m else "")
    ret = []
    for r in res[:len(pages)]:
        info = {"rep_bytes": r.rep_bytes, "def_bytes": r.def_bytes, "null_count": r.null_count,
                "min": bytes(r.min[:r.minmax_len]) if r.minmax_len else None,
                "max": bytes(r.max[:r.minmax_len]) if r.minmax_len else None}
        ret.append((out[r.payload_off:r.payload_off + r.payload_len].tobytes(), info))
    return ret
