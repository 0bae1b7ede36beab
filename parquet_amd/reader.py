"""Host-side mirror of the parquetgen-generated reader over the C ABI of libpqdecode.so.

Same names and meaning as the reference (cmd/parquetgen/gen/template.go:235-382):
NewParquetReader → ParquetReader(buf); Rows() → rows(); Levels() → levels(); Next()/Scan() → next()/scan();
Error() → error().  readRowGroup (template.go:320-351) becomes one batched decode call for all column
chunks of the row group.  All decoding happens on the GPU through the library; this module only moves
descriptors and fetches results.
"""
import ctypes as C

import numpy as np

from . import _native as N


def _check(rc, ctx=None):
    if rc != 0:
        msg = ""
        if ctx:
            m = N.decode.pqd_last_error(ctx)
            msg = m.decode() if m else ""
        raise N.PqdError(rc, msg)


class Context:
    """pqd_ctx: one per reader (the reference reader is single goroutine)."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        rc = N.decode.pqd_create(device, C.byref(self._h))
        if rc != 0:
            raise N.PqdError(rc, "pqd_create(device=%d): no usable HIP device" % device)

    @property
    def handle(self):
        return self._h

    def close(self):
        if self._h:
            N.decode.pqd_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ParquetFile:
    """pqd_file: footer + sequential page-header walk (no GPU needed)."""

    def __init__(self, buf):
        self._buf = buf  # keep alive: the file object points into it
        self._arr = np.frombuffer(buf, np.uint8)
        self._h = C.c_void_p()
        err = C.create_string_buffer(256)
        rc = N.decode.pqd_file_open(self._arr.ctypes.data, self._arr.size, C.byref(self._h), err, 256)
        if rc != 0:
            raise N.PqdError(rc, err.value.decode())
        self.num_rows = N.decode.pqd_file_num_rows(self._h)
        self.num_columns = N.decode.pqd_file_num_columns(self._h)
        self.num_row_groups = N.decode.pqd_file_num_row_groups(self._h)
        self.num_pages = N.decode.pqd_file_num_pages(self._h)
        self.columns = []
        for c in range(self.num_columns):
            ci = N.ColumnInfo()
            _check(N.decode.pqd_file_column(self._h, c, C.byref(ci)))
            name = N.decode.pqd_file_column_path(self._h, c).decode()
            self.columns.append((name, ci))

    @property
    def address(self):
        return self._arr.ctypes.data

    @property
    def size(self):
        return self._arr.size

    def row_group_rows(self, rg):
        return N.decode.pqd_file_row_group_rows(self._h, rg)

    def descriptors(self, rg_first, rg_count):
        """(chunk descriptor array, page descriptor array) for a range of row groups (copies)."""
        cp, pp = C.POINTER(N.ChunkDesc)(), C.POINTER(N.PageDesc)()
        nc, np_ = C.c_size_t(), C.c_size_t()
        _check(N.decode.pqd_file_descriptors(self._h, rg_first, rg_count, C.byref(cp), C.byref(nc), C.byref(pp), C.byref(np_)))
        chunks = (N.ChunkDesc * nc.value)()
        pages = (N.PageDesc * max(np_.value, 1))()
        C.memmove(chunks, cp, C.sizeof(N.ChunkDesc) * nc.value)
        if np_.value:
            C.memmove(pages, pp, C.sizeof(N.PageDesc) * np_.value)
        return chunks, nc.value, pages, np_.value

    def page_header(self, i):
        out = (C.c_int32 * 8)()
        pos = C.c_int64()
        _check(N.decode.pqd_file_page_header(self._h, i, out, C.byref(pos)))
        keys = ("type", "uncompressed_page_size", "compressed_page_size", "num_values", "encoding",
                "definition_level_encoding", "repetition_level_encoding", "header_len")
        d = dict(zip(keys, list(out)))
        d["pos"] = pos.value
        return d

    def chunk_meta(self, rg, chunk):
        out = (C.c_int64 * 6)()
        _check(N.decode.pqd_file_chunk_meta(self._h, rg, chunk, out))
        return dict(zip(("column", "num_values", "total_compressed_size", "start_pos", "first_page", "n_pages"), list(out)))

    def chunk_codec(self, rg, chunk):
        return N.decode.pqd_file_chunk_codec(self._h, rg, chunk)

    def close(self):
        if self._h:
            N.decode.pqd_file_close(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_DTYPES = {N.INT32: np.int32, N.INT64: np.int64, N.FLOAT: np.float32, N.DOUBLE: np.float64}


class Plan:
    """pqd_plan: a batch of column chunks planned for decode."""

    def __init__(self, ctx, chunks, n_chunks, pages, n_pages, src_host_addr, src_len):
        self.ctx = ctx
        self._chunks = chunks
        self.n_chunks = n_chunks
        self._h = C.c_void_p()
        rc = N.decode.pqd_plan_create(ctx.handle, chunks, n_chunks, pages, n_pages, src_host_addr, src_len, C.byref(self._h))
        _check(rc, ctx.handle)

    def run(self, src_device_addr):
        _check(N.decode.pqd_plan_run(self._h, src_device_addr), self.ctx.handle)

    def sync(self):
        _check(N.decode.pqd_plan_sync(self._h), self.ctx.handle)

    def last_ms(self):
        a, b = C.c_float(), C.c_float()
        _check(N.decode.pqd_plan_last_ms(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def run_ms(self, runs_back):
        a, b = C.c_float(), C.c_float()
        _check(N.decode.pqd_plan_run_ms(self._h, runs_back, C.byref(a), C.byref(b)))
        return a.value, b.value

    def bytes(self):
        i, o, c = C.c_int64(), C.c_int64(), C.c_int64()
        _check(N.decode.pqd_plan_bytes(self._h, C.byref(i), C.byref(o), C.byref(c)))
        return i.value, o.value, c.value

    def info(self, chunk):
        ci = N.ChunkInfo()
        _check(N.decode.pqd_plan_chunk_info(self._h, chunk, C.byref(ci)))
        return ci

    def device_ptr(self, chunk, which):
        return N.decode.pqd_plan_device_ptr(self._h, chunk, which)

    def string_fallbacks(self):
        """BYTE_ARRAY chunks of the last synced run that the single pass gave up (decoded by the two passes instead)."""
        return int(N.decode.pqd_plan_string_fallbacks(self._h))

    def fetch(self, chunk, which, dtype=np.uint8):
        ci = self.info(chunk)
        sizes = {
            N.OUT_DEFS: ci.n_levels, N.OUT_REPS: ci.n_levels, N.OUT_VALUES: ci.value_bytes,
            N.OUT_OFFSETS: (ci.n_values + 1) * 4, N.OUT_CHARS: ci.chars_bytes,
            N.OUT_VALIDITY: (ci.n_levels + 7) // 8, N.OUT_PAGE_NONNULL: ci.n_pages * 8,
            N.OUT_ROW_LEVEL_OFF: (ci.n_rows + 1) * 4, N.OUT_ROW_VALUE_OFF: (ci.n_rows + 1) * 4,
            N.OUT_VALUE_INDEX: ((ci.n_levels + 63) // 64 + 1) * 4,
        }
        if not self.device_ptr(chunk, which):
            return np.zeros(0, dtype)
        buf = np.zeros(max(sizes[which], 1), np.uint8)
        n = N.decode.pqd_plan_fetch(self._h, chunk, which, buf.ctypes.data, buf.size)
        if n < 0:
            _check(int(n), self.ctx.handle)
        return buf[:n].view(dtype)

    def scatter(self, chunk):
        """Null-mask scatter of one chunk (pqd_plan_scatter): one entry per level, 0 for nulls; for strings the
        n_levels + 1 slot offsets into CHARS.  Returns a numpy array copied from the device."""
        cd = self._chunks[chunk]
        ci = self.info(chunk)
        strings = cd.phys_type == N.BYTE_ARRAY
        dt = np.dtype(np.int32) if strings else np.dtype(np.uint8) if cd.phys_type == N.BOOLEAN else np.dtype(_DTYPES[cd.phys_type])
        n = ci.n_levels + (1 if strings else 0)
        nbytes = max(n * dt.itemsize, 16)
        dev = N.decode.pqd_device_alloc(self.ctx.handle, nbytes)
        if not dev:
            raise MemoryError("pqd_device_alloc")
        try:
            _check(N.decode.pqd_plan_scatter(self._h, chunk, dev, nbytes), self.ctx.handle)
            self.sync()
            out = np.zeros(nbytes, np.uint8)
            _check(N.decode.pqd_copy_to_host(self.ctx.handle, out.ctypes.data, dev, nbytes), self.ctx.handle)
        finally:
            N.decode.pqd_device_free(self.ctx.handle, dev)
        return out[: n * dt.itemsize].view(dt)

    def column(self, chunk):
        """All decoded arrays of one chunk as a dict of numpy arrays (the Field's vals / Defs / Reps)."""
        cd = self._chunks[chunk]
        out = {"defs": None, "reps": None}
        if cd.max_def > 0:
            out["defs"] = self.fetch(chunk, N.OUT_DEFS)
            out["validity"] = self.fetch(chunk, N.OUT_VALIDITY)
        if cd.max_rep > 0:
            out["reps"] = self.fetch(chunk, N.OUT_REPS)
            out["row_level_off"] = self.fetch(chunk, N.OUT_ROW_LEVEL_OFF, np.int32)
            out["row_value_off"] = self.fetch(chunk, N.OUT_ROW_VALUE_OFF, np.int32)
        elif cd.max_def > 0:
            # not repeated: row r is level r (write_optional.go:22-31); its first value is the number of
            # non-null levels before it
            n = out["defs"].size
            out["row_level_off"] = np.arange(n + 1, dtype=np.int64)
            out["row_value_off"] = value_offsets(out["validity"], self.fetch(chunk, N.OUT_VALUE_INDEX, np.uint32), n)
        out["sizes"] = self.fetch(chunk, N.OUT_PAGE_NONNULL, np.int64)
        if cd.phys_type == N.BYTE_ARRAY:
            out["offsets"] = self.fetch(chunk, N.OUT_OFFSETS, np.int32)
            out["chars"] = self.fetch(chunk, N.OUT_CHARS)
        elif cd.phys_type == N.BOOLEAN:
            out["values"] = self.fetch(chunk, N.OUT_VALUES)
        else:
            out["values"] = self.fetch(chunk, N.OUT_VALUES, _DTYPES[cd.phys_type])
        return out

    def close(self):
        if self._h:
            N.decode.pqd_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def value_offsets(validity, value_index, n_levels):
    """First value of every row of a non-repeated optional column (row r = level r), n_levels + 1 entries, from the
    device's two products: the validity bitmap and the value index (non-null levels in front of every 64-level
    word): word base + set bits of the word below the level."""
    bits = np.unpackbits(validity, bitorder="little")[:n_levels].astype(np.int64)
    pad = (-n_levels) % 64
    words = np.concatenate((bits, np.zeros(pad, np.int64))).reshape(-1, 64)
    within = np.cumsum(words, axis=1) - words            # set bits of the word below each level
    out = np.empty(n_levels + 1, np.int64)
    out[:n_levels] = (value_index[:words.shape[0], None].astype(np.int64) + within).reshape(-1)[:n_levels]
    out[n_levels] = int(value_index[(n_levels + 63) // 64])
    return out


def decode_host(ctx, chunks, n_chunks, pages, n_pages, src_addr, src_len):
    """pqd_decode: host bytes in → synced plan with results in HBM."""
    pl = Plan.__new__(Plan)
    pl.ctx, pl._chunks, pl.n_chunks, pl._h = ctx, chunks, n_chunks, C.c_void_p()
    rc = N.decode.pqd_decode(ctx.handle, chunks, n_chunks, pages, n_pages, src_addr, src_len, C.byref(pl._h))
    if rc != 0:
        m = N.decode.pqd_last_error(ctx.handle)
        pl.close()
        raise N.PqdError(rc, m.decode() if m else "")
    return pl


def snappy_decode_blocks(ctx, blocks):
    """snappy.Decode of several raw-format blocks on the device (pqd_snappy_sizes + pqd_snappy_decode).
    Returns the list of decoded byte strings."""
    blocks = [bytes(b) for b in blocks]
    src = b"".join(blocks)
    arr = (N.SnappyBlock * max(len(blocks), 1))()
    pos = 0
    for i, b in enumerate(blocks):
        arr[i].src_off, arr[i].comp_len = pos, len(b)
        pos += len(b)
    host = np.frombuffer(src or b"\0", np.uint8)
    rc = N.decode.pqd_snappy_sizes(host.ctypes.data, len(src), arr, len(blocks))
    if rc != 0:
        raise N.PqdError(rc, "bad snappy preamble")
    total = 0
    for i in range(len(blocks)):
        arr[i].dst_off = total
        total += (arr[i].decoded_len + 16 + 255) // 256 * 256
    dsrc = N.decode.pqd_device_alloc(ctx.handle, max(len(src), 1))
    ddst = N.decode.pqd_device_alloc(ctx.handle, max(total, 1))
    try:
        if not dsrc or not ddst:
            raise N.PqdError(-3, "device allocation failed")
        _check(N.decode.pqd_copy_to_device(ctx.handle, dsrc, host.ctypes.data, max(len(src), 1)), ctx.handle)
        _check(N.decode.pqd_snappy_decode(ctx.handle, arr, len(blocks), dsrc, len(src), ddst, total), ctx.handle)
        out = np.zeros(max(total, 1), np.uint8)
        _check(N.decode.pqd_copy_to_host(ctx.handle, out.ctypes.data, ddst, max(total, 1)), ctx.handle)
        return [out[arr[i].dst_off:arr[i].dst_off + arr[i].decoded_len].tobytes() for i in range(len(blocks))]
    finally:
        if dsrc:
            N.decode.pqd_device_free(ctx.handle, dsrc)
        if ddst:
            N.decode.pqd_device_free(ctx.handle, ddst)


def rle_decode(ctx, width, block, max_values, elem_size=1):
    """rle.Read for one level block (int32 length + runs): (values, number of values in the block)."""
    dt = {1: np.uint8, 2: np.uint16, 4: np.uint32}[elem_size]
    out = np.zeros(max(max_values, 1), dt)
    n = C.c_int64()
    block = bytes(block)
    rc = N.decode.pqd_rle_decode(ctx.handle, width, block, len(block), elem_size, out.ctypes.data, max_values, C.byref(n))
    _check(rc, ctx.handle)
    return out[:min(n.value, max_values)], n.value


def rle_encode(ctx, width, levels):
    """rle.(*RLE).Write + Bytes on the device: levels (uint8) -> int32 length + hybrid runs (bytes)."""
    lv = np.ascontiguousarray(levels, np.uint8)
    cap = 4 + (lv.size // 8 + 2) * max(width + 1, 6)
    out = np.zeros(cap, np.uint8)
    n = C.c_int64()
    rc = N.decode.pqd_rle_encode(ctx.handle, width, lv.ctypes.data if lv.size else None, lv.size, out.ctypes.data, cap, C.byref(n))
    _check(rc, ctx.handle)
    return out[:n.value].tobytes()


class Levels:
    """template.go:295-299"""

    def __init__(self, name, defs, reps):
        self.name, self.defs, self.reps = name, defs, reps


class ParquetReader:
    """NewParquetReader (template.go:235-270): reads the footer, then the first row group."""

    def __init__(self, buf, device=0, ctx=None):
        self._file = ParquetFile(buf)
        self._ctx = ctx or Context(device)
        self.field_names = [n for n, _ in self._file.columns]
        self._rows = self._file.num_rows
        self._cursor = 0
        self._rg_next = 0
        self._rg_cursor = 0
        self._rg_count = 0
        self._err = None
        self.fields = {}
        self._row_pos = {}
        self._read_row_group()

    def rows(self):
        return self._rows

    def error(self):
        return self._err

    def _read_row_group(self):
        """readRowGroup (template.go:320-351): every column chunk of the next row group, one batch."""
        self._rg_cursor = 0
        if self._rg_next >= self._file.num_row_groups:
            self._rg_count = 0
            return
        g = self._rg_next
        chunks, nc, pages, npg = self._file.descriptors(g, 1)
        plan = decode_host(self._ctx, chunks, nc, pages, npg, self._file.address, self._file.size)
        try:
            self.fields = {}
            for c in range(nc):
                name = self._file.columns[chunks[c].column][0]
                self.fields[name] = plan.column(c)
                self.fields[name]["desc"] = (chunks[c].phys_type, chunks[c].max_def, chunks[c].max_rep)
        finally:
            plan.close()
        self._row_pos = {n: 0 for n in self.fields}
        self._rg_count = self._file.row_group_rows(g)
        self._rg_next += 1

    def levels(self):
        """Levels (template.go:301-314) of the row group currently held."""
        out = []
        for name in self.field_names:
            f = self.fields.get(name)
            d = f["defs"] if f is not None else None
            r = f["reps"] if f is not None else None
            out.append(Levels(name, d, r))
        return out

    def next(self):
        """Next (template.go:357-371)."""
        if self._err is None and self._cursor >= self._rows:
            return False
        if self._rg_cursor >= self._rg_count:
            try:
                self._read_row_group()
            except N.PqdError as e:
                self._err = e
                return False
        self._cursor += 1
        self._rg_cursor += 1
        return True

    def scan(self):
        """Scan (template.go:373-382): the current row as {column: (values, defs, reps)}; the slices are the
        levels and values the generated write function would consume for this row."""
        if self._err is not None:
            return None
        row = {}
        for name in self.field_names:
            f = self.fields[name]
            r = self._row_pos[name]
            ptype, max_def, _ = f["desc"]
            if max_def == 0:
                v0, v1, l0, l1 = r, r + 1, 0, 0
            else:
                l0, l1 = f["row_level_off"][r], f["row_level_off"][r + 1]
                v0, v1 = f["row_value_off"][r], f["row_value_off"][r + 1]
            if ptype == N.BYTE_ARRAY:
                o = f["offsets"]
                vals = [f["chars"][o[i]:o[i + 1]].tobytes() for i in range(v0, v1)]
            else:
                vals = f["values"][v0:v1]
            row[name] = (vals, f["defs"][l0:l1] if max_def else None, f["reps"][l0:l1] if f["reps"] is not None else None)
            self._row_pos[name] = r + 1
        return row
