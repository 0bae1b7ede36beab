"""ctypes binding of oracle/_build/libpqoracle.so (the CPU restatement of the reference).

Test infrastructure only: the product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB_PATH = os.path.join(_ROOT, "oracle", "_build", "libpqoracle.so")

OK, EOF, PANIC, THRIFT, UNSUPPORTED, NOMEM = 0, -1, -2, -3, -4, -5
ERR_NAMES = {0: "ok", -1: "eof", -2: "panic", -3: "thrift", -4: "unsupported", -5: "nomem"}

BOOLEAN, INT32, INT64, FLOAT, DOUBLE, BYTE_ARRAY = 0, 1, 2, 4, 5, 6


class _Vec(C.Structure):
    _fields_ = [("p", C.POINTER(C.c_uint8)), ("n", C.c_size_t), ("cap", C.c_size_t)]


class PageHeader(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "type", "uncompressed_page_size", "compressed_page_size", "has_data_page_header",
        "num_values", "encoding", "def_encoding", "rep_encoding", "header_len")]


class _Chunk(C.Structure):
    _fields_ = [
        ("phys_type", C.c_int32), ("max_def", C.c_int32), ("max_rep", C.c_int32),
        ("pg_n", C.c_int64), ("pg_size", C.c_int64),
        ("defs", C.POINTER(C.c_uint8)), ("reps", C.POINTER(C.c_uint8)),
        ("n_defs", C.c_int64), ("n_reps", C.c_int64),
        ("values", C.POINTER(C.c_uint8)), ("values_len", C.c_int64), ("n_values", C.c_int64),
        ("offsets", C.POINTER(C.c_int32)), ("sizes", C.POINTER(C.c_int64)),
        ("n_pages", C.c_int64), ("consumed", C.c_int64), ("codec", C.c_int32),
    ]


def build():
    subprocess.run(["make", "-s", "-C", os.path.join(_ROOT, "oracle")], check=True)


def _load():
    if not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(
            os.path.join(_ROOT, "oracle", "pq_oracle.c")):
        build()
    lib = C.CDLL(_LIB_PATH)
    lib.pqo_bitpack_unpack.argtypes = [C.c_int, C.c_char_p, C.POINTER(C.c_uint8)]
    lib.pqo_bitpack_pack.argtypes = [C.c_int, C.POINTER(C.c_uint8), C.POINTER(C.c_uint8)]
    lib.pqo_bitpack_unpack_spec.argtypes = [C.c_int, C.c_char_p, C.POINTER(C.c_uint32)]
    lib.pqo_bitpack_unpack_spec.restype = None
    lib.pqo_rle_read.argtypes = [C.c_int, C.c_char_p, C.c_size_t, C.POINTER(_Vec), C.POINTER(C.c_int64)]
    lib.pqo_rle_read_u32.argtypes = [C.c_int, C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                     C.POINTER(C.c_size_t), C.POINTER(C.c_int64)]
    lib.pqo_rle_encode.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p),
                                   C.POINTER(C.c_size_t)]
    lib.pqo_free.argtypes = [C.c_void_p]
    lib.pqo_free.restype = None
    lib.pqo_page_header.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(PageHeader)]
    lib.pqo_read_chunk.argtypes = [C.c_void_p, C.c_size_t, C.c_int64, C.POINTER(_Chunk)]
    lib.pqo_snappy_decode.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    lib.pqo_snappy_decode.restype = C.c_int64
    lib.pqo_snappy_decoded_len.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.pqo_snappy_decoded_len.restype = C.c_int64
    lib.pqo_chunk_free.argtypes = [C.POINTER(_Chunk)]
    lib.pqo_chunk_free.restype = None
    lib.pqo_row_offsets.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int,
                                    C.c_void_p, C.c_void_p, C.c_int64]
    lib.pqo_row_offsets.restype = C.c_int64
    lib.pqo_validity.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
    lib.pqo_validity.restype = None
    return lib


lib = _load()


class OracleError(Exception):
    def __init__(self, code):
        super().__init__("oracle: %s" % ERR_NAMES.get(code, code))
        self.code = code


def bitpack_unpack(width, data):
    out = (C.c_uint8 * 8)()
    n = lib.pqo_bitpack_unpack(width, bytes(data) + b"\0", out)
    return list(out)[:n]


def bitpack_pack(width, vals):
    v = (C.c_uint8 * 8)(*vals)
    out = (C.c_uint8 * 8)()
    n = lib.pqo_bitpack_pack(width, v, out)
    return bytes(out)[:n]


def bitpack_unpack_spec(width, data):
    out = (C.c_uint32 * 8)()
    lib.pqo_bitpack_unpack_spec(width, bytes(data) + b"\0" * 8, out)
    return list(out)


def rle_read(width, data):
    """rle.Read: returns (all decoded uint8 values, bytes consumed)."""
    v = _Vec()
    consumed = C.c_int64()
    data = bytes(data)
    rc = lib.pqo_rle_read(width, data, len(data), C.byref(v), C.byref(consumed))
    try:
        if rc:
            raise OracleError(rc)
        out = np.ctypeslib.as_array(v.p, shape=(v.n,)).copy() if v.n else np.zeros(0, np.uint8)
    finally:
        if v.p:
            lib.pqo_free(v.p)
    return out, consumed.value


def rle_read_u32(width, data, cap):
    out = np.zeros(cap, np.uint32)
    n = C.c_size_t()
    consumed = C.c_int64()
    data = bytes(data)
    rc = lib.pqo_rle_read_u32(width, data, len(data), out.ctypes.data, cap, C.byref(n), C.byref(consumed))
    if rc:
        raise OracleError(rc)
    return out[:n.value].copy(), consumed.value


def rle_encode(width, levels):
    levels = np.ascontiguousarray(levels, dtype=np.uint8)
    p = C.c_void_p()
    n = C.c_size_t()
    rc = lib.pqo_rle_encode(width, levels.ctypes.data, levels.size, C.byref(p), C.byref(n))
    if rc:
        raise OracleError(rc)
    try:
        return C.string_at(p, n.value)
    finally:
        lib.pqo_free(p)


def page_header(data):
    h = PageHeader()
    data = bytes(data)
    rc = lib.pqo_page_header(data, len(data), C.byref(h))
    if rc:
        raise OracleError(rc)
    return h


class Chunk:
    """Result of one Field.Read (one column chunk)."""
    __slots__ = ("defs", "reps", "values", "n_values", "offsets", "sizes", "consumed", "error",
                 "max_def", "max_rep", "phys_type")


def read_chunk(file_bytes, pos, phys_type, max_def, max_rep, pg_n, pg_size, codec=0):
    """Restatement of Field.Read for one column chunk starting at file offset `pos`."""
    buf = np.frombuffer(file_bytes, dtype=np.uint8)
    c = _Chunk()
    c.phys_type, c.max_def, c.max_rep, c.pg_n, c.pg_size = phys_type, max_def, max_rep, pg_n, pg_size
    c.codec = codec
    rc = lib.pqo_read_chunk(buf.ctypes.data, buf.size, pos, C.byref(c))
    out = Chunk()
    out.error = rc
    try:
        def arr(p, n, dt):
            if not p or n == 0:
                return np.zeros(0, dt)
            return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)),
                                         shape=(n * np.dtype(dt).itemsize,)).view(dt).copy()
        out.defs = arr(c.defs, c.n_defs, np.uint8)
        out.reps = arr(c.reps, c.n_reps, np.uint8)
        out.values = arr(c.values, c.values_len, np.uint8)
        out.n_values = c.n_values
        out.offsets = arr(c.offsets, c.n_values + 1, np.int32) if c.offsets else None
        out.sizes = arr(c.sizes, c.n_pages, np.int64)
        out.consumed = c.consumed
    finally:
        lib.pqo_chunk_free(C.byref(c))
    return out


def snappy_decode(data):
    """snappy.Decode (golang/snappy v0.0.2) of one raw-format block.  Returns (bytes, 0) or (None, error)."""
    src = np.frombuffer(bytes(data), np.uint8)
    hdr = C.c_size_t()
    n = lib.pqo_snappy_decoded_len(src.ctypes.data, src.size, C.byref(hdr))
    if n < 0:
        return None, int(n)
    dst = np.zeros(max(n, 1), np.uint8)
    got = lib.pqo_snappy_decode(src.ctypes.data, src.size, dst.ctypes.data, n)
    if got < 0:
        return None, int(got)
    return dst[:n].tobytes(), 0


def row_offsets(defs, reps, max_def, max_rep):
    defs = np.ascontiguousarray(defs, np.uint8)
    n = defs.size
    reps = np.ascontiguousarray(reps, np.uint8) if max_rep > 0 else np.zeros(max(n, 1), np.uint8)
    lo = np.zeros(n + 2, np.int64)
    vo = np.zeros(n + 2, np.int64)
    rows = lib.pqo_row_offsets(defs.ctypes.data, reps.ctypes.data, n, max_def, max_rep,
                               lo.ctypes.data, vo.ctypes.data, n + 2)
    assert rows >= 0
    return lo[:rows + 1].copy(), vo[:rows + 1].copy()


def validity(defs, max_def):
    defs = np.ascontiguousarray(defs, np.uint8)
    out = np.zeros((defs.size + 7) // 8, np.uint8)
    lib.pqo_validity(defs.ctypes.data, defs.size, max_def, out.ctypes.data)
    return out
