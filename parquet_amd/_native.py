"""ctypes declarations for libpqdecode.so (include/pqdecode.h) and libpqsynth.so.

The libraries are built in-tree by parquet_amd/build.py.  There is no fallback: if the decoder
library is missing, importing this module raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "lib")
DECODE_PATH = os.environ.get("PQD_LIB") or os.path.join(LIB_DIR, "libpqdecode.so")  # PQD_LIB: another build of the library (kernel experiments)
SYNTH_PATH = os.path.join(LIB_DIR, "libpqsynth.so")

# pqd_status
OK, E_ARG, E_HIP, E_NOMEM, E_FORMAT, E_UNSUPPORTED, E_CORRUPT, E_NONCANONICAL = 0, -1, -2, -3, -4, -5, -6, -7
STATUS_NAMES = {0: "OK", -1: "E_ARG", -2: "E_HIP", -3: "E_NOMEM", -4: "E_FORMAT", -5: "E_UNSUPPORTED",
                -6: "E_CORRUPT", -7: "E_NONCANONICAL"}

BOOLEAN, INT32, INT64, INT96, FLOAT, DOUBLE, BYTE_ARRAY, FIXED_LEN_BYTE_ARRAY = range(8)

(OUT_DEFS, OUT_REPS, OUT_VALUES, OUT_OFFSETS, OUT_CHARS, OUT_VALIDITY, OUT_PAGE_NONNULL,
 OUT_ROW_LEVEL_OFF, OUT_ROW_VALUE_OFF, OUT_VALUE_INDEX) = range(10)


class ChunkDesc(C.Structure):
    _fields_ = [("phys_type", C.c_int32), ("max_def", C.c_int32), ("max_rep", C.c_int32),
                ("first_page", C.c_int32), ("n_pages", C.c_int32), ("column", C.c_int32),
                ("num_values", C.c_int64), ("codec", C.c_int32), ("reserved", C.c_int32)]


class SnappyBlock(C.Structure):
    _fields_ = [("src_off", C.c_int64), ("comp_len", C.c_int64), ("dst_off", C.c_int64), ("decoded_len", C.c_int64),
                ("hdr_len", C.c_int32), ("reserved", C.c_int32)]


class EncPage(C.Structure):
    _fields_ = [("phys_type", C.c_int32), ("is_unsigned", C.c_int32), ("max_def", C.c_int32), ("max_rep", C.c_int32),
                ("n_levels", C.c_int64), ("n_values", C.c_int64), ("defs_off", C.c_int64), ("reps_off", C.c_int64),
                ("values_off", C.c_int64), ("offsets_off", C.c_int64)]


class EncResult(C.Structure):
    _fields_ = [("payload_off", C.c_int64), ("payload_len", C.c_int64), ("rep_bytes", C.c_int32), ("def_bytes", C.c_int32),
                ("null_count", C.c_int64), ("min", C.c_uint8 * 8), ("max", C.c_uint8 * 8), ("minmax_len", C.c_int32),
                ("reserved", C.c_int32)]


class PageDesc(C.Structure):
    _fields_ = [("src_off", C.c_int64), ("payload_len", C.c_int32), ("num_values", C.c_int32),
                ("rep_bytes", C.c_int32), ("def_bytes", C.c_int32)]


class ChunkInfo(C.Structure):
    _fields_ = [("n_levels", C.c_int64), ("n_values", C.c_int64), ("value_bytes", C.c_int64),
                ("chars_bytes", C.c_int64), ("n_rows", C.c_int64), ("n_pages", C.c_int64)]


class ColumnInfo(C.Structure):
    _fields_ = [("phys_type", C.c_int32), ("converted_type", C.c_int32), ("max_def", C.c_int32),
                ("max_rep", C.c_int32), ("leaf_repetition", C.c_int32), ("reserved", C.c_int32)]


# Every symbol include/pqdecode.h declares: name → (restype, argtypes)
_P = C.c_void_p
DECODE_SYMBOLS = {
    "pqd_create": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "pqd_destroy": (None, [_P]),
    "pqd_last_error": (C.c_char_p, [_P]),
    "pqd_host_alloc": (_P, [_P, C.c_size_t]),
    "pqd_host_free": (None, [_P, _P]),
    "pqd_device_alloc": (_P, [_P, C.c_size_t]),
    "pqd_device_free": (None, [_P, _P]),
    "pqd_copy_to_device": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "pqd_copy_to_host": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "pqd_plan_create": (C.c_int, [_P, C.POINTER(ChunkDesc), C.c_size_t, C.POINTER(PageDesc), C.c_size_t,
                                  _P, C.c_size_t, C.POINTER(_P)]),
    "pqd_plan_destroy": (None, [_P]),
    "pqd_plan_run": (C.c_int, [_P, _P]),
    "pqd_plan_sync": (C.c_int, [_P]),
    "pqd_plan_last_ms": (C.c_int, [_P, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "pqd_plan_run_ms": (C.c_int, [_P, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "pqd_plan_bytes": (C.c_int, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "pqd_plan_chunk_info": (C.c_int, [_P, C.c_size_t, C.POINTER(ChunkInfo)]),
    "pqd_plan_device_ptr": (_P, [_P, C.c_size_t, C.c_int]),
    "pqd_plan_fetch": (C.c_int64, [_P, C.c_size_t, C.c_int, _P, C.c_size_t]),
    "pqd_plan_scatter": (C.c_int, [_P, C.c_size_t, _P, C.c_size_t]),
    "pqd_plan_string_fallbacks": (C.c_int64, [_P]),
    "pqd_plan_h2d_bytes": (C.c_int64, [_P]),
    "pqd_plan_shared_reps": (C.c_int64, [_P]),
    "pqd_plan_debug_stamps": (C.c_int64, [_P, _P, C.c_size_t]),
    "pqd_decode": (C.c_int, [_P, C.POINTER(ChunkDesc), C.c_size_t, C.POINTER(PageDesc), C.c_size_t,
                             _P, C.c_size_t, C.POINTER(_P)]),
    "pqd_rle_decode": (C.c_int, [_P, C.c_int, _P, C.c_size_t, C.c_int, _P, C.c_size_t, C.POINTER(C.c_int64)]),
    "pqd_rle_decode_noalloc": (C.c_int, [_P, C.c_int, _P, C.c_size_t, _P, C.c_size_t, _P]),
    "pqd_rle_encode": (C.c_int, [_P, C.c_int, _P, C.c_size_t, _P, C.c_size_t, C.POINTER(C.c_int64)]),
    "pqd_encode_pages": (C.c_int, [_P, C.POINTER(EncPage), C.c_size_t, _P, C.c_size_t, _P, C.c_size_t, C.POINTER(EncResult), C.POINTER(C.c_int64)]),
    "pqd_snappy_sizes": (C.c_int, [_P, C.c_size_t, C.POINTER(SnappyBlock), C.c_size_t]),
    "pqd_snappy_decode": (C.c_int, [_P, C.POINTER(SnappyBlock), C.c_size_t, _P, C.c_size_t, _P, C.c_size_t]),
    "pqd_level_block_sizes": (C.c_int, [_P, C.POINTER(ChunkDesc), C.c_size_t, C.POINTER(PageDesc), C.c_size_t, _P, C.c_size_t]),
    "pqd_file_open": (C.c_int, [_P, C.c_size_t, C.POINTER(_P), C.c_char_p, C.c_size_t]),
    "pqd_file_close": (None, [_P]),
    "pqd_file_num_rows": (C.c_int64, [_P]),
    "pqd_file_num_columns": (C.c_int32, [_P]),
    "pqd_file_num_row_groups": (C.c_int32, [_P]),
    "pqd_file_row_group_rows": (C.c_int64, [_P, C.c_int32]),
    "pqd_file_column": (C.c_int, [_P, C.c_int32, C.POINTER(ColumnInfo)]),
    "pqd_file_column_path": (C.c_char_p, [_P, C.c_int32]),
    "pqd_file_descriptors": (C.c_int, [_P, C.c_int32, C.c_int32, C.POINTER(C.POINTER(ChunkDesc)),
                                       C.POINTER(C.c_size_t), C.POINTER(C.POINTER(PageDesc)), C.POINTER(C.c_size_t)]),
    "pqd_file_num_pages": (C.c_int64, [_P]),
    "pqd_file_page_header": (C.c_int, [_P, C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    "pqd_file_chunk_codec": (C.c_int, [_P, C.c_int32, C.c_int32]),
    "pqd_file_chunk_meta": (C.c_int, [_P, C.c_int32, C.c_int32, C.POINTER(C.c_int64)]),
    "pqd_shard_assign": (C.c_int, [C.POINTER(C.c_int64), C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "pqd_file_shard": (C.c_int, [_P, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "pqd_multi_create": (C.c_int, [C.POINTER(C.c_int), C.c_int, C.POINTER(_P)]),
    "pqd_multi_destroy": (None, [_P]),
    "pqd_multi_decode_file": (C.c_int, [_P, _P, C.c_size_t]),
    "pqd_multi_shards": (C.c_int, [_P]),
    "pqd_multi_shard": (C.c_int, [_P, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(_P)]),
    "pqd_multi_last_error": (C.c_char_p, [_P]),
    "pqd_rccl_unique_id": (C.c_int, [_P]),
    "pqd_rccl_comm_create": (C.c_int, [C.c_int, C.c_int, C.c_int, _P, C.POINTER(_P)]),
    "pqd_rccl_comm_destroy": (None, [_P]),
    "pqd_rccl_gather_rows": (C.c_int, [_P, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
}

SYNTH_SYMBOLS = {
    "pqw_set_codec": (C.c_int, [_P, C.c_int]),
    "pqw_snappy_compress": (C.c_int64, [_P, C.c_int64, _P, C.c_int64]),
    "pqw_create": (_P, [C.c_int]),
    "pqw_destroy": (None, [_P]),
    "pqw_add_column": (C.c_int, [_P, C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_int32), C.c_int]),
    "pqw_begin_row_group": (C.c_int, [_P]),
    "pqw_write_page": (C.c_int, [_P, C.c_int, _P, _P, C.c_int64, _P, C.c_int64, _P]),
    "pqw_write_raw_page": (C.c_int, [_P, C.c_int, _P, C.c_int64, C.c_int64]),
    "pqw_end_row_group": (C.c_int, [_P, C.c_int64]),
    "pqw_finish": (C.c_int, [_P]),
    "pqw_data": (_P, [_P, C.POINTER(C.c_int64)]),
    "pqw_reserve": (C.c_int, [_P, C.c_int64]),
    "pqw_encode_levels": (C.c_int64, [C.c_int, _P, C.c_int64, _P, C.c_int64]),
}


def _bind(path, symbols):
    if not os.path.exists(path):
        raise ImportError("%s is missing: run `python parquet_amd/build.py` (needs hipcc)" % path)
    lib = C.CDLL(path)
    for name, (res, args) in symbols.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    return lib


decode = _bind(DECODE_PATH, DECODE_SYMBOLS)
synth = _bind(SYNTH_PATH, SYNTH_SYMBOLS)


class PqdError(RuntimeError):
    def __init__(self, code, msg=""):
        super().__init__("%s%s" % (STATUS_NAMES.get(code, code), (": " + msg) if msg else ""))
        self.code = code
