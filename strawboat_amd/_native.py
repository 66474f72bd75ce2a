"""ctypes binding of include/strawboat_hip.h (the C ABI of libstrawboat_hip.so).

The library is the product: if it is missing, or there is no GPU, every entry point fails
loudly — there is no CPU fallback in this package.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libstrawboat_hip.so")

SB_OK = 0
SB_ERR_OUT_OF_SPEC, SB_ERR_EXTERNAL, SB_ERR_IO, SB_ERR_NYI, SB_ERR_INVALID = -1, -2, -3, -4, -5
SB_MEM_DEVICE, SB_MEM_HOST = 0, 1
SB_WRITE_LZ4_EXACT = 1

# every symbol include/strawboat_hip.h declares
EXPORTS = ("sb_version", "sb_ctx_create", "sb_ctx_destroy", "sb_ctx_synchronize", "sb_ctx_last_error",
           "sb_ctx_stream", "sb_read_columns", "sb_read_columns_sizes", "sb_write_bound", "sb_write_columns",
           "sb_ctx_profile", "sb_ctx_profile_read", "sb_ctx_zstd_block_stats", "sb_ctx_side_forks", "sb_ctx_replays", "sb_nested_levels_bound", "sb_nested_write_levels", "sb_nested_write_levels_enqueue", "sb_nested_read_levels_enqueue",
           "sb_nested_read_levels", "sb_nested_write_levels_batch", "sb_nested_read_levels_batch", "sb_file_last_error", "sb_file_writer_open", "sb_file_writer_start",
           "sb_file_writer_write_column", "sb_file_writer_finish", "sb_file_writer_close", "sb_file_reader_open",
           "sb_file_reader_n_columns", "sb_file_reader_column", "sb_file_reader_schema", "sb_file_reader_read_pages",
           "sb_file_reader_close", "sb_stat_page", "sb_schema_last_error", "sb_schema_to_bytes", "sb_schema_from_bytes",
           "sb_schema_metadata_from_bytes")


class PageMetaC(C.Structure):
    _fields_ = [("length", C.c_uint64), ("num_values", C.c_uint64)]


class WriteOptionsC(C.Structure):
    _fields_ = [("default_compression", C.c_int32), ("has_default_compress_ratio", C.c_int32),
                ("default_compress_ratio", C.c_double), ("max_page_size", C.c_uint64),
                ("forbidden_compressions", C.c_uint32), ("force_codec", C.c_int32),
                ("force_index_codec", C.c_int32), ("flags", C.c_uint32), ("rng_seed", C.c_uint64)]


class ColumnReadC(C.Structure):
    _fields_ = [("physical_type", C.c_int32), ("is_nullable", C.c_int32), ("pages", C.c_void_p),
                ("pages_len", C.c_uint64), ("metas", C.POINTER(PageMetaC)), ("n_pages", C.c_uint64),
                ("values", C.c_void_p), ("values_capacity", C.c_uint64), ("validity", C.c_void_p),
                ("validity_capacity", C.c_uint64), ("offsets", C.c_void_p), ("offsets_capacity", C.c_uint64),
                ("rows", C.c_uint64), ("values_len", C.c_uint64), ("page_offsets", C.c_void_p)]


class ColumnWriteC(C.Structure):
    _fields_ = [("physical_type", C.c_int32), ("is_nullable", C.c_int32), ("rows", C.c_uint64),
                ("values", C.c_void_p), ("values_bit_offset", C.c_uint64), ("values_len", C.c_uint64),
                ("validity", C.c_void_p), ("validity_bit_offset", C.c_uint64), ("offsets", C.c_void_p),
                ("out_pages", C.c_void_p), ("out_capacity", C.c_uint64), ("out_metas", C.POINTER(PageMetaC)),
                ("n_pages_capacity", C.c_uint64), ("n_pages", C.c_uint64), ("out_len", C.c_uint64),
                ("page_rows", C.c_void_p), ("page_head_bytes", C.c_void_p), ("page_heads", C.c_void_p),
                ("n_pages_in", C.c_uint64), ("first_page_index", C.c_uint64), ("column_values_len", C.c_uint64)]


class NestedLevelC(C.Structure):
    _fields_ = [("validity", C.c_void_p), ("offsets", C.c_void_p), ("validity_bit_offset", C.c_uint64),
                ("length", C.c_uint64), ("kind", C.c_int32), ("is_optional", C.c_int32)]


class NestedPageC(C.Structure):
    _fields_ = [("level_bytes", C.c_uint64), ("num_values", C.c_uint64), ("leaf_start", C.c_uint64),
                ("leaf_count", C.c_uint64)]


class NestedLevelOutC(C.Structure):
    _fields_ = [("offsets", C.c_void_p), ("validity", C.c_void_p), ("offsets_capacity", C.c_uint64),
                ("validity_capacity", C.c_uint64), ("length", C.c_uint64), ("kind", C.c_int32),
                ("is_nullable", C.c_int32)]


class NestedLevelsWriteC(C.Structure):
    _fields_ = [("levels", C.POINTER(NestedLevelC)), ("n_levels", C.c_uint32), ("reserved", C.c_uint32), ("rows", C.c_uint64),
                ("out_levels", C.c_void_p), ("out_capacity", C.c_uint64), ("pages", C.POINTER(NestedPageC)),
                ("n_pages_capacity", C.c_uint64), ("n_pages", C.c_uint64)]


class NestedLevelsReadC(C.Structure):
    _fields_ = [("pages", C.c_void_p), ("pages_len", C.c_uint64), ("metas", C.c_void_p), ("n_pages", C.c_uint64),
                ("levels", C.POINTER(NestedLevelOutC)), ("n_levels", C.c_uint32), ("reserved", C.c_uint32),
                ("leaf_validity", C.c_void_p), ("leaf_validity_capacity", C.c_uint64),
                ("page_leaf_counts", C.c_void_p), ("page_block_offsets", C.c_void_p)]


class PageInfoC(C.Structure):
    _fields_ = [("codec", C.c_int32), ("has_validity_size", C.c_int32), ("validity_size", C.c_uint32),
                ("compressed_size", C.c_uint32), ("uncompressed_size", C.c_uint32), ("unique_num", C.c_uint32),
                ("exceptions_bitmap_size", C.c_uint32), ("has_nested", C.c_int32)]


class SchemaFieldC(C.Structure):
    _fields_ = [("name", C.c_char_p), ("timezone", C.c_char_p), ("type_id", C.c_int32), ("nullable", C.c_int32),
                ("n_children", C.c_int32), ("bit_width", C.c_int32), ("is_signed", C.c_int32), ("precision", C.c_int32),
                ("scale", C.c_int32), ("unit", C.c_int32), ("metadata", C.c_void_p), ("n_metadata", C.c_uint64)]


class KernelStatC(C.Structure):
    _fields_ = [("name", C.c_char_p), ("launches", C.c_uint64), ("total_ms", C.c_double)]


class NativeError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("strawboat-hip error %d: %s" % (code, message))
        self.code = code


_lib = None


def load():
    """Load libstrawboat_hip.so.  torch must be imported first so that one HIP runtime
    (SONAME libamdhip64.so.7) is shared by torch and this library."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(or `make -C strawboat_amd/csrc`)" % LIB_PATH)
    import torch  # noqa: F401  (loads the HIP runtime)
    L = C.CDLL(LIB_PATH)
    L.sb_version.restype = C.c_char_p
    L.sb_ctx_create.restype = C.c_int32
    L.sb_ctx_create.argtypes = [C.c_int32, C.c_void_p, C.POINTER(C.c_void_p)]
    L.sb_ctx_destroy.argtypes = [C.c_void_p]
    L.sb_ctx_synchronize.restype = C.c_int32
    L.sb_ctx_synchronize.argtypes = [C.c_void_p]
    L.sb_ctx_last_error.restype = C.c_char_p
    L.sb_ctx_last_error.argtypes = [C.c_void_p]
    L.sb_ctx_stream.restype = C.c_void_p
    L.sb_ctx_stream.argtypes = [C.c_void_p]
    L.sb_read_columns.restype = C.c_int32
    L.sb_read_columns.argtypes = [C.c_void_p, C.POINTER(ColumnReadC), C.c_uint64, C.c_int32]
    L.sb_read_columns_sizes.restype = C.c_int32
    L.sb_read_columns_sizes.argtypes = [C.c_void_p, C.POINTER(ColumnReadC), C.c_uint64, C.c_int32]
    L.sb_write_bound.restype = C.c_uint64
    L.sb_write_bound.argtypes = [C.c_int32, C.c_int32, C.c_uint64, C.c_uint64, C.POINTER(WriteOptionsC),
                                 C.POINTER(C.c_uint64)]
    L.sb_write_columns.restype = C.c_int32
    L.sb_write_columns.argtypes = [C.c_void_p, C.POINTER(ColumnWriteC), C.c_uint64, C.POINTER(WriteOptionsC),
                                   C.c_int32]
    L.sb_ctx_profile.restype = C.c_int32
    L.sb_ctx_profile.argtypes = [C.c_void_p, C.c_int32]
    L.sb_ctx_side_forks.restype = C.c_uint64
    L.sb_ctx_side_forks.argtypes = [C.c_void_p]
    L.sb_ctx_replays.restype = C.c_uint64
    L.sb_ctx_replays.argtypes = [C.c_void_p]
    L.sb_ctx_zstd_block_stats.restype = C.c_int32
    L.sb_ctx_zstd_block_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    L.sb_ctx_profile_read.restype = C.c_uint32
    L.sb_ctx_profile_read.argtypes = [C.c_void_p, C.POINTER(KernelStatC), C.c_uint32]
    L.sb_nested_levels_bound.restype = C.c_uint64
    L.sb_nested_levels_bound.argtypes = [C.POINTER(NestedLevelC), C.c_uint32, C.c_uint64, C.c_uint64]
    L.sb_nested_write_levels.restype = C.c_int32
    L.sb_nested_write_levels.argtypes = [C.c_void_p, C.POINTER(NestedLevelC), C.c_uint32, C.c_uint64, C.c_uint64,
                                         C.c_void_p, C.c_uint64, C.POINTER(NestedPageC), C.c_uint64,
                                         C.POINTER(C.c_uint64)]
    L.sb_nested_read_levels.restype = C.c_int32
    L.sb_nested_read_levels.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(PageMetaC), C.c_uint64,
                                        C.POINTER(NestedLevelOutC), C.c_uint32, C.c_void_p, C.c_uint64,
                                        C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.sb_nested_write_levels_batch.restype = C.c_int32
    L.sb_nested_write_levels_batch.argtypes = [C.c_void_p, C.POINTER(NestedLevelsWriteC), C.c_uint64, C.c_uint64]
    L.sb_nested_read_levels_batch.restype = C.c_int32
    L.sb_nested_read_levels_batch.argtypes = [C.c_void_p, C.POINTER(NestedLevelsReadC), C.c_uint64]
    L.sb_nested_write_levels_enqueue.restype = C.c_int32
    L.sb_nested_write_levels_enqueue.argtypes = [C.c_void_p, C.POINTER(NestedLevelsWriteC), C.c_uint64, C.c_uint64]
    L.sb_nested_read_levels_enqueue.restype = C.c_int32
    L.sb_nested_read_levels_enqueue.argtypes = [C.c_void_p, C.POINTER(NestedLevelsReadC), C.c_uint64]
    L.sb_file_last_error.restype = C.c_char_p
    L.sb_file_writer_open.restype = C.c_int32
    L.sb_file_writer_open.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
    L.sb_file_writer_start.restype = C.c_int32
    L.sb_file_writer_start.argtypes = [C.c_void_p]
    L.sb_file_writer_write_column.restype = C.c_int32
    L.sb_file_writer_write_column.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(PageMetaC), C.c_uint64]
    L.sb_file_writer_finish.restype = C.c_int32
    L.sb_file_writer_finish.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    L.sb_file_writer_close.argtypes = [C.c_void_p]
    L.sb_file_reader_open.restype = C.c_int32
    L.sb_file_reader_open.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
    L.sb_file_reader_n_columns.restype = C.c_uint64
    L.sb_file_reader_n_columns.argtypes = [C.c_void_p]
    L.sb_file_reader_column.restype = C.c_int32
    L.sb_file_reader_column.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                        C.POINTER(C.POINTER(PageMetaC))]
    L.sb_file_reader_schema.restype = C.c_int32
    L.sb_file_reader_schema.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    L.sb_file_reader_read_pages.restype = C.c_int32
    L.sb_file_reader_read_pages.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint64,
                                            C.POINTER(C.c_uint64)]
    L.sb_file_reader_close.argtypes = [C.c_void_p]
    L.sb_stat_page.restype = C.c_int32
    L.sb_stat_page.argtypes = [C.c_void_p, C.c_uint64, C.c_int32, C.c_int32, C.POINTER(PageInfoC), C.c_uint32,
                               C.POINTER(C.c_uint32)]
    L.sb_schema_last_error.restype = C.c_char_p
    L.sb_schema_to_bytes.restype = C.c_int32
    L.sb_schema_to_bytes.argtypes = [C.POINTER(SchemaFieldC), C.c_uint64, C.c_uint64, C.POINTER(C.c_char_p), C.c_uint64,
                                     C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    L.sb_schema_from_bytes.restype = C.c_int32
    L.sb_schema_from_bytes.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(SchemaFieldC), C.c_uint64, C.POINTER(C.c_uint64),
                                       C.POINTER(C.c_uint64), C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    L.sb_schema_metadata_from_bytes.restype = C.c_int32
    L.sb_schema_metadata_from_bytes.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64),
                                                C.POINTER(C.c_uint64)]
    _lib = L
    return L
