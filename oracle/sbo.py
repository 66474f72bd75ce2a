"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes wrapper around oracle/libsbo_oracle.so (the CPU restatement of strawboat's page
codecs, see oracle/sbo.h).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; the product package never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsbo_oracle.so")

# codec ids (src/compression/mod.rs:92-108)
NONE, LZ4, ZSTD, SNAPPY = 0, 1, 2, 3
RLE, DICT, ONEVALUE, FREQ, BITPACK, DELTABP, PATAS = 10, 11, 12, 13, 14, 15, 16

# physical types (oracle/sbo.h PhysType)
T_BOOL, T_I8, T_I16, T_I32, T_I64, T_U8, T_U16, T_U32, T_U64 = range(9)
T_I128, T_I256, T_F32, T_F64, T_BIN32, T_BIN64, T_NULL = range(9, 16)

WIDTH = {T_I8: 1, T_U8: 1, T_I16: 2, T_U16: 2, T_I32: 4, T_U32: 4, T_F32: 4, T_I64: 8, T_U64: 8,
         T_F64: 8, T_I128: 16, T_I256: 32}


def build(force=False):
    if force or not os.path.exists(_LIB_PATH) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
            for f in os.listdir(_HERE) if f.endswith((".cpp", ".h"))):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


class _ColumnIn(C.Structure):
    _fields_ = [("ptype", C.c_int32), ("nullable", C.c_int32), ("rows", C.c_uint64),
                ("values", C.c_void_p), ("values_bit_offset", C.c_uint64), ("values_len", C.c_uint64),
                ("validity", C.c_void_p), ("validity_bit_offset", C.c_uint64), ("offsets", C.c_void_p)]


class _Options(C.Structure):
    _fields_ = [("default_compression", C.c_uint8), ("has_ratio", C.c_uint8), ("pad_", C.c_uint8 * 6),
                ("ratio", C.c_double), ("max_page_size", C.c_uint64), ("forbidden_mask", C.c_uint32),
                ("force_codec", C.c_int32), ("force_index_codec", C.c_int32), ("pad2_", C.c_int32),
                ("rng_seed", C.c_uint64), ("page_index0", C.c_uint64)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.sbo_write_column.restype = C.c_void_p
        L.sbo_write_column.argtypes = [C.POINTER(_ColumnIn), C.POINTER(_Options), C.c_char_p, C.c_size_t]
        L.sbo_write_page.restype = C.c_void_p
        L.sbo_write_page.argtypes = [C.POINTER(_ColumnIn), C.POINTER(_Options), C.c_char_p, C.c_size_t]
        for name in ("sbo_written_len", "sbo_written_npages", "sbo_read_rows", "sbo_read_values_len",
                     "sbo_read_validity_len", "sbo_read_offsets_len"):
            getattr(L, name).restype = C.c_uint64
            getattr(L, name).argtypes = [C.c_void_p]
        for name in ("sbo_written_data", "sbo_written_metas", "sbo_read_values", "sbo_read_validity",
                     "sbo_read_offsets"):
            getattr(L, name).restype = C.c_void_p
            getattr(L, name).argtypes = [C.c_void_p]
        L.sbo_written_free.argtypes = [C.c_void_p]
        L.sbo_read_free.argtypes = [C.c_void_p]
        L.sbo_read_column.restype = C.c_void_p
        L.sbo_read_column.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                                      C.c_char_p, C.c_size_t]
        L.sbo_stat_column.restype = C.c_int32
        L.sbo_stat_column.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                                      C.c_void_p, C.c_void_p, C.c_char_p, C.c_size_t]
        L.sbo_block_compress.restype = C.c_int64
        L.sbo_block_compress.argtypes = [C.c_int32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
        L.sbo_block_bound.restype = C.c_uint64
        L.sbo_block_bound.argtypes = [C.c_int32, C.c_uint64]
        L.sbo_block_decompress.restype = C.c_int32
        L.sbo_block_decompress.argtypes = [C.c_int32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                                           C.c_char_p, C.c_size_t]
        L.sbo_patas_pack.restype = C.c_uint32
        L.sbo_patas_pack.argtypes = [C.c_uint32] * 3
        L.sbo_patas_unpack.argtypes = [C.c_uint32, C.c_void_p]
        L.sbo_bitpack_num_bits.restype = C.c_uint8
        L.sbo_bitpack_num_bits.argtypes = [C.c_void_p]
        L.sbo_bitpack_pack.argtypes = [C.c_void_p, C.c_uint8, C.c_void_p, C.c_int32, C.c_uint32]
        L.sbo_bitpack_unpack.argtypes = [C.c_void_p, C.c_uint8, C.c_void_p, C.c_int32, C.c_uint32]
        L.sbo_sample_rand.restype = C.c_uint64
        L.sbo_sample_rand.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64]
        L.sbo_mix64.restype = C.c_uint64
        L.sbo_mix64.argtypes = [C.c_uint64]
        L.sbo_time_roundtrip.restype = C.c_int32
        L.sbo_time_roundtrip.argtypes = [C.POINTER(_ColumnIn), C.POINTER(_Options), C.c_int32, C.c_void_p,
                                         C.c_char_p, C.c_size_t]
        L.sbo_time_pages_mt.restype = C.c_int32
        L.sbo_time_pages_mt.argtypes = [C.POINTER(_ColumnIn), C.c_uint64, C.POINTER(_Options), C.c_int32, C.c_int32,
                                        C.c_void_p, C.POINTER(C.c_uint64), C.c_char_p, C.c_size_t]
        L.sbo_system_codecs.restype = C.c_int32
        L.sbo_system_codecs.argtypes = [C.c_int32]
        L.sbo_system_codec_version.restype = C.c_int32
        L.sbo_system_codec_version.argtypes = [C.c_int32]
        _lib = L
    return _lib


def system_codecs(on):
    """Basic(LZ4 / Zstd) blocks through the box's liblz4.so.1 / libzstd.so.1 (what the reference's crates wrap) instead of
    the restatement's own codecs — for bench.py's cpu_baseline leg.  Returns {"lz4": version or None, "zstd": ...}."""
    L = lib()
    m = L.sbo_system_codecs(1 if on else 0)

    def ver(v):
        return "%d.%d.%d" % (v // 10000, v // 100 % 100, v % 100)
    return {"lz4": ver(L.sbo_system_codec_version(0)) if m & 1 else None, "zstd": ver(L.sbo_system_codec_version(1)) if m & 2 else None}


class OracleError(RuntimeError):
    pass


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _bytes_view(a):
    if a is None:
        return None
    a = np.ascontiguousarray(a)
    return a.view(np.uint8).reshape(-1)


def make_options(default_compression=NONE, ratio=None, max_page_size=None, forbidden=(), force_codec=-1,
                 force_index_codec=-1, rng_seed=42, page_index0=0):
    o = _Options()
    o.default_compression = default_compression
    o.has_ratio = 0 if ratio is None else 1
    o.ratio = 0.0 if ratio is None else float(ratio)
    o.max_page_size = 0 if max_page_size is None else int(max_page_size)
    mask = 0
    for c in forbidden:
        mask |= 1 << c
    o.forbidden_mask = mask
    o.force_codec = force_codec
    o.force_index_codec = force_index_codec
    o.rng_seed = rng_seed
    o.page_index0 = page_index0
    return o


def _column_in(ptype, nullable, rows, values, validity, offsets, values_bit_offset, validity_bit_offset, keep):
    c = _ColumnIn()
    c.ptype, c.nullable, c.rows = ptype, int(bool(nullable)), rows
    v = _bytes_view(values)
    vb = _bytes_view(validity)
    ob = _bytes_view(offsets)
    keep.extend([v, vb, ob])
    c.values = _ptr(v)
    c.values_bit_offset = values_bit_offset
    c.values_len = 0 if v is None else v.size
    c.validity = _ptr(vb)
    c.validity_bit_offset = validity_bit_offset
    c.offsets = _ptr(ob)
    return c


def write_column(ptype, nullable, rows, values=None, validity=None, offsets=None, options=None,
                 values_bit_offset=0, validity_bit_offset=0):
    """NativeWriter::encode_chunk for one flat leaf column -> (page bytes u8[], metas u64[n,2])."""
    L = lib()
    keep = []
    c = _column_in(ptype, nullable, rows, values, validity, offsets, values_bit_offset, validity_bit_offset, keep)
    o = options if options is not None else make_options()
    err = C.create_string_buffer(512)
    h = L.sbo_write_column(C.byref(c), C.byref(o), err, 512)
    if not h:
        raise OracleError(err.value.decode())
    try:
        n = L.sbo_written_len(h)
        npg = L.sbo_written_npages(h)
        data = np.ctypeslib.as_array(C.cast(L.sbo_written_data(h), C.POINTER(C.c_uint8)), (n,)).copy() \
            if n else np.zeros(0, np.uint8)
        metas = np.ctypeslib.as_array(C.cast(L.sbo_written_metas(h), C.POINTER(C.c_uint64)), (npg, 2)).copy()
    finally:
        L.sbo_written_free(h)
    return data, metas


def write_page(ptype, nullable, rows, values=None, validity=None, offsets=None, options=None,
               values_bit_offset=0, validity_bit_offset=0):
    """write::write_simple for ONE page (rows may be 0: the leaf block of a nested page without leaf slots)."""
    L = lib()
    keep = []
    if rows == 0:
        values = np.zeros(8, np.uint8) if values is None or np.asarray(values).size == 0 else values
        if ptype in (T_BIN32, T_BIN64) and (offsets is None or np.asarray(offsets).size == 0):
            offsets = np.zeros(1, np.int32 if ptype == T_BIN32 else np.int64)
    c = _column_in(ptype, nullable, rows, values, validity, offsets, values_bit_offset, validity_bit_offset, keep)
    if rows == 0 and ptype in (T_BIN32, T_BIN64):
        c.values_len = 0
    o = options if options is not None else make_options()
    err = C.create_string_buffer(512)
    h = L.sbo_write_page(C.byref(c), C.byref(o), err, 512)
    if not h:
        raise OracleError(err.value.decode())
    try:
        n = L.sbo_written_len(h)
        data = np.ctypeslib.as_array(C.cast(L.sbo_written_data(h), C.POINTER(C.c_uint8)), (n,)).copy() \
            if n else np.zeros(0, np.uint8)
    finally:
        L.sbo_written_free(h)
    return data


def read_column(ptype, nullable, pages, metas):
    """batch_read::read_simple for one leaf column -> dict(rows, values, validity, offsets) of u8 arrays."""
    L = lib()
    pages = np.ascontiguousarray(pages, dtype=np.uint8)
    metas = np.ascontiguousarray(metas, dtype=np.uint64).reshape(-1, 2)
    err = C.create_string_buffer(512)
    h = L.sbo_read_column(ptype, int(bool(nullable)), _ptr(pages), pages.size, _ptr(metas), metas.shape[0], err, 512)
    if not h:
        raise OracleError(err.value.decode())
    try:
        def grab(fn_len, fn_ptr):
            n = fn_len(h)
            if n == 0:
                return np.zeros(0, np.uint8)
            return np.ctypeslib.as_array(C.cast(fn_ptr(h), C.POINTER(C.c_uint8)), (n,)).copy()
        out = dict(rows=int(L.sbo_read_rows(h)),
                   values=grab(L.sbo_read_values_len, L.sbo_read_values),
                   validity=grab(L.sbo_read_validity_len, L.sbo_read_validity),
                   offsets=grab(L.sbo_read_offsets_len, L.sbo_read_offsets))
    finally:
        L.sbo_read_free(h)
    return out


def stat_column(ptype, nullable, pages, metas):
    L = lib()
    pages = np.ascontiguousarray(pages, dtype=np.uint8)
    metas = np.ascontiguousarray(metas, dtype=np.uint64).reshape(-1, 2)
    n = metas.shape[0]
    codecs = np.zeros(n, np.uint8)
    inner = np.zeros(n, np.uint8)
    err = C.create_string_buffer(512)
    if L.sbo_stat_column(ptype, int(bool(nullable)), _ptr(pages), pages.size, _ptr(metas), n, _ptr(codecs),
                         _ptr(inner), err, 512) != 0:
        raise OracleError(err.value.decode())
    return codecs, inner


def block_compress(codec, data):
    L = lib()
    data = np.ascontiguousarray(data, dtype=np.uint8)
    cap = L.sbo_block_bound(codec, data.size)
    dst = np.zeros(cap, np.uint8)
    n = L.sbo_block_compress(codec, _ptr(data), data.size, _ptr(dst), cap)
    if n < 0:
        raise OracleError("block_compress failed")
    return dst[:n].copy()


def block_decompress(codec, data, out_len):
    L = lib()
    data = np.ascontiguousarray(data, dtype=np.uint8)
    dst = np.zeros(out_len, np.uint8)
    err = C.create_string_buffer(512)
    if L.sbo_block_decompress(codec, _ptr(data), data.size, _ptr(dst), out_len, err, 512) != 0:
        raise OracleError(err.value.decode())
    return dst


def time_roundtrip(ptype, nullable, rows, values=None, validity=None, offsets=None, options=None, iters=3):
    """(seconds to write, seconds to read) for one column on ONE host core; best of `iters`."""
    L = lib()
    keep = []
    c = _column_in(ptype, nullable, rows, values, validity, offsets, 0, 0, keep)
    o = options if options is not None else make_options()
    out2 = np.zeros(2, np.float64)
    err = C.create_string_buffer(512)
    if L.sbo_time_roundtrip(C.byref(c), C.byref(o), iters, _ptr(out2), err, 512) != 0:
        raise OracleError(err.value.decode())
    return float(out2[0]), float(out2[1])


def time_pages_mt(columns, options=None, threads=1, iters=2):
    """CPU baseline over the (column, page) items of `columns` (dicts: ptype, nullable, rows, values, validity,
    offsets) with `threads` std::threads: (seconds to encode, seconds to decode, page bytes)."""
    L = lib()
    keep = []
    arr = (_ColumnIn * len(columns))()
    for i, col in enumerate(columns):
        arr[i] = _column_in(col["ptype"], col["nullable"], col["rows"], col["values"], col.get("validity"),
                            col.get("offsets"), 0, 0, keep)
    o = options if options is not None else make_options()
    out2 = np.zeros(2, np.float64)
    nbytes = C.c_uint64(0)
    err = C.create_string_buffer(512)
    if L.sbo_time_pages_mt(arr, len(columns), C.byref(o), int(threads), int(iters), _ptr(out2), C.byref(nbytes), err, 512) != 0:
        raise OracleError(err.value.decode())
    return float(out2[0]), float(out2[1]), int(nbytes.value)


# ---------------------------------------------------------------- nested (Dremel) level sections
K_PRIMITIVE, K_LIST, K_LARGE_LIST, K_STRUCT = 0, 1, 2, 3


class _NestedLevel(C.Structure):
    _fields_ = [("kind", C.c_int32), ("is_optional", C.c_int32), ("validity", C.c_void_p),
                ("validity_off", C.c_uint64), ("offsets", C.c_void_p), ("length", C.c_uint64)]


def _nested_lib():
    L = lib()
    if not getattr(L, "_nested_ready", False):
        L.sbo_nested_write.restype = C.c_void_p
        L.sbo_nested_write.argtypes = [C.POINTER(_NestedLevel), C.c_int32, C.c_uint64, C.c_uint64, C.c_char_p, C.c_size_t]
        L.sbo_nested_written_len.restype = C.c_uint64
        L.sbo_nested_written_len.argtypes = [C.c_void_p]
        L.sbo_nested_written_data.restype = C.c_void_p
        L.sbo_nested_written_data.argtypes = [C.c_void_p]
        L.sbo_nested_written_info.argtypes = [C.c_void_p, C.c_void_p]
        L.sbo_nested_written_free.argtypes = [C.c_void_p]
        L.sbo_nested_read.restype = C.c_void_p
        L.sbo_nested_read.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_int32, C.c_char_p, C.c_size_t]
        for nm in ("sbo_nested_read_consumed", "sbo_nested_read_nleaf_validity"):
            getattr(L, nm).restype = C.c_uint64
            getattr(L, nm).argtypes = [C.c_void_p]
        for nm in ("sbo_nested_read_length", "sbo_nested_read_noffsets", "sbo_nested_read_nvalidity"):
            getattr(L, nm).restype = C.c_uint64
            getattr(L, nm).argtypes = [C.c_void_p, C.c_int32]
        for nm in ("sbo_nested_read_offsets", "sbo_nested_read_validity"):
            getattr(L, nm).restype = C.c_void_p
            getattr(L, nm).argtypes = [C.c_void_p, C.c_int32]
        L.sbo_nested_read_leaf_validity.restype = C.c_void_p
        L.sbo_nested_read_leaf_validity.argtypes = [C.c_void_p]
        L.sbo_nested_read_free.argtypes = [C.c_void_p]
        L._nested_ready = True
    return L


def nested_levels_array(levels, keep):
    """levels: list of dicts(kind, is_optional, validity (packed bits or None), offsets (np or None), length)."""
    arr = (_NestedLevel * len(levels))()
    for k, lv in enumerate(levels):
        v = _bytes_view(lv.get("validity"))
        o = _bytes_view(lv.get("offsets"))
        keep.extend([v, o])
        arr[k].kind, arr[k].is_optional = lv["kind"], int(bool(lv["is_optional"]))
        arr[k].validity, arr[k].validity_off = _ptr(v), lv.get("validity_off", 0)
        arr[k].offsets, arr[k].length = _ptr(o), lv["length"]
    return arr


def nested_write_levels(levels, r0, length):
    """write_nested_validity for top-level rows [r0, r0+length): (bytes, num_values, leaf_start, leaf_count)."""
    L = _nested_lib()
    keep = []
    arr = nested_levels_array(levels, keep)
    err = C.create_string_buffer(512)
    h = L.sbo_nested_write(arr, len(levels), r0, length, err, 512)
    if not h:
        raise OracleError(err.value.decode())
    try:
        n = L.sbo_nested_written_len(h)
        data = np.ctypeslib.as_array(C.cast(L.sbo_nested_written_data(h), C.POINTER(C.c_uint8)), (n,)).copy()
        info = np.zeros(3, np.uint64)
        L.sbo_nested_written_info(h, _ptr(info))
    finally:
        L.sbo_nested_written_free(h)
    return data, int(info[0]), int(info[1]), int(info[2])


def nested_read_levels(page, num_values, kinds, nullable):
    """read_validity_nested: dict(consumed, lengths, offsets[k], validity[k], leaf_validity)."""
    L = _nested_lib()
    page = np.ascontiguousarray(page, dtype=np.uint8)
    kinds = np.ascontiguousarray(kinds, dtype=np.int32)
    nullable = np.ascontiguousarray(nullable, dtype=np.int32)
    err = C.create_string_buffer(512)
    h = L.sbo_nested_read(_ptr(page), page.size, num_values, _ptr(kinds), _ptr(nullable), kinds.size, err, 512)
    if not h:
        raise OracleError(err.value.decode())
    try:
        def grab(ptr, n, ctype, dt):
            if n == 0:
                return np.zeros(0, dt)
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), (n,)).copy()
        D = kinds.size
        out = dict(consumed=int(L.sbo_nested_read_consumed(h)),
                   lengths=[int(L.sbo_nested_read_length(h, k)) for k in range(D)],
                   offsets=[grab(L.sbo_nested_read_offsets(h, k), L.sbo_nested_read_noffsets(h, k), C.c_int64, np.int64) for k in range(D)],
                   validity=[grab(L.sbo_nested_read_validity(h, k), L.sbo_nested_read_nvalidity(h, k), C.c_uint8, np.uint8) for k in range(D)],
                   leaf_validity=grab(L.sbo_nested_read_leaf_validity(h), L.sbo_nested_read_nleaf_validity(h), C.c_uint8, np.uint8))
    finally:
        L.sbo_nested_read_free(h)
    return out
