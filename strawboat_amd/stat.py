"""Page inspector: the block structure of a column's pages without decoding them — the host-side
mirror of src/stat.rs (ColumnInfo / PageInfo / PageBody, stat_simple :61-80) over the C entry point
sb_stat_page.  Works on host bytes (what NativeReader hands to stat_simple)."""
import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from . import _native as N
from .types import Compression


@dataclass
class PageBody:
    """enum PageBody (src/stat.rs:39-49): `kind` is the variant, Dict / Freq carry their bodies"""
    kind: str                                   # "Dict" | "Freq" | "OneValue" | "Rle" | "Patas" | "Bitpack" | "DeltaBitpack" | "Common"
    common: Optional[int] = None                # Common(CommonCompression): Compression.NONE / LZ4 / ZSTD / SNAPPY
    indices: Optional["PageInfo"] = None        # DictPageBody.indices
    unique_num: Optional[int] = None            # DictPageBody.unique_num
    exceptions: Optional["PageInfo"] = None     # FreqPageBody.exceptions (None for binary columns)
    exceptions_bitmap_size: Optional[int] = None


@dataclass
class PageInfo:
    validity_size: Optional[int]
    compressed_size: int
    uncompressed_size: int
    body: PageBody
    codec: int = -1                             # the on-disk codec id (not a field upstream; handy for tests)


@dataclass
class ColumnInfo:
    physical_type: int
    is_nullable: bool
    pages: List[PageInfo] = field(default_factory=list)


_KIND = {Compression.DICT: "Dict", Compression.FREQ: "Freq", Compression.ONE_VALUE: "OneValue", Compression.RLE: "Rle",
         Compression.PATAS: "Patas", Compression.BITPACKING: "Bitpack", Compression.DELTA_BITPACKING: "DeltaBitpack"}


def _build(chain, k):
    c = chain[k]
    nested = _build(chain, k + 1) if c.has_nested else None
    codec = int(c.codec)
    if codec == Compression.DICT:
        body = PageBody("Dict", indices=nested, unique_num=int(c.unique_num))
    elif codec == Compression.FREQ:
        body = PageBody("Freq", exceptions=nested, exceptions_bitmap_size=int(c.exceptions_bitmap_size))
    elif codec in _KIND:
        body = PageBody(_KIND[codec])
    else:
        body = PageBody("Common", common=codec)
    return PageInfo(int(c.validity_size) if c.has_validity_size else None, int(c.compressed_size), int(c.uncompressed_size),
                    body, codec)


def stat_page(page: bytes, physical_type: int, is_nullable: bool) -> PageInfo:
    lib = N.load()
    buf = np.frombuffer(page, np.uint8) if not isinstance(page, np.ndarray) else np.ascontiguousarray(page, np.uint8)
    out = (N.PageInfoC * 8)()
    n = C.c_uint32(0)
    rc = lib.sb_stat_page(buf.ctypes.data_as(C.c_void_p), buf.size, int(physical_type), 1 if is_nullable else 0, out, 8, C.byref(n))
    if rc != 0:
        raise N.NativeError(rc, "sb_stat_page: page shorter than its headers" if rc == -3 else "sb_stat_page failed")
    return _build(out, 0)


def stat_simple(pages, metas, physical_type: int, is_nullable: bool) -> ColumnInfo:
    """stat_simple (src/stat.rs:61-80): `pages` = the column's pages back to back (host bytes / numpy uint8),
    `metas` = ColumnMeta.pages as an (n, 2) array or a list of PageMeta"""
    buf = np.frombuffer(pages, np.uint8) if not isinstance(pages, np.ndarray) else pages
    info = ColumnInfo(int(physical_type), bool(is_nullable))
    off = 0
    for m in metas:
        length = int(m[0]) if not hasattr(m, "length") else int(m.length)
        info.pages.append(stat_page(buf[off:off + length], physical_type, is_nullable))
        off += length
    return info
