"""Names shared by the read:: and write:: mirrors (reference: src/lib.rs:40-80,
src/compression/mod.rs:37-108, src/write/common.rs:37-45)."""
from dataclasses import dataclass, field
from typing import List, Optional


class Compression:
    """enum Compression with its on-disk ids (src/compression/mod.rs:37-51,92-108)."""
    NONE, LZ4, ZSTD, SNAPPY = 0, 1, 2, 3
    RLE, DICT, ONE_VALUE, FREQ, BITPACKING, DELTA_BITPACKING, PATAS = 10, 11, 12, 13, 14, 15, 16
    NAMES = {0: "None", 1: "Lz4", 2: "Zstd", 3: "Snappy", 10: "Rle", 11: "Dict", 12: "OneValue", 13: "Freq",
             14: "Bitpacking", 15: "DeltaBitpacking", 16: "Patas"}


class CommonCompression:
    """enum CommonCompression (src/compression/basic.rs:22-28) — the only codec type the
    reference re-exports publicly (src/lib.rs:25)."""
    NONE, LZ4, ZSTD, SNAPPY = 0, 1, 2, 3


class PhysicalType:
    """dispatch key of read_simple / write_simple (src/read/batch_read.rs:37-63)."""
    BOOLEAN, INT8, INT16, INT32, INT64, UINT8, UINT16, UINT32, UINT64 = range(9)
    INT128, INT256, FLOAT32, FLOAT64, BINARY, LARGE_BINARY, NULL = range(9, 16)
    UTF8, LARGE_UTF8 = BINARY, LARGE_BINARY  # written as Binary (src/write/serialize.rs:92-121)
    WIDTH = {1: 1, 5: 1, 2: 2, 6: 2, 3: 4, 7: 4, 11: 4, 4: 8, 8: 8, 12: 8, 9: 16, 10: 32, 13: 4, 14: 8}

    @staticmethod
    def is_binary(t):
        return t in (PhysicalType.BINARY, PhysicalType.LARGE_BINARY)


@dataclass
class PageMeta:
    """src/lib.rs:75-80"""
    length: int
    num_values: int


@dataclass
class ColumnMeta:
    """src/lib.rs:40-70"""
    offset: int
    pages: List[PageMeta] = field(default_factory=list)

    def total_len(self):
        return sum(p.length for p in self.pages)


@dataclass
class WriteOptions:
    """src/write/common.rs:37-45, plus force_codec / force_index_codec / rng_seed (the
    deterministic stand-ins for the debug-only env switches, src/util/env.rs:20-24, and for
    thread_rng() in compress_sample_ratio, src/compression/integer/mod.rs:316)."""
    default_compression: int = CommonCompression.NONE
    default_compress_ratio: Optional[float] = None
    max_page_size: Optional[int] = None
    forbidden_compressions: List[int] = field(default_factory=list)
    force_codec: int = -1
    force_index_codec: int = -1
    rng_seed: int = 42
    lz4_exact: bool = False   # SB_WRITE_LZ4_EXACT: LZ4 blocks byte-identical to LZ4_compress_default (slow serial parse)
    debug_verify_fail: bool = False   # SB_WRITE_DEBUG_VERIFY_FAIL (tests): force the exact re-selection of hashed binary pages
