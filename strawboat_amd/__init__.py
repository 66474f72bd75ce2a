"""strawboat_amd — MI355X-native page encode/decode path of strawboat (sundy-li/strawboat).

Host-side mirror of the reference's `read::` / `write::` page-level API over the C ABI of
libstrawboat_hip.so (hand-written gfx950 kernels).  PyTorch is used for device memory and
streams only.  There is no CPU fallback: without the built library and a GPU, calls raise.
"""
from .types import (ColumnMeta, CommonCompression, Compression, PageMeta, PhysicalType,  # noqa: F401
                    WriteOptions)
from .context import Context  # noqa: F401
from . import read, shard, write  # noqa: F401

__all__ = ["Context", "read", "write", "WriteOptions", "PageMeta", "ColumnMeta", "Compression",
           "CommonCompression", "PhysicalType"]
