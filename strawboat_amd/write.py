"""write:: — the page-encode API of the reference, on the GPU.

Mirrors, per leaf column, the page loop of `NativeWriter::encode_chunk`
(src/write/common.rs:54-109): the column is cut into pages of `max_page_size` rows, every
page goes through `write::write` -> `write_simple` (src/write/serialize.rs:36-132), i.e.
`write_validity` + `compress_integer|double|boolean|binary`, and `PageMeta{length, num_values}`
is recorded.  A call takes a *batch* of columns; every (column, page) is a work item scheduled
over the GPU by libstrawboat_hip.so.  The result is the byte string NativeWriter would have
`write_all`-ed for the column's pages, plus the metas.
"""
import ctypes as C
from dataclasses import dataclass
from typing import List, Optional

import numpy as np

from . import _native as N
from .types import PageMeta, PhysicalType, WriteOptions


@dataclass
class DeviceColumn:
    """Arrow buffers of one flat leaf column in HBM (torch.uint8 tensors)."""
    physical_type: int
    is_nullable: bool          # schema field nullable (pages then carry a def-level section)
    rows: int
    values: object             # primitive values / boolean bitmap / binary value bytes
    validity: Optional[object] = None   # LSB-first bitmap or None
    offsets: Optional[object] = None    # binary: rows+1 offsets (i32 / i64) as bytes
    values_bit_offset: int = 0          # boolean
    validity_bit_offset: int = 0
    first_page_index: int = 0           # pages [first, first + n) of a column encoded on their own (shard.WorkItem)
    column_values_len: int = 0          # binary page ranges: byte length of the column's whole values buffer (0 = this one)


class EncodedColumn:
    """The pages of one leaf column in HBM + ColumnMeta.pages (after Context.synchronize())."""

    def __init__(self, pages, metas_c, cstruct):
        self.pages = pages
        self._metas = metas_c
        self._c = cstruct

    @property
    def n_pages(self):
        return int(self._c.n_pages)

    @property
    def length(self):
        return int(self._c.out_len)

    @property
    def metas(self) -> List[PageMeta]:
        return [PageMeta(int(m.length), int(m.num_values)) for m in self._metas[:self.n_pages]]

    def metas_array(self):
        return np.array([[m.length, m.num_values] for m in self._metas[:self.n_pages]], dtype=np.uint64).reshape(-1, 2)

    def pages_numpy(self):
        return self.pages[:self.length].cpu().numpy()


def options_c(opts: WriteOptions):
    o = N.WriteOptionsC()
    o.default_compression = opts.default_compression
    o.has_default_compress_ratio = 0 if opts.default_compress_ratio is None else 1
    o.default_compress_ratio = 0.0 if opts.default_compress_ratio is None else float(opts.default_compress_ratio)
    o.max_page_size = 0 if opts.max_page_size is None else int(opts.max_page_size)
    mask = 0
    for c in opts.forbidden_compressions:
        mask |= 1 << c
    o.forbidden_compressions = mask
    o.force_codec = opts.force_codec
    o.force_index_codec = opts.force_index_codec
    o.rng_seed = opts.rng_seed
    o.flags = (N.SB_WRITE_LZ4_EXACT if opts.lz4_exact else 0) | ((1 << 30) if getattr(opts, "debug_verify_fail", False) else 0)
    return o


def write_bound(ctx, col: DeviceColumn, oc):
    npages = C.c_uint64(0)
    vlen = col.values.numel() if PhysicalType.is_binary(col.physical_type) else 0
    b = ctx._lib.sb_write_bound(col.physical_type, 1 if col.is_nullable else 0, col.rows, vlen, C.byref(oc),
                                C.byref(npages))
    return int(b), int(npages.value)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None and t.numel() else C.c_void_p(0)


class WriteBatch:
    """A prepared encode of a batch of leaf columns (descriptors and output buffers built
    once; enqueue() costs one C call)."""

    def __init__(self, ctx, columns: List[DeviceColumn], options: WriteOptions,
                 out: Optional[List[EncodedColumn]] = None):
        import torch
        self.ctx = ctx
        n = len(columns)
        oc = options_c(options)
        arr = (N.ColumnWriteC * n)()
        keep = [arr, oc]
        res = []
        with torch.cuda.stream(ctx.torch_stream):
            for i, col in enumerate(columns):
                c = arr[i]
                for name in ("values", "validity", "offsets"):
                    t = getattr(col, name)
                    if t is not None and (t.dtype != torch.uint8 or t.device != ctx.torch_device or
                                          not t.is_contiguous()):
                        raise ValueError("%s must be a contiguous uint8 tensor on %s" % (name, ctx.torch_device))
                c.physical_type = col.physical_type
                c.is_nullable = 1 if col.is_nullable else 0
                c.rows = col.rows
                c.values = _ptr(col.values)
                c.values_bit_offset = col.values_bit_offset
                c.values_len = col.values.numel() if col.values is not None else 0
                c.validity = _ptr(col.validity)
                c.validity_bit_offset = col.validity_bit_offset
                c.offsets = _ptr(col.offsets)
                c.first_page_index = col.first_page_index
                c.column_values_len = col.column_values_len
                if out is not None:
                    pages, metas = out[i].pages, out[i]._metas
                else:
                    bound, npages = write_bound(ctx, col, oc)
                    pages = torch.empty(max(bound, 1), dtype=torch.uint8, device=ctx.torch_device)
                    metas = (N.PageMetaC * max(npages, 1))()
                c.out_pages = _ptr(pages)
                c.out_capacity = pages.numel()
                c.out_metas = metas
                c.n_pages_capacity = len(metas)
                keep.extend([col.values, col.validity, col.offsets, pages, metas])
                res.append(EncodedColumn(pages, metas, c))
        self._arr, self._oc, self._keep, self._n = arr, oc, keep, n
        self.encoded = res

    def enqueue(self):
        ctx = self.ctx
        ctx._keep.append(self)
        ctx._check(ctx._lib.sb_write_columns(ctx._h, self._arr, self._n, C.byref(self._oc), N.SB_MEM_DEVICE))
        return self.encoded


def encode_columns(ctx, columns: List[DeviceColumn], options: WriteOptions,
                   out: Optional[List[EncodedColumn]] = None) -> List[EncodedColumn]:
    """Enqueue the encode of a batch of leaf columns on ctx's stream (valid after
    ctx.synchronize()).  `out` re-uses the buffers of a previous result."""
    return WriteBatch(ctx, columns, options, out).enqueue()


def write(ctx, column: DeviceColumn, options: WriteOptions) -> EncodedColumn:
    """Encode one leaf column (synchronous): the per-column body of encode_chunk."""
    res = encode_columns(ctx, [column], options)[0]
    ctx.synchronize()
    return res
