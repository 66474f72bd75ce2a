"""read:: — the batch page-decode API of the reference, on the GPU.

Mirrors `batch_read::read_simple` (src/read/batch_read.rs:27-64), which dispatches a leaf
column's pages to read_integer / read_double / read_boolean / read_binary
(src/read/array/integer.rs:210-238, boolean.rs:191-219, binary.rs:223-265): all pages of the
column are decoded back to back into one set of Arrow buffers.  Here a call takes a *batch*
of columns and every (column, page) is scheduled over the GPU by libstrawboat_hip.so.
"""
import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np

from . import _native as N
from .types import PageMeta, PhysicalType


@dataclass
class ColumnPages:
    """One leaf column as it sits in a strawboat file: its pages back to back plus
    ColumnMeta.pages (src/lib.rs:40-80)."""
    physical_type: int
    is_nullable: bool
    pages: object            # torch.uint8 tensor on the context's device (or host numpy for mem='host')
    metas: Sequence          # PageMeta list or an array of shape [n, 2] (length, num_values)

    def metas_array(self):
        if isinstance(self.metas, np.ndarray):
            return np.ascontiguousarray(self.metas, dtype=np.uint64).reshape(-1, 2)
        return np.array([[m.length, m.num_values] for m in self.metas], dtype=np.uint64).reshape(-1, 2)


class DeviceArray:
    """Decoded Arrow buffers of one column in HBM (values / validity bitmap / offsets)."""

    def __init__(self, physical_type, is_nullable, rows, values, validity, offsets, cstruct):
        self.physical_type = physical_type
        self.is_nullable = is_nullable
        self.rows = rows
        self.values = values
        self.validity = validity
        self.offsets = offsets
        self._c = cstruct

    @property
    def values_len(self):
        """bytes produced in `values` (valid after Context.synchronize())."""
        return int(self._c.values_len)

    # ---- host views for tests / callers that want numpy
    def values_numpy(self):
        if self.values is None:
            return np.zeros(0, np.uint8)
        n = self.values_len
        if self.physical_type == PhysicalType.BOOLEAN:
            n = (self.rows + 7) // 8
        return self.values[:n].cpu().numpy()

    def validity_numpy(self):
        if self.validity is None:
            return None
        return self.validity[:(self.rows + 7) // 8].cpu().numpy()

    def offsets_numpy(self):
        if self.offsets is None:
            return None
        w = PhysicalType.WIDTH[self.physical_type]
        return self.offsets[:(self.rows + 1) * w].cpu().numpy()


def _dev_ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None and t.numel() else C.c_void_p(0)


def _prepare(ctx, columns):
    import torch
    n = len(columns)
    arr = (N.ColumnReadC * n)()
    keep = [arr]
    for i, col in enumerate(columns):
        metas = col.metas_array()
        keep.append(metas)
        c = arr[i]
        c.physical_type = col.physical_type
        c.is_nullable = 1 if col.is_nullable else 0
        pages = col.pages
        if not isinstance(pages, torch.Tensor):
            raise TypeError("ColumnPages.pages must be a torch.uint8 tensor on the context's device")
        if pages.dtype != torch.uint8 or pages.device != ctx.torch_device or not pages.is_contiguous():
            raise ValueError("pages must be a contiguous uint8 tensor on %s" % ctx.torch_device)
        keep.append(pages)
        c.pages = _dev_ptr(pages)
        c.pages_len = pages.numel()
        c.metas = metas.ctypes.data_as(C.POINTER(N.PageMetaC))
        c.n_pages = metas.shape[0]
    return arr, keep


def batch_read_sizes(ctx, columns: List[ColumnPages]):
    """Parse only the page headers on the device: (rows, values bytes) per column.
    The reference has no such call (it over-allocates 4x the page bytes for binary columns,
    src/read/array/binary.rs:241); a device caller needs the exact size to allocate HBM."""
    arr, keep = _prepare(ctx, columns)
    ctx._keep.append(keep)
    ctx._check(ctx._lib.sb_read_columns_sizes(ctx._h, arr, len(columns), N.SB_MEM_DEVICE))
    ctx._keep.clear()
    return [(int(arr[i].rows), int(arr[i].values_len)) for i in range(len(columns))]


class ReadBatch:
    """A prepared decode of a batch of leaf columns: the C descriptors and the output buffers
    are built once; enqueue() then costs one C call (steady-state readers, bench.py)."""

    def __init__(self, ctx, columns: List[ColumnPages], values_capacity: Optional[Sequence[int]] = None,
                 out: Optional[List[DeviceArray]] = None):
        import torch
        self.ctx = ctx
        arr, keep = _prepare(ctx, columns)
        n = len(columns)
        need_sizes = [i for i, col in enumerate(columns)
                      if PhysicalType.is_binary(col.physical_type) and values_capacity is None and out is None]
        caps = list(values_capacity) if values_capacity is not None else [None] * n
        if need_sizes:
            sizes = batch_read_sizes(ctx, [columns[i] for i in need_sizes])
            for i, (_, vlen) in zip(need_sizes, sizes):
                caps[i] = vlen
        res = []
        dev = ctx.torch_device
        with torch.cuda.stream(ctx.torch_stream):
            for i, col in enumerate(columns):
                c = arr[i]
                rows = int(col.metas_array()[:, 1].sum()) if c.n_pages else 0
                t = col.physical_type
                if out is not None:
                    o = out[i]
                    values, validity, offsets = o.values, o.validity, o.offsets
                else:
                    validity = offsets = None
                    if t == PhysicalType.NULL:
                        values = None
                    elif t == PhysicalType.BOOLEAN:
                        values = torch.empty(((rows + 31) // 32) * 4, dtype=torch.uint8, device=dev)
                    elif PhysicalType.is_binary(t):
                        values = torch.empty(max(int(caps[i]), 1), dtype=torch.uint8, device=dev)
                        offsets = torch.empty((rows + 1) * PhysicalType.WIDTH[t], dtype=torch.uint8, device=dev)
                    else:
                        values = torch.empty(rows * PhysicalType.WIDTH[t], dtype=torch.uint8, device=dev)
                    if col.is_nullable and t != PhysicalType.NULL:
                        validity = torch.empty(((rows + 31) // 32) * 4, dtype=torch.uint8, device=dev)
                keep.extend([values, validity, offsets])
                c.values = _dev_ptr(values)
                c.values_capacity = values.numel() if values is not None else 0
                c.validity = _dev_ptr(validity)
                c.validity_capacity = validity.numel() if validity is not None else 0
                c.offsets = _dev_ptr(offsets)
                c.offsets_capacity = offsets.numel() if offsets is not None else 0
                res.append(DeviceArray(t, col.is_nullable, rows, values, validity, offsets, c))
        self._arr, self._keep, self._n = arr, keep, n
        self.arrays = res

    def enqueue(self):
        ctx = self.ctx
        ctx._keep.append(self)
        ctx._check(ctx._lib.sb_read_columns(ctx._h, self._arr, self._n, N.SB_MEM_DEVICE))
        return self.arrays


def batch_read_columns(ctx, columns: List[ColumnPages], values_capacity: Optional[Sequence[int]] = None,
                       out: Optional[List[DeviceArray]] = None) -> List[DeviceArray]:
    """Enqueue the decode of a batch of leaf columns on ctx's stream; returns the device
    arrays (valid after ctx.synchronize()).  `out` re-uses previously returned arrays' buffers."""
    return ReadBatch(ctx, columns, values_capacity, out).enqueue()


def read_simple(ctx, column: ColumnPages) -> DeviceArray:
    """batch_read::read_simple for one leaf column (synchronous)."""
    res = batch_read_columns(ctx, [column])[0]
    ctx.synchronize()
    return res
