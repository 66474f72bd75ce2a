"""Nested (list / struct) leaf columns — write_nested / read_nested of the reference on the GPU.

Write side mirrors `write_nested` (src/write/serialize.rs:135-198): per page
`u32 page_rows | u32 rep_len | u32 def_len | rep | def | leaf BLOCK`; the page cut follows
src/write/common.rs:79-107 (top-level rows per page, `slice_parquet_array` for the leaf range,
PageMeta.num_values = number of level entries).  Read side mirrors `read_nested_*` /
`read_validity_nested` (src/read/read_basic.rs:65-173, src/read/array/integer.rs:240-283): the
level streams rebuild list offsets, struct/list validity and leaf validity; the leaf BLOCK is an
ordinary block.

The level work runs in sb_nested_write_levels / sb_nested_read_levels; the leaf BLOCKs go
through the flat sb_write_columns / sb_read_columns with explicit paging, so every codec of
the flat path applies unchanged.
"""
import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np

from . import _native as N
from .read import ColumnPages, DeviceArray
from .types import PageMeta, PhysicalType, WriteOptions
from .write import DeviceColumn, EncodedColumn, options_c

PRIMITIVE, LIST, LARGE_LIST, STRUCT = 0, 1, 2, 3


@dataclass
class NestedLevel:
    """One node on the path root -> leaf (arrow2 `Nested`); buffers are torch.uint8 tensors in HBM."""
    kind: int
    is_optional: bool
    length: int
    validity: Optional[object] = None
    offsets: Optional[object] = None       # lists: (length + 1) i32 / i64 offsets as bytes
    validity_bit_offset: int = 0


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None and t.numel() else C.c_void_p(0)


def _levels_c(levels: Sequence[NestedLevel]):
    arr = (N.NestedLevelC * len(levels))()
    for k, lv in enumerate(levels):
        arr[k].validity = _ptr(lv.validity)
        arr[k].offsets = _ptr(lv.offsets)
        arr[k].validity_bit_offset = lv.validity_bit_offset
        arr[k].length = lv.length
        arr[k].kind = lv.kind
        arr[k].is_optional = 1 if lv.is_optional else 0
    return arr


class NestedLevels:
    """Result of write_levels: every page's level section back to back in HBM + per-page info."""

    def __init__(self, sections, info):
        self.sections = sections
        self.level_bytes = info[:, 0].copy()
        self.num_values = info[:, 1].copy()
        self.leaf_start = info[:, 2].copy()
        self.leaf_count = info[:, 3].copy()

    @property
    def n_pages(self):
        return int(self.level_bytes.size)


class _LevelsWrite:
    """descriptors + outputs of one sb_nested_write_levels_batch call, built once; call() runs it (again)"""

    def __init__(self, ctx, levels_list, max_page_size: Optional[int]):
        import torch
        self.ctx = ctx
        self.n = n = len(levels_list)
        self.mps = mps = 0 if max_page_size is None else int(max_page_size)
        self.items = (N.NestedLevelsWriteC * max(n, 1))()
        self.keep = []
        with torch.cuda.stream(ctx.torch_stream):
            for k, levels in enumerate(levels_list):
                arr = _levels_c(levels)
                rows = levels[0].length
                bound = int(ctx._lib.sb_nested_levels_bound(arr, len(levels), rows, mps))
                ps = min(mps, rows) if mps else rows
                npages = (rows + ps - 1) // ps if rows else 0
                out = torch.empty(max(bound, 1), dtype=torch.uint8, device=ctx.torch_device)
                pages = (N.NestedPageC * max(npages, 1))()
                it = self.items[k]
                it.levels, it.n_levels, it.rows = arr, len(levels), rows
                it.out_levels, it.out_capacity = _ptr(out), out.numel()
                it.pages, it.n_pages_capacity = pages, len(pages)
                self.keep.append((arr, out, pages))

    def call(self):
        if self.n:
            self.ctx._check(self.ctx._lib.sb_nested_write_levels_batch(self.ctx._h, self.items, self.n, self.mps))

    def enqueue(self):
        """the same without the round trip: info() is valid after the context's next synchronize()"""
        if self.n:
            self.ctx._check(self.ctx._lib.sb_nested_write_levels_enqueue(self.ctx._h, self.items, self.n, self.mps))

    def info(self, k):
        """(level_bytes, num_values, leaf_start, leaf_count) per page of leaf column k, as the last call() left them"""
        return np.frombuffer(self.keep[k][2], dtype=np.uint64).reshape(-1, 4)[:int(self.items[k].n_pages)]

    def results(self) -> List[NestedLevels]:
        return [NestedLevels(out, self.info(k).copy()) for k, (_, out, _) in enumerate(self.keep)]


def write_levels_batch(ctx, levels_list, max_page_size: Optional[int]) -> List[NestedLevels]:
    """write_nested_validity for every page of every leaf column of a call: ONE set of launches over all pages of all
    leaves and ONE host round trip for the page cut (sb_nested_write_levels_batch)."""
    lw = _LevelsWrite(ctx, levels_list, max_page_size)
    lw.call()
    return lw.results()


def write_levels(ctx, levels: Sequence[NestedLevel], rows: int, max_page_size: Optional[int]) -> NestedLevels:
    """write_nested_validity for every page of one nested leaf column (synchronous)."""
    assert rows == levels[0].length
    return write_levels_batch(ctx, [levels], max_page_size)[0]


class NestedEncodedColumn(EncodedColumn):
    """Pages of a nested leaf column; PageMeta.num_values counts level entries (write/common.rs:84-92)."""

    def __init__(self, pages, metas_c, cstruct, num_values):
        super().__init__(pages, metas_c, cstruct)
        self._num_values = num_values

    @property
    def metas(self) -> List[PageMeta]:
        return [PageMeta(int(m.length), int(v)) for m, v in zip(self._metas[:self.n_pages], self._num_values)]

    def metas_array(self):
        return np.array([[m.length, v] for m, v in zip(self._metas[:self.n_pages], self._num_values)],
                        dtype=np.uint64).reshape(-1, 2)


class NestedWriteBatch:
    """The two calls behind write_nested_leaves with their descriptors and output buffers built ONCE (what WriteBatch is
    for flat columns): run() = sb_nested_write_levels_batch (level sections of every leaf, one host round trip for the
    page cut) + sb_write_columns (the BLOCKs of all leaves) + synchronize.  A host that writes chunk after chunk of one
    schema keeps such an object per chunk shape instead of allocating per call."""

    def __init__(self, ctx, items, options: WriteOptions):
        import torch
        self.ctx = ctx
        self.n = n = len(items)
        self.oc = oc = options_c(options)
        oc.max_page_size = 0
        self.arr = arr = (N.ColumnWriteC * max(n, 1))()
        self.lw = _LevelsWrite(ctx, [levels for levels, _ in items], options.max_page_size)
        self.lw.call()
        self._fresh = True   # (the level sections of the first run() are already there)
        self.keep, self.outs, self.page_rows, self.heads = [], [], [], []
        for k, (levels, leaf) in enumerate(items):
            info = self.lw.info(k)
            n_pages = info.shape[0]
            if n_pages and int(info[0, 2]) != 0:
                raise ValueError("leaf slice must start at 0")
            c = arr[k]
            c.physical_type = leaf.physical_type
            c.is_nullable = 0  # the def levels carry the validity; the BLOCK has no def section
            total_leaf = int(info[:, 3].sum())
            c.rows = total_leaf
            c.values = _ptr(leaf.values)
            c.values_bit_offset = leaf.values_bit_offset
            c.values_len = leaf.values.numel() if leaf.values is not None else 0
            c.validity = _ptr(leaf.validity)
            c.validity_bit_offset = leaf.validity_bit_offset
            c.offsets = _ptr(leaf.offsets)
            # a (leaf, page range) work item (shard.WorkItem): the sampling seed and the hdr9 / total_bytes of its pages are
            # the single writer's (first page of the range, array.values().len() of the whole leaf column)
            c.first_page_index = leaf.first_page_index
            c.column_values_len = leaf.column_values_len
            vlen = c.values_len if PhysicalType.is_binary(leaf.physical_type) else 0
            npg = C.c_uint64(0)
            bound = int(ctx._lib.sb_write_bound(leaf.physical_type, 0, max(total_leaf, 1), vlen, C.byref(oc), C.byref(npg)))
            bound += n_pages * 512 + int(info[:, 0].sum())
            with torch.cuda.stream(ctx.torch_stream):
                pages = torch.empty(bound, dtype=torch.uint8, device=ctx.torch_device)
            metas = (N.PageMetaC * n_pages)()
            c.out_pages = _ptr(pages)
            c.out_capacity = pages.numel()
            c.out_metas = metas
            c.n_pages_capacity = n_pages
            page_rows = np.ascontiguousarray(info[:, 3], dtype=np.uint64)
            heads = np.ascontiguousarray(info[:, 0], dtype=np.uint64)
            c.page_rows = page_rows.ctypes.data_as(C.c_void_p)
            c.page_head_bytes = heads.ctypes.data_as(C.c_void_p)
            c.page_heads = _ptr(self.lw.keep[k][1])
            c.n_pages_in = n_pages
            self.keep.append((leaf, pages, metas))
            self.page_rows.append(page_rows)
            self.heads.append(heads)
            self.outs.append((pages, metas))

    def run(self) -> List[NestedEncodedColumn]:
        """Encodes what the buffers hold NOW.  The batch was sized by the buffers' first contents: rows per level, page
        count and output capacity are fixed, so the data may change but its level STRUCTURE (list lengths, hence leaf slots
        per page) must still fit — a page cut that does not is refused here, before anything is enqueued.  The returned
        columns alias the batch's output buffers: the next run() overwrites them."""
        ctx = self.ctx
        if not self._fresh:
            # The level sections and the leaf BLOCKs are enqueued TOGETHER, the BLOCKs with the page cut of the run before
            # (src/write/serialize.rs:217-232 writes them back to back); the cut this run's level kernels found comes back
            # with the results, and only when it differs (other list lengths) are the BLOCKs written again with it.
            before = [self.lw.info(k).copy() for k in range(self.n)]
            self.lw.enqueue()
            try:
                if self.n:
                    ctx._check(ctx._lib.sb_write_columns(ctx._h, self.arr, self.n, C.byref(self.oc), N.SB_MEM_DEVICE))
                ctx.synchronize()
                same = all(before[k].shape == self.lw.info(k).shape and np.array_equal(before[k][:, (0, 3)], self.lw.info(k)[:, (0, 3)]) for k in range(self.n))
            except Exception:   # (BLOCKs cut with a stale page table may not even fit: the level call alone, then the checks below)
                self.lw.call()
                same = False
            if same:
                return [NestedEncodedColumn(pages, metas, self.arr[k], self.lw.info(k)[:, 1].copy()) for k, (pages, metas) in enumerate(self.outs)]
            for k in range(self.n):   # the page cut of THIS call
                info = self.lw.info(k)
                if info.shape[0] != self.page_rows[k].shape[0]:
                    raise ValueError("NestedWriteBatch.run(): leaf column %d now has %d pages, the batch was built for %d" % (k, info.shape[0], self.page_rows[k].shape[0]))
                leaf_slots, heads = int(info[:, 3].sum()), int(info[:, 0].sum())
                if leaf_slots != int(self.arr[k].rows):
                    raise ValueError("NestedWriteBatch.run(): leaf column %d now has %d leaf slots, the batch was built for %d (build a new "
                                     "batch for a chunk with other list lengths)" % (k, leaf_slots, int(self.arr[k].rows)))
                if heads > int(self.heads[k].sum()) + self.page_rows[k].shape[0] * 512:
                    raise ValueError("NestedWriteBatch.run(): the level sections of leaf column %d outgrew the output capacity" % k)
                self.page_rows[k][:] = info[:, 3]
                self.heads[k][:] = info[:, 0]
        self._fresh = False
        if self.n:
            ctx._check(ctx._lib.sb_write_columns(ctx._h, self.arr, self.n, C.byref(self.oc), N.SB_MEM_DEVICE))
        ctx.synchronize()
        return [NestedEncodedColumn(pages, metas, self.arr[k], self.lw.info(k)[:, 1].copy()) for k, (pages, metas) in enumerate(self.outs)]


def write_nested_leaves(ctx, items, options: WriteOptions) -> List[NestedEncodedColumn]:
    """Encode the leaf columns of one nested array together (synchronous): `items` = [(levels, leaf), ...] — what
    NativeWriter::encode_chunk's loop over the leaves of an array does (src/write/common.rs:60-116), with the level
    sections of every leaf written first and the leaf BLOCKs of all leaves in ONE sb_write_columns call, so that the
    pages of all leaves are in flight together.  For each item leaf.rows = levels[-1].length; its validity is the leaf
    validity, also referenced by levels[-1].validity; options.max_page_size counts TOP-LEVEL rows like upstream."""
    wb = NestedWriteBatch(ctx, items, options)
    ctx._keep.append(wb)
    return wb.run()


def write_nested(ctx, levels: Sequence[NestedLevel], leaf: DeviceColumn, options: WriteOptions) -> NestedEncodedColumn:
    """Encode one nested leaf column (synchronous): write_nested_leaves with one leaf."""
    return write_nested_leaves(ctx, [(levels, leaf)], options)[0]


class NestedArray:
    """Decoded nested column: per level the column-level offsets / validity, plus the leaf array."""

    def __init__(self, kinds, nullable, lengths, offsets, validity, leaf: DeviceArray):
        self.kinds, self.nullable, self.lengths = kinds, nullable, lengths
        self.offsets, self.validity, self.leaf = offsets, validity, leaf

    def offsets_numpy(self, k):
        return self.offsets[k][:(self.lengths[k] + 1) * 8].cpu().numpy().view(np.int64)

    def validity_numpy(self, k):
        return self.validity[k][:(self.lengths[k] + 7) // 8].cpu().numpy()


class NestedReadBatch:
    """The calls behind read_nested_leaves with their descriptors and output buffers built ONCE (what ReadBatch is for
    flat columns): run() = sb_nested_read_levels_batch (offsets / validity per level, leaf validity, per page the leaf
    count and where its BLOCK starts: one host round trip) + sb_read_columns over the BLOCKs of all leaves + synchronize.
    The buffers are sized by the first pass (which includes the sizing call for binary leaves)."""

    def __init__(self, ctx, columns, kinds_list, nullable_list):
        import torch
        self.ctx = ctx
        dev = ctx.torch_device
        self.n = n = len(columns)
        self.arr = arr = (N.ColumnReadC * max(n, 1))()
        self.items = items = (N.NestedLevelsReadC * max(n, 1))()
        self.prep, self.per, self.keep, self.bufs = [], [], [], []
        for j, (column, kinds, nullable) in enumerate(zip(columns, kinds_list, nullable_list)):
            D = len(kinds)
            metas = np.ascontiguousarray(column.metas_array(), dtype=np.uint64).reshape(-1, 2)
            n_pages = metas.shape[0]
            entries = int(metas[:, 1].sum()) if n_pages else 0
            lv = (N.NestedLevelOutC * D)()
            offs, vals = [None] * D, [None] * D
            vbytes = ((entries + 31) // 32) * 4
            with torch.cuda.stream(ctx.torch_stream):
                for k in range(D):
                    lv[k].kind = kinds[k]
                    lv[k].is_nullable = 1 if nullable[k] else 0
                    if kinds[k] in (LIST, LARGE_LIST):
                        offs[k] = torch.empty((entries + 1) * 8, dtype=torch.uint8, device=dev)
                        lv[k].offsets = _ptr(offs[k])
                        lv[k].offsets_capacity = entries + 1
                    if nullable[k] and kinds[k] != PRIMITIVE:
                        vals[k] = torch.empty(max(vbytes, 4), dtype=torch.uint8, device=dev)
                        lv[k].validity = _ptr(vals[k])
                        lv[k].validity_capacity = vals[k].numel()
                leaf_validity = torch.empty(max(vbytes, 4), dtype=torch.uint8, device=dev) if nullable[-1] else None
            counts = np.zeros(max(n_pages, 1), np.uint64)
            block_offs = np.zeros(max(n_pages, 1), np.uint64)
            pages = column.pages
            it = items[j]
            it.pages, it.pages_len = _ptr(pages), pages.numel()
            it.metas, it.n_pages = metas.ctypes.data_as(C.c_void_p), n_pages
            it.levels, it.n_levels = lv, D
            it.leaf_validity = _ptr(leaf_validity)
            it.leaf_validity_capacity = leaf_validity.numel() if leaf_validity is not None else 0
            it.page_leaf_counts = counts.ctypes.data_as(C.c_void_p)
            it.page_block_offsets = block_offs.ctypes.data_as(C.c_void_p)
            starts = np.concatenate([[0], np.cumsum(metas[:, 0])[:-1]]).astype(np.uint64) if n_pages else np.zeros(0, np.uint64)
            leaf_metas = np.zeros((n_pages, 2), np.uint64)
            po = np.zeros(n_pages, np.uint64)
            self.prep.append((metas, n_pages, lv, offs, vals, leaf_validity, counts, block_offs, pages, starts, leaf_metas, po))
        self._levels()
        for j, (column, kinds, nullable) in enumerate(zip(columns, kinds_list, nullable_list)):
            metas, n_pages, lv, offs, vals, leaf_validity, counts, block_offs, pages, starts, leaf_metas, po = self.prep[j]
            t = column.physical_type
            c = arr[j]
            c.physical_type = t
            c.is_nullable = 0
            c.pages = _ptr(pages)
            c.pages_len = pages.numel()
            c.metas = leaf_metas.ctypes.data_as(C.POINTER(N.PageMetaC))
            c.n_pages = n_pages
            c.page_offsets = po.ctypes.data_as(C.c_void_p)
            self.per.append(dict(kinds=list(kinds), nullable=list(nullable), offs=offs, vals=vals, leaf_validity=leaf_validity,
                                 rows=int(counts[:n_pages].sum()), t=t))
        binary = [j for j in range(n) if PhysicalType.is_binary(self.per[j]["t"])]
        if binary:   # values_len of the binary leaves: one sizing call over all of them
            sub = (N.ColumnReadC * len(binary))()
            for q, j in enumerate(binary):
                C.memmove(C.byref(sub[q]), C.byref(arr[j]), C.sizeof(N.ColumnReadC))
            ctx._check(ctx._lib.sb_read_columns_sizes(ctx._h, sub, len(binary), N.SB_MEM_DEVICE))
            for q, j in enumerate(binary):
                self.per[j]["values_len"] = int(sub[q].values_len)
        with torch.cuda.stream(ctx.torch_stream):
            for j in range(n):
                t, rows = self.per[j]["t"], self.per[j]["rows"]
                values = offsets = None
                if t == PhysicalType.BOOLEAN:
                    values = torch.empty(((rows + 31) // 32) * 4 + 4, dtype=torch.uint8, device=dev)
                elif PhysicalType.is_binary(t):
                    values = torch.empty(max(self.per[j]["values_len"], 1), dtype=torch.uint8, device=dev)
                    offsets = torch.empty((rows + 1) * PhysicalType.WIDTH[t], dtype=torch.uint8, device=dev)
                elif t != PhysicalType.NULL:
                    values = torch.empty(max(rows * PhysicalType.WIDTH[t], 1), dtype=torch.uint8, device=dev)
                c = arr[j]
                c.values = _ptr(values)
                c.values_capacity = values.numel() if values is not None else 0
                c.offsets = _ptr(offsets)
                c.offsets_capacity = offsets.numel() if offsets is not None else 0
                self.bufs.append((values, offsets))
        self._fresh = True   # (the level outputs of the first run() are already there)

    def _levels(self, enqueue_only=False):
        """the level sections of every leaf (one set of launches, one host round trip), then the page table of the BLOCKs"""
        ctx = self.ctx
        if self.n:
            if enqueue_only:
                ctx._check(ctx._lib.sb_nested_read_levels_enqueue(ctx._h, self.items, self.n))
                return
            ctx._check(ctx._lib.sb_nested_read_levels_batch(ctx._h, self.items, self.n))
        self._page_table()

    def _page_table(self):
        """leaf counts / BLOCK offsets of the last level call -> the page table of the leaf call; False: nothing changed"""
        changed = False
        for metas, n_pages, lv, offs, vals, leaf_validity, counts, block_offs, pages, starts, leaf_metas, po in self.prep:
            lm0 = metas[:, 0] - (block_offs[:n_pages] - starts)
            if not (np.array_equal(leaf_metas[:, 0], lm0) and np.array_equal(leaf_metas[:, 1], counts[:n_pages]) and np.array_equal(po, block_offs[:n_pages])):
                changed = True
            leaf_metas[:, 0] = lm0
            leaf_metas[:, 1] = counts[:n_pages]
            po[:] = block_offs[:n_pages]
        return changed

    def run(self) -> List[NestedArray]:
        ctx = self.ctx
        if not self._fresh:
            # level sections and leaf BLOCKs enqueued together, the BLOCKs with the page table of the run before
            # (src/read/read_basic.rs:65-173 then the leaf decode); the table this run's level kernels found comes back with
            # the results, and only when it differs (other pages behind the same descriptors) are the BLOCKs read again
            self._levels(enqueue_only=True)
            try:
                if self.n:
                    ctx._check(ctx._lib.sb_read_columns(ctx._h, self.arr, self.n, N.SB_MEM_DEVICE))
                ctx.synchronize()
                again = self._page_table()
            except Exception:   # (BLOCKs read with a stale page table may look corrupt: the level call alone, then the leaf call)
                self._levels()
                again = True
            if again and self.n:
                ctx._check(ctx._lib.sb_read_columns(ctx._h, self.arr, self.n, N.SB_MEM_DEVICE))
                ctx.synchronize()
        else:
            self._fresh = False
            if self.n:
                ctx._check(ctx._lib.sb_read_columns(ctx._h, self.arr, self.n, N.SB_MEM_DEVICE))
            ctx.synchronize()
        out = []
        for j in range(self.n):
            d = self.per[j]
            lv = self.prep[j][2]
            lengths = [int(lv[k].length) for k in range(len(d["kinds"]))]
            leaf = DeviceArray(d["t"], bool(d["nullable"][-1]), d["rows"], self.bufs[j][0], d["leaf_validity"], self.bufs[j][1], self.arr[j])
            out.append(NestedArray(d["kinds"], d["nullable"], lengths, d["offs"], d["vals"], leaf))
        return out


def read_nested_leaves(ctx, columns, kinds_list, nullable_list) -> List[NestedArray]:
    """read_nested_* for the leaf columns of one nested array together (synchronous): per leaf the level sections are
    decoded (offsets / validity per level, leaf validity, per page the leaf count and where its BLOCK starts), then the
    BLOCKs of ALL leaves go through the flat decoder in one sizing call (binary leaves) and one sb_read_columns call.
    `kinds` / `nullable` per leaf are the InitNested chain root -> leaf (src/read/batch_read.rs:66-230 builds it from
    the schema)."""
    rb = NestedReadBatch(ctx, columns, kinds_list, nullable_list)
    ctx._keep.append(rb)
    return rb.run()


def read_nested(ctx, column: ColumnPages, kinds: Sequence[int], nullable: Sequence[bool]) -> NestedArray:
    """read_nested_* for one leaf column (synchronous): read_nested_leaves with one leaf."""
    return read_nested_leaves(ctx, [column], [kinds], [nullable])[0]
