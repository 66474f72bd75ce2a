"""Multi-GPU sharding of the page path.

Units of work are (leaf column, page); every page is self-contained (own def-level section, own
codec header, own dictionary: src/write/common.rs:79-109, src/compression/integer/dict.rs:42),
so page ranges of columns are dealt out to ranks with no data-path collective.  The only exchange
is the metadata a file writer needs to lay the columns out: ONE all_gather of fixed-capacity
(column, page, length, num_values) records (the capacity follows from the plan, which every rank
computes identically, so no size exchange precedes it), after which every rank derives
ColumnMeta.offset by an exclusive scan (ColumnMeta semantics: src/lib.rs:40-70,
src/write/common.rs:76,111-114).  With the NCCL backend this is RCCL over xGMI.
"""
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from .types import ColumnMeta, PageMeta

FILE_HEADER_BYTES = 8  # "ARROW2" + 2 pad bytes (src/lib.rs:34, src/write/writer.rs:98-100)


@dataclass(frozen=True)
class WorkItem:
    """pages [first_page, first_page + n_pages) of one leaf column"""
    column: int
    first_page: int
    n_pages: int
    weight: int   # uncompressed Arrow bytes of the range (the balancing key)


def plan_shards(column_bytes: Sequence[int], world: int) -> List[List[int]]:
    """Whole columns onto ranks: greedy longest-processing-time packing by uncompressed Arrow bytes (Utf8 columns are
    ~3x heavier than Boolean ones, so round-robin would skew)."""
    order = sorted(range(len(column_bytes)), key=lambda i: (-int(column_bytes[i]), i))
    load = [0] * world
    shards: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        shards[r].append(i)
        load[r] += int(column_bytes[i])
    for s in shards:
        s.sort()
    return shards


def plan_work_items(columns: Sequence[Tuple[int, int]], world: int, pages_per_item: Optional[int] = None) -> List[List[WorkItem]]:
    """(column, page-range) work items onto ranks.  `columns`: (uncompressed Arrow bytes, n_pages) per leaf column.
    A column is cut into ranges of `pages_per_item` pages (default: so that every column yields about 2 * world items;
    C4: 153 pages per column, C5: 16) — pages of a column are independent, so a heavy Utf8 column does not have to
    sit on one GPU.  Ranges are packed longest-first; every rank computes the same plan."""
    items: List[WorkItem] = []
    for c, (nbytes, npages) in enumerate(columns):
        if npages <= 0:
            continue
        step = pages_per_item or max(1, -(-npages // (2 * world)))
        for p0 in range(0, npages, step):
            n = min(step, npages - p0)
            items.append(WorkItem(c, p0, n, int(nbytes) * n // npages))
    items.sort(key=lambda it: (-it.weight, it.column, it.first_page))
    load = [0] * world
    out: List[List[WorkItem]] = [[] for _ in range(world)]
    for it in items:
        r = min(range(world), key=lambda k: (load[k], k))
        out[r].append(it)
        load[r] += it.weight
    for s in out:
        s.sort(key=lambda it: (it.column, it.first_page))
    return out


def record_capacity(plan: Sequence[Sequence[WorkItem]]) -> int:
    """records (= pages) the busiest rank contributes: the fixed size of every rank's all_gather buffer"""
    return max([sum(it.n_pages for it in s) for s in plan] + [1])


def gather_metas(local, n_columns: int, capacity: Optional[int] = None, device=None, group=None,
                 expected_pages: Optional[Sequence[int]] = None) -> List[np.ndarray]:
    """The one collective of the path.  `local`: what this rank encoded — a dict {column: metas} (whole columns) or a
    list of (column, first_page, metas) with metas = uint64 [n, 2] (length, num_values).  `capacity`: records per rank
    (record_capacity(plan)); when omitted the ranks fall back to agreeing on it first (a second, 8-byte all_gather).
    Returns the page metas of all columns, in column and page order, on every rank.  Every column's pages must arrive
    exactly once and without gaps (page indices 0..n-1; n = expected_pages[column] when the plan's counts are passed):
    a rank that contributed nothing or a range dealt twice raises here instead of shifting every later ColumnMeta.offset."""
    import torch
    import torch.distributed as dist
    if isinstance(local, dict):
        local = [(c, 0, m) for c, m in sorted(local.items())]
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rec = [(c, p0 + k, int(m[k][0]), int(m[k][1])) for c, p0, m in local for k in range(len(m))]
    if world > 1:
        mine = torch.tensor(rec, dtype=torch.int64).reshape(-1, 4)
        if capacity is None:
            count = torch.tensor([mine.shape[0]], dtype=torch.int64, device=device if device is not None else "cpu")
            counts = [torch.zeros_like(count) for _ in range(world)]
            dist.all_gather(counts, count, group=group)
            capacity = max(int(c.item()) for c in counts)
        if mine.shape[0] > capacity:
            raise ValueError("this rank holds %d page records, capacity is %d" % (mine.shape[0], capacity))
        padded = torch.full((max(capacity, 1), 4), -1, dtype=torch.int64)
        padded[:mine.shape[0]] = mine
        if device is not None:
            padded = padded.to(device)
        out = [torch.empty_like(padded) for _ in range(world)]
        dist.all_gather(out, padded, group=group)   # RCCL over xGMI when the backend is nccl
        rec = [tuple(r) for t in out for r in t.cpu().tolist() if r[0] >= 0]
    pages: Dict[int, Dict[int, tuple]] = {}
    for c, p, length, nv in rec:
        if not 0 <= c < n_columns:
            raise ValueError("page record of column %d, but the job has %d columns" % (c, n_columns))
        col = pages.setdefault(c, {})
        if p in col:
            raise ValueError("page %d of column %d arrived twice (a page range was dealt to two ranks)" % (p, c))
        col[p] = (length, nv)
    res = []
    for c in range(n_columns):
        pp = pages.get(c, {})
        n = len(pp)
        if expected_pages is not None and n != int(expected_pages[c]):
            raise ValueError("column %d: %d of its %d pages arrived" % (c, n, int(expected_pages[c])))
        if n and (min(pp) != 0 or max(pp) != n - 1):
            raise ValueError("column %d: page indices %d..%d for %d pages (a page range is missing)" % (c, min(pp), max(pp), n))
        res.append(np.array([pp[p] for p in range(n)], dtype=np.uint64).reshape(-1, 2))
    return res


def column_metas(metas: Sequence[np.ndarray], start: int = FILE_HEADER_BYTES) -> List[ColumnMeta]:
    """ColumnMeta.offset of every leaf column: columns are laid out back to back after the file
    header, pages of a column back to back (src/write/common.rs:60-116)."""
    res = []
    off = start
    for m in metas:
        m = np.asarray(m, dtype=np.uint64).reshape(-1, 2)
        res.append(ColumnMeta(off, [PageMeta(int(a), int(b)) for a, b in m]))
        off += int(m[:, 0].sum()) if len(m) else 0
    return res


def slice_column(col: dict, first_page: int, n_pages: int, page_rows: int) -> dict:
    """rows of pages [first_page, first_page + n_pages) of a flat host column (dict of numpy buffers, the shape
    workloads.py / tests use): what the rank that owns the work item uploads.  Page boundaries are multiples of
    page_rows, so the pages it writes are the very pages a single writer would (write/common.rs:54-58)."""
    r0 = first_page * page_rows
    r1 = min(col["rows"], (first_page + n_pages) * page_rows)
    out = dict(col, rows=r1 - r0)
    if col["validity"] is not None:
        bits = np.unpackbits(col["validity"], bitorder="little")[r0:r1]
        out["validity"] = np.packbits(bits, bitorder="little")
    if col["offsets"] is not None:
        offs = np.asarray(col["offsets"])
        out["offsets"] = offs[r0:r1 + 1] - offs[r0]
        out["values"] = np.asarray(col["values"])[int(offs[r0]):int(offs[r1])]
        out["column_values_len"] = int(np.asarray(col["values"]).size)   # array.values().len() of the whole column
    elif col["ptype"] == 0:  # boolean bitmap
        bits = np.unpackbits(col["values"], bitorder="little")[r0:r1]
        out["values"] = np.packbits(bits, bitorder="little")
    else:
        out["values"] = np.asarray(col["values"])[r0:r1]
    return out
