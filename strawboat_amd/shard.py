"""Multi-GPU sharding of the page path.

Units of work are (leaf column, page); every page is self-contained (own def-level section, own
codec header, own dictionary: src/write/common.rs:79-109, src/compression/integer/dict.rs:42),
so columns are dealt out to ranks with no data-path collective.  The only exchange is the
metadata a file writer needs to lay the columns out: one all_gather of the per-page
(length, num_values) pairs (16 bytes per page), after which every rank derives
ColumnMeta.offset by an exclusive scan (ColumnMeta semantics: src/lib.rs:40-70,
src/write/common.rs:76,111-114).  With the NCCL backend this is RCCL over xGMI.
"""
from typing import Dict, List, Sequence

import numpy as np

from .types import ColumnMeta, PageMeta

FILE_HEADER_BYTES = 8  # "ARROW2" + 2 pad bytes (src/lib.rs:34, src/write/writer.rs:98-100)


def plan_shards(column_bytes: Sequence[int], world: int) -> List[List[int]]:
    """Greedy longest-processing-time bin packing of columns (by uncompressed Arrow bytes) onto
    `world` ranks: Utf8 columns are ~3x heavier than Boolean ones, so round-robin would skew."""
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


def gather_metas(local: Dict[int, np.ndarray], n_columns: int, device=None, group=None) -> List[np.ndarray]:
    """all_gather of the page metas.  `local` maps the column indices this rank encoded to
    uint64 arrays [n_pages, 2] = (length, num_values).  Returns the metas of all columns, in
    column order, on every rank.  One collective: ranks first agree on the padded size."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return [np.asarray(local[i], dtype=np.uint64).reshape(-1, 2) for i in range(n_columns)]
    # flat records: (column, page, length, num_values)
    rec = [(c, p, int(m[p, 0]), int(m[p, 1])) for c, m in sorted(local.items()) for p in range(len(m))]
    mine = torch.tensor(rec, dtype=torch.int64).reshape(-1, 4)
    if device is not None:
        mine = mine.to(device)
    count = torch.tensor([mine.shape[0]], dtype=torch.int64, device=mine.device)
    counts = [torch.zeros_like(count) for _ in range(world)]
    dist.all_gather(counts, count, group=group)
    cap = max(int(c.item()) for c in counts)
    padded = torch.full((max(cap, 1), 4), -1, dtype=torch.int64, device=mine.device)
    padded[:mine.shape[0]] = mine
    out = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(out, padded, group=group)   # RCCL over xGMI when the backend is nccl
    pages: Dict[int, Dict[int, tuple]] = {}
    for t in out:
        for c, p, length, nv in t.cpu().tolist():
            if c >= 0:
                pages.setdefault(c, {})[p] = (length, nv)
    res = []
    for c in range(n_columns):
        pp = pages.get(c, {})
        res.append(np.array([pp[p] for p in sorted(pp)], dtype=np.uint64).reshape(-1, 2))
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
