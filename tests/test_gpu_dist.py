"""N > 1 with the DEVICE encoder: two ranks (two contexts on the one GPU of the test box; the metadata collective runs
over gloo here — two NCCL ranks cannot share a device — and over RCCL in bench.py) encode their (column, page-range)
work items, exchange the page metas with ONE fixed-capacity all_gather, and arrive at the ColumnMetas and page bytes of
a single-process run (which equal the oracle's)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
PAGE = 2048


def _columns():
    sys.path.insert(0, ROOT)
    from oracle import sbo as S
    from tests import gen
    return [gen.prim(S.T_I32, 30_000, uniq=900, null_density=0.1, seed=1), gen.prim(S.T_F64, 30_000, uniq=40, runs=20, null_density=0.1, seed=2),
            gen.binary(30_000, uniq=500, zipf=1.2, null_density=0.1, seed=3), gen.boolean(30_000, null_density=0.1, seed=4),
            gen.prim(S.T_I64, 7_000, uniq=1 << 40, seed=5)]


def _encode(ctx, col, first_page):
    """device encode of a (sliced) column; the per-page sampling seeds continue from `first_page` like the pages of the
    whole column"""
    import torch
    from strawboat_amd import write
    from strawboat_amd.types import WriteOptions

    def up(a):
        return None if a is None else torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(ctx.torch_device)
    dc = write.DeviceColumn(col["ptype"], col["nullable"], col["rows"], up(col["values"]), up(col["validity"]), up(col["offsets"]),
                            first_page_index=first_page, column_values_len=col.get("column_values_len", 0))
    # adaptive: the codec of every page is chosen on the device from seeded samples
    enc = write.write(ctx, dc, WriteOptions(max_page_size=PAGE, default_compress_ratio=2.0))
    return enc.pages_numpy(), enc.metas_array()


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    import strawboat_amd as sb
    from strawboat_amd import shard
    from workloads import arrow_bytes
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cols = _columns()
        ctx = sb.Context(0)
        plan = shard.plan_work_items([(arrow_bytes(c), (c["rows"] + PAGE - 1) // PAGE) for c in cols], world)
        items, blobs = [], {}
        for it in plan[rank]:
            pages, metas = _encode(ctx, shard.slice_column(cols[it.column], it.first_page, it.n_pages, PAGE), it.first_page)
            items.append((it.column, it.first_page, metas))
            blobs[(it.column, it.first_page)] = pages.tobytes()
        allm = shard.gather_metas(items, len(cols), capacity=shard.record_capacity(plan))
        cm = shard.column_metas(allm)
        q.put((rank, [c.offset for c in cm], [m.tolist() for m in allm], blobs))
        ctx.close()
    finally:
        dist.destroy_process_group()


def test_two_ranks_device_encoder(gpu_ctx):
    import torch.multiprocessing as mp
    from oracle import sbo as S
    from strawboat_amd import shard
    from tests import gen
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = 29700 + (os.getpid() % 2000)
    procs = [mpc.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, off0, m0, b0), (_, off1, m1, b1) = res
    assert off0 == off1 and m0 == m1
    cols = _columns()
    single = [_encode(gpu_ctx, c, 0) for c in cols]
    assert [m.tolist() for _, m in single] == m0, "gathered page metas differ from a single-process run"
    assert [c.offset for c in shard.column_metas([m for _, m in single])] == off0
    # the page bytes of the work items, put back in (column, page) order, are the single writer's bytes — and the oracle's
    blobs = dict(b0)
    blobs.update(b1)
    for ci, (pages, metas) in enumerate(single):
        got = b"".join(blobs[k] for k in sorted(k for k in blobs if k[0] == ci))
        assert got == pages.tobytes(), "column %d" % ci
        want, _ = gen.oracle_write(cols[ci], max_page_size=PAGE, ratio=2.0)
        assert got == want.tobytes(), "column %d vs oracle" % ci


@pytest.mark.parametrize("dc_name", ["LZ4", "ZSTD"])
def test_nested_page_range_items_write_the_single_writers_pages(gpu_ctx, dc_name):
    """(leaf, page range) work items through the NESTED API with an ADAPTIVE ratio: a rank that owns top-level rows
    [r0, r1) of a C5 leaf column (workloads.c5_slice) writes, page for page, the bytes a single writer produces for the
    whole column — the per-page sampling seed continues from first_page_index and the binary selector's total_bytes /
    the hdr9 of Dict / Freq / OneValue pages use the whole column's values length (write_nested_leaves forwards both)."""
    sys.path.insert(0, ROOT)
    import torch
    import workloads as W
    from strawboat_amd import nested, write
    from strawboat_amd.types import Compression, WriteOptions
    ctx = gpu_ctx
    page = 4096
    rows = 30_000
    la, a, lb, b = W.c5_nested(rows=rows, seed=11)
    opts = WriteOptions(max_page_size=page, default_compression=getattr(Compression, dc_name), default_compress_ratio=2.0)

    def up(x):
        return None if x is None else torch.from_numpy(np.ascontiguousarray(x).view(np.uint8).reshape(-1).copy()).to(ctx.torch_device)

    def item(levels, col, first_page):
        dl = [nested.NestedLevel(x["kind"], bool(x["is_optional"]), x["length"], up(x.get("validity")), up(x.get("offsets"))) for x in levels]
        dc = write.DeviceColumn(col["ptype"], False, col["rows"], up(col["values"]), up(col["validity"]), up(col["offsets"]),
                                first_page_index=first_page, column_values_len=col.get("column_values_len", 0))
        return dl, dc
    npages = (rows + page - 1) // page
    for levels, leaf in ((la, a), (lb, b)):
        whole = nested.write_nested_leaves(ctx, [item(levels, leaf, 0)], opts)[0]
        wm, wp = whole.metas_array(), whole.pages_numpy()
        assert len(wm) == npages
        ends = np.concatenate([[0], np.cumsum(wm[:, 0].astype(np.int64))])
        for p0, n in ((0, 3), (3, 1), (4, npages - 4)):
            r0, r1 = p0 * page, min(rows, (p0 + n) * page)
            lv, col = W.c5_slice(levels, leaf, r0, r1)
            part = nested.write_nested_leaves(ctx, [item(lv, col, p0)], opts)[0]
            assert np.array_equal(part.metas_array(), wm[p0:p0 + n]), "page sizes of pages %d..%d" % (p0, p0 + n)
            assert np.array_equal(part.pages_numpy(), wp[ends[p0]:ends[p0 + n]]), "page bytes of pages %d..%d" % (p0, p0 + n)
