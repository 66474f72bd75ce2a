"""N > 1 host path on CPU (gloo, world_size 2): columns are dealt to ranks, every rank encodes
its own columns (here with the CPU oracle standing in for the device encoder), and the only
collective is the all_gather of the page metas, from which every rank derives identical
ColumnMeta offsets — the same code path bench.py runs over RCCL."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from oracle import sbo as S
    from strawboat_amd import shard
    from tests import gen
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cols = [gen.prim(S.T_I32, 5000, seed=1), gen.prim(S.T_F64, 5000, null_density=0.1, seed=2),
                gen.binary(5000, seed=3), gen.boolean(5000, null_density=0.1, seed=4), gen.prim(S.T_I64, 300, seed=5)]
        sizes = [np.ascontiguousarray(c["values"]).nbytes for c in cols]
        mine = shard.plan_shards(sizes, world)[rank]
        local = {}
        for i in mine:
            _, metas = gen.oracle_write(cols[i], max_page_size=2048)
            local[i] = metas
        allm = shard.gather_metas(local, len(cols))
        cm = shard.column_metas(allm)
        # the same job as (column, page-range) work items and ONE fixed-capacity all_gather
        plan = shard.plan_work_items([(s, (c["rows"] + 2047) // 2048) for s, c in zip(sizes, cols)], world, pages_per_item=1)
        items = []
        for it in plan[rank]:
            part = shard.slice_column(cols[it.column], it.first_page, it.n_pages, 2048)
            # (the per-page sampling seed follows the page index inside the column)
            _, metas = gen.oracle_write(part, max_page_size=2048)
            items.append((it.column, it.first_page, metas))
        allm2 = shard.gather_metas(items, len(cols), capacity=shard.record_capacity(plan))
        q.put((rank, mine, [c.offset for c in cm], [m.tolist() for m in allm], [m.tolist() for m in allm2]))
    finally:
        dist.destroy_process_group()


def test_two_ranks_gather_metas():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    (r0, mine0, off0, m0, w0), (r1, mine1, off1, m1, w1) = res
    assert sorted(mine0 + mine1) == [0, 1, 2, 3, 4] and not set(mine0) & set(mine1)
    assert off0 == off1 and m0 == m1
    assert w0 == w1 == m0, "page-range work items must reproduce the whole-column page metas"
    # equal to what a single process computes
    sys.path.insert(0, ROOT)
    from oracle import sbo as S
    from strawboat_amd import shard
    from tests import gen
    cols = [gen.prim(S.T_I32, 5000, seed=1), gen.prim(S.T_F64, 5000, null_density=0.1, seed=2),
            gen.binary(5000, seed=3), gen.boolean(5000, null_density=0.1, seed=4), gen.prim(S.T_I64, 300, seed=5)]
    single = [gen.oracle_write(c, max_page_size=2048)[1] for c in cols]
    assert [m.tolist() for m in single] == m0
    assert [c.offset for c in shard.column_metas(single)] == off0


def test_plan_work_items_balances_and_covers():
    sys.path.insert(0, ROOT)
    from strawboat_amd import shard
    cols = [(40_000_000, 153), (40_000_000, 153), (81_250_000, 153), (81_250_000, 153), (190_000_000, 153),
            (190_000_000, 153), (2_500_000, 153), (2_500_000, 153)]
    for world in (1, 2, 4, 8):
        plan = shard.plan_work_items(cols, world)
        seen = {}
        for s in plan:
            for it in s:
                for p in range(it.first_page, it.first_page + it.n_pages):
                    assert (it.column, p) not in seen
                    seen[(it.column, p)] = 1
        assert len(seen) == sum(n for _, n in cols)
        loads = [sum(it.weight for it in s) for s in plan]
        assert max(loads) <= 1.05 * (sum(loads) / world) + 1, loads
        assert shard.record_capacity(plan) == max(sum(it.n_pages for it in s) for s in plan)
