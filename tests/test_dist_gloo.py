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
        q.put((rank, mine, [c.offset for c in cm], [m.tolist() for m in allm]))
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
    (r0, mine0, off0, m0), (r1, mine1, off1, m1) = res
    assert sorted(mine0 + mine1) == [0, 1, 2, 3, 4] and not set(mine0) & set(mine1)
    assert off0 == off1 and m0 == m1
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
