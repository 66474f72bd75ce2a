"""Development probe: where the wall time of NestedWriteBatch.run() / NestedReadBatch.run() goes on the host (C5 x 64)."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench as B
import workloads as W


def main():
    import torch
    import strawboat_amd as sb
    from strawboat_amd import nested
    from strawboat_amd.read import ColumnPages
    from strawboat_amd.types import Compression as C, WriteOptions
    arrays = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    ctx = sb.Context(0)
    h = B.GpuHarness(ctx)
    opts = WriteOptions(max_page_size=B.PAGE, default_compression=C.ZSTD)
    gen = B.gen_parallel(lambda s: W.c5_nested(seed=s), range(42, 42 + arrays))
    items = []
    for la, a, lb, b in gen:
        for lv_, leaf in ((la, a), (lb, b)):
            dc = h.dcol(leaf)
            dc.is_nullable = False
            items.append(([nested.NestedLevel(lv["kind"], bool(lv["is_optional"]), lv["length"], h.up(lv.get("validity")), h.up(lv.get("offsets"))) for lv in lv_], dc, leaf, lv_))
    wb = nested.NestedWriteBatch(ctx, [(dl, dc) for dl, dc, _, _ in items], opts)
    encs = wb.run()
    cps = [ColumnPages(c["ptype"], False, e.pages[:e.length].contiguous(), e.metas_array()) for e, (_, _, c, _) in zip(encs, items)]
    rb = nested.NestedReadBatch(ctx, cps, [[lv["kind"] for lv in lv_] for _, _, _, lv_ in items], [[bool(lv["is_optional"]) for lv in lv_] for _, _, _, lv_ in items])
    rb.run()
    for name, fn in (("write", wb.run), ("read", rb.run)):
        for _ in range(2):
            fn()
        t0 = time.perf_counter()
        for _ in range(5):
            fn()
        print("%s: %.3f ms per run" % (name, (time.perf_counter() - t0) / 5 * 1e3))
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(5):
            fn()
        pr.disable()
        st = pstats.Stats(pr)
        st.sort_stats("tottime").print_stats(8)
        ctx.profile(True)
        fn()
        prof = ctx.profile_read()
        ctx.profile(False)
        tot = sum(v[1] for v in prof.values())
        print("%s kernels: %.3f ms in %d launches" % (name, tot, sum(v[0] for v in prof.values())))
        for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])[:30]:
            print("   %-34s %3d  %.3f ms" % (k, v[0], v[1]))
        print("replays", ctx.replays())


if __name__ == "__main__":
    main()
