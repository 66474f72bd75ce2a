"""Where the host time of the C5 nested calls goes (cProfile over one batched write and one batched read of 64 arrays):
python scripts/prof_c5_host.py [arrays]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from strawboat_amd import nested
import workloads as W
from strawboat_amd.read import ColumnPages
from strawboat_amd.types import Compression as C, WriteOptions


def main():
    arrays = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    import strawboat_amd as sb
    ctx = sb.Context(0)
    h = B.GpuHarness(ctx)
    opts = WriteOptions(max_page_size=B.PAGE, default_compression=C.ZSTD)

    def dlevels(levels):
        return [nested.NestedLevel(lv["kind"], bool(lv["is_optional"]), lv["length"], h.up(lv.get("validity")), h.up(lv.get("offsets")))
                for lv in levels]
    items = []
    for la, a, lb, b in B.gen_parallel(lambda s: W.c5_nested(seed=s), range(42, 42 + arrays)):
        for lv_, leaf in ((la, a), (lb, b)):
            dc = h.dcol(leaf)
            dc.is_nullable = False
            items.append((dlevels(lv_), dc, leaf, lv_))
    pairs = [(dl, dc) for dl, dc, _, _ in items]
    encs = nested.write_nested_leaves(ctx, pairs, opts)
    cps = [ColumnPages(c["ptype"], False, e.pages[:e.length].contiguous(), e.metas_array()) for e, (_, _, c, _) in zip(encs, items)]
    kinds = [[lv["kind"] for lv in lv_] for _, _, _, lv_ in items]
    opt = [[bool(lv["is_optional"]) for lv in lv_] for _, _, _, lv_ in items]
    nested.read_nested_leaves(ctx, cps, kinds, opt)
    for name, fn in (("write", lambda: nested.write_nested_leaves(ctx, pairs, opts)), ("read", lambda: nested.read_nested_leaves(ctx, cps, kinds, opt))):
        t0 = time.perf_counter()
        fn()
        print("%s: %.2f ms wall" % (name, (time.perf_counter() - t0) * 1e3))
        pr = cProfile.Profile()
        pr.enable()
        fn()
        pr.disable()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(14)


if __name__ == "__main__":
    main()
