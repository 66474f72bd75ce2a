"""where the wall time of the C5 nested calls goes (levels per leaf, the flat call, Python)"""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench, workloads as W
import strawboat_amd as sb
from strawboat_amd import nested, write
from strawboat_amd.read import ColumnPages
from strawboat_amd.types import Compression as C, WriteOptions
ctx = sb.Context(0); h = bench.GpuHarness(ctx)
la, a, lb, b = W.c5_nested()
opts = WriteOptions(max_page_size=65536, default_compression=C.ZSTD)
def dlevels(levels):
    return [nested.NestedLevel(lv["kind"], bool(lv["is_optional"]), lv["length"], h.up(lv.get("validity")), h.up(lv.get("offsets"))) for lv in levels]
items = [(dlevels(la), h.dcol(a)), (dlevels(lb), h.dcol(b))]
for _, dc in items: dc.is_nullable = False
encs = nested.write_nested_leaves(ctx, items, opts)
T = {}
def tick(name, t0): T[name] = T.get(name, 0.0) + (time.perf_counter() - t0)
for rep in range(5):
    for levels, leaf in items:
        t0 = time.perf_counter(); lv = nested.write_levels(ctx, levels, levels[0].length, 65536); tick("write_levels x2", t0)
    t0 = time.perf_counter(); encs = nested.write_nested_leaves(ctx, items, opts); tick("write_nested_leaves (all)", t0)
cps = [ColumnPages(c["ptype"], False, e.pages[:e.length].contiguous(), e.metas_array()) for e, c in zip(encs, (a, b))]
kinds = [[lv["kind"] for lv in l] for l in (la, lb)]; opt = [[bool(lv["is_optional"]) for lv in l] for l in (la, lb)]
arrs = nested.read_nested_leaves(ctx, cps, kinds, opt)
for rep in range(5):
    t0 = time.perf_counter(); arrs = nested.read_nested_leaves(ctx, cps, kinds, opt); tick("read_nested_leaves (all)", t0)
for k, v in T.items(): print("%-28s %.3f ms per call" % (k, v / 5 * 1e3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(5): nested.read_nested_leaves(ctx, cps, kinds, opt)
pr.disable(); pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
