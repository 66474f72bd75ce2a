"""Host-side enqueue time of the continuity bool batch (512 columns x 128 pages) before and after the context has run C5
(development): is a slower step afterwards the host's or the device's?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench, workloads as W
import strawboat_amd as sb
from strawboat_amd import read, write
from strawboat_amd.types import WriteOptions, Compression as C
ctx = sb.Context(0); h = bench.GpuHarness(ctx)
o = WriteOptions(max_page_size=8192, default_compression=C.LZ4)
cols = bench.gen_parallel(lambda s: W.cont_bool(1 << 20, s), range(42, 42 + 512))
dc = [h.dcol(c) for c in cols]

def once(tag):
    enc = write.encode_columns(ctx, dc, o); ctx.synchronize()
    pages = [read.ColumnPages(c["ptype"], c["nullable"], e.pages, e.metas_array()) for c, e in zip(cols, enc)]
    dec = read.batch_read_columns(ctx, pages); ctx.synchronize()
    wb, rb = write.WriteBatch(ctx, dc, o, out=enc), read.ReadBatch(ctx, pages, out=dec)
    for _ in range(2):
        wb.enqueue(); rb.enqueue()
    ctx.synchronize()
    for name, b in (("write", wb), ("read", rb)):
        t0 = time.perf_counter()
        for _ in range(10):
            b.enqueue()
        t1 = time.perf_counter()
        ctx.synchronize()
        t2 = time.perf_counter()
        print("%s %s: host enqueue %.3f ms / call, wall %.3f ms / call" % (tag, name, (t1 - t0) / 10 * 1e3, (t2 - t0) / 10 * 1e3), flush=True)

once("fresh")
bench.run_c5(h, False)
once("after c5")
once("after c5, again")
