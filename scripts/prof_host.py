"""host time of one enqueue (table building + upload + launches) next to the GPU time per call, continuity shapes"""
import sys, time; sys.path.insert(0, '.')
import numpy as np, torch, bench, workloads as W
import strawboat_amd as sb
from strawboat_amd import read, write
from strawboat_amd.types import Compression as C, WriteOptions
ctx = sb.Context(0); h = bench.GpuHarness(ctx)
opts = WriteOptions(max_page_size=8192, default_compression=C.LZ4)
for name, gen, ncol in (("bool", W.cont_bool, 512), ("utf8", W.cont_utf8, 128), ("i64", W.cont_i64, 128)):
    col = gen(1 << 20)
    cols = [col] * ncol
    dc = [h.dcol(c) for c in cols]
    enc = write.encode_columns(ctx, dc, opts); ctx.synchronize()
    pages = [read.ColumnPages(c["ptype"], c["nullable"], e.pages, e.metas_array()) for c, e in zip(cols, enc)]
    dec = read.batch_read_columns(ctx, pages); ctx.synchronize()
    wb, rb = write.WriteBatch(ctx, dc, opts, out=enc), read.ReadBatch(ctx, pages, out=dec)
    for b, nm in ((wb, "write"), (rb, "read")):
        b.enqueue(); ctx.synchronize()
        t0 = time.perf_counter(); b.enqueue(); t1 = time.perf_counter(); ctx.synchronize(); t2 = time.perf_counter()
        print("%-5s %-5s enqueue (host) %.3f ms, enqueue + sync %.3f ms" % (name, nm, (t1 - t0) * 1e3, (t2 - t0) * 1e3))
