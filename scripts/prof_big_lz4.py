"""one 96 MB LZ4 block (12 M sorted Int64 rows, one page): wall time of write and read with the default (parallel) encoder"""
import sys, time; sys.path.insert(0, '.')
import numpy as np, torch, bench
import strawboat_amd as sb
from strawboat_amd import read, write
from strawboat_amd.types import Compression as C, WriteOptions
from oracle import sbo as S
ctx = sb.Context(0); h = bench.GpuHarness(ctx)
rng = np.random.default_rng(1)
v = np.sort(rng.integers(0, 1 << 40, 12_000_000)).astype(np.int64)
col = dict(ptype=S.T_I64, nullable=False, rows=v.size, values=v, validity=None, offsets=None)
dc = h.dcol(col)
for codec, name in ((C.LZ4, "LZ4"), (C.ZSTD, "Zstd"), (C.SNAPPY, "Snappy")):
    opts = WriteOptions(default_compression=codec)
    for rep in range(2):
        t0 = time.perf_counter(); enc = write.encode_columns(ctx, [dc], opts); ctx.synchronize(); te = time.perf_counter() - t0
    pages = [read.ColumnPages(col["ptype"], False, enc[0].pages, enc[0].metas_array())]
    t0 = time.perf_counter(); dec = read.batch_read_columns(ctx, pages); ctx.synchronize(); td = time.perf_counter() - t0
    assert np.array_equal(dec[0].values_numpy().view(np.int64), v)
    print("%-6s one page of %d MB: write %.1f ms, read %.1f ms, page %.1f MB" % (name, v.nbytes >> 20, te * 1e3, td * 1e3, enc[0].length / 1e6), flush=True)
