"""Quick decode throughput probe (development aid, not the contract bench)."""
import sys, time
import numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import sbo as S
from tests import gen
import strawboat_amd as sb
from strawboat_amd import read

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ROWS = 1_000_000
ctx = sb.Context(0)
dev = ctx.torch_device

def bench(name, col, reps=20, **opt):
    pages, metas = gen.oracle_write(col, **opt)
    want = gen.oracle_read(col, pages, metas)
    dpages = [torch.from_numpy(pages).to(dev) for _ in range(B)]   # B distinct copies in HBM
    cols = [read.ColumnPages(col["ptype"], col["nullable"], p, metas) for p in dpages]
    torch.cuda.synchronize()
    out = read.batch_read_columns(ctx, cols)
    ctx.synchronize()
    assert np.array_equal(out[-1].values_numpy(), want["values"])
    for _ in range(3):
        read.batch_read_columns(ctx, cols, out=out); 
    ctx.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(ctx.torch_stream):
        e0.record()
        for _ in range(reps):
            read.batch_read_columns(ctx, cols, out=out)
        e1.record()
    ctx.synchronize()
    ms = e0.elapsed_time(e1) / reps
    w = S.WIDTH.get(col["ptype"], 1)
    U = col["rows"] * w + (col["rows"] + 7) // 8 * (1 if col["nullable"] else 0)
    A = len(pages) + U
    print("%-28s pages=%9d B  U=%9d B  %.3f ms/batch  decode %.1f GB/s (Arrow bytes)  algorithmic %.1f GB/s (%.1f%% of 8 TB/s)"
          % (name, len(pages), U, ms, B * U / ms / 1e6, B * A / ms / 1e6, B * A / ms / 1e6 / 80))

col = dict(ptype=S.T_I64, nullable=False, rows=ROWS, values=np.random.default_rng(42).integers(0, 1 << 62, ROWS).astype(np.int64), validity=None, offsets=None)
bench("C1 i64 None 1 page", col)
bench("i64 None 64K pages", col, max_page_size=65536)
c2 = gen.prim(S.T_F64, ROWS, uniq=256, null_density=0.1, runs=32)
bench("C2 f64 None", c2, max_page_size=65536)
bench("C2 f64 RLE", c2, max_page_size=65536, force_codec=S.RLE)
bench("C2 f64 Dict(idx None)", c2, max_page_size=65536, force_codec=S.DICT)
bench("C2 f64 Dict(idx RLE)", c2, max_page_size=65536, force_codec=S.DICT, force_index_codec=S.RLE)
c2f = {k: v for k, v in c2.items()}; c2f["rows"] = 15 * 65536; c2f["values"] = c2["values"][:15 * 65536]; c2f["validity"] = c2["validity"][:15 * 65536 // 8]
bench("C2 f64 Dict(idx BP) 15pg", c2f, max_page_size=65536, force_codec=S.DICT, force_index_codec=S.BITPACK)
i32 = gen.prim(S.T_I32, 128 * 7808, uniq=1000)
bench("i32 BP", i32, max_page_size=65536, force_codec=S.BITPACK)
i32s = gen.prim(S.T_I32, 128 * 7808, uniq=1 << 20, sorted_=True)
bench("i32 DeltaBP", i32s, max_page_size=65536, force_codec=S.DELTABP)
bench("C2 f64 LZ4", c2, reps=3, max_page_size=65536, default_compression=S.LZ4)
bench("C2 f64 Zstd (libzstd level 3)", c2, reps=3, max_page_size=65536, default_compression=S.ZSTD)
bench("C2 f64 Snappy", c2, reps=3, max_page_size=65536, default_compression=S.SNAPPY)
rnd = gen.prim(S.T_I64, ROWS, uniq=1 << 40)
bench("random i64 Zstd", rnd, reps=3, max_page_size=65536, default_compression=S.ZSTD)
txt = gen.binary(ROWS, uniq=5000, zipf=1.3, maxlen=24)
bench("utf8 zipf Zstd", txt, reps=3, max_page_size=65536, default_compression=S.ZSTD)
bench("utf8 zipf LZ4", txt, reps=3, max_page_size=65536, default_compression=S.LZ4)
