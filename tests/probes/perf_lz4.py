"""LZ4 block encoder / decoder probe (development): time per block for a few byte patterns."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import strawboat_amd as sb
from strawboat_amd import read, write
from strawboat_amd.types import Compression as C, WriteOptions
from oracle import sbo as S

ctx = sb.Context(0)
dev = ctx.torch_device


def up(a):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)


def run(name, cols, opts, reps=3):
    dc = [write.DeviceColumn(c["ptype"], c["nullable"], c["rows"], up(c["values"]), up(c["validity"]), up(c["offsets"])) for c in cols]
    U = sum(np.asarray(c["values"]).nbytes for c in cols)
    enc = write.encode_columns(ctx, dc, opts); ctx.synchronize()
    pages = [read.ColumnPages(c["ptype"], c["nullable"], e.pages, e.metas_array()) for c, e in zip(cols, enc)]
    dec = read.batch_read_columns(ctx, pages); ctx.synchronize()
    pb = sum(e.length for e in enc)
    wb, rb = write.WriteBatch(ctx, dc, opts, out=enc), read.ReadBatch(ctx, pages, out=dec)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    with torch.cuda.stream(ctx.torch_stream):
        ev[0].record()
        for _ in range(reps):
            wb.enqueue()
        ev[1].record()
        for _ in range(reps):
            rb.enqueue()
        ev[2].record()
    ctx.synchronize()
    te, td = ev[0].elapsed_time(ev[1]) / reps, ev[1].elapsed_time(ev[2]) / reps
    npg = sum(len(e.metas_array()) for e in enc)
    print("%-52s %4d pages  in %7.1f MB out %7.1f MB | enc %8.3f ms %6.1f GB/s | dec %8.3f ms %6.1f GB/s" %
          (name, npg, U / 1e6, pb / 1e6, te, U / te / 1e6, td, U / td / 1e6), flush=True)


rng = np.random.default_rng(1)
def i32col(v):
    return dict(ptype=S.T_I32, nullable=False, rows=v.size, values=v.astype(np.int32), validity=None, offsets=None)

o = WriteOptions(max_page_size=65536, default_compression=C.LZ4)
for ncol in (16, 128, 1024):
    R = 65536
    run("i32 uniform [0,1000) x%d" % ncol, [i32col(rng.integers(0, 1000, R)) for _ in range(ncol)], o)
    run("i32 uniform full range x%d" % ncol, [i32col(rng.integers(-2**31, 2**31 - 1, R)) for _ in range(ncol)], o)
    run("i32 runs of ~32 x%d" % ncol, [i32col(np.repeat(rng.integers(0, 256, R // 16), rng.geometric(1 / 32, R // 16))[:R]) for _ in range(ncol)], o)
    run("i32 sorted offsets-like x%d" % ncol, [i32col(np.cumsum(rng.integers(4, 25, R))) for _ in range(ncol)], o)
    run("i32 constant x%d" % ncol, [i32col(np.full(R, 7)) for _ in range(ncol)], o)
