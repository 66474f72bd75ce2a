"""Encode / decode throughput of the BASELINE.json configurations other than the bench default (C2),
one GPU, inputs resident in HBM (development probe; numbers quoted in DESIGN.md)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
import strawboat_amd as sb
from strawboat_amd import read, write
from strawboat_amd.types import Compression as C, PhysicalType as PT, WriteOptions
from tests.test_gpu_configs import zipf_utf8
from tests import gen
from oracle import sbo as S

ctx = sb.Context(0)
dev = ctx.torch_device
ROWS = 1_000_000


def up(a):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)


def dcol(col):
    return write.DeviceColumn(col["ptype"], col["nullable"], col["rows"], up(col["values"]), up(col["validity"]), up(col["offsets"]))


def arrow_bytes(col):
    n = col["rows"]
    b = np.asarray(col["values"]).nbytes if col["ptype"] != S.T_BOOL else (n + 7) // 8
    if col["nullable"]:
        b += (n + 7) // 8
    if col["offsets"] is not None:
        b += np.asarray(col["offsets"]).nbytes
    return b


def run(name, cols, opts, reps=5):
    dc = [dcol(c) for c in cols]
    U = sum(arrow_bytes(c) for c in cols)
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
    ctx.profile(True)
    wb.enqueue(); rb.enqueue(); ctx.synchronize()
    st = ctx.profile_read(); ctx.profile(False)
    top = ", ".join("%s %.2f" % (k, v[1]) for k, v in sorted(st.items(), key=lambda kv: -kv[1][1])[:4])
    te, td = ev[0].elapsed_time(ev[1]) / reps, ev[1].elapsed_time(ev[2]) / reps
    print("%-44s Arrow %7.1f MB  pages %7.1f MB | encode %8.3f ms %7.1f GB/s | decode %8.3f ms %7.1f GB/s" %
          (name, U / 1e6, pb / 1e6, te, U / te / 1e6, td, U / td / 1e6))
    print("      kernels (ms): " + top)
    codecs = S.stat_column(cols[0]["ptype"], cols[0]["nullable"], enc[0].pages_numpy(), enc[0].metas_array())
    print("      codecs of column 0: page %s nested %s" % (sorted(set(codecs[0].tolist())), sorted(set(codecs[1].tolist()))))


rng = np.random.default_rng(42)
B = 64
c1 = [dict(ptype=S.T_I64, nullable=False, rows=ROWS, values=rng.integers(0, 2**63 - 1, ROWS), validity=None, offsets=None) for _ in range(B)]
run("C1  64 x 1M Int64, one page, None", c1, WriteOptions())
c3 = [zipf_utf8(ROWS, 42 + i) for i in range(64)]
run("C3  64 x 1M Utf8 zipf, LZ4 + ratio 2 (Dict)", c3, WriteOptions(max_page_size=65536, default_compression=C.LZ4, default_compress_ratio=2.0))
run("C3' 64 x 1M Utf8 zipf, Basic(LZ4) forced", c3, WriteOptions(max_page_size=65536, default_compression=C.LZ4), reps=2)
i32 = [dict(ptype=S.T_I32, nullable=True, rows=ROWS, values=rng.integers(0, 1000, ROWS).astype(np.int32), validity=None, offsets=None) for _ in range(128)]
run("C4 128 x 1M Int32 [0,1000), LZ4 + ratio 2", i32, WriteOptions(max_page_size=65536, default_compression=C.LZ4, default_compress_ratio=2.0))
bl = [gen.boolean(ROWS, null_density=0.1, seed=45 + i) for i in range(128)]
run("C4 128 x 1M Boolean 10% null, LZ4 + ratio 2", bl, WriteOptions(max_page_size=65536, default_compression=C.LZ4, default_compress_ratio=2.0), reps=2)
