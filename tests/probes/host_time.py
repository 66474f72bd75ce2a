"""Host-side cost of one bench step (development): how long the two enqueue calls take on the CPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
import strawboat_amd as sb
from strawboat_amd import read, write
from strawboat_amd.types import PhysicalType as PT, WriteOptions
ctx = sb.Context(0); dev = ctx.torch_device
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
vals, valid = bench.gen_c2_column(42)
cols = [write.DeviceColumn(PT.FLOAT64, True, bench.ROWS, torch.from_numpy(vals.view(np.uint8)).to(dev), torch.from_numpy(valid).to(dev)) for _ in range(B)]
opts = WriteOptions(max_page_size=65536, default_compress_ratio=2.0)
enc = write.encode_columns(ctx, cols, opts); ctx.synchronize()
pages = [read.ColumnPages(PT.FLOAT64, True, e.pages, e.metas_array()) for e in enc]
dec = read.batch_read_columns(ctx, pages); ctx.synchronize()
wb, rb = write.WriteBatch(ctx, cols, opts, out=enc), read.ReadBatch(ctx, pages, out=dec)
for _ in range(3):
    wb.enqueue(); rb.enqueue()
ctx.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    wb.enqueue()
t1 = time.perf_counter()
for _ in range(20):
    rb.enqueue()
t2 = time.perf_counter()
ctx.synchronize()
t3 = time.perf_counter()
print("columns %d: host enqueue encode %.3f ms, decode %.3f ms per step; GPU drained %.3f ms after the last enqueue; wall per step %.3f ms" %
      (B, (t1 - t0) / 20 * 1e3, (t2 - t1) / 20 * 1e3, (t3 - t2) * 1e3, (t3 - t0) / 20 * 1e3))
