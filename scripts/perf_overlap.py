"""Does running encode and decode of consecutive batches on two contexts (two HIP streams) fill the
bubbles of the one-stream step?  (development probe for bench.py --streams 2)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import strawboat_amd as sb
from strawboat_amd import read, write
from strawboat_amd.types import PhysicalType, WriteOptions

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = 20
dev = torch.device("cuda", 0)
ce, cd = sb.Context(0, stream=torch.cuda.Stream(dev)), sb.Context(0, stream=torch.cuda.Stream(dev))
opts = WriteOptions(max_page_size=65536, default_compress_ratio=2.0)
cols = []
for b in range(B):
    vals, valid = bench.gen_c2_column(42 + b)
    cols.append(write.DeviceColumn(PhysicalType.FLOAT64, True, bench.ROWS, torch.from_numpy(vals.view(np.uint8)).to(dev),
                                   torch.from_numpy(valid).to(dev)))
torch.cuda.synchronize()
U = B * (bench.ROWS * 8 + (bench.ROWS + 7) // 8)
sets = []
for k in range(2):
    enc = write.encode_columns(ce, cols, opts); ce.synchronize()
    pages = [read.ColumnPages(PhysicalType.FLOAT64, True, e.pages, e.metas_array()) for e in enc]
    dec = read.batch_read_columns(cd, pages); cd.synchronize()
    sets.append((write.WriteBatch(ce, cols, opts, out=enc), read.ReadBatch(cd, pages, out=dec),
                 torch.cuda.Event(), torch.cuda.Event()))

def run(overlap):
    for _ in range(3):
        for wb, rb, _, _ in sets:
            wb.enqueue(); ce.synchronize(); rb.enqueue(); cd.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        wb, rb, e_done, d_done = sets[i % 2]
        if overlap:
            if i >= 2:
                ce.torch_stream.wait_event(d_done)      # the decode two steps back has finished reading these pages
            wb.enqueue()
            e_done.record(ce.torch_stream)
            cd.torch_stream.wait_event(e_done)
            rb.enqueue()
            d_done.record(cd.torch_stream)
        else:
            wb.enqueue()
            e_done.record(ce.torch_stream)
            cd.torch_stream.wait_event(e_done)
            rb.enqueue()
            d_done.record(cd.torch_stream)
            ce.torch_stream.wait_event(d_done)          # strictly one after the other
    ce.synchronize(); cd.synchronize()
    dt = time.perf_counter() - t0
    print("%-28s %.3f ms/step  %.0f GB/s" % ("two streams, overlapped" if overlap else "two streams, serialized", dt / steps * 1e3, 2 * U * steps / dt / 1e9))

run(False)
run(True)
run(False)
run(True)
