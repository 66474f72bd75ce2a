"""Per-codec encode/decode timing probe on the C2 shape (development aid)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import strawboat_amd as sb
from strawboat_amd import read, write
from strawboat_amd.types import Compression as C, PhysicalType, WriteOptions

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ctx = sb.Context(0)
dev = ctx.torch_device
cols = []
for b in range(B):
    vals, valid = bench.gen_c2_column(42 + b)
    cols.append(write.DeviceColumn(PhysicalType.FLOAT64, True, bench.ROWS, torch.from_numpy(vals.view(np.uint8)).to(dev),
                                   torch.from_numpy(valid).to(dev)))
torch.cuda.synchronize()
U = B * (bench.ROWS * 8 + bench.ROWS // 8)

def run(name, opts, reps=5):
    enc = write.encode_columns(ctx, cols, opts); ctx.synchronize()
    pages = [read.ColumnPages(PhysicalType.FLOAT64, True, e.pages, e.metas_array()) for e in enc]
    dec = read.batch_read_columns(ctx, pages); ctx.synchronize()
    pb = sum(e.length for e in enc)
    wb, rb = write.WriteBatch(ctx, cols, opts, out=enc), read.ReadBatch(ctx, pages, out=dec)
    ctx.profile(True)
    for _ in range(reps):
        wb.enqueue()
    ctx.synchronize(); enc_stats = ctx.profile_read(); ctx.profile(True)
    for _ in range(reps):
        rb.enqueue()
    ctx.synchronize(); dec_stats = ctx.profile_read(); ctx.profile(False)
    te = sum(v[1] for v in enc_stats.values()) / reps
    td = sum(v[1] for v in dec_stats.values()) / reps
    top = lambda st: ", ".join("%s %.3f" % (k, v[1] / reps) for k, v in sorted(st.items(), key=lambda kv: -kv[1][1])[:3])
    print("%-22s pages %6.1f MB | enc %.3f ms (%6.0f GB/s) [%s] | dec %.3f ms (%6.0f GB/s) [%s]" %
          (name, pb / 1e6, te, U / te / 1e6, top(enc_stats), td, U / td / 1e6, top(dec_stats)))

P = 65536
run("None", WriteOptions(max_page_size=P))
run("RLE forced", WriteOptions(max_page_size=P, force_codec=C.RLE))
run("Dict idx None", WriteOptions(max_page_size=P, force_codec=C.DICT))
run("Dict idx RLE", WriteOptions(max_page_size=P, force_codec=C.DICT, force_index_codec=C.RLE))
run("adaptive r=2 (None)", WriteOptions(max_page_size=P, default_compress_ratio=2.0, forbidden_compressions=[C.FREQ, C.PATAS]))
run("LZ4 basic", WriteOptions(max_page_size=P, default_compression=C.LZ4), reps=2)
