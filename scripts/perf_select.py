import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import strawboat_amd as sb
from strawboat_amd import write
from strawboat_amd.types import Compression as C, PhysicalType, WriteOptions
B = 64
ctx = sb.Context(0); dev = ctx.torch_device
cols = []
for b in range(B):
    vals, valid = bench.gen_c2_column(42 + b)
    cols.append(write.DeviceColumn(PhysicalType.FLOAT64, True, bench.ROWS, torch.from_numpy(vals.view(np.uint8)).to(dev), torch.from_numpy(valid).to(dev)))
torch.cuda.synchronize()
def run(name, forb):
    opts = WriteOptions(max_page_size=65536, default_compress_ratio=2.0, forbidden_compressions=forb)
    enc = write.encode_columns(ctx, cols, opts); ctx.synchronize()
    wb = write.WriteBatch(ctx, cols, opts, out=enc)
    ctx.profile(True)
    for _ in range(5): wb.enqueue()
    ctx.synchronize(); st = ctx.profile_read(); ctx.profile(False)
    print("%-40s select %.3f ms" % (name, st["k_enc_select"][1] / st["k_enc_select"][0]))
run("forbid Freq,Patas", [C.FREQ, C.PATAS])
run("forbid Freq,Patas,Dict", [C.FREQ, C.PATAS, C.DICT])
run("forbid Freq,Patas,Dict,RLE", [C.FREQ, C.PATAS, C.DICT, C.RLE])
run("forbid Freq,Patas,RLE", [C.FREQ, C.PATAS, C.RLE])
run("forbid Patas", [C.PATAS])
run("forbid none", [])
