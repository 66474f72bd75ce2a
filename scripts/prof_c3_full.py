import sys; sys.path.insert(0, '.')
import numpy as np, torch, bench, workloads as W
import strawboat_amd as sb
from strawboat_amd.types import Compression as C, WriteOptions
ctx = sb.Context(0); h = bench.GpuHarness(ctx)
cols = bench.gen_parallel(lambda s: W.zipf_utf8(1 << 20, s), range(42, 42 + 64))
res = h.measure_flat(cols, WriteOptions(max_page_size=65536, default_compression=C.LZ4, default_compress_ratio=2.0), reps=3)
print("enc ms", res["enc_ms"], "dec ms", res["dec_ms"])
for k, v in sorted(res["kernels"].items(), key=lambda kv: -kv[1][1]): print("  %-32s %3d %9.3f ms" % (k, v[0], v[1]))
