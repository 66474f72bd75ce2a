import sys, numpy as np
sys.path.insert(0, "/root/repo")
import bench, workloads as W
import strawboat_amd as sb
from strawboat_amd.types import WriteOptions, Compression as C
ctx = sb.Context(0); h = bench.GpuHarness(ctx)
utf8 = W.zipf_utf8(3_000_000, 42)
for o in (WriteOptions(default_compress_ratio=2.0), WriteOptions(default_compress_ratio=2.0, default_compression=C.LZ4)):
    res = h.measure_flat([utf8], o, reps=3, check=1)
    ks = sorted(res["kernels"].items(), key=lambda kv: -kv[1][1])[:16]
    print("utf8 enc %.3f ms dec %.3f ms" % (res["enc_ms"], res["dec_ms"]), [(k, round(v[1], 3)) for k, v in ks], flush=True)
