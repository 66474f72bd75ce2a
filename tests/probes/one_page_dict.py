import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
import bench, workloads as W
import strawboat_amd as sb
from strawboat_amd.types import WriteOptions, Compression as C
ctx = sb.Context(0); h = bench.GpuHarness(ctx)
rng = np.random.default_rng(7); n = 12_000_000
lowc = dict(ptype=W.T_I32, nullable=False, rows=n, values=rng.integers(0, 500, n).astype(np.int32), validity=None, offsets=None)
sp = np.full(n, 1_000_000, dtype=np.int32)
spi = rng.random(n) < 0.02
sp[spi] = rng.integers(0, 1 << 30, int(spi.sum())).astype(np.int32)
sparse = dict(lowc, values=sp)
for nm, col in (("lowcard", lowc), ("sparse", sparse)):
    for o in (WriteOptions(default_compress_ratio=2.0), WriteOptions(default_compress_ratio=2.0, default_compression=C.LZ4)):
        res = h.measure_flat([col], o, reps=3, check=1)
        ks = sorted(res["kernels"].items(), key=lambda kv: -kv[1][1])[:14]
        print(nm, "enc %.3f ms dec %.3f ms" % (res["enc_ms"], res["dec_ms"]), [(k, round(v[1], 3)) for k, v in ks], flush=True)
