import sys, numpy as np
sys.path.insert(0, "/root/repo")
import bench, workloads as W
import strawboat_amd as sb
from strawboat_amd.types import WriteOptions
ctx = sb.Context(0); h = bench.GpuHarness(ctx)
rng = np.random.default_rng(7); n = 12_000_000
i64 = dict(ptype=W.T_I64, nullable=False, rows=n, values=np.sort(rng.integers(0, 1 << 40, n)).astype(np.int64), validity=None, offsets=None)
runs = dict(i64, values=np.repeat(rng.integers(0, 200, n // 50 + 1), 50)[:n].astype(np.int64))
for nm, col in (("sorted", i64), ("runs", runs)):
    res = h.measure_flat([col], WriteOptions(default_compress_ratio=2.0), reps=3, check=1)
    ks = sorted(res["kernels"].items(), key=lambda kv: -kv[1][1])[:8]
    print(nm, "enc %.3f ms dec %.3f ms" % (res["enc_ms"], res["dec_ms"]), [(k, round(v[1], 3)) for k, v in ks], flush=True)
