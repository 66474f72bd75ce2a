import sys, numpy as np
sys.path.insert(0, "/root/repo")
import bench, workloads as W
import strawboat_amd as sb
from strawboat_amd.types import WriteOptions
ctx = sb.Context(0); h = bench.GpuHarness(ctx)
rng = np.random.default_rng(7); n = 12_000_000
lowc = dict(ptype=W.T_I32, nullable=False, rows=n, values=rng.integers(0, 500, n).astype(np.int32), validity=None, offsets=None)
res = h.measure_flat([lowc], WriteOptions(default_compress_ratio=2.0), reps=3, check=1)
print(res["enc_ms"], res["dec_ms"])
