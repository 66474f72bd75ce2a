import sys, numpy as np
sys.path.insert(0, "/root/repo")
import bench, workloads as W
import strawboat_amd as sb
from strawboat_amd.types import WriteOptions, Compression as C
ctx = sb.Context(0); h = bench.GpuHarness(ctx)
rng = np.random.default_rng(7); n = 12_000_000
i64 = dict(ptype=W.T_I64, nullable=False, rows=n, values=np.sort(rng.integers(0, 1 << 40, n)).astype(np.int64), validity=None, offsets=None)
utf8 = W.zipf_utf8(3_000_000, 42)
for nm, col in (("i64", i64), ("utf8", utf8)):
    res = h.measure_flat([col], WriteOptions(default_compression=C.LZ4), reps=3, check=1)
    ks = sorted(res["kernels"].items(), key=lambda kv: -kv[1][1])[:10]
    print(nm, "lz4 enc %.3f ms dec %.3f ms (%.1f GB/s)" % (res["enc_ms"], res["dec_ms"], res["U"] / res["dec_ms"] / 1e6), [(k, round(v[1], 3)) for k, v in ks], flush=True)
