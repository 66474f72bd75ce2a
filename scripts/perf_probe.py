"""Quick throughput probe of one column shape at a chosen batch size (GPU box):
    python scripts/perf_probe.py int32 32 10000000 [lz4|zstd|none] [ratio|-]
prints encode / decode ms, GB/s of Arrow bytes and the per-kernel HIP-event times of one pass."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import workloads as W  # noqa: E402


def column(kind, rows, seed):
    rng = np.random.default_rng(seed)
    if kind == "int32":
        return dict(ptype=W.T_I32, nullable=True, rows=rows, values=rng.integers(0, 1000, rows).astype(np.int32), validity=None, offsets=None)
    if kind == "f64":
        return W.c2_float64(seed, rows)
    if kind == "utf8":
        return W.zipf_utf8(rows, seed, null_density=0.1)
    if kind == "utf8nn":
        return W.zipf_utf8(rows, seed)
    if kind == "bool":
        return dict(ptype=W.T_BOOL, nullable=True, rows=rows, values=W.pack_bits(rng.random(rows) < 0.5),
                    validity=W.pack_bits(rng.random(rows) >= 0.1), offsets=None)
    if kind == "offs":   # what the offsets of short strings look like: an increasing Int32 column, steps of 2..4
        return dict(ptype=W.T_I32, nullable=False, rows=rows, values=np.cumsum(rng.integers(2, 5, rows)).astype(np.int32), validity=None, offsets=None)
    if kind == "i64":
        return W.c1_int64(seed) if rows == 1_000_000 else dict(ptype=W.T_I64, nullable=False, rows=rows, values=rng.integers(0, 2**63 - 1, rows), validity=None, offsets=None)
    raise SystemExit("unknown kind " + kind)


def main():
    import torch  # noqa: F401
    import strawboat_amd as sb
    from strawboat_amd.types import Compression as C, WriteOptions
    kind, ncols, rows = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    dc = {"lz4": C.LZ4, "zstd": C.ZSTD, "none": C.NONE, "snappy": C.SNAPPY}[sys.argv[4] if len(sys.argv) > 4 else "lz4"]
    ratio = None if len(sys.argv) > 5 and sys.argv[5] == "-" else 2.0
    page = int(sys.argv[6]) if len(sys.argv) > 6 else 65536
    ctx = sb.Context(0)
    h = bench.GpuHarness(ctx)
    if kind == "c4":
        cols = [c for _, c in W.c4_columns(rows)] * ncols
    else:
        cols = bench.gen_parallel(lambda s: column(kind, rows, s), range(42, 42 + ncols))
    opts = WriteOptions(max_page_size=page or None, default_compression=dc, default_compress_ratio=ratio)
    r = h.measure_flat(cols, opts, reps=3)
    U = r["U"]
    print("%s x %d cols x %d rows: %d pages, %.1f MB arrow, %.1f MB pages" % (kind, ncols, rows, r["n_pages"], U / 1e6, r["page_bytes"] / 1e6))
    print("encode %.3f ms = %.1f GB/s   decode %.3f ms = %.1f GB/s" % (r["enc_ms"], U / r["enc_ms"] / 1e6, r["dec_ms"], U / r["dec_ms"] / 1e6))
    for k, v in sorted(r["kernels"].items(), key=lambda kv: -kv[1][1]):
        if v[1] >= 0.02:
            print("  %-34s x%d  %.3f ms" % (k, v[0], v[1]))
    try:
        print("codecs:", bench.page_codecs(cols[0], r["enc"][0]))
    except Exception as e:
        print("codecs: n/a", e)


if __name__ == "__main__":
    main()
