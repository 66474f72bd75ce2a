"""Device time of one-page columns (development probe): encode / decode of a single multi-million-row page with the
default (parallel) LZ4 encoder, columns resident in HBM, per-kernel HIP-event times of the slow ones.
    python tests/probes/one_page_time.py [rows] [case substring]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import strawboat_amd as sb
from strawboat_amd import write, read, WriteOptions
from oracle import sbo as S
from tests import gen
from tests.test_gpu_encode import to_device_column
from tests.test_gpu_freq import sparse

ctx = sb.Context(0)
ROWS = int(sys.argv[1]) if len(sys.argv) > 1 else 12_000_000
ONLY = sys.argv[2] if len(sys.argv) > 2 else ""
rng = np.random.default_rng(1)


def col(ptype, v):
    return dict(ptype=ptype, nullable=False, rows=v.size, values=v, validity=None, offsets=None)


CASES = [("random u32", col(S.T_U32, rng.integers(0, 1 << 30, ROWS).astype(np.uint32))),
         ("sorted i64", col(S.T_I64, np.sort(rng.integers(0, 1 << 40, ROWS)).astype(np.int64))),
         ("low-card i32", col(S.T_I32, rng.integers(0, 500, ROWS).astype(np.int32))),
         ("runs i64", col(S.T_I64, np.repeat(rng.integers(0, 200, ROWS // 50 + 1), 50)[:ROWS].astype(np.int64))),
         ("sparse i64", sparse(S.T_I64, ROWS, 0.02, 4)),
         ("utf8 zipf", gen.binary(ROWS // 4, uniq=5000, zipf=1.2, maxlen=24)),
         ("utf8 unique", gen.binary(ROWS // 4, uniq=ROWS // 4, maxlen=24)),
         ("bool", gen.boolean(ROWS, null_density=0.0, runs=3))]
OPTS = [("adaptive", dict(default_compress_ratio=2.0)), ("adaptive+lz4", dict(default_compress_ratio=2.0, default_compression=S.LZ4)),
        ("lz4", dict(default_compression=S.LZ4)), ("zstd", dict(default_compression=S.ZSTD))]


def top(prof, k=4):
    items = sorted(prof.items(), key=lambda kv: -kv[1][1])[:k]
    return " ".join("%s=%.2f" % (n, ms) for n, (c, ms) in items if ms > 0.05)


for name, c in CASES:
    if ONLY and ONLY not in name:
        continue
    dc = to_device_column(ctx, c)
    nbytes = c["values"].nbytes + (c["offsets"].nbytes if c["offsets"] is not None else 0)
    for oname, opt in OPTS:
        wo = WriteOptions(max_page_size=None, **opt)
        for rep in range(2):
            ctx.profile(rep == 1)
            torch.cuda.synchronize(); t = time.time()
            enc = write.write(ctx, dc, wo); ctx.synchronize()
            te = time.time() - t
        pe = ctx.profile_read()
        pages, metas = enc.pages[:enc.length], enc.metas_array()
        cp = read.ColumnPages(c["ptype"], c["nullable"], pages, metas)
        for rep in range(2):
            ctx.profile(rep == 1)
            torch.cuda.synchronize(); t = time.time()
            got = read.batch_read_columns(ctx, [cp]); ctx.synchronize()
            td = time.time() - t
        pd = ctx.profile_read()
        ctx.profile(False)
        codec = S.stat_column(c["ptype"], False, enc.pages_numpy(), metas)[0].tolist()
        print("%-13s %-13s %6.1f MB enc %8.2f ms %6.1f GB/s | dec %8.2f ms %6.1f GB/s codec %s ratio %.2f\n      enc: %s\n      dec: %s" % (
            name, oname, nbytes / 1e6, te * 1e3, nbytes / te / 1e9, td * 1e3, nbytes / td / 1e9, codec,
            nbytes / max(1, enc.pages_numpy().size), top(pe), top(pd)), flush=True)
