"""Device time of one-page columns (development probe): encode / decode of a single multi-million-row page."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import strawboat_amd as sb
from oracle import sbo as S
from tests import gen
from tests.test_gpu_encode import gpu_encode
from tests.test_gpu_decode import gpu_decode
from tests.test_gpu_freq import sparse

ctx = sb.Context(0)
ROWS = int(sys.argv[1]) if len(sys.argv) > 1 else 12_000_000
rng = np.random.default_rng(1)
def col(ptype, v):
    return dict(ptype=ptype, nullable=False, rows=v.size, values=v, validity=None, offsets=None)
CASES = [("random u32", col(S.T_U32, rng.integers(0, 1 << 30, ROWS).astype(np.uint32))),
         ("sorted i64", col(S.T_I64, np.sort(rng.integers(0, 1 << 40, ROWS)).astype(np.int64))),
         ("low-card i32", col(S.T_I32, rng.integers(0, 500, ROWS).astype(np.int32))),
         ("runs i64", col(S.T_I64, np.repeat(rng.integers(0, 200, ROWS // 50 + 1), 50)[:ROWS].astype(np.int64))),
         ("sparse i64", sparse(S.T_I64, ROWS, 0.02, 4))]
for name, c in CASES:
    for opt in (dict(ratio=2.0, forbidden=()), dict(ratio=2.0, default_compression=S.LZ4, forbidden=())):
        for rep in range(2):
            torch.cuda.synchronize(); t = time.time()
            enc = gpu_encode(ctx, c, **opt); ctx.synchronize(); torch.cuda.synchronize()
            te = time.time() - t
        pages, metas = enc.pages_numpy(), enc.metas_array()
        t = time.time()
        gpu_decode(ctx, c, pages, metas); ctx.synchronize(); torch.cuda.synchronize()
        td = time.time() - t
        print("%-14s %-58s encode %8.1f ms (incl. upload)  decode %8.1f ms  codec %s" % (
            name, opt, te * 1e3, td * 1e3, S.stat_column(c["ptype"], False, pages, metas)[0].tolist()), flush=True)
