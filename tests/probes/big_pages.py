"""Columns written without max_page_size are ONE page (development probe): multi-million-row pages of every
shape through the adaptive device writer and the device reader, byte-compared with the oracle."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import strawboat_amd as sb
from oracle import sbo as S
from tests import gen
from tests.test_gpu_select import check as sel_check
from tests.test_gpu_decode import check as dec_check
from tests.test_gpu_freq import sparse

ctx = sb.Context(0)
ROWS = int(sys.argv[1]) if len(sys.argv) > 1 else 12_000_000
ONLY = sys.argv[2].split(",") if len(sys.argv) > 2 else None   # (substrings of the case names)
LAZY = [("runs i64", lambda: gen.prim(S.T_I64, ROWS, uniq=200, runs=50, seed=1)),
         ("low-card i32", lambda: gen.prim(S.T_I32, ROWS, uniq=500, seed=2)),
         ("low-card nullable f64", lambda: gen.prim(S.T_F64, ROWS, uniq=300, null_density=0.1, seed=3)),
         ("sparse i64", lambda: sparse(S.T_I64, ROWS, 0.02, 4)),
         ("random u32", lambda: gen.prim(S.T_U32, ROWS, uniq=1 << 30, seed=5)),
         ("sorted i64", lambda: gen.prim(S.T_I64, ROWS, uniq=1 << 40, sorted_=True, seed=6)),
         ("bool runs", lambda: gen.boolean(ROWS, null_density=0.05, runs=30, seed=7)),
         ("utf8 zipf", lambda: gen.binary(ROWS // 4, uniq=5000, zipf=1.3, seed=8)),
         ("utf8 unique", lambda: gen.binary(ROWS // 8, uniq=1 << 30, seed=9, maxlen=16))]
CASES = [(n, f()) for n, f in LAZY if ONLY is None or any(o in n for o in ONLY)]
bad = 0
for name, col in CASES:
    for opt in (dict(ratio=2.0, forbidden=()), dict(ratio=2.0, default_compression=S.LZ4, forbidden=())):
        t = time.time()
        try:
            codecs = sel_check(ctx, col, **opt)
            dec_check(ctx, col, **opt)
            print("%-24s %-40s codecs %s  %.1f s" % (name, opt, codecs.tolist(), time.time() - t), flush=True)
        except Exception as e:
            bad += 1
            print("%-24s %-40s FAILED: %s" % (name, opt, str(e)[:200]), flush=True)
print("done: %d bad" % bad)
