"""Decoder robustness probe (development) for the tile-parallel parts of the binary read path: binary Dict pages of four
tiles and more (k_bin_tile_sums / k_bin_tile_scan: bit-packed, RLE, plain and LZ4 indices, i32 / i64 offsets, nulls) and
columns of 150 - 200 pages (k_colscan's wave scans, the LZ4 value blocks it queues) with the mutations of tests/fuzzing.py
(byte flips, size fields, truncation).  Every call must return; a page the device decodes to the size the oracle decodes it
to must hold the oracle's bytes.    python tests/probes/fuzz_bin_tiles.py [trials per shape, default 150]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import strawboat_amd as sb
from strawboat_amd import read
from strawboat_amd._native import NativeError
from oracle import sbo as S
from tests import gen
from tests.fuzzing import mutate
ctx = sb.Context(0)
shapes = [("dict 8 tiles bitpack", gen.binary(32_768, uniq=700, zipf=1.2, maxlen=20, seed=1), dict(force_codec=S.DICT)),
          ("dict 8 tiles rle idx nullable", gen.binary(30_000, uniq=50, null_density=0.1, maxlen=12, seed=2), dict(force_codec=S.DICT, force_index_codec=S.RLE)),
          ("dict 6 tiles plain idx large", gen.binary(24_000, uniq=3000, large=True, maxlen=30, seed=3), dict(force_codec=S.DICT, force_index_codec=S.NONE)),
          ("dict lz4 idx 5 pages", gen.binary(100_000, uniq=500, maxlen=16, seed=4), dict(max_page_size=20_000, force_codec=S.DICT, force_index_codec=S.LZ4)),
          ("200 pages adaptive lz4", gen.binary(60_000, uniq=80, null_density=0.2, maxlen=24, seed=5), dict(max_page_size=300, default_compression=S.LZ4, ratio=2.0)),
          ("150 pages basic lz4", gen.binary(45_000, uniq=5000, maxlen=24, seed=6), dict(max_page_size=300, force_codec=S.LZ4))]
for name, col, opt in shapes:
    pages, metas = gen.oracle_write(col, **opt)
    rng = np.random.default_rng(7)
    nok = nerr = bad = 0
    for t in range(int(sys.argv[1]) if len(sys.argv) > 1 else 150):
        pg, m = mutate(rng, pages, metas, t)
        if pg.size == 0:
            continue
        try:
            want = gen.oracle_read(col, pg, m)
        except Exception:
            want = None
        try:
            got = read.read_simple(ctx, read.ColumnPages(col["ptype"], col["nullable"], torch.from_numpy(pg).to(ctx.torch_device), m))
            ctx.synchronize()
            nok += 1
            if want is not None and got.values_len == want["values"].size and not np.array_equal(got.values_numpy(), want["values"]):
                bad += 1
        except NativeError:
            nerr += 1
            try:
                ctx.synchronize()
            except NativeError:
                pass
    print("%-32s %d decoded, %d rejected, %d decoded differently from the oracle" % (name, nok, nerr, bad), flush=True)
