"""Out-of-range probe (development): every input buffer of an encode call sits flush against the end of a
device allocation of its own, so a read past the column is a GPU fault instead of a silent stray load.
    python tests/probes/edge_alloc.py <first case> <last case>"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import strawboat_amd as sb
from strawboat_amd import read, write
from strawboat_amd.types import WriteOptions
from strawboat_amd._native import NativeError
from oracle import sbo as S
from tests import gen

SEG = 20 << 20
ctx = sb.Context(0)
dev = ctx.torch_device
keep = []


def at_end(a):
    if a is None:
        return None
    b = np.ascontiguousarray(a).view(np.uint8).reshape(-1)
    buf = torch.zeros(SEG, dtype=torch.uint8, device=dev)
    keep.append(buf)
    v = buf[SEG - b.size:] if b.size else buf[SEG:]
    if b.size:
        v.copy_(torch.from_numpy(b.copy()))
    return v


CASES = []
ROWS = (1, 63, 129, 1000, 4224, 12416, 16896, 16960, 70000)
for pt in (S.T_I8, S.T_I16, S.T_I32, S.T_I64, S.T_F32, S.T_F64, S.T_I128, S.T_I256):
    for rows in ROWS:
        for nd in (None, 0.2):
            CASES.append((gen.prim(pt, rows, uniq=300, runs=5, null_density=nd, seed=rows), "prim %d" % pt))
for rows in ROWS:
    for nd in (None, 0.2):
        CASES.append((gen.boolean(rows, null_density=nd, runs=9, seed=rows), "bool"))
        for large in (False, True):
            CASES.append((gen.binary(rows, uniq=50, null_density=nd, large=large, seed=rows), "binary large=%d" % large))
OPTS = [dict(ratio=2.0, forbidden=()), dict(ratio=1.1, default_compression=S.LZ4, forbidden=()), dict(default_compression=S.LZ4),
        dict(default_compression=S.ZSTD), dict(force_codec=S.RLE), dict(force_codec=S.DICT), dict(force_codec=S.DICT, force_index_codec=S.LZ4)]

lo, hi = int(sys.argv[1]), int(sys.argv[2])
n = bad = 0
for ci, (col, name) in enumerate(CASES):
    if ci < lo or ci > hi:
        continue
    for oi, opt in enumerate(OPTS):
        if col["ptype"] == S.T_BOOL and opt.get("force_codec") == S.DICT:
            continue
        print("case %d opt %d: %s rows %d nullable %s %s" % (ci, oi, name, col["rows"], col["nullable"], opt), flush=True)
        keep.clear()
        try:
            want_pages, want_metas = gen.oracle_write(col, max_page_size=65536, **opt)
        except Exception as e:
            print("   oracle refuses:", e)
            continue
        o = dict(opt)
        lz4 = o.get("default_compression") == S.LZ4 or o.get("force_index_codec") == S.LZ4
        for exact in ((True, False) if lz4 else (True,)):   # both LZ4 encoders: byte-exact (compared) and the parallel default
            wo = WriteOptions(max_page_size=65536, default_compression=o.get("default_compression", S.NONE),
                              default_compress_ratio=o.get("ratio"), forbidden_compressions=list(o.get("forbidden", ())),
                              force_codec=o.get("force_codec", -1), force_index_codec=o.get("force_index_codec", -1), lz4_exact=exact)
            dc = write.DeviceColumn(col["ptype"], col["nullable"], col["rows"], at_end(col["values"]), at_end(col["validity"]), at_end(col["offsets"]))
            try:
                enc = write.encode_columns(ctx, [dc], wo)
                ctx.synchronize()
            except NativeError as e:
                print("   device refuses:", e)
                enc = None
                break
            n += 1
            if exact and opt.get("default_compression") != S.ZSTD:
                if not np.array_equal(enc[0].pages_numpy(), want_pages):
                    bad += 1
                    print("   MISMATCH")
            if exact and lz4:   # (the parallel encoder's pages are checked by the decode below)
                got = read.read_simple(ctx, read.ColumnPages(col["ptype"], col["nullable"], at_end(enc[0].pages_numpy()), enc[0].metas_array()))
                if not np.array_equal(got.values_numpy(), gen.oracle_read(col, want_pages, want_metas)["values"]):
                    bad += 1
                    print("   DECODE MISMATCH (exact)")
        if enc is None:
            continue
        # decode from a buffer flush against the end as well
        pages = at_end(enc[0].pages_numpy())
        got = read.read_simple(ctx, read.ColumnPages(col["ptype"], col["nullable"], pages, enc[0].metas_array()))
        want = gen.oracle_read(col, enc[0].pages_numpy(), enc[0].metas_array())
        if not np.array_equal(got.values_numpy(), want["values"]):
            bad += 1
            print("   DECODE MISMATCH")
print("done: %d encodes, %d mismatches" % (n, bad))
