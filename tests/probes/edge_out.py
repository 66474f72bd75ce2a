"""Out-of-range WRITE probe (development): decode into output buffers of exactly the documented capacity,
each followed by a guard zone that must stay untouched.
    python tests/probes/edge_out.py"""
import os, sys, types
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import strawboat_amd as sb
from strawboat_amd import read
from strawboat_amd.types import PhysicalType
from oracle import sbo as S
from tests import gen

ctx = sb.Context(0)
dev = ctx.torch_device
GUARD = 4096


def guarded(nbytes):
    buf = torch.full((nbytes + GUARD,), 0xA5, dtype=torch.uint8, device=dev)
    return buf, buf[:nbytes]


ROWS = (1, 31, 63, 129, 1000, 4097, 4224, 16896, 70001)
cols = []
for pt in (S.T_I8, S.T_I16, S.T_I32, S.T_I64, S.T_F64, S.T_I128, S.T_I256):
    for rows in ROWS:
        cols.append(gen.prim(pt, rows, uniq=200, runs=5, null_density=0.2, seed=rows))
for rows in ROWS:
    cols.append(gen.boolean(rows, null_density=0.2, runs=9, seed=rows))
    for large in (False, True):
        cols.append(gen.binary(rows, uniq=50, null_density=0.2, large=large, seed=rows))
OPTS = [dict(), dict(default_compression=S.LZ4), dict(default_compression=S.ZSTD), dict(default_compression=S.SNAPPY), dict(force_codec=S.RLE),
        dict(force_codec=S.DICT), dict(force_codec=S.DICT, force_index_codec=S.RLE), dict(force_codec=S.ONEVALUE), dict(ratio=1.5),
        dict(force_codec=S.BITPACK), dict(force_codec=S.DELTABP), dict(force_codec=S.PATAS), dict(force_codec=S.FREQ)]
n = bad = 0
for col in cols:
    for opt in OPTS:
        for mps in (None, 4096, 3000):
            c2 = col
            if opt.get("force_codec") == S.ONEVALUE:
                c2 = dict(col)
                if col["ptype"] == S.T_BOOL or col["offsets"] is not None:
                    continue
                c2["values"] = np.zeros_like(col["values"])
            try:
                pages, metas = gen.oracle_write(c2, max_page_size=mps, **opt)
                want = gen.oracle_read(c2, pages, metas)
            except Exception:
                continue
            t, rows = c2["ptype"], c2["rows"]
            w = PhysicalType.WIDTH.get(t, 1)
            bm = ((rows + 31) // 32) * 4
            keep = []
            if t == S.T_BOOL:
                vb, values = guarded(bm)
            elif c2["offsets"] is not None:
                vb, values = guarded(max(len(want["values"]), 1))
            else:
                vb, values = guarded(rows * w)
            ob = offsets = None
            if c2["offsets"] is not None:
                ob, offsets = guarded((rows + 1) * w)
            lb = validity = None
            if c2["nullable"]:
                lb, validity = guarded(bm)
            out = types.SimpleNamespace(values=values, validity=validity, offsets=offsets)
            cp = read.ColumnPages(t, c2["nullable"], torch.from_numpy(pages).to(dev), metas)
            try:
                got = read.ReadBatch(ctx, [cp], out=[out]).enqueue()[0]
                ctx.synchronize()
            except Exception as e:
                print("refused:", t, rows, opt, mps, str(e)[:80])
                continue
            n += 1
            ok = np.array_equal(got.values_numpy(), want["values"])
            for b in (vb, ob, lb):
                if b is not None and not bool((b[-GUARD:] == 0xA5).all()):
                    ok = False
                    print("GUARD OVERWRITTEN", end=" ")
            if not ok:
                bad += 1
                print("BAD: ptype %d rows %d nullable %s opt %s page %s" % (t, rows, c2["nullable"], opt, mps), flush=True)
print("done: %d decodes, %d bad" % (n, bad))
