"""Random binary / integer columns through the adaptive device writer against the oracle's bytes (development probe):
k_enc_bin_page / k_enc_prim_dict (pages of 512 ... 65 536 rows: rows in multiples of 128 are finished inside the kernel,
others by the emitter), strings longer than the 24 bytes the row loop holds in registers, empty strings, pages that give
up on Dict, nulls, LargeBinary, several columns per call (hints from the call before: replays).
usage: python tests/probes/fuzz_bin_encode.py [cases = 150] [seed = 1]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import strawboat_amd as sb
from oracle import sbo as S
from tests import gen
from tests.test_gpu_encode import gpu_encode

CASES = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = sb.Context(0)


def binary_col():
    page = int(rng.choice([512, 640, 1000, 4096, 8192, 30000, 65536]))
    rows = int(page * rng.integers(1, 4) + rng.choice([0, 0, 1, 77, 128, 300]))
    uniq = int(rng.choice([1, 2, 7, 50, 400, 3000, 20000, rows]))
    maxlen = int(rng.choice([0, 3, 8, 16, 24, 25, 40]))
    minlen = int(rng.integers(0, maxlen + 1)) if rng.random() < 0.3 else 0
    nd = float(rng.choice([0, 0, 0.05, 0.5, 0.95])) or None
    zipf = float(rng.choice([0, 0, 1.1, 1.5, 3.0])) or None
    col = gen.binary(rows, uniq=max(1, uniq), null_density=nd, large=bool(rng.random() < 0.3), seed=int(rng.integers(1 << 30)),
                     zipf=zipf, minlen=minlen, maxlen=maxlen)
    if rng.random() < 0.25:   # runs of equal strings: RLE / one-value index blocks
        k = int(rng.choice([2, 16, 64, 500]))
        o = col["offsets"].astype(np.int64)
        pick = np.repeat(rng.integers(0, rows, rows // k + 1), k)[:rows]
        lens = (o[1:] - o[:-1])[pick]
        offs = np.zeros(rows + 1, np.int64)
        np.cumsum(lens, out=offs[1:])
        data = np.concatenate([col["values"][o[r]:o[r + 1]] for r in pick[::k]]) if False else None
        vals = np.zeros(int(offs[-1]), np.uint8)
        src = col["values"]
        for i in range(0, rows, k):   # (k rows share a source string)
            r = int(pick[i]); L = int(o[r + 1] - o[r])
            blk = np.tile(src[o[r]:o[r + 1]], min(k, rows - i))
            vals[offs[i]:offs[i] + blk.size] = blk
        col = dict(col, values=vals, offsets=offs.astype(col["offsets"].dtype))
    return col, page


def int_col():
    page = int(rng.choice([512, 4096, 8192, 65536]))
    rows = int(page * rng.integers(1, 4) + rng.choice([0, 1, 128, 300]))
    pt, npt = [(S.T_I8, np.int8), (S.T_I16, np.int16), (S.T_I32, np.int32), (S.T_U32, np.uint32)][int(rng.integers(0, 4))]
    hi = int(rng.choice([2, 20, 120, 3000, 12000])) if npt not in (np.int8,) else int(rng.choice([2, 20, 120]))
    v = rng.integers(0, hi, rows)
    if rng.random() < 0.3:
        v = np.sort(v)
    nd = float(rng.choice([0, 0, 0.1, 0.9])) or None
    validity = gen.make_validity(rng, rows, nd)
    return dict(ptype=pt, nullable=validity is not None, rows=rows, values=v.astype(npt), validity=validity, offsets=None), page


bad = done = 0
r0 = ctx.replays()
for case in range(CASES):
    col, page = binary_col() if rng.random() < 0.7 else int_col()
    opt = dict(max_page_size=page, ratio=float(rng.choice([1.05, 2.0, 4.0])), forbidden=(),
               default_compression=int(rng.choice([S.NONE, S.LZ4])), lz4_exact=True)
    try:
        want_pages, want_metas = gen.oracle_write(col, **{k: v for k, v in opt.items() if k != "lz4_exact"})
    except Exception as e:
        print("case %d: oracle refuses (%s)" % (case, e)); continue
    for rep in range(2):   # (the second call goes by the first one's hints)
        enc = gpu_encode(ctx, col, **opt)
        ok = np.array_equal(enc.metas_array(), want_metas) and np.array_equal(enc.pages_numpy(), want_pages)
        done += 1
        if not ok:
            bad += 1
            print("MISMATCH case %d rep %d: ptype %s rows %d page %d nullable %s opt %s" % (case, rep, col["ptype"], col["rows"], page, col["nullable"], opt))
print("done: %d encodes of %d cases, %d bad, %d replays" % (done, CASES, bad, ctx.replays() - r0))
