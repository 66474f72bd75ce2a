"""Dict pages with Freq-coded indices, inputs flush against the end of their allocation (development probe)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import strawboat_amd as sb
from strawboat_amd import read, write
from strawboat_amd.types import WriteOptions
from oracle import sbo as S
from tests import gen
SEG = 20 << 20
ctx = sb.Context(0)
keep = []
def at_end(a):
    if a is None:
        return None
    b = np.ascontiguousarray(a).view(np.uint8).reshape(-1)
    buf = torch.zeros(SEG, dtype=torch.uint8, device=ctx.torch_device); keep.append(buf)
    v = buf[SEG - b.size:]; v.copy_(torch.from_numpy(b.copy())); return v
rng = np.random.default_rng(2)
n = bad = 0
for ptype, npt in ((S.T_I64, np.int64), (S.T_I32, np.int32), (S.T_I16, np.int16)):
    for rows in (300, 4097, 8192, 20000, 70001):
        for nd in (None, 0.03):
            v = np.full(rows, -5, npt)
            exc = rng.random(rows) < 0.06
            v[exc] = -rng.integers(10, 2000, int(exc.sum())).astype(npt)
            col = dict(ptype=ptype, nullable=nd is not None, rows=rows, values=v, validity=gen.make_validity(rng, rows, nd), offsets=None)
            for dc in (S.NONE, S.LZ4):
                keep.clear()
                want_pages, want_metas = gen.oracle_write(col, max_page_size=8192, ratio=2.0, forbidden=(S.RLE,), default_compression=dc)
                wo = WriteOptions(max_page_size=8192, default_compression=dc, default_compress_ratio=2.0, forbidden_compressions=[S.RLE], lz4_exact=True)   # (byte parity with liblz4 needs the exact parse)
                dc_ = write.DeviceColumn(ptype, col["nullable"], rows, at_end(col["values"]), at_end(col["validity"]), None)
                enc = write.encode_columns(ctx, [dc_], wo); ctx.synchronize()
                n += 1
                if not np.array_equal(enc[0].pages_numpy(), want_pages):
                    bad += 1; print("ENCODE MISMATCH", ptype, rows, nd, dc)
                got = read.read_simple(ctx, read.ColumnPages(ptype, col["nullable"], at_end(want_pages), want_metas))
                want = gen.oracle_read(col, want_pages, want_metas)
                if not np.array_equal(got.values_numpy(), want["values"]):
                    bad += 1; print("DECODE MISMATCH", ptype, rows, nd, dc)
# Binary / Utf8 columns under a forced Dict codec: the Freq kernels also write the `u64 len | bytes` entries
from tests.test_gpu_freq import sparse_bin
for large in (False, True):
    for rows in (300, 4097, 8192, 20000):
        for nd in (None, 0.03):
            for top in (b"the-common-value", b""):
                col = sparse_bin(rows, 0.06, rows + (7 if nd else 0), exc_uniq=600, large=large, null_density=nd, top=top)
                for opt in (dict(force_index_codec=S.FREQ), dict(ratio=2.0, forbidden=(S.RLE,))):
                    keep.clear()
                    try:
                        want_pages, want_metas = gen.oracle_write(col, max_page_size=8192, force_codec=S.DICT, **opt)
                    except Exception as e:
                        print("oracle refuses", rows, nd, opt, e); continue
                    wo = WriteOptions(max_page_size=8192, force_codec=S.DICT, force_index_codec=opt.get("force_index_codec", -1),
                                      default_compress_ratio=opt.get("ratio"), forbidden_compressions=list(opt.get("forbidden", ())))
                    dc_ = write.DeviceColumn(col["ptype"], col["nullable"], rows, at_end(col["values"]), at_end(col["validity"]),
                                             at_end(col["offsets"]))
                    enc = write.encode_columns(ctx, [dc_], wo); ctx.synchronize()
                    n += 1
                    if not np.array_equal(enc[0].pages_numpy(), want_pages):
                        bad += 1; print("ENCODE MISMATCH binary", large, rows, nd, top, opt)
                    got = read.read_simple(ctx, read.ColumnPages(col["ptype"], col["nullable"], at_end(want_pages), want_metas))
                    want = gen.oracle_read(col, want_pages, want_metas)
                    if not (np.array_equal(got.values_numpy(), want["values"]) and np.array_equal(got.offsets_numpy(), want["offsets"])):
                        bad += 1; print("DECODE MISMATCH binary", large, rows, nd, top, opt)
print("done: %d cases, %d bad" % (n, bad))
