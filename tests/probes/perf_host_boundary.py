"""PCIe-inclusive rate (development probe): the C2 workload handed over as HOST buffers (SB_MEM_HOST, the shape the
reference's callers have) — host Arrow buffers in, host page bytes out, and back into host buffers.  Never bench.py's
`value`; the number goes into DESIGN.md §4."""
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import strawboat_amd as sb
from strawboat_amd import _native as N
from strawboat_amd.types import WriteOptions
from strawboat_amd.write import options_c
from oracle import sbo as S

COLS = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ROWS, PAGE = 1_000_000, 65536
ctx = sb.Context(0)
lib, h = ctx._lib, ctx._h
oc = options_c(WriteOptions(max_page_size=PAGE, default_compress_ratio=2.0))
rng = np.random.default_rng(0)

def buf(n, pinned):
    t = torch.zeros(n, dtype=torch.uint8)
    return t.pin_memory() if pinned else t

for pinned in (False, True):
    vals, valids, outs, metas, vouts, bouts = [], [], [], [], [], []
    npg = C.c_uint64()
    bound = lib.sb_write_bound(S.T_F64, 1, ROWS, 0, C.byref(oc), C.byref(npg))
    for c in range(COLS):
        v = np.repeat(rng.integers(0, 1000, ROWS // 40 + 1), 40)[:ROWS].astype(np.float64)
        b = np.packbits(rng.random(ROWS) >= 0.1, bitorder="little")
        tv, tb = buf(ROWS * 8, pinned), buf(b.size, pinned)
        tv.numpy()[:] = v.view(np.uint8); tb.numpy()[:] = b
        vals.append(tv); valids.append(tb)
        outs.append(buf(bound, pinned)); metas.append((N.PageMetaC * npg.value)())
        vouts.append(buf(ROWS * 8, pinned)); bouts.append(buf((ROWS + 31) // 32 * 4, pinned))
    cw = (N.ColumnWriteC * COLS)()
    for c in range(COLS):
        cw[c].physical_type, cw[c].is_nullable, cw[c].rows = S.T_F64, 1, ROWS
        cw[c].values, cw[c].validity = vals[c].data_ptr(), valids[c].data_ptr()
        cw[c].out_pages, cw[c].out_capacity = outs[c].data_ptr(), bound
        cw[c].out_metas, cw[c].n_pages_capacity = metas[c], npg.value
    best_e = best_d = 1e9
    for rep in range(4):
        t = time.time()
        ctx._check(lib.sb_write_columns(h, cw, COLS, C.byref(oc), N.SB_MEM_HOST)); ctx.synchronize()
        best_e = min(best_e, time.time() - t)
        cr = (N.ColumnReadC * COLS)()
        for c in range(COLS):
            cr[c].physical_type, cr[c].is_nullable = S.T_F64, 1
            cr[c].pages, cr[c].pages_len = outs[c].data_ptr(), int(cw[c].out_len)
            cr[c].metas, cr[c].n_pages = metas[c], int(cw[c].n_pages)
            cr[c].values, cr[c].values_capacity = vouts[c].data_ptr(), ROWS * 8
            cr[c].validity, cr[c].validity_capacity = bouts[c].data_ptr(), bouts[c].numel()
        t = time.time()
        ctx._check(lib.sb_read_columns(h, cr, COLS, N.SB_MEM_HOST)); ctx.synchronize()
        best_d = min(best_d, time.time() - t)
    ok = True
    for c in range(COLS):  # null slots carry no value: compare the valid rows and the bitmaps
        m = np.unpackbits(valids[c].numpy(), bitorder="little")[:ROWS].astype(bool)
        ok &= bool(np.array_equal(vouts[c].numpy().view(np.float64)[m], vals[c].numpy().view(np.float64)[m]))
        ok &= bool(np.array_equal(bouts[c].numpy()[:(ROWS + 7) // 8], valids[c].numpy()))
    arrow = COLS * (ROWS * 8 + (ROWS + 7) // 8)
    print("%s host buffers, %d columns: encode %.1f ms = %.1f GB/s, decode %.1f ms = %.1f GB/s, both %.1f GB/s of Arrow bytes; "
          "round trip %s" % ("pinned" if pinned else "pageable", COLS, best_e * 1e3, arrow / best_e / 1e9, best_d * 1e3,
                             arrow / best_d / 1e9, 2 * arrow / (best_e + best_d) / 1e9, "ok" if ok else "MISMATCH"), flush=True)
