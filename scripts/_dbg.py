import sys, numpy as np
sys.path.insert(0, ".")
import strawboat_amd as sb
from oracle import sbo as S
from tests import gen
from tests.test_gpu_encode import gpu_encode
from tests.test_gpu_big_pages import CASES, OPTS
ctx = sb.Context(0)
for name in sys.argv[1:]:
    col = CASES[name]
    for opt in OPTS:
        wp, wm = gen.oracle_write(col, **opt)
        enc = gpu_encode(ctx, col, **opt)
        gp, gm = enc.pages_numpy(), enc.metas_array()
        wc = S.stat_column(col["ptype"], col["nullable"], wp, wm)
        gc = S.stat_column(col["ptype"], col["nullable"], gp, gm)
        same = gp.size == wp.size and np.array_equal(gp, wp)
        first = -1 if same else int(np.argmax(gp[:min(gp.size, wp.size)] != wp[:min(gp.size, wp.size)]))
        print(name, opt, "want", wc[0].tolist(), wc[1].tolist(), wm.tolist(), "got", gc[0].tolist(), gc[1].tolist(), gm.tolist(), "same", same, "first diff", first, flush=True)
