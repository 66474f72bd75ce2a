"""binary pages on the section-parallel path, several per column / call (development probe; build with
make EXTRA=-DSB_BIN_BIG_ROWS=65536 to send 64 Ki-row pages there)"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import strawboat_amd as sb
from oracle import sbo as S
from tests import gen
from tests.test_gpu_select import check as sel_check
from tests.test_gpu_decode import check as dec_check
ctx = sb.Context(0)
which = sys.argv[1]
rows, mps = int(sys.argv[2]), int(sys.argv[3])
col = gen.binary(rows, uniq=3000, zipf=1.2, maxlen=20, seed=5, null_density=0.1 if which == "nulls" else None)
opt = dict(ratio=2.0, max_page_size=mps, forbidden=(S.DICT,) if which == "nodict" else ())
print(which, rows, mps, sel_check(ctx, col, **opt).tolist()[:6], flush=True)
dec_check(ctx, col, **opt)
print("ok", flush=True)
