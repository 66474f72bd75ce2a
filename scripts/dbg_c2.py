import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import strawboat_amd as sb
from oracle import sbo as S
from tests import gen
from tests.test_gpu_encode import gpu_encode
ctx = sb.Context(0)
vals, valid = bench.gen_c2_column(42)
col = dict(ptype=S.T_F64, nullable=True, rows=vals.size, values=vals, validity=valid, offsets=None)
opt = dict(max_page_size=65536, ratio=2.0, forbidden=(S.FREQ,))
wp, wm = gen.oracle_write(col, **opt)
enc = gpu_encode(ctx, col, **opt)
gp, gm = enc.pages_numpy(), enc.metas_array()
print(wm[0], gm[0])
defb = 4 + 3 + 8192
w = wp[defb + 9: int(wm[0, 0])]
g = gp[defb + 9: int(gm[0, 0])]
nw, ng = len(w) // 12, len(g) // 12
wr = np.frombuffer(w[:nw * 12].tobytes(), dtype=[("c", "<u4"), ("v", "<f8")])
gr = np.frombuffer(g[:ng * 12].tobytes(), dtype=[("c", "<u4"), ("v", "<f8")])
print(nw, ng)
for i in range(min(nw, ng)):
    if wr[i] != gr[i]:
        print("first diff at record", i, "row", int(wr["c"][:i].sum()))
        print("want", wr[max(0, i - 2): i + 3])
        print("got ", gr[max(0, i - 2): i + 3])
        r = int(wr["c"][:i].sum())
        vb = np.unpackbits(valid, bitorder="little")[:vals.size]
        print("rows", r - 3, "..", r + 40)
        print(vals[r - 3:r + 40])
        print(vb[r - 3:r + 40])
        break
