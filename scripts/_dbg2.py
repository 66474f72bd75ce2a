import sys, numpy as np
sys.path.insert(0, ".")
from oracle import sbo as S
from tests import gen
rows=120_000
for k in (2,6,10):
    col = gen.binary(rows, uniq=300, null_density=0.1, seed=k)
    pages, metas = gen.oracle_write(col, max_page_size=32768, ratio=2.0, forbidden=())
    want = gen.oracle_read(col, pages, metas)
    print(k, "values", col["values"].size, "decoded", want["values"].size, "offsets last", int(col["offsets"][-1]), int(want["offsets"].view(np.int32)[-1]) if want["offsets"].dtype==np.uint8 else int(want["offsets"][-1]))
