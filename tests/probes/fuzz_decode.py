"""Decoder robustness probe (development): oracle-written pages with random byte flips / truncations are
decoded on the device; any status is fine, a GPU fault or a hang is not.  One case per process:
    for i in $(seq 0 62); do timeout 300 python tests/probes/fuzz_decode.py $i 150 || echo "CASE $i FAILED"; done"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import strawboat_amd as sb
from strawboat_amd import read
from strawboat_amd._native import NativeError
from tests import gen
from tests.fuzzing import CASES, mutate

if __name__ == "__main__":
    ci, trials = int(sys.argv[1]), int(sys.argv[2])
    if ci >= len(CASES):
        print("no case", ci); sys.exit(0)
    name, mk = CASES[ci]
    col, opt = mk()
    pages, metas = gen.oracle_write(col, **opt)
    ctx = sb.Context(0)
    rng = np.random.default_rng(1000 + ci)
    nerr = nok = 0
    for t in range(trials):
        pg, m = mutate(rng, pages, metas, t)
        if pg.size == 0:
            continue
        try:
            cp = read.ColumnPages(col["ptype"], col["nullable"], torch.from_numpy(pg).to(ctx.torch_device), m)
            read.read_simple(ctx, cp)
            nok += 1
        except NativeError:
            nerr += 1
            try:
                ctx.synchronize()
            except NativeError:
                pass
    print("case %d %-32s trials %d: %d decoded, %d rejected" % (ci, name, trials, nok, nerr), flush=True)
