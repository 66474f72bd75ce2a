"""Nested level-section robustness probe (development): many corrupted variants per shape; any status is fine,
a GPU fault or a hang is not.   python tests/probes/fuzz_nested.py <trials>"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import strawboat_amd as sb
from strawboat_amd import nested
from strawboat_amd.read import ColumnPages
from strawboat_amd._native import NativeError
from oracle import sbo as S
from tests.fuzzing import mutate
from tests.nested_gen import make_nested
from tests.test_gpu_nested import SHAPES, leaf_values, oracle_pages, up

ctx = sb.Context(0)
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for shape in SHAPES:
    levels, rows = make_nested(shape, 3000, 5)
    values, w = leaf_values(levels, S.T_I32, 3)
    pages, metas = oracle_pages(levels, S.T_I32, values, rows, 700, force_codec=S.NONE)
    kinds, opt = [lv["kind"] for lv in levels], [bool(lv["is_optional"]) for lv in levels]
    rng = np.random.default_rng(1)
    ok = bad = 0
    for t in range(trials):
        pg, m = mutate(rng, pages, metas, t)
        if pg.size == 0:
            continue
        print("shape %s trial %d" % (shape, t), flush=True)
        try:
            nested.read_nested(ctx, ColumnPages(S.T_I32, False, up(ctx, pg), m), kinds, opt)
            ok += 1
        except NativeError:
            bad += 1
            try:
                ctx.synchronize()
            except NativeError:
                pass
    print("%-14s %d decoded, %d rejected" % (shape, ok, bad), flush=True)
