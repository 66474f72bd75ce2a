"""Corrupted pages never take the device down: every codec / column family, a few mutations each
(byte flips, size fields set to extreme values, truncated last page).  Any status is acceptable —
the reference itself panics or errors on such input — a GPU fault or a hang is not.
tests/probes/fuzz_decode.py runs the same cases with many more trials."""
import numpy as np
import pytest

from tests import gen
from tests.fuzzing import CASES, mutate

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("ci", range(len(CASES)))
def test_corrupted_pages_are_survived(gpu_ctx, ci):
    import torch
    from strawboat_amd import read
    from strawboat_amd._native import NativeError
    name, mk = CASES[ci]
    col, opt = mk()
    pages, metas = gen.oracle_write(col, **opt)
    rng = np.random.default_rng(1000 + ci)
    for t in range(12):
        pg, m = mutate(rng, pages, metas, t)
        if pg.size == 0:
            continue
        try:
            cp = read.ColumnPages(col["ptype"], col["nullable"], torch.from_numpy(pg).to(gpu_ctx.torch_device), m)
            read.read_simple(gpu_ctx, cp)
        except NativeError:
            try:
                gpu_ctx.synchronize()
            except NativeError:
                pass
    # the context still works: the untouched pages decode to what the oracle reads
    want = gen.oracle_read(col, pages, metas)
    cp = read.ColumnPages(col["ptype"], col["nullable"], torch.from_numpy(pages).to(gpu_ctx.torch_device), metas)
    got = read.read_simple(gpu_ctx, cp)
    assert np.array_equal(got.values_numpy(), want["values"])
