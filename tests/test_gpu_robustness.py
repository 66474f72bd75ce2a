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


@pytest.mark.parametrize("shape", ["list", "list_list", "list_struct", "struct_list"])
def test_corrupted_nested_pages_are_survived(gpu_ctx, shape):
    """level sections (page_rows | rep_len | def_len | rep | def) with flipped bytes, extreme length fields and
    truncation: read_nested reports a status or decodes garbage, and the context keeps working"""
    from strawboat_amd import nested
    from strawboat_amd.read import ColumnPages
    from strawboat_amd._native import NativeError
    from oracle import sbo as S
    from tests.nested_gen import make_nested
    from tests.test_gpu_nested import check_decoded, leaf_values, oracle_pages, up
    levels, rows = make_nested(shape, 3000, 5)
    values, w = leaf_values(levels, S.T_I32, 3)
    pages, metas = oracle_pages(levels, S.T_I32, values, rows, 700, force_codec=S.NONE)
    kinds, opt = [lv["kind"] for lv in levels], [bool(lv["is_optional"]) for lv in levels]
    rng = np.random.default_rng(11)
    for t in range(24):
        pg, m = mutate(rng, pages, metas, t)
        if pg.size == 0:
            continue
        try:
            nested.read_nested(gpu_ctx, ColumnPages(S.T_I32, False, up(gpu_ctx, pg), m), kinds, opt)
        except NativeError:
            try:
                gpu_ctx.synchronize()
            except NativeError:
                pass
    arr = nested.read_nested(gpu_ctx, ColumnPages(S.T_I32, False, up(gpu_ctx, pages), metas), kinds, opt)
    check_decoded(arr, levels, rows, values, w)
