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


@pytest.mark.parametrize("shape", ["bitpack", "delta", "dict_bitpacked_indices"])
def test_a_last_page_that_consumes_less_than_its_length_reads_like_upstream(gpu_ctx, shape):
    """a header whose compressed size is SMALLER than the body that follows (tests/probes/fuzz_long_bp.py found 4 of 240
    damaged pages where only the oracle refused): the reference never compares what a page's decoders consumed with
    PageMeta.length — the Extend codecs see the rest of the buffer (src/compression/integer/mod.rs:108-114) and the reader drops
    what is left of the page (src/read/array/integer.rs:69-81) — so such a page, the last of its column, decodes to the original
    values.  Pinned in the oracle (read_column) and on the device; a page that consumes MORE than its length is refused by both."""
    from oracle import sbo as S
    from strawboat_amd import read
    from strawboat_amd._native import NativeError
    import torch
    rng = np.random.default_rng(4)
    n = 128 * 600
    if shape == "bitpack":
        pt, v, o = S.T_U32, rng.integers(256, 512, n).astype(np.uint32), S.make_options(force_codec=S.BITPACK)
    elif shape == "delta":
        pt, v, o = S.T_U32, np.cumsum(rng.integers(0, 4000, n)).astype(np.uint32), S.make_options(force_codec=S.DELTABP)
    else:
        pt, v, o = S.T_I64, (rng.integers(0, 3000, n) * 1_000_003).astype(np.int64), S.make_options(force_codec=S.DICT)
    page, metas = S.write_column(pt, False, n, v, options=o)
    page = np.array(page, dtype=np.uint8)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(gpu_ctx.torch_device)
    csize = int.from_bytes(page[1:5].tobytes(), "little")
    want = np.ascontiguousarray(v).view(np.uint8).reshape(-1)
    for smaller in (csize - 1, csize // 2, 1):
        b = page.copy()
        b[1:5] = np.frombuffer(int(smaller).to_bytes(4, "little"), np.uint8)
        if shape == "dict_bitpacked_indices":   # (the Dict page's own size covers the nested block and the values: not an Extend codec's input)
            continue
        got_o = S.read_column(pt, False, b, metas)["values"]
        assert np.array_equal(np.ascontiguousarray(got_o).view(np.uint8).reshape(-1), want), (shape, smaller)
        got = read.read_simple(gpu_ctx, read.ColumnPages(pt, False, up(b), np.array(metas, np.uint64))).values_numpy()
        assert np.array_equal(got, want), (shape, smaller)
    if shape == "dict_bitpacked_indices":   # the nested index block's compressed size made smaller: the same rule one level down
        b = page.copy()
        ic = int.from_bytes(b[10:14].tobytes(), "little")
        b[10:14] = np.frombuffer(int(ic - 3).to_bytes(4, "little"), np.uint8)
        try:
            got_o = np.ascontiguousarray(S.read_column(pt, False, b, metas)["values"]).view(np.uint8).reshape(-1)
        except Exception:
            got_o = None
        try:
            got = read.read_simple(gpu_ctx, read.ColumnPages(pt, False, up(b), np.array(metas, np.uint64))).values_numpy()
        except NativeError:
            got = None
        assert (got is None) == (got_o is None), (got is None, got_o is None)
        if got is not None:
            assert np.array_equal(got, got_o)
    # a page that consumes MORE than PageMeta.length (the meta says one byte less): refused by both
    m = np.array(metas, np.uint64).copy()
    m[0, 0] -= 1
    with pytest.raises(Exception):
        S.read_column(pt, False, page[:-1], m)
    with pytest.raises(NativeError):
        read.read_simple(gpu_ctx, read.ColumnPages(pt, False, up(page[:-1].copy()), m))
