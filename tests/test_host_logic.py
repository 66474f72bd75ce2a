"""Host-side logic that needs no GPU: sharding plan, ColumnMeta offsets, option mapping."""
import numpy as np

from strawboat_amd import shard
from strawboat_amd.types import Compression, WriteOptions
from strawboat_amd.write import options_c


def test_plan_shards_balances_by_bytes():
    # C4: 8 columns, Utf8 ~3x heavier than Boolean
    sizes = [40, 40, 80, 80, 240, 240, 2, 2]
    shards = shard.plan_shards(sizes, 8)
    assert sorted(sum(shards, [])) == list(range(8)) and all(len(s) == 1 for s in shards)
    shards = shard.plan_shards(sizes, 2)
    loads = [sum(sizes[i] for i in s) for s in shards]
    assert abs(loads[0] - loads[1]) <= 4
    assert shard.plan_shards([5, 5, 5], 4)[3] == []


def test_column_metas_offsets():
    metas = [np.array([[100, 10], [50, 5]], np.uint64), np.array([[7, 1]], np.uint64), np.zeros((0, 2), np.uint64)]
    cm = shard.column_metas(metas)
    assert [c.offset for c in cm] == [8, 158, 165]
    assert cm[0].pages[1].num_values == 5 and cm[0].total_len() == 150


def test_gather_metas_single_process():
    local = {0: np.array([[10, 1]], np.uint64), 1: np.array([[20, 2], [30, 3]], np.uint64)}
    out = shard.gather_metas(local, 2)
    assert out[1].tolist() == [[20, 2], [30, 3]]


def test_gather_metas_refuses_gaps_and_duplicates():
    """(column, page range) work items: every page of every column exactly once, or an error — never a shorter column
    whose later ColumnMeta offsets are silently wrong"""
    import pytest
    m = lambda n: np.array([[10 + k, 1] for k in range(n)], np.uint64)
    ok = shard.gather_metas([(0, 0, m(2)), (0, 2, m(1)), (1, 0, m(1))], 2, expected_pages=[3, 1])
    assert [len(x) for x in ok] == [3, 1]
    with pytest.raises(ValueError, match="missing"):
        shard.gather_metas([(0, 0, m(2)), (0, 3, m(1))], 1)                     # page 2 never arrived
    with pytest.raises(ValueError, match="twice"):
        shard.gather_metas([(0, 0, m(2)), (0, 1, m(2))], 1)                     # page 1 dealt twice
    with pytest.raises(ValueError, match="of its 4 pages"):
        shard.gather_metas([(0, 0, m(3))], 1, expected_pages=[4])               # the rank owning the tail sent nothing
    with pytest.raises(ValueError, match="columns"):
        shard.gather_metas([(5, 0, m(1))], 2)


def test_options_mapping():
    o = options_c(WriteOptions(default_compression=Compression.LZ4, default_compress_ratio=2.0, max_page_size=8192,
                               forbidden_compressions=[Compression.FREQ, Compression.PATAS], force_codec=10, rng_seed=7))
    assert o.default_compression == 1 and o.has_default_compress_ratio == 1 and o.default_compress_ratio == 2.0
    assert o.max_page_size == 8192 and o.forbidden_compressions == (1 << 13) | (1 << 16)
    assert o.force_codec == 10 and o.force_index_codec == -1 and o.rng_seed == 7
    o = options_c(WriteOptions())
    assert o.has_default_compress_ratio == 0 and o.max_page_size == 0


def test_c5_page_range_slices_are_columns_of_their_own():
    """workloads.c5_slice (what a rank holds when it owns a page range of a C5 leaf column, SURVEY 8e work items): the level
    sections and leaf ranges the oracle writes for the pages of the slice equal those of the same pages of the whole column"""
    import workloads as W
    from oracle import sbo as S
    la, a, lb, b = W.c5_nested(rows=20_000, seed=7)
    page = 4096
    for levels, leaf in ((la, a), (lb, b)):
        for r0, r1 in ((0, 8192), (8192, 16384), (16384, 20_000), (4096, 20_000)):
            lv, col = W.c5_slice(levels, leaf, r0, r1)
            assert col["rows"] == lv[-1]["length"] == lv[1]["length"]
            whole_leaf0 = None
            for q0 in range(r0, r1, page):
                ln = min(page, r1 - q0)
                want, nv, ls, lc = S.nested_write_levels(levels, q0, ln)
                got, nv2, ls2, lc2 = S.nested_write_levels(lv, q0 - r0, ln)
                assert np.array_equal(got, want) and nv == nv2 and lc == lc2
                if whole_leaf0 is None:
                    whole_leaf0 = ls - ls2          # the slice's leaf rows are the whole column's, shifted by a constant
                assert ls - ls2 == whole_leaf0
            e0 = whole_leaf0
            if leaf["offsets"] is None:
                assert np.array_equal(col["values"], np.asarray(leaf["values"])[e0:e0 + col["rows"]])
            else:
                bo = np.asarray(leaf["offsets"])
                assert np.array_equal(col["offsets"], bo[e0:e0 + col["rows"] + 1] - bo[e0])
                assert np.array_equal(col["values"], leaf["values"][int(bo[e0]):int(bo[e0 + col["rows"]])])
