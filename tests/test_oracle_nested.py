"""Nested (Dremel) level sections of the CPU oracle: encode -> decode round trips on the nested
shapes of the reference's integration tests (tests/it/io.rs:167-278: list, list_list, list_struct,
struct_list; offsets step gen_range(0..3), 10 % null lists), plus hand-checked small cases."""
import numpy as np
import pytest

from oracle import sbo as S
from tests import gen
from tests.nested_gen import expected_state, make_nested


def test_list_int32_hand_checked():
    # rows: [1, null], [], null, [3]
    lv = [dict(kind=S.K_LIST, is_optional=True, validity=gen.pack_bits([1, 1, 0, 1]),
               offsets=np.array([0, 2, 2, 2, 3], np.int32), length=4),
          dict(kind=S.K_PRIMITIVE, is_optional=True, validity=gen.pack_bits([1, 0, 1]), length=3)]
    b, nv, ls, lc = S.nested_write_levels(lv, 0, 4)
    # u32 rows=4 | u32 rep_len=2 | u32 def_len=3 | rep: 03 02 | def: 03 1b 03
    assert bytes(b).hex() == "040000000200000003000000" + "0302" + "031b03"
    assert (nv, ls, lc) == (5, 0, 3)
    r = S.nested_read_levels(b, nv, [1, 0], [1, 1])
    assert r["consumed"] == len(b) and r["lengths"] == [4, 3]
    assert r["offsets"][0].tolist() == [0, 2, 2, 2] and r["validity"][0].tolist() == [1, 1, 0, 1]
    assert r["leaf_validity"].tolist() == [1, 0, 1]


def test_required_struct_of_required_leaf_has_no_levels():
    lv = [dict(kind=S.K_STRUCT, is_optional=False, length=10), dict(kind=S.K_PRIMITIVE, is_optional=False, length=10)]
    b, nv, ls, lc = S.nested_write_levels(lv, 2, 5)
    assert bytes(b).hex() == "05000000" + "00000000" + "00000000" and (nv, ls, lc) == (5, 2, 5)
    r = S.nested_read_levels(b, nv, [3, 0], [0, 0])
    assert r["lengths"] == [5, 5]


@pytest.mark.parametrize("shape", ["list", "large_list", "list_list", "list_struct", "struct_list", "struct_struct",
                                   "list_required"])
@pytest.mark.parametrize("seed", [1, 2])
def test_round_trip(shape, seed):
    levels, rows = make_nested(shape, 3000, seed)
    kinds = [lv["kind"] for lv in levels]
    nullable = [int(lv["is_optional"]) for lv in levels]
    for r0, ln in ((0, rows), (0, 1000), (1000, 1000), (2999, 1), (1234, 777)):
        b, nv, ls, lc = S.nested_write_levels(levels, r0, ln)
        got = S.nested_read_levels(b, nv, kinds, nullable)
        want = expected_state(levels, r0, ln)
        assert got["consumed"] == len(b)
        assert (ls, lc) == (want["leaf_start"], want["leaf_count"])
        assert got["lengths"] == want["lengths"]
        for k in range(len(levels)):
            assert got["offsets"][k].tolist() == want["offsets"][k], "offsets of level %d" % k
            assert got["validity"][k].tolist() == want["validity"][k], "validity of level %d" % k
        assert got["leaf_validity"].tolist() == want["leaf_validity"]
        assert nv == want["num_values"]
