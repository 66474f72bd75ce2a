"""Pins the oracle's Dremel level arithmetic and its hybrid-RLE bit packing against an INDEPENDENT implementation:
tests/golden/parquet_levels.json holds the repetition / definition level values parquet-cpp (pyarrow.parquet, data
page V2) wrote for the same arrays (generator: tests/golden/make_parquet_levels.py).  The oracle packs each stream as
one bit-packed run (arrow2's encoder, reference call sites src/write/serialize.rs:200-232); unpacking that run must
give the same VALUES parquet-cpp encoded with its own RLE / bit-packed mix."""
import json
import os

import numpy as np
import pytest

from oracle import sbo as S

FIX = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "parquet_levels.json")))
CASES = FIX["cases"]


def uleb(buf, p):
    out, sh = 0, 0
    while True:
        v = int(buf[p])
        p += 1
        out |= (v & 0x7F) << sh
        sh += 7
        if not v & 0x80:
            return out, p


def unpack_bitpacked_run(buf, bit_width, count):
    """one bit-packed hybrid run: ULEB128((groups << 1) | 1) | groups * bit_width bytes (read_basic.rs:83-84 bit width)"""
    if bit_width == 0:
        assert len(buf) == 0
        return [0] * count
    h, p = uleb(buf, 0)
    assert h & 1, "the oracle (like arrow2) writes bit-packed runs only"
    groups = h >> 1
    assert groups == (count + 7) // 8
    # arrow2's encoder writes ceil(count * bit_width / 8) bytes: the last group of 8 is cut at the byte that holds the
    # last value; a reader that wants whole groups sees the missing bits as zero
    assert len(buf) - p == (count * bit_width + 7) // 8
    bits = int.from_bytes(bytes(buf[p:]), "little")
    mask = (1 << bit_width) - 1
    return [(bits >> (i * bit_width)) & mask for i in range(count)]


def levels_from_fixture(case):
    lv = []
    for d in case["levels"]:
        e = dict(kind=d["kind"], is_optional=d["is_optional"], length=d["length"])
        if d["validity"] is not None:
            e["validity"] = np.frombuffer(bytes.fromhex(d["validity"]), np.uint8)
        if d["offsets"] is not None:
            e["offsets"] = np.array(d["offsets"], np.int64 if d["kind"] == S.K_LARGE_LIST else np.int32)
        lv.append(e)
    return lv


@pytest.mark.parametrize("i", [k for k, c in enumerate(CASES) if c["kind"] == "nested"])
def test_nested_level_values_equal_parquet_cpp(i):
    case = CASES[i]
    pq = case["parquet"]
    levels = levels_from_fixture(case)
    b, nv, ls, lc = S.nested_write_levels(levels, 0, case["rows"])
    b = np.asarray(b)
    rows, rep_len, def_len = (int.from_bytes(bytes(b[k:k + 4]), "little") for k in (0, 4, 8))
    assert rows == case["rows"] and 12 + rep_len + def_len == len(b)
    assert nv == pq["num_values"], "number of level entries"
    wbits = lambda mx: 0 if mx == 0 else int(mx).bit_length()   # get_bit_width
    rep = unpack_bitpacked_run(b[12:12 + rep_len], wbits(pq["max_rep"]), nv)
    deff = unpack_bitpacked_run(b[12 + rep_len:], wbits(pq["max_def"]), nv)
    assert rep == pq["rep"], "%s rows=%d: repetition levels differ from parquet-cpp's" % (case["shape"], case["rows"])
    assert deff == pq["def"], "%s rows=%d: definition levels differ from parquet-cpp's" % (case["shape"], case["rows"])


@pytest.mark.parametrize("i", [k for k, c in enumerate(CASES) if c["kind"] == "flat"])
def test_flat_def_levels_equal_parquet_cpp(i):
    case = CASES[i]
    rows = case["rows"]
    validity = np.frombuffer(bytes.fromhex(case["validity"]), np.uint8)
    pages, metas = S.write_column(S.T_I64, True, rows, np.arange(rows, dtype=np.int64), validity=validity,
                                  options=S.make_options(force_codec=S.NONE))
    def_len = int.from_bytes(bytes(pages[:4]), "little")
    got = unpack_bitpacked_run(pages[4:4 + def_len], 1, rows)
    assert got == case["parquet"]["def"]
    assert case["parquet"]["max_def"] == 1 and case["parquet"]["max_rep"] == 0
