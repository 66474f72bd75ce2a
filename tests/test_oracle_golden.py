"""Pins the CPU oracle (oracle/) against the known-answer vectors of this path: the hand-derived
byte vectors of SURVEY.md Appendix C (tests/golden/appendix_c.json), the reference's own patas
pack/unpack test (src/compression/double/patas.rs:191-202) and its codec-selection test
(src/stat.rs:228-269)."""
import json
import os

import numpy as np
import pytest

from oracle import sbo as S
from tests import gen

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "appendix_c.json")))
NP = dict(gen.NP_OF)


def build(vec):
    ptype = getattr(S, vec["ptype"])
    valid = vec.get("valid")
    validity = gen.pack_bits(valid) if valid is not None else None
    offsets = None
    if "gen" in vec:
        n = vec["n"]
        vals = np.array([int(eval(vec["gen"], {"j": j})) for j in range(n)], NP[ptype])
        rows = n
    elif ptype == S.T_BOOL:
        vals = gen.pack_bits(vec["values"])
        rows = len(vec["values"])
    elif ptype in (S.T_BIN32, S.T_BIN64):
        bs = [v.encode() for v in vec["values"]]
        offsets = np.cumsum([0] + [len(b) for b in bs]).astype(np.int32 if ptype == S.T_BIN32 else np.int64)
        vals = np.frombuffer(b"".join(bs), np.uint8)
        rows = len(bs)
    else:
        vals = np.array(vec["values"], NP[ptype])
        rows = len(vec["values"])
    force = getattr(S, vec["force"]) if "force" in vec else -1
    return S.write_column(ptype, vec["nullable"], rows, vals, validity=validity, offsets=offsets,
                          options=S.make_options(force_codec=force))


@pytest.mark.parametrize("vec", GOLD["vectors"], ids=[v["id"] for v in GOLD["vectors"]])
def test_appendix_c_vector(vec):
    data, metas = build(vec)
    got = bytes(data).hex()
    if "hex" in vec:
        assert got == vec["hex"].replace(" ", "").lower()
    else:
        assert got.startswith(vec["hex_prefix"].replace(" ", "").lower())
        assert len(data) == vec["length"]
    assert int(metas[:, 0].sum()) == len(data)


def test_patas_pack_unpack_reference_kat():
    L = S.lib()
    for a, b, c, want in GOLD["patas_pack"]:
        assert L.sbo_patas_pack(a, b, c) == want
        out = np.zeros(3, np.uint32)
        L.sbo_patas_unpack(want, out.ctypes.data)
        assert tuple(out) == (a, b, c)


def test_def_levels_65536_all_valid():
    data, _ = S.write_column(S.T_I8, True, 65536, np.zeros(65536, np.int8))
    assert bytes(data[:7]).hex() == GOLD["def_levels_65536_all_valid_prefix"].replace(" ", "").lower()
    assert (data[7:7 + 8192] == 255).all()


def test_roaring_portable_bytes_inside_freq():
    # Freq over [0,9,0,9,0...] style data: exceptions {1,3} -> the 20-byte portable bitmap of App. C (M)
    vals = np.array([5, 7, 5, 7, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5], np.uint32) + 1000
    data, _ = S.write_column(S.T_U32, False, vals.size, vals, options=S.make_options(force_codec=S.FREQ))
    want = GOLD["roaring_1_3"].replace(" ", "").lower()
    assert want in bytes(data).hex()


def test_stat_rs_codec_choice_one_value():
    # src/stat.rs:228-239: 20 480 x "a" LargeBinary, page 2048, LZ4, ratio 1.2 -> every page OneValue
    n = 20480
    vals = np.frombuffer(b"a" * n, np.uint8)
    offs = np.arange(n + 1, dtype=np.int64)
    opts = S.make_options(default_compression=S.LZ4, ratio=1.2, max_page_size=2048)
    data, metas = S.write_column(S.T_BIN64, False, n, vals, offsets=offs, options=opts)
    codecs, _ = S.stat_column(S.T_BIN64, False, data, metas)
    assert len(codecs) == 10 and (codecs == S.ONEVALUE).all()
    # :241-252 with the DICT switch -> Dict{unique_num = 1, indices OneValue}
    opts = S.make_options(default_compression=S.LZ4, ratio=1.2, max_page_size=2048, force_codec=S.DICT)
    data, metas = S.write_column(S.T_BIN64, False, n, vals, offsets=offs, options=opts)
    codecs, inner = S.stat_column(S.T_BIN64, False, data, metas)
    assert (codecs == S.DICT).all() and (inner == S.ONEVALUE).all()
    # :254-268 with the FREQ switch -> Freq (no exceptions)
    opts = S.make_options(default_compression=S.LZ4, ratio=1.2, max_page_size=2048, force_codec=S.FREQ)
    data, metas = S.write_column(S.T_BIN64, False, n, vals, offsets=offs, options=opts)
    codecs, _ = S.stat_column(S.T_BIN64, False, data, metas)
    assert (codecs == S.FREQ).all()


def test_c2_adaptive_choice_is_rle():
    """bench.py's C2 data with the bench's options (the reference's defaults: ratio 2.0, NOTHING forbidden): the
    selector prefers RLE (sampled ratio ~14) over Dict (7.6), Freq and Patas."""
    import bench
    vals, valid = bench.gen_c2_column(42)
    opts = S.make_options(default_compression=S.LZ4, ratio=2.0, max_page_size=65536)
    data, metas = S.write_column(S.T_F64, True, vals.size, vals, validity=valid, options=opts)
    codecs, _ = S.stat_column(S.T_F64, True, data, metas)
    assert (codecs == S.RLE).all()
