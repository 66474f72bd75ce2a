"""Page inspector (src/stat.rs): sb_stat_page / strawboat_amd.stat against pages written by the oracle.
Mirrors the shapes of the reference's own tests (stat.rs:228-269: a column of identical strings picks
OneValue / Dict / Freq depending on the forced codec) on the CPU: the inspector is host-only code."""
import numpy as np
import pytest

from oracle import sbo as S
from tests import gen
from strawboat_amd import stat
from strawboat_amd._native import NativeError


def info_of(col, **opt):
    pages, metas = gen.oracle_write(col, **opt)
    ci = stat.stat_simple(pages, metas, col["ptype"], col["nullable"])
    codecs, inner = S.stat_column(col["ptype"], col["nullable"], pages, metas)
    assert [p.codec for p in ci.pages] == codecs.tolist()
    return ci, pages, metas, inner


def test_plain_and_basic_codecs():
    col = gen.prim(S.T_I64, 5000, uniq=100)
    ci, pages, metas, _ = info_of(col, max_page_size=2048)
    assert [p.body.kind for p in ci.pages] == ["Common"] * 3 and ci.pages[0].body.common == S.NONE
    assert [p.compressed_size for p in ci.pages] == [2048 * 8, 2048 * 8, 904 * 8]
    assert all(p.uncompressed_size == p.compressed_size and p.validity_size is None for p in ci.pages)
    ci, *_ = info_of(col, max_page_size=2048, default_compression=S.LZ4)
    assert ci.pages[0].body.common == S.LZ4 and ci.pages[0].uncompressed_size == 2048 * 8


def test_dict_body_and_nested_indices():
    col = gen.prim(S.T_I32, 128 * 40, uniq=37, null_density=0.1)
    ci, pages, metas, inner = info_of(col, max_page_size=128 * 20, force_codec=S.DICT, force_index_codec=S.BITPACK)
    for p, ic in zip(ci.pages, inner.tolist()):
        assert p.body.kind == "Dict" and p.body.indices.codec == ic == S.BITPACK and p.body.indices.body.kind == "Bitpack"
        assert p.body.unique_num == 37 and p.body.indices.validity_size is None
        # upstream reports the u32 that FOLLOWS the def-level section (stat.rs:72-76): codec byte + 3 size bytes
        assert p.validity_size == (S.DICT | (p.compressed_size & 0xFFFFFF) << 8)


def test_freq_bodies():
    v = np.full(6000, 7, np.int64)
    v[::50] = np.arange(120) + 1000
    col = dict(ptype=S.T_I64, nullable=False, rows=v.size, values=v, validity=None, offsets=None)
    ci, pages, metas, inner = info_of(col, max_page_size=3000, force_codec=S.FREQ)
    for p in ci.pages:
        assert p.body.kind == "Freq" and p.body.exceptions is not None and p.body.exceptions_bitmap_size > 0
        assert p.body.exceptions.uncompressed_size == 60 * 8
    col = gen.binary(4000, uniq=2, zipf=3.0, seed=3)
    ci, *_ = info_of(col, max_page_size=2000, force_codec=S.FREQ)
    assert all(p.body.kind == "Freq" and p.body.exceptions is None for p in ci.pages)  # binary: plain exceptions


def test_identical_strings_like_the_reference_test():
    # stat.rs:228-269: 20 480 x "a" with LZ4 + ratio 1.2 -> OneValue; Dict / Freq when forced
    rows = 20480
    col = dict(ptype=S.T_BIN32, nullable=False, rows=rows, values=np.full(rows, ord("a"), np.uint8), validity=None,
               offsets=np.arange(rows + 1, dtype=np.int32))
    ci, *_ = info_of(col, max_page_size=2048, default_compression=S.LZ4, ratio=1.2)
    assert len(ci.pages) == 10 and all(p.body.kind == "OneValue" for p in ci.pages)
    ci, *_ = info_of(col, max_page_size=2048, default_compression=S.LZ4, ratio=1.2, force_codec=S.DICT)
    assert all(p.body.kind == "Dict" and p.body.unique_num == 1 for p in ci.pages)
    ci, *_ = info_of(col, max_page_size=2048, default_compression=S.LZ4, ratio=1.2, force_codec=S.FREQ)
    assert all(p.body.kind == "Freq" for p in ci.pages)


def test_other_variants_and_errors():
    kinds = {}
    for codec, name in ((S.RLE, "Rle"), (S.ONEVALUE, "OneValue"), (S.PATAS, "Patas"), (S.DELTABP, "DeltaBitpack")):
        if codec == S.PATAS:
            col = gen.prim(S.T_F64, 1000, uniq=10)
        elif codec == S.DELTABP:
            col = gen.prim(S.T_U32, 128 * 8, uniq=5000, sorted_=True)
        else:
            col = gen.prim(S.T_I32, 1000, uniq=1)
        ci, pages, metas, _ = info_of(col, force_codec=codec)
        assert ci.pages[0].body.kind == name
        kinds[name] = (pages, metas, col)
    pages, metas, col = kinds["Rle"]
    with pytest.raises(NativeError) as e:   # truncated page: a slice panic upstream
        stat.stat_page(pages[:int(metas[0, 0]) - 3], col["ptype"], col["nullable"])
    assert e.value.code == -3
    bad = pages.copy()
    bad[0] = 9                               # unknown codec id: Compression::from_codec errors (compression/mod.rs:78-80)
    with pytest.raises(NativeError) as e:
        stat.stat_page(bad, col["ptype"], col["nullable"])
    assert e.value.code == -1
