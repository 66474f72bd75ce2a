"""Round-trip properties of the CPU oracle on the workload shapes of the reference's integration
tests (tests/it/io.rs:72-278, 417-438): every chunk goes through the 4 default compressions with
max_page_size = 2048 and default_compress_ratio = 2.0 and must read back logically equal."""
import numpy as np
import pytest

from oracle import sbo as S
from tests import gen

WRITE_PAGE = 2048
DEFAULTS = [S.LZ4, S.ZSTD, S.SNAPPY, S.NONE]


def logical_equal(col, out):
    rows = col["rows"]
    assert out["rows"] == rows
    valid = np.ones(rows, bool)
    if col["validity"] is not None:
        valid = np.unpackbits(col["validity"], bitorder="little")[:rows].astype(bool)
    if col["nullable"]:
        got_valid = np.unpackbits(out["validity"], bitorder="little")[:rows].astype(bool)
        assert np.array_equal(got_valid, valid)
    t = col["ptype"]
    if t == S.T_BOOL:
        a = np.unpackbits(col["values"], bitorder="little")[:rows]
        b = np.unpackbits(out["values"], bitorder="little")[:rows]
        assert np.array_equal(a[valid], b[valid])
    elif t in (S.T_BIN32, S.T_BIN64):
        dt = np.int32 if t == S.T_BIN32 else np.int64
        o0, o1 = col["offsets"].astype(np.int64), out["offsets"].view(dt).astype(np.int64)
        assert o1.size == rows + 1
        src, dst = col["values"].tobytes(), out["values"].tobytes()
        for i in np.nonzero(valid)[0][:: max(1, rows // 500)]:
            assert src[o0[i]:o0[i + 1]] == dst[o1[i]:o1[i + 1]]
    else:
        w = S.WIDTH[t]
        a = np.ascontiguousarray(col["values"]).view(np.uint8).reshape(rows, w)
        b = out["values"].reshape(rows, w)
        if t in (S.T_F32, S.T_F64):   # float RLE keeps the first value of a run: compare as numbers
            fa = a.view(gen.NP_OF[t]).reshape(-1)
            fb = b.view(gen.NP_OF[t]).reshape(-1)
            assert np.array_equal(fa[valid], fb[valid])
        else:
            assert np.array_equal(a[valid], b[valid])


def roundtrip(col, **opt):
    pages, metas = gen.oracle_write(col, **opt)
    out = gen.oracle_read(col, pages, metas)
    logical_equal(col, out)
    return pages, metas


SHAPES = {
    "random": lambda: [gen.boolean(10000, null_density=0.3), gen.prim(S.T_I32, 10000, null_density=0.2),
                       gen.prim(S.T_F64, 10000, null_density=0.5), gen.binary(10000, null_density=0.1, large=True)],
    "random_nonull": lambda: [gen.boolean(10000), gen.prim(S.T_I32, 10000), gen.prim(S.T_F64, 10000),
                              gen.binary(10000, large=True)],
    "dict": lambda: [gen.prim(S.T_I64, 10000, uniq=8), gen.prim(S.T_F32, 10000, uniq=8), gen.binary(10000, uniq=8)],
    "freq": lambda: [dict(ptype=S.T_U32, nullable=False, rows=2048 * 5,
                          values=np.tile(np.concatenate([np.full(2045, 20), np.full(3, 10000)]), 5).astype(np.uint32),
                          validity=None, offsets=None)],
    "bitpacking": lambda: [gen.prim(S.T_U32, 2048 * 5, uniq=8), gen.prim(S.T_I32, 2048 * 5, uniq=8)],
    "delta_bitpacking": lambda: [dict(ptype=S.T_U32, nullable=False, rows=10240, values=np.arange(10240, dtype=np.uint32),
                                      validity=None, offsets=None),
                                 dict(ptype=S.T_I32, nullable=False, rows=10240, values=np.arange(10240, dtype=np.int32),
                                      validity=None, offsets=None)],
    "onevalue": lambda: [gen.prim(S.T_I16, 10000, uniq=1), gen.boolean(10000, p_true=1.0), gen.binary(10000, uniq=1)],
    "float": lambda: [gen.prim(S.T_F32, 10000, null_density=0.1, runs=5), gen.prim(S.T_F64, 10000, runs=20)],
    "wide": lambda: [gen.prim(S.T_I128, 5000, null_density=0.1), gen.prim(S.T_I256, 5000), gen.prim(S.T_U8, 5000),
                     gen.prim(S.T_U64, 5000, uniq=1 << 40)],
}


@pytest.mark.parametrize("default", DEFAULTS)
@pytest.mark.parametrize("shape", sorted(SHAPES))
def test_write_read(shape, default):
    for col in SHAPES[shape]():
        roundtrip(col, default_compression=default, ratio=2.0, max_page_size=WRITE_PAGE)
        roundtrip(col, default_compression=default, max_page_size=WRITE_PAGE)  # ratio None => Basic


def test_basic_six_rows():
    # tests/it/io.rs:48-70 (a 6-row chunk of many types, some with nulls)
    valid = gen.pack_bits([1, 0, 1, 1, 0, 1])
    for t in (S.T_I8, S.T_I16, S.T_I32, S.T_I64, S.T_U8, S.T_U16, S.T_U32, S.T_U64, S.T_F32, S.T_F64):
        vals = np.array([1, 2, 3, 4, 5, 6], gen.NP_OF[t])
        col = dict(ptype=t, nullable=True, rows=6, values=vals, validity=valid, offsets=None)
        roundtrip(col)
        roundtrip(col, default_compression=S.LZ4, ratio=2.0)


def test_forced_codecs_selection_and_sizes():
    col = gen.prim(S.T_I32, 2048 * 4, uniq=8)
    p_none, _ = roundtrip(col, max_page_size=2048)
    p_bp, m = roundtrip(col, max_page_size=2048, default_compression=S.LZ4, ratio=2.0)
    codecs, _ = S.stat_column(S.T_I32, False, p_bp, m)
    assert set(codecs.tolist()) <= {S.BITPACK, S.DICT, S.RLE}
    assert len(p_bp) < len(p_none) / 2
    srt = dict(col, values=np.sort(col["values"]))
    p, m = roundtrip(srt, max_page_size=2048, default_compression=S.LZ4, ratio=2.0, forbidden=(S.RLE, S.DICT))
    codecs, _ = S.stat_column(S.T_I32, False, p, m)
    assert S.DELTABP in codecs.tolist()


def test_sampling_is_seeded():
    col = gen.prim(S.T_I32, 128 * 512, uniq=300, runs=3)
    a, _ = gen.oracle_write(col, max_page_size=65536, ratio=1.1, default_compression=S.LZ4, rng_seed=1)
    b, _ = gen.oracle_write(col, max_page_size=65536, ratio=1.1, default_compression=S.LZ4, rng_seed=1)
    assert np.array_equal(a, b)


def test_errors():
    col = gen.prim(S.T_I32, 1000, uniq=10)
    pages, metas = gen.oracle_write(col)
    bad = pages.copy()
    bad[0] = 77
    with pytest.raises(S.OracleError, match="Unknown compression codec"):
        gen.oracle_read(col, bad, metas)
    with pytest.raises(S.OracleError):
        gen.oracle_read(col, pages[:-10], metas)
    with pytest.raises(S.OracleError):   # Bitpacking needs whole 128-blocks
        gen.oracle_write(gen.prim(S.T_I32, 1000), force_codec=S.BITPACK)
