"""LZ4 / Zstd / Snappy block codecs of the oracle against bytes produced by the real third-party
libraries (fixtures made by tests/golden/make_block_fixtures.py with liblz4 1.9.3 and pyarrow)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from oracle import sbo as S

DIR = os.path.join(os.path.dirname(__file__), "golden", "blocks")
INDEX = json.load(open(os.path.join(DIR, "index.json")))
CASES = [c["name"] for c in INDEX["cases"]]


def raw(name):
    return np.fromfile(os.path.join(DIR, name + ".raw"), np.uint8)


@pytest.mark.parametrize("name", CASES)
def test_lz4_compress_bytes_equal_liblz4(name):
    a = raw(name)
    want = np.fromfile(os.path.join(DIR, name + ".lz4"), np.uint8)
    got = S.block_compress(S.LZ4, a)
    assert np.array_equal(got, want)  # restatement of LZ4_compress_default is byte-exact


@pytest.mark.parametrize("name", CASES)
def test_lz4_decompress_liblz4_stream(name):
    a = raw(name)
    comp = np.fromfile(os.path.join(DIR, name + ".lz4"), np.uint8)
    assert np.array_equal(S.block_decompress(S.LZ4, comp, a.size), a)


@pytest.mark.parametrize("level", [1, 3])
@pytest.mark.parametrize("name", CASES)
def test_zstd_decompress_libzstd_frame(name, level):
    a = raw(name)
    comp = np.fromfile(os.path.join(DIR, "%s.zstd%d" % (name, level)), np.uint8)
    assert np.array_equal(S.block_decompress(S.ZSTD, comp, a.size), a)


@pytest.mark.parametrize("name", CASES)
def test_snappy_decompress_snappy_stream(name):
    a = raw(name)
    comp = np.fromfile(os.path.join(DIR, name + ".snappy"), np.uint8)
    assert np.array_equal(S.block_decompress(S.SNAPPY, comp, a.size), a)


@pytest.mark.parametrize("codec", [S.ZSTD, S.SNAPPY])
@pytest.mark.parametrize("name", CASES)
def test_own_frames_round_trip(name, codec):
    a = raw(name)
    assert np.array_equal(S.block_decompress(codec, S.block_compress(codec, a), a.size), a)


def test_own_frames_accepted_by_third_party_decoders():
    pa = pytest.importorskip("pyarrow")
    for name in CASES:
        a = raw(name)
        if a.size == 0:
            continue
        z = S.block_compress(S.ZSTD, a).tobytes()
        assert pa.Codec("zstd").decompress(z, decompressed_size=a.size).to_pybytes() == a.tobytes()
        s = S.block_compress(S.SNAPPY, a).tobytes()
        assert pa.Codec("snappy").decompress(s, decompressed_size=a.size).to_pybytes() == a.tobytes()


def test_lz4_live_cross_check_against_system_liblz4():
    try:
        lz = C.CDLL("liblz4.so.1")
    except OSError:
        pytest.skip("no system liblz4")
    lz.LZ4_compress_default.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    lz.LZ4_compressBound.argtypes = [C.c_int]
    rng = np.random.default_rng(3)
    for n in (100, 65546, 65547, 200000):
        a = np.repeat(rng.integers(0, 20, n // 5 + 1), 5)[:n].astype(np.uint8)
        cap = lz.LZ4_compressBound(a.size)
        dst = np.zeros(cap, np.uint8)
        k = lz.LZ4_compress_default(a.ctypes.data, dst.ctypes.data, a.size, cap)
        assert np.array_equal(S.block_compress(S.LZ4, a), dst[:k])


def test_corrupt_streams_raise():
    a = raw("i32_runs")
    comp = np.fromfile(os.path.join(DIR, "i32_runs.lz4"), np.uint8).copy()
    with pytest.raises(S.OracleError):
        S.block_decompress(S.LZ4, comp[:-3], a.size)
    z = np.fromfile(os.path.join(DIR, "i32_runs.zstd3"), np.uint8).copy()
    z[0] ^= 0xFF
    with pytest.raises(S.OracleError):
        S.block_decompress(S.ZSTD, z, a.size)
