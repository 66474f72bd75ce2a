"""GPU parity for Zstd pages (codec id 2).  The pages are built from the oracle's None pages by
re-compressing every block with the real libzstd (through pyarrow, level 3 = what
zstd::bulk::compress_to_buffer(.., 0) uses, src/compression/basic.rs:122-135), so the device
decoder sees genuine Huffman / FSE streams; the result must equal the oracle's decode."""
import numpy as np
import pytest

from oracle import sbo as S
from tests import gen
from tests.test_gpu_decode import gpu_decode

pytestmark = pytest.mark.gpu


def recompress(col, pages, metas, level=3):
    pa = pytest.importorskip("pyarrow")
    codec = pa.Codec("zstd", compression_level=level)
    out, new_metas, off = [], [], 0
    nblocks = 2 if col["ptype"] in (S.T_BIN32, S.T_BIN64) else 1
    for length, nv in metas:
        page = bytes(pages[off:off + int(length)])
        off += int(length)
        cur, buf = 0, b""
        if col["nullable"]:
            dl = int.from_bytes(page[0:4], "little")
            buf += page[:4 + dl]
            cur = 4 + dl
        for _ in range(nblocks):
            assert page[cur] == 0
            csize = int.from_bytes(page[cur + 1:cur + 5], "little")
            usize = int.from_bytes(page[cur + 5:cur + 9], "little")
            body = page[cur + 9:cur + 9 + csize]
            z = codec.compress(body, asbytes=True)
            buf += bytes([2]) + len(z).to_bytes(4, "little") + usize.to_bytes(4, "little") + z
            cur += 9 + csize
        assert cur == len(page)
        out.append(buf)
        new_metas.append((len(buf), int(nv)))
    return np.frombuffer(b"".join(out), np.uint8).copy(), np.array(new_metas, np.uint64).reshape(-1, 2)


def check(ctx, col, level=3, **opt):
    pages, metas = gen.oracle_write(col, **opt)
    zp, zm = recompress(col, pages, metas, level)
    want = gen.oracle_read(col, zp, zm)          # the oracle's own Zstd decoder (pinned against libzstd)
    plain = gen.oracle_read(col, pages, metas)
    assert np.array_equal(want["values"], plain["values"])
    got = gpu_decode(ctx, col, zp, zm)
    assert np.array_equal(got.values_numpy(), want["values"])
    if col["nullable"]:
        assert np.array_equal(got.validity_numpy(), want["validity"])
    if col["offsets"] is not None:
        assert np.array_equal(got.offsets_numpy(), want["offsets"])


@pytest.mark.parametrize("level", [1, 3, 9])
def test_zstd_primitive_pages(gpu_ctx, level):
    check(gpu_ctx, gen.prim(S.T_I32, 30_000, uniq=50, null_density=0.1, runs=4), level, max_page_size=8192)
    check(gpu_ctx, gen.prim(S.T_F64, 30_000, uniq=1000), level, max_page_size=8192)
    check(gpu_ctx, gen.prim(S.T_I64, 20_000, uniq=1 << 40), level, max_page_size=8192)   # raw blocks
    check(gpu_ctx, gen.prim(S.T_U8, 40_000, uniq=1), level, max_page_size=16384)          # RLE blocks


def test_zstd_large_blocks(gpu_ctx):
    # pages larger than one 128 KiB zstd block: several blocks per frame, repeat offsets across blocks
    check(gpu_ctx, gen.prim(S.T_F64, 200_000, uniq=300, runs=3), max_page_size=65536)
    check(gpu_ctx, gen.prim(S.T_I32, 400_000, uniq=1 << 12, sorted_=True), max_page_size=262144)


def test_zstd_boolean_and_binary(gpu_ctx):
    check(gpu_ctx, gen.boolean(100_000, null_density=0.2, runs=5), max_page_size=32768)
    check(gpu_ctx, gen.binary(30_000, uniq=400, null_density=0.1, zipf=1.2), max_page_size=8192)
    check(gpu_ctx, gen.binary(30_000, uniq=5000, large=True), max_page_size=8192)


def test_zstd_corrupt_frame_raises(gpu_ctx):
    from strawboat_amd._native import NativeError
    col = gen.prim(S.T_I32, 5000, uniq=20, runs=3)
    pages, metas = gen.oracle_write(col)
    zp, zm = recompress(col, pages, metas)
    zp[9] ^= 0xFF   # break the magic number
    with pytest.raises(NativeError) as e:
        gpu_decode(gpu_ctx, col, zp, zm)
    assert e.value.code == -2


def test_zstd_encode_frames(gpu_ctx):
    """default_compression = Zstd on the device: frames of raw / RLE blocks, byte-identical to the
    oracle's encoder, accepted by libzstd, and decoded back by the device."""
    pa = pytest.importorskip("pyarrow")
    from tests.test_gpu_encode import gpu_encode
    for col, opt in ((gen.prim(S.T_I64, 40_000, uniq=1 << 30, null_density=0.1), dict(max_page_size=8192)),
                     (gen.prim(S.T_U8, 300_000, uniq=1), dict(max_page_size=262144)),
                     (gen.boolean(50_000, runs=4), dict(max_page_size=8192)),
                     (gen.binary(20_000, uniq=100, zipf=1.3), dict(max_page_size=4096)),
                     (gen.prim(S.T_F64, 20_000, uniq=30, runs=5), dict(max_page_size=4096, force_codec=S.DICT))):
        want_pages, want_metas = gen.oracle_write(col, default_compression=S.ZSTD, **opt)
        enc = gpu_encode(gpu_ctx, col, default_compression=S.ZSTD, **opt)
        assert np.array_equal(enc.metas_array(), want_metas)
        assert np.array_equal(enc.pages_numpy(), want_pages)
        got = gpu_decode(gpu_ctx, col, enc.pages_numpy(), enc.metas_array())
        want = gen.oracle_read(col, want_pages, want_metas)
        assert np.array_equal(got.values_numpy(), want["values"])
    # libzstd accepts the frame of a primitive page
    col = gen.prim(S.T_I32, 5000, uniq=9)
    enc = gpu_encode(gpu_ctx, col, default_compression=S.ZSTD)
    page = enc.pages_numpy().tobytes()
    assert page[0] == 2
    csize = int.from_bytes(page[1:5], "little")
    raw = pa.Codec("zstd").decompress(page[9:9 + csize], decompressed_size=col["rows"] * 4).to_pybytes()
    assert raw == col["values"].tobytes()
