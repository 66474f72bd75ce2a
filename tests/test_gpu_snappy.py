"""GPU parity for Snappy pages (codec id 3, src/compression/basic.rs:99-106,137-152).
Decode: pages are built from the oracle's None pages by re-compressing every block with the real
libsnappy (through pyarrow), so the device decoder sees genuine copy-1/2/4 elements; the result must
equal the oracle's decode.  Encode: the device writes literal-only Snappy streams — format-valid
(libsnappy and the oracle read them back), not byte-identical to the `snap` crate (unpinned upstream)."""
import numpy as np
import pytest

from oracle import sbo as S
from tests import gen
from tests.test_gpu_decode import gpu_decode
from tests.test_gpu_encode import gpu_encode

pytestmark = pytest.mark.gpu


def recompress(col, pages, metas):
    pa = pytest.importorskip("pyarrow")
    codec = pa.Codec("snappy")
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
            z = codec.compress(page[cur + 9:cur + 9 + csize], asbytes=True)
            buf += bytes([3]) + len(z).to_bytes(4, "little") + usize.to_bytes(4, "little") + z
            cur += 9 + csize
        assert cur == len(page)
        out.append(buf)
        new_metas.append((len(buf), int(nv)))
    return np.frombuffer(b"".join(out), np.uint8).copy(), np.array(new_metas, np.uint64).reshape(-1, 2)


def check_decode(ctx, col, **opt):
    pages, metas = gen.oracle_write(col, **opt)
    zp, zm = recompress(col, pages, metas)
    want = gen.oracle_read(col, zp, zm)
    got = gpu_decode(ctx, col, zp, zm)
    assert np.array_equal(got.values_numpy(), want["values"])
    if col["nullable"]:
        assert np.array_equal(got.validity_numpy(), want["validity"])
    if col["offsets"] is not None:
        assert np.array_equal(got.offsets_numpy(), want["offsets"])


def test_snappy_decode(gpu_ctx):
    check_decode(gpu_ctx, gen.prim(S.T_I32, 30_000, uniq=50, null_density=0.1, runs=4), max_page_size=8192)
    check_decode(gpu_ctx, gen.prim(S.T_F64, 200_000, uniq=300, runs=3), max_page_size=65536)   # offsets > 64 Ki: copy-4
    check_decode(gpu_ctx, gen.prim(S.T_I64, 20_000, uniq=1 << 40), max_page_size=8192)          # incompressible: long literals
    check_decode(gpu_ctx, gen.prim(S.T_U8, 40_000, uniq=1), max_page_size=16384)                # overlapping copies
    check_decode(gpu_ctx, gen.boolean(100_000, null_density=0.2, runs=5), max_page_size=32768)
    check_decode(gpu_ctx, gen.binary(30_000, uniq=400, null_density=0.1, zipf=1.2), max_page_size=8192)
    # oracle-written Snappy pages (its own greedy matcher) incl. Dict pages whose indices are Snappy blocks
    col = gen.prim(S.T_I64, 20_000, uniq=100, runs=2)
    for opt in (dict(default_compression=S.SNAPPY), dict(default_compression=S.SNAPPY, force_codec=S.DICT)):
        pages, metas = gen.oracle_write(col, max_page_size=4096, **opt)
        got = gpu_decode(gpu_ctx, col, pages, metas)
        assert np.array_equal(got.values_numpy(), gen.oracle_read(col, pages, metas)["values"])


def test_snappy_encode_is_readable(gpu_ctx):
    pa = pytest.importorskip("pyarrow")
    for col in (gen.prim(S.T_I32, 20_000, uniq=50, null_density=0.1, runs=4), gen.boolean(20_000, runs=7),
                gen.binary(10_000, uniq=100, null_density=0.2), gen.prim(S.T_U16, 50, uniq=3)):
        enc = gpu_encode(gpu_ctx, col, max_page_size=4096, default_compression=S.SNAPPY)
        pages, metas = enc.pages_numpy(), enc.metas_array()
        codecs, _ = S.stat_column(col["ptype"], col["nullable"], pages, metas)
        assert (codecs == S.SNAPPY).all()
        want = gen.oracle_read(col, pages, metas)                      # the oracle's decoder accepts the stream
        plain = gen.oracle_read(col, *gen.oracle_write(col, max_page_size=4096))
        assert np.array_equal(want["values"], plain["values"])
        got = gpu_decode(gpu_ctx, col, pages, metas)                   # and so does the device's
        assert np.array_equal(got.values_numpy(), plain["values"])
    # libsnappy reads a device-written block
    col = gen.prim(S.T_I64, 1000, uniq=5)
    enc = gpu_encode(gpu_ctx, col, default_compression=S.SNAPPY)
    page = bytes(enc.pages_numpy())
    csize = int.from_bytes(page[1:5], "little")
    raw = pa.Codec("snappy").decompress(page[9:9 + csize], decompressed_size=8000, asbytes=True)
    assert raw == col["values"].tobytes()
