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


def _zstd_frames_of_page(page, ptype, nullable, rows):
    """(frame bytes, decompressed size) of every Basic(Zstd) block of one page"""
    pos = 0
    if nullable:
        pos = 4 + int.from_bytes(page[0:4], "little")
    out = []
    nblocks = 2 if ptype in (S.T_BIN32, S.T_BIN64) else 1
    for _ in range(nblocks):
        assert page[pos] == S.ZSTD
        csize = int.from_bytes(page[pos + 1:pos + 5], "little")
        usize = int.from_bytes(page[pos + 5:pos + 9], "little")
        out.append((bytes(page[pos + 9:pos + 9 + csize]), usize))
        pos += 9 + csize
    return out


SHAPES_Z = [("i64_random", lambda: gen.prim(S.T_I64, 40_000, uniq=1 << 30, null_density=0.1), dict(max_page_size=8192)),
            ("u8_constant", lambda: gen.prim(S.T_U8, 300_000, uniq=1), dict(max_page_size=262144)),
            ("bool_runs", lambda: gen.boolean(50_000, runs=4), dict(max_page_size=8192)),
            ("utf8_zipf", lambda: gen.binary(20_000, uniq=100, zipf=1.3), dict(max_page_size=4096)),
            ("utf8_big_pages", lambda: gen.binary(200_000, uniq=3000, zipf=1.1, maxlen=24), dict(max_page_size=65536)),
            ("i32_small_values", lambda: gen.prim(S.T_I32, 100_000, uniq=100), dict(max_page_size=65536)),
            ("i64_runs", lambda: gen.prim(S.T_I64, 100_000, uniq=50, runs=9), dict(max_page_size=32768)),
            ("f64_dict_indices", lambda: gen.prim(S.T_F64, 20_000, uniq=30, runs=5), dict(max_page_size=4096, force_codec=S.DICT)),
            ("tiny", lambda: gen.prim(S.T_I32, 5, uniq=3), dict()),
            ("one_row", lambda: gen.prim(S.T_I64, 1, uniq=3), dict())]


@pytest.mark.parametrize("name", [s[0] for s in SHAPES_Z])
def test_zstd_encode_frames(gpu_ctx, name):
    """default_compression = Zstd on the device: real Zstd frames (LDS matcher, Huffman literals, FSE sequences with
    the predefined tables).  Format-valid, not libzstd's bytes: the oracle's decoder, libzstd (pyarrow,
    where importable) and the device's decoder read them back to the input; page structure as the oracle writes it."""
    from tests.test_gpu_encode import gpu_encode
    mk, opt = {n: (m, o) for n, m, o in SHAPES_Z}[name]
    col = mk()
    want_pages, want_metas = gen.oracle_write(col, default_compression=S.ZSTD, **opt)
    enc = gpu_encode(gpu_ctx, col, default_compression=S.ZSTD, **opt)
    pages, metas = enc.pages_numpy(), enc.metas_array()
    assert np.array_equal(metas[:, 1], want_metas[:, 1])
    want = gen.oracle_read(col, want_pages, want_metas)
    got = gen.oracle_read(col, pages, metas)                       # the oracle's RFC 8878 decoder
    for k in ("values", "validity", "offsets"):
        assert np.array_equal(got[k], want[k]), k
    assert np.array_equal(S.stat_column(col["ptype"], col["nullable"], pages, metas)[0],
                          S.stat_column(col["ptype"], col["nullable"], want_pages, want_metas)[0])
    back = gpu_decode(gpu_ctx, col, pages, metas)                   # the device's decoder
    assert np.array_equal(back.values_numpy(), want["values"])
    if opt.get("force_codec") is None:
        try:
            import pyarrow as pa
        except ImportError:
            pa = None
        off = 0
        total_c = total_u = 0
        for length, rows in metas:
            page = pages[off:off + int(length)]
            for frame, usize in _zstd_frames_of_page(page, col["ptype"], col["nullable"], int(rows)):
                if pa is not None and col["ptype"] != S.T_BOOL:
                    raw = pa.Codec("zstd").decompress(frame, decompressed_size=usize).to_pybytes()   # libzstd
                    assert len(raw) == usize
                total_c += len(frame)
                total_u += usize if col["ptype"] != S.T_BOOL else (usize + 7) // 8
            off += int(length)
        # never larger than stored + framing; compressible shapes must actually compress
        assert total_c <= total_u + 16 * len(metas) * 2 + 12 * (total_u // 16384 + len(metas) * 2)
        if name in ("u8_constant", "utf8_zipf", "utf8_big_pages", "i32_small_values", "i64_runs"):
            assert total_c < 0.7 * total_u, (name, total_c, total_u)


def test_zstd_ratio_on_golden_blocks(gpu_ctx):
    """the golden inputs of tests/golden/blocks: device frame size next to libzstd level 3's (reported; the device uses
    the predefined FSE tables and direct Huffman weights only, so small highly repetitive inputs stay behind)"""
    import json
    import os
    from tests.test_gpu_encode import gpu_encode
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "blocks")
    rows = []
    for case in json.load(open(os.path.join(d, "index.json")))["cases"]:
        raw = np.fromfile(os.path.join(d, case["name"] + ".raw"), np.uint8)
        if raw.size == 0:
            continue
        ref = os.path.getsize(os.path.join(d, case["name"] + ".zstd3"))
        col = dict(ptype=S.T_U8, nullable=False, rows=int(raw.size), values=raw, validity=None, offsets=None)
        enc = gpu_encode(gpu_ctx, col, default_compression=S.ZSTD)
        page = enc.pages_numpy()
        csize = int.from_bytes(bytes(page[1:5]), "little")
        got = gen.oracle_read(col, page, enc.metas_array())
        assert np.array_equal(got["values"], raw), case["name"]
        rows.append((case["name"], raw.size, csize, ref))
        assert csize <= raw.size + 16
    print("\n".join("%-16s raw %6d  device %6d  libzstd-3 %6d  x%.2f" % (n, r, c, z, c / z) for n, r, c, z in rows))
    # compressible inputs compress
    by = {n: (r, c, z) for n, r, c, z in rows}
    for n in ("abcd_x500", "i32_runs", "low_entropy_8k", "zipf_words", "zeros_70k", "f64_small_set"):
        assert by[n][1] < 0.75 * by[n][0], (n, by[n])


@pytest.mark.parametrize("kind", ["zeros", "text", "random"])
@pytest.mark.parametrize("n", [16384 + 1, 16384 + 31, 16384 + 32, 2 * 16384, 32768 + 1, 2 * 32768 + 5, 5 * 32768 - 3, 300_007])
def test_zstd_frames_of_parallel_blocks(gpu_ctx, kind, n):
    """buffers of more than 16 KiB: every 16 KiB piece is compressed by a wave of its own (k_enc_zstd_chunks) into a FRAME of
    its own, and the frames are written back to back — one valid buffer for ZSTD_decompress ("any number of frames
    concatenated"), which is what the reference's zstd::bulk::decompress_to_buffer calls; libzstd (pyarrow), the oracle and
    the device (one job per frame, or the whole buffer by one wave) must read it back, whatever the size of the last piece"""
    pa = pytest.importorskip("pyarrow")
    from tests.test_gpu_encode import gpu_encode
    rng = np.random.default_rng(n % 997)
    if kind == "zeros":
        data = np.zeros(n, np.uint8)
    elif kind == "text":
        words = [b"w%d" % i + b"y" * (i % 7) for i in range(200)]
        data = np.frombuffer(b"".join(words[i] for i in rng.zipf(1.3, n // 3) % 200)[:n].ljust(n, b"."), np.uint8)
    else:
        data = rng.integers(0, 256, n, dtype=np.uint8)
    col = dict(ptype=S.T_U8, nullable=False, rows=n, values=data, validity=None, offsets=None)
    enc = gpu_encode(gpu_ctx, col, default_compression=S.ZSTD)
    pages, metas = enc.pages_numpy(), enc.metas_array()
    assert metas.shape[0] == 1 and pages[0] == S.ZSTD
    csize = int.from_bytes(bytes(pages[1:5]), "little")
    assert csize == int(metas[0, 0]) - 9
    raw = pa.Codec("zstd").decompress(bytes(pages[9:9 + csize]), decompressed_size=n).to_pybytes()
    assert raw == bytes(data)
    assert np.array_equal(gen.oracle_read(col, pages, metas)["values"], data)
    assert np.array_equal(gpu_decode(gpu_ctx, col, pages, metas).values_numpy(), data)
    assert csize <= n + 16 + 12 * (n // 16384 + 1)     # (frame header + block header per 16 KiB piece)
    if kind != "random":
        assert csize < 0.5 * n


def test_concatenated_libzstd_frames(gpu_ctx):
    """a Zstd payload made of several libzstd frames back to back (valid input for ZSTD_decompress, which the reference's
    zstd::bulk::decompress_to_buffer ends in): the device queues one job per frame when every frame states its content
    size, and decodes the payload frame after frame on one wave otherwise; both must give the oracle's bytes"""
    pa = pytest.importorskip("pyarrow")
    rng = np.random.default_rng(7)
    data = np.repeat(rng.integers(0, 50, 40_000), rng.integers(1, 6, 40_000)).astype(np.int64)[:100_000]
    raw = data.tobytes()
    cuts = [0, 70_000, 70_008, 300_000, len(raw)]
    codec = pa.Codec("zstd", compression_level=3)
    frames = b"".join(codec.compress(raw[a:b], asbytes=True) for a, b in zip(cuts[:-1], cuts[1:]))
    page = bytes([S.ZSTD]) + len(frames).to_bytes(4, "little") + len(raw).to_bytes(4, "little") + frames
    pages = np.frombuffer(page, np.uint8).copy()
    metas = np.array([[len(page), data.size]], np.uint64)
    col = dict(ptype=S.T_I64, nullable=False, rows=data.size, values=data, validity=None, offsets=None)
    assert np.array_equal(np.asarray(gen.oracle_read(col, pages, metas)["values"]).view(np.int64), data)
    assert np.array_equal(gpu_decode(gpu_ctx, col, pages, metas).values_numpy().view(np.int64), data)
    # a skippable frame in front: not split (no content size to place the frames by), decoded serially
    skip = (0x184D2A50).to_bytes(4, "little") + (5).to_bytes(4, "little") + b"hello"
    payload = skip + frames
    page = bytes([S.ZSTD]) + len(payload).to_bytes(4, "little") + len(raw).to_bytes(4, "little") + payload
    pages = np.frombuffer(page, np.uint8).copy()
    metas = np.array([[len(page), data.size]], np.uint64)
    assert np.array_equal(np.asarray(gen.oracle_read(col, pages, metas)["values"]).view(np.int64), data)
    assert np.array_equal(gpu_decode(gpu_ctx, col, pages, metas).values_numpy().view(np.int64), data)
    # a truncated last frame must be refused, not split into garbage
    from strawboat_amd._native import NativeError
    bad = frames[:-7]
    page = bytes([S.ZSTD]) + len(bad).to_bytes(4, "little") + len(raw).to_bytes(4, "little") + bad
    pages = np.frombuffer(page, np.uint8).copy()
    metas = np.array([[len(page), data.size]], np.uint64)
    with pytest.raises(NativeError):
        gpu_decode(gpu_ctx, col, pages, metas)


def test_batch_of_frames_takes_the_lane_per_frame_sequence_decoder(gpu_ctx):
    """A read with >= 4 frames per wave of the inflate pool (here ~10 000 frames of 16 KiB in 160 MB of Arrow bytes) decodes
    the FSE sequence streams LANE PER FRAME before the waves execute them (k_inflate, z_lane_frame): same bytes out as the
    one-wave path, for frames with many short sequences (small integers, runs), with long matches (constant stretches),
    with raw literals only (random bytes) and for the two-block frames of binary pages — device decode == input ==
    the oracle's decode of the device's pages."""
    from tests.test_gpu_encode import gpu_encode
    rng = np.random.default_rng(99)
    n = 10_000_000
    v = rng.integers(-2**40, 2**40, n)                      # 3-byte matches every 8 bytes
    v[2_000_000:3_000_000] = rng.integers(0, 50, 1_000_000)  # small values: long zero matches + 1 literal
    v[3_000_000:3_500_000] = 7                               # one long match per frame
    v[3_500_000:4_000_000] = rng.integers(-2**62, 2**62, 500_000)   # incompressible: raw blocks
    col = dict(ptype=S.T_I64, nullable=False, rows=n, values=v, validity=None, offsets=None)
    words = gen.binary(3_000_000, uniq=500, maxlen=6, seed=5)
    for c in (col, words):
        enc = gpu_encode(gpu_ctx, c, lz4_exact=False, max_page_size=65536, default_compression=S.ZSTD)
        pages, metas = enc.pages_numpy(), enc.metas_array()
        got = gpu_decode(gpu_ctx, c, pages, metas)
        want = gen.oracle_read(c, pages, metas)
        assert np.array_equal(got.values_numpy(), want["values"])
        assert np.array_equal(want["values"], np.ascontiguousarray(c["values"]).view(np.uint8))
        if c["offsets"] is not None:
            assert np.array_equal(got.offsets_numpy(), want["offsets"])
