"""LZ4 on the device (strawboat_amd/csrc/sb_lz4.h; reference src/compression/basic.rs:87-91,108-120).

Encoder: the default parallel parse is format-valid, not liblz4's bytes: the CPU oracle's decoder
— and pyarrow's liblz4 where importable — must read the device's pages back to the input, the page structure
(codec ids, hdr9 sizes) must equal the oracle's, and the blocks must not be much larger than liblz4's.
Decoder: liblz4-produced blocks (tests/golden/blocks/*.lz4, oracle-written pages whose LZ4 bytes equal liblz4's)
and hand-built blocks that exercise the format's corners decode to the same bytes as the oracle's decoder."""
import os

import numpy as np
import pytest

from oracle import sbo as S
from tests import gen

pytestmark = pytest.mark.gpu
DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "blocks")


def up(ctx, a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(ctx.torch_device)


def bytes_column(data):
    return dict(ptype=S.T_U8, nullable=False, rows=int(data.size), values=np.ascontiguousarray(data, np.uint8),
                validity=None, offsets=None)


def device_write(ctx, col, **kw):
    from strawboat_amd import write
    from strawboat_amd.types import WriteOptions
    dc = write.DeviceColumn(col["ptype"], col["nullable"], col["rows"], up(ctx, col["values"]),
                            None if col["validity"] is None else up(ctx, col["validity"]),
                            None if col["offsets"] is None else up(ctx, col["offsets"]))
    return write.write(ctx, dc, WriteOptions(default_compression=S.LZ4, **kw))


def device_read(ctx, col, pages, metas):
    from strawboat_amd import read
    return read.read_simple(ctx, read.ColumnPages(col["ptype"], col["nullable"], up(ctx, pages), metas))


def lz4_page(block, n_out):
    """hdr9 (codec 1) + block = a non-nullable UInt8 page of n_out rows"""
    hdr = bytes([S.LZ4]) + int(len(block)).to_bytes(4, "little") + int(n_out).to_bytes(4, "little")
    return np.frombuffer(hdr + bytes(block), np.uint8).copy(), np.array([[9 + len(block), n_out]], np.uint64)


def shapes():
    rng = np.random.default_rng(11)
    words = [b"w%d" % i + b"x" * (i % 9) for i in range(300)]
    yield "runs_i32", np.repeat(rng.integers(0, 50, 20_000), rng.integers(1, 9, 20_000)).astype(np.int32).view(np.uint8)
    yield "small_ints_u32", rng.integers(0, 1000, 70_000).astype(np.uint32).view(np.uint8)       # Dict-index like
    yield "random", rng.integers(0, 256, 200_000, dtype=np.uint8)                                # incompressible
    yield "zeros", np.zeros(300_000, np.uint8)                                                   # one long match
    yield "period3", np.tile(np.frombuffer(b"abc", np.uint8), 50_000)
    yield "period7_long", np.tile(rng.integers(0, 256, 7, dtype=np.uint8), 30_000)
    yield "period5000", np.tile(rng.integers(0, 256, 5000, dtype=np.uint8), 40)                  # long far matches
    yield "zipf_words", np.frombuffer(b"".join(words[i] for i in rng.zipf(1.2, 60_000) % 300), np.uint8)
    yield "sorted_i64", np.cumsum(rng.integers(0, 9, 40_000)).astype(np.int64).view(np.uint8)
    yield "f64_small_set", (rng.integers(0, 16, 50_000) * 0.25).astype(np.float64).view(np.uint8)
    yield "low_entropy", rng.integers(0, 4, 100_000, dtype=np.uint8)
    yield "mixed", np.concatenate([rng.integers(0, 256, 5000, dtype=np.uint8), np.zeros(7000, np.uint8),
                                   rng.integers(0, 256, 400, dtype=np.uint8), np.tile(np.arange(16, dtype=np.uint8), 900),
                                   rng.integers(0, 256, 70_000, dtype=np.uint8), np.full(66_000, 7, np.uint8)])
    for n in (0, 1, 4, 12, 13, 14, 17, 63, 64, 65, 300):
        yield "tiny_%d" % n, rng.integers(0, 3, n, dtype=np.uint8)


SHAPES = list(shapes())


@pytest.mark.parametrize("name", [s[0] for s in SHAPES])
@pytest.mark.parametrize("page", [None, 8192, 1000])
def test_fast_encoder_is_format_valid(gpu_ctx, name, page):
    data = dict(SHAPES)[name]
    if data.size == 0:
        pytest.skip("encode_chunk on an empty chunk panics upstream")
    col = bytes_column(data)
    enc = device_write(gpu_ctx, col, max_page_size=page)
    pages, metas = enc.pages_numpy(), enc.metas_array()
    want_pages, want_metas = gen.oracle_write(col, max_page_size=page, default_compression=S.LZ4)
    assert np.array_equal(metas[:, 1], want_metas[:, 1])
    # the oracle's LZ4 decoder (a restatement of LZ4_decompress_safe) reads the device's pages back
    got = gen.oracle_read(col, pages, metas)
    assert np.array_equal(got["values"], data), "oracle decode of the device's LZ4 pages differs from the input"
    # page structure: codec id and uncompressed size of every page as the oracle writes them
    off = 0
    total = 0
    for (length, rows), (wl, _) in zip(metas, want_metas):
        assert pages[off] == S.LZ4 and int.from_bytes(bytes(pages[off + 5:off + 9]), "little") == rows
        assert int.from_bytes(bytes(pages[off + 1:off + 5]), "little") == length - 9
        try:
            import pyarrow as pa
            out = pa.Codec("lz4_raw").decompress(bytes(pages[off + 9:off + int(length)]), decompressed_size=int(rows), asbytes=True)
            assert out == bytes(data[total:total + int(rows)]), "liblz4 decodes the device's block to something else"
        except ImportError:
            pass
        off += int(length)
        total += int(rows)
    # not much larger than liblz4's greedy parse (the oracle's bytes equal LZ4_compress_default's)
    if data.size >= 4096:
        assert pages.size <= 1.25 * want_pages.size + 64 * metas.shape[0], (pages.size, want_pages.size)
    # and the device reads its own pages
    back = device_read(gpu_ctx, col, pages, metas)
    assert np.array_equal(back.values_numpy(), data)


@pytest.mark.parametrize("name", [s[0] for s in SHAPES])
def test_decoder_reads_liblz4_parse(gpu_ctx, name):
    """oracle-written pages: their LZ4 blocks are byte-identical to LZ4_compress_default (tests/test_oracle_blocks.py)"""
    data = dict(SHAPES)[name]
    if data.size == 0:
        pytest.skip("no page")
    col = bytes_column(data)
    for page in (None, 5000):
        pages, metas = gen.oracle_write(col, max_page_size=page, default_compression=S.LZ4)
        got = device_read(gpu_ctx, col, pages, metas)
        assert np.array_equal(got.values_numpy(), data), page


def test_decoder_reads_golden_liblz4_blocks(gpu_ctx):
    import json
    for case in json.load(open(os.path.join(DIR, "index.json")))["cases"]:
        raw = np.fromfile(os.path.join(DIR, case["name"] + ".raw"), np.uint8)
        blk = np.fromfile(os.path.join(DIR, case["name"] + ".lz4"), np.uint8)
        if raw.size == 0:
            continue
        pages, metas = lz4_page(blk, raw.size)
        got = device_read(gpu_ctx, bytes_column(raw), pages, metas)
        assert np.array_equal(got.values_numpy(), raw), case["name"]


def _seq(lit, match_len=None, off=None):
    """one LZ4 sequence: literals + (optional) match"""
    out = bytearray()
    ml = 0 if match_len is None else match_len - 4
    tok = (min(len(lit), 15) << 4) | min(ml, 15)
    out.append(tok)
    if len(lit) >= 15:
        r = len(lit) - 15
        while r >= 255:
            out.append(255)
            r -= 255
        out.append(r)
    out += lit
    if match_len is not None:
        out += int(off).to_bytes(2, "little")
        if ml >= 15:
            r = ml - 15
            while r >= 255:
                out.append(255)
                r -= 255
            out.append(r)
    return bytes(out)


def test_decoder_format_corners(gpu_ctx):
    """hand-built blocks: length extensions of exactly 255 / 510, offsets 1 and 65535, overlapping matches of every
    small period, literals of 14 / 15 / 270 / 300 / 301 / 5000 bytes, matches crossing the 8 KiB staging window and
    reading from flushed output, a match of 100 000 bytes"""
    rng = np.random.default_rng(3)
    cases = []
    for L in (0, 1, 14, 15, 16, 269, 270, 271, 300, 301, 525, 5000):
        lit = bytes(rng.integers(0, 256, max(L, 1), dtype=np.uint8))
        cases.append(_seq(lit[:max(L, 1)] if L else b"\x07", 4 + 15 + 255 if L % 2 else 19, 1) + _seq(b"tail!"))
    for off in (1, 2, 3, 4, 5, 7, 8, 15, 16, 17, 63, 64, 65, 100):
        lit = bytes(rng.integers(0, 256, off, dtype=np.uint8))
        cases.append(_seq(lit, 4 + 200, off) + _seq(lit, 4, off) + _seq(b"12345"))
    big = bytes(rng.integers(0, 256, 66000, dtype=np.uint8))
    cases.append(_seq(big, 4 + 15 + 510, 65535) + _seq(b"x", 70, 65535) + _seq(b"abcde"))
    cases.append(_seq(b"ab", 100_000, 2) + _seq(b"zz", 9000, 50_000) + _seq(b"q" * 40, 8200, 60_001) + _seq(b"end.."))
    many = b"".join(_seq(bytes([65 + (i % 26)]) * (1 + i % 3), 4 + (i % 11), 1 + (i * 7) % (1 + i)) for i in range(3000))
    cases.append(_seq(b"0123456789abcdef", 4, 16) + many + _seq(b"final"))
    for blk in cases:
        # expected output from a straightforward python LZ4 decode
        out = bytearray()
        ip = 0
        while ip < len(blk):
            tok = blk[ip]; ip += 1
            ll = tok >> 4
            if ll == 15:
                while True:
                    b = blk[ip]; ip += 1; ll += b
                    if b != 255:
                        break
            out += blk[ip:ip + ll]; ip += ll
            if ip >= len(blk):
                break
            off = blk[ip] | (blk[ip + 1] << 8); ip += 2
            ml = tok & 15
            if ml == 15:
                while True:
                    b = blk[ip]; ip += 1; ml += b
                    if b != 255:
                        break
            ml += 4
            assert 0 < off <= len(out)
            for _ in range(ml):
                out.append(out[-off])
        want = np.frombuffer(bytes(out), np.uint8)
        assert np.array_equal(S.block_decompress(S.LZ4, np.frombuffer(blk, np.uint8), want.size), want)   # the oracle agrees
        pages, metas = lz4_page(blk, want.size)
        for shift in (0, 1, 5):   # the output position inside the column decides the 16-byte alignment of the flushes
            pre = np.arange(shift, dtype=np.uint8)
            if shift:
                p0 = np.frombuffer(bytes([S.NONE]) + shift.to_bytes(4, "little") * 2 + bytes(pre), np.uint8)
                pg = np.concatenate([p0, pages])
                mt = np.concatenate([np.array([[9 + shift, shift]], np.uint64), metas])
            else:
                pg, mt = pages, metas
            got = device_read(gpu_ctx, bytes_column(np.concatenate([pre, want])), pg, mt)
            assert np.array_equal(got.values_numpy(), np.concatenate([pre, want])), (len(blk), shift)


@pytest.mark.parametrize("bad", ["offset0", "offset_too_far", "truncated", "overrun", "no_last_literals"])
def test_decoder_rejects_malformed_blocks(gpu_ctx, bad):
    from strawboat_amd._native import NativeError
    if bad == "offset0":
        blk, n = _seq(b"abcd", 8, 0) + _seq(b"12345"), 17
    elif bad == "offset_too_far":
        blk, n = _seq(b"abcd", 8, 5) + _seq(b"12345"), 17
    elif bad == "truncated":
        blk, n = (_seq(b"abcd", 8, 2) + _seq(b"12345"))[:-3], 17
    elif bad == "overrun":
        blk, n = _seq(b"abcd", 80, 2) + _seq(b"12345"), 17
    else:
        blk, n = _seq(b"abcd", 13, 2), 17
    pages, metas = lz4_page(blk, n)
    with pytest.raises(NativeError) as e:
        device_read(gpu_ctx, bytes_column(np.zeros(n, np.uint8)), pages, metas)
    assert e.value.code == -2      # Error::External (LZ4_decompress_safe < 0 upstream)
    # the context keeps working
    ok = _seq(b"abcd", 8, 2) + _seq(b"12345")
    pages, metas = lz4_page(ok, 17)
    got = device_read(gpu_ctx, bytes_column(np.zeros(17, np.uint8)), pages, metas)
    assert bytes(got.values_numpy()) == b"abcd" + b"cdcdcdcd" + b"12345"


def test_binary_and_boolean_and_nested_blocks_fast_path(gpu_ctx):
    """every place an LZ4 block is written: binary offsets + values blocks, boolean bitmaps, Dict index blocks"""
    from strawboat_amd import write
    from strawboat_amd.types import WriteOptions
    for col, kw in ((gen.binary(30_000, uniq=300, null_density=0.1, zipf=1.3), dict(max_page_size=4096)),
                    (gen.binary(9_000, uniq=50, large=True), dict(max_page_size=9000)),
                    (gen.boolean(50_003, null_density=0.2, runs=7), dict(max_page_size=8192)),
                    (gen.prim(S.T_F64, 30_000, uniq=100, runs=5), dict(max_page_size=8192, force_codec=S.DICT, force_index_codec=S.LZ4)),
                    (gen.prim(S.T_I64, 20_000, uniq=1 << 40), dict(max_page_size=4096, default_compress_ratio=2.0))):
        dc = write.DeviceColumn(col["ptype"], col["nullable"], col["rows"], up(gpu_ctx, col["values"]),
                                None if col["validity"] is None else up(gpu_ctx, col["validity"]),
                                None if col["offsets"] is None else up(gpu_ctx, col["offsets"]))
        enc = write.write(gpu_ctx, dc, WriteOptions(default_compression=S.LZ4, **kw))
        pages, metas = enc.pages_numpy(), enc.metas_array()
        okw = dict(kw)
        if "default_compress_ratio" in okw:
            okw["ratio"] = okw.pop("default_compress_ratio")
        wp, wm = gen.oracle_write(col, default_compression=S.LZ4, **okw)
        want = gen.oracle_read(col, wp, wm)
        got = gen.oracle_read(col, pages, metas)          # CPU decode of the device's pages
        for k in ("values", "validity", "offsets"):
            assert np.array_equal(got[k], want[k]), k
        assert np.array_equal(S.stat_column(col["ptype"], col["nullable"], pages, metas)[0],
                              S.stat_column(col["ptype"], col["nullable"], wp, wm)[0])
        back = device_read(gpu_ctx, col, pages, metas)    # device decode of the device's pages
        assert np.array_equal(back.values_numpy(), want["values"])


def _liblz4_check(pages, off, length, data_bytes):
    """liblz4 (through pyarrow) must accept the block at pages[off + 9 : off + length] and give data_bytes"""
    try:
        import pyarrow as pa
    except ImportError:
        return
    out = pa.Codec("lz4_raw").decompress(bytes(pages[off + 9:off + int(length)]), decompressed_size=len(data_bytes), asbytes=True)
    assert out == data_bytes


@pytest.mark.parametrize("kind", ["zeros", "text", "random", "runs"])
@pytest.mark.parametrize("n", [65536 + 1, 65536 + 4, 65536 + 5, 65536 + 11, 65536 + 12, 65536 + 13, 65536 + 100,
                               2 * 65536, 2 * 65536 + 7, 5 * 65536 - 3, 1_000_003])
def test_chunked_blocks_end_rules(gpu_ctx, kind, n):
    """blocks of more than one 64 KiB chunk are compressed chunk by chunk and joined (k_enc_lz4_plan/_chunks/_stitch):
    the joined block must be one valid LZ4 block — liblz4's LZ4_decompress_safe enforces the end-of-block rules (last
    5 bytes literals, last match >= 12 bytes before the end) — whatever the size of the last chunk"""
    rng = np.random.default_rng(n % 1000)
    if kind == "zeros":
        data = np.zeros(n, np.uint8)
    elif kind == "text":
        words = [b"w%d" % i + b"y" * (i % 7) for i in range(200)]
        data = np.frombuffer(b"".join(words[i] for i in rng.zipf(1.3, n // 3) % 200)[:n].ljust(n, b"."), np.uint8)
    elif kind == "random":
        data = rng.integers(0, 256, n, dtype=np.uint8)
    else:
        data = np.repeat(rng.integers(0, 256, n // 40 + 1, dtype=np.uint8), 40)[:n]
    col = bytes_column(data)
    enc = device_write(gpu_ctx, col, max_page_size=None)
    pages, metas = enc.pages_numpy(), enc.metas_array()
    assert metas.shape[0] == 1 and int(metas[0, 1]) == n
    assert pages[0] == S.LZ4 and int.from_bytes(bytes(pages[1:5]), "little") == int(metas[0, 0]) - 9
    _liblz4_check(pages, 0, metas[0, 0], bytes(data))
    got = gen.oracle_read(col, pages, metas)
    assert np.array_equal(got["values"], data)
    back = device_read(gpu_ctx, col, pages, metas)
    assert np.array_equal(back.values_numpy(), data)
    if kind != "random":
        want_pages, _ = gen.oracle_write(col, max_page_size=None, default_compression=S.LZ4)
        assert pages.size <= 1.25 * want_pages.size + 64 * (n // 65536 + 1), (pages.size, want_pages.size)


def test_chunked_binary_boolean_nullable_pages(gpu_ctx):
    """every flat page shape with a block of more than one chunk: binary (offsets block and values block, i32 and i64
    offsets, nullable), boolean with a bit offset inside the column, wide primitives; adaptive mode with LZ4 as default"""
    from strawboat_amd import write
    from strawboat_amd.types import WriteOptions
    cases = ((gen.binary(120_000, uniq=5000, null_density=0.1, zipf=1.2, maxlen=40), dict(max_page_size=50_000)),
             (gen.binary(40_000, uniq=40_000, large=True, minlen=10, maxlen=30), dict(max_page_size=None)),
             (gen.boolean(2_000_003, null_density=0.2, runs=5), dict(max_page_size=700_001)),
             (gen.prim(S.T_I64, 200_000, uniq=1 << 40, null_density=0.3), dict(max_page_size=None)),
             (gen.prim(S.T_I32, 300_000, uniq=1 << 30), dict(max_page_size=100_000, default_compress_ratio=50.0)))
    for col, kw in cases:
        dc = write.DeviceColumn(col["ptype"], col["nullable"], col["rows"], up(gpu_ctx, col["values"]),
                                None if col["validity"] is None else up(gpu_ctx, col["validity"]),
                                None if col["offsets"] is None else up(gpu_ctx, col["offsets"]))
        enc = write.write(gpu_ctx, dc, WriteOptions(default_compression=S.LZ4, **kw))
        pages, metas = enc.pages_numpy(), enc.metas_array()
        okw = dict(kw)
        if "default_compress_ratio" in okw:
            okw["ratio"] = okw.pop("default_compress_ratio")
        wp, wm = gen.oracle_write(col, default_compression=S.LZ4, **okw)
        assert np.array_equal(metas[:, 1], wm[:, 1])
        want = gen.oracle_read(col, wp, wm)
        got = gen.oracle_read(col, pages, metas)
        for k in ("values", "validity", "offsets"):
            assert np.array_equal(got[k], want[k]), k
        assert np.array_equal(S.stat_column(col["ptype"], col["nullable"], pages, metas)[0],
                              S.stat_column(col["ptype"], col["nullable"], wp, wm)[0])
        back = device_read(gpu_ctx, col, pages, metas)
        assert np.array_equal(back.values_numpy(), want["values"])
        # (the matcher's history is its 8 KiB LDS ring, liblz4's is 64 KiB: a vocabulary of 5000 strings costs ~45 %)
        assert pages.size <= 1.6 * wp.size + 4096


# ---------------------------------------------------------------------------------------------------------------------
# blocks of 64 KiB of compressed bytes and more: the workgroup decoder (strawboat_amd/csrc/sb_lz4_big.h)
def _py_lz4(blk):
    out = bytearray()
    ip = 0
    while ip < len(blk):
        tok = blk[ip]; ip += 1
        ll = tok >> 4
        if ll == 15:
            while True:
                b = blk[ip]; ip += 1; ll += b
                if b != 255:
                    break
        out += blk[ip:ip + ll]; ip += ll
        if ip >= len(blk):
            break
        off = blk[ip] | (blk[ip + 1] << 8); ip += 2
        ml = tok & 15
        if ml == 15:
            while True:
                b = blk[ip]; ip += 1; ml += b
                if b != 255:
                    break
        ml += 4
        assert 0 < off <= len(out)
        if off >= ml:
            out += out[len(out) - off:len(out) - off + ml]
        else:
            seg = bytes(out[len(out) - off:])
            out += (seg * (ml // off + 1))[:ml]
    return np.frombuffer(bytes(out), np.uint8)


def _big_blocks():
    rng = np.random.default_rng(23)
    words = [b"w%d" % i + b"x" * (i % 9) for i in range(3000)]
    text = np.frombuffer(b" ".join(words[i] for i in rng.zipf(1.1, 260_000) % 3000), np.uint8)
    yield "text", bytes(S.block_compress(S.LZ4, text))
    yield "random", bytes(S.block_compress(S.LZ4, rng.integers(0, 256, 300_000, dtype=np.uint8)))
    yield "small_ints", bytes(S.block_compress(S.LZ4, rng.integers(0, 1000, 200_000).astype(np.uint32).view(np.uint8)))
    yield "low_entropy", bytes(S.block_compress(S.LZ4, rng.integers(0, 4, 1_200_000, dtype=np.uint8)))
    yield "sorted_i64", bytes(S.block_compress(S.LZ4, np.cumsum(rng.integers(0, 9, 150_000)).astype(np.int64).view(np.uint8)))
    # more than 1024 sequences in 4 KiB of input: 3-byte sequences (no literals, 4..18-byte matches)
    dense = b"".join(_seq(b"", 4 + (i % 15), 1 + (i * 13) % 64) for i in range(40_000))
    yield "dense", _seq(bytes(range(64)), 4, 64) + dense + _seq(b"final")
    # literal runs that do not fit the staged input, between dense stretches; matches longer than a window; far offsets
    lit = lambda n: bytes(rng.integers(0, 256, n, dtype=np.uint8))
    parts = [_seq(lit(70_000), 300_000, 1), _seq(lit(281), 4, 65_000), _seq(lit(5000), 40_000, 65_535)]
    for i in range(6000):
        parts.append(_seq(lit(i % 5), 4 + (i % 40), 1 + (i * 31) % 60_000))
        if i % 500 == 0:
            parts.append(_seq(lit(300 + i), 20_000 + i, 3 + i % 7))
    parts.append(_seq(lit(66_000), 5, 2))
    parts.append(_seq(b"", 100_000, 65_535))
    parts.append(_seq(b"abcde"))
    yield "long_and_far", b"".join(parts)
    # literal length extensions that end exactly at the staged border, sequences straddling chunk borders
    parts = [_seq(lit(20), 8, 3)]
    for i in range(2000):
        parts.append(_seq(lit(255 + 15 + (i % 40)), 4 + 15 + (255 if i % 3 == 0 else i % 7), 1 + (i * 17) % 20))
    parts.append(_seq(b"tail!"))
    yield "ext_borders", b"".join(parts)


BIG = None


def _big():
    global BIG
    if BIG is None:
        BIG = dict(_big_blocks())
    return BIG


@pytest.mark.parametrize("name", ["text", "random", "small_ints", "low_entropy", "sorted_i64", "dense", "long_and_far", "ext_borders"])
def test_big_block_decoder(gpu_ctx, name):
    blk = _big()[name]
    assert len(blk) >= 65536, len(blk)
    want = _py_lz4(blk)
    assert np.array_equal(S.block_decompress(S.LZ4, np.frombuffer(blk, np.uint8), want.size), want)
    pages, metas = lz4_page(blk, want.size)
    for shift in (0, 3):
        pre = np.arange(shift, dtype=np.uint8)
        if shift:
            p0 = np.frombuffer(bytes([S.NONE]) + shift.to_bytes(4, "little") * 2 + bytes(pre), np.uint8)
            pg = np.concatenate([p0, pages])
            mt = np.concatenate([np.array([[9 + shift, shift]], np.uint64), metas])
        else:
            pg, mt = pages, metas
        got = device_read(gpu_ctx, bytes_column(np.concatenate([pre, want])), pg, mt).values_numpy()
        exp = np.concatenate([pre, want])
        bad = np.flatnonzero(got != exp)
        assert bad.size == 0, (name, shift, int(bad[0]), int(bad.size))


def test_big_blocks_many_per_call(gpu_ctx):
    """several big blocks and small ones in one call (both LZ4 kernels walk the same queue)"""
    from strawboat_amd import read
    B = _big()
    names = ["text", "dense", "small_ints", "ext_borders"]
    cols, want = [], []
    for k in range(10):
        blk = B[names[k % 4]] if k % 3 else _seq(b"abcd", 8, 2) + _seq(b"12345")
        w = _py_lz4(blk)
        pages, metas = lz4_page(blk, w.size)
        cols.append(read.ColumnPages(S.T_U8, False, up(gpu_ctx, pages), metas))
        want.append(w)
    got = read.batch_read_columns(gpu_ctx, cols)
    gpu_ctx.synchronize()
    for g, w in zip(got, want):
        assert np.array_equal(g.values_numpy(), w)


@pytest.mark.parametrize("bad", ["offset0", "offset_too_far", "truncated", "short_output", "long_output", "no_last_literals",
                                 "garbage_tail", "literal_overrun"])
def test_big_block_decoder_rejects_malformed(gpu_ctx, bad):
    from strawboat_amd._native import NativeError
    rng = np.random.default_rng(5)
    lit = lambda n: bytes(rng.integers(0, 256, n, dtype=np.uint8))
    body = [_seq(lit(1 + i % 9), 4 + i % 30, 1 + (i * 7) % (1 + i)) for i in range(20_000)]
    good = b"".join(body) + _seq(b"final")
    n = _py_lz4(good).size
    if bad == "offset0":
        blk = b"".join(body[:9000]) + _seq(b"ab", 9, 0) + b"".join(body[9000:]) + _seq(b"final")
        n += 11
    elif bad == "offset_too_far":
        blk = _seq(b"abcd", 8, 5) + good
        n += 12
    elif bad == "truncated":
        blk = good[:-3]
    elif bad == "short_output":
        blk, n = good, n - 1
    elif bad == "long_output":
        blk, n = good, n + 1
    elif bad == "no_last_literals":
        blk = b"".join(body)
        n -= 5
    elif bad == "garbage_tail":
        blk = good + b"\x00"
    else:
        blk = b"".join(body) + bytes([0xF0, 255, 255, 255, 10]) + b"xy"
        n = n - 5 + 2
    assert len(blk) >= 65536
    pages, metas = lz4_page(blk, n)
    with pytest.raises(NativeError) as e:
        device_read(gpu_ctx, bytes_column(np.zeros(n, np.uint8)), pages, metas)
    assert e.value.code == -2
    pages, metas = lz4_page(good, _py_lz4(good).size)
    got = device_read(gpu_ctx, bytes_column(np.zeros(1, np.uint8)), pages, metas)
    assert np.array_equal(got.values_numpy(), _py_lz4(good))


# ---------------------------------------------------------------------------------------------------------------------
# blocks of 2 MiB of compressed bytes and more: the block-parallel decoder (strawboat_amd/csrc/sb_lz4_giant.h).  The
# reference's default paging makes a column ONE page and a Basic(LZ4) page is ONE block (src/compression/basic.rs:87-91).
def _lz4_out_len(blk):
    """output bytes of a well-formed block (headers only)"""
    ip, out, n = 0, 0, len(blk)
    while ip < n:
        tok = blk[ip]; ip += 1
        ll = tok >> 4
        if ll == 15:
            while True:
                b = blk[ip]; ip += 1; ll += b
                if b != 255:
                    break
        ip += ll; out += ll
        if ip >= n:
            break
        ip += 2
        ml = tok & 15
        if ml == 15:
            while True:
                b = blk[ip]; ip += 1; ml += b
                if b != 255:
                    break
        out += ml + 4
    return out


def _giant_blocks():
    rng = np.random.default_rng(29)
    lit = lambda n: bytes(rng.integers(0, 256, n, dtype=np.uint8))
    # what the reference's writer produces for one-page columns (liblz4's parse through the oracle)
    yield "sorted_i64", bytes(S.block_compress(S.LZ4, np.cumsum(rng.integers(0, 1 << 20, 1_500_000)).astype(np.int64).view(np.uint8)))
    words = [b"w%d" % i + b"x" * (i % 9) for i in range(3000)]
    text = np.frombuffer(b" ".join(words[i] for i in rng.zipf(1.1, 2_200_000) % 3000), np.uint8)
    yield "text", bytes(S.block_compress(S.LZ4, text))
    yield "random", bytes(S.block_compress(S.LZ4, rng.integers(0, 256, 2_500_000, dtype=np.uint8)))   # ONE literal run: 9 800 length bytes of 255
    yield "small_ints", bytes(S.block_compress(S.LZ4, rng.integers(0, 1000, 1_500_000).astype(np.uint32).view(np.uint8)))
    v = np.zeros(40_000_000, np.uint8)
    v[::1_000_003] = 7
    yield "zeros", bytes(S.block_compress(S.LZ4, v)) + b""     # tiny: stays with the other decoders (a control)
    # hand-built: three-byte sequences (more than 1 300 sequence starts per chunk of 4 KiB), chains of matches through
    # EVERY window (offset 1 / 8 / 65 535), literal runs longer than the chunk tables cover, sequences across chunk borders
    dense = [_seq(bytes(range(64)), 4, 64)]
    for i in range(900_000):
        dense.append(_seq(b"", 4 + (i % 15), 1 + (i * 13) % 64))
    dense.append(_seq(b"final"))
    yield "dense", b"".join(dense)
    parts = [_seq(lit(70_000), 300_000, 1), _seq(lit(281), 4, 65_000), _seq(lit(5000), 40_000, 65_535)]
    for i in range(260_000):
        parts.append(_seq(lit(i % 11), 4 + (i % 40), 1 + (i * 31) % 60_000))
        if i % 5000 == 0:
            parts.append(_seq(lit(300 + i % 9000), 20_000 + i, 3 + i % 7))
        if i % 40_000 == 7:
            parts.append(_seq(lit(17_000 + i % 3000), 70_000, 8))          # headers of more than 64 length bytes
    parts.append(_seq(lit(66_000), 5, 2))
    parts.append(_seq(b"", 1_000_000, 65_535))
    parts.append(_seq(b"abcde"))
    yield "long_and_far", b"".join(parts)
    parts = [_seq(lit(20), 8, 3)]
    for i in range(9000):
        parts.append(_seq(lit(255 + 15 + (i % 40)), 4 + 15 + (255 if i % 3 == 0 else i % 7), 1 + (i * 17) % 20))
    parts.append(_seq(b"tail!"))
    yield "ext_borders", b"".join(parts)


GIANT = None


def _giant():
    global GIANT
    if GIANT is None:
        GIANT = dict(_giant_blocks())
    return GIANT


@pytest.mark.parametrize("name", ["sorted_i64", "text", "random", "small_ints", "zeros", "dense", "long_and_far", "ext_borders"])
def test_giant_block_decoder(gpu_ctx, name):
    blk = _giant()[name]
    n_out = _lz4_out_len(blk)
    want = S.block_decompress(S.LZ4, np.frombuffer(blk, np.uint8), n_out)
    assert name == "zeros" or len(blk) >= 2 << 20, len(blk)
    pages, metas = lz4_page(blk, n_out)
    for shift in (0, 5):
        pre = np.arange(shift, dtype=np.uint8)
        if shift:
            p0 = np.frombuffer(bytes([S.NONE]) + shift.to_bytes(4, "little") * 2 + bytes(pre), np.uint8)
            pg = np.concatenate([p0, pages])
            mt = np.concatenate([np.array([[9 + shift, shift]], np.uint64), metas])
        else:
            pg, mt = pages, metas
        got = device_read(gpu_ctx, bytes_column(np.concatenate([pre, want])), pg, mt).values_numpy()
        exp = np.concatenate([pre, want])
        bad = np.flatnonzero(got != exp)
        assert bad.size == 0, (name, shift, int(bad[0]), int(bad.size))


def test_giant_blocks_next_to_others(gpu_ctx):
    """three giant blocks, big ones and small ones in one call: every decoder takes its own entries of the queue"""
    from strawboat_amd import read
    G, B = _giant(), _big()
    blks = [G["text"], B["dense"], G["dense"], _seq(b"abcd", 8, 2) + _seq(b"12345"), G["small_ints"], B["text"]]
    cols, want = [], []
    for blk in blks:
        n_out = _lz4_out_len(blk)
        pages, metas = lz4_page(blk, n_out)
        cols.append(read.ColumnPages(S.T_U8, False, up(gpu_ctx, pages), metas))
        want.append(S.block_decompress(S.LZ4, np.frombuffer(blk, np.uint8), n_out))
    for _ in range(2):   # (the second call reuses the pool)
        got = read.batch_read_columns(gpu_ctx, cols)
        gpu_ctx.synchronize()
        for g, w in zip(got, want):
            assert np.array_equal(g.values_numpy(), w)


@pytest.mark.parametrize("bad", ["offset0", "offset_too_far", "truncated", "short_output", "long_output", "no_last_literals",
                                 "garbage_tail", "literal_overrun"])
def test_giant_block_decoder_rejects_malformed(gpu_ctx, bad):
    from strawboat_amd._native import NativeError
    rng = np.random.default_rng(6)
    lit = lambda n: bytes(rng.integers(0, 256, n, dtype=np.uint8))
    body = [_seq(lit(1 + i % 9), 4 + i % 30, 1 + (i * 7) % min(1 + i, 60_000)) for i in range(320_000)]
    good = b"".join(body) + _seq(b"final")
    assert len(good) >= 2 << 20
    n = _lz4_out_len(good)
    if bad == "offset0":
        blk = b"".join(body[:190_000]) + _seq(b"ab", 9, 0) + b"".join(body[190_000:]) + _seq(b"final")
        n += 11
    elif bad == "offset_too_far":
        blk = _seq(b"abcd", 8, 5) + good
        n += 12
    elif bad == "truncated":
        blk = good[:-3]
    elif bad == "short_output":
        blk, n = good, n - 1
    elif bad == "long_output":
        blk, n = good, n + 1
    elif bad == "no_last_literals":
        blk = b"".join(body)
        n -= 5
    elif bad == "garbage_tail":
        blk = good + b"\x00\x00\x00"
    else:
        blk = b"".join(body) + bytes([0xF0, 255, 255, 3]) + b"xy"
    pages, metas = lz4_page(blk, n)
    with pytest.raises(NativeError):
        device_read(gpu_ctx, bytes_column(np.zeros(n, np.uint8)), pages, metas)
    # the context still works
    test_giant_block_decoder(gpu_ctx, "ext_borders")


def test_giant_block_after_plain_long_pages_is_replayed(gpu_ctx):
    """a context whose last intervals read long pages without an LZ4 block of megabytes stops launching the block-parallel
    chain; when such a block shows up after all, k_inflate_lz4_big leaves it alone (one workgroup took 0.8 s for 68 MB),
    the interval is issued again with the chain (sb_ctx_replays) and the bytes are right"""
    import os
    if os.environ.get("SB_NO_HINTS", "0") != "0":
        pytest.skip("SB_NO_HINTS: every kernel is launched, nothing to replay")
    import time
    from strawboat_amd import read
    blk = _giant()["sorted_i64"]
    n_out = _lz4_out_len(blk)
    want = S.block_decompress(S.LZ4, np.frombuffer(blk, np.uint8), n_out)
    pages, metas = lz4_page(blk, n_out)
    plain = np.random.default_rng(1).integers(0, 255, 3 << 20).astype(np.uint8)
    ppages = np.frombuffer(bytes([S.NONE]) + len(plain).to_bytes(4, "little") * 2 + plain.tobytes(), np.uint8)
    pmetas = np.array([[9 + len(plain), len(plain)]], np.uint64)
    pcol = read.ColumnPages(S.T_U8, False, up(gpu_ctx, ppages), pmetas)
    gcol = read.ColumnPages(S.T_U8, False, up(gpu_ctx, pages), metas)
    for _ in range(5):      # long pages, no giant block: the chain is dropped
        got = read.batch_read_columns(gpu_ctx, [pcol])
        gpu_ctx.synchronize()
        assert np.array_equal(got[0].values_numpy(), plain)
    r0 = gpu_ctx.replays()
    t0 = time.perf_counter()
    got = read.batch_read_columns(gpu_ctx, [gcol])
    gpu_ctx.synchronize()
    dt = time.perf_counter() - t0
    assert np.array_equal(got[0].values_numpy(), want)
    assert gpu_ctx.replays() == r0 + 1
    assert dt < 0.1, dt
    got = read.batch_read_columns(gpu_ctx, [gcol])     # the chain is on again: no replay
    gpu_ctx.synchronize()
    assert np.array_equal(got[0].values_numpy(), want) and gpu_ctx.replays() == r0 + 1
