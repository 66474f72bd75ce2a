"""GPU parity: pages written by the CPU oracle, decoded by the HIP path through the C ABI,
compared bit for bit with the oracle's decode of the same pages."""
import numpy as np
import pytest

from oracle import sbo as S
from tests import gen

pytestmark = pytest.mark.gpu


def gpu_decode(ctx, col, pages, metas):
    import torch
    from strawboat_amd import read
    cp = read.ColumnPages(col["ptype"], col["nullable"], torch.from_numpy(pages).to(ctx.torch_device), metas)
    return read.read_simple(ctx, cp)


def check(ctx, col, **opt):
    pages, metas = gen.oracle_write(col, **opt)
    want = gen.oracle_read(col, pages, metas)
    got = gpu_decode(ctx, col, pages, metas)
    assert got.rows == want["rows"]
    assert np.array_equal(got.values_numpy(), want["values"]), "values differ"
    if col["nullable"] and col["ptype"] != S.T_NULL:
        assert np.array_equal(got.validity_numpy(), want["validity"]), "validity differs"
    if col["offsets"] is not None:
        assert np.array_equal(got.offsets_numpy(), want["offsets"]), "offsets differ"
    return pages, metas


PRIMS = [S.T_I8, S.T_I16, S.T_I32, S.T_I64, S.T_U8, S.T_U16, S.T_U32, S.T_U64, S.T_F32, S.T_F64, S.T_I128, S.T_I256]


@pytest.mark.parametrize("ptype", PRIMS)
@pytest.mark.parametrize("codec", [S.NONE, S.RLE, S.DICT, S.ONEVALUE])
def test_prim_codecs(gpu_ctx, ptype, codec):
    uniq = 1 if codec == S.ONEVALUE else 100
    col = gen.prim(ptype, 10_000, uniq=uniq, null_density=0.2, runs=8)
    check(gpu_ctx, col, max_page_size=2048, force_codec=codec)
    col = gen.prim(ptype, 10_000, uniq=uniq, runs=3)
    check(gpu_ctx, col, max_page_size=4100, force_codec=codec)


@pytest.mark.parametrize("ptype", [S.T_I32, S.T_U32])
@pytest.mark.parametrize("codec", [S.BITPACK, S.DELTABP])
def test_bitpacking(gpu_ctx, ptype, codec):
    col = gen.prim(ptype, 128 * 100, uniq=1 << 13, sorted_=(codec == S.DELTABP))
    check(gpu_ctx, col, max_page_size=128 * 40, force_codec=codec)
    col = gen.prim(ptype, 128 * 1024, uniq=1 << 30, sorted_=(codec == S.DELTABP), seed=7)
    check(gpu_ctx, col, max_page_size=65536, force_codec=codec)


@pytest.mark.parametrize("icodec", [S.NONE, S.RLE, S.BITPACK, S.DELTABP, S.ONEVALUE, S.LZ4])
def test_dict_index_codecs(gpu_ctx, icodec):
    uniq = 1 if icodec == S.ONEVALUE else 200
    col = gen.prim(S.T_F64, 128 * 300, uniq=uniq, null_density=0.1, runs=16, sorted_=(icodec == S.DELTABP))
    check(gpu_ctx, col, max_page_size=128 * 100, force_codec=S.DICT, force_index_codec=icodec)


def test_c1_int64_single_page(gpu_ctx):
    rng = np.random.default_rng(42)
    vals = rng.integers(0, 1 << 62, 1_000_000).astype(np.int64)
    col = dict(ptype=S.T_I64, nullable=False, rows=vals.size, values=vals, validity=None, offsets=None)
    pages, metas = check(gpu_ctx, col)
    assert metas.shape[0] == 1 and metas[0, 0] == 8_000_009


def test_c2_float64_pages(gpu_ctx):
    col = gen.prim(S.T_F64, 1_000_000, uniq=256, null_density=0.1, runs=32)
    for codec in (S.RLE, S.DICT, S.NONE):
        check(gpu_ctx, col, max_page_size=65536, force_codec=codec)
    # the last page (16 960 rows) is not a multiple of 128: Bitpacking is ineligible there
    # (integer/bp.rs:92-99), so the forced Dict->Bitpacking variant covers the 15 full pages
    full = {k: (v[:15 * 65536] if k == "values" else v) for k, v in col.items()}
    full["rows"] = 15 * 65536
    full["validity"] = col["validity"][:15 * 65536 // 8]
    check(gpu_ctx, full, max_page_size=65536, force_codec=S.DICT, force_index_codec=S.BITPACK)


@pytest.mark.parametrize("codec", [S.NONE, S.RLE, S.ONEVALUE, S.LZ4])
def test_boolean(gpu_ctx, codec):
    p = 1.0 if codec == S.ONEVALUE else 0.5
    col = gen.boolean(100_003, null_density=0.3, p_true=p, runs=5)
    check(gpu_ctx, col, max_page_size=8192, force_codec=codec)
    col = gen.boolean(10_000, p_true=p, runs=40)
    check(gpu_ctx, col, max_page_size=1000, force_codec=codec)  # pages not multiples of 32 rows


@pytest.mark.parametrize("large", [False, True])
@pytest.mark.parametrize("codec", [S.NONE, S.DICT, S.ONEVALUE, S.LZ4])
def test_binary(gpu_ctx, codec, large):
    uniq = 1 if codec == S.ONEVALUE else 300
    col = gen.binary(20_000, uniq=uniq, null_density=0.1, large=large, zipf=1.3)
    check(gpu_ctx, col, max_page_size=4096, force_codec=codec)
    col = gen.binary(5_000, uniq=uniq, large=large)
    check(gpu_ctx, col, max_page_size=5000, force_codec=codec, force_index_codec=S.RLE)


@pytest.mark.parametrize("ptype", [S.T_I32, S.T_I64, S.T_F64])
def test_lz4_pages(gpu_ctx, ptype):
    col = gen.prim(ptype, 50_000, uniq=50, null_density=0.1, runs=4)
    check(gpu_ctx, col, max_page_size=8192, default_compression=S.LZ4)


def test_corrupt_codec_raises(gpu_ctx):
    from strawboat_amd._native import NativeError
    col = gen.prim(S.T_I32, 1000, uniq=10)
    pages, metas = gen.oracle_write(col)
    pages = pages.copy()
    pages[0] = 77  # unknown codec id
    with pytest.raises(NativeError) as e:
        gpu_decode(gpu_ctx, col, pages, metas)
    assert e.value.code == -1


# ---- hand-built RLE pages: what the page-level RLE kernel must get right beyond oracle-written data
def _rle_page(runs, w, dtype):
    body = b"".join(int(c).to_bytes(4, "little") + np.array([v], dtype).tobytes() for c, v in runs)
    return np.frombuffer(bytes([S.RLE]) + len(body).to_bytes(4, "little") + (0).to_bytes(4, "little") + body, np.uint8)


def _decode_rle_pages(ctx, ptype, page_list, rows_list):
    import torch
    from strawboat_amd import read
    pages = np.concatenate(page_list)
    metas = np.array([[p.size, n] for p, n in zip(page_list, rows_list)], np.uint64)
    cp = read.ColumnPages(ptype, False, torch.from_numpy(pages).to(ctx.torch_device), metas)
    got = read.read_simple(ctx, cp)
    want = S.read_column(ptype, False, pages, metas)
    assert got.rows == want["rows"] == sum(rows_list)
    assert np.array_equal(got.values_numpy(), want["values"])
    return got


@pytest.mark.parametrize("ptype,dtype,w", [(S.T_U8, np.uint8, 1), (S.T_I16, np.int16, 2), (S.T_I32, np.int32, 4),
                                           (S.T_F64, np.float64, 8)])
def test_rle_hand_built_pages(gpu_ctx, ptype, dtype, w):
    rng = np.random.default_rng(5)
    # zero-count runs, a run that overshoots the page, > 1024 runs (several chunks), rows % 4096 != 0
    runs = [(0, 99), (3, 1), (0, 2), (0, 3), (5000, 4)] + [(int(c), int(v)) for c, v in zip(rng.integers(0, 9, 3000), rng.integers(0, 100, 3000))]
    n1 = sum(c for c, _ in runs)
    one = [(12345, 42)]                       # a page that is one run
    # three pages with odd row counts: the second and third start at odd output rows; trailing runs
    # after the page is full are never read (the decoder stops at N rows)
    _decode_rle_pages(gpu_ctx, ptype, [_rle_page(runs, w, dtype), _rle_page(one + [(7, 1)], w, dtype), _rle_page(runs[:50], w, dtype)],
                      [n1, 12345, sum(c for c, _ in runs[:50])])


def test_rle_run_overshooting_the_page_is_out_of_spec(gpu_ctx):
    from strawboat_amd._native import NativeError
    with pytest.raises(NativeError) as e:  # upstream: assert on the decoded length (read/array/integer.rs:81)
        _decode_rle_pages(gpu_ctx, S.T_I64, [_rle_page([(10, 1), (10, 2)], 8, np.int64)], [16])
    assert e.value.code == -1


def test_rle_runs_end_before_the_page_is_full(gpu_ctx):
    from strawboat_amd._native import NativeError
    with pytest.raises(NativeError) as e:  # read_u32 hits EOF upstream (integer/rle.rs:128-131)
        _decode_rle_pages(gpu_ctx, S.T_I32, [_rle_page([(10, 1), (0, 2), (5, 3)], 4, np.int32)], [16])
    assert e.value.code == -3


def test_binary_dict_entries_with_zero_bytes_and_empty_strings(gpu_ctx):
    """k_plan finds the entries of a binary Dict page in parallel by looking for length fields (four zero bytes in front of
    text) and verifies the chain it found; dictionaries whose strings hold zero bytes, are empty or are raw little-endian
    integers defeat the pattern and must come out the same through the serial walk"""
    from tests.test_gpu_encode import gpu_encode
    rng = np.random.default_rng(17)
    n = 150_000
    vocab = [b"", b"\0", b"\0\0\0\0\0\0\0\0", b"a\0b", b"tail\0\0\0\0", b"\0\0\0\0head"] + \
            [int(v).to_bytes(8, "little") for v in rng.integers(0, 1 << 20, 300)] + [("w%d" % k).encode() for k in range(200)]
    idx = rng.integers(0, len(vocab), n)
    lens = np.array([len(vocab[i]) for i in idx], np.int64)
    offs = np.zeros(n + 1, np.int64)
    np.cumsum(lens, out=offs[1:])
    data = np.frombuffer(b"".join(vocab[i] for i in idx), np.uint8).copy()
    for nulls in (None, 0.2):
        col = dict(ptype=S.T_BIN32, nullable=nulls is not None, rows=n, values=data,
                   validity=None if nulls is None else gen.pack_bits(rng.random(n) >= nulls), offsets=offs.astype(np.int32))
        pages, metas = gen.oracle_write(col, max_page_size=65536, force_codec=S.DICT)
        want = gen.oracle_read(col, pages, metas)
        got = gpu_decode(gpu_ctx, col, pages, metas)
        assert np.array_equal(got.values_numpy(), want["values"])
        assert np.array_equal(got.offsets_numpy(), want["offsets"])
        enc = gpu_encode(gpu_ctx, col, max_page_size=65536, default_compression=S.LZ4, ratio=2.0)
        wp, wm = gen.oracle_write(col, max_page_size=65536, default_compression=S.LZ4, ratio=2.0)
        assert np.array_equal(enc.metas_array(), wm) and np.array_equal(enc.pages_numpy(), wp)


@pytest.mark.parametrize("large", [False, True])
def test_binary_dict_tile_totals_by_tile(gpu_ctx, large):
    """Binary Dict pages of four tiles (4096 rows each) or more leave the value bytes of their tiles to k_bin_tile_sums /
    k_bin_tile_scan — a workgroup per tile instead of a walk inside the page's k_plan workgroup — shorter pages keep the
    walk: pages around the limit (3 tiles, exactly 4, 4 and a row, a last tile of one row), pages of both kinds in one
    call, every index codec the reference nests in a Dict block (binary/dict.rs:60-62), nulls, and a 700 000-row page."""
    for rows, mps in ((12_288, None), (12_289, None), (16_384, None), (16_385, None), (20_000, None), (100_000, 16_385),
                      (90_000, 20_000), (163_840, 16_384), (700_001, None)):
        for icodec in (None, S.RLE, S.BITPACK, S.LZ4):
            if rows > 100_000 and icodec in (S.RLE, S.LZ4):
                continue
            col = gen.binary(rows, uniq=700, null_density=0.1 if rows % 2 else None, large=large, zipf=1.2, maxlen=20, seed=rows)
            opt = dict(max_page_size=mps, force_codec=S.DICT)
            if icodec is not None:
                opt["force_index_codec"] = icodec
            if icodec == S.BITPACK and (rows % 128 or (mps or 128) % 128):   # (the crate asserts whole 128-blocks)
                continue
            check(gpu_ctx, col, **opt)
    # adaptive, LZ4 default: Dict pages with bit-packed indices next to a short last page, two columns in one call
    import torch
    from strawboat_amd import read
    cols = [gen.binary(150_000, uniq=900, zipf=1.1, maxlen=24, large=large, seed=5),
            gen.binary(70_000, uniq=50, null_density=0.2, large=large, seed=6)]
    enc = [gen.oracle_write(c, max_page_size=65536, default_compression=S.LZ4, ratio=2.0) for c in cols]
    cps = [read.ColumnPages(c["ptype"], c["nullable"], torch.from_numpy(p).to(gpu_ctx.torch_device), m) for c, (p, m) in zip(cols, enc)]
    got = read.batch_read_columns(gpu_ctx, cps)
    gpu_ctx.synchronize()
    for c, (p, m), g in zip(cols, enc, got):
        want = gen.oracle_read(c, p, m)
        assert np.array_equal(g.values_numpy(), want["values"])
        assert np.array_equal(g.offsets_numpy(), want["offsets"])
        if c["nullable"]:
            assert np.array_equal(g.validity_numpy(), want["validity"])


def test_binary_freq_page_tile_totals_by_tile(gpu_ctx):
    """a binary Freq page is read as a virtual Dict page (entry 0 = the top value, then the exception records behind the
    bitmap): 40 000 rows = 10 tiles, so its tile totals come from the tile kernels too (the gap between entry 0 and 1)"""
    rng = np.random.default_rng(3)
    n = 40_000
    words = [b"top-value"] + [("exception-%d" % k).encode() for k in range(400)]
    idx = np.where(rng.random(n) < 0.04, rng.integers(1, len(words), n), 0)
    lens = np.array([len(w) for w in words], np.int64)[idx]
    offs = np.zeros(n + 1, np.int64)
    np.cumsum(lens, out=offs[1:])
    data = np.frombuffer(b"".join(words[i] for i in idx), np.uint8).copy()
    for nulls in (None, 0.1):
        col = dict(ptype=S.T_BIN32, nullable=nulls is not None, rows=n, values=data,
                   validity=None if nulls is None else gen.pack_bits(rng.random(n) >= nulls), offsets=offs.astype(np.int32))
        pages, metas = check(gpu_ctx, col, force_codec=S.FREQ)
        assert S.stat_column(col["ptype"], col["nullable"], pages, metas)[0].tolist() == [S.FREQ]


def test_binary_dict_long_page_with_an_index_out_of_range(gpu_ctx):
    """an index beyond the dictionary in a tile of a long page: OutOfSpec from the tile kernel, like the walk (and the oracle)"""
    from strawboat_amd._native import NativeError
    seen = set()
    for uniq, value in ((200, 255), (200, 3), (2000, 1999), (2000, 40_000)):
        c = gen.binary(30_000, uniq=uniq, maxlen=10, seed=9)
        pages, metas = gen.oracle_write(c, force_codec=S.DICT, force_index_codec=S.NONE)
        bad = pages.copy()
        # plain u32 indices: hdr9 Dict | hdr9 None | N * 4 bytes; row 25 000 (tile 6) gets another index
        pos = 9 + 9 + 25_000 * 4
        bad[pos:pos + 4] = np.frombuffer(np.uint32(value).tobytes(), np.uint8)
        try:
            want = gen.oracle_read(c, bad, metas)
        except Exception:
            want = None
        seen.add(want is None)
        try:
            got = gpu_decode(gpu_ctx, c, bad, metas)
            gpu_ctx.synchronize()
        except NativeError as e:
            assert want is None and e.code == -1
            continue
        assert want is not None, "the oracle refuses index %d, the device decoded the page" % value
        assert np.array_equal(got.values_numpy(), want["values"]) and np.array_equal(got.offsets_numpy(), want["offsets"])
    assert seen == {True, False}


def test_read_kernels_left_out_on_a_hint_are_replayed():
    """a read call leaves out the inflate kernels of queue A and the tile kernel of primitives when the context's last read
    interval queued nothing for them (RLE pages: a page kernel expands them); pages that need them after all — LZ4 blocks,
    plain tiles, Dict pages with an LZ4 index block — show up: the interval is issued again with everything (sb_ctx_replays)
    and decodes to the oracle's values"""
    import os
    if os.environ.get("SB_NO_HINTS", "0") != "0":
        pytest.skip("SB_NO_HINTS: every kernel is launched, nothing to replay")
    import strawboat_amd as sb
    ctx = sb.Context(0)
    try:
        rle = gen.prim(S.T_F64, 200_000, uniq=50, null_density=0.1, runs=40, seed=5)
        plain = gen.prim(S.T_I64, 200_000, uniq=1 << 40, seed=6)
        lz4 = gen.prim(S.T_I32, 200_000, uniq=300, runs=5, seed=7)
        dict_lz4 = gen.prim(S.T_F64, 128 * 300, uniq=200, null_density=0.1, runs=16, seed=8)
        r0 = ctx.replays()
        for _ in range(3):
            check(ctx, rle, max_page_size=65536, force_codec=S.RLE)          # (neither jobs nor tiles)
        r1 = ctx.replays()
        assert r1 == r0
        check(ctx, plain, max_page_size=65536, force_codec=S.NONE)            # tiles after all
        assert ctx.replays() == r1 + 1
        check(ctx, rle, max_page_size=65536, force_codec=S.RLE)
        check(ctx, rle, max_page_size=65536, force_codec=S.RLE)
        r2 = ctx.replays()
        check(ctx, lz4, max_page_size=65536, force_codec=S.LZ4)              # inflate jobs after all
        assert ctx.replays() == r2 + 1
        check(ctx, rle, max_page_size=65536, force_codec=S.RLE)
        check(ctx, rle, max_page_size=65536, force_codec=S.RLE)
        r3 = ctx.replays()
        check(ctx, dict_lz4, max_page_size=128 * 100, force_codec=S.DICT, force_index_codec=S.LZ4)   # a plan that needs inflated indices
        assert ctx.replays() == r3 + 1
        check(ctx, dict_lz4, max_page_size=128 * 100, force_codec=S.DICT, force_index_codec=S.LZ4)   # (now launched: no replay)
        assert ctx.replays() == r3 + 1
    finally:
        ctx.close()
