"""GPU parity for the block-parallel Zstd decoder (strawboat_amd/csrc/sb_zstd_blocks.h): pages whose blocks are frames
written by the real libzstd (what zstd::bulk::compress_to_buffer produces at the reference's call site,
src/compression/basic.rs:122-135: ONE frame per buffer, many <= 128 KiB blocks), decoded block by block on the device.
The shapes aim at what a block inherits from the blocks before it — repeat offsets (periodic data), Huffman trees
("treeless" literal sections), repeated FSE tables — and at the paths around the fast one: raw / RLE blocks inside a
frame, matches longer than a block, offsets beyond 2^22 (more than 64 bits in one sequence), frames the pipeline hands
back.  Every result is compared with the oracle's decode of the same pages (pinned against libzstd) and with the input."""
import os

import numpy as np
import pytest

from oracle import sbo as S
from tests import gen
from tests.test_gpu_decode import gpu_decode
from tests.test_gpu_zstd import recompress

pytestmark = pytest.mark.gpu


def _ctx(mode):
    import strawboat_amd as sb
    old = os.environ.get("SB_ZSTD_BLOCKS")
    if mode is None:
        os.environ.pop("SB_ZSTD_BLOCKS", None)
    else:
        os.environ["SB_ZSTD_BLOCKS"] = mode
    try:
        return sb.Context(0)
    finally:
        if old is None:
            os.environ.pop("SB_ZSTD_BLOCKS", None)
        else:
            os.environ["SB_ZSTD_BLOCKS"] = old


@pytest.fixture(scope="module")
def zb_ctx():
    ctx = _ctx("1")
    yield ctx
    ctx.close()


def i64col(v):
    v = np.ascontiguousarray(v, dtype=np.int64)
    return dict(ptype=S.T_I64, nullable=False, rows=v.size, values=v, validity=None, offsets=None)


def u8col(v):
    v = np.ascontiguousarray(v, dtype=np.uint8)
    return dict(ptype=S.T_U8, nullable=False, rows=v.size, values=v, validity=None, offsets=None)


def roundtrip(ctx, col, level=3, **opt):
    pages, metas = gen.oracle_write(col, **opt)
    zp, zm = recompress(col, pages, metas, level)
    want = gen.oracle_read(col, zp, zm)
    got = gpu_decode(ctx, col, zp, zm)
    assert np.array_equal(got.values_numpy(), want["values"]), "values differ from the oracle's decode"
    assert np.array_equal(want["values"], np.ascontiguousarray(col["values"]).view(np.uint8)), "oracle decode != input"
    if col["nullable"]:
        assert np.array_equal(got.validity_numpy(), want["validity"])
    if col["offsets"] is not None:
        assert np.array_equal(got.offsets_numpy(), want["offsets"])
    return zp, zm


def shapes():
    rng = np.random.default_rng(2024)
    out = {}
    # literals-dominated blocks, trees reused by the blocks that follow (5 random bytes + 3 sign bytes per value)
    out["i64_40bit"] = i64col(rng.integers(-2**40, 2**40, 300_000))
    # short matches everywhere: ~30 000 sequences per 128 KiB block, offsets repeat (the same few distances)
    words = [b"s%d" % k for k in range(500)]
    out["words"] = u8col(np.frombuffer(b"".join(words[i] for i in rng.integers(0, 500, 250_000)), np.uint8))
    # a period of 24 bytes with noise: repeat-offset codes carried across block borders
    base = rng.integers(0, 256, 24).astype(np.uint8)
    per = np.tile(base, 40_000)
    noise = rng.random(per.size) < 0.01
    per[noise] = rng.integers(0, 256, int(noise.sum()))
    out["periodic"] = u8col(per)
    # runs: RLE blocks and matches longer than a block (one sequence of > 64 KiB)
    out["zeros_then_noise"] = u8col(np.concatenate([np.zeros(400_000, np.uint8), rng.integers(0, 256, 100_000).astype(np.uint8),
                                                    np.full(300_000, 7, np.uint8)]))
    # incompressible: raw blocks inside the frame
    out["random"] = u8col(rng.integers(0, 256, 400_000))
    # text-like, long-ish matches, a mix of literal lengths
    syll = [bytes(rng.integers(97, 123, int(rng.integers(2, 9))).astype(np.uint8)) for _ in range(3000)]
    zipf = np.minimum(rng.zipf(1.3, 120_000), 3000) - 1
    out["zipf_text"] = u8col(np.frombuffer(b" ".join(syll[i] for i in zipf), np.uint8))
    # small integers: few distinct byte values, Huffman codes of 1-3 bits, many sequences
    out["small_ints"] = i64col(rng.integers(0, 12, 200_000))
    out["sorted"] = i64col(np.sort(rng.integers(0, 1 << 33, 250_000)))
    return out


SHAPES = shapes()


@pytest.mark.parametrize("level", [1, 3, 9, 19])
@pytest.mark.parametrize("name", sorted(SHAPES))
def test_libzstd_frames_block_parallel(zb_ctx, name, level):
    col = SHAPES[name]
    before = zb_ctx.zstd_block_stats()
    roundtrip(zb_ctx, col, level)                                   # one page = one frame of many blocks
    roundtrip(zb_ctx, col, level, max_page_size=max(1, col["rows"] // 7))
    after = zb_ctx.zstd_block_stats()
    assert after[0] > before[0], "the block pipeline decoded no frame"
    assert after[1] == before[1], "a well-formed libzstd frame was handed back to the frame-serial decoder"


def test_nullable_and_binary_columns(zb_ctx):
    roundtrip(zb_ctx, gen.prim(S.T_F64, 200_000, uniq=300, runs=3, null_density=0.1), max_page_size=65536)
    roundtrip(zb_ctx, gen.prim(S.T_I32, 400_000, uniq=1 << 12, sorted_=True), max_page_size=262144)
    roundtrip(zb_ctx, gen.boolean(1_000_000, null_density=0.2, runs=5), max_page_size=500_000)
    roundtrip(zb_ctx, gen.binary(300_000, uniq=400, null_density=0.1, zipf=1.2), max_page_size=100_000)
    roundtrip(zb_ctx, gen.binary(100_000, uniq=50_000, large=True), max_page_size=30_000)


def test_offsets_beyond_4_mib_take_the_second_window(zb_ctx):
    """a match 9 MB back: offset code 23 + 16 extra bits of a long match + the state bits do not fit 64 bits"""
    rng = np.random.default_rng(5)
    head = rng.integers(0, 256, 3_000_000).astype(np.uint8)
    mid = np.repeat(rng.integers(0, 256, 60_000).astype(np.uint8), 100)     # cheap filler, compressible
    data = np.concatenate([head, mid, head[:2_500_000], rng.integers(0, 4, 100_000).astype(np.uint8), head[100_000:900_000]])
    pa = pytest.importorskip("pyarrow")
    # (window log 23+ needs a level whose window covers 9 MB: 19 has 8 MiB, ultra levels more; long matches are found anyway
    # inside the window, so check only that whatever libzstd wrote decodes)
    for level in (3, 19, 22):
        roundtrip(zb_ctx, u8col(data), level)


def test_many_frames_in_one_call(zb_ctx):
    """64 pages = 64 frames of 3-4 blocks in one call, next to LZ4 and uncompressed pages of other columns"""
    import torch
    from strawboat_amd import read
    rng = np.random.default_rng(11)
    cols, pages = [], []
    for k in range(6):
        c = i64col(rng.integers(-2**(20 + 4 * k), 2**(20 + 4 * k), 400_000))
        p, m = gen.oracle_write(c, max_page_size=50_000)
        if k % 3 == 0:
            p, m = recompress(c, p, m, 3)
        elif k % 3 == 1:
            p, m = gen.oracle_write(c, max_page_size=50_000, default_compression=S.LZ4)
        cols.append(c)
        pages.append((p, m))
    w = gen.binary(200_000, uniq=700, zipf=1.1)
    p, m = gen.oracle_write(w, max_page_size=40_000)
    cols.append(w)
    pages.append(recompress(w, p, m, 3))
    cps = [read.ColumnPages(c["ptype"], c["nullable"], torch.from_numpy(p).to(zb_ctx.torch_device), m) for c, (p, m) in zip(cols, pages)]
    got = read.batch_read_columns(zb_ctx, cps)
    zb_ctx.synchronize()
    for c, g, (p, m) in zip(cols, got, pages):
        want = gen.oracle_read(c, p, m)
        assert np.array_equal(g.values_numpy(), want["values"])
        if c["offsets"] is not None:
            assert np.array_equal(g.offsets_numpy(), want["offsets"])


def test_corrupted_frames_agree_with_the_oracle(zb_ctx):
    """single-byte damage anywhere in a multi-block frame: the device either reports an error or — frames carry no
    checksum — decodes the damaged stream to exactly what the oracle's decoder makes of it; never a fault or a hang"""
    from strawboat_amd._native import NativeError
    rng = np.random.default_rng(3)
    col = SHAPES["zipf_text"]
    pages, metas = gen.oracle_write(col)
    zp, zm = recompress(col, pages, metas, 3)
    n_err = n_same = 0
    positions = np.concatenate([np.arange(9, 60), rng.integers(60, zp.size, 150)])
    for pos in positions:
        bad = zp.copy()
        bad[pos] ^= 1 << int(rng.integers(0, 8))
        try:
            want = gen.oracle_read(col, bad, zm)["values"]
        except Exception:
            want = None
        try:
            got = gpu_decode(zb_ctx, col, bad, zm).values_numpy()
        except NativeError:
            got = None
        if want is None:
            assert got is None, "the oracle refuses the frame damaged at byte %d, the device decoded it" % pos
            n_err += 1
        elif got is not None:
            assert np.array_equal(got, want), "damage at byte %d: device and oracle decode differently" % pos
            n_same += 1
        else:
            n_err += 1   # (the device may refuse what the oracle's laxer checks let through)
    assert n_err > 20
    roundtrip(zb_ctx, col)   # the context still works


def test_auto_mode_learns_from_the_first_call():
    """without SB_ZSTD_BLOCKS a context launches the block pipeline once it has met a Zstd buffer (the kinds word comes back
    with every synchronize): the first call decodes frame-serially, the second block-parallel — same bytes"""
    import os
    if os.environ.get("SB_NO_HINTS", "0") != "0":
        pytest.skip("SB_NO_HINTS: the pipeline is launched from the first call")
    ctx = _ctx(None)
    try:
        col = SHAPES["words"]
        roundtrip(ctx, col)
        s1 = ctx.zstd_block_stats()
        roundtrip(ctx, col)
        s2 = ctx.zstd_block_stats()
        assert s1[0] == 0 and s2[0] > 0, (s1, s2)
    finally:
        ctx.close()


def test_pipeline_off_matches_on():
    """SB_ZSTD_BLOCKS=0 keeps the frame-serial decoder: the two must agree on every shape (cheap cross-check of both)"""
    ctx = _ctx("0")
    try:
        for name in ("i64_40bit", "words", "periodic"):
            roundtrip(ctx, SHAPES[name])
        assert ctx.zstd_block_stats()[0] == 0
    finally:
        ctx.close()


def test_partial_pool_exhaustion_on_a_reused_context():
    """a frame whose pool reservation fails part-way must leave INERT block descriptors behind — the block
    pool is not cleared between calls, so the slots it gave up still hold the previous call's descriptors, which name frame
    indices that are live in this call.  A context first decodes a large call (the pools fill with descriptors), then — with
    the pool estimates divided so that only some frames fit — calls with many tiny blocks and many sequences per byte; the
    refused frames go through the frame-serial decoder, and every column must equal the oracle's decode."""
    import torch
    from strawboat_amd import read
    rng = np.random.default_rng(77)
    old = os.environ.get("SB_ZSTD_BLOCKS_POOL_DIV")
    try:
        for div in ("1", "6", "40", "400"):
            os.environ["SB_ZSTD_BLOCKS_POOL_DIV"] = div
            ctx = _ctx("1")
            try:
                # 1. a large call: descriptors of many frames / blocks stay in the pool afterwards
                roundtrip(ctx, SHAPES["words"], 3, max_page_size=SHAPES["words"]["rows"] // 9)
                roundtrip(ctx, SHAPES["zipf_text"], 3)
                # 2. calls of several columns in which only part of the frames fits the (divided) pools
                cols, pages = [], []
                for k in range(5):
                    if k % 2 == 0:   # many sequences per stream byte: short repeats of few words
                        words = [b"w%d" % j for j in range(30)]
                        c = u8col(np.frombuffer(b"".join(words[i] for i in rng.integers(0, 30, 120_000)), np.uint8))
                    else:            # highly compressible: 128 KiB blocks of a few hundred stream bytes
                        c = u8col(np.tile(rng.integers(0, 256, 100 + k).astype(np.uint8), 4000))
                    p, m = gen.oracle_write(c, max_page_size=c["rows"] // (3 + k))
                    cols.append(c)
                    pages.append(recompress(c, p, m, 3 if k < 3 else 19))
                before = ctx.zstd_block_stats()
                for _ in range(2):
                    cps = [read.ColumnPages(c["ptype"], c["nullable"], torch.from_numpy(p).to(ctx.torch_device), m) for c, (p, m) in zip(cols, pages)]
                    got = read.batch_read_columns(ctx, cps)
                    ctx.synchronize()
                    for c, g, (p, m) in zip(cols, got, pages):
                        want = gen.oracle_read(c, p, m)
                        assert np.array_equal(g.values_numpy(), want["values"]), "pool divisor %s: a column differs from the oracle's decode" % div
                after = ctx.zstd_block_stats()
                if div == "400":
                    assert after[1] > before[1], "no frame was refused although the pools were cut to 1/400"
                if div == "1":
                    assert after[1] == before[1] and after[0] > before[0]
            finally:
                ctx.close()
    finally:
        if old is None:
            os.environ.pop("SB_ZSTD_BLOCKS_POOL_DIV", None)
        else:
            os.environ["SB_ZSTD_BLOCKS_POOL_DIV"] = old
