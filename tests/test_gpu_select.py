"""GPU parity of the adaptive selector: with default_compress_ratio = Some(r) the codec of every
page is chosen on the device (statistics + seeded sampling); the choice and the page bytes must
equal what the CPU oracle produces for the same options and seed."""
import numpy as np
import pytest

from oracle import sbo as S
from tests import gen
from tests.test_gpu_encode import gpu_encode

pytestmark = pytest.mark.gpu

NOT_ON_DEVICE = ()  # every codec has a device encoder now: the reference's default options (nothing forbidden)


def check(ctx, col, **opt):
    opt.setdefault("forbidden", NOT_ON_DEVICE)
    want_pages, want_metas = gen.oracle_write(col, **opt)
    want_codecs, want_inner = S.stat_column(col["ptype"], col["nullable"], want_pages, want_metas)
    enc = gpu_encode(ctx, col, **opt)
    got_metas = enc.metas_array()
    got = enc.pages_numpy()
    got_codecs, got_inner = S.stat_column(col["ptype"], col["nullable"], got, got_metas)
    assert got_codecs.tolist() == want_codecs.tolist(), "page codecs differ"
    assert got_inner.tolist() == want_inner.tolist(), "nested codecs differ"
    assert np.array_equal(got_metas, want_metas)
    assert np.array_equal(got, want_pages), "page bytes differ"
    return want_codecs


@pytest.mark.parametrize("ptype", [S.T_I8, S.T_I16, S.T_I32, S.T_U32, S.T_I64, S.T_U64, S.T_F32, S.T_F64, S.T_I128])
def test_prim_shapes(gpu_ctx, ptype):
    seen = set()
    for kw in (dict(uniq=1), dict(uniq=8), dict(uniq=200, runs=20), dict(uniq=1 << 20), dict(uniq=3, null_density=0.3),
               dict(uniq=50, runs=40, null_density=0.1), dict(uniq=1000, sorted_=True)):
        uq = min(kw.get("uniq", 1000), 100 if ptype == S.T_I8 else 1 << 30)
        col = gen.prim(ptype, 128 * 160, **dict(kw, uniq=uq))
        for ratio in (1.2, 2.0):
            seen |= set(check(gpu_ctx, col, max_page_size=128 * 64, ratio=ratio).tolist())
    assert len(seen) >= 2, "the shapes should exercise several codecs, got %s" % seen


def test_small_pages_use_whole_array_trials(gpu_ctx):
    # N / 10 <= 64: compress_sample_ratio compresses the whole page (no sampling)
    col = gen.prim(S.T_I32, 640 * 5, uniq=4, runs=6, null_density=0.1)
    check(gpu_ctx, col, max_page_size=640, ratio=1.5)
    col = gen.prim(S.T_U32, 512 * 4, uniq=1 << 10, sorted_=True)
    check(gpu_ctx, col, max_page_size=512, ratio=1.1)
    col = gen.prim(S.T_F64, 600 * 3, uniq=5, runs=9)
    check(gpu_ctx, col, max_page_size=600, ratio=1.5)


def test_seed_changes_sampling_not_parity(gpu_ctx):
    col = gen.prim(S.T_I32, 128 * 512, uniq=300, runs=3)
    for seed in (1, 2, 3):
        check(gpu_ctx, col, max_page_size=65536, ratio=1.3, rng_seed=seed)


def test_c2_adaptive(gpu_ctx):
    import bench
    vals, valid = bench.gen_c2_column(42)
    col = dict(ptype=S.T_F64, nullable=True, rows=vals.size, values=vals, validity=valid, offsets=None)
    codecs = check(gpu_ctx, col, max_page_size=65536, ratio=2.0)
    assert (codecs == S.RLE).all()


def test_boolean(gpu_ctx):
    for kw in (dict(p_true=1.0), dict(p_true=0.0, null_density=0.2), dict(runs=100), dict(runs=2), dict(runs=30, null_density=0.3)):
        col = gen.boolean(20_000, **kw)
        check(gpu_ctx, col, max_page_size=8192, ratio=1.5)
    check(gpu_ctx, gen.boolean(600, runs=50), max_page_size=600, ratio=1.2)


@pytest.mark.parametrize("large", [False, True])
def test_binary(gpu_ctx, large):
    for kw in (dict(uniq=1), dict(uniq=20, zipf=1.5), dict(uniq=5000), dict(uniq=50, null_density=0.2)):
        col = gen.binary(12_000, large=large, **kw)
        check(gpu_ctx, col, max_page_size=4096, ratio=1.5)


def test_unsupported_choice_is_loud(gpu_ctx):
    # a codec the column family does not have fails loudly instead of writing something else
    from strawboat_amd._native import NativeError
    with pytest.raises(NativeError) as e:
        gpu_encode(gpu_ctx, gen.boolean(8192, seed=5), force_codec=S.FREQ)   # "Unknown compression codec Freq for boolean"
    assert e.value.code == -1


def test_rle_chosen_after_the_speculation_stopped(gpu_ctx):
    """runs of ~2.5 rows: the fused select + RLE pass gives up writing records (< 4 rows per run), the
    selector still picks RLE (everything else is forbidden or worse), and the ordinary RLE kernel encodes"""
    for ptype in (S.T_I64, S.T_F64, S.T_I32, S.T_F32):
        col = gen.prim(ptype, 128 * 512, uniq=1 << 20, runs=3, null_density=0.05)
        codecs = check(gpu_ctx, col, max_page_size=128 * 128, ratio=1.1, forbidden=(S.DICT, S.FREQ, S.PATAS, S.BITPACK, S.DELTABP))
        assert (codecs == S.RLE).all()
    # and the other way round: long runs in a page whose first rows are all different
    v = np.concatenate([np.arange(3000), np.repeat(np.arange(50), 1000)]).astype(np.int64)
    col = dict(ptype=S.T_I64, nullable=False, rows=v.size, values=v, validity=None, offsets=None)
    check(gpu_ctx, col, ratio=1.5, forbidden=(S.DICT,))


def test_inputs_at_the_end_of_an_allocation(gpu_ctx):
    """pages whose last 4096-row chunk is short, values placed flush against the end of their device
    allocation: the kernels must not read past the column (a stray prefetch there is a GPU fault)"""
    import torch
    from strawboat_amd import write
    from strawboat_amd.types import WriteOptions
    rng = np.random.default_rng(9)
    seg = 20 << 20  # a whole allocator segment of its own
    for ptype, npt, rows in ((S.T_I32, np.int32, 4224), (S.T_I64, np.int64, 12416), (S.T_F64, np.float64, 16896)):
        w = np.dtype(npt).itemsize
        buf = torch.zeros(seg, dtype=torch.uint8, device=gpu_ctx.torch_device)
        v = rng.integers(0, 1000, rows).astype(npt)
        view = buf[seg - rows * w:]
        view.copy_(torch.from_numpy(v.view(np.uint8)))
        col = dict(ptype=ptype, nullable=False, rows=rows, values=v, validity=None, offsets=None)
        want_pages, want_metas = gen.oracle_write(col, max_page_size=65536, default_compression=S.LZ4, ratio=2.0)
        opts = WriteOptions(max_page_size=65536, default_compression=S.LZ4, default_compress_ratio=2.0, lz4_exact=True)
        enc = write.encode_columns(gpu_ctx, [write.DeviceColumn(ptype, False, rows, view)], opts)
        gpu_ctx.synchronize()
        assert np.array_equal(enc[0].pages_numpy(), want_pages)
        del buf


@pytest.mark.parametrize("rows", [1, 2, 63, 129, 255, 256])
def test_pages_smaller_than_a_workgroup(gpu_ctx, rows):
    """fewer rows than threads, every codec a candidate: the majority vote of the Freq ratio (wide integers,
    binary) must only look at rows of the page"""
    for col in (gen.prim(S.T_I128, rows, uniq=300, seed=rows), gen.prim(S.T_I256, rows, uniq=300, seed=rows + 1),
                gen.prim(S.T_I64, rows, uniq=300, seed=rows + 2), gen.prim(S.T_F64, rows, uniq=3, null_density=0.2, seed=rows + 3),
                gen.binary(rows, uniq=40, seed=rows + 4), gen.binary(rows, uniq=2, null_density=0.3, large=True, seed=rows + 5),
                gen.boolean(rows, seed=rows + 6)):
        check(gpu_ctx, col, ratio=2.0, forbidden=())
        check(gpu_ctx, col, ratio=1.1, default_compression=S.LZ4, forbidden=())


def test_binary_column_of_empty_strings(gpu_ctx):
    for rows in (1, 300):
        col = gen.binary(rows, uniq=1, maxlen=0)
        assert col["values"].size == 0
        check(gpu_ctx, col, ratio=2.0, forbidden=())
        check(gpu_ctx, col)


def test_run_level_kernel_shapes(gpu_ctx):
    """the lane = raw run kernel (sb_select_runs.h): null blocks around the 4096-row chunk seams, runs whose
    bits differ but whose keys are equal (+0 / -0, NaN payloads), a page that falls back to the row-level
    kernel half way, leading / trailing nulls, one long run"""
    rng = np.random.default_rng(77)
    N = 3 * 65536 + 5000

    def col_of(vals, valid_bool, ptype):
        validity = None if valid_bool is None else gen.pack_bits(valid_bool)
        return dict(ptype=ptype, nullable=validity is not None, rows=vals.size, values=vals, validity=validity, offsets=None)

    # null blocks that straddle chunk seams, values changing exactly at / next to the seams
    v = np.repeat(rng.integers(0, 50, N // 37 + 1), 37)[:N].astype(np.float64)
    ok = np.ones(N, bool)
    for seam in range(4096, N, 4096):
        a, b = seam - int(rng.integers(0, 9)), seam + int(rng.integers(0, 9))
        ok[a:b] = False
        if seam % 8192 == 0:
            v[seam - 1:seam + 40] = 1000.0 + seam     # a raw run that starts on the last row of a chunk
    check(gpu_ctx, col_of(v, ok, S.T_F64), max_page_size=65536, ratio=1.5, forbidden=())
    check(gpu_ctx, col_of(v.astype(np.int64), ok, S.T_I64), max_page_size=65536, ratio=1.5, forbidden=())
    check(gpu_ctx, col_of(v.astype(np.int32), ok, S.T_I32), max_page_size=65536, ratio=1.5, forbidden=())
    check(gpu_ctx, col_of(v.astype(np.float32), None, S.T_F32), max_page_size=65536, ratio=1.5, forbidden=())
    # equal keys with different bits: blocks of +0.0 / -0.0 and of NaNs with different payloads
    z = np.zeros(N, np.float64)
    z[(np.arange(N) // 16) % 2 == 1] = -0.0
    nan_bits = np.where((np.arange(N) // 24) % 2 == 0, 0x7FF8000000000000, 0xFFF8000000000123).astype(np.uint64)
    z2 = np.where((np.arange(N) // 3000) % 2 == 0, z, nan_bits.view(np.float64))
    okz = rng.random(N) < 0.9
    check(gpu_ctx, col_of(z2, okz, S.T_F64), max_page_size=65536, ratio=1.5, forbidden=())
    # half of a page in long runs, the other half random: the page leaves the run-level kernel half way
    h = np.concatenate([np.repeat(rng.integers(0, 9, 1024), 32), rng.integers(0, 1 << 40, 32768)]).astype(np.int64)
    check(gpu_ctx, col_of(h, rng.random(h.size) < 0.95, S.T_I64), max_page_size=65536, ratio=1.2, forbidden=())
    # leading and trailing nulls, an all-null page, one run
    w = np.full(2 * 65536, 7.5)
    okw = np.ones(w.size, bool)
    okw[:5000] = False
    okw[60000:70000] = False
    okw[-3:] = False
    check(gpu_ctx, col_of(w, okw, S.T_F64), max_page_size=65536, ratio=1.5, forbidden=())
    check(gpu_ctx, col_of(np.arange(70000, dtype=np.float64) // 1000, np.zeros(70000, bool), S.T_F64), max_page_size=65536, ratio=1.5, forbidden=())
    # sorted / unsorted / negative 4-byte integers in runs (Bitpacking and DeltaBitpacking eligibility)
    for arr in (np.repeat(np.arange(2048), 64), np.repeat(np.arange(2048)[::-1], 64), np.repeat(np.arange(-5, 2043), 64),
                np.repeat(np.r_[np.arange(1000), 999, np.arange(1000, 2047)], 64)):
        check(gpu_ctx, col_of(arr.astype(np.int32), None, S.T_I32), max_page_size=65536, ratio=1.05, forbidden=())
        check(gpu_ctx, col_of(arr.astype(np.uint32), None, S.T_U32), max_page_size=65536, ratio=1.05, forbidden=())


@pytest.mark.parametrize("nulls", [None, 0.15])
def test_hashed_binary_selection_redo_path(gpu_ctx, nulls):
    """Adaptive Utf8 pages count their keys with a tag table over 64-bit row hashes and hand the dictionary to the builder;
    a string check over every row guards it, and a page that fails is selected again exactly and built by the exact
    builder.  SB_WRITE_DEBUG_VERIFY_FAIL makes every page fail: the bytes must still be the oracle's (Dict pages with
    Bitpacking / LZ4 indices, a Freq-heavy column, a high-cardinality column that stays Basic)."""
    from tests.test_gpu_encode import gpu_encode
    cols = [gen.binary(200_000, uniq=3000, zipf=1.2, null_density=nulls, seed=11),
            gen.binary(150_000, uniq=40, zipf=3.0, null_density=nulls, seed=12),
            gen.binary(120_000, uniq=100_000, null_density=nulls, seed=13, maxlen=9)]
    for col in cols:
        opt = dict(max_page_size=65536, default_compression=S.LZ4, ratio=2.0)
        want_pages, want_metas = gen.oracle_write(col, **opt)
        for dbg in (False, True):
            enc = gpu_encode(gpu_ctx, col, debug_verify_fail=dbg, **opt)
            assert np.array_equal(enc.metas_array(), want_metas), "debug_verify_fail=%s" % dbg
            assert np.array_equal(enc.pages_numpy(), want_pages), "debug_verify_fail=%s" % dbg


@pytest.mark.parametrize("nulls", [None, 0.2])
def test_fused_page_kernels_and_the_kernel_chain_write_the_same_pages(nulls, monkeypatch):
    """Binary and 1- / 2- / 4-byte integer Dict pages of 512 ... 65 536 rows are selected and built by one 1024-thread
    workgroup per page (k_enc_bin_page / k_enc_prim_dict); SB_BIN_FUSED=0 (read at sb_ctx_create) sends them through the
    hash -> select -> verify -> emit chain, which still serves short / long pages and forced codecs.  Both write the
    oracle's bytes: Dict pages with bit-packed, RLE and LZ4-coded indices, strings longer than the 24 bytes the fused
    kernel holds in registers, a page that gives up on Dict midway, integer pages with more distinct values than the
    fused table takes."""
    import strawboat_amd as sb
    rng = np.random.default_rng(77)

    def icol(a, pt):
        validity = None if nulls is None else gen.pack_bits(rng.random(len(a)) >= nulls)
        return dict(ptype=pt, nullable=validity is not None, rows=a.size, values=a, validity=validity, offsets=None)

    cols = [gen.binary(200_000, uniq=3000, zipf=1.2, null_density=nulls, seed=21),
            gen.binary(140_000, uniq=300, zipf=1.1, null_density=nulls, seed=22, maxlen=60),     # slow rows: > 24 bytes
            gen.binary(131_072, uniq=100_000, null_density=nulls, seed=23, maxlen=9),           # stays Basic
            gen.binary(70_000, uniq=2, zipf=1.5, null_density=nulls, seed=24),
            icol(rng.integers(0, 500, 200_000).astype(np.int32), S.T_I32),
            icol(np.sort(rng.integers(0, 9000, 150_000)).astype(np.uint32), S.T_U32),
            icol(rng.integers(-300, 300, 100_000).astype(np.int16), S.T_I16),
            icol(rng.integers(0, 20_000, 131_072).astype(np.int32), S.T_I32),                     # > 10 240 keys per page
            icol(rng.integers(0, 100, 90_000).astype(np.int8), S.T_I8)]
    opt = dict(max_page_size=65536, default_compression=S.LZ4, ratio=2.0)
    want = [gen.oracle_write(c, **opt) for c in cols]
    for fused in ("1", "0"):
        monkeypatch.setenv("SB_BIN_FUSED", fused)
        ctx = sb.Context(0)
        try:
            for rep in range(2):                      # (the second call runs on the first one's launch hints)
                for c, (wp, wm) in zip(cols, want):
                    enc = gpu_encode(ctx, c, **opt)
                    assert np.array_equal(enc.metas_array(), wm), "SB_BIN_FUSED=%s" % fused
                    assert np.array_equal(enc.pages_numpy(), wp), "SB_BIN_FUSED=%s" % fused
        finally:
            ctx.close()


def test_a_skipped_dict_emitter_is_replayed(gpu_ctx):
    """k_enc_bin_page finishes the Dict pages whose index block it bit-packs, and a call leaves k_enc_emit_pages<-4, Dict> out
    when the last call with the same plan needed it for no page (codec_counts[30]).  A column of the same shape whose
    indices come in runs (an RLE index block: the emitter's job) then finds the page unwritten: the interval is issued again
    with everything launched, the bytes are the oracle's."""
    import os
    if os.environ.get("SB_NO_HINTS", "0") != "0" or os.environ.get("SB_BIN_FUSED", "1") == "0":
        pytest.skip("the switches launch the emitter every time")
    rng = np.random.default_rng(99)
    n = 2 * 65536                                   # full pages only: row counts that are multiples of 128
    words = np.array([b"w%05d" % k for k in range(800)], dtype=object)

    def col_of(ids):
        vals = b"".join(words[i] for i in ids)
        offs = np.zeros(n + 1, np.int32)
        offs[1:] = np.cumsum([len(words[i]) for i in ids])
        return dict(ptype=S.T_BIN32, nullable=False, rows=n, values=np.frombuffer(vals, np.uint8),
                    validity=None, offsets=offs)

    scattered = col_of(rng.integers(0, 800, n))                       # bit-packed indices: finished by k_enc_bin_page
    runs = col_of(np.repeat(rng.integers(0, 800, n // 64), 64))       # indices in runs of 64: an RLE index block
    opt = dict(max_page_size=65536, ratio=2.0, forbidden=())
    want_s, want_r = gen.oracle_write(scattered, **opt), gen.oracle_write(runs, **opt)
    inner_s = S.stat_column(scattered["ptype"], False, *want_s)[1].tolist()
    inner_r = S.stat_column(runs["ptype"], False, *want_r)[1].tolist()
    assert inner_s != inner_r, "the two columns were meant to choose different index codecs"
    r0 = gpu_ctx.replays()
    for k, (col, want) in enumerate([(scattered, want_s), (scattered, want_s), (runs, want_r), (scattered, want_s), (runs, want_r)]):
        enc = gpu_encode(gpu_ctx, col, **opt)
        assert np.array_equal(enc.metas_array(), want[1]), k
        assert np.array_equal(enc.pages_numpy(), want[0]), k
    assert gpu_ctx.replays() > r0, "the page behind a skipped emitter was written without a replay?"
    # a writer that alternates the two kinds under one plan keeps both sets of kernels (the hints are what the last TWO calls
    # with the plan needed): no replay every other call
    r1 = gpu_ctx.replays()
    for k, (col, want) in enumerate([(scattered, want_s), (runs, want_r)] * 3):
        enc = gpu_encode(gpu_ctx, col, **opt)
        assert np.array_equal(enc.pages_numpy(), want[0]), k
    assert gpu_ctx.replays() == r1, "alternating data under one plan is replayed again and again"
