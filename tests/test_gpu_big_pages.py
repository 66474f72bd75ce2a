"""GPU parity for LONG pages: a column written without max_page_size is ONE page (src/write/common.rs:54-58), so a page can
hold millions of rows.  Pages of 2^18 rows or more are selected section-parallel (strawboat_amd/csrc/sb_select_big.h) and
their plain / LZ4 / Zstd blocks are written by many workgroups; whatever the path, the bytes must be the oracle's and the
pages must decode to the input.  (The shapes of tests/probes/big_pages.py at a size the oracle writes in seconds.)"""
import numpy as np
import pytest

from oracle import sbo as S
from tests import gen
from tests.test_gpu_decode import check as dec_check
from tests.test_gpu_freq import sparse
from tests.test_gpu_select import check as sel_check

pytestmark = pytest.mark.gpu

ROWS = 128 * 4800          # 614 400 rows = 38 sections of 16 384
ROWS_ODD = 700_001         # not a multiple of 128 (no bit-packing), a short last section


def cases():
    rng = np.random.default_rng(77)
    out = {
        "runs_i64": gen.prim(S.T_I64, ROWS, uniq=200, runs=50, seed=1),
        "short_runs_i32": gen.prim(S.T_I32, ROWS, uniq=40, runs=3, seed=11),
        "lowcard_i32": gen.prim(S.T_I32, ROWS, uniq=500, seed=2),
        "lowcard_nullable_f64": gen.prim(S.T_F64, ROWS_ODD, uniq=300, null_density=0.1, seed=3),
        "midcard_i64": gen.prim(S.T_I64, ROWS, uniq=150_000, seed=12),      # more keys than the LDS sets take, fewer than N / 3
        "highcard_i64": gen.prim(S.T_I64, ROWS, uniq=230_000, seed=13),     # just above / below Dict's limit of N / 3 distinct
        "sparse_i64": sparse(S.T_I64, ROWS, 0.02, 4),
        "sparse_f32": sparse(S.T_F32, ROWS_ODD, 0.05, 5),
        "mostly_null_i32": gen.prim(S.T_I32, ROWS, uniq=1000, null_density=0.93, seed=14),
        "random_u32": gen.prim(S.T_U32, ROWS, uniq=1 << 30, seed=5),
        "small_u32": gen.prim(S.T_U32, ROWS, uniq=1 << 9, seed=15),         # bit-packing territory
        "sorted_u32": gen.prim(S.T_U32, ROWS, uniq=1 << 28, sorted_=True, seed=16),
        "sorted_i64": gen.prim(S.T_I64, ROWS_ODD, uniq=1 << 40, sorted_=True, seed=6),
        "negative_i32": gen.prim(S.T_I32, ROWS, uniq=1 << 20, seed=17),
        "one_value_i64": gen.prim(S.T_I64, ROWS, uniq=1, seed=18),
        "one_value_then_other": None,
        "f64_random": gen.prim(S.T_F64, ROWS, uniq=1 << 30, seed=19),
    }
    v = np.full(ROWS, 7, np.int64)
    v[-3] = 9                                                            # all-equal fails in the LAST section only
    out["one_value_then_other"] = dict(ptype=S.T_I64, nullable=False, rows=ROWS, values=v, validity=None, offsets=None)
    k = rng.integers(0, 1 << 40, 3000)
    v = k[rng.integers(0, 3000, ROWS)].astype(np.int64)                 # > 2048 keys in the page, fewer in a section?  no: all
    out["union_overflows_i64"] = dict(ptype=S.T_I64, nullable=False, rows=ROWS, values=v, validity=None, offsets=None)
    # 1- and 2-byte values go through the same long-page kernels
    out["lowcard_i8"] = gen.prim(S.T_I8, ROWS, uniq=90, seed=41)
    out["runs_nullable_u8"] = gen.prim(S.T_U8, ROWS_ODD, uniq=60, runs=30, null_density=0.1, seed=42)
    out["midcard_u16"] = gen.prim(S.T_U16, ROWS, uniq=40_000, seed=43)          # more keys than the LDS sets take
    out["runs_i16"] = gen.prim(S.T_I16, ROWS, uniq=500, runs=20, seed=44)
    out["sparse_i16"] = sparse(S.T_I16, ROWS, 0.03, 45)
    out["runs_nullable_f64"] = gen.prim(S.T_F64, ROWS_ODD, uniq=50, runs=40, null_density=0.1, seed=31)
    c = gen.prim(S.T_I32, ROWS, uniq=30, runs=25, null_density=0.02, seed=32)
    bits = np.unpackbits(c["validity"], bitorder="little")[:ROWS].copy()
    bits[20_000:70_000] = 0            # whole sections without a valid row (runs carried across them)
    bits[300_000:300_010] = 0
    bits[:5] = 0                       # leading nulls join the first run
    bits[-40_000:] = 0                 # the page ends in nulls
    c["validity"] = np.packbits(bits, bitorder="little")
    out["runs_null_sections_i32"] = c
    c = gen.prim(S.T_I64, ROWS, uniq=9, runs=2000, null_density=0.5, seed=33)
    out["long_runs_half_null_i64"] = c
    v = out["midcard_i64"]["values"].copy()
    v[::1000] = -1                                                       # the all-ones key is the count table's "empty" word
    out["midcard_with_minus_one_i64"] = dict(ptype=S.T_I64, nullable=False, rows=ROWS, values=v, validity=None, offsets=None)
    v = (np.arange(ROWS) // 16384 * 50 + rng.integers(0, 50, ROWS)).astype(np.int32)   # 50 keys per section, 1 900 in the page
    out["sets_unite_i32"] = dict(ptype=S.T_I32, nullable=False, rows=ROWS, values=v, validity=None, offsets=None)
    v = (np.arange(ROWS) // 16384 * 60 + rng.integers(0, 60, ROWS)).astype(np.int32)   # 60 per section: 2 280 > 2 048 in the page
    out["sets_unite_overflow_i32"] = dict(ptype=S.T_I32, nullable=False, rows=ROWS, values=v, validity=None, offsets=None)
    # Dict pages written section-parallel (sb_dict_big.h): float keys are raw bits, NaNs never equal anything (every NaN row is
    # a dictionary entry of its own), +0.0 / -0.0 are two entries, a leading null interns T::default()
    v = rng.integers(0, 40, ROWS).astype(np.float64)
    v[rng.integers(0, ROWS, 300)] = np.nan
    v[rng.integers(0, ROWS, 50)] = np.frombuffer(np.uint64(0x7FF8000000000123).tobytes(), np.float64)[0]   # a NaN with a payload
    v[rng.integers(0, ROWS, 2000)] = -0.0
    out["lowcard_f64_nans_and_zeros"] = dict(ptype=S.T_F64, nullable=False, rows=ROWS, values=v, validity=None, offsets=None)
    c = gen.prim(S.T_I32, ROWS_ODD, uniq=700, null_density=0.3, seed=51)
    bits = np.unpackbits(c["validity"], bitorder="little")[:ROWS_ODD].copy()
    bits[:9] = 0                       # leading nulls: row 0 interns 0
    bits[100_000:140_000] = 0          # sections without a keyed row (the index is carried across them)
    c["validity"] = np.packbits(bits, bitorder="little")
    out["lowcard_null_sections_i32"] = c
    v = np.repeat(rng.integers(0, 3000, ROWS // 64 + 1), 64)[:ROWS].astype(np.int64)   # Dict whose indices come in runs (nested RLE)
    out["dict_runs_i64"] = dict(ptype=S.T_I64, nullable=False, rows=ROWS, values=v * 1_000_003, validity=None, offsets=None)
    v = rng.integers(0, 20, ROWS).astype(np.int64)
    v[rng.random(ROWS) < 0.97] = 5     # mostly one value with a maximum below 256: Dict with Freq-coded indices
    out["dict_freq_indices_i64"] = dict(ptype=S.T_I64, nullable=False, rows=ROWS, values=v, validity=None, offsets=None)
    # Freq pages prepared container-parallel (sb_freq_big.h); 65 536 exceptions or more: their block is selected and written
    # section-parallel too
    big = 1_600_000
    out["sparse_many_exceptions_i32"] = sparse(S.T_I32, big, 0.08, 61)              # 128 k random exceptions -> a plain / LZ4 block
    c = gen.prim(S.T_I64, big, uniq=5000, null_density=0.93, seed=62)               # >= 90 % null: every valid row is an exception
    out["mostly_null_many_exceptions_i64"] = c
    v = np.full(big, 77, np.int32)
    m = rng.random(big) < 0.07
    v[m] = np.repeat(rng.integers(1000, 1200, int(m.sum()) // 40 + 1), 40)[:int(m.sum())]   # exceptions in runs (nested RLE)
    out["sparse_exceptions_in_runs_i32"] = dict(ptype=S.T_I32, nullable=False, rows=big, values=v, validity=None, offsets=None)
    v = np.full(big, 1 << 20, np.int32)
    v[m] = rng.integers(100_000, 100_300, int(m.sum()))                             # low-cardinality exceptions (nested Dict: one workgroup)
    out["sparse_exceptions_lowcard_i32"] = dict(ptype=S.T_I32, nullable=False, rows=big, values=v, validity=None, offsets=None)
    v = np.full(big, 1 << 20, np.int32)
    mm = rng.random(big) < 0.09
    e = np.full(int(mm.sum()), 4242, np.int32)      # 144 k exceptions: the first quarter one value, the rest 60 % distinct values
    rare = rng.random(e.size) < 0.60                 # -> more than N / 3 distinct (no Dict), which only a count over ALL sections'
    rare[: e.size // 4] = False                      # rows sees (the count kernel runs BIG_COUNT_SPLIT workgroups per section)
    e[rare] = rng.integers(0, 1 << 30, int(rare.sum()))
    v[mm] = e
    out["sparse_exceptions_late_distinct_i32"] = dict(ptype=S.T_I32, nullable=False, rows=big, values=v, validity=None, offsets=None)
    # integers that never step down: the distinct keys are the runs, counted by the section kernel itself (no table pass)
    out["sorted_midcard_i64"] = dict(ptype=S.T_I64, nullable=False, rows=ROWS, values=np.sort(rng.integers(-50_000, 50_000, ROWS)).astype(np.int64),
                                     validity=None, offsets=None)                                  # 100 000 keys < N / 3: Dict is a candidate
    out["sorted_highcard_i64"] = dict(ptype=S.T_I64, nullable=False, rows=ROWS, values=np.sort(rng.integers(0, 230_000, ROWS)).astype(np.int64),
                                      validity=None, offsets=None)                                 # just around Dict's limit of N / 3
    out["sorted_midcard_u16"] = dict(ptype=S.T_U16, nullable=False, rows=ROWS_ODD, values=np.sort(rng.integers(0, 40_000, ROWS_ODD)).astype(np.uint16),
                                     validity=None, offsets=None)
    out["sorted_descending_i32"] = dict(ptype=S.T_I32, nullable=False, rows=ROWS, values=np.sort(rng.integers(0, 100_000, ROWS))[::-1].astype(np.int32).copy(),
                                        validity=None, offsets=None)                               # steps down: the count pass
    c = dict(ptype=S.T_I64, nullable=True, rows=ROWS, values=np.sort(rng.integers(0, 100_000, ROWS)).astype(np.int64),
             validity=np.packbits(rng.random(ROWS) < 0.9, bitorder="little"), offsets=None)
    out["sorted_midcard_nullable_i64"] = c
    v = np.full(big, 3.5, np.float64)
    v[rng.random(big) < 0.9] = 3.5
    v[m] = rng.random(int(m.sum()))
    v[::65536 * 3] = np.nan
    c = dict(ptype=S.T_F64, nullable=True, rows=big, values=v, validity=np.packbits(rng.random(big) < 0.97, bitorder="little"), offsets=None)
    out["sparse_nullable_f64"] = c
    return out


CASES = cases()
OPTS = [dict(ratio=2.0, forbidden=()), dict(ratio=1.2, default_compression=S.LZ4, forbidden=()),
        dict(ratio=3.0, default_compression=S.LZ4, forbidden=(S.DICT,))]
# the default encoder (parallel LZ4 matcher / Zstd frames per piece: chunk, plan and stitch kernels shared by many
# workgroups on a long page): format-valid streams of another parse, checked by structure and round trips
DEFAULT_ENC = {"random_u32": [dict(default_compression=S.LZ4), dict(ratio=3.0, default_compression=S.ZSTD, forbidden=(S.DICT,))],
               "sorted_i64": [dict(ratio=2.0, default_compression=S.LZ4), dict(default_compression=S.ZSTD)],
               "lowcard_nullable_f64": [dict(default_compression=S.LZ4), dict(default_compression=S.SNAPPY)],
               "runs_i64": [dict(default_compression=S.ZSTD), dict(default_compression=S.LZ4)]}


@pytest.mark.parametrize("name", sorted(CASES))
def test_one_page_columns_match_the_oracle(gpu_ctx, name):
    col = CASES[name]
    assert col["rows"] >= 1 << 18
    for opt in OPTS:
        sel_check(gpu_ctx, col, **opt)      # codecs, metas and bytes == the oracle's
    dec_check(gpu_ctx, col, **OPTS[0])


@pytest.mark.parametrize("name", sorted(DEFAULT_ENC))
def test_one_page_columns_default_encoder(gpu_ctx, name):
    from tests.test_gpu_configs import default_encoder_parity
    col = CASES[name]
    for opt in DEFAULT_ENC[name]:   # (bound: the chunked LZ4 matcher's small table against liblz4 on 8-byte floats: 1.96 x)
        default_encoder_parity(gpu_ctx, col, 2.2, **opt)


def test_one_page_binary_and_boolean_default_encoder(gpu_ctx):
    """the staging loops of the chunk planner (re-based offsets, re-packed bitmaps) are shared by many workgroups too"""
    from tests.test_gpu_configs import default_encoder_parity
    for col, opts in ((gen.binary(400_000, uniq=5000, zipf=1.2, maxlen=24, seed=8), (dict(default_compression=S.LZ4),)),
                      (gen.binary(300_000, uniq=250_000, maxlen=16, null_density=0.1, seed=9), (dict(default_compression=S.ZSTD),)),
                      (gen.boolean(3_000_001, null_density=0.05, runs=30, seed=7), (dict(default_compression=S.LZ4), dict(default_compression=S.ZSTD)))):
        for opt in opts:
            default_encoder_parity(gpu_ctx, col, 2.2, **opt)


def test_long_and_short_pages_in_one_call(gpu_ctx):
    """explicit paging that leaves long and short pages side by side, several columns of both widths in one call"""
    import torch
    from strawboat_amd import write, WriteOptions
    from tests.test_gpu_encode import to_device_column
    from strawboat_amd import read
    names = ["lowcard_i32", "midcard_i64", "random_u32", "sparse_i64", "small_u32", "runs_i64", "runs_null_sections_i32", "long_runs_half_null_i64"]
    cols = [CASES[n] for n in names]
    for mps in (None, 300_000, 262_144):
        wo = WriteOptions(default_compress_ratio=2.0, max_page_size=mps, lz4_exact=True, forbidden_compressions=[S.DICT] if mps == 300_000 else [])
        forb = (S.DICT,) if mps == 300_000 else ()
        encs = write.encode_columns(gpu_ctx, [to_device_column(gpu_ctx, c) for c in cols], wo)
        gpu_ctx.synchronize()
        for c, e in zip(cols, encs):
            want_pages, want_metas = gen.oracle_write(c, ratio=2.0, max_page_size=mps, forbidden=forb)
            assert np.array_equal(e.metas_array(), want_metas)
            assert np.array_equal(e.pages_numpy(), want_pages)
        # and back, all columns in one call (long RLE / Dict / plain pages next to short ones)
        cps = [read.ColumnPages(c["ptype"], c["nullable"], e.pages[:e.length], e.metas_array()) for c, e in zip(cols, encs)]
        got = read.batch_read_columns(gpu_ctx, cps)
        gpu_ctx.synchronize()
        for c, e, g in zip(cols, encs, got):
            want = gen.oracle_read(c, e.pages_numpy(), e.metas_array())
            assert np.array_equal(g.values_numpy(), want["values"])
            if c["nullable"]:
                assert np.array_equal(g.validity_numpy(), want["validity"])


def test_damaged_long_rle_page(gpu_ctx):
    """a long RLE page whose run counts were changed: the parts of the page agree with the oracle — an error, never a fault"""
    import torch
    from strawboat_amd import read
    from strawboat_amd._native import NativeError
    col = CASES["runs_i64"]
    pages, metas = gen.oracle_write(col, ratio=2.0, forbidden=())
    assert S.stat_column(col["ptype"], col["nullable"], pages, metas)[0].tolist() == [S.RLE]
    for pos, val in ((9 + 12 * 1000, 7), (9 + 12 * 5000, 0xFF), (pages.size - 12, 1)):   # a count in the middle, a huge one, the last run
        bad = pages.copy()
        bad[pos] = (int(bad[pos]) + val) & 0xFF
        try:
            want = gen.oracle_read(col, bad, metas)["values"]
        except Exception:
            want = None
        try:
            got = read.read_simple(gpu_ctx, read.ColumnPages(col["ptype"], False, torch.from_numpy(bad).to(gpu_ctx.torch_device), metas)).values_numpy()
        except NativeError:
            got = None
        if want is None:
            assert got is None, "the oracle refuses the page damaged at byte %d, the device decoded it" % pos
        else:
            assert got is not None and np.array_equal(got, want)
    dec_check(gpu_ctx, col, ratio=2.0, forbidden=())   # the context still works


def test_five_million_rows_one_page(gpu_ctx):
    """sections longer than 16 384 rows (256 sections at most per page)"""
    col = gen.prim(S.T_I32, 5_000_064, uniq=70_000, seed=21)
    sel_check(gpu_ctx, col, ratio=2.0, forbidden=())
    col = gen.prim(S.T_I64, 4_500_001, uniq=1 << 40, sorted_=True, seed=22)
    sel_check(gpu_ctx, col, ratio=2.0, default_compression=S.LZ4, forbidden=())


def _zstd_one_page_round_trip(ctx, values):
    """write one Basic(Zstd) page with the default encoder (one frame per 32 KiB piece), read it back"""
    from strawboat_amd import read, write, WriteOptions
    from tests.test_gpu_encode import to_device_column
    col = dict(ptype=S.T_U8, nullable=False, rows=values.size, values=values, validity=None, offsets=None)
    enc = write.write(ctx, to_device_column(ctx, col), WriteOptions(default_compression=S.ZSTD))
    assert enc.n_pages == 1
    pages, metas = enc.pages_numpy(), enc.metas_array()
    want = gen.oracle_read(col, pages, metas)["values"]           # the oracle reads what the device wrote
    assert np.array_equal(want, values)
    for _ in range(2):   # (the second read runs with the context knowing about Zstd: block pipeline + scan-based frame split)
        got = read.read_simple(ctx, read.ColumnPages(col["ptype"], False, enc.pages[:enc.length], metas))
        assert np.array_equal(got.values_numpy(), values)
    return pages, metas


def test_long_multi_frame_zstd_buffers_split_by_scan(gpu_ctx):
    """buffers of >= 1 MiB made of many frames: the frames are found by a scan for the frame magic (k_zsplit_scan /
    k_zsplit_chain).  Incompressible data is stored in raw blocks, so a magic number planted in the DATA shows up inside
    the frames: sparse plants are candidates off the chain, dense ones (> 7 per 16 KiB) send the buffer to the one-lane walk"""
    from strawboat_amd._native import NativeError
    from strawboat_amd import read
    rng = np.random.default_rng(99)
    magic = np.frombuffer((0xFD2FB528).to_bytes(4, "little"), np.uint8)
    plain = rng.integers(0, 256, 6_000_000).astype(np.uint8)
    _zstd_one_page_round_trip(gpu_ctx, plain)
    sparse_plants = plain.copy()
    for pos in range(1000, sparse_plants.size - 8, 50_001):
        sparse_plants[pos:pos + 4] = magic
    _zstd_one_page_round_trip(gpu_ctx, sparse_plants)
    dense = plain.copy()
    for pos in range(100, dense.size - 8, 997):
        dense[pos:pos + 4] = magic
    _zstd_one_page_round_trip(gpu_ctx, dense)
    text = np.frombuffer(b" ".join(b"w%d" % (i * 7919 % 5000) for i in range(1_500_000)), np.uint8).copy()   # frames with sequences
    pages, metas = _zstd_one_page_round_trip(gpu_ctx, text)
    # damage in the middle of the buffer: an error or the oracle's bytes, never a fault
    col = dict(ptype=S.T_U8, nullable=False, rows=text.size, values=text, validity=None, offsets=None)
    for pos in (pages.size // 2, pages.size // 3 + 5, 40):
        bad = pages.copy()
        bad[pos] ^= 0x40
        try:
            want = gen.oracle_read(col, bad, metas)["values"]
        except Exception:
            want = None
        try:
            import torch
            got = read.read_simple(gpu_ctx, read.ColumnPages(col["ptype"], False, torch.from_numpy(bad).to(gpu_ctx.torch_device), metas)).values_numpy()
        except NativeError:
            got = None
        if want is None:
            assert got is None
        elif got is not None:
            assert np.array_equal(got, want)
    _zstd_one_page_round_trip(gpu_ctx, plain[:2_000_000])   # the context still works


def binary_cases():
    rng = np.random.default_rng(91)
    out = {
        "utf8_zipf": gen.binary(ROWS_ODD, uniq=8000, zipf=1.2, maxlen=24, seed=71),                 # a Dict page, bit-packing impossible (rows % 128)
        "utf8_zipf_128": gen.binary(ROWS, uniq=5000, zipf=1.3, maxlen=16, seed=72),                 # Dict, bit-packed indices
        "utf8_lowcard_nullable": gen.binary(ROWS_ODD, uniq=300, null_density=0.15, maxlen=20, seed=73),
        "large_utf8_midcard": gen.binary(ROWS, uniq=120_000, maxlen=10, large=True, seed=74),       # 64-bit offsets, 120 k entries
        "utf8_unique": gen.binary(400_000, uniq=600_000, minlen=8, maxlen=14, seed=75),             # no Dict: most strings differ
        "utf8_one_value": gen.binary(300_000, uniq=1, minlen=5, maxlen=5, seed=76),
        "utf8_sparse": None,
        "utf8_leading_nulls": None,
        "utf8_runs": None,
    }
    c = gen.binary(ROWS, uniq=2, minlen=3, maxlen=3, seed=77)                                       # one string in 97 % of the rows -> Freq
    words = [b"top", b"x1", b"yy22", b"zzz333"]
    idx = np.where(rng.random(ROWS) < 0.97, 0, rng.integers(1, 4, ROWS))
    lens = np.array([len(w) for w in words], np.int64)[idx]
    offs = np.zeros(ROWS + 1, np.int64)
    np.cumsum(lens, out=offs[1:])
    out["utf8_sparse"] = dict(c, values=np.frombuffer(b"".join(words[i] for i in idx), np.uint8).copy(), offsets=offs.astype(np.int32))
    c = gen.binary(ROWS_ODD, uniq=900, null_density=0.3, maxlen=12, seed=78)
    bits = np.unpackbits(c["validity"], bitorder="little")[:ROWS_ODD].copy()
    bits[:7] = 0                        # row 0 is null: its slot's bytes are interned all the same (binary/dict.rs:55-93)
    bits[200_000:260_000] = 0           # sections without a keyed row
    c["validity"] = np.packbits(bits, bitorder="little")
    out["utf8_leading_nulls"] = c
    c = gen.binary(ROWS, uniq=4000, maxlen=9, seed=79)                                              # Dict whose indices come in runs (nested RLE)
    vocab_idx = np.repeat(rng.integers(0, 4000, ROWS // 50 + 1), 50)[:ROWS]
    voc = [("k%d" % i).encode() for i in range(4000)]
    lens = np.array([len(v) for v in voc], np.int64)[vocab_idx]
    offs = np.zeros(ROWS + 1, np.int64)
    np.cumsum(lens, out=offs[1:])
    out["utf8_runs"] = dict(c, values=np.frombuffer(b"".join(voc[i] for i in vocab_idx), np.uint8).copy(), offsets=offs.astype(np.int32))
    return out


BIN_CASES = binary_cases()


@pytest.mark.parametrize("name", sorted(BIN_CASES))
def test_one_page_binary_columns_match_the_oracle(gpu_ctx, name):
    """one-page ADAPTIVE Utf8 columns (Dict pages of 600 k+ rows, low cardinality, sparse, unique),
    bytes == the oracle's, through the section-parallel binary selector and Dict writer"""
    col = BIN_CASES[name]
    assert col["rows"] >= 1 << 18
    for opt in (dict(ratio=2.0, forbidden=()), dict(ratio=1.2, default_compression=S.LZ4, forbidden=())):
        sel_check(gpu_ctx, col, **opt)
    dec_check(gpu_ctx, col, ratio=2.0, forbidden=())


def test_one_page_binary_dict_falls_back_when_the_string_check_fails(gpu_ctx):
    """SB_WRITE_DEBUG_VERIFY_FAIL: the hash-formed dictionary of a long page is declared bad, the exact builder writes the page"""
    from tests.test_gpu_select import gpu_encode
    col = BIN_CASES["utf8_zipf"]
    want_pages, want_metas = gen.oracle_write(col, ratio=2.0, forbidden=())
    enc = gpu_encode(gpu_ctx, col, ratio=2.0, forbidden=(), debug_verify_fail=True)
    assert np.array_equal(enc.metas_array(), want_metas)
    assert np.array_equal(enc.pages_numpy(), want_pages)


def test_launch_hints_never_change_the_bytes(gpu_ctx):
    """a write call remembers which codecs the last call with the same plan chose and skips the long-page Dict / Freq
    launches nobody needed; a column of the same shape that needs them after all is written by the one-workgroup kernels:
    the same bytes either way.  Columns of one shape (plan) whose pages choose Dict, Freq, plain, Dict, Freq in turn."""
    from tests.test_gpu_select import gpu_encode
    rng = np.random.default_rng(123)
    n = ROWS
    sp = np.full(n, 1_000_000, np.int32)
    m = rng.random(n) < 0.03
    sp[m] = rng.integers(0, 1 << 30, int(m.sum()))
    variants = [rng.integers(0, 400, n).astype(np.int32), sp, rng.integers(0, 1 << 30, n).astype(np.int32),
                rng.integers(0, 300, n).astype(np.int32), sp[::-1].copy()]
    for k, v in enumerate(variants + variants[:2]):
        col = dict(ptype=S.T_I32, nullable=False, rows=n, values=v, validity=None, offsets=None)
        want_pages, want_metas = gen.oracle_write(col, ratio=2.0, forbidden=())
        enc = gpu_encode(gpu_ctx, col, ratio=2.0, forbidden=())
        assert np.array_equal(enc.metas_array(), want_metas), k
        assert np.array_equal(enc.pages_numpy(), want_pages), k


@pytest.mark.parametrize("nulls", [None, 0.1])
def test_several_long_binary_pages_per_column(gpu_ctx, nulls):
    """a binary page's slot starts behind the column's value bytes before it — at ANY byte: the long-page selector's records
    (8-byte words, targets of atomics) must not sit there unaligned.  Columns of four long pages each (the first page's slot
    is aligned by construction, the others are not), two columns in one call."""
    cols = [gen.binary(1_200_000, uniq=3000, zipf=1.2, maxlen=20, seed=5, null_density=nulls),
            gen.binary(900_001, uniq=40_000, maxlen=11, seed=6, null_density=nulls, large=True)]
    for col in cols:
        for opt in (dict(ratio=2.0, max_page_size=300_000, forbidden=()), dict(ratio=1.2, max_page_size=262_144, default_compression=S.LZ4, forbidden=())):
            sel_check(gpu_ctx, col, **opt)
        dec_check(gpu_ctx, col, ratio=2.0, max_page_size=300_000, forbidden=())


@pytest.mark.parametrize("rows", [262_144, 315_000, 393_217, 524_289])
def test_exact_count_table_fits_its_area(gpu_ctx, rows):
    """the exact distinct count of a long page works on a table of 8-byte keys in the page's aux area whose size is a power
    of two that depends on the row count, the sections and the workgroups per section: row counts on both sides of the
    rounding steps, keys that need the exact count (more than the section sets hold), Int64 and Utf8"""
    col = gen.prim(S.T_I64, rows, uniq=rows // 5, seed=rows % 97)
    sel_check(gpu_ctx, col, ratio=2.0, forbidden=())
    col = gen.binary(rows, uniq=rows // 40, maxlen=12, seed=rows % 89, null_density=0.05)
    sel_check(gpu_ctx, col, ratio=1.5, forbidden=())
    dec_check(gpu_ctx, col, ratio=1.5, forbidden=())


@pytest.mark.parametrize("shape", ["one_width", "head_grows", "changes_midway", "changes_often"])
@pytest.mark.parametrize("codec", [S.BITPACK, S.DELTABP])
def test_long_bitpacked_pages_block_widths(gpu_ctx, shape, codec):
    """a bit-packed page of 4096 blocks and more: the reader walks the head of the body block by block, takes the stretch of
    equal-width blocks behind it from k_bp_guess (checked by many workgroups) and guesses / walks the rest — whatever the
    widths do (src/compression/integer/bp.rs:67-94, delta_bp.rs:73-103: a width byte in front of every 128 values)"""
    from tests.test_gpu_decode import gpu_decode
    rng = np.random.default_rng(5)
    n = 128 * 6000 + (0 if codec == S.BITPACK else 0)
    if shape == "one_width":
        v = rng.integers(256, 512, n)
    elif shape == "head_grows":        # the ids of a Dict page: 7, 8, 9 ... bits in the first blocks
        v = np.minimum(rng.integers(0, 1 << 20, n), np.arange(n) // 64 + 1)
    elif shape == "changes_midway":    # the guessed stretch ends at block 3000
        v = rng.integers(0, 1 << 9, n)
        v[128 * 3000:] = rng.integers(0, 1 << 21, n - 128 * 3000)
    else:                              # a new width every few blocks: the guesses give up, one lane walks
        w = rng.integers(1, 25, n // 128 // 3 + 1).repeat(3)[: n // 128]
        v = rng.integers(0, 1 << 30, n) & ((1 << w.repeat(128)) - 1)
    if codec == S.DELTABP:
        v = np.cumsum(v % (1 << 12))    # (deltas of the widths above, sums that stay below 2^32)
    col = dict(ptype=S.T_U32, nullable=False, rows=n, values=v.astype(np.uint32), validity=None, offsets=None)
    page, metas = S.write_column(S.T_U32, False, n, col["values"], options=S.make_options(force_codec=codec))
    got = gpu_decode(gpu_ctx, col, page, metas)
    assert np.array_equal(got.values_numpy(), col["values"].view(np.uint8))


def test_long_bitpacked_page_with_a_damaged_width_byte(gpu_ctx):
    """a width byte > 32 in the guessed stretch, and one that makes the body end early: refused like the oracle refuses them"""
    from tests.test_gpu_decode import gpu_decode
    rng = np.random.default_rng(6)
    n = 128 * 5000
    col = dict(ptype=S.T_U32, nullable=False, rows=n, values=rng.integers(0, 1 << 9, n).astype(np.uint32), validity=None, offsets=None)
    page, metas = S.write_column(S.T_U32, False, n, col["values"], options=S.make_options(force_codec=S.BITPACK))
    for blk, byte in ((2500, 77), (4000, 32), (10, 1)):
        bad = np.array(page, dtype=np.uint8).copy()
        bad[9 + blk * (1 + 16 * 9)] = byte
        try:
            want = S.read_column(S.T_U32, False, bad, metas)["values"]
        except Exception:
            want = None
        try:
            got = gpu_decode(gpu_ctx, col, bad, metas).values_numpy()
        except Exception:
            got = None
        if want is None:
            assert got is None, (blk, byte)
        elif got is not None:
            assert np.array_equal(got, want.view(np.uint8)), (blk, byte)


def test_wrong_hints_are_replayed_not_walked(gpu_ctx):
    """a call skips the long-page Dict / Freq kernels that the last call with the same plan did not need; when a page needs
    them after all it is left undone, the device says so and sb_ctx_synchronize issues the interval again with everything
    launched (sb_ctx_replays counts): the bytes are the oracle's, and no call takes the one-workgroup walk (75 ms for a
    Freq page, 80 ms for a Dict page of this size before).  src/write/common.rs:54-58: one page per column is the default."""
    import os
    if os.environ.get("SB_NO_HINTS", "0") != "0":
        pytest.skip("SB_NO_HINTS: every kernel is launched, nothing to replay")
    import time
    from tests.test_gpu_select import gpu_encode
    rng = np.random.default_rng(321)
    n = 128 * 16000    # 2 M rows
    sp = np.full(n, 1_000_000, np.int32)
    m = rng.random(n) < 0.03
    sp[m] = rng.integers(0, 1 << 30, int(m.sum()))
    plain = rng.integers(0, 1 << 30, n).astype(np.int32)
    dic = rng.integers(0, 400, n).astype(np.int32)
    order = [plain, plain, dic, plain, sp, plain, dic, sp]
    want = {}
    for v in (plain, dic, sp):
        col = dict(ptype=S.T_I32, nullable=False, rows=n, values=v, validity=None, offsets=None)
        want[id(v)] = gen.oracle_write(col, ratio=2.0, forbidden=())
    r0 = gpu_ctx.replays()
    worst = 0.0
    for k, v in enumerate(order):
        col = dict(ptype=S.T_I32, nullable=False, rows=n, values=v, validity=None, offsets=None)
        t0 = time.perf_counter()
        enc = gpu_encode(gpu_ctx, col, ratio=2.0, forbidden=())
        dt = time.perf_counter() - t0
        if k:
            worst = max(worst, dt)
        want_pages, want_metas = want[id(v)]
        assert np.array_equal(enc.metas_array(), want_metas), k
        assert np.array_equal(enc.pages_numpy(), want_pages), k
    assert gpu_ctx.replays() - r0 >= 2, "the Dict and Freq pages behind plain ones were written without a replay?"
    # (host-side wall time of upload + encode + read-back of an 8 MB column: a replay doubles the ~1 ms of kernels)
    assert worst < 0.060, worst
