"""GPU parity for Freq pages of primitives (codec id 13, src/compression/integer/freq.rs:90-127,
double/freq.rs): pages written by the oracle (top value | Roaring bitmap | nested exceptions block),
decoded on the device in two passes (exceptions block through the normal decoder, then a scatter),
compared with the oracle's decode."""
import numpy as np
import pytest

from oracle import sbo as S
from tests import gen
from tests.test_gpu_decode import check, gpu_decode

pytestmark = pytest.mark.gpu

NP = {S.T_U8: np.uint8, S.T_I32: np.int32, S.T_I64: np.int64, S.T_F64: np.float64, S.T_F32: np.float32, S.T_I16: np.int16}


def sparse(ptype, rows, p_exc, seed, null_density=None, exc_uniq=1 << 20, top=7):
    rng = np.random.default_rng(seed)
    v = np.full(rows, top, dtype=NP[ptype])
    exc = rng.random(rows) < p_exc
    hi = min(exc_uniq, 100 if ptype == S.T_U8 else exc_uniq)
    v[exc] = (rng.integers(300 if ptype != S.T_U8 else 8, 300 + hi if ptype != S.T_U8 else 8 + hi, int(exc.sum()))).astype(NP[ptype])
    validity = gen.make_validity(rng, rows, null_density)
    return dict(ptype=ptype, nullable=validity is not None, rows=rows, values=v, validity=validity, offsets=None)


@pytest.mark.parametrize("ptype", [S.T_U8, S.T_I16, S.T_I32, S.T_I64, S.T_F32, S.T_F64])
def test_freq_pages(gpu_ctx, ptype):
    check(gpu_ctx, sparse(ptype, 20_000, 0.05, 1), max_page_size=4096, force_codec=S.FREQ)
    check(gpu_ctx, sparse(ptype, 20_000, 0.05, 2, null_density=0.1), max_page_size=5000, force_codec=S.FREQ)
    check(gpu_ctx, sparse(ptype, 9_000, 0.0, 3), max_page_size=3000, force_codec=S.FREQ)            # no exceptions at all


def test_freq_mostly_null_pages(gpu_ctx):
    # >= 90 % nulls: the top value "is null" and every valid row is an exception (freq.rs:46-48)
    col = gen.prim(S.T_I64, 30_000, uniq=1000, null_density=0.95, seed=4)
    check(gpu_ctx, col, max_page_size=8192, force_codec=S.FREQ)


def test_freq_roaring_containers(gpu_ctx):
    # > 4096 exceptions inside one 64 Ki-row container -> bitmap container; 200 000-row pages -> 4 containers
    check(gpu_ctx, sparse(S.T_I32, 65_536, 0.09, 5), max_page_size=65_536, force_codec=S.FREQ)
    check(gpu_ctx, sparse(S.T_I64, 400_000, 0.08, 6), max_page_size=200_000, force_codec=S.FREQ)


def test_freq_nested_exception_codecs(gpu_ctx):
    # the exceptions block is an ordinary adaptive block: few distinct values -> Dict / RLE / bit-packing, LZ4 default ...
    seen = set()
    for kw, opt in ((dict(exc_uniq=3), dict(ratio=1.2)), (dict(exc_uniq=1 << 20), dict(default_compression=S.LZ4)),
                    (dict(exc_uniq=200), dict(ratio=1.1, default_compression=S.ZSTD)), (dict(exc_uniq=1), dict(ratio=1.5))):
        col = sparse(S.T_I32, 128 * 700, 0.07, 7, **kw)
        pages, metas = check(gpu_ctx, col, max_page_size=128 * 350, force_codec=S.FREQ, **opt)
        seen |= set(S.stat_column(col["ptype"], col["nullable"], pages, metas)[1].tolist())
    assert len(seen) >= 3, "the nested exceptions blocks should use several codecs, got %s" % seen


def test_freq_and_other_pages_in_one_batch(gpu_ctx):
    import torch
    from strawboat_amd import read
    cols, want = [], []
    for k, (col, opt) in enumerate([(sparse(S.T_I64, 30_000, 0.04, 8), dict(force_codec=S.FREQ)),
                                    (gen.prim(S.T_I64, 30_000, uniq=50, runs=9), dict(force_codec=S.RLE)),
                                    (sparse(S.T_F64, 30_000, 0.06, 9, null_density=0.2), dict(force_codec=S.FREQ))]):
        pages, metas = gen.oracle_write(col, max_page_size=8192, **opt)
        want.append(gen.oracle_read(col, pages, metas))
        cols.append(read.ColumnPages(col["ptype"], col["nullable"], torch.from_numpy(pages).to(gpu_ctx.torch_device), metas))
    got = read.batch_read_columns(gpu_ctx, cols)
    gpu_ctx.synchronize()
    for g, w in zip(got, want):
        assert np.array_equal(g.values_numpy(), w["values"])


# ---- encode: Freq pages written on the device equal the oracle's (integer/freq.rs:34-88)
def test_freq_encode_matches_oracle(gpu_ctx):
    from tests.test_gpu_encode import check as enc_check
    for ptype in (S.T_U8, S.T_I16, S.T_I32, S.T_I64, S.T_F32, S.T_F64):
        enc_check(gpu_ctx, sparse(ptype, 20_000, 0.05, 21), max_page_size=4096, force_codec=S.FREQ)
        enc_check(gpu_ctx, sparse(ptype, 20_000, 0.05, 22, null_density=0.1), max_page_size=5000, force_codec=S.FREQ)
        enc_check(gpu_ctx, sparse(ptype, 9_000, 0.0, 23), max_page_size=3000, force_codec=S.FREQ)      # no exceptions
    enc_check(gpu_ctx, gen.prim(S.T_I64, 30_000, uniq=1000, null_density=0.95, seed=24), max_page_size=8192, force_codec=S.FREQ)
    enc_check(gpu_ctx, sparse(S.T_I32, 65_536, 0.09, 25), max_page_size=65_536, force_codec=S.FREQ)   # bitmap container
    enc_check(gpu_ctx, sparse(S.T_I64, 400_000, 0.08, 26), max_page_size=200_000, force_codec=S.FREQ)  # 4 containers


def test_freq_encode_nested_codecs(gpu_ctx):
    from tests.test_gpu_encode import check as enc_check
    for kw, opt in ((dict(exc_uniq=3), dict(ratio=1.2)), (dict(exc_uniq=1 << 20), dict(default_compression=S.LZ4)),
                    (dict(exc_uniq=200), dict(ratio=1.1, default_compression=S.ZSTD)), (dict(exc_uniq=1), dict(ratio=1.5)),
                    (dict(exc_uniq=50), dict(force_index_codec=S.RLE)), (dict(exc_uniq=50), dict(force_index_codec=S.DICT))):
        if opt.get("default_compression") == S.ZSTD:
            continue  # Zstd blocks are format-valid but not byte-pinned
        enc_check(gpu_ctx, sparse(S.T_I32, 128 * 700, 0.07, 27, **kw), max_page_size=128 * 350, force_codec=S.FREQ, **opt)


def test_adaptive_selection_with_nothing_forbidden(gpu_ctx):
    """the reference's default options: every codec is a candidate, and sparse pages come out as Freq"""
    from tests.test_gpu_select import check as sel_check
    seen = set()
    for col in (sparse(S.T_I64, 128 * 300, 0.03, 31), sparse(S.T_F64, 128 * 300, 0.04, 32, null_density=0.05),
                sparse(S.T_I32, 128 * 300, 0.02, 33, top=1000), gen.prim(S.T_I64, 128 * 300, uniq=500, null_density=0.95, seed=34),
                gen.prim(S.T_I32, 128 * 300, uniq=40, runs=30, seed=35)):
        for ratio in (1.2, 2.0):
            seen |= set(sel_check(gpu_ctx, col, max_page_size=128 * 100, ratio=ratio, forbidden=()).tolist())
    assert S.FREQ in seen and len(seen) >= 2, seen


# ---- binary / Utf8 Freq pages (binary/freq.rs:44-145) and 128- / 256-bit integers
def sparse_bin(rows, p_exc, seed, null_density=None, large=False, top=b"the-common-value", exc_uniq=500, maxlen=20):
    rng = np.random.default_rng(seed)
    exc = rng.random(rows) < p_exc
    vocab = [(b"x%d-" % i) * int(rng.integers(0, maxlen // 3 + 1)) for i in range(exc_uniq)]
    pick = rng.integers(0, exc_uniq, rows)
    items = [vocab[pick[i]] if exc[i] else top for i in range(rows)]
    lens = np.fromiter((len(b) for b in items), np.int64, rows)
    offs = np.zeros(rows + 1, np.int64)
    np.cumsum(lens, out=offs[1:])
    data = np.frombuffer(b"".join(items), np.uint8).copy() if rows else np.zeros(0, np.uint8)
    validity = gen.make_validity(rng, rows, null_density)
    return dict(ptype=S.T_BIN64 if large else S.T_BIN32, nullable=validity is not None, rows=rows, values=data,
                validity=validity, offsets=offs.astype(np.int64 if large else np.int32))


def sparse_wide(ptype, rows, p_exc, seed, null_density=None):
    rng = np.random.default_rng(seed)
    w = S.WIDTH[ptype] // 8
    vals = np.zeros((rows, w), np.int64)
    vals[:, 0] = 1000
    vals[:, -1] = 5
    exc = rng.random(rows) < p_exc
    vals[exc, 0] = rng.integers(300, 100000, int(exc.sum()))
    vals[exc, -1] = -rng.integers(0, 3, int(exc.sum()))
    validity = gen.make_validity(rng, rows, null_density)
    return dict(ptype=ptype, nullable=validity is not None, rows=rows, values=vals.reshape(-1), validity=validity, offsets=None)


@pytest.mark.parametrize("large", [False, True])
def test_binary_freq_decode(gpu_ctx, large):
    check(gpu_ctx, sparse_bin(20_000, 0.05, 41, large=large), max_page_size=4096, force_codec=S.FREQ)
    check(gpu_ctx, sparse_bin(20_000, 0.05, 42, null_density=0.1, large=large), max_page_size=5000, force_codec=S.FREQ)
    check(gpu_ctx, sparse_bin(9_000, 0.0, 43, large=large), max_page_size=3000, force_codec=S.FREQ)      # no exceptions
    check(gpu_ctx, sparse_bin(9_000, 0.04, 44, top=b"", large=large), max_page_size=3000, force_codec=S.FREQ)  # empty top value
    check(gpu_ctx, sparse_bin(30_000, 0.5, 45, null_density=0.95, large=large), max_page_size=8192, force_codec=S.FREQ)  # top "is null"
    check(gpu_ctx, sparse_bin(140_000, 0.08, 46, large=large), max_page_size=70_000, force_codec=S.FREQ)  # bitmap + array containers
    check(gpu_ctx, sparse_bin(5_000, 0.07, 47, maxlen=900, exc_uniq=40, large=large), max_page_size=2500, force_codec=S.FREQ)  # long exceptions


@pytest.mark.parametrize("large", [False, True])
def test_binary_freq_encode_matches_oracle(gpu_ctx, large):
    from tests.test_gpu_encode import check as enc_check
    enc_check(gpu_ctx, sparse_bin(20_000, 0.05, 51, large=large), max_page_size=4096, force_codec=S.FREQ)
    enc_check(gpu_ctx, sparse_bin(20_000, 0.05, 52, null_density=0.1, large=large), max_page_size=5000, force_codec=S.FREQ)
    enc_check(gpu_ctx, sparse_bin(9_000, 0.0, 53, large=large), max_page_size=3000, force_codec=S.FREQ)
    enc_check(gpu_ctx, sparse_bin(9_000, 0.04, 54, top=b"", large=large), max_page_size=3000, force_codec=S.FREQ)
    enc_check(gpu_ctx, sparse_bin(30_000, 0.5, 55, null_density=0.95, large=large), max_page_size=8192, force_codec=S.FREQ)
    enc_check(gpu_ctx, sparse_bin(140_000, 0.08, 56, large=large), max_page_size=70_000, force_codec=S.FREQ)
    enc_check(gpu_ctx, sparse_bin(5_000, 0.07, 57, maxlen=900, exc_uniq=40, large=large), max_page_size=2500, force_codec=S.FREQ)


def test_binary_adaptive_picks_freq(gpu_ctx):
    """a Utf8 column that is mostly one string: choose_compressor takes Freq (binary/freq.rs:147-169)"""
    from tests.test_gpu_select import check as sel_check
    seen = set()
    for col in (sparse_bin(128 * 300, 0.03, 61), sparse_bin(128 * 300, 0.04, 62, null_density=0.05, top=b""),
                sparse_bin(128 * 300, 0.5, 63, null_density=0.93)):
        seen |= set(sel_check(gpu_ctx, col, max_page_size=128 * 100, ratio=2.0, forbidden=()).tolist())
    assert S.FREQ in seen, seen


@pytest.mark.parametrize("ptype", [S.T_I128, S.T_I256])
def test_wide_integer_freq(gpu_ctx, ptype):
    from tests.test_gpu_encode import check as enc_check
    from tests.test_gpu_select import check as sel_check
    for col, ps in ((sparse_wide(ptype, 20_000, 0.05, 71), 4096), (sparse_wide(ptype, 20_000, 0.05, 72, null_density=0.1), 5000),
                    (sparse_wide(ptype, 70_000, 0.08, 73), 70_000)):
        check(gpu_ctx, col, max_page_size=ps, force_codec=S.FREQ)
        enc_check(gpu_ctx, col, max_page_size=ps, force_codec=S.FREQ)
    seen = set(sel_check(gpu_ctx, sparse_wide(ptype, 128 * 300, 0.03, 74), max_page_size=128 * 100, ratio=2.0, forbidden=()).tolist())
    assert S.FREQ in seen, seen


def test_forced_freq_without_a_majority_value(gpu_ctx):
    """force_codec = Freq on pages where no value holds half of the rows (unreachable for choose_compressor): the
    top value is the exact arg-max of the counts over all slots, ties going to the earliest first occurrence"""
    from tests.test_gpu_encode import check as enc_check
    for ptype in (S.T_U8, S.T_I16, S.T_I32, S.T_I64, S.T_F32, S.T_F64, S.T_I128, S.T_I256):
        enc_check(gpu_ctx, gen.prim(ptype, 9000, uniq=7, null_density=0.2, seed=3), max_page_size=3000, force_codec=S.FREQ)
        enc_check(gpu_ctx, gen.prim(ptype, 5000, uniq=100 if ptype == S.T_U8 else 3000, seed=4), max_page_size=2500, force_codec=S.FREQ)
    # ties: every value exactly twice -> the value that occurs first wins
    v = np.tile(np.arange(500, dtype=np.int64), 2)[np.random.default_rng(5).permutation(1000)]
    enc_check(gpu_ctx, dict(ptype=S.T_I64, nullable=False, rows=1000, values=v, validity=None, offsets=None), force_codec=S.FREQ)
    # floats: -0.0 / +0.0 and NaNs with different payloads count as one value
    f = np.array([0.0, -0.0, 1.0, np.nan, 2.0, -0.0] * 200, np.float64)
    f.view(np.uint64)[3::12] = 0xFFF8000000000001
    enc_check(gpu_ctx, dict(ptype=S.T_F64, nullable=False, rows=f.size, values=f, validity=None, offsets=None), force_codec=S.FREQ)
    for large in (False, True):
        enc_check(gpu_ctx, gen.binary(8192, uniq=40, seed=5, large=large), max_page_size=4096, force_codec=S.FREQ)
        enc_check(gpu_ctx, gen.binary(6000, uniq=9, null_density=0.3, seed=6, large=large), max_page_size=2000, force_codec=S.FREQ)


def test_dict_pages_with_freq_indices(gpu_ctx):
    """integers that are mostly one value but whose maximum is below 256 may not use Freq themselves (freq.rs:146):
    choose_compressor takes Dict, and the u32 indices — mostly index 0, more than 256 distinct — become a nested
    Freq block.  The reference writes such pages; they decode on the device"""
    rng = np.random.default_rng(1)
    for ptype, npt in ((S.T_I64, np.int64), (S.T_I32, np.int32)):
        n = 3 * 8192
        v = np.full(n, -5, npt)
        exc = rng.random(n) < 0.06
        v[exc] = -rng.integers(10, 2000, int(exc.sum())).astype(npt)
        col = dict(ptype=ptype, nullable=False, rows=n, values=v, validity=None, offsets=None)
        seen = set()
        for opt in (dict(ratio=2.0), dict(ratio=2.0, default_compression=S.LZ4), dict(ratio=2.0, default_compression=S.ZSTD)):
            pages, metas = check(gpu_ctx, col, max_page_size=8192, forbidden=(S.RLE,), **opt)
            codecs, inner = S.stat_column(ptype, False, pages, metas)
            seen |= set(zip(codecs.tolist(), inner.tolist()))
        assert (S.DICT, S.FREQ) in seen, seen
    # one distinct exception index -> the exceptions block is OneValue: forced Dict over { -5 x many, 300 other values once
    # each at the start (ids 1..300), then one value repeated }
    v = np.concatenate([[-5], -np.arange(10, 310), np.full(8192 - 301, -5)]).astype(np.int64)
    v[400::97] = -309
    check(gpu_ctx, dict(ptype=S.T_I64, nullable=False, rows=v.size, values=v, validity=None, offsets=None), force_codec=S.DICT, ratio=1.5)
    # exactly 256 exceptions with ascending ids: the exceptions block is (Delta)Bitpacking
    v = np.full(8192, -5, np.int64)
    v[np.sort(np.random.default_rng(3).choice(np.arange(1, 8192), 256, replace=False))] = -(np.arange(256) + 10)
    for opt in (dict(ratio=2.0), dict(ratio=2.0, default_compression=S.LZ4)):
        pages, metas = check(gpu_ctx, dict(ptype=S.T_I64, nullable=False, rows=8192, values=v, validity=None, offsets=None),
                             forbidden=(S.RLE,), **opt)
        from strawboat_amd import stat
        q = stat.stat_page(pages, S.T_I64, False)
        assert (q.codec, q.body.indices.codec, q.body.indices.body.exceptions.codec) == (S.DICT, S.FREQ, S.DELTABP)
    # binary: forced Dict, nested selection on the indices
    b = sparse_bin(20_000, 0.05, 91, exc_uniq=600)
    pages, metas = check(gpu_ctx, b, max_page_size=20000, force_codec=S.DICT, ratio=2.0, forbidden=(S.RLE,))
    assert S.FREQ in set(S.stat_column(b["ptype"], b["nullable"], pages, metas)[1].tolist())


def test_dict_pages_with_freq_indices_encode(gpu_ctx):
    """the same pages written on the device: Dict block = Freq block over the u32 indices (top index, Roaring bitmap,
    nested exceptions block chosen on the device) + dictionary, byte-equal to the oracle"""
    from tests.test_gpu_select import check as sel_check
    rng = np.random.default_rng(1)
    for ptype, npt in ((S.T_I64, np.int64), (S.T_I32, np.int32)):
        n = 3 * 8192
        v = np.full(n, -5, npt)
        exc = rng.random(n) < 0.06
        v[exc] = -rng.integers(10, 2000, int(exc.sum())).astype(npt)
        for validity in (None, gen.make_validity(rng, n, 0.02)):
            col = dict(ptype=ptype, nullable=validity is not None, rows=n, values=v, validity=validity, offsets=None)
            seen = set()
            for opt in (dict(ratio=2.0), dict(ratio=2.0, default_compression=S.LZ4), dict(ratio=1.2)):
                codecs = sel_check(gpu_ctx, col, max_page_size=8192, forbidden=(S.RLE,), **opt)
                seen |= set(codecs.tolist())
            assert S.DICT in seen
    # a leading null (the dictionary's first entry is T::default()) and f64 values
    g = np.where(rng.random(8192) < 0.07, -rng.integers(10, 3000, 8192).astype(np.float64), -5.0)
    valid = np.ones(8192, bool)
    valid[0] = False
    sel_check(gpu_ctx, dict(ptype=S.T_F64, nullable=True, rows=8192, values=g, validity=gen.pack_bits(valid), offsets=None),
              ratio=2.0, forbidden=(S.RLE,))


def test_binary_and_forced_dict_pages_with_freq_indices_encode(gpu_ctx):
    """Dict pages whose u32 indices are a Freq block, written on the device for Binary / Utf8 columns (entries
    `u64 len | bytes`, binary/dict.rs:55-93) and for a *forced* Dict codec (nested selection, or Freq forced on the
    indices): byte-equal to the oracle"""
    from tests.test_gpu_encode import check as enc_check
    for large in (False, True):
        for nd in (None, 0.03):
            b = sparse_bin(20_000, 0.05, 91, exc_uniq=600, large=large, null_density=nd)
            for opt in (dict(force_codec=S.DICT, ratio=2.0, forbidden=(S.RLE,)), dict(force_codec=S.DICT, force_index_codec=S.FREQ),
                        dict(force_codec=S.DICT, force_index_codec=S.FREQ, default_compression=S.LZ4)):
                enc_check(gpu_ctx, b, max_page_size=8192, **opt)
                pages, metas = gen.oracle_write(b, max_page_size=8192, **opt)
                assert S.FREQ in set(S.stat_column(b["ptype"], b["nullable"], pages, metas)[1].tolist()), opt
    # empty strings as the top value and as exceptions, one page shorter than the others
    e = sparse_bin(9_000, 0.06, 92, exc_uniq=700, top=b"", maxlen=9)
    enc_check(gpu_ctx, e, max_page_size=4096, force_codec=S.DICT, force_index_codec=S.FREQ)
    # primitives of every width under a forced Dict
    rng = np.random.default_rng(7)
    for ptype, npt in ((S.T_I64, np.int64), (S.T_I32, np.int32), (S.T_I16, np.int16), (S.T_F64, np.float64)):
        n = 2 * 8192
        v = np.full(n, -5, npt)
        exc = rng.random(n) < 0.06
        v[exc] = (-rng.integers(10, 2000, int(exc.sum()))).astype(npt)
        col = dict(ptype=ptype, nullable=False, rows=n, values=v, validity=None, offsets=None)
        for opt in (dict(force_codec=S.DICT, force_index_codec=S.FREQ), dict(force_codec=S.DICT, ratio=2.0, forbidden=(S.RLE,))):
            enc_check(gpu_ctx, col, max_page_size=8192, **opt)
            pages, metas = gen.oracle_write(col, max_page_size=8192, **opt)
            if "force_index_codec" in opt:
                assert set(S.stat_column(ptype, False, pages, metas)[1].tolist()) == {S.FREQ}


def test_freq_pages_of_more_than_four_mi_rows(gpu_ctx):
    """without max_page_size a column is ONE page (write/common.rs: page size = rows): 4.3 M rows are 66 Roaring
    containers, bitmap and array containers mixed"""
    from tests.test_gpu_encode import check as enc_check
    from tests.test_gpu_select import check as sel_check
    c = sparse(S.T_I64, 4_300_000, 0.03, 5)
    c["values"][70_000:140_000:3] = 999    # one dense (bitmap) container
    c["values"][200_000:262_144] = 7       # one empty container
    enc_check(gpu_ctx, c, force_codec=S.FREQ)
    check(gpu_ctx, c, force_codec=S.FREQ)
    assert set(sel_check(gpu_ctx, c, ratio=2.0, forbidden=()).tolist()) == {S.FREQ}
    n = sparse(S.T_I32, 4_200_000, 0.02, 6, null_density=0.05)
    enc_check(gpu_ctx, n, force_codec=S.FREQ)
    check(gpu_ctx, n, force_codec=S.FREQ)
    b = sparse_bin(4_300_000, 0.03, 6)
    enc_check(gpu_ctx, b, force_codec=S.FREQ)
    check(gpu_ctx, b, force_codec=S.FREQ)
