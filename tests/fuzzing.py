"""Corrupted-page cases shared by tests/probes/fuzz_decode.py and tests/test_gpu_robustness.py: oracle-written
pages of every codec / column family, and the mutations applied to them (byte flips, size fields,
truncation).  The reference panics or errors on such input; the device must report a
status — never fault, never hang."""
import numpy as np

from oracle import sbo as S
from tests import gen

CASES = []
for pt in (S.T_I32, S.T_I64, S.T_F64, S.T_U8, S.T_I128):
    for codec in (S.NONE, S.RLE, S.DICT, S.ONEVALUE, S.LZ4, S.ZSTD, S.SNAPPY, S.FREQ):
        CASES.append(("prim %d codec %d" % (pt, codec), lambda pt=pt, codec=codec: (
            gen.prim(pt, 9000, uniq=1 if codec == S.ONEVALUE else 60, null_density=0.1, runs=6 if codec != S.FREQ else 200),
            dict(max_page_size=3000, **({"default_compression": codec} if codec <= 3 else {"force_codec": codec})))))
for codec in (S.BITPACK, S.DELTABP):
    CASES.append(("i32 codec %d" % codec, lambda codec=codec: (gen.prim(S.T_I32, 128 * 60, uniq=5000, sorted_=codec == S.DELTABP),
                                                                dict(max_page_size=128 * 20, force_codec=codec))))
for ic in (S.RLE, S.BITPACK, S.LZ4, S.ONEVALUE):
    CASES.append(("i64 dict idx %d" % ic, lambda ic=ic: (gen.prim(S.T_I64, 128 * 60, uniq=1 if ic == S.ONEVALUE else 90, runs=5),
                                                          dict(max_page_size=128 * 20, force_codec=S.DICT, force_index_codec=ic))))
CASES.append(("f64 patas", lambda: (gen.prim(S.T_F64, 6000, uniq=50, runs=4), dict(max_page_size=2000, force_codec=S.PATAS))))
for large in (False, True):
    for codec in (S.NONE, S.LZ4, S.ZSTD, S.DICT, S.ONEVALUE, S.FREQ):
        CASES.append(("binary large=%d codec %d" % (large, codec), lambda large=large, codec=codec: (
            gen.binary(6000, uniq=1 if codec == S.ONEVALUE else 40, null_density=0.1, large=large, zipf=2.5 if codec == S.FREQ else None),
            dict(max_page_size=2000, **({"default_compression": codec} if codec <= 3 else {"force_codec": codec})))))
for codec in (S.NONE, S.LZ4, S.RLE, S.ONEVALUE):
    CASES.append(("bool codec %d" % codec, lambda codec=codec: (
        gen.boolean(9000, null_density=0.1, runs=7, p_true=1.0 if codec == S.ONEVALUE else 0.5),
        dict(max_page_size=3000, **({"default_compression": codec} if codec <= 3 else {"force_codec": codec})))))



def _dominant(npt, ptype, seed, ascending=False):
    rng = np.random.default_rng(seed)
    v = np.full(8192, -5, npt)
    if ascending:   # exactly 256 exceptions with ascending ids: (Delta)Bitpacking exceptions block
        v[np.sort(rng.choice(np.arange(1, 8192), 256, replace=False))] = -(np.arange(256) + 10)
    else:
        exc = rng.random(8192) < 0.06
        v[exc] = -rng.integers(10, 2000, int(exc.sum())).astype(npt)
    return dict(ptype=ptype, nullable=False, rows=8192, values=v, validity=None, offsets=None)


# Dict pages whose u32 indices are a nested Freq block (plain / LZ4 / bit-packed exceptions)
for dc in (S.NONE, S.LZ4):
    CASES.append(("i64 dict(freq idx) default %d" % dc, lambda dc=dc: (_dominant(np.int64, S.T_I64, 5),
                                                                        dict(ratio=2.0, forbidden=(S.RLE,), default_compression=dc))))
CASES.append(("i64 dict(freq idx, dbp exceptions)", lambda: (_dominant(np.int64, S.T_I64, 6, ascending=True), dict(ratio=2.0, forbidden=(S.RLE,)))))


def mutate(rng, pages, metas, t):
    """the t-th mutation of (pages, metas): returns (pages', metas')"""
    pg = pages.copy()
    m = metas.copy()
    starts = np.concatenate([[0], np.cumsum(metas[:, 0])[:-1]]).astype(np.int64)
    kind = t % 4
    if kind == 0:    # flips anywhere
        for _ in range(int(rng.integers(1, 6))):
            pg[int(rng.integers(0, pg.size))] = int(rng.integers(0, 256))
    elif kind == 1:  # flips in the first bytes of a page (headers, sizes)
        p = int(rng.integers(0, len(starts)))
        for _ in range(int(rng.integers(1, 4))):
            pg[int(starts[p] + rng.integers(0, min(48, metas[p, 0])))] = int(rng.integers(0, 256))
    elif kind == 2:  # a size field becomes huge / tiny
        p = int(rng.integers(0, len(starts)))
        o = int(starts[p] + rng.integers(0, min(40, metas[p, 0] - 4)))
        v = int(rng.choice([0, 1, 0xFFFFFFFF, 0x7FFFFFFF, 65536, int(rng.integers(0, 1 << 20))]))
        pg[o:o + 4] = np.frombuffer(np.uint32(v).tobytes(), np.uint8)
    else:            # the last page is cut short
        p = len(starts) - 1
        cut = int(rng.integers(0, metas[p, 0]))
        m[p, 0] = cut
        pg = pg[:int(starts[p]) + cut].copy()
    return pg, m
