"""GPU parity for Patas pages (codec id 16, f64; src/compression/double/patas.rs:106-133): pages
written by the oracle, decoded on the device, compared with the oracle's decode."""
import numpy as np
import pytest

from oracle import sbo as S
from tests import gen
from tests.test_gpu_decode import check, gpu_decode

pytestmark = pytest.mark.gpu


def fcol(values, null_density=None, seed=3):
    rng = np.random.default_rng(seed)
    v = np.asarray(values, np.float64)
    validity = gen.make_validity(rng, v.size, null_density)
    return dict(ptype=S.T_F64, nullable=validity is not None, rows=v.size, values=v, validity=validity, offsets=None)


def test_patas_pages(gpu_ctx):
    rng = np.random.default_rng(1)
    shapes = [
        rng.integers(0, 50, 20_000).astype(np.float64),                       # references to repeated values
        np.repeat(rng.random(400), 50),                                        # long equal runs (tz = 63 records)
        rng.random(20_000),                                                    # no repeats: ref = i - 1, 8 significant bytes
        np.cumsum(rng.integers(0, 3, 20_000)).astype(np.float64) * 0.25,      # few significant bytes
        np.array([0.0, -0.0, np.nan, np.inf, -np.inf, 1.0] * 3000),           # special bit patterns
        np.tile(np.arange(200, dtype=np.float64), 100),                        # period 200 > 128: refs fall back to i - 1
    ]
    for v in shapes:
        check(gpu_ctx, fcol(v), max_page_size=4096, force_codec=S.PATAS)
        check(gpu_ctx, fcol(v, null_density=0.2), max_page_size=5000, force_codec=S.PATAS)
    check(gpu_ctx, fcol(shapes[0][:1]), force_codec=S.PATAS)                   # a page of one value
    check(gpu_ctx, fcol(rng.integers(0, 9, 200_000).astype(np.float64)), max_page_size=65536, force_codec=S.PATAS)


def test_patas_truncated_page_raises(gpu_ctx):
    from strawboat_amd._native import NativeError
    col = fcol(np.random.default_rng(2).random(1000))
    pages, metas = gen.oracle_write(col, force_codec=S.PATAS)
    cut = pages[:-5].copy()
    cut[1:5] = np.frombuffer(int(cut.size - 9).to_bytes(4, "little"), np.uint8)
    m = metas.copy()
    m[0, 0] = cut.size
    with pytest.raises(NativeError) as e:
        gpu_decode(gpu_ctx, col, cut, m)
    assert e.value.code == -1


# ---- encode: the device writes the bytes the oracle writes (double/patas.rs:36-104)
def test_patas_encode_matches_oracle(gpu_ctx):
    from tests.test_gpu_encode import check as enc_check
    rng = np.random.default_rng(11)
    shapes = [
        rng.integers(0, 50, 20_000).astype(np.float64),
        np.repeat(rng.random(400), 50),
        rng.random(20_000),
        np.cumsum(rng.integers(0, 3, 20_000)).astype(np.float64) * 0.25,
        np.array([0.0, -0.0, np.nan, np.inf, -np.inf, 1.0] * 3000),
        np.tile(np.arange(200, dtype=np.float64), 100),      # period > 128: references fall back to the previous row
        np.arange(300, dtype=np.float64),                      # every value unseen: row 0 while i < 128
    ]
    for v in shapes:
        enc_check(gpu_ctx, fcol(v), max_page_size=4096, force_codec=S.PATAS)
        enc_check(gpu_ctx, fcol(v, null_density=0.2), max_page_size=5000, force_codec=S.PATAS)
        f32 = dict(fcol(v), ptype=S.T_F32, values=v.astype(np.float32))
        enc_check(gpu_ctx, f32, max_page_size=4100, force_codec=S.PATAS)
    enc_check(gpu_ctx, fcol(shapes[0][:1]), force_codec=S.PATAS)
    enc_check(gpu_ctx, fcol(rng.integers(0, 9, 200_000).astype(np.float64)), max_page_size=65536, force_codec=S.PATAS)


def test_patas_for_integers_is_out_of_spec(gpu_ctx):
    from strawboat_amd._native import NativeError
    from tests.test_gpu_encode import gpu_encode
    with pytest.raises(NativeError) as e:
        gpu_encode(gpu_ctx, gen.prim(S.T_I64, 1000, uniq=5), force_codec=S.PATAS)
    assert e.value.code in (-1, -4)


def test_adaptive_selection_with_patas_allowed(gpu_ctx):
    """float pages with Freq alone forbidden: the selector may pick Patas, and the pages still equal the oracle's"""
    from tests.test_gpu_select import check as sel_check
    rng = np.random.default_rng(12)
    seen = set()
    for v in (rng.random(128 * 200), np.cumsum(rng.integers(0, 3, 128 * 200)).astype(np.float64) * 0.25,
              np.repeat(rng.random(800), 32), rng.integers(0, 30, 128 * 200).astype(np.float64)):
        for ratio in (1.05, 1.5):
            seen |= set(sel_check(gpu_ctx, fcol(v), max_page_size=128 * 50, ratio=ratio, forbidden=(S.FREQ,)).tolist())
            sel_check(gpu_ctx, dict(fcol(v), ptype=S.T_F32, values=v.astype(np.float32)), max_page_size=128 * 50, ratio=ratio,
                      forbidden=(S.FREQ,))
    assert S.PATAS in seen, "expected Patas to win on at least one shape, got %s" % seen
