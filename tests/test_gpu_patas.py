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
