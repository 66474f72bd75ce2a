"""The device against the layout vectors of tests/golden/layout_vectors.json (RoaringBitmap containers inside Freq pages —
bitmap container, two containers, a full container — and BitPacker4x blocks at 31 / 32 bits and with wrapped deltas):
the pages the device writes are the oracle's bytes (which tests/test_oracle_layouts.py pins), and it reads them back."""
import numpy as np
import pytest

from oracle import sbo as S
from tests.test_gpu_decode import gpu_decode
from tests.test_gpu_encode import check as encode_check
from tests.test_oracle_layouts import VEC, bitpack_column, freq_column

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", VEC["freq"], ids=[c["name"] for c in VEC["freq"]])
def test_freq_pages_with_every_roaring_container_kind(gpu_ctx, case):
    col = freq_column(case)
    encode_check(gpu_ctx, col, force_codec=S.FREQ)
    page, metas = S.write_column(S.T_U32, False, col["rows"], col["values"], options=S.make_options(force_codec=S.FREQ))
    got = gpu_decode(gpu_ctx, col, page, metas)
    assert np.array_equal(got.values_numpy(), col["values"].view(np.uint8))


@pytest.mark.parametrize("case", VEC["bitpack"], ids=[c["name"] for c in VEC["bitpack"]])
def test_bitpacker4x_blocks(gpu_ctx, case):
    col = bitpack_column(case)
    codec = S.DELTABP if case["delta"] else S.BITPACK
    encode_check(gpu_ctx, col, force_codec=codec)
    page, metas = S.write_column(S.T_U32, False, col["rows"], col["values"], options=S.make_options(force_codec=codec))
    want = S.read_column(S.T_U32, False, page, metas)["values"]     # (for wrapped deltas: what upstream's decoder makes of them)
    got = gpu_decode(gpu_ctx, col, page, metas)
    assert np.array_equal(got.values_numpy(), want)
