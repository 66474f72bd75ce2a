"""Pins the oracle's two third-party layouts — RoaringBitmap portable serialisation inside Freq pages
(src/compression/integer/freq.rs:58-76) and BitPacker4x blocks (integer/bp.rs:45-61, delta_bp.rs:45-65) — against
tests/golden/layout_vectors.json, which tests/golden/make_layout_vectors.py derives from the published format rules with
plain Python integers (no code shared with oracle/).  No second implementation of either crate is importable in this image:
these vectors and SURVEY.md Appendix C (M, N, O, P, Q) are the pin."""
import hashlib
import importlib.util
import json
import os

import numpy as np
import pytest

from oracle import sbo as S

HERE = os.path.dirname(os.path.abspath(__file__))
VEC = json.load(open(os.path.join(HERE, "golden", "layout_vectors.json")))
_spec = importlib.util.spec_from_file_location("make_layout_vectors", os.path.join(HERE, "golden", "make_layout_vectors.py"))
MK = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(MK)


def freq_column(case):
    _, ex, vals = MK.freq_case(case["name"], case["rows"], case["top"], MK.exception_rows(case["exceptions"]), case["seed"])
    v = np.full(case["rows"], case["top"], np.uint32)
    v[ex] = vals.astype(np.uint32)
    return dict(ptype=S.T_U32, nullable=False, rows=v.size, values=v, validity=None, offsets=None)


def bitpack_column(case):
    vals, delta = MK.bp_case(case["name"])
    v = np.array(vals, np.uint32)
    return dict(ptype=S.T_U32, nullable=False, rows=v.size, values=v, validity=None, offsets=None)


def check_digest(got, want):
    assert len(got) == want["length"]
    assert bytes(got[:48]).hex() == want["head_hex"]
    assert hashlib.sha256(bytes(got)).hexdigest() == want["sha256"]


@pytest.mark.parametrize("case", VEC["freq"], ids=[c["name"] for c in VEC["freq"]])
def test_roaring_bytes_inside_freq_pages(case):
    col = freq_column(case)
    page, metas = S.write_column(S.T_U32, False, col["rows"], col["values"], options=S.make_options(force_codec=S.FREQ))
    assert len(metas) == 1 and page[0] == S.FREQ
    top = int.from_bytes(bytes(page[9:13]), "little")
    blen = int.from_bytes(bytes(page[13:17]), "little")
    assert top == case["top"]
    check_digest(page[17:17 + blen], case["roaring"])
    back = S.read_column(S.T_U32, False, page, metas)
    assert np.array_equal(back["values"], col["values"].view(np.uint8))


@pytest.mark.parametrize("case", VEC["bitpack"], ids=[c["name"] for c in VEC["bitpack"]])
def test_bitpacker4x_blocks(case):
    col = bitpack_column(case)
    codec = S.DELTABP if case["delta"] else S.BITPACK
    page, metas = S.write_column(S.T_U32, False, col["rows"], col["values"], options=S.make_options(force_codec=codec))
    assert page[0] == codec
    body = page[9:]
    assert [int(body[0]), int(body[1 + 16 * int(body[0])])] == case["num_bits"]
    check_digest(body, case["blocks"])
    if case["name"] != "delta_wrap":   # (a wrapped delta does not survive num_bits taken from the raw values: upstream's hazard, App. B)
        back = S.read_column(S.T_U32, False, page, metas)
        assert np.array_equal(back["values"], col["values"].view(np.uint8))
