"""The C-ABI library loads without a GPU and exports every symbol include/strawboat_hip.h declares;
host-only entry points behave (no compute calls here)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    from strawboat_amd import _native
    if not os.path.exists(_native.LIB_PATH):
        g.build()
    return _native.load()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "strawboat_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sb_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_are_exported(lib):
    from strawboat_amd import _native
    names = declared_symbols()
    assert set(names) == set(_native.EXPORTS), "keep _native.EXPORTS in sync with the header"
    for n in names:
        assert hasattr(lib, n), "libstrawboat_hip.so does not export %s" % n


def test_version_and_no_gpu_behaviour(lib):
    import torch
    assert b"gfx950" in lib.sb_version()
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = C.c_void_p()
    rc = lib.sb_ctx_create(0, None, C.byref(h))
    assert rc != 0 and not h.value        # fails loudly, no CPU fallback
    import strawboat_amd as sb
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        sb.Context(0)


def test_write_bound_page_arithmetic(lib):
    # page arithmetic of encode_chunk (src/write/common.rs:54-58,79-86)
    from strawboat_amd import _native as N
    from strawboat_amd.types import WriteOptions
    from strawboat_amd.write import options_c
    npages = C.c_uint64()
    o = options_c(WriteOptions(max_page_size=65536))
    b = lib.sb_write_bound(12, 1, 1_000_000, 0, C.byref(o), C.byref(npages))
    assert npages.value == 16 and b > 1_000_000 * 8
    o = options_c(WriteOptions())
    lib.sb_write_bound(4, 0, 1_000_000, 0, C.byref(o), C.byref(npages))
    assert npages.value == 1
    o = options_c(WriteOptions(max_page_size=2048))
    lib.sb_write_bound(4, 0, 10_000, 0, C.byref(o), C.byref(npages))
    assert npages.value == 5
    assert lib.sb_write_bound(4, 0, 0, 0, C.byref(o), C.byref(npages)) == 0 and npages.value == 0


def test_struct_layouts_match_header():
    from strawboat_amd import _native as N
    assert C.sizeof(N.PageMetaC) == 16
    assert C.sizeof(N.WriteOptionsC) == 48
    assert C.sizeof(N.ColumnReadC) == 112
    assert C.sizeof(N.ColumnWriteC) == 160
