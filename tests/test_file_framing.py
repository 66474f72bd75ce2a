"""File framing through the C ABI (host only — runs without a GPU): the bytes sb_file_writer_* emit
equal the oracle's restatement of NativeWriter::start/finish, the reader returns what was written,
and the writer's state machine raises the reference's errors (src/write/writer.rs:91-135)."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import sbo as S
from oracle import sbo_file as F
from tests import gen


@pytest.fixture(scope="module")
def lib():
    from strawboat_amd import _native as N
    return N.load()


def metas_c(metas):
    from strawboat_amd import _native as N
    arr = (N.PageMetaC * max(len(metas), 1))()
    for i, (l, v) in enumerate(metas):
        arr[i].length, arr[i].num_values = int(l), int(v)
    return arr


def make_columns():
    cols = []
    for k, col in enumerate([gen.prim(S.T_I64, 5000, null_density=0.1, seed=1), gen.prim(S.T_F64, 5000, runs=8, seed=2),
                             gen.binary(5000, uniq=50, null_density=0.2, seed=3), gen.boolean(5000, seed=4)]):
        pages, metas = gen.oracle_write(col, max_page_size=1024, force_codec=[S.NONE, S.RLE, S.DICT, S.NONE][k])
        cols.append((pages, [tuple(int(x) for x in m) for m in metas]))
    return cols


def write_with_abi(lib, path, cols, schema):
    h = C.c_void_p()
    assert lib.sb_file_writer_open(path.encode(), C.byref(h)) == 0
    assert lib.sb_file_writer_start(h) == 0
    for pages, metas in cols:
        pg = np.ascontiguousarray(pages, dtype=np.uint8)
        assert lib.sb_file_writer_write_column(h, pg.ctypes.data_as(C.c_void_p), pg.size, metas_c(metas), len(metas)) == 0
    total = C.c_uint64(0)
    sb = np.frombuffer(schema, dtype=np.uint8)
    assert lib.sb_file_writer_finish(h, sb.ctypes.data_as(C.c_void_p), sb.size, C.byref(total)) == 0
    lib.sb_file_writer_close(h)
    return int(total.value)


def test_file_bytes_equal_oracle_and_read_back(lib, tmp_path):
    cols = make_columns()
    schema = bytes(range(200)) * 3  # opaque to the framing
    path = str(tmp_path / "t.sb")
    total = write_with_abi(lib, path, cols, schema)
    got = open(path, "rb").read()
    want = F.write_file(cols, schema)
    assert total == len(want) and got == want
    assert got[:8] == b"ARROW2\x00\x00" and got[-8:] == b"\xff\xff\xff\xff\x00\x00\x00\x00"
    # reader
    r = C.c_void_p()
    assert lib.sb_file_reader_open(path.encode(), C.byref(r)) == 0
    assert lib.sb_file_reader_n_columns(r) == len(cols)
    om = F.read_meta(got)
    from strawboat_amd import _native as N
    for i, (pages, metas) in enumerate(cols):
        off, npg = C.c_uint64(), C.c_uint64()
        pm = C.POINTER(N.PageMetaC)()
        assert lib.sb_file_reader_column(r, i, C.byref(off), C.byref(npg), C.byref(pm)) == 0
        assert (int(off.value), [(int(pm[k].length), int(pm[k].num_values)) for k in range(npg.value)]) == (om[i][0], list(metas))
        assert om[i][1] == list(metas)
        # a page range in the middle: ColumnMeta::slice arithmetic
        first, n = 1, len(metas) - 2
        want_off = sum(m[0] for m in metas[:first])
        want_len = sum(m[0] for m in metas[first:first + n])
        dst = np.zeros(want_len, np.uint8)
        nread = C.c_uint64()
        assert lib.sb_file_reader_read_pages(r, i, first, n, dst.ctypes.data_as(C.c_void_p), dst.size, C.byref(nread)) == 0
        assert nread.value == want_len and bytes(dst) == bytes(np.asarray(pages)[want_off:want_off + want_len])
    sp, sl = C.c_void_p(), C.c_uint64()
    assert lib.sb_file_reader_schema(r, C.byref(sp), C.byref(sl)) == 0
    assert C.string_at(sp, sl.value) == schema == F.schema_bytes(got)
    lib.sb_file_reader_close(r)


def test_writer_state_machine_errors(lib, tmp_path):
    h = C.c_void_p()
    assert lib.sb_file_writer_open(str(tmp_path / "e.sb").encode(), C.byref(h)) == 0
    m = metas_c([(0, 0)])
    assert lib.sb_file_writer_write_column(h, None, 0, m, 1) == -1   # OutOfSpec: not started
    assert b"must be started before" in lib.sb_file_last_error()
    assert lib.sb_file_writer_finish(h, None, 0, None) == -1
    assert b"must be written before it can be finished" in lib.sb_file_last_error()
    assert lib.sb_file_writer_start(h) == 0
    assert lib.sb_file_writer_start(h) == -1
    assert b"can only be started once" in lib.sb_file_last_error()
    assert lib.sb_file_writer_write_column(h, None, 0, m, 1) == 0
    assert lib.sb_file_writer_finish(h, None, 0, None) == 0
    assert lib.sb_file_writer_write_column(h, None, 0, m, 1) == -1
    assert b"only accept one RowGroup" in lib.sb_file_last_error()
    lib.sb_file_writer_close(h)


def test_reader_rejects_truncated_files(lib, tmp_path):
    p = tmp_path / "bad.sb"
    p.write_bytes(b"ARROW2\x00\x00" + b"\x00" * 4)
    r = C.c_void_p()
    assert lib.sb_file_reader_open(str(p).encode(), C.byref(r)) == -3
    good = F.write_file([(b"abc", [(3, 1)])], b"schema")
    p.write_bytes(good[:-20] + b"\xff" * 4 + good[-16:])   # meta_size pointing outside the file
    bad = bytearray(good)
    bad[-12:-8] = (10 ** 6).to_bytes(4, "little")
    p.write_bytes(bytes(bad))
    assert lib.sb_file_reader_open(str(p).encode(), C.byref(r)) == -3
