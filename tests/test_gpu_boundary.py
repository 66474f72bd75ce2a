"""Boundary behaviour of the C ABI that the parity tests do not reach:
  * more enqueued calls than the staging ring has slots before ONE synchronize (results of the early
    calls must not be overwritten by the later ones);
  * a binary column decoded into a values buffer that is too small: an error, values_len still
    reported, and nothing written past values_capacity;
  * a failed synchronize interval that held Freq pages does not leak its records into the next one.
"""
import ctypes as C

import numpy as np
import pytest

from oracle import sbo as S
from tests import gen

pytestmark = pytest.mark.gpu


def _dev(ctx, a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(ctx.torch_device)


def _dcol(ctx, col):
    from strawboat_amd import write
    return write.DeviceColumn(col["ptype"], col["nullable"], col["rows"], _dev(ctx, col["values"]),
                              None if col["validity"] is None else _dev(ctx, col["validity"]),
                              None if col["offsets"] is None else _dev(ctx, col["offsets"]))


def test_many_calls_before_one_synchronize(gpu_ctx):
    """20 distinct encode calls and 20 distinct decode calls, one synchronize each: every call gets its OWN
    metas / lengths back (the pinned staging ring has 8 slots)."""
    from strawboat_amd import read, write
    from strawboat_amd.types import WriteOptions
    cols, encs, wants = [], [], []
    for k in range(20):
        rows = 3000 + 517 * k
        col = gen.prim(S.T_I32 if k % 2 else S.T_I64, rows, uniq=5 + 40 * k, null_density=0.1 if k % 3 else None,
                       seed=100 + k, runs=8 if k % 4 == 0 else None)
        cols.append(col)
        ps = 500 + 37 * k
        wants.append(gen.oracle_write(col, max_page_size=ps, ratio=2.0))
        encs.append(write.encode_columns(gpu_ctx, [_dcol(gpu_ctx, col)], WriteOptions(max_page_size=ps, default_compress_ratio=2.0))[0])
    gpu_ctx.synchronize()
    for k in range(20):
        assert np.array_equal(encs[k].metas_array(), wants[k][1]), "call %d: metas of another call" % k
        assert np.array_equal(encs[k].pages_numpy(), wants[k][0]), "call %d" % k
    outs = []
    bins = [gen.binary(2000 + 301 * k, uniq=20 + 11 * k, null_density=0.2, seed=k) for k in range(20)]
    pages = [gen.oracle_write(b, max_page_size=700 + k, ratio=2.0) for k, b in enumerate(bins)]
    sizes = []
    for k in range(20):
        want = gen.oracle_read(bins[k], *pages[k])
        sizes.append(want["values"].size)
        cp = read.ColumnPages(bins[k]["ptype"], True, _dev(gpu_ctx, pages[k][0]), pages[k][1])
        outs.append((read.batch_read_columns(gpu_ctx, [cp], values_capacity=[want["values"].size + 64])[0], want))
    gpu_ctx.synchronize()
    for k, (got, want) in enumerate(outs):
        assert got.values_len == sizes[k], "call %d: values_len of another call" % k
        assert np.array_equal(got.values_numpy(), want["values"]), k
        assert np.array_equal(got.offsets_numpy().view(np.int32), want["offsets"].view(np.int32)), k


def test_staging_slots_outgrown_while_results_are_pending(gpu_ctx):
    """calls whose page tables grow 3 x from one to the next, none synchronized: every one of the eight pinned staging slots
    is outgrown (twice) while the results of the calls before still sit in it — they are moved out first, the old buffers are
    freed at the synchronize, every call gets its own metas and pages"""
    from strawboat_amd import write
    from strawboat_amd.types import WriteOptions
    encs, wants = [], []
    npages = 2
    for k in range(18):
        rows = npages * 64
        col = gen.prim(S.T_I32, rows, uniq=7 + k, null_density=0.1 if k % 2 else None, seed=300 + k, runs=5 if k % 3 == 0 else None)
        wants.append(gen.oracle_write(col, max_page_size=64, ratio=2.0))
        encs.append(write.encode_columns(gpu_ctx, [_dcol(gpu_ctx, col)], WriteOptions(max_page_size=64, default_compress_ratio=2.0))[0])
        npages = npages * 3 if k % 2 else npages + 5      # (2 ... ~20 000 pages)
    gpu_ctx.synchronize()
    for k in range(18):
        assert np.array_equal(encs[k].metas_array(), wants[k][1]), "call %d: metas of another call" % k
        assert np.array_equal(encs[k].pages_numpy(), wants[k][0]), "call %d" % k


@pytest.mark.parametrize("codec", [S.ONEVALUE, S.DICT, S.NONE, S.LZ4, S.FREQ])
def test_undersized_values_capacity_is_not_overrun(gpu_ctx, codec):
    """binary pages that expand beyond the caller's values buffer (a OneValue / Dict page expands far
    beyond 4x its page bytes): SB_ERR_INVALID, the real size in values_len, the guard bytes untouched"""
    import torch
    from strawboat_amd import _native as N
    from strawboat_amd import read
    rows = 20000
    if codec == S.ONEVALUE:
        col = gen.binary(rows, uniq=1, seed=3, minlen=40, maxlen=40)
    elif codec == S.FREQ:
        col = gen.binary(rows, uniq=30, seed=3, minlen=30, maxlen=40)
        # mostly one value
        idx = np.zeros(rows, np.int64)
        idx[::50] = 1
        o = col["offsets"].astype(np.int64)
        lens = (o[1:] - o[:-1])
        first = col["values"][o[0]:o[1]].copy()
        second = col["values"][o[1]:o[2]].copy()
        parts = [second if i else first for i in idx]
        col["values"] = np.concatenate(parts)
        col["offsets"] = np.concatenate([[0], np.cumsum([p.size for p in parts])]).astype(np.int32)
        del lens
    else:
        col = gen.binary(rows, uniq=16, seed=3, minlen=30, maxlen=40)
    pages, metas = gen.oracle_write(col, max_page_size=4096, force_codec=codec)
    want = gen.oracle_read(col, pages, metas)
    need = want["values"].size
    cap = need // 3
    buf = torch.full((need + 4096,), 0xAB, dtype=torch.uint8, device=gpu_ctx.torch_device)
    cp = read.ColumnPages(col["ptype"], col["nullable"], _dev(gpu_ctx, pages), metas)
    batch = read.ReadBatch(gpu_ctx, [cp], values_capacity=[need])
    c = batch._arr[0]
    c.values = C.c_void_p(buf.data_ptr())
    c.values_capacity = cap
    batch.enqueue()
    with pytest.raises(N.NativeError) as ei:
        gpu_ctx.synchronize()
    assert ei.value.code == N.SB_ERR_INVALID
    assert int(c.values_len) == need          # the caller can retry with the right size
    tail = buf[cap:].cpu().numpy()
    assert (tail == 0xAB).all(), "%d bytes written past values_capacity" % int((tail != 0xAB).sum())
    # retry with the reported size
    c.values_capacity = need
    batch.enqueue()
    gpu_ctx.synchronize()
    assert np.array_equal(buf[:need].cpu().numpy(), want["values"])


def test_freq_records_of_a_failed_interval_are_dropped(gpu_ctx):
    """interval 1: a Freq column + a corrupt column -> error at synchronize.  interval 2: another Freq column
    decodes correctly (the first interval's FreqEntry records, which point into freed buffers, are gone)."""
    import torch
    from strawboat_amd import _native as N
    from strawboat_amd import read
    def freq_col(seed, rows):
        rng = np.random.default_rng(seed)
        v = np.full(rows, 1000 + seed, np.int64)
        k = rng.choice(rows, rows // 50, replace=False)
        v[k] = rng.integers(300, 100000, k.size)
        return dict(ptype=S.T_I64, nullable=False, rows=rows, values=v, validity=None, offsets=None)
    a = freq_col(1, 30000)
    pa, ma = gen.oracle_write(a, max_page_size=4096, force_codec=S.FREQ)
    bad = gen.prim(S.T_I32, 5000, seed=9)
    pb, mb = gen.oracle_write(bad, max_page_size=1000, force_codec=S.NONE)
    pb = pb.copy()
    pb[0] = 99  # unknown codec id
    ta = _dev(gpu_ctx, pa)
    read.batch_read_columns(gpu_ctx, [read.ColumnPages(S.T_I64, False, ta, ma)])
    read.batch_read_columns(gpu_ctx, [read.ColumnPages(S.T_I32, False, _dev(gpu_ctx, pb), mb)])
    with pytest.raises(N.NativeError):
        gpu_ctx.synchronize()
    del ta
    torch.cuda.empty_cache()
    for seed in (2, 3):
        b = freq_col(seed, 20000 + seed)
        p2, m2 = gen.oracle_write(b, max_page_size=4096, force_codec=S.FREQ)
        got = read.read_simple(gpu_ctx, read.ColumnPages(S.T_I64, False, _dev(gpu_ctx, p2), m2))
        assert np.array_equal(got.values_numpy().view(np.int64), b["values"])


def test_columns_of_many_pages_take_their_bases_from_wave_scans(gpu_ctx):
    """k_colscan gives every page of a binary column its value-byte base and offset base: a wave per column, 64 pages per
    step (binary/mod.rs:121,136-144: the running `last`).  Columns of 1, 63, 64, 65, 130 and 333 pages in ONE call, Dict /
    Basic(LZ4) / plain / OneValue pages (the LZ4 value blocks are queued by the scan), i32 and i64 offsets, nulls; then a
    values buffer that ends inside the 100th page: SB_ERR_INVALID, the size reported, nothing written behind the buffer."""
    import torch
    from strawboat_amd import _native as N
    from strawboat_amd import read
    specs = [(1, S.DICT, False), (63, S.LZ4, False), (64, S.NONE, True), (65, S.DICT, True), (130, S.LZ4, True), (333, S.DICT, False),
             (70, S.ONEVALUE, False)]
    cols, encs = [], []
    for k, (npages, codec, large) in enumerate(specs):
        rows = npages * 300 - 7
        col = gen.binary(rows, uniq=1 if codec == S.ONEVALUE else 40 + k, null_density=0.1 if k % 2 else None, large=large, maxlen=30, seed=20 + k)
        pages, metas = gen.oracle_write(col, max_page_size=300, force_codec=codec)
        assert metas.shape[0] == npages
        cols.append(col)
        encs.append((pages, metas))
    cps = [read.ColumnPages(c["ptype"], c["nullable"], _dev(gpu_ctx, p), m) for c, (p, m) in zip(cols, encs)]
    got = read.batch_read_columns(gpu_ctx, cps)
    gpu_ctx.synchronize()
    for c, (p, m), g in zip(cols, encs, got):
        want = gen.oracle_read(c, p, m)
        assert np.array_equal(g.values_numpy(), want["values"])
        assert np.array_equal(g.offsets_numpy(), want["offsets"])
        if c["nullable"]:
            assert np.array_equal(g.validity_numpy(), want["validity"])
    # the 333-page Dict column into a buffer that ends inside page 100
    col, (pages, metas) = cols[5], encs[5]
    want = gen.oracle_read(col, pages, metas)
    need = want["values"].size
    cap = int(want["offsets"][100 * 300 + 150])
    buf = torch.full((need + 4096,), 0xAB, dtype=torch.uint8, device=gpu_ctx.torch_device)
    batch = read.ReadBatch(gpu_ctx, [cps[5]], values_capacity=[need])
    c = batch._arr[0]
    c.values = C.c_void_p(buf.data_ptr())
    c.values_capacity = cap
    batch.enqueue()
    with pytest.raises(N.NativeError) as ei:
        gpu_ctx.synchronize()
    assert ei.value.code == N.SB_ERR_INVALID
    assert int(c.values_len) == need
    assert (buf[cap:].cpu().numpy() == 0xAB).all()
    c.values_capacity = need
    batch.enqueue()
    gpu_ctx.synchronize()
    assert np.array_equal(buf[:need].cpu().numpy(), want["values"])


def test_host_memory_many_columns_in_groups(gpu_ctx):
    """SB_MEM_HOST with 64 columns (> 32 MB): the call runs in groups — the pages of one group travel back on the copy
    stream while the next group's Arrow buffers travel in; fixed-size outputs of a read are sent as soon as a group's kernels
    are done.  Pageable host buffers, mixed types (Float64 RLE, Int64 plain, Utf8 Dict, an Int32 column that becomes a Freq
    page: its values are written by the second pass AT the synchronize, after the early copy): every page equals the oracle's
    and every buffer reads back (src/write/serialize.rs:36-49, src/read/batch_read.rs:27-64)."""
    import ctypes as C
    from strawboat_amd import _native as N
    from strawboat_amd.types import WriteOptions
    from strawboat_amd.write import options_c
    rows = 120_000
    cols = []
    for k in range(64):
        if k % 4 == 0:
            cols.append(gen.prim(S.T_F64, rows, uniq=64, null_density=0.1, runs=9, seed=k))
        elif k % 4 == 1:
            cols.append(gen.prim(S.T_I64, rows, uniq=1 << 40, seed=k))
        elif k % 4 == 2:
            cols.append(gen.binary(rows, uniq=300, null_density=0.1, seed=k))
        else:
            sp = np.full(rows, 1_000_000, np.int32)
            rng = np.random.default_rng(k)
            m = rng.random(rows) < 0.02
            sp[m] = rng.integers(0, 1 << 30, int(m.sum()))
            cols.append(dict(ptype=S.T_I32, nullable=False, rows=rows, values=sp, validity=None, offsets=None))
    opt = dict(max_page_size=32768, ratio=2.0, forbidden=())
    lib, h = gpu_ctx._lib, gpu_ctx._h
    oc = options_c(WriteOptions(max_page_size=32768, default_compress_ratio=2.0))
    n = len(cols)
    cw = (N.ColumnWriteC * n)()
    keep, outs, metas = [], [], []
    total = 0
    for k, col in enumerate(cols):
        vals = np.ascontiguousarray(col["values"]).view(np.uint8)
        vlen = vals.size if col["offsets"] is not None else 0
        npg = C.c_uint64()
        bound = lib.sb_write_bound(col["ptype"], 1 if col["nullable"] else 0, rows, vlen, C.byref(oc), C.byref(npg))
        out = np.zeros(bound, np.uint8)
        mt = (N.PageMetaC * npg.value)()
        cw[k].physical_type, cw[k].is_nullable, cw[k].rows = col["ptype"], 1 if col["nullable"] else 0, rows
        cw[k].values, cw[k].values_len = vals.ctypes.data, vlen
        if col["validity"] is not None:
            cw[k].validity = col["validity"].ctypes.data
        if col["offsets"] is not None:
            cw[k].offsets = col["offsets"].ctypes.data
        cw[k].out_pages, cw[k].out_capacity = out.ctypes.data, out.size
        cw[k].out_metas, cw[k].n_pages_capacity = mt, npg.value
        keep.append(vals)
        outs.append(out)
        metas.append(mt)
        total += vals.size
    assert total > (32 << 20)
    for rep in range(2):   # (the second call finds the plan of the groups)
        gpu_ctx._check(lib.sb_write_columns(h, cw, n, C.byref(oc), N.SB_MEM_HOST))
        gpu_ctx.synchronize()
        for kk in (list(range(n)) if rep == 0 else [0, 1, 2, 3, n - 1]):
            want_pages, want_metas = gen.oracle_write(cols[kk], **opt)
            assert cw[kk].out_len == want_pages.size and np.array_equal(outs[kk][:cw[kk].out_len], want_pages), (rep, kk)
    # read everything back into pageable host buffers
    cr = (N.ColumnReadC * n)()
    bufs = []
    for k, col in enumerate(cols):
        w = 8 if col["ptype"] in (S.T_F64, S.T_I64) else 4
        m = np.array([[metas[k][q].length, metas[k][q].num_values] for q in range(int(cw[k].n_pages))], np.uint64)
        want = None
        if col["offsets"] is not None:   # (a null row of a Dict page reads back as the string before it: sizes and bytes from the oracle's decode)
            wp, wm = gen.oracle_write(col, **opt)
            want = gen.oracle_read(col, wp, wm)
        v_out = np.zeros(want["values"].size + 64 if want is not None else rows * w, np.uint8)
        b_out = np.zeros((rows + 31) // 32 * 4, np.uint8)
        o_out = np.zeros((rows + 1) * 4, np.uint8)
        cr[k].physical_type, cr[k].is_nullable = col["ptype"], 1 if col["nullable"] else 0
        cr[k].pages, cr[k].pages_len = outs[k].ctypes.data, int(cw[k].out_len)
        cr[k].metas, cr[k].n_pages = m.ctypes.data_as(C.POINTER(N.PageMetaC)), m.shape[0]
        cr[k].values, cr[k].values_capacity = v_out.ctypes.data, v_out.size
        if col["nullable"]:
            cr[k].validity, cr[k].validity_capacity = b_out.ctypes.data, b_out.size
        if col["offsets"] is not None:
            cr[k].offsets, cr[k].offsets_capacity = o_out.ctypes.data, o_out.size
        bufs.append((m, v_out, b_out, o_out, want))
    for rep in range(2):
        for _, v_out, b_out, o_out, _w in bufs:
            v_out[:] = 0
        gpu_ctx._check(lib.sb_read_columns(h, cr, n, N.SB_MEM_HOST))
        gpu_ctx.synchronize()
        for k, col in enumerate(cols):
            m, v_out, b_out, o_out, want = bufs[k]
            if col["offsets"] is not None:
                assert cr[k].values_len == want["values"].size
                assert np.array_equal(o_out, want["offsets"]), (rep, k)
                assert np.array_equal(v_out[:cr[k].values_len], want["values"]), (rep, k)
                assert np.array_equal(b_out[:(rows + 7) // 8], want["validity"]), (rep, k)
            elif col["validity"] is not None:
                valid = np.unpackbits(col["validity"], bitorder="little")[:rows].astype(bool)
                w = 8
                assert np.array_equal(v_out.view(np.uint64)[valid], keep[k].view(np.uint64)[valid]), (rep, k)
                assert np.array_equal(b_out[:(rows + 7) // 8], col["validity"][:(rows + 7) // 8]), (rep, k)
            else:
                assert np.array_equal(v_out, keep[k]), (rep, k)
