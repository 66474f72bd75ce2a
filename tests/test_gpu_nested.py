"""GPU parity of the nested (Dremel) path through the C ABI:
  * level sections produced by sb_nested_write_levels == oracle's write_nested_validity, byte for byte;
  * whole nested pages (levels + leaf BLOCK through the flat encoder with explicit paging) == the
    oracle's composition of the same page;
  * sb_nested_read_levels + flat decode rebuild the Arrow buffers the levels were made from
    (the shapes of the reference's tests/it/io.rs:167-278: list, list_list, list_struct, struct_list)."""
import numpy as np
import pytest

from oracle import sbo as S
from tests import gen
from tests.nested_gen import expected_state, make_nested

pytestmark = pytest.mark.gpu

SHAPES = ["list", "large_list", "list_list", "list_struct", "struct_list", "struct_struct", "list_required"]


def up(ctx, a):
    import torch
    if a is None:
        return None
    return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(ctx.torch_device)


def device_levels(ctx, levels):
    from strawboat_amd.nested import NestedLevel
    return [NestedLevel(lv["kind"], bool(lv["is_optional"]), lv["length"], up(ctx, lv.get("validity")),
                        up(ctx, lv.get("offsets"))) for lv in levels]


def leaf_values(levels, ptype, seed, runs=None):
    n = levels[-1]["length"]
    col = gen.prim(ptype, max(n, 1), uniq=50, seed=seed, runs=runs)
    w = col["values"].dtype.itemsize
    return col["values"][:n], w


def oracle_pages(levels, ptype, values, rows, page_rows, **opt):
    """The bytes write_nested would emit page by page: level section + compress_* of the leaf slice."""
    out, metas = [], []
    leaf = levels[-1]
    for r0 in range(0, rows, page_rows):
        ln = min(page_rows, rows - r0)
        b, nv, ls, lc = S.nested_write_levels(levels, r0, ln)
        o = S.make_options(max_page_size=None, **opt)
        if lc:
            blk, _ = S.write_column(ptype, False, lc, values[ls:ls + lc], leaf.get("validity"), None, o,
                                    validity_bit_offset=ls)
        else:  # write_nested still compresses the (empty) leaf slice: hdr9 + Basic(default) of zero bytes
            blk = S.write_page(ptype, False, 0, options=o)
        out += [np.asarray(b), blk]
        metas.append((len(b) + blk.size, nv))
    return np.concatenate(out), np.array(metas, np.uint64).reshape(-1, 2)


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("seed", [1, 2])
def test_level_sections_match_oracle(gpu_ctx, shape, seed):
    from strawboat_amd import nested
    levels, rows = make_nested(shape, 3000, seed)
    dl = device_levels(gpu_ctx, levels)
    for mps in (None, 1000, 777, 1):
        if mps == 1 and seed == 2:
            continue
        got = nested.write_levels(gpu_ctx, dl, rows, mps)
        ps = rows if mps is None else mps
        sec = got.sections.cpu().numpy()
        off = 0
        assert got.n_pages == (rows + ps - 1) // ps
        for p, r0 in enumerate(range(0, rows, ps)):
            ln = min(ps, rows - r0)
            b, nv, ls, lc = S.nested_write_levels(levels, r0, ln)
            assert (int(got.num_values[p]), int(got.leaf_start[p]), int(got.leaf_count[p])) == (nv, ls, lc), (p, r0)
            assert int(got.level_bytes[p]) == len(b), (p, r0)
            assert bytes(sec[off:off + len(b)]) == bytes(b), "page %d (rows %d..%d, mps %s)" % (p, r0, r0 + ln, mps)
            off += len(b)


@pytest.mark.parametrize("shape", ["list", "list_list", "list_struct", "struct_list"])
@pytest.mark.parametrize("codec,ptype", [(S.NONE, S.T_I32), (S.RLE, S.T_I64), (S.DICT, S.T_F64), (S.LZ4, S.T_I32)])
def test_nested_pages_match_oracle(gpu_ctx, shape, codec, ptype):
    from strawboat_amd import WriteOptions, nested
    from strawboat_amd.write import DeviceColumn
    levels, rows = make_nested(shape, 5000, 3)
    values, w = leaf_values(levels, ptype, 5, runs=6)
    leaf = levels[-1]
    opt = dict(force_codec=codec) if codec != S.LZ4 else dict(default_compression=S.LZ4)
    want_pages, want_metas = oracle_pages(levels, ptype, values, rows, 1024, **opt)
    wo = WriteOptions(default_compression=opt.get("default_compression", 0), max_page_size=1024,
                      force_codec=opt.get("force_codec", -1), lz4_exact=True)
    dcol = DeviceColumn(ptype, False, leaf["length"], up(gpu_ctx, values), up(gpu_ctx, leaf.get("validity")))
    enc = nested.write_nested(gpu_ctx, device_levels(gpu_ctx, levels), dcol, wo)
    assert np.array_equal(enc.metas_array(), want_metas)
    got = enc.pages_numpy()
    assert got.size == want_pages.size
    assert np.array_equal(got, want_pages), "first mismatch at %d" % int(np.argmax(got != want_pages))


def check_decoded(arr, levels, rows, values, w):
    want = expected_state(levels, 0, rows)
    assert arr.lengths == want["lengths"]
    for k, lv in enumerate(levels):
        if lv["kind"] in (S.K_LIST, S.K_LARGE_LIST):
            offs = arr.offsets_numpy(k)
            assert offs[:-1].tolist() == want["offsets"][k], "offsets of level %d" % k
            assert int(offs[-1]) == want["lengths"][k + 1]
        if lv["is_optional"] and lv["kind"] != S.K_PRIMITIVE:
            bits = np.unpackbits(arr.validity_numpy(k), bitorder="little")[:want["lengths"][k]]
            assert bits.tolist() == want["validity"][k], "validity of level %d" % k
    if levels[-1]["is_optional"]:
        bits = np.unpackbits(arr.leaf.validity_numpy(), bitorder="little")[:want["leaf_count"]]
        assert bits.tolist() == want["leaf_validity"]
    got = arr.leaf.values[:want["leaf_count"] * w].cpu().numpy()
    assert np.array_equal(got, np.ascontiguousarray(values).view(np.uint8)[:want["leaf_count"] * w])


@pytest.mark.parametrize("shape", SHAPES)
def test_read_oracle_written_nested_pages(gpu_ctx, shape):
    from strawboat_amd import nested
    from strawboat_amd.read import ColumnPages
    levels, rows = make_nested(shape, 4000, 7)
    values, w = leaf_values(levels, S.T_I32, 11)
    pages, metas = oracle_pages(levels, S.T_I32, values, rows, 900, force_codec=S.NONE)
    col = ColumnPages(S.T_I32, False, up(gpu_ctx, pages), metas)
    arr = nested.read_nested(gpu_ctx, col, [lv["kind"] for lv in levels], [bool(lv["is_optional"]) for lv in levels])
    check_decoded(arr, levels, rows, values, w)


@pytest.mark.parametrize("shape", ["list_list", "list_struct"])
def test_large_round_trip_on_device(gpu_ctx, shape):
    """size-independent property at a size the oracle would not finish quickly: encode -> decode on
    the device rebuilds the Arrow buffers the levels came from"""
    from strawboat_amd import WriteOptions, nested
    from strawboat_amd.read import ColumnPages
    from strawboat_amd.write import DeviceColumn
    levels, rows = make_nested(shape, 300_000, 9)
    values, w = leaf_values(levels, S.T_I64, 13)
    leaf = levels[-1]
    dcol = DeviceColumn(S.T_I64, False, leaf["length"], up(gpu_ctx, values), up(gpu_ctx, leaf.get("validity")))
    enc = nested.write_nested(gpu_ctx, device_levels(gpu_ctx, levels), dcol,
                              WriteOptions(max_page_size=65536, force_codec=S.NONE))
    col = ColumnPages(S.T_I64, False, enc.pages[:enc.length].contiguous(), enc.metas_array())
    arr = nested.read_nested(gpu_ctx, col, [lv["kind"] for lv in levels], [bool(lv["is_optional"]) for lv in levels])
    want = expected_state(levels, 0, rows)
    assert arr.lengths == want["lengths"]
    for k, lv in enumerate(levels):
        if lv["kind"] in (S.K_LIST, S.K_LARGE_LIST):
            assert np.array_equal(arr.offsets_numpy(k), np.asarray(lv["offsets"]).astype(np.int64))
    got = arr.leaf.values[:want["leaf_count"] * w].cpu().numpy()
    assert np.array_equal(got, values.view(np.uint8)[:want["leaf_count"] * w])
    bits = np.unpackbits(arr.leaf.validity_numpy(), bitorder="little")[:want["leaf_count"]]
    assert np.array_equal(bits, np.unpackbits(leaf["validity"], bitorder="little")[:want["leaf_count"]])


def _leaf_column(gpu_ctx, levels, ptype, seed):
    """leaf buffers of `ptype` for the leaf level: (DeviceColumn, host values, host offsets or None)"""
    from strawboat_amd.write import DeviceColumn
    leaf = levels[-1]
    n = leaf["length"]
    if ptype == S.T_BIN32:
        col = gen.binary(max(n, 1), uniq=20, seed=seed, maxlen=9)
        offs = col["offsets"][:n + 1]
        vals = col["values"][:int(offs[-1])]
        return DeviceColumn(ptype, False, n, up(gpu_ctx, vals), up(gpu_ctx, leaf.get("validity")), up(gpu_ctx, offs)), vals, offs
    if ptype == S.T_BOOL:
        col = gen.boolean(max(n, 8), seed=seed)
        return DeviceColumn(ptype, False, n, up(gpu_ctx, col["values"]), up(gpu_ctx, leaf.get("validity"))), col["values"], None
    values, _ = leaf_values(levels, ptype, seed)
    return DeviceColumn(ptype, False, n, up(gpu_ctx, values), up(gpu_ctx, leaf.get("validity"))), values, None


def _oracle_nested_pages(levels, ptype, values, offsets, rows, page_rows, **opt):
    out, metas = [], []
    leaf = levels[-1]
    for r0 in range(0, rows, page_rows):
        ln = min(page_rows, rows - r0)
        b, nv, ls, lc = S.nested_write_levels(levels, r0, ln)
        o = S.make_options(max_page_size=None, **opt)
        if ptype == S.T_BIN32:
            blk = S.write_page(ptype, False, lc, values, leaf.get("validity"), offsets[ls:ls + lc + 1], o,
                               validity_bit_offset=ls)
        elif ptype == S.T_BOOL:
            blk = S.write_page(ptype, False, lc, values, leaf.get("validity"), None, o, values_bit_offset=ls,
                               validity_bit_offset=ls)
        else:
            blk = S.write_page(ptype, False, lc, values[ls:ls + lc], leaf.get("validity"), None, o,
                               validity_bit_offset=ls)
        out += [np.asarray(b), blk]
        metas.append((len(b) + blk.size, nv))
    return np.concatenate(out), np.array(metas, np.uint64).reshape(-1, 2)


@pytest.mark.parametrize("dc", [S.NONE, S.LZ4, S.ZSTD, S.SNAPPY])
@pytest.mark.parametrize("ptype", [S.T_I32, S.T_F64, S.T_BIN32, S.T_BOOL])
@pytest.mark.parametrize("ratio", [None, 2.0])
def test_pages_without_leaf_slots(gpu_ctx, dc, ptype, ratio):
    """2 top-level rows per page: many pages hold only empty / null lists.  write_nested still compresses the
    empty leaf slice (write/serialize.rs:134-198): hdr9 + Basic(default) of zero bytes — and the reader needs it
    (read_compress_header on an absent block is an UnexpectedEof upstream)."""
    from strawboat_amd import WriteOptions, nested
    from strawboat_amd.read import ColumnPages
    levels, rows = make_nested("list", 400, 21)
    dcol, values, offsets = _leaf_column(gpu_ctx, levels, ptype, 4)
    want_pages, want_metas = _oracle_nested_pages(levels, ptype, values, offsets, rows, 2, default_compression=dc,
                                                  ratio=ratio)
    enc = nested.write_nested(gpu_ctx, device_levels(gpu_ctx, levels), dcol,
                              WriteOptions(default_compression=dc, max_page_size=2, default_compress_ratio=ratio, lz4_exact=True))
    got = enc.pages_numpy()
    if dc in (S.SNAPPY, S.ZSTD):   # valid Snappy / Zstd streams, not the oracle's bytes (literal-only Snappy; the device's Zstd
        # encoder compresses, the oracle's stores): same page count / entries, and the pages decode
        assert np.array_equal(enc.metas_array()[:, 1], want_metas[:, 1])
        back = nested.read_nested(gpu_ctx, ColumnPages(ptype, False, up(gpu_ctx, got), enc.metas_array()),
                                  [lv["kind"] for lv in levels], [bool(lv["is_optional"]) for lv in levels])
        assert back.lengths == expected_state(levels, 0, rows)["lengths"]
    else:
        assert np.array_equal(enc.metas_array(), want_metas)
        assert np.array_equal(got, want_pages), "first mismatch at %d" % int(np.argmax(got[:want_pages.size] != want_pages[:got.size]))
    # and the oracle-written pages decode (zero-row blocks are parsed, not skipped)
    arr = nested.read_nested(gpu_ctx, ColumnPages(ptype, False, up(gpu_ctx, want_pages), want_metas),
                             [lv["kind"] for lv in levels], [bool(lv["is_optional"]) for lv in levels])
    want = expected_state(levels, 0, rows)
    assert arr.lengths == want["lengths"]
    if ptype == S.T_I32 or ptype == S.T_F64:
        w = values.dtype.itemsize
        assert np.array_equal(arr.leaf.values[:want["leaf_count"] * w].cpu().numpy(), values.view(np.uint8)[:want["leaf_count"] * w])
    elif ptype == S.T_BIN32:
        assert np.array_equal(arr.leaf.offsets_numpy().view(np.int32), offsets)
        assert np.array_equal(arr.leaf.values_numpy(), values)


@pytest.mark.parametrize("ptype", [S.T_I64, S.T_BIN32])
def test_column_of_empty_lists(gpu_ctx, ptype):
    """every list empty or null: the leaf array has zero slots in total, every page still carries a block"""
    from strawboat_amd import WriteOptions, nested
    rng = np.random.default_rng(5)
    rows = 1000
    valid = rng.random(rows) > 0.3
    levels = [dict(kind=S.K_LIST, is_optional=True, validity=gen.pack_bits(valid), offsets=np.zeros(rows + 1, np.int32), length=rows),
              dict(kind=S.K_PRIMITIVE, is_optional=True, validity=None, length=0)]
    dcol, values, offsets = _leaf_column(gpu_ctx, levels, ptype, 4)
    for dc in (S.NONE, S.LZ4):
        want_pages, want_metas = _oracle_nested_pages(levels, ptype, values, offsets, rows, 300, default_compression=dc, ratio=2.0)
        enc = nested.write_nested(gpu_ctx, device_levels(gpu_ctx, levels), dcol,
                                  WriteOptions(default_compression=dc, max_page_size=300, default_compress_ratio=2.0, lz4_exact=True))
        assert np.array_equal(enc.metas_array(), want_metas)
        assert np.array_equal(enc.pages_numpy(), want_pages)


@pytest.mark.parametrize("dc", [S.NONE, S.LZ4, S.ZSTD])
def test_leaves_of_an_array_in_one_call(gpu_ctx, dc):
    """write_nested_leaves / read_nested_leaves (all leaves of a nested array through ONE sb_write_columns / sb_read_columns
    call, as encode_chunk loops over an array's leaves, src/write/common.rs:60-116): the pages of every leaf are the
    pages the one-leaf calls write, and the batch read rebuilds the same buffers — Int64, Utf8 and Boolean leaves under
    two different nestings"""
    from strawboat_amd import WriteOptions, nested
    from strawboat_amd.read import ColumnPages
    items, singles, metas = [], [], []
    opts = WriteOptions(max_page_size=4096, default_compression=dc)
    for shape, ptype, seed in (("list_struct", S.T_I64, 3), ("list_struct", S.T_BIN32, 4), ("list_list", S.T_BOOL, 5)):
        levels, rows = make_nested(shape, 20_000, seed)
        dcol, vals, offs = _leaf_column(gpu_ctx, levels, ptype, seed + 10)
        dl = device_levels(gpu_ctx, levels)
        items.append((dl, dcol))
        one = nested.write_nested(gpu_ctx, dl, dcol, opts)
        singles.append((one.pages_numpy().copy(), one.metas_array().copy()))
        metas.append((levels, ptype))
    encs = nested.write_nested_leaves(gpu_ctx, items, opts)
    assert len(encs) == 3
    cps, kinds, nul = [], [], []
    for e, (pages1, metas1), (levels, ptype) in zip(encs, singles, metas):
        assert np.array_equal(e.metas_array(), metas1)
        assert np.array_equal(e.pages_numpy(), pages1), "a leaf's pages differ between the batch and the one-leaf call"
        cps.append(ColumnPages(ptype, False, e.pages[:e.length].contiguous(), e.metas_array()))
        kinds.append([lv["kind"] for lv in levels])
        nul.append([bool(lv["is_optional"]) for lv in levels])
    arrs = nested.read_nested_leaves(gpu_ctx, cps, kinds, nul)
    for arr, cp, k, o, (levels, ptype) in zip(arrs, cps, kinds, nul, metas):
        one = nested.read_nested(gpu_ctx, cp, k, o)
        assert arr.lengths == one.lengths
        for lvl, lv in enumerate(levels):
            if lv["kind"] in (S.K_LIST, S.K_LARGE_LIST):
                assert np.array_equal(arr.offsets_numpy(lvl), one.offsets_numpy(lvl))
                assert np.array_equal(arr.offsets_numpy(lvl), np.asarray(lv["offsets"]).astype(np.int64))
        assert np.array_equal(arr.leaf.values_numpy(), one.leaf.values_numpy())
        assert np.array_equal(arr.leaf.validity_numpy(), one.leaf.validity_numpy())
        if ptype == S.T_BIN32:
            assert np.array_equal(arr.leaf.offsets_numpy(), one.leaf.offsets_numpy())


def test_batch_objects_run_repeatedly(gpu_ctx):
    """NestedWriteBatch / NestedReadBatch (descriptors and buffers built once): every run() gives the pages / buffers of the
    one-shot calls, also after the output buffers were overwritten in between"""
    from strawboat_amd import WriteOptions, nested
    from strawboat_amd.read import ColumnPages
    opts = WriteOptions(max_page_size=2048, default_compression=S.ZSTD)
    items, shapes = [], []
    for shape, ptype, seed in (("list_struct", S.T_I64, 7), ("list_struct", S.T_BIN32, 8)):
        levels, rows = make_nested(shape, 30_000, seed)
        dcol, vals, offs = _leaf_column(gpu_ctx, levels, ptype, seed + 10)
        items.append((device_levels(gpu_ctx, levels), dcol))
        shapes.append((levels, ptype))
    want = [(e.pages_numpy().copy(), e.metas_array().copy()) for e in nested.write_nested_leaves(gpu_ctx, items, opts)]
    wb = nested.NestedWriteBatch(gpu_ctx, items, opts)
    for rep in range(3):
        encs = wb.run()
        for e, (p, m) in zip(encs, want):
            assert np.array_equal(e.metas_array(), m), rep
            assert np.array_equal(e.pages_numpy(), p), rep
        for e in encs:
            e.pages.zero_()   # the next run has to write everything again
    encs = wb.run()
    cps = [ColumnPages(pt, False, e.pages[:e.length].contiguous(), e.metas_array()) for e, (_, pt) in zip(encs, shapes)]
    kinds = [[lv["kind"] for lv in levels] for levels, _ in shapes]
    nul = [[bool(lv["is_optional"]) for lv in levels] for levels, _ in shapes]
    ref = nested.read_nested_leaves(gpu_ctx, cps, kinds, nul)
    ref_vals = [(a.leaf.values_numpy().copy(), a.leaf.validity_numpy().copy(), a.offsets_numpy(0).copy(), list(a.lengths)) for a in ref]
    rb = nested.NestedReadBatch(gpu_ctx, cps, kinds, nul)
    for rep in range(3):
        arrs = rb.run()
        for a, (v, vd, o, ln) in zip(arrs, ref_vals):
            assert a.lengths == ln
            assert np.array_equal(a.leaf.values_numpy(), v), rep
            assert np.array_equal(a.leaf.validity_numpy(), vd), rep
            assert np.array_equal(a.offsets_numpy(0), o), rep
        for a in arrs:
            a.leaf.values.zero_()
            a.offsets[0].zero_()


@pytest.mark.parametrize("what", ["lowcard_i64", "sparse_i32", "utf8", "runs_f64"])
def test_long_leaf_pages_adaptive(gpu_ctx, what):
    """a nested array written without max_page_size: its leaf pages hold hundreds of thousands of slots and go through the
    section-parallel selector and the long-page Dict / Freq / RLE writers (sb_select_big.h, sb_dict_big.h, sb_freq_big.h)
    behind their level sections; bytes == the oracle's, and the pages read back"""
    from strawboat_amd import WriteOptions, nested
    from strawboat_amd.read import ColumnPages
    from strawboat_amd.write import DeviceColumn
    levels, rows = make_nested("list_struct", 700_000, 21)
    leaf = levels[-1]
    n = leaf["length"]
    assert n >= 2 * (1 << 18) + 40_000, n
    rng = np.random.default_rng(8)
    offs = None
    if what == "lowcard_i64":
        ptype, values = S.T_I64, rng.integers(0, 300, n).astype(np.int64) * 1_000_003
    elif what == "sparse_i32":
        ptype, values = S.T_I32, np.where(rng.random(n) < 0.02, rng.integers(0, 1 << 30, n), 1_000_000).astype(np.int32)
    elif what == "runs_f64":
        ptype, values = S.T_F64, np.repeat(rng.integers(0, 50, n // 40 + 1), 40)[:n].astype(np.float64)
    else:
        ptype = S.T_BIN32
        col = gen.binary(n, uniq=2000, zipf=1.2, maxlen=16, seed=3)
        values, offs = col["values"], col["offsets"]
    page_rows = 350_000   # two pages: the second one's leaf slots start in the middle of the buffers
    want_pages, want_metas = _oracle_nested_pages(levels, ptype, values, offs, rows, page_rows, ratio=2.0, forbidden=())
    dcol = DeviceColumn(ptype, False, n, up(gpu_ctx, values), up(gpu_ctx, leaf.get("validity")), up(gpu_ctx, offs))
    enc = nested.write_nested(gpu_ctx, device_levels(gpu_ctx, levels), dcol,
                              WriteOptions(max_page_size=page_rows, default_compress_ratio=2.0, forbidden_compressions=[], lz4_exact=True))
    assert np.array_equal(enc.metas_array(), want_metas)
    got = enc.pages_numpy()
    assert got.size == want_pages.size and np.array_equal(got, want_pages), "first mismatch at %d" % int(np.argmax(got[:want_pages.size] != want_pages[:got.size]))
    arr = nested.read_nested(gpu_ctx, ColumnPages(ptype, False, enc.pages[:enc.length].contiguous(), enc.metas_array()),
                             [lv["kind"] for lv in levels], [bool(lv["is_optional"]) for lv in levels])
    valid = np.unpackbits(np.asarray(leaf["validity"]), bitorder="little")[:n].astype(bool) if leaf.get("validity") is not None else np.ones(n, bool)
    if offs is None:
        gv = arr.leaf.values_numpy().view(values.dtype)[:n]
        assert np.array_equal(gv[valid], values[valid])
    else:   # (a null slot decodes to the string of the index before it: the valid slots are the input's)
        go = arr.leaf.offsets_numpy().view(np.int32)[:n + 1].astype(np.int64)
        gvals = arr.leaf.values_numpy()
        o64 = offs.astype(np.int64)
        assert np.array_equal((go[1:] - go[:-1])[valid], (o64[1:n + 1] - o64[:n])[valid])
        for i in np.flatnonzero(valid)[::211]:
            assert bytes(gvals[go[i]:go[i + 1]]) == bytes(values[o64[i]:o64[i + 1]]), i


def test_batch_objects_enqueue_levels_and_leaves_together(gpu_ctx):
    """run() after the first enqueues the level call and the leaf call TOGETHER, the leaf BLOCKs with the page cut of the
    run before, and checks the cut when the results are back.  A run whose list lengths moved a leaf slot across a page
    border (same rows, pages and slots in total) must notice and write / read the BLOCKs again with the new cut."""
    import torch
    from strawboat_amd import WriteOptions, nested
    from strawboat_amd.read import ColumnPages
    opts = WriteOptions(max_page_size=2048, default_compression=S.LZ4)
    levels_a, rows = make_nested("list_struct", 20_000, 3)
    levels_b = [dict(lv) for lv in levels_a]
    offs = np.asarray(levels_a[0]["offsets"]).copy()
    r1 = next(r for r in range(100, 2048) if offs[r + 1] - offs[r] >= 1)       # a non-empty list in page 0 ...
    lvalid = np.unpackbits(np.asarray(levels_a[0]["validity"]), bitorder="little")
    r2 = next(r for r in range(2048 + 100, 4096) if lvalid[r])                  # ... gives one element to a (non-null) row of page 1
    offs[r1 + 1:r2 + 1] -= 1
    levels_b[0] = dict(levels_a[0], offsets=offs)
    want = {}
    for name, levels in (("a", levels_a), ("b", levels_b)):
        dcol, vals, offs_leaf = _leaf_column(gpu_ctx, levels, S.T_I64, 13)
        enc = nested.write_nested_leaves(gpu_ctx, [(device_levels(gpu_ctx, levels), dcol)], opts)[0]
        want[name] = (enc.pages_numpy().copy(), enc.metas_array().copy())
    assert not np.array_equal(want["a"][1], want["b"][1])
    dl = device_levels(gpu_ctx, levels_a)
    dcol, vals, offs_leaf = _leaf_column(gpu_ctx, levels_a, S.T_I64, 13)
    wb = nested.NestedWriteBatch(gpu_ctx, [(dl, dcol)], opts)
    for name in ("a", "a", "b", "b", "a"):
        src = levels_a if name == "a" else levels_b
        dl[0].offsets.copy_(torch.from_numpy(np.ascontiguousarray(src[0]["offsets"]).view(np.uint8).reshape(-1).copy()).to(dl[0].offsets.device))
        gpu_ctx.synchronize()
        enc = wb.run()[0]
        assert np.array_equal(enc.metas_array(), want[name][1]), name
        assert np.array_equal(enc.pages_numpy(), want[name][0]), name
    # read side: a stale page table (what a run over other pages would leave) is noticed and the BLOCKs are read again
    enc = wb.run()[0]
    kinds = [[lv["kind"] for lv in levels_a]]
    nul = [[bool(lv["is_optional"]) for lv in levels_a]]
    cp = ColumnPages(S.T_I64, False, enc.pages[:enc.length].contiguous(), enc.metas_array())
    ref = nested.read_nested_leaves(gpu_ctx, [cp], kinds, nul)[0]
    rv, ro = ref.leaf.values_numpy().copy(), ref.offsets_numpy(0).copy()
    rb = nested.NestedReadBatch(gpu_ctx, [cp], kinds, nul)
    for rep in range(3):
        if rep == 1:
            leaf_metas = rb.prep[0][10]
            leaf_metas[0, 1] -= 1     # (page 0 one slot short, page 1 one too many)
            leaf_metas[1, 1] += 1
        arr = rb.run()[0]
        assert np.array_equal(arr.leaf.values_numpy(), rv), rep
        assert np.array_equal(arr.offsets_numpy(0), ro), rep
