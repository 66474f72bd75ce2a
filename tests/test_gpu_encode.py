"""GPU parity: Arrow buffers encoded by the HIP path through the C ABI must equal, byte for
byte, the pages the CPU oracle writes for the same input and options (deterministic codecs),
and decode back (GPU) to what the oracle decodes."""
import numpy as np
import pytest

from oracle import sbo as S
from tests import gen

pytestmark = pytest.mark.gpu


def to_device_column(ctx, col):
    import torch
    from strawboat_amd.write import DeviceColumn
    dev = ctx.torch_device

    def up(a):
        if a is None:
            return None
        return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)
    return DeviceColumn(col["ptype"], col["nullable"], col["rows"], up(col["values"]), up(col["validity"]),
                        up(col["offsets"]))


def gpu_encode(ctx, col, **opt):
    from strawboat_amd import write, WriteOptions
    wo = WriteOptions(default_compression=opt.get("default_compression", 0),
                      default_compress_ratio=opt.get("ratio"), max_page_size=opt.get("max_page_size"),
                      forbidden_compressions=list(opt.get("forbidden", ())), force_codec=opt.get("force_codec", -1),
                      force_index_codec=opt.get("force_index_codec", -1), rng_seed=opt.get("rng_seed", 42),
                      debug_verify_fail=opt.get("debug_verify_fail", False),
                      lz4_exact=opt.get("lz4_exact", True))   # byte parity with the oracle (== liblz4) needs the exact parse;
                                                              # lz4_exact=False = the default (parallel) encoder bench.py times
    return write.write(ctx, to_device_column(ctx, col), wo)


def check(ctx, col, **opt):
    want_pages, want_metas = gen.oracle_write(col, **opt)
    enc = gpu_encode(ctx, col, **opt)
    got_metas = enc.metas_array()
    assert np.array_equal(got_metas, want_metas), "PageMeta differ:\n%s\n%s" % (got_metas[:4], want_metas[:4])
    got = enc.pages_numpy()
    if not np.array_equal(got, want_pages):
        bad = int(np.argmax(got != want_pages[:got.size])) if got.size == want_pages.size else -1
        raise AssertionError("page bytes differ (first mismatch at byte %d of %d)" % (bad, want_pages.size))


PRIMS = [S.T_I8, S.T_I16, S.T_I32, S.T_I64, S.T_U8, S.T_U16, S.T_U32, S.T_U64, S.T_F32, S.T_F64, S.T_I128, S.T_I256]


@pytest.mark.parametrize("ptype", PRIMS)
@pytest.mark.parametrize("codec", [S.NONE, S.RLE, S.DICT, S.ONEVALUE])
def test_prim_codecs(gpu_ctx, ptype, codec):
    uniq = 1 if codec == S.ONEVALUE else 100
    col = gen.prim(ptype, 10_000, uniq=uniq, null_density=0.2, runs=8)
    check(gpu_ctx, col, max_page_size=2048, force_codec=codec)
    col = gen.prim(ptype, 10_000, uniq=uniq, runs=3)
    check(gpu_ctx, col, max_page_size=4100, force_codec=codec)


def test_rle_edge_cases(gpu_ctx):
    # leading nulls, all-null page, nulls between equal values, -0.0/+0.0 and NaN runs
    vals = np.array([5, 5, 7, 7, 7, 1, 1, 2] * 700, np.int32)
    valid = np.ones(vals.size, bool)
    valid[:3] = False
    valid[4096:8192] = False          # pages of 2048 rows that are entirely null
    valid[9000:9005] = False
    col = dict(ptype=S.T_I32, nullable=True, rows=vals.size, values=vals, validity=gen.pack_bits(valid), offsets=None)
    check(gpu_ctx, col, max_page_size=2048, force_codec=S.RLE)
    check(gpu_ctx, col, max_page_size=2048, force_codec=S.DICT)
    f = np.array([0.0, -0.0, np.nan, np.nan, 1.5, 1.5, -0.0, 0.0] * 600, np.float64)
    f.view(np.uint64)[3::8] |= 1      # a second NaN payload
    col = dict(ptype=S.T_F64, nullable=False, rows=f.size, values=f, validity=None, offsets=None)
    check(gpu_ctx, col, max_page_size=1000, force_codec=S.RLE)
    check(gpu_ctx, col, max_page_size=1000, force_codec=S.DICT)


@pytest.mark.parametrize("ptype", [S.T_I32, S.T_U32])
@pytest.mark.parametrize("codec", [S.BITPACK, S.DELTABP])
def test_bitpacking(gpu_ctx, ptype, codec):
    col = gen.prim(ptype, 128 * 100, uniq=1 << 13, sorted_=(codec == S.DELTABP))
    check(gpu_ctx, col, max_page_size=128 * 40, force_codec=codec)
    col = gen.prim(ptype, 128 * 1024, uniq=1 << 30, sorted_=(codec == S.DELTABP), seed=7)
    check(gpu_ctx, col, max_page_size=65536, force_codec=codec)
    col = gen.prim(ptype, 128 * 64, uniq=2)   # 0/1 bit widths
    check(gpu_ctx, col, max_page_size=128 * 64, force_codec=codec)


@pytest.mark.parametrize("icodec", [S.NONE, S.RLE, S.BITPACK, S.DELTABP, S.ONEVALUE])
def test_dict_index_codecs(gpu_ctx, icodec):
    uniq = 1 if icodec == S.ONEVALUE else 200
    col = gen.prim(S.T_F64, 128 * 300, uniq=uniq, null_density=0.1, runs=16, sorted_=(icodec == S.DELTABP))
    check(gpu_ctx, col, max_page_size=128 * 100, force_codec=S.DICT, force_index_codec=icodec)


def test_c1_int64_single_page(gpu_ctx):
    rng = np.random.default_rng(42)
    vals = rng.integers(0, 1 << 62, 1_000_000).astype(np.int64)
    col = dict(ptype=S.T_I64, nullable=False, rows=vals.size, values=vals, validity=None, offsets=None)
    check(gpu_ctx, col)


def test_c2_float64_pages(gpu_ctx):
    col = gen.prim(S.T_F64, 1_000_000, uniq=256, null_density=0.1, runs=32)
    for codec in (S.RLE, S.DICT, S.NONE):
        check(gpu_ctx, col, max_page_size=65536, force_codec=codec)


@pytest.mark.parametrize("codec", [S.NONE, S.RLE, S.ONEVALUE])
def test_boolean(gpu_ctx, codec):
    p = 1.0 if codec == S.ONEVALUE else 0.5
    col = gen.boolean(100_003, null_density=0.3, p_true=p, runs=5)
    check(gpu_ctx, col, max_page_size=8192, force_codec=codec)
    col = gen.boolean(10_000, p_true=p, runs=40)
    check(gpu_ctx, col, max_page_size=1000, force_codec=codec)  # pages start at non byte-aligned bits


@pytest.mark.parametrize("large", [False, True])
@pytest.mark.parametrize("codec", [S.NONE, S.DICT, S.ONEVALUE])
def test_binary(gpu_ctx, codec, large):
    uniq = 1 if codec == S.ONEVALUE else 300
    col = gen.binary(20_000, uniq=uniq, null_density=0.1, large=large, zipf=1.3)
    check(gpu_ctx, col, max_page_size=4096, force_codec=codec)
    col = gen.binary(5_000, uniq=uniq, large=large)
    check(gpu_ctx, col, max_page_size=5000, force_codec=codec, force_index_codec=S.RLE)


def test_roundtrip_gpu_only(gpu_ctx):
    """encode on the GPU, decode on the GPU, compare with the input (tests/it/io.rs:440-528)."""
    import torch
    from strawboat_amd import read, write, WriteOptions
    col = gen.prim(S.T_I64, 300_000, uniq=1 << 40)
    dc = to_device_column(gpu_ctx, col)
    enc = write.write(gpu_ctx, dc, WriteOptions(max_page_size=65536))
    dec = read.read_simple(gpu_ctx, read.ColumnPages(col["ptype"], False, enc.pages[:enc.length].contiguous(),
                                                     enc.metas_array()))
    assert torch.equal(dec.values, dc.values)


@pytest.mark.parametrize("ptype", [S.T_I8, S.T_I32, S.T_I64, S.T_F64, S.T_I256])
def test_lz4_default_compression(gpu_ctx, ptype):
    """Basic(Lz4) pages: the device runs the same greedy parse as LZ4_compress_default, so the
    bytes equal the oracle's (which equal liblz4's, tests/test_oracle_blocks.py)."""
    col = gen.prim(ptype, 40_000, uniq=50, null_density=0.1, runs=4)
    check(gpu_ctx, col, max_page_size=8192, default_compression=S.LZ4)
    col = gen.prim(ptype, 3000, uniq=1 << 20)          # incompressible, small blocks (< 64 KiB: u16 table)
    check(gpu_ctx, col, max_page_size=1000, default_compression=S.LZ4)
    col = gen.prim(ptype, 70_000, uniq=3, runs=100)    # long matches, one block > 64 KiB (u32 table)
    check(gpu_ctx, col, default_compression=S.LZ4)


def test_lz4_boolean_and_binary(gpu_ctx):
    check(gpu_ctx, gen.boolean(50_003, null_density=0.2, runs=7), max_page_size=8192, default_compression=S.LZ4)
    check(gpu_ctx, gen.boolean(10_000, runs=3), max_page_size=1001, default_compression=S.LZ4)
    for large in (False, True):
        col = gen.binary(20_000, uniq=300, null_density=0.1, large=large, zipf=1.3)
        check(gpu_ctx, col, max_page_size=4096, default_compression=S.LZ4)


def test_lz4_nested_and_adaptive(gpu_ctx):
    col = gen.prim(S.T_F64, 30_000, uniq=100, runs=5)
    check(gpu_ctx, col, max_page_size=8192, default_compression=S.LZ4, force_codec=S.DICT)   # indices: Basic(Lz4)
    check(gpu_ctx, col, max_page_size=8192, default_compression=S.LZ4, force_codec=S.DICT, force_index_codec=S.LZ4)
    # adaptive with LZ4 as the fallback: incompressible pages fall back to Basic(Lz4)
    rnd = gen.prim(S.T_I64, 20_000, uniq=1 << 40)
    check(gpu_ctx, rnd, max_page_size=4096, default_compression=S.LZ4, ratio=2.0, forbidden=(S.FREQ, S.PATAS))
    check(gpu_ctx, col, max_page_size=4096, default_compression=S.LZ4, ratio=2.0, forbidden=(S.FREQ, S.PATAS))


def test_host_memory_boundary(gpu_ctx):
    """SB_MEM_HOST: host Arrow buffers in, host page bytes out, and back (the shape the reference's
    callers have); the library stages over PCIe itself."""
    import ctypes as C
    from strawboat_amd import _native as N
    from strawboat_amd.types import WriteOptions
    from strawboat_amd.write import options_c
    col = gen.prim(S.T_F64, 50_000, uniq=64, null_density=0.1, runs=9)
    want_pages, want_metas = gen.oracle_write(col, max_page_size=8192, force_codec=S.RLE)
    lib, h = gpu_ctx._lib, gpu_ctx._h
    oc = options_c(WriteOptions(max_page_size=8192, force_codec=S.RLE))
    vals = np.ascontiguousarray(col["values"]).view(np.uint8)
    npg = C.c_uint64()
    bound = lib.sb_write_bound(col["ptype"], 1, col["rows"], 0, C.byref(oc), C.byref(npg))
    out = np.zeros(bound, np.uint8)
    metas = (N.PageMetaC * npg.value)()
    cw = (N.ColumnWriteC * 1)()
    cw[0].physical_type, cw[0].is_nullable, cw[0].rows = col["ptype"], 1, col["rows"]
    cw[0].values, cw[0].validity = vals.ctypes.data, col["validity"].ctypes.data
    cw[0].out_pages, cw[0].out_capacity = out.ctypes.data, out.size
    cw[0].out_metas, cw[0].n_pages_capacity = metas, npg.value
    gpu_ctx._check(lib.sb_write_columns(h, cw, 1, C.byref(oc), N.SB_MEM_HOST))
    gpu_ctx.synchronize()
    assert cw[0].out_len == want_pages.size and np.array_equal(out[:cw[0].out_len], want_pages)
    # decode from host page bytes into host buffers
    want = gen.oracle_read(col, want_pages, want_metas)
    cr = (N.ColumnReadC * 1)()
    m = np.ascontiguousarray(want_metas)
    v_out = np.zeros(col["rows"] * 8, np.uint8)
    b_out = np.zeros((col["rows"] + 31) // 32 * 4, np.uint8)
    cr[0].physical_type, cr[0].is_nullable = col["ptype"], 1
    cr[0].pages, cr[0].pages_len = out.ctypes.data, int(cw[0].out_len)
    cr[0].metas, cr[0].n_pages = m.ctypes.data_as(C.POINTER(N.PageMetaC)), m.shape[0]
    cr[0].values, cr[0].values_capacity = v_out.ctypes.data, v_out.size
    cr[0].validity, cr[0].validity_capacity = b_out.ctypes.data, b_out.size
    gpu_ctx._check(lib.sb_read_columns(h, cr, 1, N.SB_MEM_HOST))
    gpu_ctx.synchronize()
    assert np.array_equal(v_out, want["values"]) and np.array_equal(b_out[:(col["rows"] + 7) // 8], want["validity"])


def test_host_memory_boundary_binary(gpu_ctx):
    """SB_MEM_HOST for a Utf8 column: host page bytes in, host offsets / values out; only `values_len` bytes of the
    values buffer and `out_len` bytes of a page buffer are written back"""
    import ctypes as C
    from strawboat_amd import _native as N
    from strawboat_amd.types import WriteOptions
    from strawboat_amd.write import options_c
    col = gen.binary(30_000, uniq=300, null_density=0.1, seed=3)
    want_pages, want_metas = gen.oracle_write(col, max_page_size=8192, ratio=2.0)
    want = gen.oracle_read(col, want_pages, want_metas)
    lib, h = gpu_ctx._lib, gpu_ctx._h
    # write: host Arrow buffers -> host page bytes
    oc = options_c(WriteOptions(max_page_size=8192, default_compress_ratio=2.0))
    npg = C.c_uint64()
    bound = lib.sb_write_bound(col["ptype"], 1, col["rows"], col["values"].size, C.byref(oc), C.byref(npg))
    out = np.full(bound, 0xA5, np.uint8)
    metas = (N.PageMetaC * npg.value)()
    cw = (N.ColumnWriteC * 1)()
    cw[0].physical_type, cw[0].is_nullable, cw[0].rows = col["ptype"], 1, col["rows"]
    cw[0].values, cw[0].values_len = col["values"].ctypes.data, col["values"].size
    cw[0].validity, cw[0].offsets = col["validity"].ctypes.data, col["offsets"].ctypes.data
    cw[0].out_pages, cw[0].out_capacity = out.ctypes.data, out.size
    cw[0].out_metas, cw[0].n_pages_capacity = metas, npg.value
    gpu_ctx._check(lib.sb_write_columns(h, cw, 1, C.byref(oc), N.SB_MEM_HOST))
    gpu_ctx.synchronize()
    assert cw[0].out_len == want_pages.size and np.array_equal(out[:cw[0].out_len], want_pages)
    assert (out[cw[0].out_len:] == 0xA5).all()
    # read back into host buffers with spare capacity
    m = np.ascontiguousarray(want_metas)
    cap = want["values"].size + 4096
    v_out = np.full(cap, 0x5A, np.uint8)
    o_out = np.zeros((col["rows"] + 1) * 4, np.uint8)
    b_out = np.zeros((col["rows"] + 31) // 32 * 4, np.uint8)
    cr = (N.ColumnReadC * 1)()
    cr[0].physical_type, cr[0].is_nullable = col["ptype"], 1
    cr[0].pages, cr[0].pages_len = out.ctypes.data, int(cw[0].out_len)
    cr[0].metas, cr[0].n_pages = m.ctypes.data_as(C.POINTER(N.PageMetaC)), m.shape[0]
    cr[0].values, cr[0].values_capacity = v_out.ctypes.data, v_out.size
    cr[0].validity, cr[0].validity_capacity = b_out.ctypes.data, b_out.size
    cr[0].offsets, cr[0].offsets_capacity = o_out.ctypes.data, o_out.size
    gpu_ctx._check(lib.sb_read_columns(h, cr, 1, N.SB_MEM_HOST))
    gpu_ctx.synchronize()
    assert cr[0].values_len == want["values"].size
    assert np.array_equal(v_out[:cr[0].values_len], want["values"]) and (v_out[cr[0].values_len:] == 0x5A).all()
    assert np.array_equal(o_out, want["offsets"]) and np.array_equal(b_out[:(col["rows"] + 7) // 8], want["validity"])


def test_columns_of_many_pages_in_one_call(gpu_ctx):
    """k_enc_layout places the pages of a column behind each other — a wave per column, 64 pages per step: columns of 1, 63,
    64, 65, 130 and 333 pages of mixed types in ONE call, adaptive (RLE / Dict / plain / LZ4 pages of very different sizes),
    PageMeta and bytes against the oracle's (write::write walks the pages one after the other, src/write/serialize.rs:36-49)"""
    from strawboat_amd import write, WriteOptions
    specs = [(1, S.T_I64), (63, S.T_F64), (64, S.T_I32), (65, None), (130, S.T_U16), (333, S.T_I64), (70, "bool")]
    cols = []
    for k, (npages, what) in enumerate(specs):
        rows = npages * 500 - 3
        if what is None:
            cols.append(gen.binary(rows, uniq=60, null_density=0.1, maxlen=20, seed=50 + k))
        elif what == "bool":
            cols.append(gen.boolean(rows, null_density=0.1, runs=7, seed=50 + k))
        else:
            cols.append(gen.prim(what, rows, uniq=[3, 1 << 30, 200][k % 3], runs=[9, 1, 2][k % 3], null_density=0.2 if k % 2 else None, seed=50 + k))
    for opt in (dict(max_page_size=500, ratio=2.0), dict(max_page_size=500, default_compression=S.LZ4)):
        wo = WriteOptions(default_compression=opt.get("default_compression", 0), default_compress_ratio=opt.get("ratio"),
                          max_page_size=500, lz4_exact=True)
        encs = write.encode_columns(gpu_ctx, [to_device_column(gpu_ctx, c) for c in cols], wo)
        gpu_ctx.synchronize()
        for (npages, _), c, e in zip(specs, cols, encs):
            want_pages, want_metas = gen.oracle_write(c, **opt)
            assert want_metas.shape[0] == npages
            assert np.array_equal(e.metas_array(), want_metas)
            assert np.array_equal(e.pages_numpy(), want_pages)
