"""BASELINE.json's configurations at their full per-GPU sizes (SURVEY.md §8d inputs): the pages the
device writes equal the oracle's byte for byte (same options, same sampling seed), and the device
decodes them back to the Arrow buffers they came from.

  C1  1 M-row non-nullable Int64, one page, no compression
  C3  1 M-row Utf8, zipf(1.1) over 10 000 words of length 4..24, LZ4 default, ratio 2.0 -> Dict pages
  C4  the per-GPU shard of the mixed schema: Int32 / Float64 / Utf8 / Boolean columns, 64 Ki-row pages,
      LZ4 default, ratio 2.0 (1 M rows per column, and the 10 M rows BASELINE.json states: 153 pages, 38 528-row tail)
  C5  1 M-row List<Struct<Int64, Utf8>>, Zstd default, whole-file round trip
(C2 is tests/test_gpu_select.py::test_c2_adaptive and bench.py itself.)"""
import numpy as np
import pytest

from oracle import sbo as S
from tests import gen
from tests.test_gpu_decode import gpu_decode
import workloads
from tests.test_gpu_encode import gpu_encode

pytestmark = pytest.mark.gpu

ROWS = 1_000_000
NOT_ON_DEVICE = ()  # every codec has a device encoder now: the reference's default options (nothing forbidden)


def encode_matches_oracle_and_round_trips(ctx, col, **opt):
    want_pages, want_metas = gen.oracle_write(col, **opt)
    enc = gpu_encode(ctx, col, **opt)
    assert np.array_equal(enc.metas_array(), want_metas)
    assert np.array_equal(enc.pages_numpy(), want_pages), "page bytes differ from the oracle's"
    got = gpu_decode(ctx, col, want_pages, want_metas)
    want = gen.oracle_read(col, want_pages, want_metas)
    assert got.rows == col["rows"]
    assert np.array_equal(got.values_numpy(), want["values"])
    if col["nullable"]:
        assert np.array_equal(got.validity_numpy(), want["validity"])
    if col["offsets"] is not None:
        assert np.array_equal(got.offsets_numpy(), want["offsets"])
    return S.stat_column(col["ptype"], col["nullable"], want_pages, want_metas)[0]


def _chain(info):
    """(codec, uncompressed_size, unique_num) of every block of a page, outermost first; the compressed sizes are left
    out (they are what a different LZ4 / Zstd parse changes), and so is PageInfo.validity_size, which upstream reads from
    the 4 bytes BEHIND the def-level section, i.e. from the hdr9 (src/stat.rs:72-77)"""
    out = []
    while info is not None:
        out.append((info.codec, info.uncompressed_size, info.body.unique_num))
        info = info.body.indices or info.body.exceptions
    return out


def default_encoder_parity(ctx, col, ratio_bound, **opt):
    """The encoder bench.py times (WriteOptions without lz4_exact: parallel LZ4 matcher, chunk + stitch kernels, Zstd
    frames per piece) under the config's own options.  Its LZ4 / Zstd bytes are format-valid streams of another parse
    (BASELINE.md section 6), so instead of byte equality: (1) page structure equals the oracle's — page count, num_values,
    the codec chain of every page with every hdr9 uncompressed_size / unique_num / def-level size; (2) the ORACLE decodes
    the device's pages to what it decodes from its own; (3) the device decodes both; (4) page bytes stay within
    `ratio_bound` x the oracle's (whose LZ4 is byte-identical to liblz4, tests/test_oracle_blocks.py).
    Follows compress_buffer / decompress_buffer, src/compression/basic.rs:87-135."""
    from strawboat_amd import stat
    opt = {k: v for k, v in opt.items() if k != "forbidden" or v}
    want_pages, want_metas = gen.oracle_write(col, **opt)
    enc = gpu_encode(ctx, col, lz4_exact=False, **opt)
    got_pages, got_metas = enc.pages_numpy(), enc.metas_array()
    assert got_metas.shape == want_metas.shape, "page count"
    assert np.array_equal(got_metas[:, 1], want_metas[:, 1]), "num_values per page"
    gi = stat.stat_simple(got_pages, got_metas, col["ptype"], col["nullable"])
    wi = stat.stat_simple(want_pages, want_metas, col["ptype"], col["nullable"])
    go = wo = 0
    for k, (g, w) in enumerate(zip(gi.pages, wi.pages)):
        assert _chain(g) == _chain(w), "page %d: block structure %s vs the oracle's %s" % (k, _chain(g), _chain(w))
        if col["nullable"]:   # the def-level section (u32 length + hybrid-RLE bits) byte for byte
            dl = 4 + int(want_pages[wo:wo + 4].view(np.uint32)[0])
            assert np.array_equal(got_pages[go:go + dl], want_pages[wo:wo + dl]), "page %d: def-level section" % k
        go += int(got_metas[k, 0])
        wo += int(want_metas[k, 0])
    want = gen.oracle_read(col, want_pages, want_metas)
    back = gen.oracle_read(col, got_pages, got_metas)           # the CPU reader reads what the device wrote
    for key in ("values", "validity", "offsets"):
        if want.get(key) is not None:
            assert np.array_equal(back[key], want[key]), "oracle decode of the device's pages: %s" % key
    for pages, metas in ((got_pages, got_metas), (want_pages, want_metas)):
        got = gpu_decode(ctx, col, pages, metas)
        assert got.rows == col["rows"]
        assert np.array_equal(got.values_numpy(), want["values"])
        if col["nullable"]:
            assert np.array_equal(got.validity_numpy(), want["validity"])
        if col["offsets"] is not None:
            assert np.array_equal(got.offsets_numpy(), want["offsets"])
    r = got_pages.size / max(want_pages.size, 1)
    print("default encoder: %d page bytes vs %d (oracle / liblz4 parse) = %.3f x" % (got_pages.size, want_pages.size, r))
    assert r <= ratio_bound, "page bytes %.3f x the oracle's (bound %.2f)" % (r, ratio_bound)
    return r


def zipf_utf8(rows, seed, null_density=None):
    return workloads.zipf_utf8(rows, seed, null_density)


def test_c1_int64_one_page_no_compression(gpu_ctx):
    rng = np.random.default_rng(42)
    col = dict(ptype=S.T_I64, nullable=False, rows=ROWS, values=rng.integers(0, 2**63 - 1, ROWS), validity=None, offsets=None)
    codecs = encode_matches_oracle_and_round_trips(gpu_ctx, col)
    assert codecs.tolist() == [S.NONE]


def test_c3_utf8_zipf_dict_lz4(gpu_ctx):
    col = zipf_utf8(ROWS, 42)
    codecs = encode_matches_oracle_and_round_trips(gpu_ctx, col, max_page_size=65536, default_compression=S.LZ4, ratio=2.0,
                                                   forbidden=NOT_ON_DEVICE)
    assert (codecs == S.DICT).all()
    # and the forced Basic(LZ4) variant: offsets block + values block
    encode_matches_oracle_and_round_trips(gpu_ctx, zipf_utf8(200_000, 7), max_page_size=65536, default_compression=S.LZ4)


def test_c3_default_encoder(gpu_ctx):
    """C3 and C3' at 1 M rows with the DEFAULT encoder (the timed path): Dict pages whose short last page carries
    LZ4-coded indices, and Basic(LZ4) pages with 0.94 MB value blocks through the chunk + stitch kernels"""
    col = zipf_utf8(ROWS, 42)
    default_encoder_parity(gpu_ctx, col, 1.02, max_page_size=65536, default_compression=S.LZ4, ratio=2.0)
    default_encoder_parity(gpu_ctx, col, 1.35, max_page_size=65536, default_compression=S.LZ4)


@pytest.mark.parametrize("name", ["int32_0", "int32_1", "float64_0", "float64_1", "utf8_0", "utf8_1", "boolean_0", "boolean_1"])
def test_c4_default_encoder_at_the_size_baseline_states(gpu_ctx, name):
    """every C4 column type at 10 M rows with the default encoder (Boolean pages are LZ4 blocks of incompressible bitmaps)"""
    col = dict(workloads.c4_columns(10_000_000))[name]
    default_encoder_parity(gpu_ctx, col, 1.05, max_page_size=65536, default_compression=S.LZ4, ratio=2.0)


def test_c5_leaves_default_encoder(gpu_ctx):
    """C5's two leaf columns (Int64 with 20 % nulls, Utf8) as flat non-nullable columns, Zstd default, 1 M top-level rows'
    worth of leaves: the Zstd frames the device writes (one per 16 KiB piece) are read by the oracle's decoder; size
    against the oracle's STORE-ONLY frames must be below 1 (the device compresses)"""
    la, a, lb, b = workloads.c5_nested()
    for leaf in (a, b):
        col = dict(leaf, nullable=False, validity=None)
        r = default_encoder_parity(gpu_ctx, col, 1.0, max_page_size=65536, default_compression=S.ZSTD)
        assert r < 1.0


@pytest.mark.parametrize("kind", ["int32", "float64", "utf8", "boolean"])
def test_c4_mixed_schema_shard(gpu_ctx, kind):
    import bench
    rng = np.random.default_rng(42)
    if kind == "int32":
        col = dict(ptype=S.T_I32, nullable=True, rows=ROWS, values=rng.integers(0, 1000, ROWS).astype(np.int32),
                   validity=None, offsets=None)
    elif kind == "float64":
        vals, valid = bench.gen_c2_column(43)
        col = dict(ptype=S.T_F64, nullable=True, rows=vals.size, values=vals, validity=valid, offsets=None)
    elif kind == "utf8":
        col = zipf_utf8(ROWS, 44, null_density=0.1)
    else:
        col = gen.boolean(ROWS, null_density=0.1, seed=45)
    encode_matches_oracle_and_round_trips(gpu_ctx, col, max_page_size=65536, default_compression=S.LZ4, ratio=2.0,
                                          forbidden=NOT_ON_DEVICE)


@pytest.mark.parametrize("name", ["int32_0", "int32_1", "float64_0", "float64_1", "utf8_0", "utf8_1", "boolean_0", "boolean_1"])
def test_c4_at_the_size_baseline_states(gpu_ctx, name):
    """C4 as BASELINE.json states it: 10 M rows per column = 153 pages, the last one of 38 528 rows (= 301 x 128: its
    Dict indices bit-pack, unlike the 16 960-row tail of a 1 M-row column) — the columns bench.py's `c4` entry times,
    byte for byte against the oracle."""
    col = dict(workloads.c4_columns(10_000_000))[name]
    codecs = encode_matches_oracle_and_round_trips(gpu_ctx, col, max_page_size=65536, default_compression=S.LZ4, ratio=2.0)
    assert len(codecs) == 153


def test_c5_nested_list_struct_zstd_file(gpu_ctx, tmp_path):
    import pyarrow as pa
    from strawboat_amd import WriteOptions, file as F
    from strawboat_amd.types import Compression as C
    rng = np.random.default_rng(42)
    list_valid = rng.random(ROWS) > 0.1
    lens = np.where(list_valid, rng.integers(0, 3, ROWS), 0)            # Uniform{0,1,2}, null lists are empty
    offs = np.zeros(ROWS + 1, np.int32)
    np.cumsum(lens, out=offs[1:])
    n = int(offs[-1])
    a = pa.array(rng.integers(-2**40, 2**40, n), mask=rng.random(n) < 0.2)
    words = np.array(["s%d" % k for k in range(500)], dtype=object)
    b = pa.array(words[rng.integers(0, 500, n)], mask=rng.random(n) < 0.2)
    st = pa.StructArray.from_arrays([a, b], fields=[pa.field("a", pa.int64()), pa.field("b", pa.string())])
    col = pa.ListArray.from_arrays(pa.array(offs), st, mask=pa.array(~list_valid))
    t = pa.Table.from_arrays([col], schema=pa.schema([pa.field("ls", col.type)]))
    path = tmp_path / "c5.sb"
    with F.NativeWriter(gpu_ctx, path, t.schema, WriteOptions(max_page_size=65536, default_compression=C.ZSTD)) as w:
        w.start()
        w.write(t)
        w.finish()
        assert [len(m.pages) for m in w.metas] == [16, 16]          # 2 leaf columns x 16 pages
    got = F.read_table(gpu_ctx, path)
    assert got.column("ls").combine_chunks().equals(t.column("ls").combine_chunks())


def test_c5_level_sections_match_oracle_full_size(gpu_ctx):
    """C5 at the size BASELINE.json states (1 M top-level rows, 64 Ki-row pages, ~1.1 M level entries per leaf): the rep / def
    level section of every page of both leaf columns, byte for byte against the oracle's write_nested_validity
    (src/write/serialize.rs:217-232) — a round trip alone would accept a section that is self-consistently wrong"""
    import torch
    from strawboat_amd import nested
    la, a, lb, b = workloads.c5_nested()
    rows = la[0]["length"]

    def up(x):
        return None if x is None else torch.from_numpy(np.ascontiguousarray(x).view(np.uint8).reshape(-1)).to(gpu_ctx.torch_device)
    for levels in (la, lb):
        dl = [nested.NestedLevel(lv["kind"], bool(lv["is_optional"]), lv["length"], up(lv.get("validity")), up(lv.get("offsets"))) for lv in levels]
        got = nested.write_levels(gpu_ctx, dl, rows, 65536)
        sec = got.sections.cpu().numpy()
        assert got.n_pages == 16
        off = 0
        for p, r0 in enumerate(range(0, rows, 65536)):
            ln = min(65536, rows - r0)
            want, nv, ls, lc = S.nested_write_levels(levels, r0, ln)
            assert (int(got.num_values[p]), int(got.leaf_start[p]), int(got.leaf_count[p])) == (nv, ls, lc), p
            assert int(got.level_bytes[p]) == len(want), p
            assert bytes(sec[off:off + len(want)]) == bytes(want), "level section of page %d differs" % p
            off += len(want)


def test_c4_all_columns_in_one_call(gpu_ctx):
    """C4's eight columns in ONE adaptive sb_write_columns call and ONE sb_read_columns call — the shape bench.py's `c4`
    entry times.  A call that mixes binary and primitive kinds runs the binary chain (k_enc_bin_hash -> selector ->
    k_enc_bin_verify -> page emitters) on the high-priority side stream next to the primitive kinds, joined by events
    (sb_encode.hip `multi`; the reader likewise, sb_decode.hip): the scratch areas a page's kernels hand to each other
    cross kernel boundaries on ANOTHER stream there (the memory-order contract at the top of csrc/sb_common.h).  Every
    page must still be the oracle's, byte for byte, on repeated calls (the plan cache reuses the device page table), and
    sb_ctx_side_forks must show that the multi-stream path is the one that ran.  Primitive kinds alone alternate between
    the call's stream and a second side stream."""
    from strawboat_amd import read, write, WriteOptions
    from tests.test_gpu_encode import to_device_column
    named = workloads.c4_columns(ROWS)
    opt = dict(max_page_size=65536, default_compression=S.LZ4, ratio=2.0)
    want = {n: gen.oracle_write(c, **opt) for n, c in named}
    wo = WriteOptions(default_compression=S.LZ4, default_compress_ratio=2.0, max_page_size=65536, lz4_exact=True)
    for names in ([n for n, _ in named],                                       # binary + primitives + boolean
                  ["int32_0", "float64_0", "int32_1", "float64_1"],            # two primitive kinds
                  ["utf8_1", "boolean_0", "int32_1"]):
        cols = [dict(named)[n] for n in names]
        dev = [to_device_column(gpu_ctx, c) for c in cols]
        for rep in range(3):
            before = gpu_ctx.side_forks()
            encs = write.encode_columns(gpu_ctx, dev, wo)
            gpu_ctx.synchronize()
            assert gpu_ctx.side_forks() > before, "the call did not use the side streams"
            for n, e in zip(names, encs):
                assert np.array_equal(e.metas_array(), want[n][1]), "%s (call %d): PageMeta" % (n, rep)
                assert np.array_equal(e.pages_numpy(), want[n][0]), "%s (call %d): page bytes differ from the oracle's" % (n, rep)
        cps = [read.ColumnPages(c["ptype"], c["nullable"], e.pages[:e.length], e.metas_array()) for c, e in zip(cols, encs)]
        for rep in range(2):
            before = gpu_ctx.side_forks()
            got = read.batch_read_columns(gpu_ctx, cps)
            gpu_ctx.synchronize()
            if any(c["offsets"] is not None for c in cols):
                assert gpu_ctx.side_forks() > before, "the read did not use the side streams"
            for n, c, g in zip(names, cols, got):
                back = gen.oracle_read(c, *want[n])
                assert np.array_equal(g.values_numpy(), back["values"]), n
                if c["nullable"]:
                    assert np.array_equal(g.validity_numpy(), back["validity"]), n
                if c["offsets"] is not None:
                    assert np.array_equal(g.offsets_numpy(), back["offsets"]), n
