"""Host-side logic that needs no GPU: sharding plan, ColumnMeta offsets, option mapping."""
import numpy as np

from strawboat_amd import shard
from strawboat_amd.types import Compression, WriteOptions
from strawboat_amd.write import options_c


def test_plan_shards_balances_by_bytes():
    # C4: 8 columns, Utf8 ~3x heavier than Boolean
    sizes = [40, 40, 80, 80, 240, 240, 2, 2]
    shards = shard.plan_shards(sizes, 8)
    assert sorted(sum(shards, [])) == list(range(8)) and all(len(s) == 1 for s in shards)
    shards = shard.plan_shards(sizes, 2)
    loads = [sum(sizes[i] for i in s) for s in shards]
    assert abs(loads[0] - loads[1]) <= 4
    assert shard.plan_shards([5, 5, 5], 4)[3] == []


def test_column_metas_offsets():
    metas = [np.array([[100, 10], [50, 5]], np.uint64), np.array([[7, 1]], np.uint64), np.zeros((0, 2), np.uint64)]
    cm = shard.column_metas(metas)
    assert [c.offset for c in cm] == [8, 158, 165]
    assert cm[0].pages[1].num_values == 5 and cm[0].total_len() == 150


def test_gather_metas_single_process():
    local = {0: np.array([[10, 1]], np.uint64), 1: np.array([[20, 2], [30, 3]], np.uint64)}
    out = shard.gather_metas(local, 2)
    assert out[1].tolist() == [[20, 2], [30, 3]]


def test_gather_metas_refuses_gaps_and_duplicates():
    """(column, page range) work items: every page of every column exactly once, or an error — never a shorter column
    whose later ColumnMeta offsets are silently wrong"""
    import pytest
    m = lambda n: np.array([[10 + k, 1] for k in range(n)], np.uint64)
    ok = shard.gather_metas([(0, 0, m(2)), (0, 2, m(1)), (1, 0, m(1))], 2, expected_pages=[3, 1])
    assert [len(x) for x in ok] == [3, 1]
    with pytest.raises(ValueError, match="missing"):
        shard.gather_metas([(0, 0, m(2)), (0, 3, m(1))], 1)                     # page 2 never arrived
    with pytest.raises(ValueError, match="twice"):
        shard.gather_metas([(0, 0, m(2)), (0, 1, m(2))], 1)                     # page 1 dealt twice
    with pytest.raises(ValueError, match="of its 4 pages"):
        shard.gather_metas([(0, 0, m(3))], 1, expected_pages=[4])               # the rank owning the tail sent nothing
    with pytest.raises(ValueError, match="columns"):
        shard.gather_metas([(5, 0, m(1))], 2)


def test_options_mapping():
    o = options_c(WriteOptions(default_compression=Compression.LZ4, default_compress_ratio=2.0, max_page_size=8192,
                               forbidden_compressions=[Compression.FREQ, Compression.PATAS], force_codec=10, rng_seed=7))
    assert o.default_compression == 1 and o.has_default_compress_ratio == 1 and o.default_compress_ratio == 2.0
    assert o.max_page_size == 8192 and o.forbidden_compressions == (1 << 13) | (1 << 16)
    assert o.force_codec == 10 and o.force_index_codec == -1 and o.rng_seed == 7
    o = options_c(WriteOptions())
    assert o.has_default_compress_ratio == 0 and o.max_page_size == 0


def test_c5_page_range_slices_are_columns_of_their_own():
    """workloads.c5_slice (what a rank holds when it owns a page range of a C5 leaf column, SURVEY 8e work items): the level
    sections and leaf ranges the oracle writes for the pages of the slice equal those of the same pages of the whole column"""
    import workloads as W
    from oracle import sbo as S
    la, a, lb, b = W.c5_nested(rows=20_000, seed=7)
    page = 4096
    for levels, leaf in ((la, a), (lb, b)):
        for r0, r1 in ((0, 8192), (8192, 16384), (16384, 20_000), (4096, 20_000)):
            lv, col = W.c5_slice(levels, leaf, r0, r1)
            assert col["rows"] == lv[-1]["length"] == lv[1]["length"]
            whole_leaf0 = None
            for q0 in range(r0, r1, page):
                ln = min(page, r1 - q0)
                want, nv, ls, lc = S.nested_write_levels(levels, q0, ln)
                got, nv2, ls2, lc2 = S.nested_write_levels(lv, q0 - r0, ln)
                assert np.array_equal(got, want) and nv == nv2 and lc == lc2
                if whole_leaf0 is None:
                    whole_leaf0 = ls - ls2          # the slice's leaf rows are the whole column's, shifted by a constant
                assert ls - ls2 == whole_leaf0
            e0 = whole_leaf0
            if leaf["offsets"] is None:
                assert np.array_equal(col["values"], np.asarray(leaf["values"])[e0:e0 + col["rows"]])
            else:
                bo = np.asarray(leaf["offsets"])
                assert np.array_equal(col["offsets"], bo[e0:e0 + col["rows"] + 1] - bo[e0])
                assert np.array_equal(col["values"], leaf["values"][int(bo[e0]):int(bo[e0 + col["rows"]])])


def _canned_configs():
    """the shape of bench.run_configs' result with long kernel names and every leg present (worst case for the line length)"""
    def direction(g, f):
        return {"GBps": g, "ms": 1.234, "frac_hbm": f, "top_kernel": "k_enc_emit_pages<-4, 11>", "top_kernel_ms": 0.92, "top_kernel_share": 0.3,
                "kernels_ms": {"void sb::k_very_long_kernel_name_number_%d<(int)-4, (int)11>" % i: 0.123 for i in range(8)}}

    def cpu():
        return {"value": 0.56, "unit": "GB/s", "cores": 1, "kind": "port", "sample": "s" * 300,
                "one_thread": {"value": 0.56, "encode": 0.4, "decode": 1.0},
                "all_cores": {"value": 123.4, "encode": 100.0, "decode": 150.0, "cores": 256, "sample_replicas": 64},
                "cpu_model": "AMD EPYC 9575F 64-Core Processor", "block_codecs": {"lz4": "x" * 80, "zstd": "y" * 80}, "note": "n" * 200}

    def entry(g1, g2, **kw):
        e = {"arrow_MB": 1234.5, "page_MB": 123.4, "pages": 1024, "encdec_GBps": 456.7, "encode": direction(g1, 0.0531),
             "decode": direction(g2, 0.0971), "cpu_baseline": cpu(), "workload": "w" * 250}
        e.update(kw)
        return e

    ref = {"decode": direction(95.9, 0.02), "written_by": "libzstd 1.4.8"}
    one = {nm: entry(0.6, 158.0) for nm in ("int64_adaptive", "int64_runs_adaptive", "int64_zstd", "int64_lz4", "utf8_zstd", "utf8_adaptive",
                                            "int32_lowcard_adaptive", "int32_sparse_adaptive", "utf8_lz4", "bool_adaptive")}
    return {"c1": entry(2408.0, 2262.3), "c3": entry(354.0, 652.0), "c3_lz4": entry(83.4, 149.4), "c3_lz4_reference_written": ref,
            "c4": entry(298.0, 658.0, per_column_type={t: {"encode_GBps": 1.0, "decode_GBps": 2.0, "codecs": ["Dict>Bitpacking"]} for t in "abcd"}),
            "one_page": one,
            "host_boundary": {"c2": {"encode_GBps": 44.3, "decode_GBps": 44.5}, "c1": {"encode_GBps": 26.5, "decode_GBps": 26.2},
                              "cpu_1t_GBps": {"c2": 1.5, "c1": 0.09}, "note": "n" * 300, "pcie_peak_GBps": 63.0},
            "c5": entry(158.0, 205.0, leaf_pages_reference_written=ref, single_array={"encode_ms": 0.74, "decode_ms": 0.93}),
            "continuity": {nm: entry(676.0, 1612.3, single_column_latency={"2^%d" % p: {"write_ms": 0.024, "read_ms": 0.105} for p in range(10, 21, 2)})
                           for nm in ("bool", "utf8", "i64")}}


def test_bench_line_is_short_and_parses():
    """the driver parses ONE short stdout line; round 4's line (40 KB) came back `parsed: null`.  The line
    must stay under 8 KB whatever the configurations carry, round-trip through json, keep every string under the driver's
    128-character cut, and hold the keys a reader needs: roofline, cpu_baseline, north_star_decode, config.workload and one
    string per configuration (C1-C5, one_page_*, host_boundary, continuity_*)."""
    import json
    import bench
    head = {"metric": bench.METRIC, "value": 3625.44, "unit": "GB/s", "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": 2.2949,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64 bit patterns (integer/bit work, no arithmetic)",
            "data": "synthetic"}
    cfg = {"workload": "C2: 512 x 1M-row nullable Float64 columns per GPU, 64Ki-row pages, codec adaptive, inputs resident in HBM",
           "columns_per_gpu": 512, "rows_per_column": 1000000, "page_rows": 65536, "arrow_bytes_per_step": 4160000000,
           "page_bytes_per_step": 254766424, "parallelism": "pages of independent columns sharded across 1 GPU(s)", "note": "n" * 100}
    roof = {"bound": "hbm", "kernel": "k_enc_select_runs<8, 2>", "achieved": 3719.9, "peak": 8000.0, "unit": "GB/s", "frac": 0.465,
            "traffic": 4861421764, "avg_kernel_ms": 1.1868, "algorithmic_bytes_per_launch": 4414766424, "traffic_source": "t" * 100}
    north = {"config": "C1", "target_frac": 0.40, "frac_end_to_end": 0.566, "decode_ms": 0.4526, "frac_kernel": 0.69, "kernel": "k_expand", "kernel_ms": 0.368}
    configs = _canned_configs()
    out = bench.assemble_line(head, cfg, roof, north, configs["c1"]["cpu_baseline"], configs)
    line = bench.bench_line(out)
    assert "\n" not in line and len(line) < 8192, len(line)
    back = json.loads(line)
    assert back["roofline"]["frac"] == 0.465 and back["cpu_baseline"]["cores"] == 1 and back["cpu_baseline"]["kind"] == "port"
    assert back["north_star_decode"]["frac_end_to_end"] == 0.566
    assert back["config"]["workload"].startswith("C2")
    assert back["ms_per_step"] == 2.2949 and back["steps"] == 20 and back["warmup"] == 5
    c = back["config"]
    for k in ("c1", "c3", "c3_lz4", "c3_lz4_reference_written", "c4", "c5", "host_boundary", "north_star_decode",
              "one_page_int64_lz4", "one_page_utf8_adaptive", "continuity_bool", "continuity_i64"):
        assert isinstance(c[k], str) and 0 < len(c[k]) <= 128, (k, c.get(k))
    assert "enc 354" in c["c3"] and "dec 652" in c["c3"] and "CPU 1t" in c["c3"]
    assert "kernels" not in back and "configs" not in back and "summary" not in back

    def strings(o):
        if isinstance(o, dict):
            for v in o.values():
                yield from strings(v)
        elif isinstance(o, str):
            yield o
    assert max(len(s) for s in strings(back)) <= 128
    # a pathological case (hundreds of entries) sheds text instead of growing past the limit
    many = {"cfg%03d" % i: configs["c3"] for i in range(150)}
    line = bench.bench_line(bench.assemble_line(head, cfg, roof, north, None, many))
    assert len(line) < 8192 and json.loads(line)["roofline"]["frac"] == 0.465


def test_kernel_source_sha_covers_every_kernel_source():
    import os
    import bench
    d = os.path.join(os.path.dirname(bench.__file__), "strawboat_amd", "csrc")
    srcs = [f for f in os.listdir(d) if f.endswith((".hip", ".h", ".cpp"))]
    assert {"sb_decode.hip", "sb_encode.hip", "sb_zstd_blocks.h", "sb_lz4_big.h", "sb_nested.hip"} <= set(srcs)
    a = bench.kernel_source_sha()
    assert len(a) == 16 and a == bench.kernel_source_sha()
