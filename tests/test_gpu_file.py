"""Whole files on the GPU path: pyarrow Table -> NativeWriter (pages encoded on the device) -> file
-> read_meta / infer_schema / read_table (pages decoded on the device) -> the same Table.  The column
shapes follow the reference's round-trip tests (tests/it/io.rs:72-278)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def make_table(rows, seed=1):
    import pyarrow as pa
    rng = np.random.default_rng(seed)

    def nullable(a, p=0.2):
        return pa.array(a, mask=rng.random(len(a)) < p)
    runs = np.repeat(rng.integers(0, 50, rows // 8 + 1), 8)[:rows]
    words = np.array(["w%d" % k + "x" * int(k % 7) for k in range(200)], dtype=object)
    lists = [None if rng.random() < 0.1 else [int(x) if rng.random() > 0.2 else None for x in rng.integers(0, 9, rng.integers(0, 3))]
             for _ in range(rows)]
    ls = [None if rng.random() < 0.1 else [dict(a=int(x), b=str(words[x % 200]) if x % 5 else None) for x in rng.integers(0, 1000, rng.integers(0, 3))]
          for _ in range(rows)]
    st = [dict(x=float(i % 13), y=int(i % 100) if i % 7 else None) for i in range(rows)]
    cols = dict(
        i64=nullable(rng.integers(-2**40, 2**40, rows)),
        f64=nullable(runs.astype(np.float64), 0.1),
        u32=pa.array(np.sort(rng.integers(0, 1000, rows)).astype(np.uint32)),
        s=nullable(words[rng.integers(0, 200, rows)]),
        b=nullable(rng.random(rows) < 0.5),
        li=pa.array(lists, type=pa.list_(pa.int32())),
        ls=pa.array(ls, type=pa.list_(pa.struct([("a", pa.int64()), ("b", pa.string())]))),
        st=pa.array(st, type=pa.struct([pa.field("x", pa.float32(), nullable=False), ("y", pa.int16())])),
        dec=pa.array([None if i % 11 == 0 else i * 1001 - 5000 for i in range(rows)], type=pa.decimal128(20, 3)),
        n=pa.nulls(rows),
    )
    fields = [pa.field(k, v.type, nullable=(k not in ("u32", "st"))) for k, v in cols.items()]
    return pa.Table.from_arrays(list(cols.values()), schema=pa.schema(fields))


@pytest.mark.parametrize("opt", ["none", "adaptive", "lz4", "zstd_adaptive"])
def test_table_round_trip(gpu_ctx, tmp_path, opt):
    from strawboat_amd import WriteOptions, file as F
    from strawboat_amd.types import Compression as C
    t = make_table(5000)
    wo = dict(none=WriteOptions(max_page_size=1000),
              adaptive=WriteOptions(max_page_size=1024, default_compress_ratio=2.0),
              lz4=WriteOptions(max_page_size=2048, default_compression=C.LZ4),
              zstd_adaptive=WriteOptions(max_page_size=4096, default_compression=C.ZSTD, default_compress_ratio=1.5))[opt]
    path = tmp_path / "t.sb"
    with F.NativeWriter(gpu_ctx, path, t.schema, wo) as w:
        w.start()
        w.write(t)
        w.finish()
        metas = w.metas
        assert w.total_size == path.stat().st_size
    assert F.read_meta(path) == metas
    assert len(metas) == 12  # leaves: i64 f64 u32 s b li ls.a ls.b st.x st.y dec n
    assert F.infer_schema(path).equals(t.schema)
    got = F.read_table(gpu_ctx, path)
    assert got.schema.equals(t.schema)
    for name in t.column_names:
        assert got.column(name).combine_chunks().equals(t.column(name).combine_chunks()), name


def test_writer_refuses_a_second_row_group(gpu_ctx, tmp_path):
    from strawboat_amd import WriteOptions, file as F
    from strawboat_amd._native import NativeError
    import pyarrow as pa
    t = pa.table({"a": pa.array([1, 2, 3], pa.int32())})
    with F.NativeWriter(gpu_ctx, tmp_path / "x.sb", t.schema, WriteOptions()) as w:
        with pytest.raises(NativeError, match="must be started"):
            w.write(t)
        w.start()
        w.write(t)
        w.finish()
        with pytest.raises(NativeError, match="one RowGroup"):
            w.write(t)
