"""The reference's own round-trip tests (tests/it/io.rs:72-278), shape by shape, on the GPU path: a chunk
is written with NativeWriter for each default compression (LZ4 / Zstd / Snappy / None) with
default_compress_ratio = Some(2.0), max_page_size = 2048 and nothing forbidden (io.rs:417-438), read back
with read_table and compared.  Generators follow io.rs:281-416 (create_random_* / create_list / create_map /
create_struct); the RNG is numpy's, not StdRng."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

WRITE_PAGE = 2048


def _pa():
    import pyarrow as pa
    return pa


def random_bool(size, null_density, seed=42):
    pa, rng = _pa(), np.random.default_rng(seed)
    return pa.array(rng.random(size) < 0.5, mask=(rng.random(size) <= null_density) if null_density else None)


def random_index(size, null_density, uniq, seed=42):
    pa, rng = _pa(), np.random.default_rng(seed)
    return pa.array(rng.integers(0, max(uniq, 1), size).astype(np.int32), mask=(rng.random(size) <= null_density) if null_density else None)


def random_double(size, null_density, uniq, seed=42):
    pa, rng = _pa(), np.random.default_rng(seed)
    return pa.array(rng.integers(0, max(uniq, 1), size).astype(np.float64), mask=(rng.random(size) <= null_density) if null_density else None)


def random_string(size, null_density, uniq, seed=42):
    pa, rng = _pa(), np.random.default_rng(seed)
    v = np.array([str(x).encode() for x in rng.integers(0, max(uniq, 1), size)], dtype=object)
    return pa.array(v, type=pa.large_binary(), mask=(rng.random(size) <= null_density) if null_density else None)


def random_offsets(size, null_density, seed=42):
    rng = np.random.default_rng(seed)
    valid = rng.random(size) > null_density
    lens = np.where(valid, rng.integers(0, 3, size), 0)
    offs = np.zeros(size + 1, np.int32)
    np.cumsum(lens, out=offs[1:])
    return offs, valid


def create_list(size, null_density):
    pa = _pa()
    offs, valid = random_offsets(size, 0.1)
    length = int(offs[-1])
    return pa.ListArray.from_arrays(pa.array(offs), random_index(length, null_density, length), mask=pa.array(~valid))


def create_struct(size, null_density, uniq):
    pa = _pa()
    return pa.StructArray.from_arrays([random_string(size, null_density, uniq), random_index(size, null_density, uniq)],
                                      fields=[pa.field("name", pa.large_binary()), pa.field("age", pa.int32())])


def create_map(size, null_density):
    pa = _pa()
    offs, valid = random_offsets(size, 0.1)
    length = int(offs[-1])
    t = pa.map_(pa.int32(), pa.large_binary())
    entries = pa.StructArray.from_arrays([random_index(length, 0.0, length), random_string(length, null_density, length)],
                                         fields=[t.key_field, t.item_field])
    vb = pa.py_buffer(np.packbits(valid, bitorder="little").tobytes())
    return pa.Array.from_buffers(t, size, [vb, pa.py_buffer(offs.tobytes())], children=[entries])


def list_of_pairs(child):
    """io.rs:196-254: a non-nullable list whose offsets are 0, 2, 4, ... over a 2000-element child"""
    pa = _pa()
    offs = np.arange(0, 1001, 2, dtype=np.int32)
    return pa.ListArray.from_arrays(pa.array(offs), child.slice(0, 1000))


def write_read(gpu_ctx, tmp_path, arrays, nullable=None):
    """test_write_read (io.rs:417-438): field nullable iff the array has a validity bitmap"""
    from strawboat_amd import WriteOptions, file as F
    from strawboat_amd.types import Compression as C
    pa = _pa()
    fields = [pa.field("name%d" % i, a.type, nullable=(a.null_count > 0 or a.buffers()[0] is not None) if nullable is None else nullable)
              for i, a in enumerate(arrays)]
    t = pa.Table.from_arrays(arrays, schema=pa.schema(fields))
    for comp in (C.LZ4, C.ZSTD, C.SNAPPY, C.NONE):
        path = tmp_path / ("t%d.sb" % comp)
        wo = WriteOptions(max_page_size=WRITE_PAGE, default_compression=comp, default_compress_ratio=2.0, forbidden_compressions=[])
        with F.NativeWriter(gpu_ctx, path, t.schema, wo) as w:
            w.start()
            w.write(t)
            w.finish()
            metas = w.metas
        assert F.read_meta(path) == metas
        got = F.read_table(gpu_ctx, path)
        for name in t.column_names:
            assert got.column(name).combine_chunks().equals(t.column(name).combine_chunks()), (name, comp)


def test_basic(gpu_ctx, tmp_path):
    pa = _pa()
    s = ["1.1", "2.2", "3.3", "4.4", "5.5", "6.6"]
    write_read(gpu_ctx, tmp_path, [
        pa.array([True, True, True, False, False, False]),
        pa.array([1, 2, 3, 4, 5, 6], pa.int8()), pa.array([1, 2, 3, 4, 5, 6], pa.int16()), pa.array([1, 2, 3, 4, 5, 6], pa.int32()),
        pa.array([1, 2, 3, 4, 5, 6], pa.int64()), pa.array([1, 2, 3, 4, 5, 6], pa.uint8()), pa.array([1, 2, 3, 4, 5, 6], pa.uint16()),
        pa.array([1, 2, 3, 4, 5, 6], pa.uint32()), pa.array([1, 2, 3, 4, 5, 6], pa.uint64()),
        pa.array([1.1, 2.2, 3.3, 4.4, 5.5, 6.6], pa.float32()), pa.array([1.1, 2.2, 3.3, 4.4, 5.5, 6.6], pa.float64()),
        pa.array(s, pa.string()), pa.array([x.encode() for x in s], pa.large_binary())])


def test_random_nonull(gpu_ctx, tmp_path):
    n = 10000
    write_read(gpu_ctx, tmp_path, [random_bool(n, 0.0), random_index(n, 0.0, n), random_double(n, 0.0, n), random_string(n, 0.0, n)])


def test_random(gpu_ctx, tmp_path):
    n = 10000
    write_read(gpu_ctx, tmp_path, [random_bool(n, 0.1), random_index(n, 0.1, n), random_index(n, 0.2, n, 43), random_index(n, 0.3, n, 44),
                                   random_index(n, 0.4, n, 45), random_double(n, 0.5, n), random_string(n, 0.4, n)])


def test_dict(gpu_ctx, tmp_path):
    n = 10000
    write_read(gpu_ctx, tmp_path, [random_bool(n, 0.1), random_index(n, 0.1, 8), random_index(n, 0.2, 8, 43), random_index(n, 0.3, 8, 44),
                                   random_index(n, 0.4, 8, 45), random_double(n, 0.5, 8), random_string(n, 0.4, 8)])


def test_freq(gpu_ctx, tmp_path):
    pa = _pa()
    v = np.tile(np.r_[np.full(WRITE_PAGE - 3, 20), [10000] * 3], 5).astype(np.uint32)
    write_read(gpu_ctx, tmp_path, [pa.array(v)])


def test_bitpacking(gpu_ctx, tmp_path):
    n = WRITE_PAGE * 5
    write_read(gpu_ctx, tmp_path, [random_index(n, 0.1, 8), random_index(n, 0.5, 8, 43)])


def test_delta_bitpacking(gpu_ctx, tmp_path):
    pa = _pa()
    n = WRITE_PAGE * 5
    write_read(gpu_ctx, tmp_path, [pa.array(np.arange(n, dtype=np.uint32)), pa.array(np.arange(n, dtype=np.int32))])


def test_onevalue(gpu_ctx, tmp_path):
    pa = _pa()
    n = 10000
    write_read(gpu_ctx, tmp_path, [pa.array(np.ones(n, bool)), pa.array(np.zeros(n, bool)), pa.array(np.full(n, 3, np.uint32)),
                                   random_index(n, 0.3, 1), random_string(n, 0.4, 1)])


def test_struct(gpu_ctx, tmp_path):
    write_read(gpu_ctx, tmp_path, [create_struct(1000, 0.2, 1000)], nullable=False)


def test_float(gpu_ctx, tmp_path):
    write_read(gpu_ctx, tmp_path, [random_double(1000, 0.5, 1000)])


def test_list(gpu_ctx, tmp_path):
    write_read(gpu_ctx, tmp_path, [create_list(1000, 0.2)])


def test_map(gpu_ctx, tmp_path):
    write_read(gpu_ctx, tmp_path, [create_map(1000, 0.2)])


def test_list_list(gpu_ctx, tmp_path):
    write_read(gpu_ctx, tmp_path, [list_of_pairs(create_list(2000, 0.2))], nullable=False)


def test_list_struct(gpu_ctx, tmp_path):
    write_read(gpu_ctx, tmp_path, [list_of_pairs(create_struct(2000, 0.2, 2000))], nullable=False)


def test_list_map(gpu_ctx, tmp_path):
    write_read(gpu_ctx, tmp_path, [list_of_pairs(create_map(2000, 0.2))], nullable=False)


def test_struct_list(gpu_ctx, tmp_path):
    pa = _pa()
    n = 10000
    lst = create_list(n, 0.2)
    st = pa.StructArray.from_arrays([random_string(n, 0.2, n), lst],
                                    fields=[pa.field("name", pa.large_binary()), pa.field("age", lst.type)])
    write_read(gpu_ctx, tmp_path, [st], nullable=False)
