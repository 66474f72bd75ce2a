"""Whole files: `write::NativeWriter` and `read::reader::{read_meta, infer_schema, NativeReader}` of the
reference over the C ABI, with pyarrow as the Arrow implementation on the caller's side (arrow2
upstream).

  writer:  NativeWriter.start / write / finish  (src/write/writer.rs:47-175; the per-column page loop
           of encode_chunk, src/write/common.rs:54-116, runs on the GPU through write.encode_columns
           / nested.write_nested);
  reader:  read_meta (src/read/reader.rs:168-178), infer_schema (:227-241), and read_table =
           batch_read_array over every field (src/read/batch_read.rs:66-230), which assembles
           list / struct arrays from the per-leaf nested states like read/array/{list,struct_}.rs.

Leaf order is arrow2's to_leaves order: depth first over the schema.  Utf8 is written as Binary
(src/write/serialize.rs:92-121).  The schema travels as an Arrow IPC Schema message flatbuffer.
"""
import ctypes as C
import struct
from typing import List

import numpy as np

from . import _native as N
from . import nested as NE
from . import read as R
from . import write as W
from .types import ColumnMeta, PageMeta, PhysicalType as PT, WriteOptions


def _pa():
    import pyarrow
    return pyarrow


def physical_type(dt):
    """Arrow logical type -> the physical kind write_simple dispatches on (serialize.rs:62-129)."""
    pa = _pa()
    t = pa.types
    table = [(t.is_boolean, PT.BOOLEAN), (t.is_int8, PT.INT8), (t.is_int16, PT.INT16), (t.is_int32, PT.INT32),
             (t.is_int64, PT.INT64), (t.is_uint8, PT.UINT8), (t.is_uint16, PT.UINT16), (t.is_uint32, PT.UINT32),
             (t.is_uint64, PT.UINT64), (t.is_float32, PT.FLOAT32), (t.is_float64, PT.FLOAT64),
             (t.is_date32, PT.INT32), (t.is_time32, PT.INT32), (t.is_date64, PT.INT64), (t.is_time64, PT.INT64),
             (t.is_timestamp, PT.INT64), (t.is_duration, PT.INT64), (t.is_decimal128, PT.INT128),
             (t.is_decimal256, PT.INT256), (t.is_large_string, PT.LARGE_BINARY), (t.is_large_binary, PT.LARGE_BINARY),
             (t.is_string, PT.BINARY), (t.is_binary, PT.BINARY), (t.is_null, PT.NULL)]
    for pred, p in table:
        if pred(dt):
            return p
    raise NotImplementedError("unsupported Arrow type %s (the reference: Float16/interval/FixedSizeBinary/Dictionary/Union "
                              "are unimplemented, src/write/primitive.rs:90-92, README.md:84-97)" % dt)


def _is_list(dt):
    """List, LargeList and Map: a Map is a list of a non-nullable `entries` struct (arrow2 to_nested treats
    DataType::Map like List, src/read/array/map.rs rebuilds it from the same offsets)"""
    t = _pa().types
    return t.is_list(dt) or t.is_large_list(dt) or t.is_map(dt)


def _value_field(dt):
    """the child field of a list-like type"""
    pa = _pa()
    if pa.types.is_map(dt):
        return pa.field("entries", pa.struct([dt.key_field, dt.item_field]), nullable=False)
    return dt.value_field


def _buf(b, start=0, length=None):
    if b is None:
        return None
    a = np.frombuffer(b, dtype=np.uint8)
    return a[start:] if length is None else a[start:start + length]


class _Leaf:
    """One leaf column on the host: the Nested chain above it (empty for a flat column) + its buffers."""

    def __init__(self, levels, ptype, nullable, rows, values, validity, validity_off, offsets, values_bit_off=0):
        self.levels, self.ptype, self.nullable, self.rows = levels, ptype, nullable, rows
        self.values, self.validity, self.validity_off, self.offsets = values, validity, validity_off, offsets
        self.values_bit_off = values_bit_off


def to_leaves(field, arr, chain=None) -> List[_Leaf]:
    """arrow2 to_nested + to_leaves for one field: depth-first leaves with their Nested chains."""
    pa = _pa()
    chain = list(chain or [])
    dt = arr.type
    bufs = arr.buffers()
    n, off = len(arr), arr.offset
    validity = _buf(bufs[0]) if bufs[0] is not None else None
    if _is_list(dt):
        w = 8 if pa.types.is_large_list(dt) else 4
        offs = np.frombuffer(bufs[1], dtype=np.int64 if w == 8 else np.int32)[off:off + n + 1]
        child = arr.values
        first = int(offs[0]) if n else 0
        if first or child.offset:  # re-base so that child element 0 is offsets[0]
            child = child.slice(first)
            offs = offs - offs[0]
        chain.append(dict(kind=NE.LARGE_LIST if w == 8 else NE.LIST, is_optional=field.nullable, length=n,
                          validity=validity, validity_off=off, offsets=np.ascontiguousarray(offs)))
        return to_leaves(_value_field(dt), child.slice(0, int(offs[-1]) if n else 0), chain)
    if pa.types.is_struct(dt):
        chain.append(dict(kind=NE.STRUCT, is_optional=field.nullable, length=n, validity=validity, validity_off=off,
                          offsets=None))
        out = []
        for i in range(dt.num_fields):
            out += to_leaves(dt.field(i), arr.field(i), chain)
        return out
    p = physical_type(dt)
    if chain:
        chain.append(dict(kind=NE.PRIMITIVE, is_optional=field.nullable, length=n, validity=validity, validity_off=off,
                          offsets=None))
    if p == PT.NULL:
        return [_Leaf(chain, p, field.nullable, n, None, None, 0, None)]
    if p == PT.BOOLEAN:
        return [_Leaf(chain, p, field.nullable, n, _buf(bufs[1]), validity, off, None, values_bit_off=off)]
    if PT.is_binary(p):
        w = PT.WIDTH[p]
        offs = np.frombuffer(bufs[1], dtype=np.int64 if w == 8 else np.int32)[off:off + n + 1]
        return [_Leaf(chain, p, field.nullable, n, _buf(bufs[2]) if bufs[2] is not None else np.zeros(0, np.uint8),
                      validity, off, np.ascontiguousarray(offs))]
    w = PT.WIDTH[p]
    return [_Leaf(chain, p, field.nullable, n, _buf(bufs[1], off * w, n * w), validity, off, None)]


def _up(ctx, a):
    import torch
    if a is None:
        return None
    a = np.ascontiguousarray(a).view(np.uint8).reshape(-1)
    return torch.from_numpy(a.copy()).to(ctx.torch_device)


def _check_file(rc):
    if rc != N.SB_OK:
        raise N.NativeError(rc, N.load().sb_file_last_error().decode())


class NativeWriter:
    """write::NativeWriter: `start()`, one `write(batch)`, `finish()`; `metas` then holds ColumnMeta."""

    def __init__(self, ctx, path, schema, options: WriteOptions):
        self.ctx, self.schema, self.options = ctx, schema, options
        self._lib = N.load()
        h = C.c_void_p()
        _check_file(self._lib.sb_file_writer_open(str(path).encode(), C.byref(h)))
        self._h = h
        self.metas: List[ColumnMeta] = []
        self.total_size = 0
        self._offset = 8

    def start(self):
        _check_file(self._lib.sb_file_writer_start(self._h))

    def _write_leaf(self, leaf: _Leaf):
        ctx = self.ctx
        col = W.DeviceColumn(leaf.ptype, leaf.nullable if not leaf.levels else False, leaf.rows, _up(ctx, leaf.values),
                             _up(ctx, leaf.validity), _up(ctx, leaf.offsets), leaf.values_bit_off, leaf.validity_off)
        if leaf.ptype == PT.NULL and not leaf.levels:  # empty pages, one PageMeta per page of rows (serialize.rs:63)
            ps = min(self.options.max_page_size or leaf.rows, leaf.rows) or 1
            return np.zeros(0, np.uint8), [PageMeta(0, min(ps, leaf.rows - r)) for r in range(0, leaf.rows, ps)]
        if leaf.levels:
            lv = [NE.NestedLevel(d["kind"], bool(d["is_optional"]), d["length"], _up(ctx, d["validity"]),
                                 _up(ctx, d["offsets"]), d["validity_off"]) for d in leaf.levels]
            enc = NE.write_nested(ctx, lv, col, self.options)
        else:
            enc = W.write(ctx, col, self.options)
        return enc.pages_numpy(), enc.metas

    def write(self, batch):
        """One RecordBatch / Table (one row group per file, writer.rs:108-112)."""
        pa = _pa()
        if isinstance(batch, pa.Table):
            batch = batch.combine_chunks().to_batches()[0] if batch.num_rows else pa.RecordBatch.from_pylist([], schema=batch.schema)
        assert batch.num_columns == len(self.schema), "chunk and schema disagree (writer.rs:119)"
        for i, field in enumerate(self.schema):
            for leaf in to_leaves(field, batch.column(i)):
                pages, metas = self._write_leaf(leaf)
                arr = (N.PageMetaC * max(len(metas), 1))()
                for k, m in enumerate(metas):
                    arr[k].length, arr[k].num_values = m.length, m.num_values
                pages = np.ascontiguousarray(pages, dtype=np.uint8)
                _check_file(self._lib.sb_file_writer_write_column(self._h, pages.ctypes.data_as(C.c_void_p), pages.size,
                                                                  arr, len(metas)))
                self.metas.append(ColumnMeta(self._offset, list(metas)))
                self._offset += pages.size

    def finish(self):
        raw = schema_to_bytes(self.schema)
        total = C.c_uint64(0)
        b = np.frombuffer(raw, dtype=np.uint8)
        _check_file(self._lib.sb_file_writer_finish(self._h, b.ctypes.data_as(C.c_void_p), b.size, C.byref(total)))
        self.total_size = int(total.value)

    def close(self):
        if self._h:
            self._lib.sb_file_writer_close(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


from .schema import schema_from_bytes, schema_to_bytes  # noqa: E402,F401  (the library's own flatbuffer writer / reader)


class FileReader:
    """Footer of a strawboat file + page reads (sb_file_reader_*)."""

    def __init__(self, path):
        self._lib = N.load()
        h = C.c_void_p()
        _check_file(self._lib.sb_file_reader_open(str(path).encode(), C.byref(h)))
        self._h = h
        self.metas: List[ColumnMeta] = []
        for i in range(self._lib.sb_file_reader_n_columns(h)):
            off, npg = C.c_uint64(), C.c_uint64()
            pm = C.POINTER(N.PageMetaC)()
            _check_file(self._lib.sb_file_reader_column(h, i, C.byref(off), C.byref(npg), C.byref(pm)))
            self.metas.append(ColumnMeta(int(off.value), [PageMeta(int(pm[k].length), int(pm[k].num_values))
                                                          for k in range(npg.value)]))
        sp, sl = C.c_void_p(), C.c_uint64()
        _check_file(self._lib.sb_file_reader_schema(h, C.byref(sp), C.byref(sl)))
        self.schema_bytes = C.string_at(sp, sl.value)

    def read_pages(self, col, first_page=0, n_pages=None):
        m = self.metas[col]
        n_pages = len(m.pages) - first_page if n_pages is None else n_pages
        size = sum(p.length for p in m.pages[first_page:first_page + n_pages])
        dst = np.zeros(max(size, 1), np.uint8)
        got = C.c_uint64()
        _check_file(self._lib.sb_file_reader_read_pages(self._h, col, first_page, n_pages, dst.ctypes.data_as(C.c_void_p),
                                                        dst.size, C.byref(got)))
        return dst[:got.value]

    def close(self):
        if self._h:
            self._lib.sb_file_reader_close(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def read_meta(path) -> List[ColumnMeta]:
    with FileReader(path) as r:
        return r.metas


def infer_schema(path):
    with FileReader(path) as r:
        return schema_from_bytes(r.schema_bytes)


# ---------------------------------------------------------------------------------- read_table
def _leaf_fields(field, chain=None):
    """DFS leaves of a field with the InitNested chain (kinds, nullable) above each (batch_read.rs:66-230)."""
    pa = _pa()
    chain = list(chain or [])
    dt = field.type
    if _is_list(dt):
        chain.append((NE.LARGE_LIST if pa.types.is_large_list(dt) else NE.LIST, field.nullable))
        return _leaf_fields(_value_field(dt), chain)
    if pa.types.is_struct(dt):
        chain.append((NE.STRUCT, field.nullable))
        out = []
        for i in range(dt.num_fields):
            out += _leaf_fields(dt.field(i), chain)
        return out
    if chain:
        chain.append((NE.PRIMITIVE, field.nullable))
    return [(field, chain)]


def _np(t, nbytes=None):
    a = t.cpu().numpy() if t is not None else np.zeros(0, np.uint8)
    return a if nbytes is None else a[:nbytes]


def _leaf_array(field, dev: R.DeviceArray, rows, validity_np):
    """pyarrow array of one decoded leaf"""
    pa = _pa()
    p = physical_type(field.type)
    vb = pa.py_buffer(validity_np.tobytes()) if validity_np is not None else None
    if p == PT.NULL:
        return pa.nulls(rows, field.type)
    if p == PT.BOOLEAN:
        return pa.Array.from_buffers(field.type, rows, [vb, pa.py_buffer(_np(dev.values, (rows + 7) // 8).tobytes())])
    if PT.is_binary(p):
        w = PT.WIDTH[p]
        offs = _np(dev.offsets, (rows + 1) * w)
        return pa.Array.from_buffers(field.type, rows, [vb, pa.py_buffer(offs.tobytes()),
                                                        pa.py_buffer(_np(dev.values, dev.values_len).tobytes())])
    return pa.Array.from_buffers(field.type, rows, [vb, pa.py_buffer(_np(dev.values, rows * PT.WIDTH[p]).tobytes())])


def _assemble(field, leaves, depth):
    """(array, state of the first leaf under this field): list / struct assembly from nested states
    (read/array/list.rs:21-60, struct_.rs)"""
    pa = _pa()
    dt = field.type
    if _is_list(dt):
        child, st = _assemble(_value_field(dt), leaves, depth + 1)
        n = st.lengths[depth]
        offs = st.offsets_numpy(depth)
        offs = offs.astype(np.int64 if pa.types.is_large_list(dt) else np.int32)
        vb = pa.py_buffer(st.validity_numpy(depth).tobytes()) if field.nullable else None
        return pa.Array.from_buffers(dt, n, [vb, pa.py_buffer(offs.tobytes())], children=[child]), st
    if pa.types.is_struct(dt):
        kids, first = [], None
        for i in range(dt.num_fields):
            a, st = _assemble(dt.field(i), leaves, depth + 1)
            kids.append(a)
            first = first or st
        n = first.lengths[depth]
        vb = pa.py_buffer(first.validity_numpy(depth).tobytes()) if field.nullable else None
        return pa.Array.from_buffers(dt, n, [vb], children=kids), first
    st = next(leaves)
    rows = st.leaf.rows
    v = st.leaf.validity_numpy() if field.nullable and st.leaf.validity is not None else None
    return _leaf_array(field, st.leaf, rows, v), st


def read_table(ctx, path):
    """Every column of the file as a pyarrow Table (batch_read over all fields)."""
    import torch
    pa = _pa()
    with FileReader(path) as r:
        schema = schema_from_bytes(r.schema_bytes)
        col = 0
        arrays = []
        for field in schema:
            states = []
            for leaf_field, chain in _leaf_fields(field):
                m = r.metas[col]
                metas = np.array([[p.length, p.num_values] for p in m.pages], dtype=np.uint64).reshape(-1, 2)
                host = r.read_pages(col)
                pages = torch.from_numpy(np.ascontiguousarray(host).copy()).to(ctx.torch_device) if host.size else \
                    torch.zeros(1, dtype=torch.uint8, device=ctx.torch_device)
                p = physical_type(leaf_field.type)
                if chain:
                    cp = R.ColumnPages(p, False, pages, metas)
                    st = NE.read_nested(ctx, cp, [k for k, _ in chain], [nl for _, nl in chain])
                    states.append(st)
                else:
                    rows = int(metas[:, 1].sum()) if metas.size else 0
                    if p == PT.NULL:
                        arrays.append(pa.nulls(rows, leaf_field.type))
                    else:
                        dev = R.read_simple(ctx, R.ColumnPages(p, leaf_field.nullable, pages, metas))
                        v = dev.validity_numpy() if leaf_field.nullable else None
                        arrays.append(_leaf_array(leaf_field, dev, rows, v))
                col += 1
            if states:
                arrays.append(_assemble(field, iter(states), 0)[0])
        return pa.Table.from_arrays(arrays, schema=schema)
