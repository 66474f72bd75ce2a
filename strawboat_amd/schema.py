"""The footer's schema bytes: arrow2's `schema_to_bytes` / `deserialize_schema` as the reference calls them
(src/write/writer.rs:137-139, src/read/reader.rs:227-241), over the C entry points sb_schema_to_bytes /
sb_schema_from_bytes (strawboat_amd/csrc/sb_schema.cpp: its own flatbuffer writer and reader).

pyarrow is only the host-side object model of a schema here (the mirror of arrow2's `Schema` / `Field` / `DataType`);
the bytes are produced and parsed by the library."""
import ctypes as C

from . import _native as N

# Type union tags of Schema.fbs (include/strawboat_hip.h SB_ARROW_*)
NULL, INT, FLOAT, BINARY, UTF8, BOOL, DECIMAL, DATE, TIME, TIMESTAMP, INTERVAL, LIST, STRUCT = 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13
FIXED_SIZE_BINARY, FIXED_SIZE_LIST, MAP, DURATION, LARGE_BINARY, LARGE_UTF8, LARGE_LIST = 15, 16, 17, 18, 19, 20, 21
_UNITS = {"s": 0, "ms": 1, "us": 2, "ns": 3}
_UNIT_NAMES = {v: k for k, v in _UNITS.items()}


def _pa():
    import pyarrow as pa
    return pa


def _flatten(field, out):
    """pre-order entries (dicts) of one pyarrow field"""
    pa = _pa()
    t = field.type
    e = dict(name=field.name, nullable=field.nullable, type_id=0, n_children=0, bit_width=0, is_signed=0, precision=0, scale=0,
             unit=0, timezone=None, metadata=dict(field.metadata or {}))
    kids = []
    T = pa.types
    if T.is_null(t):
        e["type_id"] = NULL
    elif T.is_boolean(t):
        e["type_id"] = BOOL
    elif T.is_integer(t):
        e.update(type_id=INT, bit_width=t.bit_width, is_signed=int(T.is_signed_integer(t)))
    elif T.is_floating(t):
        e.update(type_id=FLOAT, precision={16: 0, 32: 1, 64: 2}[t.bit_width])
    elif T.is_decimal(t):
        e.update(type_id=DECIMAL, precision=t.precision, scale=t.scale, bit_width=t.bit_width)
    elif T.is_date32(t):
        e.update(type_id=DATE, unit=0)
    elif T.is_date64(t):
        e.update(type_id=DATE, unit=1)
    elif T.is_time(t):
        e.update(type_id=TIME, unit=_UNITS[t.unit], bit_width=t.bit_width)
    elif T.is_timestamp(t):
        e.update(type_id=TIMESTAMP, unit=_UNITS[t.unit], timezone=t.tz)
    elif T.is_duration(t):
        e.update(type_id=DURATION, unit=_UNITS[t.unit])
    elif T.is_interval(t):
        e.update(type_id=INTERVAL, unit=2)          # pyarrow's only interval type: month_day_nano
    elif T.is_large_binary(t):
        e["type_id"] = LARGE_BINARY
    elif T.is_large_string(t):
        e["type_id"] = LARGE_UTF8
    elif T.is_fixed_size_binary(t):
        e.update(type_id=FIXED_SIZE_BINARY, bit_width=t.byte_width)
    elif T.is_binary(t):
        e["type_id"] = BINARY
    elif T.is_string(t):
        e["type_id"] = UTF8
    elif T.is_map(t):
        e.update(type_id=MAP, is_signed=int(t.keys_sorted))
        kids = [pa.field("entries", pa.struct([t.key_field, t.item_field]), nullable=False)]
    elif T.is_large_list(t):
        e["type_id"] = LARGE_LIST
        kids = [t.value_field]
    elif T.is_fixed_size_list(t):
        e.update(type_id=FIXED_SIZE_LIST, bit_width=t.list_size)
        kids = [t.value_field]
    elif T.is_list(t):
        e["type_id"] = LIST
        kids = [t.value_field]
    elif T.is_struct(t):
        e["type_id"] = STRUCT
        kids = [t.field(i) for i in range(t.num_fields)]
    else:
        raise NotImplementedError("no strawboat page layout for %s (Dictionary / Union / extension types are `unreachable!()` "
                                  "upstream, src/write/serialize.rs:124-129)" % t)
    e["n_children"] = len(kids)
    out.append(e)
    for k in kids:
        _flatten(k, out)


def _pack_kv(md) -> bytes:
    """{key: value} -> b"key\\0value\\0..." (the C ABI's packed KeyValue form)"""
    out = b""
    for k, v in md.items():
        k, v = bytes(k), bytes(v)
        if b"\0" in k or b"\0" in v:
            raise ValueError("metadata keys / values with an embedded NUL cannot cross the C ABI")
        out += k + b"\0" + v + b"\0"
    return out


def _unpack_kv(addr, n):
    """n pairs of NUL-terminated strings starting at address `addr` -> dict"""
    md, p = {}, addr
    for _ in range(n):
        k = C.string_at(p)
        p += len(k) + 1
        v = C.string_at(p)
        p += len(v) + 1
        md[k] = v
    return md


def schema_to_bytes(schema) -> bytes:
    """arrow2 `schema_to_bytes(&schema, &default_ipc_fields(&schema.fields))`: the bare IPC Message flatbuffer"""
    lib = N.load()
    ents = []
    for f in schema:
        _flatten(f, ents)
    arr = (N.SchemaFieldC * max(len(ents), 1))()
    keep, held = [], []
    for i, e in enumerate(ents):
        nm = e["name"].encode()
        tz = e["timezone"].encode() if e["timezone"] else None
        keep += [nm, tz]
        arr[i].name, arr[i].timezone = nm, tz
        if e["metadata"]:
            mb = C.create_string_buffer(_pack_kv(e["metadata"]))
            keep.append(mb.raw)
            held.append(mb)
            arr[i].metadata, arr[i].n_metadata = C.cast(mb, C.c_void_p), len(e["metadata"])
        for k in ("type_id", "n_children", "bit_width", "is_signed", "precision", "scale", "unit"):
            setattr(arr[i], k, int(e[k]))
        arr[i].nullable = 1 if e["nullable"] else 0
    md = schema.metadata or {}
    kv = (C.c_char_p * max(2 * len(md), 1))()
    for i, (k, v) in enumerate(md.items()):
        kv[2 * i], kv[2 * i + 1] = bytes(k), bytes(v)
    n = C.c_uint64(0)
    # a KeyValue costs ~48 bytes of flatbuffer structure (two padded, length-prefixed strings, a table, a vtable, a
    # vector slot) on top of its characters; should the estimate still be short, the call reports the length it needs
    n_pairs = len(md) + sum(len(e["metadata"]) for e in ents if e["metadata"])
    cap = 256 + 128 * len(ents) + sum(len(x) for x in keep if x) + sum(len(k) + len(v) for k, v in md.items()) + 64 * n_pairs
    for _ in range(2):
        buf = C.create_string_buffer(cap)
        rc = lib.sb_schema_to_bytes(arr, len(ents), len(schema), kv if md else None, len(md), buf, cap, C.byref(n))
        if rc == N.SB_OK:
            return buf.raw[:n.value]
        if n.value <= cap:
            break
        cap = n.value
    raise N.NativeError(rc, lib.sb_schema_last_error().decode())


def _build(ents, pos):
    """(pyarrow field, next position) for the entry at `pos`"""
    pa = _pa()
    e = ents[pos]
    pos += 1
    kids = []
    for _ in range(e.n_children):
        k, pos = _build(ents, pos)
        kids.append(k)
    tid = e.type_id
    unit = _UNIT_NAMES.get(e.unit, "ms")
    if tid == NULL:
        t = pa.null()
    elif tid == BOOL:
        t = pa.bool_()
    elif tid == INT:
        t = {(8, 1): pa.int8(), (16, 1): pa.int16(), (32, 1): pa.int32(), (64, 1): pa.int64(), (8, 0): pa.uint8(),
             (16, 0): pa.uint16(), (32, 0): pa.uint32(), (64, 0): pa.uint64()}[(e.bit_width, e.is_signed)]
    elif tid == FLOAT:
        t = [pa.float16(), pa.float32(), pa.float64()][e.precision]
    elif tid == DECIMAL:
        t = pa.decimal256(e.precision, e.scale) if e.bit_width == 256 else pa.decimal128(e.precision, e.scale)
    elif tid == DATE:
        t = pa.date32() if e.unit == 0 else pa.date64()
    elif tid == TIME:
        t = pa.time32(unit) if e.bit_width == 32 else pa.time64(unit)
    elif tid == TIMESTAMP:
        t = pa.timestamp(unit, tz=e.timezone.decode() if e.timezone else None)
    elif tid == DURATION:
        t = pa.duration(unit)
    elif tid == INTERVAL:
        t = pa.month_day_nano_interval()
    elif tid == BINARY:
        t = pa.binary()
    elif tid == UTF8:
        t = pa.string()
    elif tid == LARGE_BINARY:
        t = pa.large_binary()
    elif tid == LARGE_UTF8:
        t = pa.large_string()
    elif tid == FIXED_SIZE_BINARY:
        t = pa.binary(e.bit_width)
    elif tid == LIST:
        t = pa.list_(kids[0])
    elif tid == LARGE_LIST:
        t = pa.large_list(kids[0])
    elif tid == FIXED_SIZE_LIST:
        t = pa.list_(kids[0], e.bit_width)
    elif tid == STRUCT:
        t = pa.struct(kids)
    elif tid == MAP:
        ent = kids[0].type
        t = pa.map_(ent.field(0), ent.field(1), keys_sorted=bool(e.is_signed))
    else:
        raise NotImplementedError("Arrow type tag %d" % tid)
    md = _unpack_kv(e.metadata, e.n_metadata) if e.n_metadata else None
    return pa.field(e.name.decode(), t, nullable=bool(e.nullable), metadata=md), pos


def schema_from_bytes(raw: bytes):
    """arrow2 `deserialize_schema(&schema_bytes)` -> pyarrow.Schema (field names, types, nullability, field and schema
    custom_metadata)"""
    pa = _pa()
    lib = N.load()
    nf, nt, sl = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
    buf = C.create_string_buffer(bytes(raw), len(raw))
    cap, scap = 64, 4096
    while True:
        arr = (N.SchemaFieldC * cap)()
        strings = C.create_string_buffer(scap)
        rc = lib.sb_schema_from_bytes(buf, len(raw), arr, cap, C.byref(nf), C.byref(nt), strings, scap, C.byref(sl))
        if rc == N.SB_ERR_INVALID and (nf.value > cap or sl.value > scap):
            cap, scap = max(cap, nf.value), max(scap, sl.value)
            continue
        if rc != N.SB_OK:
            raise N.NativeError(rc, lib.sb_schema_last_error().decode())
        break
    fields, pos = [], 0
    for _ in range(nt.value):
        f, pos = _build(arr, pos)
        fields.append(f)
    np_, ml = C.c_uint64(0), C.c_uint64(0)
    mcap = 4096
    while True:
        mbuf = C.create_string_buffer(mcap)
        rc = lib.sb_schema_metadata_from_bytes(buf, len(raw), mbuf, mcap, C.byref(np_), C.byref(ml))
        if rc == N.SB_ERR_INVALID and ml.value > mcap:
            mcap = ml.value
            continue
        if rc != N.SB_OK:
            raise N.NativeError(rc, lib.sb_schema_last_error().decode())
        break
    md = _unpack_kv(C.addressof(mbuf), np_.value) if np_.value else None
    return pa.schema(fields, metadata=md)
