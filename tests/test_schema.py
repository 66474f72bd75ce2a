"""The footer's schema bytes (strawboat_amd/csrc/sb_schema.cpp behind sb_schema_to_bytes / sb_schema_from_bytes): what
the reference gets from arrow2's `schema_to_bytes` (src/write/writer.rs:137-139) and reads with `deserialize_schema`
(src/read/reader.rs:227-241) — a bare Arrow IPC `Message` flatbuffer holding a `Schema`.

Pinned by an independent implementation in both directions: Arrow C++ (pyarrow.ipc.read_schema) parses the library's
bytes to the same schema, and the library parses Arrow C++'s bytes (Schema.serialize()) to the same schema."""
import struct

import pytest

pa = pytest.importorskip("pyarrow")

from strawboat_amd import schema as SC  # noqa: E402


def message_of(raw: bytes) -> bytes:
    """an encapsulated IPC message (continuation marker, length, flatbuffer, no body) around the bare flatbuffer"""
    pad = (-len(raw)) % 8
    return b"\xff\xff\xff\xff" + struct.pack("<i", len(raw) + pad) + raw + b"\0" * pad


def bare_of(msg: bytes) -> bytes:
    assert msg[:4] == b"\xff\xff\xff\xff"
    n = struct.unpack("<i", msg[4:8])[0]
    return msg[8:8 + n]


SCHEMAS = {
    "primitives": pa.schema([pa.field("i8", pa.int8()), pa.field("u8", pa.uint8(), nullable=False), pa.field("i16", pa.int16()),
                             pa.field("u16", pa.uint16()), pa.field("i32", pa.int32()), pa.field("u32", pa.uint32()),
                             pa.field("i64", pa.int64(), nullable=False), pa.field("u64", pa.uint64()),
                             pa.field("f32", pa.float32()), pa.field("f64", pa.float64()), pa.field("b", pa.bool_()),
                             pa.field("n", pa.null())]),
    "strings": pa.schema([pa.field("s", pa.utf8()), pa.field("ls", pa.large_utf8()), pa.field("bin", pa.binary()),
                          pa.field("lbin", pa.large_binary(), nullable=False), pa.field("fsb", pa.binary(16))]),
    "temporal": pa.schema([pa.field("d32", pa.date32()), pa.field("d64", pa.date64()), pa.field("t32", pa.time32("ms")),
                           pa.field("t64", pa.time64("ns")), pa.field("ts", pa.timestamp("us")),
                           pa.field("tsz", pa.timestamp("ns", tz="Europe/Berlin")), pa.field("dur", pa.duration("s")),
                           pa.field("dec", pa.decimal128(20, 3)), pa.field("dec256", pa.decimal256(40, 7))]),
    "nested": pa.schema([pa.field("l", pa.list_(pa.field("item", pa.int32()))),
                         pa.field("ll", pa.large_list(pa.field("item", pa.utf8(), nullable=False)), nullable=False),
                         pa.field("st", pa.struct([pa.field("a", pa.int64()), pa.field("b", pa.utf8())])),
                         pa.field("ls", pa.list_(pa.field("item", pa.struct([pa.field("x", pa.int64()), pa.field("y", pa.utf8())])))),
                         pa.field("m", pa.map_(pa.utf8(), pa.int32())),
                         pa.field("fl", pa.list_(pa.field("item", pa.float32()), 4))]),
    "metadata": pa.schema([pa.field("a", pa.int32())], metadata={b"writer": b"strawboat", b"k2": b""}),
    "empty": pa.schema([]),
    # the schema of the reference's own test files (tests/it/io.rs:72-278): Utf8 + primitives + List<Struct<..>>
    "c5": pa.schema([pa.field("c", pa.list_(pa.field("item", pa.struct([pa.field("a", pa.int64()), pa.field("b", pa.utf8())]))))]),
}


@pytest.mark.parametrize("name", sorted(SCHEMAS))
def test_arrow_cpp_reads_our_bytes(name):
    sch = SCHEMAS[name]
    raw = SC.schema_to_bytes(sch)
    got = pa.ipc.read_schema(pa.py_buffer(message_of(raw)))
    assert got.equals(sch, check_metadata=True), (got, sch)


@pytest.mark.parametrize("name", sorted(SCHEMAS))
def test_we_read_arrow_cpp_bytes(name):
    sch = SCHEMAS[name]
    raw = bare_of(sch.serialize().to_pybytes())
    got = SC.schema_from_bytes(raw)
    assert got.equals(sch.remove_metadata()), (got, sch)


@pytest.mark.parametrize("name", sorted(SCHEMAS))
def test_own_round_trip(name):
    sch = SCHEMAS[name]
    assert SC.schema_from_bytes(SC.schema_to_bytes(sch)).equals(sch.remove_metadata())


def test_message_header_layout():
    """the fixed part of the flatbuffer: root offset, Message{version = V5, header_type = Schema, bodyLength = 0} — the
    fields arrow2's schema_to_bytes sets (arrow2 io/ipc/write/schema.rs: MetadataVersion::V5, MessageHeader::Schema)"""
    raw = SC.schema_to_bytes(SCHEMAS["primitives"])
    root = struct.unpack_from("<I", raw, 0)[0]
    vt = root - struct.unpack_from("<i", raw, root)[0]
    vt_len, _tbl_len = struct.unpack_from("<HH", raw, vt)
    slots = struct.unpack_from("<%dH" % ((vt_len - 4) // 2), raw, vt + 4)
    # Message: version (i16) | header_type (u8) | header (offset) | bodyLength (i64) | custom_metadata
    assert slots[0] and struct.unpack_from("<h", raw, root + slots[0])[0] == 4          # MetadataVersion::V5
    assert slots[1] and raw[root + slots[1]] == 1                                       # MessageHeader::Schema
    assert slots[2]
    if len(slots) > 3 and slots[3]:
        assert struct.unpack_from("<q", raw, root + slots[3])[0] == 0


def test_malformed_bytes_are_refused():
    raw = SC.schema_to_bytes(SCHEMAS["nested"])
    from strawboat_amd._native import NativeError
    for bad in (raw[:10], b"", b"\x00" * 64, raw[:len(raw) // 2]):
        with pytest.raises(NativeError):
            SC.schema_from_bytes(bad)
