"""ORACLE — TEST INFRASTRUCTURE ONLY.  File framing restated from the reference:
NativeWriter::start / finish (src/write/writer.rs:91-167), ColumnMeta offsets
(src/write/common.rs:76,111-114), read_meta / deserialize_meta (src/read/reader.rs:148-178),
infer_schema's byte arithmetic (src/read/reader.rs:227-241), magic + EOS (src/lib.rs:34-35,
src/write/common.rs:124-128).  The schema flatbuffer (arrow2 schema_to_bytes [3P]) is opaque."""
import struct

MAGIC = b"ARROW2\x00\x00"
EOS = b"\xff\xff\xff\xff\x00\x00\x00\x00"


def write_file(columns, schema_bytes):
    """columns: [(page_bytes, [(length, num_values), ...]), ...] in leaf order -> file bytes"""
    out = bytearray(MAGIC)
    metas = []
    for pages, pm in columns:
        metas.append((len(out), list(pm)))
        out += bytes(pages)
    out += bytes(schema_bytes)
    meta = bytearray(struct.pack("<Q", len(metas)))
    for off, pm in metas:
        meta += struct.pack("<QQ", off, len(pm))
        for length, nv in pm:
            meta += struct.pack("<QQ", int(length), int(nv))
    out += meta
    out += struct.pack("<II", len(schema_bytes), len(meta))
    out += EOS
    return bytes(out)


def read_meta(buf):
    (meta_size,) = struct.unpack_from("<I", buf, len(buf) - 12)
    pos = len(buf) - 16 - meta_size
    (n,) = struct.unpack_from("<Q", buf, pos)
    pos += 8
    cols = []
    for _ in range(n):
        off, npg = struct.unpack_from("<QQ", buf, pos)
        pos += 16
        pages = []
        for _ in range(npg):
            pages.append(struct.unpack_from("<QQ", buf, pos))
            pos += 16
        cols.append((off, pages))
    return cols


def schema_bytes(buf):
    schema_size, meta_size = struct.unpack_from("<II", buf, len(buf) - 16)
    start = len(buf) - 16 - meta_size - schema_size
    return bytes(buf[start:start + schema_size])
