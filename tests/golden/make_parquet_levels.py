"""Generates tests/golden/parquet_levels.json: repetition / definition level VALUES that an independent
implementation — parquet-cpp through pyarrow.parquet, data page V2 — writes for the nested shapes the oracle's
level generator (oracle/sbo_nested.cpp, a restatement of arrow2's write_rep_and_def / to_nested, reference call sites
src/write/serialize.rs:217-232, src/read/read_basic.rs:65-173) is tested on, and the definition levels of flat
nullable columns (reference src/write/serialize.rs:200-215, src/read/read_basic.rs:36-63).

Dremel levels are defined by the schema and the data, not by the writer, so the VALUES must agree even though the
two writers pack them differently (arrow2: one bit-packed hybrid run; parquet-cpp: a mix of RLE and bit-packed
runs).  Run in the build container (needs pyarrow); the fixture (inputs + expected levels, data only) is committed
and checked by tests/test_oracle_parquet_levels.py on CPU.
"""
import io
import json
import os
import sys

import numpy as np
import pyarrow as pa
import pyarrow.parquet as pq

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tests.nested_gen import make_nested  # noqa: E402

K_PRIMITIVE, K_LIST, K_LARGE_LIST, K_STRUCT = 0, 1, 2, 3


# ------------------------------------------------------------------ a minimal Thrift compact-protocol reader
class Compact:
    def __init__(self, buf, pos=0):
        self.b, self.p = buf, pos

    def byte(self):
        v = int(self.b[self.p])
        self.p += 1
        return v

    def varint(self):
        out, sh = 0, 0
        while True:
            v = self.byte()
            out |= (v & 0x7F) << sh
            sh += 7
            if not v & 0x80:
                return out

    def zigzag(self):
        v = self.varint()
        return (v >> 1) ^ -(v & 1)

    def skip(self, t):
        if t in (1, 2):          # bool true / false (in the field header)
            return
        if t == 3:
            self.p += 1
        elif t in (4, 5, 6):     # i16 / i32 / i64
            self.varint()
        elif t == 7:
            self.p += 8
        elif t == 8:             # binary
            n = self.varint()
            self.p += n
        elif t in (9, 10):       # list / set
            h = self.byte()
            n, et = h >> 4, h & 15
            if n == 15:
                n = self.varint()
            for _ in range(n):
                if et in (1, 2):
                    self.p += 1
                else:
                    self.skip(et)
        elif t == 12:
            self.struct(lambda fid, ft: None)
        else:
            raise ValueError("thrift type %d" % t)

    def struct(self, on_field):
        """on_field(field id, type) -> True if it consumed the value itself"""
        last = 0
        while True:
            h = self.byte()
            if h == 0:
                return
            delta, t = h >> 4, h & 15
            fid = last + delta if delta else self.zigzag()
            last = fid
            if not on_field(fid, t):
                self.skip(t)


def parse_data_page_v2(buf, pos):
    """-> (page type, header dict, position of the page body)"""
    c = Compact(buf, pos)
    out = {}

    def v2(fid, t):
        if fid in (1, 2, 3, 4, 5, 6) and t in (4, 5, 6):
            out[{1: "num_values", 2: "num_nulls", 3: "num_rows", 4: "encoding", 5: "def_len", 6: "rep_len"}[fid]] = c.zigzag()
            return True
        return False

    def top(fid, t):
        if fid == 1:
            out["type"] = c.zigzag()
            return True
        if fid == 2:
            out["uncompressed"] = c.zigzag()
            return True
        if fid == 3:
            out["compressed"] = c.zigzag()
            return True
        if fid == 8 and t == 12:
            c.struct(v2)
            return True
        return False
    c.struct(top)
    return out, c.p


def hybrid_decode(buf, bit_width, count):
    """parquet RLE / bit-packed hybrid -> `count` values"""
    out = []
    if bit_width == 0:
        return [0] * count
    p = 0
    nbytes = (bit_width + 7) // 8
    while len(out) < count:
        h, sh = 0, 0
        while True:
            v = int(buf[p])
            p += 1
            h |= (v & 0x7F) << sh
            sh += 7
            if not v & 0x80:
                break
        if h & 1:   # bit-packed run of (h >> 1) groups of 8 values
            n = (h >> 1) * 8
            bits = int.from_bytes(bytes(buf[p:p + (h >> 1) * bit_width]), "little")
            p += (h >> 1) * bit_width
            mask = (1 << bit_width) - 1
            for i in range(n):
                out.append((bits >> (i * bit_width)) & mask)
        else:       # RLE run
            v = int.from_bytes(bytes(buf[p:p + nbytes]), "little")
            p += nbytes
            out += [v] * (h >> 1)
    return out[:count]


def bit(lv, i):
    if lv.get("validity") is None:
        return True
    return bool((lv["validity"][i >> 3] >> (i & 7)) & 1)


def to_arrow(levels, k, leaf_values):
    """(pyarrow array, field nullable) for the sub-tree rooted at level k"""
    lv = levels[k]
    n = lv["length"]
    if lv["kind"] == K_PRIMITIVE:
        mask = None
        if lv["is_optional"] and lv.get("validity") is not None:
            mask = np.array([not bit(lv, i) for i in range(n)], bool)
        return pa.array(leaf_values[:n], type=pa.int32(), mask=mask), bool(lv["is_optional"])
    child, child_nullable = to_arrow(levels, k + 1, leaf_values)
    if lv["kind"] == K_STRUCT:
        f = pa.field("f", child.type, nullable=child_nullable)
        return pa.StructArray.from_arrays([child], fields=[f]), bool(lv["is_optional"])
    large = lv["kind"] == K_LARGE_LIST
    offs = np.asarray(lv["offsets"]).astype(np.int64 if large else np.int32)
    item = pa.field("item", child.type, nullable=child_nullable)
    typ = pa.large_list(item) if large else pa.list_(item)
    mask = None
    if lv["is_optional"] and lv.get("validity") is not None:
        mask = pa.array([not bit(lv, i) for i in range(n)], pa.bool_())
    cls = pa.LargeListArray if large else pa.ListArray
    return cls.from_arrays(pa.array(offs), child, type=typ, mask=mask), bool(lv["is_optional"])


def parquet_levels(arr, nullable):
    """write one column as ONE V2 data page (no dictionary, no compression) and read its level values back"""
    schema = pa.schema([pa.field("c", arr.type, nullable=nullable)])
    t = pa.Table.from_arrays([arr], schema=schema)
    sink = io.BytesIO()
    pq.write_table(t, sink, data_page_version="2.0", use_dictionary=False, compression="NONE",
                   data_page_size=1 << 30, write_statistics=False)
    buf = np.frombuffer(sink.getvalue(), np.uint8)
    md = pq.ParquetFile(io.BytesIO(sink.getvalue()))
    assert md.metadata.num_row_groups == 1 and md.metadata.row_group(0).num_columns == 1
    col = md.metadata.row_group(0).column(0)
    sc = md.schema.column(0)
    hdr, body = parse_data_page_v2(buf, col.data_page_offset)
    assert hdr["type"] == 3, "expected a DATA_PAGE_V2, got page type %r" % hdr.get("type")
    assert hdr["num_values"] == col.num_values, "more than one data page"

    def width(mx):
        return 0 if mx == 0 else int(mx).bit_length()
    rep = hybrid_decode(buf[body:body + hdr["rep_len"]], width(sc.max_repetition_level), hdr["num_values"])
    deff = hybrid_decode(buf[body + hdr["rep_len"]:body + hdr["rep_len"] + hdr["def_len"]], width(sc.max_definition_level),
                         hdr["num_values"])
    return dict(max_rep=sc.max_repetition_level, max_def=sc.max_definition_level, num_values=hdr["num_values"],
                num_rows=hdr["num_rows"], rep=rep, **{"def": deff})


def main():
    cases = []
    for shape in ("list", "large_list", "list_required", "list_list", "list_struct", "struct_list", "struct_struct"):
        for rows, seed in ((40, 1), (300, 2), (1000, 3)):
            levels, _ = make_nested(shape, rows, seed)
            leaf = np.arange(levels[-1]["length"] + 1, dtype=np.int32)
            arr, nullable = to_arrow(levels, 0, leaf)
            got = parquet_levels(arr, nullable)
            assert got["num_rows"] == rows
            ser = []
            for lv in levels:
                ser.append(dict(kind=lv["kind"], is_optional=bool(lv["is_optional"]), length=int(lv["length"]),
                                validity=None if lv.get("validity") is None else bytes(lv["validity"]).hex(),
                                offsets=None if lv.get("offsets") is None else np.asarray(lv["offsets"]).astype(np.int64).tolist()))
            cases.append(dict(kind="nested", shape=shape, rows=rows, seed=seed, levels=ser, parquet=got))
    # flat nullable columns: the def levels ARE the validity bits
    rng = np.random.default_rng(5)
    for rows, density in ((1, 0.0), (7, 0.5), (8, 0.5), (9, 1.0), (100, 0.3), (1000, 0.9), (4097, 0.1)):
        valid = rng.random(rows) >= density
        arr = pa.array(np.arange(rows, dtype=np.int64), mask=~valid)
        got = parquet_levels(arr, True)
        cases.append(dict(kind="flat", rows=rows, validity=bytes(np.packbits(valid, bitorder="little")).hex(), parquet=got))
    out = dict(pyarrow=pa.__version__, note="levels written by parquet-cpp (pyarrow.parquet, data page V2), decoded by "
                                            "tests/golden/make_parquet_levels.py", cases=cases)
    json.dump(out, open(os.path.join(HERE, "parquet_levels.json"), "w"), separators=(",", ":"))
    print("wrote %d cases" % len(cases))


if __name__ == "__main__":
    main()
