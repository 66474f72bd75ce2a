"""Synthetic inputs of the BASELINE.json configurations and of the
reference's own bench shapes (benches/write_strawboat.rs:30-67), shared by bench.py and tests/.

A column is a dict: ptype (oracle/sbo.h PhysType == SB_TYPE_*), nullable, rows, values (numpy),
validity (packed LSB-first bits or None), offsets (numpy or None).  Everything is seeded numpy;
nothing here touches the GPU, the oracle or the reference.
"""
import numpy as np

# physical types (include/strawboat_hip.h SB_TYPE_*)
T_BOOL, T_I32, T_I64, T_F64, T_BIN32 = 0, 3, 4, 12, 13
PAGE = 65536


def pack_bits(b):
    return np.packbits(np.asarray(b, dtype=bool), bitorder="little")


def arrow_bytes(col):
    """U(column) of BASELINE.md §3: values + ceil(N/8) if nullable + (N+1)*sizeof(offset) if binary"""
    n = col["rows"]
    b = (n + 7) // 8 if col["ptype"] == T_BOOL else np.asarray(col["values"]).nbytes
    if col["nullable"]:
        b += (n + 7) // 8
    if col["offsets"] is not None:
        b += np.asarray(col["offsets"]).nbytes
    return int(b)


def c1_int64(seed=42, rows=1_000_000):
    """C1: Int64 uniform in [0, 2^63), non-nullable; written as ONE page, no compression"""
    rng = np.random.default_rng(seed)
    return dict(ptype=T_I64, nullable=False, rows=rows, values=rng.integers(0, 2**63 - 1, rows), validity=None, offsets=None)


def c2_values(seed, rows=1_000_000):
    rng = np.random.default_rng(seed)
    nrun = rows // 16 + 64
    lens = rng.geometric(1.0 / 32.0, nrun)
    while lens.sum() < rows:
        lens = np.concatenate([lens, rng.geometric(1.0 / 32.0, nrun)])
    k = rng.integers(0, 256, lens.size)
    vals = np.repeat(k, lens)[:rows].astype(np.float64)
    valid = np.packbits(rng.random(rows) < 0.9, bitorder="little")
    return vals, valid


def c2_float64(seed=42, rows=1_000_000):
    """C2: nullable Float64, value = float(k), k piecewise constant (runs ~ Geometric(mean 32)), 10 % nulls"""
    vals, valid = c2_values(seed, rows)
    return dict(ptype=T_F64, nullable=True, rows=rows, values=vals, validity=valid, offsets=None)


def zipf_utf8(rows, seed, null_density=None):
    """C3: Utf8<i32>, 10 000-word vocabulary ("w{rank}" padded to Uniform[4,24] bytes), rank ~ Zipf(1.1) truncated"""
    rng = np.random.default_rng(seed)
    lens_v = np.empty(10_000, np.int64)
    mat = np.full((10_000, 24), ord("x"), np.uint8)
    for k in range(10_000):
        L = int(rng.integers(4, 25))
        w = ("w%d" % k).encode()
        mat[k, :len(w)] = np.frombuffer(w, np.uint8)
        lens_v[k] = max(L, len(w))   # str.ljust never truncates
    rank = np.minimum(rng.zipf(1.1, rows), 10_000) - 1
    lens = lens_v[rank]
    offs = np.zeros(rows + 1, np.int64)
    np.cumsum(lens, out=offs[1:])
    data = np.empty(int(offs[-1]), np.uint8)
    step = 1 << 20   # bounded temporaries for 10 M-row columns
    for r0 in range(0, rows, step):
        r1 = min(rows, r0 + step)
        m = mat[rank[r0:r1]]
        data[offs[r0]:offs[r1]] = m[np.arange(24)[None, :] < lens[r0:r1, None]]
    validity = None if null_density is None else pack_bits(rng.random(rows) >= null_density)
    return dict(ptype=T_BIN32, nullable=True, rows=rows, values=data, validity=validity, offsets=offs.astype(np.int32))


def c4_columns(rows=10_000_000, seed=42):
    """C4: the 8-column mixed schema {Int32 x2, Float64 x2, Utf8 x2, Boolean x2 (10 % null)}"""
    cols = []
    for j in range(2):
        rng = np.random.default_rng(seed + j)
        cols.append(("int32_%d" % j, dict(ptype=T_I32, nullable=True, rows=rows,
                                          values=rng.integers(0, 1000, rows).astype(np.int32), validity=None, offsets=None)))
    for j in range(2):
        cols.append(("float64_%d" % j, c2_float64(seed + 10 + j, rows)))
    for j in range(2):
        cols.append(("utf8_%d" % j, zipf_utf8(rows, seed + 20 + j, null_density=0.1)))
    for j in range(2):
        rng = np.random.default_rng(seed + 30 + j)
        cols.append(("boolean_%d" % j, dict(ptype=T_BOOL, nullable=True, rows=rows, values=pack_bits(rng.random(rows) < 0.5),
                                            validity=pack_bits(rng.random(rows) >= 0.1), offsets=None)))
    return cols


def c5_nested(rows=1_000_000, seed=42):
    """C5: List<Struct<Int64, Utf8>>: list length Uniform{0,1,2}, 10 % null lists (a null list is empty), struct
    non-null, leaves 20 % null.  Returns (levels_a, leaf_a, levels_b, leaf_b): per leaf column the level descriptors
    root -> leaf (dicts: kind, is_optional, validity, offsets, length) and the leaf column."""
    K_PRIMITIVE, K_LIST, K_STRUCT = 0, 1, 3
    rng = np.random.default_rng(seed)
    list_valid = rng.random(rows) > 0.1
    lens = np.where(list_valid, rng.integers(0, 3, rows), 0)
    offs = np.zeros(rows + 1, np.int32)
    np.cumsum(lens, out=offs[1:])
    n = int(offs[-1])
    a_vals = rng.integers(-2**40, 2**40, n)
    a_valid = pack_bits(rng.random(n) >= 0.2)
    widx = rng.integers(0, 500, n)
    words = [("s%d" % k).encode() for k in range(500)]
    wl = np.array([len(w) for w in words], np.int64)
    mat = np.zeros((500, 4), np.uint8)
    for k, w in enumerate(words):
        mat[k, :len(w)] = np.frombuffer(w, np.uint8)
    bl = wl[widx]
    boffs = np.zeros(n + 1, np.int64)
    np.cumsum(bl, out=boffs[1:])
    b_vals = mat[widx][np.arange(4)[None, :] < bl[:, None]]
    b_valid = pack_bits(rng.random(n) >= 0.2)

    def levels(leaf_valid):
        return [dict(kind=K_LIST, is_optional=True, validity=pack_bits(list_valid), offsets=offs, length=rows),
                dict(kind=K_STRUCT, is_optional=True, validity=None, length=n),
                dict(kind=K_PRIMITIVE, is_optional=True, validity=leaf_valid, length=n)]
    leaf_a = dict(ptype=T_I64, nullable=True, rows=n, values=a_vals, validity=a_valid, offsets=None)
    leaf_b = dict(ptype=T_BIN32, nullable=True, rows=n, values=b_vals, validity=b_valid, offsets=boffs.astype(np.int32))
    return levels(a_valid), leaf_a, levels(b_valid), leaf_b


def _bits(bm, lo, hi):
    """bits [lo, hi) of an LSB-first bitmap, re-packed from bit 0"""
    return pack_bits(np.unpackbits(bm, bitorder="little")[lo:hi].astype(bool))


def c5_slice(levels, leaf, r0, r1):
    """top-level rows [r0, r1) of one C5 leaf column (levels list -> struct -> leaf) as a column of its own: offsets
    re-based to 0, bitmaps re-packed from bit 0 — what a rank holds when it owns a page range of the leaf column"""
    lst, st, lf = levels
    offs = np.asarray(lst["offsets"])
    e0, e1 = int(offs[r0]), int(offs[r1])
    lv = [dict(kind=lst["kind"], is_optional=lst["is_optional"], validity=_bits(lst["validity"], r0, r1),
               offsets=(offs[r0:r1 + 1] - offs[r0]).astype(offs.dtype), length=r1 - r0),
          dict(kind=st["kind"], is_optional=st["is_optional"], validity=None, length=e1 - e0),
          dict(kind=lf["kind"], is_optional=lf["is_optional"], validity=_bits(lf["validity"], e0, e1), length=e1 - e0)]
    if leaf["offsets"] is None:
        col = dict(ptype=leaf["ptype"], nullable=leaf["nullable"], rows=e1 - e0, values=leaf["values"][e0:e1],
                   validity=_bits(leaf["validity"], e0, e1), offsets=None)
    else:
        bo = np.asarray(leaf["offsets"])
        col = dict(ptype=leaf["ptype"], nullable=leaf["nullable"], rows=e1 - e0, values=leaf["values"][int(bo[e0]):int(bo[e1])],
                   validity=_bits(leaf["validity"], e0, e1), offsets=(bo[e0:e1 + 1] - bo[e0]).astype(bo.dtype),
                   column_values_len=int(np.asarray(leaf["values"]).size))   # array.values().len() of the whole leaf column
    return lv, col


# ---- the reference's bench shapes (benches/write_strawboat.rs:53-67; arrow2 util::bench_util generators):
# nullable field, LZ4, max_page_size 8192, default_compress_ratio None
def cont_bool(rows, seed=42):
    """create_boolean_array(size, null_density 0.1, true_density 0.5)"""
    rng = np.random.default_rng(seed)
    return dict(ptype=T_BOOL, nullable=True, rows=rows, values=pack_bits(rng.random(rows) < 0.5),
                validity=pack_bits(rng.random(rows) >= 0.1), offsets=None)


def cont_utf8(rows, seed=42):
    """create_string_array::<i32>(size, 4, 0.1, 42): random alphanumeric strings of length 4, 10 % null (empty)"""
    rng = np.random.default_rng(seed)
    alnum = np.frombuffer(b"ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789", np.uint8)
    valid = rng.random(rows) >= 0.1
    lens = np.where(valid, 4, 0)
    offs = np.zeros(rows + 1, np.int64)
    np.cumsum(lens, out=offs[1:])
    data = alnum[rng.integers(0, 62, int(offs[-1]))]
    return dict(ptype=T_BIN32, nullable=True, rows=rows, values=data, validity=pack_bits(valid), offsets=offs.astype(np.int32))


def cont_i64(rows, seed=42):
    """create_primitive_array::<i64>(size, 0.0): uniform random i64, no nulls, nullable field"""
    rng = np.random.default_rng(seed)
    return dict(ptype=T_I64, nullable=True, rows=rows, values=rng.integers(-2**63, 2**63 - 1, rows), validity=None, offsets=None)
