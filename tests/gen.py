"""Seeded synthetic columns shaped after the reference's tests/it/io.rs generators
(io.rs:343-415: StdRng seed 42, value = gen_range(0..uniq), null density 0..0.5) and the
BASELINE.md configs.  A column is a dict:
  ptype, nullable, rows, values (np array), validity (packed bits or None), offsets (np or None)
"""
import numpy as np

from oracle import sbo as S

NP_OF = {S.T_I8: np.int8, S.T_I16: np.int16, S.T_I32: np.int32, S.T_I64: np.int64, S.T_U8: np.uint8,
         S.T_U16: np.uint16, S.T_U32: np.uint32, S.T_U64: np.uint64, S.T_F32: np.float32, S.T_F64: np.float64}


def pack_bits(b):
    return np.packbits(np.asarray(b, dtype=bool), bitorder="little")


def make_validity(rng, rows, null_density):
    if null_density is None:
        return None
    return pack_bits(rng.random(rows) >= null_density)


def prim(ptype, rows, uniq=1000, null_density=None, nullable=None, seed=42, sorted_=False, runs=None):
    rng = np.random.default_rng(seed)
    if ptype in (S.T_I128, S.T_I256):
        w = S.WIDTH[ptype]
        base = rng.integers(0, uniq, rows).astype(np.int64)
        vals = np.zeros((rows, w // 8), np.int64)
        vals[:, 0] = base
        vals[:, -1] = -(base % 3 == 0).astype(np.int64)  # some negative numbers
        vals = vals.reshape(-1)
    else:
        if runs:
            nrun = rows // runs + 1
            vals = np.repeat(rng.integers(0, uniq, nrun), rng.geometric(1.0 / runs, nrun))[:rows]
            if vals.size < rows:
                vals = np.concatenate([vals, np.full(rows - vals.size, vals[-1] if vals.size else 0)])
        else:
            vals = rng.integers(0, uniq, rows)
        if sorted_:
            vals = np.sort(vals)
        vals = vals.astype(NP_OF[ptype])
    validity = make_validity(rng, rows, null_density)
    if nullable is None:
        nullable = validity is not None
    return dict(ptype=ptype, nullable=nullable, rows=rows, values=vals, validity=validity, offsets=None)


def boolean(rows, null_density=None, nullable=None, seed=42, p_true=0.5, runs=None):
    rng = np.random.default_rng(seed)
    if runs:
        nrun = rows // runs + 1
        b = np.repeat(rng.random(nrun) < p_true, rng.geometric(1.0 / runs, nrun))[:rows]
        if b.size < rows:
            b = np.concatenate([b, np.zeros(rows - b.size, bool)])
    else:
        b = rng.random(rows) < p_true
    validity = make_validity(rng, rows, null_density)
    if nullable is None:
        nullable = validity is not None
    return dict(ptype=S.T_BOOL, nullable=nullable, rows=rows, values=pack_bits(b), validity=validity, offsets=None)


def binary(rows, uniq=1000, null_density=None, nullable=None, seed=42, large=False, zipf=None, minlen=0, maxlen=12):
    rng = np.random.default_rng(seed)
    vocab = []
    for i in range(uniq):
        L = int(rng.integers(minlen, maxlen + 1))
        s = ("w%d" % i).encode()
        vocab.append((s * (L // len(s) + 1))[:L] if L else b"")
    if zipf:
        idx = (rng.zipf(zipf, rows) - 1) % uniq
    else:
        idx = rng.integers(0, uniq, rows)
    lens = np.array([len(vocab[i]) for i in range(uniq)], np.int64)[idx]
    offs = np.zeros(rows + 1, np.int64)
    np.cumsum(lens, out=offs[1:])
    data = np.frombuffer(b"".join(vocab[i] for i in idx), np.uint8).copy() if rows else np.zeros(0, np.uint8)
    validity = make_validity(rng, rows, null_density)
    if nullable is None:
        nullable = validity is not None
    return dict(ptype=S.T_BIN64 if large else S.T_BIN32, nullable=nullable, rows=rows, values=data,
                validity=validity, offsets=offs.astype(np.int64 if large else np.int32))


def oracle_write(col, **opt):
    return S.write_column(col["ptype"], col["nullable"], col["rows"], col["values"], validity=col["validity"],
                          offsets=col["offsets"], options=S.make_options(**opt))


def oracle_read(col, pages, metas):
    return S.read_column(col["ptype"], col["nullable"], pages, metas)
