"""Random nested columns (level descriptors from the root to ONE leaf) shaped after the reference's
tests/it/io.rs:279-415 generators: list offsets step gen_range(0..3), 10 % null lists (a null list
is empty), leaves 20 % null, structs without nulls (the reference never tests null structs, and its
decoder drops the leaf slot of a null struct, src/read/read_basic.rs:103-106).

`expected_state` derives what read_validity_nested must rebuild for a row range straight from the
Arrow buffers — independently of any level arithmetic."""
import numpy as np

from oracle import sbo as S
from tests import gen


def _offsets(rng, n, null_density, large=False):
    valid = rng.random(n) > null_density
    step = np.where(valid, rng.integers(0, 3, n), 0)
    offs = np.zeros(n + 1, np.int64)
    np.cumsum(step, out=offs[1:])
    return offs.astype(np.int64 if large else np.int32), valid


def make_nested(shape, rows, seed=1):
    rng = np.random.default_rng(seed)
    lv = []

    def lst(n, optional=True, large=False, null_density=0.1):
        offs, valid = _offsets(rng, n, null_density if optional else 0.0, large)
        lv.append(dict(kind=S.K_LARGE_LIST if large else S.K_LIST, is_optional=optional,
                       validity=gen.pack_bits(valid) if optional else None, offsets=offs, length=n))
        return int(offs[-1])

    def struct(n, optional):
        lv.append(dict(kind=S.K_STRUCT, is_optional=optional, validity=None, length=n))
        return n

    def prim(n, optional=True):
        valid = rng.random(n) > 0.2
        lv.append(dict(kind=S.K_PRIMITIVE, is_optional=optional, validity=gen.pack_bits(valid) if optional else None,
                       length=n))

    n = rows
    if shape == "list":
        n = lst(n)
    elif shape == "large_list":
        n = lst(n, large=True)
    elif shape == "list_required":
        n = lst(n, optional=False)
    elif shape == "list_list":
        n = lst(lst(n))
    elif shape == "list_struct":
        n = struct(lst(n), True)
    elif shape == "struct_list":
        n = lst(struct(n, False))
    elif shape == "struct_struct":
        n = struct(struct(n, True), False)
    else:
        raise ValueError(shape)
    prim(n, optional=shape != "struct_struct")
    return lv, rows


def _bit(lv, i):
    if not lv["is_optional"]:
        return None
    if lv.get("validity") is None:
        return 1
    k = lv.get("validity_off", 0) + i
    return int((lv["validity"][k >> 3] >> (k & 7)) & 1)


def expected_state(levels, r0, length):
    D = len(levels)
    s, e = r0, r0 + length
    lengths, offsets, validity = [], [], []
    num_values = None
    # entries: leaf slots + one per empty list at any level
    extra = 0
    for k, lv in enumerate(levels):
        lengths.append(e - s)
        if lv["kind"] in (S.K_LIST, S.K_LARGE_LIST):
            offs = np.asarray(lv["offsets"]).astype(np.int64)
            offsets.append((offs[s:e] - offs[s]).tolist())
            validity.append([_bit(lv, i) for i in range(s, e)] if lv["is_optional"] else [])
            extra += int(np.count_nonzero(offs[s + 1:e + 1] == offs[s:e]))
            s, e = int(offs[s]), int(offs[e])
        elif lv["kind"] == S.K_STRUCT:
            offsets.append([])
            validity.append([_bit(lv, i) for i in range(s, e)] if lv["is_optional"] else [])
        else:
            offsets.append([])
            validity.append([])
    leaf = levels[-1]
    leaf_validity = [_bit(leaf, i) for i in range(s, e)] if leaf["is_optional"] else []
    num_values = (e - s) + extra
    return dict(lengths=lengths, offsets=offsets, validity=validity, leaf_validity=leaf_validity,
                leaf_start=s, leaf_count=e - s, num_values=num_values)
