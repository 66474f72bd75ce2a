"""Known-answer vectors for the two third-party layouts the reference delegates to crates that are not under /root/reference
and for which no second implementation is importable in this image (no pyroaring, no bitpacking binding):

  * RoaringBitmap portable serialisation (roaring 0.10.1, `serialize_into`; src/compression/integer/freq.rs:58-76) — derived here
    from the published RoaringFormatSpec: cookie 12346 (no run containers: roaring-rs never writes them), u32 container count,
    (u16 key, u16 cardinality - 1) per container, u32 byte offset per container, then per container a sorted u16 array
    (cardinality <= 4096) or a 1024 x u64 bitset;
  * BitPacker4x (bitpacking 0.8; src/compression/integer/bp.rs:45-61, delta_bp.rs:45-65) — derived from the crate's documented
    register layout: 128 values = 32 vectors of 4 lanes, lane l packs values l, l + 4, ... LSB first, word k of lane l at u32
    index 4 k + l; the packer ORs the (delta) value into the word WITHOUT masking it to num_bits, and delta_bp.rs:50 takes
    num_bits from the RAW values — so a delta that wraps (an unsorted block) spills its high bits over the slots behind it.

The script is a second, independent statement of those rules (plain Python integers, no code shared with oracle/): it writes
tests/golden/layout_vectors.json = inputs (as seeds / formulas), the expected bytes as sha256 + length + head, and
tests/test_oracle_layouts.py checks the oracle against them.  Run: python tests/golden/make_layout_vectors.py"""
import hashlib
import json
import os
import struct

import numpy as np


def roaring_portable(rows):
    rows = sorted(set(int(r) for r in rows))
    cont = {}
    for r in rows:
        cont.setdefault(r >> 16, []).append(r & 0xFFFF)
    keys = sorted(cont)
    out = struct.pack("<II", 12346, len(keys))
    for k in keys:
        out += struct.pack("<HH", k, len(cont[k]) - 1)
    pos = len(out) + 4 * len(keys)
    offs, body = b"", b""
    for k in keys:
        offs += struct.pack("<I", pos + len(body))
        lows = cont[k]
        if len(lows) <= 4096:
            body += b"".join(struct.pack("<H", x) for x in lows)
        else:
            words = [0] * 1024
            for x in lows:
                words[x >> 6] |= 1 << (x & 63)
            body += b"".join(struct.pack("<Q", w) for w in words)
    return out + offs + body


def bp4x_block(vals, delta, initial):
    """(num_bits, 16 * num_bits bytes) of one 128-value block"""
    assert len(vals) == 128
    acc = 0
    for v in vals:
        acc |= v
    nb = acc.bit_length()
    words = [0] * (4 * nb)
    prev = initial
    for j, v in enumerate(vals):
        x = (v - prev) & 0xFFFFFFFF if delta else v
        prev = v
        lane, slot = j & 3, j >> 2
        bit = slot * nb
        w, sh = bit >> 5, bit & 31
        if nb:
            words[4 * w + lane] = (words[4 * w + lane] | (x << sh)) & 0xFFFFFFFF       # no mask to num_bits
            if sh + nb > 32:
                words[4 * (w + 1) + lane] |= x >> (32 - sh)
    return nb, b"".join(struct.pack("<I", w) for w in words)


def digest(b):
    return {"length": len(b), "sha256": hashlib.sha256(b).hexdigest(), "head_hex": b[:48].hex()}


def freq_case(name, n, top, exc_rows, seed):
    """column of n u32 rows: `top` everywhere, exception rows hold seeded values != top"""
    rng = np.random.default_rng(seed)
    ex = np.array(sorted(exc_rows), np.int64)
    vals = rng.integers(1, 1 << 30, ex.size, dtype=np.int64) * 2 + 1      # odd: never equal to the even top value
    return {"name": name, "rows": n, "top": top, "seed": seed, "exceptions": name, "n_exceptions": int(ex.size),
            "roaring": digest(roaring_portable(ex))}, ex, vals


def exception_rows(name):
    if name == "bitmap_container":          # 5000 of the first 65 536 rows: one container, past the 4096 limit of the array form
        return list(range(3, 65536, 13))[:5000]
    if name == "two_containers":            # rows on both sides of 65 536, arrays
        return [3, 70000, 70001, 131071]
    if name == "full_container":            # every row of [65 536, 131 072): cardinality 65 536, stored as 65 535; + one row in front
        return [10] + list(range(65536, 131072))
    raise KeyError(name)


def bp_case(name):
    rng = np.random.default_rng({"nb31": 31, "nb32": 32, "delta_wrap": 7, "delta_sorted": 8, "nb0_then_nb17": 9}[name])
    if name == "nb31":
        v = rng.integers(0, 1 << 31, 256)
        v[5] |= 1 << 30
        return [int(x) for x in v], False
    if name == "nb32":
        v = rng.integers(0, 1 << 32, 256)
        v[200] |= 1 << 31
        return [int(x) for x in v], False
    if name == "delta_wrap":                # unsorted: every other delta wraps around 2^32; num_bits = 8 from the raw values
        v = rng.integers(0, 256, 256)
        return [int(x) for x in v], True
    if name == "delta_sorted":
        v = np.cumsum(rng.integers(0, 1000, 256))
        return [int(x) for x in v], True
    v = np.concatenate([np.zeros(128, np.int64), rng.integers(0, 1 << 17, 128)])
    return [int(x) for x in v], False


def main():
    out = {"_comment": __doc__.split("\n\n")[0], "freq": [], "bitpack": []}
    for name, n, top in (("bitmap_container", 100_000, 20), ("two_containers", 140_000, 1000), ("full_container", 300_000, 42)):
        case, _, _ = freq_case(name, n, top, exception_rows(name), 1)
        out["freq"].append(case)
    for name in ("nb31", "nb32", "delta_wrap", "delta_sorted", "nb0_then_nb17"):
        vals, delta = bp_case(name)
        body, initial = b"", 0
        for b in range(0, len(vals), 128):
            nb, packed = bp4x_block(vals[b:b + 128], delta, initial)
            body += bytes([nb]) + packed
            initial = vals[b + 127]
        out["bitpack"].append({"name": name, "delta": delta, "rows": len(vals), "blocks": digest(body),
                               "num_bits": [body[0], body[1 + 16 * body[0]]]})
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "layout_vectors.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
