"""Damaged long bit-packed pages (top-level Bitpacking / DeltaBitpacking pages and Dict pages with bit-packed indices, 4096
blocks and more: the reader's head walk + k_bp_guess + guessed stretches + one-lane walk) against the oracle's verdict:
a page the oracle refuses must be refused, a page it decodes must decode to the same bytes (or be refused), and no call may
fault or hang.   python tests/probes/fuzz_long_bp.py [trials]

Known difference (kind 4, a flipped bit in the header's compressed size that makes it SMALLER): the oracle checks that a page's
decoder consumed exactly PageMeta.length bytes; the reference's readers hand the page buffer to the decoder and ignore what is
left of it (src/read/array/integer.rs:69-81), and for the Extend codecs the decoder sees the rest of the buffer anyway
(src/compression/integer/mod.rs:108-114) — the device follows the reference there and decodes such a page to the original values."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import strawboat_amd as sb  # noqa: E402
from oracle import sbo as S  # noqa: E402
from strawboat_amd import read  # noqa: E402
from strawboat_amd._native import NativeError  # noqa: E402


def main():
    import torch
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    ctx = sb.Context(0)
    rng = np.random.default_rng(11)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(ctx.torch_device)
    n = 128 * 5000
    grow = np.minimum(rng.integers(0, 1 << 20, n), np.arange(n) // 64 + 1).astype(np.uint32)
    shapes = {
        "bitpack_one_width": (S.T_U32, rng.integers(256, 512, n).astype(np.uint32), S.make_options(force_codec=S.BITPACK)),
        "bitpack_head_grows": (S.T_U32, grow, S.make_options(force_codec=S.BITPACK)),
        "delta_sorted": (S.T_U32, np.cumsum(rng.integers(0, 4000, n)).astype(np.uint32), S.make_options(force_codec=S.DELTABP)),
        "dict_bitpacked_indices": (S.T_I64, (rng.integers(0, 3000, n) * 1_000_003).astype(np.int64), S.make_options(force_codec=S.DICT)),
    }
    for name, (pt, v, o) in shapes.items():
        page, metas = S.write_column(pt, False, n, v, options=o)
        page = np.array(page, dtype=np.uint8)
        same = refused_both = refused_only_dev = differ = 0
        for t in range(trials):
            b = page.copy()
            kind = t % 5
            if kind == 0:      # a width byte somewhere (positions that ARE headers when all blocks share a width)
                w = int(b[9]) if name.startswith("bitpack_one") else 9
                blk = int(rng.integers(0, n // 128))
                pos = min(len(b) - 1, 9 + blk * (1 + 16 * w))
                b[pos] = int(rng.integers(0, 64))
            elif kind == 1:    # random bytes
                for _ in range(1 + t % 5):
                    b[int(rng.integers(9, len(b)))] = int(rng.integers(0, 256))
            elif kind == 2:    # the head of the body
                a = int(rng.integers(9, 9 + 4096))
                b[a:a + 8] = rng.integers(0, 256, 8).astype(np.uint8)
            elif kind == 3:    # truncated (the meta keeps the length: the header's sizes disagree)
                b = b[: int(rng.integers(len(b) // 2, len(b)))]
            else:              # header sizes
                b[1 + int(rng.integers(0, 8))] ^= 1 << int(rng.integers(0, 8))
            m = np.array(metas, dtype=np.uint64).copy()
            m[0, 0] = len(b)
            try:
                want = S.read_column(pt, False, b, m)["values"]
            except Exception:
                want = None
            try:
                got = read.read_simple(ctx, read.ColumnPages(pt, False, up(b), m)).values_numpy()
            except NativeError:
                got = None
            if want is None and got is None:
                refused_both += 1
            elif want is None:
                differ += 1      # the oracle refuses, the device decodes
                print("   %s trial %d kind %d: the oracle refuses, the device decodes" % (name, t, kind), flush=True)
            elif got is None:
                refused_only_dev += 1
            elif np.array_equal(got, np.ascontiguousarray(want).view(np.uint8).reshape(-1)):
                same += 1
            else:
                differ += 1
                w8 = np.ascontiguousarray(want).view(np.uint8).reshape(-1)
                bad = np.flatnonzero(got[: w8.size] != w8) if got.size == w8.size else np.array([-1])
                print("   %s trial %d kind %d: decoded differently (%d bytes differ, first at %d, sizes %d / %d)" % (name, t, kind, bad.size, int(bad[0]), got.size, w8.size), flush=True)
        print("%-24s %3d equal, %3d refused by both, %3d refused by the device only, %3d DIFFER" % (name, same, refused_both, refused_only_dev, differ), flush=True)


main()
