"""Damaged LZ4 blocks of megabytes through the block-parallel decoder (sb_lz4_giant.h): every call must return — a status,
or (the format has no checksum) bytes equal to what the oracle's decoder makes of the damaged block — without a GPU
fault or a hang:  python tests/probes/fuzz_giant.py [trials per shape]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import strawboat_amd as sb  # noqa: E402
from oracle import sbo as S  # noqa: E402
from strawboat_amd import read  # noqa: E402
from strawboat_amd._native import NativeError  # noqa: E402
from tests import test_gpu_lz4 as TL  # noqa: E402


def main():
    import torch
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    ctx = sb.Context(0)
    rng = np.random.default_rng(11)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(ctx.torch_device)
    refused = same = differ = 0
    for name, blk in TL._giant().items():
        if len(blk) < 2 << 20:
            continue
        n_out = TL._lz4_out_len(blk)
        for t in range(trials):
            b = bytearray(blk)
            kind = t % 5
            if kind == 0:
                for _ in range(1 + t % 7):
                    b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
            elif kind == 1:
                a = int(rng.integers(0, len(b) - 64))
                b[a:a + 32] = bytes(32)
            elif kind == 2:
                a = int(rng.integers(0, len(b) - 400))
                b[a:a + 300] = b"\xff" * 300            # a run of 255s: every position a header of > 64 length bytes
            elif kind == 3:
                b = b[:int(rng.integers(len(b) // 2, len(b)))]
            else:
                a = int(rng.integers(0, len(b) - 8))
                b[a:a + 2] = b"\x00\x00"                # (maybe) an offset of 0
            n = n_out if t % 3 else int(n_out + rng.integers(-5, 6))
            pages, metas = TL.lz4_page(bytes(b), max(n, 1))
            try:
                want = S.block_decompress(S.LZ4, np.frombuffer(bytes(b), np.uint8), max(n, 1))
            except Exception:
                want = None
            try:
                got = read.read_simple(ctx, read.ColumnPages(S.T_U8, False, up(pages), metas)).values_numpy()
            except NativeError:
                got = None
            if got is None:
                refused += 1
                assert True
            elif want is None:
                differ += 1
                print("  %s trial %d: the oracle refuses, the device decoded" % (name, t), flush=True)
            elif np.array_equal(got, want):
                same += 1
            else:
                differ += 1
                print("  %s trial %d: decoded bytes differ from the oracle's" % (name, t), flush=True)
        print("%-14s done" % name, flush=True)
    print("giant LZ4 blocks: %d refused, %d decoded like the oracle, %d DIFFER" % (refused, same, differ), flush=True)


if __name__ == "__main__":
    main()
