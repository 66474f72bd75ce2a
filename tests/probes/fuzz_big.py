"""Corrupted inputs for the two decoders of this round: LZ4 blocks of the workgroup decoder (sb_lz4_big.h) and batches of
literals-only Zstd frames read lane per stream (k_inflate's batch mode).  Every call must return — a status or garbage
bytes — without a GPU fault or hang:  python tests/probes/fuzz_big.py [trials]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import strawboat_amd as sb  # noqa: E402
from oracle import sbo as S  # noqa: E402
from strawboat_amd import read, write  # noqa: E402
from strawboat_amd._native import NativeError  # noqa: E402
from strawboat_amd.types import Compression as C, WriteOptions  # noqa: E402
from tests import test_gpu_lz4 as TL  # noqa: E402


def main():
    import torch
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    ctx = sb.Context(0)
    rng = np.random.default_rng(7)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(ctx.torch_device)
    refused = accepted = 0
    # ---- big LZ4 blocks
    for name, blk in TL._big().items():
        want = TL._py_lz4(blk)
        for t in range(trials):
            b = bytearray(blk)
            kind = t % 4
            if kind == 0:
                for _ in range(1 + t % 7):
                    b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
            elif kind == 1:
                a = int(rng.integers(0, len(b) - 64))
                b[a:a + 32] = bytes(32)
            elif kind == 2:
                a = int(rng.integers(0, len(b) - 64))
                b[a:a + 16] = b"\xff" * 16
            else:
                b = b[:int(rng.integers(len(b) // 2, len(b)))]
            n_out = want.size if t % 3 else int(want.size + rng.integers(-5, 6))
            pages, metas = TL.lz4_page(bytes(b), max(n_out, 1))
            try:
                read.read_simple(ctx, read.ColumnPages(S.T_U8, False, up(pages), metas))
                accepted += 1
            except NativeError:
                refused += 1
    print("LZ4 big blocks: %d refused, %d decoded" % (refused, accepted), flush=True)
    # ---- batches of literals-only Zstd frames (offset-like Int32 columns: 16 frames per page, 128 columns = 32768 frames)
    cols = []
    for s in range(128):
        v = np.cumsum(np.random.default_rng(s).integers(2, 5, 1 << 20)).astype(np.int32)
        cols.append(write.DeviceColumn(S.T_I32, False, v.size, up(v), None, None))
    enc = write.encode_columns(ctx, cols, WriteOptions(default_compression=C.ZSTD, max_page_size=65536))
    ctx.synchronize()
    host = [e.pages[:e.length].cpu().numpy().copy() for e in enc]
    refused = accepted = 0
    for t in range(max(trials // 6, 4)):
        pages = []
        for h, e in zip(host, enc):
            b = h.copy()
            for _ in range(20):
                a = int(rng.integers(9, b.size - 40))
                k = t % 3
                if k == 0:
                    b[a] = rng.integers(0, 256)
                elif k == 1:
                    b[a:a + 24] = 0
                else:
                    b[a:a + 8] = 255
            pages.append(read.ColumnPages(S.T_I32, False, up(b), e.metas_array()))
        try:
            read.batch_read_columns(ctx, pages)
            ctx.synchronize()
            accepted += 1
        except NativeError:
            refused += 1
    print("Zstd batches: %d calls refused, %d decoded" % (refused, accepted), flush=True)


if __name__ == "__main__":
    main()
