"""Decoder robustness probe (development): device-written Zstd pages whose payloads are several frames (one per 16 KiB
piece) with random byte flips / truncations / frame-header damage are decoded on the device; any status is fine, a GPU
fault or a hang is not, and an accepted page must decode to the original bytes or raise.
    python tests/probes/fuzz_zstd_frames.py [trials]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import strawboat_amd as sb
from strawboat_amd import read
from strawboat_amd._native import NativeError
from oracle import sbo as S
from tests import gen
from tests.test_gpu_encode import gpu_encode

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 200
ctx = sb.Context(0)
rng = np.random.default_rng(5)
cols = [gen.prim(S.T_I64, 60_000, uniq=300, runs=3), gen.binary(40_000, uniq=500, zipf=1.2, maxlen=30),
        gen.prim(S.T_U8, 200_000, uniq=1 << 7), gen.boolean(900_000, runs=3)]
nerr = nok = nsame = 0
for col in cols:
    enc = gpu_encode(ctx, col, default_compression=S.ZSTD, max_page_size=None if col["ptype"] != S.T_BIN32 else 20_000)
    pages, metas = enc.pages_numpy().copy(), enc.metas_array().copy()
    want = read.read_simple(ctx, read.ColumnPages(col["ptype"], col["nullable"], torch.from_numpy(pages).to(ctx.torch_device), metas)).values_numpy().copy()
    for t in range(trials):
        pg = pages.copy()
        kind = t % 4
        if kind == 0:      # random byte flips
            for _ in range(int(rng.integers(1, 6))):
                pg[int(rng.integers(0, pg.size))] ^= np.uint8(1 << int(rng.integers(0, 8)))
        elif kind == 1:    # damage near a frame header (magic numbers are easy to find)
            idx = np.flatnonzero((pg[:-4] == 0x28) & (pg[1:-3] == 0xB5) & (pg[2:-2] == 0x2F) & (pg[3:-1] == 0xFD))
            if idx.size:
                k = int(idx[int(rng.integers(0, idx.size))]) + int(rng.integers(0, 12))
                pg[min(k, pg.size - 1)] = np.uint8(rng.integers(0, 256))
        elif kind == 2:    # zero a random span
            a = int(rng.integers(0, pg.size)); pg[a:a + int(rng.integers(1, 64))] = 0
        else:              # swap two random 16-byte spans
            a, b = (int(x) for x in rng.integers(0, max(pg.size - 16, 1), 2))
            tmp = pg[a:a + 16].copy(); pg[a:a + 16] = pg[b:b + 16]; pg[b:b + 16] = tmp
        try:
            got = read.read_simple(ctx, read.ColumnPages(col["ptype"], col["nullable"], torch.from_numpy(pg).to(ctx.torch_device), metas))
            nok += 1
            nsame += int(np.array_equal(got.values_numpy(), want))
        except NativeError:
            nerr += 1
print("done: %d accepted (%d of them decode to the original), %d refused" % (nok, nsame, nerr))
