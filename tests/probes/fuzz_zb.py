"""Decoder robustness probe (development) for the block-parallel Zstd pipeline (sb_zstd_blocks.h): pages whose blocks are
multi-block frames written by libzstd, with random byte flips / zeroed spans / swapped spans / damage inside the block and
section headers, decoded with SB_ZSTD_BLOCKS=1.  Any status is fine; a GPU fault or a hang is not; a page both the oracle and
the device accept must decode to the same bytes.
    SB_ZSTD_BLOCKS=1 python tests/probes/fuzz_zb.py [trials per shape]"""
import os, sys
os.environ.setdefault("SB_ZSTD_BLOCKS", "1")
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import strawboat_amd as sb
from strawboat_amd import read
from strawboat_amd._native import NativeError
from tests import gen
from tests.test_gpu_zstd import recompress
import tests.test_gpu_zstd_blocks as T

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 100
ctx = sb.Context(0)
rng = np.random.default_rng(11)
nerr = nok = nsame = ndiff_ok = 0
for name in ("i64_40bit", "words", "periodic", "zipf_text", "small_ints", "zeros_then_noise"):
    col = T.SHAPES[name]
    p0, m0 = gen.oracle_write(col)
    zp, zm = recompress(col, p0, m0, 3)
    # block headers of the frame (positions relative to the page): damage lands there half of the time
    hdr = [9 + 4 + 1 + 4]   # after hdr9 + magic + FHD + (fcs 4): first block header (approximate for single-segment frames)
    for t in range(trials):
        pg = zp.copy()
        kind = t % 5
        if kind == 0:
            for _ in range(int(rng.integers(1, 6))):
                pg[int(rng.integers(9, pg.size))] ^= np.uint8(1 << int(rng.integers(0, 8)))
        elif kind == 1:
            k = hdr[0] + int(rng.integers(0, 24))
            pg[min(k, pg.size - 1)] = np.uint8(rng.integers(0, 256))
        elif kind == 2:
            a = int(rng.integers(9, pg.size)); pg[a:a + int(rng.integers(1, 64))] = 0
        elif kind == 3:
            a, b = (int(x) for x in rng.integers(9, max(pg.size - 16, 10), 2))
            tmp = pg[a:a + 16].copy(); pg[a:a + 16] = pg[b:b + 16]; pg[b:b + 16] = tmp
        else:   # a run of 0xFF (maximal lengths / sizes wherever it lands)
            a = int(rng.integers(9, pg.size)); pg[a:a + int(rng.integers(1, 8))] = 0xFF
        try:
            want = gen.oracle_read(col, pg, zm)["values"]
        except Exception:
            want = None
        try:
            got = read.read_simple(ctx, read.ColumnPages(col["ptype"], col["nullable"], torch.from_numpy(pg).to(ctx.torch_device), zm)).values_numpy()
            nok += 1
            if want is not None:
                if np.array_equal(got, want):
                    nsame += 1
                else:
                    print("MISMATCH", name, t, kind)
                    sys.exit(1)
            else:
                ndiff_ok += 1
        except NativeError:
            nerr += 1
print("done: %d accepted (%d equal to the oracle's decode, %d the oracle refuses), %d refused; stats %s" % (nok, nsame, ndiff_ok, nerr, ctx.zstd_block_stats()))
