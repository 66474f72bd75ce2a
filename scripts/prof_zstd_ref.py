"""Development probe: decode of the C5 leaf columns as pages written by libzstd (one frame per page buffer,
src/compression/basic.rs:122-135) — the shape `bench.py` reports as c5.leaf_pages_reference_written.
    python scripts/prof_zstd_ref.py [arrays] [reps]
Set SB_ZSTD_BLOCKS=0 / 1 to compare the one-wave path with the block-parallel pipeline (sb_zstd_blocks.h)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B  # noqa: E402
import workloads as W  # noqa: E402


def main():
    arrays = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    which = sys.argv[2] if len(sys.argv) > 2 else "ab"   # a: the Int64 leaves, b: the Utf8 leaves
    import torch
    import strawboat_amd as sb
    from oracle import sbo
    ctx = sb.Context(0)
    h = B.GpuHarness(ctx)
    t0 = time.time()
    gen = B.gen_parallel(lambda s: W.c5_nested(seed=s), range(42, 42 + arrays))
    cols = []
    for la, a, lb, b in gen:
        if "a" in which:
            cols.append(dict(a, nullable=False, validity=None))
        if "b" in which:
            cols.append(dict(b, nullable=False, validity=None))
    o = sbo.make_options(default_compression=sbo.ZSTD, max_page_size=B.PAGE)
    print("generated in %.1fs" % (time.time() - t0), flush=True)
    for rep in range(1):   # (the second pass runs with the context knowing about Zstd: SB_ZSTD_BLOCKS unset = auto)
        r = B.measure_reference_pages(h, cols, o, reps=3, check_all=True)
        d = r["decode"]
        print("pass %d: arrow %.1f MB pages %.1f MB: decode %.3f ms = %.1f GB/s" % (rep, r["arrow_MB"], r["page_MB"], d["ms"], d["GBps"]))
        for k, v in d.get("kernels_ms", {}).items():
            print("    %-28s %8.3f ms" % (k, v))
    import ctypes as C
    lib = ctx._lib
    if hasattr(lib, "sb_debug_zb_timers"):   # a development build (-DZB_TL): phase clocks of zb_exec / LzSeqExec::run
        out = (C.c_uint64 * 20)()
        lib.sb_debug_zb_timers(ctx._h, out)
        v = [int(x) for x in out]
        print("zb_exec (2 frames): rec %d rep %d scans %d run %d rest %d batches %d" % (v[0], v[1], v[2], v[3], v[4], v[11]))
        print("run: hbm-mode %d (n=%d) literals %d classify %d matches %d flush %d | rounds %d serial matches %d" % (v[12], v[18], v[13], v[14], v[15], v[16], v[19], v[17]))
    print("ok")


if __name__ == "__main__":
    main()
