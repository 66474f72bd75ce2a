#!/usr/bin/env python3
"""bench.py — encode+decode GB/s of uncompressed Arrow bytes on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one batch of synthetic input that is already
resident in HBM: encode the batch (Arrow buffers -> strawboat pages) and decode the pages it
produced back (pages -> Arrow buffers), through the C ABI of libstrawboat_hip.so.

Workload at N=1 (BASELINE.json configs[1], "C2"): columns of 1 M-row nullable Float64,
64 Ki-row pages, value = float(k) with k piecewise constant (run length ~ Geometric(mean 32),
k uniform in [0,256)), 10 % nulls.  The batch is `--columns` such columns (default 512 =
512 M rows, 4.16 GB of Arrow bytes, 8192 pages): far beyond the 256 MB Infinity Cache
(SURVEY.md §8d), and enough pages that the one-workgroup-per-page kernels run several rounds per
CU instead of exactly one (with 64 columns = 1024 pages = 4 per CU every phase of every workgroup
runs in lockstep and the fixed ~0.15 ms of small kernels and launch gaps weighs 20 %; at 256 columns it
still weighs 10 %, at 512 columns 6 %).  Default mode "adaptive": default_compress_ratio = 2.0 and the codec of every
page is chosen on the device by the reference's selector (it picks RLE for this data: sampled
ratio ~14 vs Dict 7.6 vs Patas < 4; the CPU oracle agrees, tests/test_oracle_golden.py).  Nothing is in
forbidden_compressions: every codec of the reference is a candidate, as with its default options.

Multi-GPU (torchrun, one rank per GPU): every rank owns its own `--columns` columns (weak
scaling, pages of independent columns shard with no data-path collective); the only collective
is one RCCL all_gather of the page metas at the end of the timed region (the metadata a file
writer needs for ColumnMeta offsets).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (MI355X_MICROARCH.md: 8 TB/s spec, ~6.3 TB/s achievable)
ROWS = 1_000_000
PAGE = 65536


def gen_c2_column(seed, rows=ROWS):
    rng = np.random.default_rng(seed)
    nrun = rows // 16 + 64
    lens = rng.geometric(1.0 / 32.0, nrun)
    while lens.sum() < rows:
        lens = np.concatenate([lens, rng.geometric(1.0 / 32.0, nrun)])
    k = rng.integers(0, 256, lens.size)
    vals = np.repeat(k, lens)[:rows].astype(np.float64)
    valid = np.packbits(rng.random(rows) < 0.9, bitorder="little")
    return vals, valid


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--columns", type=int, default=512, help="1 M-row columns per GPU in one batch")
    ap.add_argument("--codec", default="adaptive", choices=["adaptive", "rle", "none", "dict"],
                    help="adaptive = default_compress_ratio 2.0, codec chosen per page on the device")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback in strawboat_amd)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)  # RCCL

    import strawboat_amd as sb
    from strawboat_amd import read, write
    from strawboat_amd.types import Compression, PhysicalType, WriteOptions

    dev = torch.device("cuda", local_rank)
    ctx = sb.Context(local_rank)
    B = args.columns
    codec = {"adaptive": -1, "rle": Compression.RLE, "none": Compression.NONE, "dict": Compression.DICT}[args.codec]
    if codec < 0:   # the reference's adaptive mode with its default options: nothing forbidden
        opts = WriteOptions(max_page_size=PAGE, default_compress_ratio=2.0,
                            forbidden_compressions=[])
    else:
        opts = WriteOptions(max_page_size=PAGE, force_codec=codec)

    # ---- synthetic batch, resident in HBM before the timed region
    cols = []
    host0 = None
    for b in range(B):
        vals, valid = gen_c2_column(42 + 1000 * rank + b)
        if b == 0:
            host0 = (vals, valid)
        cols.append(write.DeviceColumn(PhysicalType.FLOAT64, True, ROWS,
                                       torch.from_numpy(vals.view(np.uint8)).to(dev),
                                       torch.from_numpy(valid).to(dev)))
    torch.cuda.synchronize()
    U_col = ROWS * 8 + (ROWS + 7) // 8          # uncompressed Arrow bytes of one column
    U = U_col * B

    # ---- one untimed pass to allocate outputs and to verify the round trip bit for bit
    enc = write.encode_columns(ctx, cols, opts)
    ctx.synchronize()
    pages = [read.ColumnPages(PhysicalType.FLOAT64, True, e.pages, e.metas_array()) for e in enc]
    dec = read.batch_read_columns(ctx, pages)
    ctx.synchronize()
    page_bytes = sum(e.length for e in enc)
    valid_rows = torch.from_numpy(np.unpackbits(host0[1], bitorder="little")[:ROWS].astype(bool)).to(dev)
    assert torch.equal(dec[0].validity[:(ROWS + 7) // 8], cols[0].validity), "validity round trip failed"
    got = dec[0].values.view(torch.float64)
    ref = cols[0].values.view(torch.float64)
    assert torch.equal(got[valid_rows], ref[valid_rows]), "value round trip failed"

    wbatch = write.WriteBatch(ctx, cols, opts, out=enc)   # descriptors built once, outside the loop
    rbatch = read.ReadBatch(ctx, pages, out=dec)

    def step():
        wbatch.enqueue()
        rbatch.enqueue()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    ctx.synchronize()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.synchronize()
    if world > 1:  # the one collective of the path: page metas of every rank (RCCL all_gather)
        from strawboat_amd import shard
        local = {rank * B + i: e.metas_array() for i, e in enumerate(enc)}
        all_metas = shard.gather_metas(local, world * B, device=dev)
        assert len(shard.column_metas(all_metas)) == world * B
    barrier()
    t1 = time.perf_counter()
    # ---- the same K steps once more with HIP events around every kernel launch (recorded by the
    # library on the stream the kernels run on): per-kernel durations for the roofline figure.  Kept
    # out of the timed region above because 2 event records per launch cost ~15 % of a 0.6 ms step.
    ctx.profile(True)
    for _ in range(args.steps):
        step()
    ctx.synchronize()
    stats = ctx.profile_read()
    ctx.profile(False)

    elapsed = t1 - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ms_per_step = elapsed / args.steps * 1e3
    value = world * 2.0 * U * args.steps / elapsed / 1e9   # whole job: encode + decode bytes

    if rank == 0:
        # ---- roofline of the dominant kernel: algorithmic bytes / HIP-event time
        A = {"k_expand": page_bytes + U,            # A_dec = page bytes read + Arrow bytes written
             "k_expand_rle": page_bytes + U,
             "k_enc_emit_pages": U + page_bytes,    # A_enc = Arrow bytes read + page bytes written
             "k_enc_emit_tiles": U + page_bytes,
             "k_enc_select": U + page_bytes}        # fused selection + RLE: Arrow bytes read once, pages written
        base = lambda k: k.split("<")[0]
        dom = max((k for k in stats if base(k) in A), key=lambda k: stats[k][1], default=None)
        roof = None
        if dom:
            n, tot = stats[dom]
            avg_ms = tot / n
            A = {k: A[base(k)] for k in stats if base(k) in A}
            achieved = A[dom] / (avg_ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                    "avg_kernel_ms": round(avg_ms, 4), "algorithmic_bytes_per_launch": A[dom]}
        kernels = {k: {"launches": v[0], "avg_ms": round(v[1] / v[0], 4)} for k, v in stats.items()}
        # HBM traffic of the dominant kernel from the PMC counters (rocprofv3 --pmc FETCH_SIZE /
        # WRITE_SIZE, separate passes of this same command; summary committed under profiles/)
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
            if roof and pmc["config"] == {"workload": "C2", "columns_per_gpu": B, "codec": args.codec}:
                cand = [rec["hbm_bytes_per_launch"] for name, rec in pmc["kernels"].items() if name.split("<")[0].startswith(dom.split("<")[0])]
                if cand:
                    roof["traffic"] = max(cand)   # several template instances share a name: the one that did the work
                roof["traffic_source"] = "profiles/r01_pmc_traffic.json (PMC, per launch, gfx950 FETCH_SIZE x2 correction)"
        except (OSError, KeyError, ValueError):
            pass

        cpu = None
        if not args.no_cpu_baseline:
            from oracle import sbo
            if codec < 0:
                o = sbo.make_options(max_page_size=PAGE, ratio=2.0, forbidden=())
            else:
                o = sbo.make_options(max_page_size=PAGE, force_codec=codec)
            tw, tr = sbo.time_roundtrip(sbo.T_F64, True, ROWS, host0[0], validity=host0[1], options=o, iters=3)
            cpu = {"value": round(2.0 * U_col / (tw + tr) / 1e9, 3), "unit": "GB/s", "cores": 1, "kind": "port",
                   "sample": "1 column (1 M rows, 16 pages) of the same workload, encode+decode, best of 3, "
                             "single thread (the reference is single-threaded); C++ restatement, not the Rust binary",
                   "encode_s": round(tw, 4), "decode_s": round(tr, 4)}
        out = {
            "metric": "encode+decode GB/s (uncompressed Arrow bytes) per GPU; 1/2/4/8-GPU scaling",
            "value": round(value, 2), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64 bit patterns (integer/bit work, no arithmetic)",
            "data": "synthetic",
            "config": {"workload": "C2: %d x 1M-row nullable Float64 columns per GPU, 64Ki-row pages, codec %s, "
                                   "inputs resident in HBM" % (B, args.codec),
                       "columns_per_gpu": B, "rows_per_column": ROWS, "page_rows": PAGE,
                       "arrow_bytes_per_step": U, "page_bytes_per_step": page_bytes,
                       "parallelism": "pages of independent columns sharded across %d GPU(s)" % world},
            "roofline": roof, "cpu_baseline": cpu, "kernels": kernels,
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
