#!/usr/bin/env python3
"""bench.py — encode+decode GB/s of uncompressed Arrow bytes on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one batch of synthetic input that is already
resident in HBM: encode the batch (Arrow buffers -> strawboat pages) and decode the pages it
produced back (pages -> Arrow buffers), through the C ABI of libstrawboat_hip.so.

Headline workload at N=1 (BASELINE.json configs[1], "C2"): columns of 1 M-row nullable Float64,
64 Ki-row pages, value = float(k) with k piecewise constant (run length ~ Geometric(mean 32),
k uniform in [0,256)), 10 % nulls.  The batch is `--columns` such columns (default 512 =
512 M rows, 4.16 GB of Arrow bytes, 8192 pages): far beyond the 256 MB Infinity Cache
(SURVEY.md §8d), and enough pages that the one-workgroup-per-page kernels run several rounds per
CU.  Mode "adaptive": default_compress_ratio = 2.0 and the codec of every page is chosen on the
device by the reference's selector (RLE for this data; the CPU oracle agrees).  Nothing is in
forbidden_compressions: every codec of the reference is a candidate, as with its default options.

`configs` (same JSON line, N=1 only): every other BASELINE.json configuration and the reference's
own bench shapes (benches/write_strawboat.rs:30-67), each with encode / decode GB/s of Arrow bytes,
the fraction of the HBM roofline its algorithmic bytes reach per direction, the kernel that
dominates each direction, and the CPU restatement timed on this box (1 thread and all cores).
Every entry lists the five largest kernels of each direction, `page_bytes_vs_reference` (this library's page bytes
over the bytes of the same pages written with the box's liblz4 / libzstd), and — `c3_lz4_reference_written`,
`c5.leaf_pages_reference_written` — the decode of pages the reference's codecs wrote (one LZ4 block / one libzstd
frame per buffer).  `config.summary` repeats one line per configuration inside the keys the driver keeps.
C2 is the FRIENDLIEST configuration (16x compressible runs); the LZ4 / Dict / nested ones are
one to two orders of magnitude slower per Arrow byte — read `configs`, not only `value`.

Multi-GPU (torchrun, one rank per GPU): every rank owns its own `--columns` columns (weak
scaling, pages of independent columns shard with no data-path collective; `configs.c4_sharded` / `c5_sharded`: the two
configurations BASELINE names for 8 GPUs, work items over the N ranks, strong scaling); the only collective
is one RCCL all_gather of the page metas at the end of the timed region (the metadata a file
writer needs for ColumnMeta offsets).
"""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import workloads as W  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (MI355X_MICROARCH.md: 8 TB/s spec, ~6.3 TB/s achievable)
ROWS = 1_000_000
PAGE = 65536
METRIC = "encode+decode GB/s (uncompressed Arrow bytes) per GPU; 1/2/4/8-GPU scaling"


def gen_c2_column(seed, rows=ROWS):
    return W.c2_values(seed, rows)


def host_cores():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def gen_parallel(fn, seeds):
    with ThreadPoolExecutor(max_workers=min(16, host_cores())) as ex:
        return list(ex.map(fn, seeds))


def cpu_baseline(cols, sbo_opts, arrow_bytes, sample_desc, rep_for_all_cores=True, all_cores=True, iters=2):
    """CPU restatement (oracle) over the (column, page) items of `cols`: 1 thread — how the reference runs —
    and page-parallel over all host cores.  Bounded: the sample is sized by the caller."""
    from oracle import sbo
    cores = host_cores()
    sysc = sbo.system_codecs(True)   # Basic(LZ4 / Zstd) blocks through the box's liblz4 / libzstd
    tw, tr, _ = sbo.time_pages_mt(cols, sbo_opts, threads=1, iters=iters)
    one = {"value": round(2.0 * arrow_bytes / (tw + tr) / 1e9, 3), "encode": round(arrow_bytes / tw / 1e9, 3),
           "decode": round(arrow_bytes / tr / 1e9, 3)}
    rep = 1
    if not all_cores:   # (a one-page column is ONE work item for the reference: there is no all-cores leg)
        sbo.system_codecs(False)
        return {"value": one["value"], "unit": "GB/s", "cores": 1, "kind": "port", "sample": sample_desc, "one_thread": one,
                "all_cores": None, "cpu_model": cpu_model()}
    if rep_for_all_cores:
        ps = sbo_opts.max_page_size or max(c["rows"] for c in cols)
        pages = sum((c["rows"] + ps - 1) // ps for c in cols)
        rep = max(1, -(-4 * cores // max(pages, 1)))   # >= 4 work items per core
        rep = min(rep, 64)
    twm, trm, _ = sbo.time_pages_mt(cols * rep, sbo_opts, threads=cores, iters=2)
    allc = {"value": round(2.0 * arrow_bytes * rep / (twm + trm) / 1e9, 3), "encode": round(arrow_bytes * rep / twm / 1e9, 3),
            "decode": round(arrow_bytes * rep / trm / 1e9, 3), "cores": cores, "sample_replicas": rep}
    sbo.system_codecs(False)
    return {"value": one["value"], "unit": "GB/s", "cores": 1, "kind": "port", "sample": sample_desc,
            "one_thread": one, "all_cores": allc, "cpu_model": cpu_model(),
            "block_codecs": {"lz4": "liblz4 %s (the box's liblz4.so.1)" % sysc["lz4"] if sysc["lz4"] else "the restatement's own LZ4 (== LZ4_compress_default bytes)",
                             "zstd": "libzstd %s (the box's libzstd.so.1, default level)" % sysc["zstd"] if sysc["zstd"] else "the restatement's own Zstd (store-only encoder)",
                             "snappy": "the restatement's own Snappy"},
            "note": "C++ restatement of sundy-li/strawboat's algorithm (oracle/), not the Rust binary; Basic(LZ4 / Zstd) blocks go "
                    "through the box's liblz4 / libzstd when they load (block_codecs says which)"}


def reference_page_bytes(cols, sbo_opts):
    """bytes of the pages the reference's CPU path writes for these columns: the restatement with the box's liblz4 / libzstd
    behind Basic(LZ4 / Zstd) blocks (one LZ4 block / one Zstd frame per buffer, src/compression/basic.rs:108-135), all host
    cores, untimed setup.  None when neither library loads."""
    from oracle import sbo
    sysc = sbo.system_codecs(True)
    try:
        if not (sysc["lz4"] or sysc["zstd"]):
            return None, sysc
        _, _, nbytes = sbo.time_pages_mt(cols, sbo_opts, threads=host_cores(), iters=1)
        return nbytes, sysc
    finally:
        sbo.system_codecs(False)


class GpuHarness:
    def __init__(self, ctx):
        import torch
        self.torch = torch
        self.ctx = ctx
        self.dev = ctx.torch_device

    def up(self, a):
        if a is None:
            return None
        a = np.ascontiguousarray(a)
        return self.torch.from_numpy(a.view(np.uint8).reshape(-1)).to(self.dev)

    def dcol(self, col):
        from strawboat_amd import write
        return write.DeviceColumn(col["ptype"], col["nullable"], col["rows"], self.up(col["values"]),
                                  self.up(col["validity"]), self.up(col["offsets"]))

    def check_round_trip(self, col, dec):
        """decoded buffers == the Arrow buffers they came from (null slots of primitives excepted: RLE / Dict / Freq
        pages do not keep them, like the reference)"""
        torch = self.torch
        n = col["rows"]
        if col["nullable"] and col["validity"] is not None:
            assert np.array_equal(dec.validity_numpy(), col["validity"][:(n + 7) // 8]), "validity round trip failed"
        if col["offsets"] is not None:
            go, gv = dec.offsets_numpy().view(np.int32).astype(np.int64), dec.values_numpy()
            ro, rv = col["offsets"].astype(np.int64), col["values"]
            if col["validity"] is None:
                assert np.array_equal(go, ro) and np.array_equal(gv, rv), "binary round trip failed"
                return
            m = np.unpackbits(col["validity"], bitorder="little")[:n].astype(bool)   # null slots: Dict pages repeat a neighbour
            assert np.array_equal((go[1:] - go[:-1])[m], (ro[1:] - ro[:-1])[m]), "string lengths round trip failed"
            rows = np.flatnonzero(m)[::max(1, int(m.sum()) // 2000)]
            for r in rows:
                assert np.array_equal(gv[go[r]:go[r + 1]], rv[ro[r]:ro[r + 1]]), "string bytes round trip failed (row %d)" % r
            return
        if col["ptype"] == W.T_BOOL:
            got = np.unpackbits(dec.values_numpy(), bitorder="little")[:n]
            ref = np.unpackbits(col["values"], bitorder="little")[:n]
        else:
            got = dec.values_numpy().view(col["values"].dtype)
            ref = col["values"]
        if col["validity"] is not None:
            m = np.unpackbits(col["validity"], bitorder="little")[:n].astype(bool)
            got, ref = got[m], ref[m]
        assert np.array_equal(got.view(np.uint8), np.ascontiguousarray(ref).view(np.uint8)), "value round trip failed"
        del torch

    def measure_flat(self, cols, opts, reps=5, check=1):
        """encode / decode of a batch of flat columns: ms per direction (HIP events on the context's stream), Arrow and
        page bytes, per-kernel HIP-event times of one profiled pass"""
        from strawboat_amd import read, write
        torch, ctx = self.torch, self.ctx
        dc = [self.dcol(c) for c in cols]
        U = sum(W.arrow_bytes(c) for c in cols)
        enc = write.encode_columns(ctx, dc, opts)
        ctx.synchronize()
        pages = [read.ColumnPages(c["ptype"], c["nullable"], e.pages, e.metas_array()) for c, e in zip(cols, enc)]
        dec = read.batch_read_columns(ctx, pages)
        ctx.synchronize()
        for i in range(min(check, len(cols))):
            self.check_round_trip(cols[i], dec[i])
        pb = sum(e.length for e in enc)
        npages = sum(e.n_pages for e in enc)
        wb, rb = write.WriteBatch(ctx, dc, opts, out=enc), read.ReadBatch(ctx, pages, out=dec)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        # warm-up: a context keeps eight pinned staging slots, each grown (hipHostMalloc, ~0.3 ms for the 4 MB of tables a
        # 65 536-page batch uploads) the first time a call of this size lands on it — once per context and size, not per step
        for _ in range(4):
            wb.enqueue()
            rb.enqueue()
        ctx.synchronize()
        with torch.cuda.stream(ctx.torch_stream):
            ev[0].record()
            for _ in range(reps):
                wb.enqueue()
            ev[1].record()
            for _ in range(reps):
                rb.enqueue()
            ev[2].record()
        ctx.synchronize()
        te, td = ev[0].elapsed_time(ev[1]) / reps, ev[1].elapsed_time(ev[2]) / reps
        ctx.profile(True)
        wb.enqueue()
        rb.enqueue()
        ctx.synchronize()
        st = ctx.profile_read()
        ctx.profile(False)
        return dict(U=U, page_bytes=pb, n_pages=npages, enc_ms=te, dec_ms=td, kernels=st, enc=enc)


def cold_leg(col, o):
    """A one-page column in a context whose last call with the same plan wrote a PLAIN page (random values of the same shape):
    the launch hints are wrong for it, the page is left undone and the interval is issued again (sb_ctx_replays) — wall ms of
    that first write / read (host clock, synchronize included) next to the second call's.  Before round 6 such a call fell
    to the one-workgroup kernels (75 ms for a Freq page, 832 ms for a 68 MB LZ4 block)."""
    import strawboat_amd as sb
    from strawboat_amd import read, write
    ctx2 = sb.Context(0)
    h2 = GpuHarness(ctx2)
    rng = np.random.default_rng(99)
    v = np.asarray(col["values"])
    prime = dict(col)
    prime["values"] = (rng.integers(0, 256, v.nbytes, dtype=np.uint8).view(v.dtype) if col["offsets"] is None
                       else rng.integers(97, 123, v.size, dtype=np.uint8))
    out = {}
    for name, c in (("prime", prime), ("first", col), ("second", col)):
        dc = [h2.dcol(c)]
        ctx2.synchronize()
        t0 = time.perf_counter()
        enc = write.encode_columns(ctx2, dc, o)
        ctx2.synchronize()
        te = (time.perf_counter() - t0) * 1e3
        pages = [read.ColumnPages(c["ptype"], c["nullable"], enc[0].pages, enc[0].metas_array())]
        t0 = time.perf_counter()
        dec = read.batch_read_columns(ctx2, pages)
        ctx2.synchronize()
        out[name] = [round(te, 3), round((time.perf_counter() - t0) * 1e3, 3)]
        del dec
    return {"first_call_after_a_plain_page_ms": out["first"], "second_call_ms": out["second"], "replays": ctx2.replays(),
            "note": "host wall clock incl. synchronize; [write, read]"}


def measure_reference_pages(h, cols, sbo_opts, reps=3, check_all=False):
    """decode of pages the REFERENCE's codecs wrote: the restatement with the box's liblz4 / libzstd writes the columns'
    pages in the untimed setup (one LZ4 block / one libzstd frame per buffer, src/compression/basic.rs:108-135), the device
    reads them back; returns None when the libraries do not load"""
    from oracle import sbo
    from strawboat_amd import read
    torch, ctx = h.torch, h.ctx
    sysc = sbo.system_codecs(True)
    try:
        if not (sysc["lz4"] and sysc["zstd"]):
            return None

        def one(c):
            return sbo.write_column(c["ptype"], c["nullable"], c["rows"], c["values"], validity=c["validity"], offsets=c["offsets"], options=sbo_opts)
        with ThreadPoolExecutor(max_workers=min(32, host_cores())) as ex:
            written = list(ex.map(one, cols))
    finally:
        sbo.system_codecs(False)
    pages = [read.ColumnPages(c["ptype"], c["nullable"], torch.from_numpy(p).to(h.dev), m) for c, (p, m) in zip(cols, written)]
    dec = read.batch_read_columns(ctx, pages)
    ctx.synchronize()
    for c, d in zip(cols if check_all else cols[:2], dec):
        h.check_round_trip(c, d)
    rb = read.ReadBatch(ctx, pages, out=dec)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    rb.enqueue()
    ctx.synchronize()
    with torch.cuda.stream(ctx.torch_stream):
        ev[0].record()
        for _ in range(reps):
            rb.enqueue()
        ev[1].record()
    ctx.synchronize()
    td = ev[0].elapsed_time(ev[1]) / reps
    ctx.profile(True)
    rb.enqueue()
    ctx.synchronize()
    st = ctx.profile_read()
    ctx.profile(False)
    U = sum(W.arrow_bytes(c) for c in cols)
    pb = sum(int(p.size) for p, _ in written)
    d = direction_summary(U, pb, td, st, False)
    return {"arrow_MB": round(U / 1e6, 1), "page_MB": round(pb / 1e6, 1), "pages": sum(len(m) for _, m in written), "decode": d,
            "written_by": "the restatement with liblz4 %s / libzstd %s (untimed setup)" % (sysc["lz4"], sysc["zstd"])}


def run_host_boundary(h, which, ncols=64, reps=3):
    """The drop-in boundary as the reference's callers see it: HOST Arrow buffers in, host page bytes out (write::write,
    src/write/serialize.rs:36-49) and host pages in, host Arrow buffers out (read_simple, src/read/batch_read.rs:27-64), through
    sb_write_columns / sb_read_columns with SB_MEM_HOST.  PCIe-inclusive: never `value`.  which = "c2" (nullable f64, adaptive
    -> RLE pages) or "c1" (Int64, one page per column, no compression)."""
    import ctypes as C
    import torch
    from strawboat_amd import _native as N
    from strawboat_amd.types import WriteOptions
    from strawboat_amd.write import options_c
    ctx = h.ctx
    lib, hh = ctx._lib, ctx._h
    if which == "c2":
        oc = options_c(WriteOptions(max_page_size=PAGE, default_compress_ratio=2.0))
        ptype, nullable = W.T_F64, 1
        gen = gen_parallel(gen_c2_column, [7000 + b for b in range(ncols)])
    else:
        from strawboat_amd.types import Compression
        oc = options_c(WriteOptions(max_page_size=None, force_codec=Compression.NONE))
        ptype, nullable = W.T_I64, 0
        rng = np.random.default_rng(7)
        gen = [(rng.integers(-2**62, 2**62, ROWS), None) for _ in range(ncols)]
    npg = C.c_uint64()
    bound = lib.sb_write_bound(ptype, nullable, ROWS, 0, C.byref(oc), C.byref(npg))

    def pinned(n):
        return torch.zeros(n, dtype=torch.uint8).pin_memory()
    vals, valids, outs, metas, vouts, bouts = [], [], [], [], [], []
    for v, m in gen:
        tv = pinned(ROWS * 8)
        tv.numpy()[:] = np.ascontiguousarray(v).view(np.uint8)
        vals.append(tv)
        if nullable:
            tb = pinned(m.size)
            tb.numpy()[:] = m
            valids.append(tb)
            bouts.append(pinned((ROWS + 31) // 32 * 4))
        outs.append(pinned(bound))
        metas.append((N.PageMetaC * npg.value)())
        vouts.append(pinned(ROWS * 8))
    cw = (N.ColumnWriteC * ncols)()
    for c in range(ncols):
        cw[c].physical_type, cw[c].is_nullable, cw[c].rows = ptype, nullable, ROWS
        cw[c].values = vals[c].data_ptr()
        if nullable:
            cw[c].validity = valids[c].data_ptr()
        cw[c].out_pages, cw[c].out_capacity = outs[c].data_ptr(), bound
        cw[c].out_metas, cw[c].n_pages_capacity = metas[c], npg.value
    te = td = 1e9
    for _ in range(reps + 1):
        t = time.perf_counter()
        ctx._check(lib.sb_write_columns(hh, cw, ncols, C.byref(oc), N.SB_MEM_HOST))
        ctx.synchronize()
        te = min(te, time.perf_counter() - t)
        cr = (N.ColumnReadC * ncols)()
        for c in range(ncols):
            cr[c].physical_type, cr[c].is_nullable = ptype, nullable
            cr[c].pages, cr[c].pages_len = outs[c].data_ptr(), int(cw[c].out_len)
            cr[c].metas, cr[c].n_pages = metas[c], int(cw[c].n_pages)
            cr[c].values, cr[c].values_capacity = vouts[c].data_ptr(), ROWS * 8
            if nullable:
                cr[c].validity, cr[c].validity_capacity = bouts[c].data_ptr(), bouts[c].numel()
        t = time.perf_counter()
        ctx._check(lib.sb_read_columns(hh, cr, ncols, N.SB_MEM_HOST))
        ctx.synchronize()
        td = min(td, time.perf_counter() - t)
    if nullable:
        m = np.unpackbits(valids[0].numpy(), bitorder="little")[:ROWS].astype(bool)
        assert np.array_equal(vouts[0].numpy().view(np.float64)[m], vals[0].numpy().view(np.float64)[m]), "host-boundary round trip failed"
    else:
        assert np.array_equal(vouts[0].numpy(), vals[0].numpy()), "host-boundary round trip failed"
    arrow = ncols * (ROWS * 8 + ((ROWS + 7) // 8 if nullable else 0))
    pages = sum(int(cw[c].out_len) for c in range(ncols))
    return {"columns": ncols, "arrow_MB": round(arrow / 1e6, 1), "page_MB": round(pages / 1e6, 1),
            "encode_GBps": round(arrow / te / 1e9, 1), "decode_GBps": round(arrow / td / 1e9, 1),
            "encode_ms": round(te * 1e3, 2), "decode_ms": round(td * 1e3, 2),
            # what crosses the link per direction: Arrow bytes one way, page bytes the other
            "pcie_GBps_encode": round((arrow + pages) / te / 1e9, 1), "pcie_GBps_decode": round((arrow + pages) / td / 1e9, 1)}


def direction_summary(U, pb, ms, kernels, encode):
    """GB/s of Arrow bytes, fraction of the HBM roofline reached by the direction's algorithmic bytes
    (A_enc = Arrow bytes read + page bytes written, A_dec = page bytes read + Arrow bytes written),
    and the kernel that dominates the direction"""
    ks = {k: v for k, v in kernels.items() if k.startswith(("k_enc", "k_sel_", "k_rle_big", "k_dict_big", "k_freq_big", "k_nested_big")) == encode}
    tot = sum(v[1] for v in ks.values()) or 1.0
    top = max(ks.items(), key=lambda kv: kv[1][1]) if ks else (None, (0, 0.0))
    A = U + pb
    return {"GBps": round(U / ms / 1e6, 1), "ms": round(ms, 3), "frac_hbm": round(A / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "top_kernel": top[0], "top_kernel_ms": round(top[1][1], 3), "top_kernel_share": round(top[1][1] / tot, 3),
            "kernels_ms": {k: round(v[1], 3) for k, v in sorted(ks.items(), key=lambda kv: -kv[1][1])[:8]}}


def config_entry(name, res, cpu, extra=None):
    U, pb = res["U"], res["page_bytes"]
    e = {"arrow_MB": round(U / 1e6, 1), "page_MB": round(pb / 1e6, 1), "pages": res["n_pages"],
         "encdec_GBps": round(2.0 * U / (res["enc_ms"] + res["dec_ms"]) / 1e6, 1),
         "encode": direction_summary(U, pb, res["enc_ms"], res["kernels"], True),
         "decode": direction_summary(U, pb, res["dec_ms"], res["kernels"], False),
         "cpu_baseline": cpu}
    if extra:
        e.update(extra)
    return e


def sbo_options(opts):
    from oracle import sbo
    return sbo.make_options(default_compression=opts.default_compression, ratio=opts.default_compress_ratio,
                            max_page_size=opts.max_page_size, forbidden=tuple(opts.forbidden_compressions),
                            force_codec=opts.force_codec, force_index_codec=opts.force_index_codec,
                            rng_seed=opts.rng_seed)


def page_codecs(col, enc):
    """codec ids of a column's pages (page inspector, strawboat_amd.stat)"""
    from strawboat_amd import stat
    from strawboat_amd.types import Compression
    ids = set()
    pages = enc.pages_numpy()
    off = 0
    for m in enc.metas:
        info = stat.stat_page(pages[off:off + m.length], col["ptype"], col["nullable"])
        chain = []
        while info is not None:
            chain.append(Compression.NAMES.get(info.codec, str(info.codec)))
            info = info.body.indices or info.body.exceptions
        ids.add(">".join(chain))
        off += m.length
    return sorted(ids)


def run_configs(h, only, cpu_on, log):
    """every BASELINE.json configuration but the headline, and the reference's bench shapes, on one GPU"""
    from strawboat_amd.types import Compression as C, WriteOptions
    out = {}

    def want(k):
        return only is None or k in only

    def flat(key, desc, cols, opts, cpu_cols, reps=5, codecs=True):
        t0 = time.time()
        res = h.measure_flat(cols, opts, reps=reps)
        cpu = None
        if cpu_on:
            cu = sum(W.arrow_bytes(c) for c in cpu_cols)
            cpu = cpu_baseline(cpu_cols, sbo_options(opts), cu,
                               "%d column(s) x %d rows of this configuration, encode+decode" % (len(cpu_cols), cpu_cols[0]["rows"]))
        extra = {"workload": desc}
        if cpu_on:   # the size the reference's own codecs (liblz4 / libzstd) give the same pages
            rb, sysc = reference_page_bytes(cols, sbo_options(opts))
            if rb:
                extra["page_MB_reference_codecs"] = round(rb / 1e6, 1)
                extra["page_bytes_vs_reference"] = round(res["page_bytes"] / rb, 4)
                extra["reference_codecs"] = "liblz4 %s, libzstd %s" % (sysc["lz4"], sysc["zstd"])
        if codecs:
            try:
                extra["codecs_of_column_0"] = page_codecs(cols[0], res["enc"][0])
            except Exception as e:  # the inspector is informative only
                extra["codecs_of_column_0"] = "n/a (%s)" % type(e).__name__
        out[key] = config_entry(key, res, cpu, extra)
        log("%s: encode %.1f GB/s, decode %.1f GB/s (%.1f s)" % (key, out[key]["encode"]["GBps"], out[key]["decode"]["GBps"], time.time() - t0))

    if want("c1"):
        cols = gen_parallel(W.c1_int64, range(42, 42 + 128))
        flat("c1", "C1: 128 x 1M-row non-nullable Int64, one page per column, no compression (the north-star "
                   "'1 M-row primitive-page decode' shape)", cols, WriteOptions(), cols[:2], reps=10)
    if want("c3") or want("c3_lz4") or want("c3_512"):
        cols = gen_parallel(lambda s: W.zipf_utf8(ROWS, s), range(42, 42 + 64))
        if want("c3"):
            flat("c3", "C3: 64 x 1M-row Utf8 (zipf 1.1 over 10 000 words of 4..24 B), 64Ki-row pages, LZ4 default, ratio 2.0 "
                       "-> Dict pages with Bitpacking / LZ4 indices", cols,
                 WriteOptions(max_page_size=PAGE, default_compression=C.LZ4, default_compress_ratio=2.0), cols[:1], reps=3)
        if want("c3_512"):   # the same 64 columns eight times over: C3 at the headline's batch size (8 192 pages), so that the
            # per-kernel floors of a 1 024-page batch and the throughput can be told apart
            flat("c3_512", "C3 x 8: 512 x 1M-row Utf8 columns (the 64 columns of c3 eight times), same options", cols * 8,
                 WriteOptions(max_page_size=PAGE, default_compression=C.LZ4, default_compress_ratio=2.0), cols[:1], reps=2, codecs=False)
        if want("c3_lz4"):
            o_lz4 = WriteOptions(max_page_size=PAGE, default_compression=C.LZ4)
            flat("c3_lz4", "C3': the same columns, Basic(LZ4) pages (offsets block + values block), ratio None", cols, o_lz4, cols[:1], reps=3)
            if cpu_on:
                r = measure_reference_pages(h, cols, sbo_options(o_lz4))
                if r:
                    r["workload"] = "C3' pages written by liblz4 (one block per buffer, 64 KiB match window), decoded on the device"
                    out["c3_lz4_reference_written"] = r
                    log("c3_lz4_reference_written: decode %.1f GB/s" % r["decode"]["GBps"])
        del cols
    if want("c4"):
        named = W.c4_columns(10_000_000)
        cols = [c for _, c in named]
        opts = WriteOptions(max_page_size=PAGE, default_compression=C.LZ4, default_compress_ratio=2.0)
        cpu_cols = [dict(c, rows=ROWS, values=(c["values"][:ROWS] if c["offsets"] is None and c["ptype"] != W.T_BOOL else c["values"]),
                         offsets=None if c["offsets"] is None else c["offsets"][:ROWS + 1]) for c in cols[::2]]
        for c in cpu_cols:
            if c["offsets"] is not None:
                c["values"] = c["values"][:int(c["offsets"][-1])]
        flat("c4", "C4: the 8-column mixed schema {Int32 x2, Float64 x2, Utf8 x2, Boolean x2} x 10 M rows (153 pages per "
                   "column), LZ4 default, ratio 2.0, all 8 columns on this GPU; CPU sample = the first 1 M rows of one column "
                   "per type", cols, opts, cpu_cols, reps=3, codecs=False)
        per = {}
        for (nm, c) in named[::2]:
            r = h.measure_flat([c], opts, reps=3, check=0)
            per[nm.split("_")[0]] = {"encode_GBps": round(r["U"] / r["enc_ms"] / 1e6, 1), "decode_GBps": round(r["U"] / r["dec_ms"] / 1e6, 1),
                                     "codecs": page_codecs(c, r["enc"][0])}
        out["c4"]["per_column_type"] = per
        del named, cols
    if want("one_page"):
        # the reference's default paging (WriteOptions.max_page_size = None, src/write/common.rs:54-58): a column is ONE page.
        # Pages this long run the section-parallel selector (sb_select_big.h) and many-workgroup plain / LZ4 / Zstd writers;
        # Dict / RLE / Freq pages and a one-block LZ4 buffer are still one workgroup's work.
        rng = np.random.default_rng(7)
        n64 = 12_000_000
        i64 = dict(ptype=W.T_I64, nullable=False, rows=n64, values=np.sort(rng.integers(0, 1 << 40, n64)).astype(np.int64), validity=None, offsets=None)
        utf8 = W.zipf_utf8(3_000_000, 42)
        runs = dict(i64, values=np.repeat(rng.integers(0, 200, n64 // 50 + 1), 50)[:n64].astype(np.int64))
        lowc = dict(ptype=W.T_I32, nullable=False, rows=n64, values=rng.integers(0, 500, n64).astype(np.int32), validity=None, offsets=None)
        sp = np.full(n64, 1_000_000, dtype=np.int32)
        spi = rng.random(n64) < 0.02
        sp[spi] = rng.integers(0, 1 << 30, int(spi.sum())).astype(np.int32)
        sparse = dict(lowc, values=sp)
        one = {}
        for nm, col, o, desc in (
                ("int64_adaptive", i64, WriteOptions(default_compress_ratio=2.0), "sorted Int64 (40-bit), adaptive (ratio 2.0), no default compression -> a plain page"),
                ("int64_runs_adaptive", runs, WriteOptions(default_compress_ratio=2.0), "Int64 in runs of 50 rows, adaptive -> ONE RLE page (written and read section-parallel)"),
                ("int64_zstd", i64, WriteOptions(default_compression=C.ZSTD), "the same column, Basic(Zstd)"),
                ("int64_lz4", i64, WriteOptions(default_compression=C.LZ4), "the same column, Basic(LZ4): one LZ4 block of 68 MB, decoded by one workgroup"),
                ("utf8_zstd", utf8, WriteOptions(default_compression=C.ZSTD), "Utf8 (zipf over 10 000 words), Basic(Zstd)"),
                ("utf8_adaptive", utf8, WriteOptions(default_compress_ratio=2.0), "the same column, adaptive -> one Dict page"),
                ("int32_lowcard_adaptive", lowc, WriteOptions(default_compress_ratio=2.0), "Int32 with 500 distinct values, adaptive -> one Dict page"),
                ("int32_sparse_adaptive", sparse, WriteOptions(default_compress_ratio=2.0), "Int32, 98 % one value, adaptive -> one Freq page")):
            res = h.measure_flat([col], o, reps=3, check=1)
            cpu = None
            if cpu_on:   # the reference works through a page on ONE thread
                cpu = cpu_baseline([col], sbo_options(o), W.arrow_bytes(col), "the same one-page column, encode+decode, 1 thread", all_cores=False, iters=1)
            one[nm] = config_entry(nm, res, cpu, {"workload": "ONE page of %d rows: %s" % (col["rows"], desc)})
            try:
                one[nm]["cold"] = cold_leg(col, o)
            except Exception as e:
                one[nm]["cold"] = {"error": "%s: %s" % (type(e).__name__, e)}
            log("one_page %s: encode %.1f GB/s, decode %.1f GB/s" % (nm, one[nm]["encode"]["GBps"], one[nm]["decode"]["GBps"]))
        out["one_page"] = one
        del i64, utf8, runs, lowc, sparse, sp
    if want("host_boundary"):
        try:
            out["host_boundary"] = {"c2": run_host_boundary(h, "c2"), "c1": run_host_boundary(h, "c1"),
                                    "note": "SB_MEM_HOST: pinned host Arrow buffers <-> pinned host page bytes, wall time of the calls incl. "
                                            "both PCIe directions; the link is PCIe Gen5 x16, 63 GB/s per direction by spec (never `value`)",
                                    "pcie_peak_GBps": 63.0}
            log("host_boundary: c2 %.1f / %.1f GB/s, c1 %.1f / %.1f GB/s" % (out["host_boundary"]["c2"]["encode_GBps"], out["host_boundary"]["c2"]["decode_GBps"],
                                                                               out["host_boundary"]["c1"]["encode_GBps"], out["host_boundary"]["c1"]["decode_GBps"]))
        except Exception as e:   # (a bench leg must not take the headline line down)
            out["host_boundary"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if want("c5"):
        out["c5"] = run_c5(h, cpu_on)
        log("c5: encode %.2f GB/s, decode %.2f GB/s" % (out["c5"]["encode"]["GBps"], out["c5"]["decode"]["GBps"]))
    if want("continuity"):
        o = WriteOptions(max_page_size=8192, default_compression=C.LZ4)
        cont = {}
        for nm, fn, B in (("bool", W.cont_bool, 512), ("utf8", W.cont_utf8, 128), ("i64", W.cont_i64, 128)):
            cols = gen_parallel(lambda s: fn(1 << 20, s), range(42, 42 + B))
            res = h.measure_flat(cols, o, reps=3)
            cpu = None
            if cpu_on:
                cpu = cpu_baseline(cols[:1], sbo_options(o), W.arrow_bytes(cols[0]), "1 column x 2^20 rows, encode+decode")
            e = config_entry(nm, res, cpu, {"workload": "%d x 2^20-row %s columns, nullable field, LZ4, page 8192, ratio None "
                                                        "(benches/write_strawboat.rs:30-67), batched" % (B, nm)})
            lat = {}
            for p in range(10, 21, 2):   # the reference's own sizes, one column per call: wall ms per write (criterion's unit)
                c1 = fn(1 << p, 42)
                r = h.measure_flat([c1], o, reps=5, check=1)
                lat["2^%d" % p] = {"write_ms": round(r["enc_ms"], 3), "read_ms": round(r["dec_ms"], 3)}
            e["single_column_latency"] = lat
            cont[nm] = e
            log("continuity %s: encode %.1f GB/s, decode %.1f GB/s" % (nm, e["encode"]["GBps"], e["decode"]["GBps"]))
            del cols
        out["continuity"] = cont
    return out


def run_c5(h, cpu_on, arrays=64):
    """C5: `arrays` x (1 M-row List<Struct<Int64,Utf8>>), Zstd default, ratio None: per array 2 leaf columns x 16 pages
    through the nested API — the level sections of ALL leaves in one batch of launches (one host round trip for the page
    cut), then the leaf BLOCKs of all leaves through the flat path in one call.  Wall time of the calls (they synchronise),
    plus the one-array call as `single_array` (the round-2 shape: latency of a 22.7 MB job)."""
    from strawboat_amd import nested
    from strawboat_amd.read import ColumnPages
    from strawboat_amd.types import Compression as C, WriteOptions
    ctx = h.ctx
    opts = WriteOptions(max_page_size=PAGE, default_compression=C.ZSTD)

    def dlevels(levels):
        return [nested.NestedLevel(lv["kind"], bool(lv["is_optional"]), lv["length"], h.up(lv.get("validity")), h.up(lv.get("offsets")))
                for lv in levels]
    gen = gen_parallel(lambda s: W.c5_nested(seed=s), range(42, 42 + arrays))
    items, U = [], 0
    for la, a, lb, b in gen:
        rows = la[0]["length"]
        for lv_, leaf in ((la, a), (lb, b)):
            dc = h.dcol(leaf)
            dc.is_nullable = False
            items.append((dlevels(lv_), dc, leaf, lv_))
        U += W.arrow_bytes(a) + W.arrow_bytes(b) + 2 * ((rows + 1) * 4 + (rows + 7) // 8)   # leaves + list offsets / validity per leaf path
    la, a, lb, b = gen[0]
    del gen

    def measure(its, reps):
        pairs = [(dl, dc) for dl, dc, _, _ in its]
        wb = nested.NestedWriteBatch(ctx, pairs, opts)      # descriptors and buffers once, like WriteBatch for flat columns
        encs = wb.run()                                     # warm-up + outputs
        tes = []
        for _ in range(reps):                               # (a run ends in a synchronize: timed one by one, the MEDIAN reported —
            t0 = time.perf_counter()                        #  one run in ~30 of a day's runs stalled for 0.3 s on the host side)
            encs = wb.run()
            tes.append((time.perf_counter() - t0) * 1e3)
        te = sorted(tes)[len(tes) // 2]
        cps = [ColumnPages(c["ptype"], False, e.pages[:e.length].contiguous(), e.metas_array()) for e, (_, _, c, _) in zip(encs, its)]
        kinds = [[lv["kind"] for lv in lv_] for _, _, _, lv_ in its]
        opt = [[bool(lv["is_optional"]) for lv in lv_] for _, _, _, lv_ in its]
        rb = nested.NestedReadBatch(ctx, cps, kinds, opt)
        arrs = rb.run()
        tds = []
        for _ in range(reps):
            t0 = time.perf_counter()
            arrs = rb.run()
            tds.append((time.perf_counter() - t0) * 1e3)
        td = sorted(tds)[len(tds) // 2]
        measure.spread = {"encode_ms_runs": [round(x, 3) for x in tes], "decode_ms_runs": [round(x, 3) for x in tds]}
        return pairs, encs, cps, kinds, opt, arrs, te, td, wb, rb
    pairs, encs, cps, kinds, opt, arrs, te, td, wb, rb = measure(items, 3)
    # round trip: list offsets and leaf buffers of the first array
    assert np.array_equal(arrs[0].offsets_numpy(0), la[0]["offsets"].astype(np.int64)), "C5 list offsets round trip failed"
    assert np.array_equal(arrs[1].leaf.values_numpy(), b["values"]), "C5 Utf8 leaf round trip failed"
    m = np.unpackbits(a["validity"], bitorder="little")[:a["rows"]].astype(bool)
    assert np.array_equal(arrs[0].leaf.values_numpy().view(np.int64)[m], a["values"][m]), "C5 Int64 leaf round trip failed"
    ctx.profile(True)
    wb.run()
    st_e = ctx.profile_read()
    rb.run()
    st_d = ctx.profile_read()       # accumulated since profile(True): take the write's share out
    st_d = {k: (v[0] - st_e.get(k, (0, 0.0))[0], v[1] - st_e.get(k, (0, 0.0))[1]) for k, v in st_d.items()}
    st_d = {k: v for k, v in st_d.items() if v[0] > 0}
    ctx.profile(False)
    pb = sum(e.length for e in encs)
    res = dict(U=U, page_bytes=pb, n_pages=sum(e.n_pages for e in encs), enc_ms=te, dec_ms=td, kernels={})
    spread = getattr(measure, "spread", None)
    cpu = None
    if cpu_on:   # one array: its two leaf columns' blocks + the level sections of their 2 x 16 pages (write_nested_validity / read_validity_nested)
        from oracle import sbo
        o = sbo.make_options(default_compression=sbo.ZSTD, max_page_size=PAGE)
        U1c = W.arrow_bytes(a) + W.arrow_bytes(b) + 2 * ((la[0]["length"] + 1) * 4 + (la[0]["length"] + 7) // 8)
        cpu = cpu_baseline([dict(a, nullable=False), dict(b, nullable=False)], o, U1c,
                           "one array (1 M rows): the blocks of its two leaf columns + the rep / def level sections of their 2 x 16 pages")
        rows1 = la[0]["length"]
        items_l = [(lv_, r0, min(PAGE, rows1 - r0)) for lv_ in (la, lb) for r0 in range(0, rows1, PAGE)]
        kinds_l = [[lv["kind"] for lv in lv_] for lv_ in (la, lb)]
        opt_l = [[int(bool(lv["is_optional"])) for lv in lv_] for lv_ in (la, lb)]

        def lv_write(it):
            return sbo.nested_write_levels(it[0], it[1], it[2])

        def lv_time(threads, rep):
            work = items_l * rep
            t0 = time.perf_counter()
            if threads == 1:
                written = [lv_write(it) for it in work]
            else:
                with ThreadPoolExecutor(max_workers=threads) as ex:
                    written = list(ex.map(lv_write, work, chunksize=4))
            tw_ = time.perf_counter() - t0
            rd = [(w[0], w[1], kinds_l[0 if it[0] is la else 1], opt_l[0 if it[0] is la else 1]) for w, it in zip(written, work)]
            t0 = time.perf_counter()
            if threads == 1:
                for r in rd:
                    sbo.nested_read_levels(*r)
            else:
                with ThreadPoolExecutor(max_workers=threads) as ex:
                    list(ex.map(lambda r: sbo.nested_read_levels(*r), rd, chunksize=4))
            return tw_, time.perf_counter() - t0
        lw1, lr1 = lv_time(1, 1)
        repl = cpu["all_cores"]["sample_replicas"]
        lwm, lrm = lv_time(host_cores(), repl)
        # fold the level times into the leaf times (GB/s of the array's Arrow bytes incl. list offsets / validity)
        def fold(leg, tw_, tr_, n):
            te_ = U1c * n / (leg["encode"] * 1e9) + tw_
            td_ = U1c * n / (leg["decode"] * 1e9) + tr_
            leg.update(encode=round(U1c * n / te_ / 1e9, 3), decode=round(U1c * n / td_ / 1e9, 3), value=round(2.0 * U1c * n / (te_ + td_) / 1e9, 3),
                       level_sections_s={"write": round(tw_, 4), "read": round(tr_, 4)})
        fold(cpu["one_thread"], lw1, lr1, 1)
        fold(cpu["all_cores"], lwm, lrm, repl)
        cpu["value"] = cpu["one_thread"]["value"]
    e = config_entry("c5", res, cpu, {"workload": "C5: %d x 1 M-row List<Struct<Int64,Utf8>> (list length U{0,1,2}, 10 %% null lists, leaves 20 %% "
                                                  "null), 64Ki-row pages, Zstd default, ratio None; %d leaf columns x 16 pages through the nested API "
                                                  "(NestedWriteBatch / NestedReadBatch: descriptors and buffers built once; a run = level sections of all "
                                                  "leaves in one batch, then the BLOCKs of all leaves in one call); wall time of the runs incl. their "
                                                  "host round trips" % (arrays, 2 * arrays)})

    def top(st, direction):
        if st:
            k, v = max(st.items(), key=lambda kv: kv[1][1])
            tot = sum(x[1] for x in st.values()) or 1.0
            e[direction].update(top_kernel=k, top_kernel_ms=round(v[1], 3), top_kernel_share=round(v[1] / tot, 3),
                                kernels_ms={kk: round(vv[1], 3) for kk, vv in sorted(st.items(), key=lambda kv: -kv[1][1])[:6]})
    top(st_e, "encode")
    top(st_d, "decode")
    if spread:
        e["runs_ms"] = spread   # (every timed run; encode / decode above are their medians)
    if cpu_on:
        from oracle import sbo
        o = sbo.make_options(default_compression=sbo.ZSTD, max_page_size=PAGE)
        flat_leaves = [dict(a, nullable=False, validity=None), dict(b, nullable=False, validity=None)]
        rbytes, sysc = reference_page_bytes(flat_leaves, o)
        if rbytes:
            e["leaf_page_MB_reference_codecs_one_array"] = round(rbytes / 1e6, 2)
            e["leaf_page_MB_one_array"] = round((encs[0].length + encs[1].length) / 1e6, 2)
            e["reference_codecs"] = "libzstd %s (one frame per buffer, default level)" % sysc["zstd"]
        # the leaf columns of all arrays written by libzstd (the reference's frames: one per page buffer), read here
        r = measure_reference_pages(h, [dict(it[2], nullable=False, validity=None) for it in items], o)
        if r:
            r["workload"] = "the leaf BLOCKs of the %d arrays as flat columns, pages written by libzstd (ONE frame per page buffer: the device " \
                            "decodes such a frame on one wave), decoded on the device" % arrays
            e["leaf_pages_reference_written"] = r
    if arrays > 1:   # the round-2 shape: one array per call
        _, _, _, _, _, _, te1, td1, _, _ = measure(items[:2], 3)
        U1 = W.arrow_bytes(a) + W.arrow_bytes(b) + 2 * ((la[0]["length"] + 1) * 4 + (la[0]["length"] + 7) // 8)
        e["single_array"] = {"arrow_MB": round(U1 / 1e6, 1), "write_ms": round(te1, 3), "read_ms": round(td1, 3),
                             "encode_GBps": round(U1 / te1 / 1e6, 2), "decode_GBps": round(U1 / td1 / 1e6, 2)}
    return e


def run_c4_sharded(h, world, rank, dist, steps=3, on_device=True):
    """C4 as BASELINE.json states it: the 8-column mixed schema x 10 M rows, (column, page-range) work items dealt to the
    ranks by strawboat_amd.shard.plan_work_items, every rank encodes and decodes its own items with the device codecs,
    ONE all_gather of the page metas (RCCL over xGMI).  Strong scaling: the job is fixed, value = job bytes / max-over-
    ranks time.  Returns the entry for `configs` on rank 0 (None elsewhere)."""
    import torch
    from strawboat_amd import read, shard, write
    from strawboat_amd.types import Compression as C, WriteOptions
    ctx = h.ctx
    rows = 10_000_000
    npages = (rows + PAGE - 1) // PAGE
    # per-column Arrow bytes are a function of the generator alone: every rank computes the same plan without the data
    est = {"int32": rows * 4 + rows // 8, "float64": rows * 8 + rows // 8, "utf8": rows * 18 + rows // 8, "boolean": rows // 4}
    names = ["int32_0", "int32_1", "float64_0", "float64_1", "utf8_0", "utf8_1", "boolean_0", "boolean_1"]
    plan = shard.plan_work_items([(est[n.split("_")[0]], npages) for n in names], world)
    mine = plan[rank]
    need = sorted({it.column for it in mine})
    full = {}
    for ci in need:   # the columns this rank touches (seeds as in workloads.c4_columns)
        kind, j = names[ci].split("_")
        j = int(j)
        if kind == "int32":
            rng = np.random.default_rng(42 + j)
            full[ci] = dict(ptype=W.T_I32, nullable=True, rows=rows, values=rng.integers(0, 1000, rows).astype(np.int32), validity=None, offsets=None)
        elif kind == "float64":
            full[ci] = W.c2_float64(52 + j, rows)
        elif kind == "utf8":
            full[ci] = W.zipf_utf8(rows, 62 + j, null_density=0.1)
        else:
            rng = np.random.default_rng(72 + j)
            full[ci] = dict(ptype=W.T_BOOL, nullable=True, rows=rows, values=W.pack_bits(rng.random(rows) < 0.5),
                            validity=W.pack_bits(rng.random(rows) >= 0.1), offsets=None)
    opts = WriteOptions(max_page_size=PAGE, default_compression=C.LZ4, default_compress_ratio=2.0)
    cdev = h.dev if (world > 1 and on_device) else None

    def all_ok(ok):
        """every rank reaches this: True only if no rank failed (a rank that raised alone would leave the others waiting in
        the next collective)"""
        if world == 1:
            return ok
        f = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=cdev if cdev is not None else "cpu")
        dist.all_reduce(f, op=dist.ReduceOp.MIN)
        return bool(f.item() > 0.5)

    err = None
    try:
        parts = [shard.slice_column(full[it.column], it.first_page, it.n_pages, PAGE) for it in mine]
        U_local = sum(W.arrow_bytes(c) for c in parts)
        dcs = []
        for it, c in zip(mine, parts):
            d = h.dcol(c)
            d.first_page_index = it.first_page
            d.column_values_len = c.get("column_values_len", 0)
            dcs.append(d)
        del full
        enc = write.encode_columns(ctx, dcs, opts)
        ctx.synchronize()
        cps = [read.ColumnPages(c["ptype"], c["nullable"], e.pages, e.metas_array()) for c, e in zip(parts, enc)]
        dec = read.batch_read_columns(ctx, cps)
        ctx.synchronize()
        if parts:
            h.check_round_trip(parts[0], dec[0])
        wb, rb = write.WriteBatch(ctx, dcs, opts, out=enc), read.ReadBatch(ctx, cps, out=dec)
    except Exception as e:   # reported, not fatal: the headline line must not depend on this configuration
        err = "%s: %s" % (type(e).__name__, e)
    if not all_ok(err is None):
        return {"error": err or "another rank failed while preparing its work items"} if rank == 0 else None
    cap = shard.record_capacity(plan)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    items = []
    try:
        for _ in range(steps):
            wb.enqueue()
            rb.enqueue()
        ctx.synchronize()
        items = [(it.column, it.first_page, e.metas_array()) for it, e in zip(mine, enc)]
    except Exception as e:
        err = "%s: %s" % (type(e).__name__, e)
    allm = None
    try:   # (every rank takes part in the collective, whatever happened above; gaps / duplicates are reported after it)
        allm = shard.gather_metas(items, len(names), capacity=cap, device=cdev, expected_pages=[npages] * len(names))
    except ValueError as e:
        err = err or "%s: %s" % (type(e).__name__, e)
    barrier()
    el = time.perf_counter() - t0
    tt = torch.tensor([el, float(U_local)], dtype=torch.float64, device=h.dev if on_device else "cpu")
    if world > 1:
        mx = tt.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = tt.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        el, U = float(mx[0].item()), float(sm[1].item())
    else:
        U = float(U_local)
    if err is None:
        try:
            cm = shard.column_metas(allm)
            assert all(len(m.pages) == npages for m in cm), "every column must come back with all of its pages"
        except Exception as e:
            err = "%s: %s" % (type(e).__name__, e)
    if not all_ok(err is None):
        return {"error": err or "another rank failed in the timed region"} if rank == 0 else None
    if rank != 0:
        return None
    return {"workload": "C4: 8-column mixed schema x 10 M rows, %d (column, page-range) work items over %d GPU(s), LZ4 default, ratio 2.0, "
                        "encode+decode of every item + ONE all_gather of %d page metas" % (sum(len(s) for s in plan), world, len(names) * npages),
            "scaling": "strong", "n_gpus": world, "steps": steps, "arrow_MB": round(U / 1e6, 1),
            "ms_per_step": round(el / steps * 1e3, 3), "encdec_GBps": round(2.0 * U * steps / el / 1e9, 1),
            "items_per_rank": [len(s) for s in plan], "record_capacity": cap}


def run_c5_sharded(h, world, rank, dist, steps=3, on_device=True):
    """C5 as SURVEY 8(e) splits it: 2 leaf columns x 16 pages = 32 (leaf, page range) work items dealt to the ranks; every
    rank writes and reads its own items through the nested API (a page range of top-level rows is a nested column of its
    own: same level sections, same leaf pages), ONE all_gather of the page metas.  Strong scaling, wall time of synchronous
    calls.  Failures are handled like run_c4_sharded's."""
    import torch
    from strawboat_amd import nested, shard
    from strawboat_amd.read import ColumnPages
    from strawboat_amd.types import Compression as C, WriteOptions
    ctx = h.ctx
    rows = 1_000_000
    npages = (rows + PAGE - 1) // PAGE
    cdev = h.dev if (world > 1 and on_device) else None

    def all_ok(ok):
        if world == 1:
            return ok
        f = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=cdev if cdev is not None else "cpu")
        dist.all_reduce(f, op=dist.ReduceOp.MIN)
        return bool(f.item() > 0.5)

    plan = shard.plan_work_items([(rows * 9, npages), (rows * 4, npages)], world)
    mine = plan[rank]
    err, U_local = None, 0.0
    try:
        la, a, lb, b = W.c5_nested()
        opts = WriteOptions(max_page_size=PAGE, default_compression=C.ZSTD)
        pairs, metas_kinds = [], []
        for it in mine:
            levels, leaf = (la, a) if it.column == 0 else (lb, b)
            r0, r1 = it.first_page * PAGE, min(rows, (it.first_page + it.n_pages) * PAGE)
            lv, col = W.c5_slice(levels, leaf, r0, r1)
            dl = [nested.NestedLevel(x["kind"], bool(x["is_optional"]), x["length"], h.up(x.get("validity")), h.up(x.get("offsets"))) for x in lv]
            dc = h.dcol(col)
            dc.is_nullable = False
            dc.first_page_index = it.first_page
            dc.column_values_len = col.get("column_values_len", 0)
            pairs.append((dl, dc))
            metas_kinds.append((col, [x["kind"] for x in lv], [bool(x["is_optional"]) for x in lv]))
            U_local += W.arrow_bytes(col) + (r1 - r0 + 1) * 4 + (r1 - r0 + 7) // 8
        encs = nested.write_nested_leaves(ctx, pairs, opts)
        cps = [ColumnPages(c["ptype"], False, e.pages[:e.length].contiguous(), e.metas_array()) for e, (c, _, _) in zip(encs, metas_kinds)]
        arrs = nested.read_nested_leaves(ctx, cps, [k for _, k, _ in metas_kinds], [o for _, _, o in metas_kinds])
        for arr, (c, _, _) in zip(arrs, metas_kinds):   # leaf round trip of every item
            if c["offsets"] is None:
                m = np.unpackbits(c["validity"], bitorder="little")[:c["rows"]].astype(bool)
                assert np.array_equal(arr.leaf.values_numpy().view(np.int64)[m], np.asarray(c["values"])[m]), "C5 Int64 leaf round trip failed"
            else:
                assert np.array_equal(arr.leaf.values_numpy(), c["values"]), "C5 Utf8 leaf round trip failed"
    except Exception as e:
        err = "%s: %s" % (type(e).__name__, e)
    if not all_ok(err is None):
        return {"error": err or "another rank failed while preparing its work items"} if rank == 0 else None
    cap = shard.record_capacity(plan)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    items = []
    try:
        for _ in range(steps):
            encs = nested.write_nested_leaves(ctx, pairs, opts)
            nested.read_nested_leaves(ctx, cps, [k for _, k, _ in metas_kinds], [o for _, _, o in metas_kinds])
        items = [(it.column, it.first_page, e.metas_array()) for it, e in zip(mine, encs)]
    except Exception as e:
        err = "%s: %s" % (type(e).__name__, e)
    allm = None
    try:
        allm = shard.gather_metas(items, 2, capacity=cap, device=cdev, expected_pages=[npages, npages])
    except ValueError as e:
        err = err or "%s: %s" % (type(e).__name__, e)
    barrier()
    el = time.perf_counter() - t0
    tt = torch.tensor([el, float(U_local)], dtype=torch.float64, device=h.dev if on_device else "cpu")
    if world > 1:
        mx = tt.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = tt.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        el, U = float(mx[0].item()), float(sm[1].item())
    else:
        U = float(U_local)
    if err is None:
        try:
            cm = shard.column_metas(allm)
            assert all(len(m.pages) == npages for m in cm), "every leaf column must come back with all of its pages"
        except Exception as e:
            err = "%s: %s" % (type(e).__name__, e)
    if not all_ok(err is None):
        return {"error": err or "another rank failed in the timed region"} if rank == 0 else None
    if rank != 0:
        return None
    return {"workload": "C5: 1 M-row List<Struct<Int64,Utf8>>, Zstd, %d (leaf, page-range) work items over %d GPU(s), nested write + read of "
                        "every item (synchronous calls, wall time) + ONE all_gather of %d page metas" % (sum(len(x) for x in plan), world, 2 * npages),
            "scaling": "strong", "n_gpus": world, "steps": steps, "arrow_MB": round(U / 1e6, 1),
            "ms_per_step": round(el / steps * 1e3, 3), "encdec_GBps": round(2.0 * U * steps / el / 1e9, 2),
            "items_per_rank": [len(x) for x in plan], "record_capacity": cap}


def self_launch(n, backend):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU under
    torch.distributed.run on 127.0.0.1) with the same arguments; rank 0 prints the JSON line.  With the RCCL backend
    every rank needs a device of its own: fewer visible devices than ranks is an error, never a silent 1-GPU run."""
    import socket
    import subprocess
    if backend == "nccl":
        import torch
        ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if ndev < n:
            raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible; RCCL needs one device per rank "
                             "(--backend gloo shares devices between ranks: a functional check, not a number)" % (n, ndev))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def kernel_source_sha():
    """first 16 hex digits of the sha256 over EVERY kernel source of the library (csrc/*.hip, *.h, *.cpp, sorted by name): a
    PMC profile is this run's traffic only if it was collected for the same sources (scripts/summarize_pmc.py stamps it)"""
    import hashlib
    hsh = hashlib.sha256()
    d = os.path.join(ROOT, "strawboat_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h", ".cpp")):
            with open(os.path.join(d, f), "rb") as fh:
                hsh.update(f.encode() + b"\0" + fh.read())
    return hsh.hexdigest()[:16]


LINE_LIMIT = 8192      # the driver's record parser takes ONE short stdout line (round 4's 40 KB line came back `parsed: null`)
STR_LIMIT = 120        # ... and cuts strings at 128 characters


def _f(x):
    """3 significant digits, no exponent"""
    if x is None:
        return "n/a"
    x = float(x)
    if x >= 100:
        return "%.0f" % x
    if x >= 10:
        return "%.1f" % x
    if x >= 1:
        return "%.2f" % x
    if x >= 0.01 or x == 0:
        return ("%.3f" % x).lstrip("0") or "0"
    return ("%.5f" % x).lstrip("0")


def config_summary(configs):
    """ONE short string per configuration (the driver's record keeps `config` and cuts strings at 128 characters): GB/s of
    Arrow bytes and the fraction of the 8 TB/s HBM peak reached by the direction's algorithmic bytes, per direction, and
    the CPU restatement (1 thread / all cores, encode+decode) where the entry has one"""
    out = {}
    for k, e in (configs or {}).items():
        if not isinstance(e, dict):
            continue
        if "encode" in e and "decode" in e:
            s = "enc %s GB/s (%s of HBM) dec %s (%s)" % (_f(e["encode"]["GBps"]), _f(e["encode"]["frac_hbm"]),
                                                             _f(e["decode"]["GBps"]), _f(e["decode"]["frac_hbm"]))
            r = e.get("leaf_pages_reference_written")
            if isinstance(r, dict) and "decode" in r:
                s += "; libzstd pages dec %s" % _f(r["decode"]["GBps"])
            c = e.get("cpu_baseline")
            if isinstance(c, dict):
                s += "; CPU 1t %s" % _f(c["one_thread"]["value"])
                if isinstance(c.get("all_cores"), dict):
                    s += " %dt %s" % (c["all_cores"]["cores"], _f(c["all_cores"]["value"]))
            sa = e.get("single_array")
            if isinstance(sa, dict) and "encode_ms" in sa:
                s += "; 1 array %s/%s ms" % (_f(sa["encode_ms"]), _f(sa["decode_ms"]))
            cold = e.get("cold")
            if isinstance(cold, dict) and "first_call_after_a_plain_page_ms" in cold:   # (wrong launch hints: replayed, not walked)
                f, w = cold["first_call_after_a_plain_page_ms"], cold["second_call_ms"]
                s += "; cold %s/%s ms (warm %s/%s)" % (_f(f[0]), _f(f[1]), _f(w[0]), _f(w[1]))
            out[k] = s
        elif "decode" in e and "written_by" in e:
            out[k] = "dec %s GB/s (%s of HBM peak)" % (_f(e["decode"]["GBps"]), _f(e["decode"]["frac_hbm"]))
        elif "encdec_GBps" in e:
            out[k] = "enc+dec %s GB/s over %d GPU(s), %s ms per step" % (_f(e["encdec_GBps"]), e.get("n_gpus", 1), _f(e.get("ms_per_step", 0.0)))
        elif k == "host_boundary" and "c2" in e:
            out[k] = "PCIe-inclusive, pinned host (link 63): C2 enc %s dec %s, C1 enc %s dec %s GB/s" % (
                _f(e["c2"]["encode_GBps"]), _f(e["c2"]["decode_GBps"]), _f(e["c1"]["encode_GBps"]), _f(e["c1"]["decode_GBps"]))
            if e.get("cpu_1t_GBps"):
                out[k] += "; CPU 1t C2 %s C1 %s" % (_f(e["cpu_1t_GBps"].get("c2")), _f(e["cpu_1t_GBps"].get("c1")))
        elif "error" in e:
            out[k] = ("error: %s" % e["error"])
        else:
            sub = config_summary(e)
            for kk, vv in sub.items():
                out["%s_%s" % (k, kk)] = vv
    return {k: v[:STR_LIMIT] for k, v in out.items()}


def assemble_line(head, cfg_head, roof, north, cpu, configs):
    """The ONE stdout line of a run (dict; `bench_line` turns it into text).  Short by construction: the headline keys of the
    bench contract, `config` = the workload + one short string per configuration, `roofline`, `north_star_decode`,
    `cpu_baseline`.  Everything longer (per-configuration entries with their kernels) goes to the sidecar file, not here."""
    cfg = dict(cfg_head)
    for k, v in config_summary(configs).items():
        cfg[k] = v
    if north:
        cfg["north_star_decode"] = ("C1 1 M-row Int64 page decode: %s of HBM peak end to end, %s for k_expand alone (target 0.40)"
                                    % (_f(north.get("frac_end_to_end")), _f(north.get("frac_kernel"))))[:STR_LIMIT]
    cb = None
    if cpu:
        cb = {k: cpu[k] for k in ("value", "unit", "cores", "kind") if k in cpu}
        cb["sample"] = str(cpu.get("sample", ""))[:STR_LIMIT]
        cb["cpu_model"] = str(cpu.get("cpu_model", ""))[:60]
        if isinstance(cpu.get("all_cores"), dict):
            cb["all_cores_value"] = cpu["all_cores"]["value"]
            cb["all_cores"] = cpu["all_cores"]["cores"]
        cb["note"] = "C++ restatement of the reference's algorithm (oracle/), Basic blocks via the box's liblz4 / libzstd"
    out = dict(head)
    out["config"] = cfg
    out["roofline"] = roof
    out["north_star_decode"] = north
    out["cpu_baseline"] = cb
    return out


def bench_line(out):
    """json text of the line; sheds the longest config strings rather than exceed LINE_LIMIT"""
    line = json.dumps(out, separators=(",", ":"))
    cfg = out.get("config") or {}
    while len(line) >= LINE_LIMIT:
        longest = max((k for k, v in cfg.items() if isinstance(v, str) and k != "workload"), key=lambda k: len(cfg[k]), default=None)
        if longest is None or len(cfg[longest]) <= 24:
            break
        cfg[longest] = cfg[longest][:len(cfg[longest]) // 2]
        line = json.dumps(out, separators=(",", ":"))
    return line


def write_detail(detail):
    """the long record (every configuration with its kernels, the headline's kernels): `bench_detail.json` at the repo root
    and under gpurun_out/ (what travels back from a GPU box), and one line on stderr"""
    txt = json.dumps(detail)
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        try:
            os.makedirs(d, exist_ok=True)
            with open(os.path.join(d, "bench_detail.json"), "w") as fh:
                fh.write(txt)
        except OSError:
            pass
    print("[bench-detail] " + txt, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--columns", type=int, default=512, help="1 M-row columns per GPU in one batch")
    ap.add_argument("--codec", default="adaptive", choices=["adaptive", "rle", "none", "dict"],
                    help="adaptive = default_compress_ratio 2.0, codec chosen per page on the device")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the sweep over the other configurations")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend; nccl = RCCL over xGMI (the measured configuration).  gloo lets the N > 1 path be "
                         "exercised on a box with fewer GPUs than ranks (ranks then share devices): a functional check, not a number")
    ap.add_argument("--only", default=None, help="comma list of configs (c1,c3,c3_lz4,c4,c5,one_page,host_boundary,continuity): run ONLY these, "
                                                 "without the headline (profiling runs)")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args.gpus, args.backend)   # does not return

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d; launch one rank per GPU (or run `python bench.py --gpus N` "
                         "without a launcher: it starts the N ranks itself)" % (args.gpus, world))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback in strawboat_amd)")
    ndev = torch.cuda.device_count()
    if args.backend == "nccl" and world > 1 and local_rank >= ndev:
        raise SystemExit("rank %d has no GPU of its own (%d visible); RCCL needs one device per rank" % (rank, ndev))
    local_rank %= ndev
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # (the gloo backend reports its peer connections on stdout; stdout carries the ONE JSON line, so it is pointed at
        # stderr while the group comes up)
        sys.stdout.flush()
        saved_out = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group(args.backend, rank=rank, world_size=world)  # nccl == RCCL on ROCm
            dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved_out, 1)
            os.close(saved_out)

    import strawboat_amd as sb
    from strawboat_amd import read, write
    from strawboat_amd.types import Compression, PhysicalType, WriteOptions

    def log(msg):
        print("[bench] " + msg, file=sys.stderr, flush=True)

    dev = torch.device("cuda", local_rank)
    ctx = sb.Context(local_rank)
    harness = GpuHarness(ctx)
    if args.only:
        res = run_configs(harness, set(args.only.split(",")), not args.no_cpu_baseline, log)
        write_detail({"metric": METRIC, "configs": res, "hbm_peak_GBps": HBM_PEAK_GBS, "kernel_source_sha16": kernel_source_sha()})
        print(bench_line({"metric": METRIC, "config": dict({"workload": "only: " + args.only}, **config_summary(res)), "hbm_peak_GBps": HBM_PEAK_GBS}))
        return

    B = args.columns
    codec = {"adaptive": -1, "rle": Compression.RLE, "none": Compression.NONE, "dict": Compression.DICT}[args.codec]
    if codec < 0:   # the reference's adaptive mode with its default options: nothing forbidden
        opts = WriteOptions(max_page_size=PAGE, default_compress_ratio=2.0,
                            forbidden_compressions=[])
    else:
        opts = WriteOptions(max_page_size=PAGE, force_codec=codec)

    # ---- synthetic batch, resident in HBM before the timed region (every column from its own seed)
    gen = gen_parallel(gen_c2_column, [42 + 1000 * rank + b for b in range(B)])
    host0 = gen[0]
    cols = [write.DeviceColumn(PhysicalType.FLOAT64, True, ROWS, torch.from_numpy(v.view(np.uint8)).to(dev),
                               torch.from_numpy(m).to(dev)) for v, m in gen]
    del gen
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
        all_metas = shard.gather_metas(local, world * B, capacity=B * ((ROWS + PAGE - 1) // PAGE),
                                       device=dev if args.backend == "nccl" else None)
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
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    c4s = c5s = None
    if world > 1 and not args.no_configs:   # the configuration BASELINE names for 8 GPUs, sharded by page ranges
        del wbatch, rbatch
        try:   # (its own failures are caught inside, rank-collectively; this guards what is not: the headline line comes first)
            c4s = run_c4_sharded(harness, world, rank, dist, on_device=args.backend == "nccl")
        except Exception as e:
            c4s = {"error": "%s: %s" % (type(e).__name__, e)}
        try:
            c5s = run_c5_sharded(harness, world, rank, dist, on_device=args.backend == "nccl")
        except Exception as e:
            c5s = {"error": "%s: %s" % (type(e).__name__, e)}
    ms_per_step = elapsed / args.steps * 1e3
    value = world * 2.0 * U * args.steps / elapsed / 1e9   # whole job: encode + decode bytes

    if rank == 0:
        # ---- roofline of the dominant kernel: algorithmic bytes / HIP-event time
        Abytes = {"k_expand": page_bytes + U,            # A_dec = page bytes read + Arrow bytes written
                  "k_expand_rle": page_bytes + U,
                  "k_enc_emit_pages": U + page_bytes,    # A_enc = Arrow bytes read + page bytes written
                  "k_enc_emit_tiles": U + page_bytes,
                  "k_enc_select_runs": U + page_bytes,   # fused selection + RLE: Arrow bytes read once, pages written
                  "k_enc_select_rle": U + page_bytes,
                  "k_enc_select": U + page_bytes}
        base = lambda k: k.split("<")[0]
        dom = max((k for k in stats if base(k) in Abytes), key=lambda k: stats[k][1], default=None)
        roof = None
        if dom:
            n, tot = stats[dom]
            avg_ms = tot / n
            achieved = Abytes[base(dom)] / (avg_ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                    "avg_kernel_ms": round(avg_ms, 4), "algorithmic_bytes_per_launch": Abytes[base(dom)]}
        kernels = {k: {"launches": v[0], "avg_ms": round(v[1] / v[0], 4)} for k, v in stats.items()}
        # HBM traffic of the dominant kernel from the PMC counters (rocprofv3 --pmc FETCH_SIZE /
        # WRITE_SIZE, separate passes of this same command; summary committed under profiles/)
        src_sha = kernel_source_sha()
        for pf in ("r06_pmc_traffic.json", "r05_pmc_traffic.json"):
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", pf)))
                pc = pmc["config"]
                if pmc.get("kernel_source_sha16") != src_sha:   # the counters belong to another version of the kernel: not this run's traffic
                    if roof:
                        roof["traffic_stale"] = "profiles/%s: kernel sources %s, this build %s" % (pf, pmc.get("kernel_source_sha16"), src_sha)
                    continue
                if roof and str(pc.get("workload", "")).lower().startswith("c2") and pc.get("columns_per_gpu", B) == B \
                        and pc.get("codec", args.codec) == args.codec:
                    cand = [rec["hbm_bytes_per_launch"] for name, rec in pmc["kernels"].items() if name.split("<")[0].startswith(dom.split("<")[0])]
                    if cand:
                        roof["traffic"] = max(cand)   # several template instances share a name: the one that did the work
                        roof["traffic_source"] = "profiles/%s (PMC, per launch, gfx950 FETCH_SIZE x2 correction)" % pf
                        break
            except (OSError, KeyError, ValueError):
                pass

        cpu = None
        if not args.no_cpu_baseline and world == 1:   # (the CPU leg is a property of the box: timed at N = 1 only)
            from oracle import sbo
            if codec < 0:
                o = sbo.make_options(max_page_size=PAGE, ratio=2.0, forbidden=())
            else:
                o = sbo.make_options(max_page_size=PAGE, force_codec=codec)
            c0 = dict(ptype=sbo.T_F64, nullable=True, rows=ROWS, values=host0[0], validity=host0[1], offsets=None)
            cpu = cpu_baseline([c0], o, U_col, "1 column (1 M rows, 16 pages) of the same workload, encode+decode, best of 2; the all-cores "
                                               "leg runs replicas of it page-parallel over std::threads")
        configs = None
        if world > 1 and not args.no_configs:
            configs = {"c4_sharded": c4s, "c5_sharded": c5s}
        if world == 1 and not args.no_configs:
            del wbatch, rbatch, enc, dec, pages, cols
            torch.cuda.empty_cache()
            configs = run_configs(harness, None, not args.no_cpu_baseline, log)
            hb = configs.get("host_boundary")
            if isinstance(hb, dict) and "c2" in hb and cpu:   # the CPU path has no PCIe leg: its C2 / C1 rates are the ones to hold against
                c1cpu = (configs.get("c1") or {}).get("cpu_baseline") or {}
                hb["cpu_1t_GBps"] = {"c2": cpu["value"], "c1": c1cpu.get("value")}
        north = None
        c1e = (configs or {}).get("c1")
        if isinstance(c1e, dict) and "decode" in c1e:   # BASELINE.json north_star: >= 40 % of HBM peak on 1 M-row primitive-page decode
            dk = c1e["decode"]
            A1 = (c1e["arrow_MB"] + c1e["page_MB"]) * 1e6
            kms = dk["kernels_ms"].get("k_expand")
            north = {"config": "C1: 128 x 1 M-row Int64, one page per column, no compression", "target_frac": 0.40,
                     "frac_end_to_end": dk["frac_hbm"], "decode_ms": dk["ms"],
                     "frac_kernel": round(A1 / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if kms else None, "kernel": "k_expand", "kernel_ms": kms}
        cfg = {"workload": "C2: %d x 1M-row nullable Float64 columns per GPU, 64Ki-row pages, codec %s, "
                           "inputs resident in HBM" % (B, args.codec),
               "columns_per_gpu": B, "rows_per_column": ROWS, "page_rows": PAGE,
               "arrow_bytes_per_step": U, "page_bytes_per_step": page_bytes,
               "parallelism": "pages of independent columns sharded across %d GPU(s)" % world,
               "note": "C2 is the FRIENDLIEST config (RLE 16x); the keys below: one line per other config, GB/s of Arrow bytes"}
        head = {
            "metric": METRIC,
            "value": round(value, 2), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64 bit patterns (integer/bit work, no arithmetic)",
            "data": "synthetic",
        }
        out = assemble_line(head, cfg, roof, north, cpu, configs)
        write_detail(dict(head, config=cfg, roofline=roof, north_star_decode=north, cpu_baseline=cpu, kernels=kernels, configs=configs,
                          kernel_source_sha16=src_sha))
        print(bench_line(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
