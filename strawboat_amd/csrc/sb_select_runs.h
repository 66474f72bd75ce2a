// strawboat-hip: fused adaptive selection + speculative RLE encoding of one page, RUN-level version
// (included by sb_encode.hip inside namespace sb, after sb_select_rle.h).
//
// select_rle_page spends ~50 VALU instructions on every row (canonical key, change detection, vote,
// RLE boundary) and is ALU-bound at a third of the HBM rate.  But everything the selector and the RLE
// encoder need is constant inside a run of rows with identical BITS:
//   * gen_stats (integer/mod.rs:179-229, double/mod.rs:178-229): distinct keys, all-equal, max,
//     sortedness and sign can only change where the bits change; the Freq vote takes a run as one
//     weighted step; the null count comes from the validity words;
//   * RLE (integer/rle.rs:64-104): a run boundary is the first VALID row of a raw run whose canonical
//     key differs from the key of the last raw run that held a valid row (nulls extend the run).
// So the per-row work shrinks to a comparison with the row before (3 instructions); the positions
// where the bits change are compacted into an LDS list, and a second step handles one raw RUN per
// lane.  For pages with runs of ~32 rows that is 1/32 of the old per-row work.
//
// The run values travel through LDS.  A chunk's rows are requested at the top of its iteration: requesting the next
// chunk before step 2 (measured, round 2) keeps 32 more VGPRs live across step 2, which either spills at 4 waves /
// SIMD (1.58 ms on C2 instead of 1.10) or drops the kernel to 3 waves / SIMD (1.33 ms) — the other three
// workgroups of the CU are what hides the load latency.  More workgroups per CU do not help either (measured, round 2,
// with a 16 KiB key set so that LDS allows 6): 5 waves / SIMD (96 VGPRs, 252 B of scratch) 1.22 ms, 6 waves / SIMD
// (80 VGPRs, 320 B of scratch) 1.38 ms, against 1.19 ms on the same box — the kernel is bound by instruction issue.
//
// A chunk (4096 rows) with more than RUNS_CAP raw runs (runs shorter than ~6 rows on average) makes
// the page FALL BACK to select_rle_page (k_enc_select_rle runs after this kernel and takes the pages
// marked CODEC_PENDING): lane = run pays off only when there are runs.
#ifndef SB_RUNS_TOUCH
#define SB_RUNS_TOUCH 0   // EXPERIMENT (scripts/micro/runs_dma.hip): one dword of the NEXT chunk's 128 bytes per thread, to have them in L2
#endif
#ifndef SB_RUNS_DMA
#define SB_RUNS_DMA 0
#endif
constexpr int32_t CODEC_PENDING = -100;
constexpr uint32_t RUNS_CAP = 640;

template <int W, int FK>
__device__ uint32_t select_runs_page(const EncodeArgs& a, const EncCol& c, const EncPage& p, uint32_t page, const SelectOpts& o,
                                     const SelScratch& sc, uint32_t* s_cnt2 /* s_kcnt, s_ksent */, bool* rle_kept, bool* fallback) {
    static_assert(W == 4 || W == 8, "fused select + RLE: 4- and 8-byte values");
    constexpr int K = 16;
    constexpr uint32_t CHUNK = WG * K;
    constexpr int REC = 4 + W;
    constexpr uint64_t SENT = ~0ull;
    constexpr uint32_t KSLOTS = SEL_LDS_SLOTS / 2, KCAP = KSLOTS / 2;
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint64_t lt = (1ull << lane) - 1;
    const uint64_t N = p.rows;
    const uint8_t* vals = c.values + p.row0 * W;
    const ValidView vv{c.validity, c.validity_bit_offset + p.row0};
    const uint32_t nk = c.nk;
    const bool is_float = nk >= NK_F32;
    auto getv = [=](uint64_t i) { return ld_val<W>(vals + i * W); };
    auto forbidden = [&](uint32_t cd) { return (o.forbidden >> cd) & 1u; };
    auto k64 = [&](const Val<W>& k) {
        uint64_t x = 0;
        __builtin_memcpy(&x, &k, W);
        return x;
    };
    *rle_kept = false;
    *fallback = false;
    // ---- selector state
    const Val<W> k0 = stat_key<W>(getv(0), nk);
    uint32_t f_neq0 = 0, f_unsorted = 0, f_neg = 0, nulls = 0;
    Val<W> tmax = getv(0);
    uint64_t vote_k = 0;
    uint32_t vote_n = 0;
    unsigned long long* kset = (unsigned long long*)sc.lds_tab;
    uint32_t& s_kcnt = s_cnt2[0];
    uint32_t& s_ksent = s_cnt2[1];
    const bool want_set = !forbidden(SB_CODEC_DICT) && N >= 3;
    const bool want_vote = !forbidden(SB_CODEC_FREQ);
    for (uint32_t i = t; i < KSLOTS; i += WG) kset[i] = SENT;
    if (t == 0) {
        s_kcnt = 0;
        s_ksent = 0;
    }
    // LDS: sc.sample_mem and sc.s_misc are one contiguous pool here (k_enc_select_runs), free until decide_prim
    uint16_t* runs = (uint16_t*)sc.sample_mem;                         // RUNS_CAP + 2 chunk-relative run starts
    uint32_t* s_vb = (uint32_t*)(sc.sample_mem + 2 * (RUNS_CAP + 8));   // CHUNK / 32 validity words of the chunk
    uint32_t* s_x = s_vb + CHUNK / 32;                                  // 2 parities x 16 words of wave records
    Val<W>* s_lastk = (Val<W>*)(s_x + 32);                              // 2 parities x 4 keys
    Val<W>* rvals = s_lastk + 8;                                        // RUNS_CAP + 1 raw values: [0] = the row before
                                                                        // the chunk, [1 + r] = value of run r
    __syncthreads();
    // ---- RLE state (speculative)
    bool spec = !forbidden(SB_CODEC_RLE);
    uint8_t* slot = page_slot(a, c, p);
    const uint64_t pos = c.nullable ? def_section_bytes(N) : 0;
    uint8_t* dst = slot + pos + 9;
    uint64_t run_start = 0;
    uint32_t nrec = 0;
    bool have = false;
    Val<W> last = val_zero<W>();
    const uint64_t vtotal = vv.off + N;
    auto lds_barrier = []() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    };
    uint32_t par = 0;
    // rows of one chunk (thread = K consecutive rows), the row before a wave's first row, and a validity word
    Val<W> v[K];
    Val<W> pv0 = val_zero<W>();
    uint32_t vword = 0;
    auto request = [&](uint64_t cb) {
        const uint32_t n = (uint32_t)min((uint64_t)CHUNK, N - cb);
        const uint32_t r0 = (uint32_t)t * K;
        if (r0 + K <= n) {
            constexpr int NV = K * W / 16;
            u32x4 q[NV];
#pragma unroll
            for (int u = 0; u < NV; u++) q[u] = ldu128(vals + (cb + r0) * W + 16 * u);
            __builtin_memcpy(v, q, K * W);
        } else {
#pragma unroll
            for (int j = 0; j < K; j++) v[j] = getv(cb + (r0 + j < n ? r0 + j : n - 1));
        }
        if (lane == 0) pv0 = getv(cb + r0 > 0 ? min(cb + r0 - 1, N - 1) : 0);
        vword = 0xFFFFFFFFu;
        if (t < (int)(CHUNK / 32) && (uint32_t)t * 32 < n && vv.bits) vword = bits32(vv.bits, vv.off + cb + (uint32_t)t * 32, vtotal);
    };
#if SB_RUNS_DMA
    // EXPERIMENT (scripts/micro/runs_dma.hip): the rows of chunk k + 1 travel HBM -> LDS with
    // global_load_lds_dwordx4 while chunk k is worked on — no VGPRs are held for them — and are read out of the stage
    // (8 x ds_read_b128 per thread, conflict-free: the stage is [wave][piece][lane] x 16 bytes, the layout the DMA writes)
    // at the top of the next iteration.  One stage of CHUNK * W bytes (32 KB for 8-byte values) on top of the 32 KB key
    // set: 2 workgroups per CU instead of 4.
    __shared__ __attribute__((aligned(16))) uint8_t dma_stage[CHUNK * W];
    constexpr int NV_DMA = K * W / 16;
    auto dma_issue = [&](uint64_t cb) {
        const uint8_t* g = vals + (cb + (uint64_t)t * K) * W;
#pragma unroll
        for (int u = 0; u < NV_DMA; u++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + 16 * u),
                                             (__attribute__((address_space(3))) void*)(dma_stage + ((w * NV_DMA + u) * 64) * 16), 16, 0, 0);
    };
    if (N >= CHUNK) dma_issue(0);
#endif
    STL(0);
    for (uint64_t cb = 0; cb < N; cb += CHUNK) {
        XTL(0);
#if SB_RUNS_DMA
        if (N - cb >= CHUNK) {
            __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): my pieces have landed
            {
                u32x4 q[NV_DMA];
#pragma unroll
                for (int u = 0; u < NV_DMA; u++) q[u] = *(const u32x4*)(dma_stage + ((w * NV_DMA + u) * 64 + lane) * 16);
                __builtin_memcpy(v, q, K * W);
            }
            if (lane == 0) pv0 = getv(cb + (uint64_t)t * K > 0 ? cb + (uint64_t)t * K - 1 : 0);
            vword = 0xFFFFFFFFu;
            if (t < (int)(CHUNK / 32) && vv.bits) vword = bits32(vv.bits, vv.off + cb + (uint32_t)t * 32, vtotal);
            // (a wave reads only what it wrote: no barrier between the read-out and the next issue)
            if (N - (cb + CHUNK) >= CHUNK && cb + CHUNK < N) dma_issue(cb + CHUNK);
        } else {
            request(cb);
        }
#else
        request(cb);
#endif
#if SB_RUNS_TOUCH
        uint32_t touch = 0;
        if (cb + CHUNK + (uint64_t)(t + 1) * K <= N) {
            const uint8_t* tp = vals + (cb + CHUNK + (uint64_t)t * K) * W;
            asm volatile("global_load_dword %0, %1, off" : "=v"(touch) : "v"(tp) : "memory");
        }
#endif
        const uint32_t n = (uint32_t)min((uint64_t)CHUNK, N - cb);
        const uint32_t r0 = (uint32_t)t * K;
        const uint32_t mine = r0 < n ? min((uint32_t)K, n - r0) : 0u;
        // ---- step 1: where do the bits change?
        if (t < (int)(CHUNK / 32)) {  // the chunk's validity words (null count rides along)
            const uint32_t b0 = (uint32_t)t * 32;
            uint32_t word = 0;
            if (b0 < n) {
                const uint32_t nb = min(32u, n - b0);
                const uint32_t m = nb >= 32 ? 0xFFFFFFFFu : (1u << nb) - 1;
                word = vword & m;
                nulls += nb - (uint32_t)__popc(word);
            }
            s_vb[t] = word;
        }
        Val<W> pvrow = shfl_val<W>(v[K - 1], (lane + 63) & 63);
        if (lane == 0) pvrow = pv0;
        if (t == 0) rvals[0] = pvrow;
        uint32_t rbm = 0;
        {
            Val<W> pr = pvrow;
#pragma unroll
            for (int j = 0; j < K; j++) {
                if (!bits_eq<W>(v[j], pr)) rbm |= 1u << j;
                pr = v[j];
            }
            if (r0 == 0) rbm |= 1u;  // every chunk starts a run: the list is per chunk (a raw run that continues from the
                                     // chunk before may hold its first valid row here)
            rbm &= mine >= (uint32_t)K ? 0xFFFFu : ((1u << mine) - 1);
        }
        XTL(1);
        uint32_t* s_rc = s_x + par * 16 + 12;  // [4] raw runs starting in each wave
        const uint32_t cnt1 = (uint32_t)__popc(rbm);
        const uint32_t incl1 = wave_incl_scan(cnt1);
        if (lane == 63) s_rc[w] = incl1;
        lds_barrier();
        XTL(2);
        const uint32_t total = s_rc[0] + s_rc[1] + s_rc[2] + s_rc[3];
        if (total > RUNS_CAP) {  // short runs: lane = run does not pay, the row-level kernel takes the page
            *fallback = true;
            return 0;
        }
        {
            uint32_t at = incl1 - cnt1;
            for (int pw = 0; pw < 3; pw++)
                if (pw < w) at += s_rc[pw];
            while (rbm) {  // (a handful of iterations: the value is picked with a select chain, not 16 guarded stores)
                const int j = __ffs((int)rbm) - 1;
                rbm &= rbm - 1;
                Val<W> x = v[0];
#pragma unroll
                for (int q = 1; q < K; q++)
                    if (j == q) x = v[q];
                runs[at] = (uint16_t)(r0 + (uint32_t)j);
                rvals[1 + at] = x;
                at++;
            }
            if (t == 0) runs[total] = (uint16_t)n;  // (n <= 4096 fits)
        }
        XTL(3);
        lds_barrier();
        XTL(4);
        // ---- step 2: one raw run per lane
        for (uint32_t rb = 0; rb < total; rb += WG, par ^= 1) {
            const uint32_t r = rb + (uint32_t)t;
            const bool act = r < total;
            const uint32_t start = act ? runs[r] : 0, end = act ? runs[r + 1] : 0;
            const uint64_t row = cb + start;
            const Val<W> val = rvals[act ? 1 + r : 0];
            const Val<W> kk = stat_key<W>(val, nk);
            if (act) {
                if (!bits_eq<W>(kk, k0)) f_neq0 = 1;
                if (!is_float) {
                    if (int_lt<W>(tmax, val, nk)) tmax = val;
                    if (W == 4 && nk == NK_SIGNED && (int32_t)as_i64<W>(val, nk) < 0) f_neg = 1;
                    if (W == 4 && row > 0 && int_lt<W>(val, rvals[r], nk)) f_unsorted = 1;  // rvals[r]: the row before
                }
                if (want_vote) {  // Boyer-Moore with the run length as weight (== feeding its rows one by one)
                    const uint64_t x = k64(kk);
                    const uint32_t wgt = end - start;
                    if (vote_k == x) {
                        vote_n += wgt;
                    } else if (vote_n >= wgt) {
                        vote_n -= wgt;
                    } else {
                        vote_k = x;
                        vote_n = wgt - vote_n;
                    }
                }
                if (want_set && s_kcnt <= KCAP) {
                    const uint64_t x = k64(kk);
                    if (x == SENT) {
                        s_ksent = 1;
                    } else {
                        uint32_t h = (((uint32_t)x ^ (uint32_t)(x >> 32) * 0x85EBCA6Bu) * 0x9E3779B1u >> 15) & (KSLOTS - 1);
                        for (;;) {
                            unsigned long long cur = kset[h];
                            if (cur == x) break;
                            if (cur == SENT) {
                                const unsigned long long old = atomicCAS(&kset[h], (unsigned long long)SENT, (unsigned long long)x);
                                if (old == SENT) {
                                    atomicAdd(&s_kcnt, 1u);
                                    break;
                                }
                                if (old == x) break;
                            }
                            h = (h + 1) & (KSLOTS - 1);
                        }
                    }
                }
            }
            XTL(5);
            if (!spec) continue;
            // first valid row of my run (chunk relative), -1 if it has none
            int fv = -1;
            if (act) {
                const uint32_t w0 = start >> 5, w1 = (end - 1) >> 5;
                for (uint32_t wd = w0; wd <= w1; wd++) {
                    uint32_t bits = s_vb[wd];
                    if (wd == w0) bits &= 0xFFFFFFFFu << (start & 31);
                    if (wd == w1 && (end & 31)) bits &= (1u << (end & 31)) - 1;
                    if (bits) {
                        fv = (int)(wd * 32) + __ffs((int)bits) - 1;
                        break;
                    }
                }
            }
            const bool hv = fv >= 0;
            uint32_t* s_has = s_x + par * 16;
            uint32_t* s_cnt = s_x + par * 16 + 4;
            uint32_t* s_blast = s_x + par * 16 + 8;
            Val<W>* s_last = s_lastk + par * 4;  // canonical key of the wave's last run with a valid row
            const uint64_t hm = __ballot(hv);
            const bool has_w = hm != 0;
            const Val<W> last_w = readlane_val<W>(kk, has_w ? top_bit(hm) : 0);
            if (lane == 0) {
                s_has[w] = has_w;
                s_last[w] = last_w;
            }
            XTL(6);
            lds_barrier();
            XTL(7);
            bool chas = have;
            Val<W> cval = last;
            for (int pw = 0; pw < 3; pw++)
                if (pw < w && s_has[pw]) {
                    chas = true;
                    cval = s_last[pw];
                }
            const uint64_t pm = hm & lt;
            const Val<W> pvs = shfl_val<W>(kk, pm ? top_bit(pm) : 0);
            const bool pc = pm ? true : chas;
            const Val<W> pvk = pm ? pvs : cval;
            if (hv && !pc) st_val<W>(dst + 4, val);  // the page's first valid row: value of the first record
            const bool b = hv && pc && !bits_eq<W>(pvk, kk);
            const uint64_t bmk = __ballot(b);
            const uint32_t blast = b ? (uint32_t)fv + 1 : 0;
            const uint64_t pb = bmk & lt;
            const uint32_t prev_blast = __shfl(blast, pb ? top_bit(pb) : 0, 64);
            const uint32_t cnt_w = (uint32_t)__popcll(bmk);
            const uint32_t blast_w = (uint32_t)__builtin_amdgcn_readlane((int)blast, bmk ? top_bit(bmk) : 0);
            if (lane == 0) {
                s_cnt[w] = cnt_w;
                s_blast[w] = bmk ? blast_w : 0;
            }
            XTL(8);
            lds_barrier();
            XTL(9);
            uint32_t base = nrec;
            uint64_t start_prev = run_start;
            for (int pw = 0; pw < 3; pw++)
                if (pw < w) {
                    base += s_cnt[pw];
                    if (s_blast[pw]) start_prev = cb + s_blast[pw] - 1;
                }
            if (b) {  // the run that ends at my first valid row, and the value of the one that starts there
                uint8_t* rec = dst + (uint64_t)(base + (uint32_t)__popcll(pb)) * REC;
                const uint64_t sr = pb ? cb + prev_blast - 1 : start_prev;
                stu32(rec, (uint32_t)(cb + (uint32_t)fv - sr));
                st_val<W>(rec + REC + 4, val);
            }
            for (int pw = 0; pw < 4; pw++) {
                if (s_has[pw]) {
                    have = true;
                    last = s_last[pw];
                }
                nrec += s_cnt[pw];
                if (s_blast[pw]) run_start = cb + s_blast[pw] - 1;
            }
        }
        XTL(10);
#if SB_RUNS_TOUCH
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (long landed: the chunk's own loads were waited for above)
        asm volatile("" ::"v"(touch));                      // the register stays the load's until here
#endif
        if (!spec) lds_barrier();  // (the run list and the validity words are rewritten by the next chunk)
        // runs shorter than 4 rows on average: RLE is unlikely to be chosen, stop paying for it
        if ((uint64_t)nrec * 4 > cb + n + 256) spec = false;
    }
    __syncthreads();  // every wave's keys are in the set before its size is read
    STL(1);
    PrimPartials<W> pp{f_neq0, f_unsorted, f_neg, nulls, tmax, vote_k, vote_n};
    const uint32_t s_k = want_set ? s_kcnt : 0u, s_s = want_set ? s_ksent : 0u;
    SamplePre<W> none;
    __builtin_memset(&none, 0, sizeof none);
    const uint32_t codec = decide_prim<W>(getv, vv, N, nk, o, sc, pp, want_set, want_vote, s_k, s_s, none, none, none, none, false);
    if (codec == SB_CODEC_RLE && spec) {  // keep the records: close the last run, add the def levels and the header
        if (t == 0) {
            uint8_t* r = dst + (uint64_t)nrec * REC;
            stu32(r, (uint32_t)(N - run_start));
            if (!have) st_val<W>(r + 4, val_zero<W>());
        }
        if (c.nullable) {
            uint8_t* bits = def_header(slot, N);
            def_bits_page(bits, ValidView{c.validity, c.validity_bit_offset}, p.row0, N, c.rows);
        }
        const uint64_t body = (uint64_t)(nrec + 1) * REC;
        if (t == 0) {
            put_hdr9(slot + pos, SB_CODEC_RLE, (uint32_t)body, (uint32_t)(N * W));
            EncOut out;
            out.length = pos + 9 + body;
            out.out_off = 0;
            out.slot = slot;
            out.codec = SB_CODEC_RLE;
            out.pad = 1;  // emitted here: k_enc_emit_pages<., RLE> leaves the page alone
            a.outs[page] = out;
        }
        *rle_kept = true;
    }
    return codec;
}
