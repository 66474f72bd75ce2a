// strawboat-hip: fused adaptive selection + speculative RLE encoding of one page (included by
// sb_encode.hip inside namespace sb, after decide_prim and the RLE helpers).
//
// In adaptive mode every page is read twice: once by the selector (gen_stats + the sampled trials),
// once by the encoder of the codec it chose.  For run-heavy pages that codec is RLE, so this kernel
// writes the RLE records into the page's slot WHILE it gathers the selector's statistics, and keeps
// them if choose_compressor then says RLE — the page is read from HBM once.  When another codec
// wins the records are simply overwritten by that codec's kernel (same slot); when runs turn out to
// be shorter than 4 rows on average the speculation stops and RLE, if it still wins, is encoded by
// the ordinary RLE kernel.  The decision itself is decide_prim, shared with k_enc_select.
//
// Layout: a thread owns K = 8 CONSECUTIVE rows, loaded straight from HBM with 16-byte loads
// (scripts/micro/stream_read.hip: same 6.2 TB/s as lane-coalesced loads).  Statistics: null count
// from the validity bits; typed max / negative / unsorted for integers; distinct keys appended to a
// per-wave LDS buffer only where the canonical key changes from one row to the next (the previous
// row of a thread's first row comes from the lane before it) and probed into the LDS hash set 64 at
// a time; the Freq vote is a per-thread Boyer-Moore over its rows.  RLE: as enc_rle_rows, without
// the LDS transpose.
template <int W, int FK>
__device__ uint32_t select_rle_page(const EncodeArgs& a, const EncCol& c, const EncPage& p, uint32_t page, const SelectOpts& o,
                                    const SelScratch& sc, uint32_t* s_cnt2 /* s_kcnt, s_ksent */, bool* rle_kept) {
    static_assert(W == 4 || W == 8, "fused select + RLE: 4- and 8-byte values");
    constexpr int K = 16;
    constexpr uint32_t CHUNK = WG * K;
    constexpr int REC = 4 + W;
    constexpr uint64_t SENT = ~0ull;
    constexpr uint32_t KSLOTS = SEL_LDS_SLOTS / 2, KCAP = KSLOTS / 2;
    constexpr uint32_t CBUF = 96;
    using KE = typename std::conditional<(W == 8), unsigned long long, uint32_t>::type;
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
    // (the trials' sample rows are loaded by decide_prim: prefetching them here costs the registers K = 16 needs)
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
    KE* cbuf = (KE*)sc.sample_mem + w * CBUF;          // values whose key changed, waiting to be probed (per wave)
    uint32_t* sA = (uint32_t*)(sc.sample_mem + 4 * CBUF * sizeof(KE));  // RLE wave records (64 words)
    uint32_t* sB = sA + 64;                             // RLE wave values (8 * W bytes)
    __syncthreads();
    uint32_t ccount = 0;
    auto flush = [&]() {
        for (uint32_t base = 0; base < ccount; base += 64) {
            const bool act = base + lane < ccount;
            const uint64_t rawx = act ? (uint64_t)cbuf[base + lane] : 0;
            Val<W> rv;
            __builtin_memcpy(&rv, &rawx, W);
            const Val<W> kk = stat_key<W>(rv, nk);  // the buffer holds raw values: one canonicalisation per 64 keys
            const uint64_t x = k64(kk);
            if (act && !bits_eq<W>(kk, k0)) f_neq0 = 1;
            if (want_set && act && s_kcnt <= KCAP) {
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
        ccount = 0;
    };
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
    for (uint64_t cb = 0; cb < N; cb += CHUNK, par ^= 1) {
        const uint32_t n = (uint32_t)min((uint64_t)CHUNK, N - cb);
        const uint32_t r0 = (uint32_t)t * K;
        const uint32_t mine = r0 < n ? min((uint32_t)K, n - r0) : 0u;
        // ---- my rows
        Val<W> v[K];
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
        VWord vw = vword_issue(vv.bits, vv.off + cb + r0, vtotal, mine);
        if (!mine) vw.mask = 0;
        // the row before my first one (lane 0 of a wave: fetched, it belongs to another wave or chunk)
        Val<W> pvrow = shfl_val<W>(v[K - 1], (lane + 63) & 63);
        if (lane == 0) pvrow = getv(cb + r0 > 0 ? min(cb + r0 - 1, N - 1) : 0);  // (waves past the end of a short chunk stay in range)
        const bool has_prev_row = cb + r0 > 0;
        const uint32_t m = vw.word();
        // ---- one walk over my rows: selector statistics and RLE boundaries share the canonical keys
        // (OrderedFloat equality == equality of the canonical bits: NaNs collapsed, -0 -> +0)
        nulls += mine - (uint32_t)__popc(m);
        uint32_t sbm = 0;    // rows whose key differs from the row before (null slots included): distinct keys
        uint32_t bmask = 0;  // valid rows whose key differs from the previous VALID row of this thread: run starts
        Val<W> ek = val_zero<W>(), firstk = val_zero<W>(), firstv = val_zero<W>();
        bool seen = false;
        {
            Val<W> pk = stat_key<W>(pvrow, nk);
            Val<W> pv_int = pvrow;
#pragma unroll
            for (int j = 0; j < K; j++) {
                const bool in = (uint32_t)j < mine;
                const Val<W> kj = stat_key<W>(v[j], nk);
                if (in && ((j == 0 && !has_prev_row) || !bits_eq<W>(kj, pk))) sbm |= 1u << j;
                if (!is_float && in) {
                    if (int_lt<W>(tmax, v[j], nk)) tmax = v[j];
                    if (W == 4 && nk == NK_SIGNED && (int32_t)as_i64<W>(v[j], nk) < 0) f_neg = 1;
                    if (W == 4 && (j > 0 || has_prev_row) && int_lt<W>(v[j], pv_int, nk)) f_unsorted = 1;
                }
                if (want_vote && in) {
                    const uint64_t x = k64(kj);
                    if (vote_n == 0) {
                        vote_k = x;
                        vote_n = 1;
                    } else if (vote_k == x) {
                        vote_n++;
                    } else {
                        vote_n--;
                    }
                }
                if ((m >> j) & 1) {
                    if (!seen) {
                        firstk = kj;
                        firstv = v[j];
                    } else if (!bits_eq<W>(kj, ek)) {
                        bmask |= 1u << j;
                    }
                    ek = kj;
                    seen = true;
                }
                pk = kj;
                pv_int = v[j];
            }
        }
        // changed keys -> per-wave buffer -> hash set (and the comparison with row 0's key)
        {
            const uint32_t mycnt = (uint32_t)__popc(sbm);
            const uint32_t incl = wave_incl_scan(mycnt);
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            if (total) {
                if (ccount + total > CBUF) flush();
                if (total <= CBUF) {
                    uint32_t at = ccount + incl - mycnt;
#pragma unroll
                    for (int j = 0; j < K; j++)
                        if ((sbm >> j) & 1) cbuf[at++] = (KE)k64(v[j]);
                    ccount += total;
                } else {  // more changes than the buffer holds (no runs at all): row position by row position
#pragma unroll
                    for (int j = 0; j < K; j++) {
                        const bool b = (sbm >> j) & 1;
                        const uint64_t bm = __ballot(b);
                        const uint32_t nb = (uint32_t)__popcll(bm);
                        if (ccount + nb > CBUF) flush();
                        if (b) cbuf[ccount + mbcnt64(bm)] = (KE)k64(v[j]);
                        ccount += nb;
                    }
                }
            }
        }
        if (!spec) continue;
        // ---- speculative RLE (see enc_rle_rows): carries and ranks across threads
        uint32_t* s_has = sA + par * 16;
        uint32_t* s_cnt = sA + par * 16 + 4;
        uint32_t* s_blast = sA + par * 16 + 8;
        Val<W>* s_last = (Val<W>*)sB + par * 4;  // canonical key of the wave's last valid row
        const Val<W> lastk = ek;
        const uint64_t hm = __ballot(m != 0);
        const bool has_w = hm != 0;
        const Val<W> last_w = readlane_val<W>(lastk, has_w ? top_bit(hm) : 0);
        if (lane == 0) {
            s_has[w] = has_w;
            s_last[w] = last_w;
        }
        lds_barrier();
        bool chas = have;
        Val<W> cval = last;
        for (int pw = 0; pw < 3; pw++)
            if (pw < w && s_has[pw]) {
                chas = true;
                cval = s_last[pw];
            }
        const uint64_t pm = hm & lt;
        const Val<W> pvs = shfl_val<W>(lastk, pm ? top_bit(pm) : 0);
        const bool pc = pm ? true : chas;
        const Val<W> pvk = pm ? pvs : cval;
        if (m) {
            const int f = __ffs((int)m) - 1;
            if (!pc)
                st_val<W>(dst + 4, firstv);
            else if (!bits_eq<W>(pvk, firstk))
                bmask |= 1u << f;
        }
        const uint32_t cnt = (uint32_t)__popc(bmask);
        const uint32_t blast = bmask ? r0 + (31u - (uint32_t)__clz((int)bmask)) + 1 : 0;
        const uint32_t incl = wave_incl_scan(cnt);
        const uint64_t bmk = __ballot(bmask != 0);
        const uint64_t pb = bmk & lt;
        const uint32_t prev_blast = __shfl(blast, pb ? top_bit(pb) : 0, 64);
        const uint32_t cnt_w = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        const uint32_t blast_w = (uint32_t)__builtin_amdgcn_readlane((int)blast, bmk ? top_bit(bmk) : 0);
        if (lane == 0) {
            s_cnt[w] = cnt_w;
            s_blast[w] = bmk ? blast_w : 0;
        }
        lds_barrier();
        uint32_t base = nrec;
        uint64_t start_prev = run_start;
        for (int pw = 0; pw < 3; pw++)
            if (pw < w) {
                base += s_cnt[pw];
                if (s_blast[pw]) start_prev = cb + s_blast[pw] - 1;
            }
        if (bmask) {  // my records: one after the other, the start of each run is the boundary before it
            uint8_t* rec = dst + (uint64_t)(base + incl - cnt) * REC;
            uint64_t start = pb ? cb + prev_blast - 1 : start_prev;
#pragma unroll
            for (int j = 0; j < K; j++) {
                if (!((bmask >> j) & 1)) continue;
                const uint64_t row = cb + r0 + j;
                stu32(rec, (uint32_t)(row - start));  // count of the run that ends here
                st_val<W>(rec + REC + 4, v[j]);       // value of the run that starts here
                rec += REC;
                start = row;
            }
        }
        for (int pw = 0; pw < 4; pw++) {
            if (s_has[pw]) {
                have = true;
                last = s_last[pw];
            }
            nrec += s_cnt[pw];
            if (s_blast[pw]) run_start = cb + s_blast[pw] - 1;
        }
        // runs shorter than 4 rows on average: RLE is unlikely to be chosen, stop paying for it
        if ((uint64_t)nrec * 4 > cb + n + 256) spec = false;
    }
    flush();
    __syncthreads();  // every wave's keys are in the set before its size is read
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
