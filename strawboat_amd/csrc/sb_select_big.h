// strawboat-hip: adaptive selection of LONG pages, section-parallel (included by sb_encode.hip inside namespace sb, after
// sb_select_runs.h).
//
// The reference's default paging is one page per column and chunk (src/write/common.rs:54-58: max_page_size = None), so a
// page can hold millions of rows.  The page selectors above stream a page through ONE workgroup (3 000 latency-bound
// chunk iterations for 12 M rows: 60-85 ms), and when the page holds more distinct values than the LDS set takes they
// count them exactly on an HBM table — still one workgroup, one probe chain at a time.  Pages of SEL_BIG_ROWS rows or more
// (4- and 8-byte values) go through these kernels instead; the choice is decide_prim's, fed with the same statistics:
//
//   k_sel_big_sec    (sections x pages)   gen_stats of one section of <= 256 per page: flags, null count, typed maximum,
//                                         Boyer-Moore vote, the section's distinct keys (LDS set, <= 2048, dumped to HBM)
//   k_sel_big_merge  (1 workgroup / page) the sections' partials merged, the key sets united in LDS; when neither an exact
//                                         distinct count nor a majority count is missing the codec is chosen here
//   k_sel_big_clear  (sections x pages)   pages that need the HBM table: cleared
//   k_sel_big_count  (sections x pages)   every section inserts its rows into the page's table (agent-scope CAS, the count
//                                         in the page record; all sections stop once it passes Dict's limit) and counts
//                                         the rows equal to the vote's candidate
//   k_sel_big_decide (1 workgroup / page) decide_prim with those two numbers
//
// The sections' records live in the page's own output slot (nothing has been written there yet: speculative RLE is off
// for these pages; >= 12 bytes of slot per row against 16.5 KB per section of >= 16 384 rows).  integer/mod.rs:179-308,
// double/mod.rs:178-307.
constexpr uint64_t SEL_BIG_ROWS = 1ull << 18;
#ifndef SB_BIN_BIG_ROWS
#define SB_BIN_BIG_ROWS (1ull << 18)
#endif
constexpr uint64_t BIN_BIG_ROWS = SB_BIN_BIG_ROWS;   // binary pages of this many rows take the section-parallel selector / Dict writer
constexpr uint32_t SEL_BIG_SECTIONS = 256;    // at most, per page
constexpr uint32_t SEL_BIG_MIN_SEC = 16384;   // rows of a section: a power of two, at least this
constexpr uint32_t BIG_COUNT_SPLIT = 4;       // workgroups per section in k_sel_big_count (12 M rows, round 5 with the LDS set in front: sorted / distinct 0.37 ms either way, 98 % one value 0.42 -> 0.13 ms)
constexpr uint32_t BIG_SEC_SPLIT = 4;         // workgroups per section in k_sel_big_sec, a partial record each (12 M rows are 184 sections: fewer workgroups than CUs, each a chain of 16 chunk steps)
constexpr uint32_t BIG_KCAP = SEL_LDS_SLOTS / 4;   // keys of an LDS set (KSLOTS / 2 of the page selectors)

__host__ __device__ __forceinline__ uint64_t big_sec_rows(uint64_t N) {
    uint64_t r = SEL_BIG_MIN_SEC;
    while (r * SEL_BIG_SECTIONS < N) r <<= 1;
    return r;
}
struct BigSec {   // 64 bytes: the record of the section's first part, the other parts' behind it (BIG_SEC_STRIDE bytes per section)
    uint32_t flags, nulls, vote_n, kcnt;   // flags: neq0 | unsorted << 1 | neg << 2 | step down << 3; kcnt > BIG_KCAP: the set overflowed
    uint64_t tmax, vote_k;
    uint32_t ksent, runs, pad[6];          // runs: rows whose key differs from the row before (the page's first row counts)
};
struct BigPage {   // 256 bytes at the start of the slot
    uint32_t flags, nulls, maj_n, set_ok;
    uint64_t tmax, maj_k;
    uint32_t set_unique, ksent, need_uq, need_mc;
    uint32_t uq, mc;   // k_sel_big_count's results (atomics)
    uint32_t uq_sent;  // the all-ones key (the table's "empty") was met
    uint32_t sorted_uq;   // integers that never step down: the distinct keys are the runs (run_uq), no count pass
    unsigned long long tus;   // binary pages: bytes of the distinct strings (8 + len each: binary/dict.rs:43-53), k_sel_big_count
    uint32_t run_uq, pad[45];
};
constexpr uint32_t BIG_SEC_STRIDE = 64 + BIG_KCAP * 8;
static_assert(sizeof(BigSec) == 64 && sizeof(BigPage) == 256, "records of the long-page selector");
// slots of the page's HBM table: every section stops once the count has passed Dict's limit (N - 1) / 3, checked before each
// batch of WG * 8 rows, so the table never holds more than limit + sections * WG * 8 keys — a quarter of the 2 N slots the
// page selectors clear
// The slots hold the KEYS (8 bytes; the page selectors' tables hold row indices and read the row's key to compare): a
// probe is one load, an insert one CAS, nothing else.  M * 8 < 15 N bytes always fits the aux area ((2^k >= 2 N) + 3 N words).
__host__ __device__ __forceinline__ uint64_t big_tab_slots(uint64_t N) {
    const uint64_t SR = big_sec_rows(N), nsec = (N + SR - 1) / SR;
    const uint64_t need = 2 * ((N - 1) / 3 + nsec * BIG_COUNT_SPLIT * WG * 8 + 64);
    uint64_t M = 64;
    while (M < need) M <<= 1;
    return M;
}
// (the slot of a binary page starts behind the column's value bytes before it — any byte; the records hold 8-byte words and
// words that are the target of atomics, so they start at the next multiple of 16)
__device__ __forceinline__ uint8_t* big_rec_base(uint8_t* slot) { return (uint8_t*)(((uintptr_t)slot + 15) & ~(uintptr_t)15); }
__device__ __forceinline__ BigPage* big_page_rec(uint8_t* slot) { return (BigPage*)big_rec_base(slot); }
__device__ __forceinline__ BigSec* big_sec_rec(uint8_t* slot, uint32_t s, uint32_t part = 0) {
    return (BigSec*)(big_rec_base(slot) + 256 + (uint64_t)s * BIG_SEC_STRIDE + part * 64);
}
static_assert(BIG_SEC_SPLIT * 64 <= BIG_SEC_STRIDE && SEL_BIG_MIN_SEC % BIG_SEC_SPLIT == 0, "partial records of a section");

// voff: 0 = the pages of the list; n_pages = their VIRTUAL pages (the u32 index array of a long Dict page, sb_dict_big.h),
// whose table entries exist only once k_dict_big_idx has written them — the codec word says so
// The union of the sections' key sets: a table of BIG_UNION_SLOTS keys behind the section records, filled by the sections
// themselves at their end (agent-scope CAS; the count beside it).  One workgroup uniting 184 sets of 500 keys took 0.27 ms.
constexpr uint32_t BIG_UNION_SLOTS = 8192, BIG_UNION_PROBES = 32;
__device__ __forceinline__ unsigned long long* big_union_tab(uint8_t* slot, uint32_t nsec) {
    return (unsigned long long*)(big_rec_base(slot) + 256 + (uint64_t)nsec * BIG_SEC_STRIDE + 64);
}
__device__ __forceinline__ uint32_t* big_union_cnt(uint8_t* slot, uint32_t nsec) { return (uint32_t*)(big_rec_base(slot) + 256 + (uint64_t)nsec * BIG_SEC_STRIDE); }

__device__ __forceinline__ bool big_page_of(const EncodeArgs& a, const uint32_t* big, int W, uint32_t* page, EncPage* p, EncCol* c, uint32_t voff = 0) {
    *page = big[blockIdx.y] + voff;
    if (a.codecs[*page] != CODEC_PENDING) return false;   // (k_enc_select_runs may have taken the page)
    *p = get_page(a, *page);
    *c = get_col(a, p->col);
    if (c->ptype == SB_TYPE_BINARY || c->ptype == SB_TYPE_LARGE_BINARY) {
        // a long binary page: the statistics run over one 64-bit hash per row (k_enc_bin_hash wrote them before the selector;
        // equal hashes count as equal strings, as in choose_bin) — the page as a column of u64 keys.  ptype and offsets stay:
        // page_slot and the string lengths need them.
        if (W != 8 || p->h64_off == ~0ull) return false;
        c->values = a.scratch + p->h64_off - p->row0 * 8;
        c->width = 8;
        c->nk = NK_UNSIGNED;
        c->fkind = 0;
        return true;
    }
    return (int)c->width == W;
}
__device__ __forceinline__ bool big_is_bin(const EncCol& c) { return c.ptype == SB_TYPE_BINARY || c.ptype == SB_TYPE_LARGE_BINARY; }
// 8 + len of row i of a binary page (c: the column as big_page_of left it)
__device__ __forceinline__ uint32_t big_bin_weight(const EncCol& c, const EncPage& p, uint64_t i) {
    if (c.ptype == SB_TYPE_BINARY) {
        const uint8_t* o = c.offsets + (p.row0 + i) * 4;
        return ldu32(o + 4) - ldu32(o) + 8;
    }
    const uint8_t* o = c.offsets + (p.row0 + i) * 8;
    return (uint32_t)(ldu64(o + 8) - ldu64(o)) + 8;
}

// one step of the merge of two Boyer-Moore states
__device__ __forceinline__ void vote_merge(unsigned long long& k0, uint32_t& n0, unsigned long long k1, uint32_t n1) {
    if (!n1) return;
    if (n0 == 0) {
        k0 = k1;
        n0 = n1;
    } else if (k0 == k1) {
        n0 += n1;
    } else if (n1 > n0) {
        k0 = k1;
        n0 = n1 - n0;
    } else {
        n0 -= n1;
    }
}

template <int W>
__global__ void __launch_bounds__(WG, 4) k_sel_big_sec(EncodeArgs a, const uint32_t* big, uint32_t voff) {
    constexpr int K = 16;
    constexpr uint32_t CHUNK = WG * K;
    constexpr uint64_t SENT = ~0ull;
    constexpr uint32_t KSLOTS = SEL_LDS_SLOTS / 2;
    constexpr uint32_t CBUF = 96;
    using KE = typename std::conditional<(W == 8), unsigned long long, uint32_t>::type;
    __shared__ unsigned long long kset[KSLOTS];
    __shared__ KE cbufs[4 * CBUF];
    __shared__ unsigned long long s_vk[WG];
    __shared__ uint32_t s_vn[WG];
    __shared__ uint32_t s4[4];
    __shared__ uint32_t s_kcnt, s_ksent;
    uint32_t page;
    EncPage p;
    EncCol c;
    if (!big_page_of(a, big, W, &page, &p, &c, voff)) return;
    // (blockIdx.x = section * BIG_SEC_SPLIT + part: a part is a section of its own to everything below but the record's place)
    const uint64_t N = p.rows, SR_full = big_sec_rows(N), SR = SR_full / BIG_SEC_SPLIT;
    const uint64_t s0 = (uint64_t)blockIdx.x * SR;
    if (s0 >= N) return;
    const uint64_t s1 = min(N, s0 + SR);
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint8_t* vals = c.values + p.row0 * W;
    const ValidView vv{c.validity, c.validity_bit_offset + p.row0};
    const uint32_t nk = c.nk;
    const bool is_float = nk >= NK_F32;
    const uint32_t forb = a.forbidden | p.forb_extra;
    auto getv = [=](uint64_t i) { return ld_val<W>(vals + i * W); };
    auto k64 = [&](const Val<W>& k) {
        uint64_t x = 0;
        __builtin_memcpy(&x, &k, W);
        return x;
    };
    const Val<W> k0 = stat_key<W>(getv(0), nk);
    uint32_t f_neq0 = 0, f_unsorted = 0, f_neg = 0, f_down = 0, nulls = 0, runs = 0;
    Val<W> tmax = getv(0);
    uint64_t vote_k = 0;
    uint32_t vote_n = 0;
    // (binary pages need the bytes of the distinct strings beside their number: always the exact count, k_sel_big_count)
    const bool want_set = !((forb >> SB_CODEC_DICT) & 1) && N >= 3 && !big_is_bin(c);
    const bool want_vote = !((forb >> SB_CODEC_FREQ) & 1);
    for (uint32_t i = t; i < KSLOTS; i += WG) kset[i] = SENT;
    if (t == 0) {
        s_kcnt = 0;
        s_ksent = 0;
    }
    KE* cbuf = cbufs + w * CBUF;
    __syncthreads();
    uint32_t ccount = 0;
    auto flush = [&]() {
        for (uint32_t base = 0; base < ccount; base += 64) {
            const bool act = base + lane < ccount;
            const uint64_t rawx = act ? (uint64_t)cbuf[base + lane] : 0;
            Val<W> rv;
            __builtin_memcpy(&rv, &rawx, W);
            const Val<W> kk = stat_key<W>(rv, nk);
            const uint64_t x = k64(kk);
            if (act && !bits_eq<W>(kk, k0)) f_neq0 = 1;
            if (want_set && act && s_kcnt <= BIG_KCAP) {
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
    const uint64_t vtotal = vv.off + N;
    for (uint64_t cb = s0; cb < s1; cb += CHUNK) {
        const uint32_t n = (uint32_t)min((uint64_t)CHUNK, s1 - cb);
        const uint32_t r0 = (uint32_t)t * K;
        const uint32_t mine = r0 < n ? min((uint32_t)K, n - r0) : 0u;
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
        // the row before my first one (lane 0 of a wave: fetched — it belongs to another wave, chunk or section)
        Val<W> pvrow = shfl_val<W>(v[K - 1], (lane + 63) & 63);
        if (lane == 0) pvrow = getv(cb + r0 > 0 ? min(cb + r0 - 1, N - 1) : 0);
        const bool has_prev_row = cb + r0 > 0;
        const uint32_t m = vw.word();
        nulls += mine - (uint32_t)__popc(m);
        uint32_t sbm = 0;   // rows whose key differs from the row before (null slots included)
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
                    if (W != 4 && (j > 0 || has_prev_row) && int_lt<W>(v[j], pv_int, nk)) f_down = 1;
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
                pk = kj;
                pv_int = v[j];
            }
        }
        runs += (uint32_t)__popc(sbm);
        // a section's first row always goes to the section's set (the key may be new to THIS section)
        if (cb == s0 && t == 0 && mine) sbm |= 1u;
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
                } else {
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
    }
    flush();
    __syncthreads();
    // ---- the section's record
    const uint32_t flags = wg_or32(f_neq0 | (f_unsorted << 1) | (f_neg << 2) | ((f_down | f_unsorted) << 3), s4);
    const uint32_t null_count = wg_sum32(nulls, s4);
    const uint32_t run_count = wg_sum32(runs, s4);
    s_vk[t] = vote_k;
    s_vn[t] = vote_n;
    __syncthreads();
    for (int stride = WG / 2; stride > 0; stride >>= 1) {
        if (t < stride) {
            unsigned long long kk = s_vk[t];
            uint32_t nn = s_vn[t];
            vote_merge(kk, nn, s_vk[t + stride], s_vn[t + stride]);
            s_vk[t] = kk;
            s_vn[t] = nn;
        }
        __syncthreads();
    }
    const uint64_t mk = s_vk[0];
    const uint32_t mn = s_vn[0];
    __syncthreads();
    uint64_t mx = 0;
    if (!is_float) {   // typed maximum
        s_vk[t] = k64(tmax);
        __syncthreads();
        for (int stride = WG / 2; stride > 0; stride >>= 1) {
            if (t < stride) {
                Val<W> x, y;
                const unsigned long long xa = s_vk[t], ya = s_vk[t + stride];
                __builtin_memcpy(&x, &xa, W);
                __builtin_memcpy(&y, &ya, W);
                if (int_lt<W>(x, y, nk)) s_vk[t] = ya;
            }
            __syncthreads();
        }
        mx = s_vk[0];
    }
    uint8_t* slot = page_slot(a, c, p);
    BigSec* rec = big_sec_rec(slot, blockIdx.x / BIG_SEC_SPLIT, blockIdx.x % BIG_SEC_SPLIT);
    const uint32_t kc = want_set ? s_kcnt : 0u;
    if (t == 0) {
        BigSec r;
        __builtin_memset(&r, 0, sizeof r);
        r.flags = flags;
        r.nulls = null_count;
        r.vote_n = want_vote ? mn : 0u;
        r.vote_k = mk;
        r.kcnt = kc;
        r.tmax = mx;
        r.ksent = want_set ? s_ksent : 0u;
        r.runs = run_count;
        *rec = r;
    }
    if (want_set) {   // the section's keys join the page's set
        // One add per SECTION to the count (8 192 single adds to one word took 0.5 ms: same-address atomics queue up in L2), and
        // no count to stop at: a probe that meets BIG_UNION_PROBES occupied slots in a row calls the union too big — at a
        // quarter full that does not happen, and a false alarm only sends the page to the exact count (k_sel_big_count).
        const uint32_t nsec = (uint32_t)((N + SR_full - 1) / SR_full);
        unsigned long long* gt = big_union_tab(slot, nsec);
        uint32_t* gc = big_union_cnt(slot, nsec);
        // In batches of one key per thread, the batch's new keys added to the count before the next one: a page with far more
        // keys than the table holds (2 % random exceptions in 12 M rows: 240 000) is found out after the first batches — 700
        // workgroups pushing 300 keys each into 8 192 slots until probes met 32 occupied slots in a row took 0.2 ms.
        __syncthreads();   // (every thread has read the set's count)
        if (t == 0) s_kcnt = kc > BIG_KCAP ? 1u : 0u;   // (from here on: "the union is too big")
        __syncthreads();
        for (uint32_t i0 = 0; i0 < KSLOTS; i0 += WG) {
            if (s_kcnt) break;
            const unsigned long long x = kset[i0 + t];
            uint32_t newc = 0, over = 0;
            if (x != SENT) {
                uint32_t h = (((uint32_t)x ^ (uint32_t)(x >> 32) * 0x85EBCA6Bu) * 0x9E3779B1u >> 13) & (BIG_UNION_SLOTS - 1);
                for (uint32_t steps = 0;; steps++) {
                    if (steps >= BIG_UNION_PROBES) {
                        over = 1;
                        break;
                    }
                    const unsigned long long cur = __hip_atomic_load(gt + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (cur == x) break;
                    if (cur == SENT) {
                        unsigned long long e = SENT;
                        __hip_atomic_compare_exchange_strong(gt + h, &e, x, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (e == SENT) {
                            newc = 1;
                            break;
                        }
                        if (e == x) break;
                    }
                    h = (h + 1) & (BIG_UNION_SLOTS - 1);
                }
            }
            const uint32_t both = wg_sum32(newc | (over << 16), s4);
            if (t == 0) {
                const uint32_t tot = both & 0xFFFFu;
                bool ov = (both >> 16) != 0;
                if (tot && __hip_atomic_fetch_add(gc, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + tot > BIG_KCAP) ov = true;
                if (!ov && __hip_atomic_load(gc + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) ov = true;   // somebody saw it overflow
                if (ov) s_kcnt = 1;
            }
            __syncthreads();
        }
        if (t == 0 && s_kcnt) __hip_atomic_store(gc + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// the union table of every pending long page cleared (before k_sel_big_sec)
__global__ void __launch_bounds__(WG) k_sel_big_init(EncodeArgs a, const uint32_t* big, uint32_t voff) {
    const uint32_t page = big[blockIdx.y] + voff;
    if (a.codecs[page] != CODEC_PENDING) return;
    const EncPage p = get_page(a, page);
    const EncCol c = get_col(a, p.col);
    if (c.width > 8) return;
    const uint64_t N = p.rows, SR = big_sec_rows(N);
    const uint32_t nsec = (uint32_t)((N + SR - 1) / SR);
    uint8_t* slot = page_slot(a, c, p);
    unsigned long long* gt = big_union_tab(slot, nsec);
    for (uint32_t i = blockIdx.x * WG + threadIdx.x; i < BIG_UNION_SLOTS; i += gridDim.x * WG) gt[i] = ~0ull;
    if (blockIdx.x == 0 && threadIdx.x < 16) big_union_cnt(slot, nsec)[threadIdx.x] = 0;
}

// the sections of a page merged: flags, nulls, maximum, vote -> thread 0's PrimPartials (the other threads: neutral)
template <int W>
__device__ PrimPartials<W> big_merged_partials(const BigPage& bp, const Val<W>& first) {
    PrimPartials<W> pp;
    const bool lead = threadIdx.x == 0;
    pp.f_neq0 = lead ? (bp.flags & 1u) : 0u;
    pp.f_unsorted = lead ? ((bp.flags >> 1) & 1u) : 0u;
    pp.f_neg = lead ? ((bp.flags >> 2) & 1u) : 0u;
    pp.nulls = lead ? bp.nulls : 0u;
    pp.tmax = first;
    if (lead) __builtin_memcpy(&pp.tmax, &bp.tmax, W);
    pp.vote_k = bp.maj_k;
    pp.vote_n = lead ? bp.maj_n : 0u;
    return pp;
}

template <int W>
__device__ void big_decide(const EncodeArgs& a, const EncCol& c, const EncPage& p, uint32_t page, const BigPage& bp, const PrimCounts& pc,
                           uint32_t* lds_tab, uint32_t* s_misc, uint8_t* sample_mem) {
    const uint64_t N = p.rows;
    const uint8_t* vals = c.values + p.row0 * W;
    const ValidView vv{c.validity, c.validity_bit_offset + p.row0};
    auto getv = [=](uint64_t i) { return ld_val<W>(vals + i * W); };
    SelectOpts so{a.ratio, a.has_ratio, a.forbidden | p.forb_extra, a.default_compression, -1, p.seed, p.depth};
    SelScratch sc{lds_tab, s_misc, sample_mem, nullptr, 0};
    const uint32_t forb = so.forbidden;
    const bool want_set = !((forb >> SB_CODEC_DICT) & 1) && N >= 3;
    const bool want_vote = !((forb >> SB_CODEC_FREQ) & 1);
    const PrimPartials<W> pp = big_merged_partials<W>(bp, getv(0));
    SamplePre<W> none;
    __builtin_memset(&none, 0, sizeof none);
    // (s_kcnt as decide_prim reads it: <= KCAP means "the set holds every key")
    const uint32_t s_k = bp.set_ok ? bp.set_unique - bp.ksent : BIG_KCAP + 1, s_s = bp.ksent;
    const uint32_t codec = decide_prim<W>(getv, vv, N, c.nk, so, sc, pp, want_set, want_vote, want_set ? s_k : 0u, want_set ? s_s : 0u, none, none,
                                          none, none, false, &pc);
    if (threadIdx.x == 0) {
        a.codecs[page] = (int32_t)codec;
        atomicAdd(&a.codec_counts[codec & 31], 1u);
        if (!has_device_encoder(codec))
            raise(a.status, SB_ERR_NYI, page, 700 + codec);
        else if (codec == SB_CODEC_FREQ)
            atomicAdd(a.freq_count, 1u);
    }
}

// choose_compressor for a long binary page (binary/mod.rs:293-348; choose_bin_impl with the numbers the sections and
// k_sel_big_count found): thread 0 decides
__device__ void big_decide_bin(const EncodeArgs& a, const EncCol& c, const EncPage& p, uint32_t page, const BigPage& bp) {
    if (threadIdx.x) return;
    const uint64_t N = p.rows;
    const uint32_t forb = a.forbidden | p.forb_extra;
    const bool all_equal = !(bp.flags & 1u);
    const double tuple_count = (double)N;
    const uint64_t ow = c.ptype == SB_TYPE_BINARY ? 4 : 8;
    const double total_bytes = (double)(c.values_len_total + (N + 1) * ow);
    double max_ratio = a.ratio;
    uint32_t result = a.default_compression;
    static const uint8_t ORDER[3] = {SB_CODEC_ONEVALUE, SB_CODEC_FREQ, SB_CODEC_DICT};
    for (int oi = 0; oi < 3; oi++) {
        const uint32_t cd = ORDER[oi];
        if ((forb >> cd) & 1u) continue;
        double r = 0.0;
        if (cd == SB_CODEC_ONEVALUE) {
            r = all_equal ? tuple_count : 0.0;
        } else if (cd == SB_CODEC_FREQ) {
            if (!all_equal) {
                if ((double)bp.nulls / tuple_count >= 0.9) r = (double)(N - 1);
                else if (bp.need_mc && (double)bp.mc / tuple_count >= 0.9) r = (double)(N - 1);   // (no need_mc: the vote rules a 90 % majority out)
            }
        } else if (N >= 3 && bp.need_uq && p.aux_bytes >= big_tab_slots(N) * 8) {
            const uint64_t uq = (uint64_t)bp.uq + bp.uq_sent;
            if (uq * 3 < N) {
                uint64_t after = bp.tus + N * (uint64_t)(bits_needed(uq) / 8);
                after += N * 2 / 128;
                r = total_bytes / (double)after;
            }
        }
        if (r > max_ratio) {
            max_ratio = r;
            result = cd;
            if (r == tuple_count) break;
        }
    }
    a.codecs[page] = (int32_t)result;
    atomicAdd(&a.codec_counts[result & 31], 1u);
    if (!has_device_encoder(result)) raise(a.status, SB_ERR_NYI, page, 700 + result);
    else if (result == SB_CODEC_FREQ) atomicAdd(a.freq_count, 1u);
}

template <int W>
__global__ void __launch_bounds__(WG, 2) k_sel_big_merge(EncodeArgs a, const uint32_t* big, uint32_t voff) {
    __shared__ uint32_t lds_tab[SEL_LDS_SLOTS];
    __shared__ uint32_t s_misc[2 * WG + 16];
    __shared__ __attribute__((aligned(16))) uint8_t sample_mem[SAMPLE_CAP * (W + 1) + 16];
    __shared__ BigPage s_bp;
    uint32_t page;
    EncPage p;
    EncCol c;
    if (!big_page_of(a, big, W, &page, &p, &c, voff)) return;
    const uint64_t N = p.rows, SR = big_sec_rows(N);
    const uint32_t nsec = (uint32_t)((N + SR - 1) / SR);
    const int t = threadIdx.x;
    const uint32_t nk = c.nk;
    const bool is_float = nk >= NK_F32;
    const uint32_t forb = a.forbidden | p.forb_extra;
    const bool want_set = !((forb >> SB_CODEC_DICT) & 1) && N >= 3;
    const bool want_vote = !((forb >> SB_CODEC_FREQ) & 1);
    uint8_t* slot = page_slot(a, c, p);
    uint32_t* s4 = s_misc + 2 * WG;
    // ---- thread = section
    BigSec r;
    __builtin_memset(&r, 0, sizeof r);
    const bool has = (uint32_t)t < nsec;
    if (has) {   // the records of the section's parts (the first one always exists)
        r = *big_sec_rec(slot, (uint32_t)t);
        for (uint32_t q = 1; q < BIG_SEC_SPLIT && (uint64_t)t * SR + q * (SR / BIG_SEC_SPLIT) < N; q++) {
            const BigSec o = *big_sec_rec(slot, (uint32_t)t, q);
            r.flags |= o.flags;
            r.nulls += o.nulls;
            r.runs += o.runs;
            r.ksent |= o.ksent;
            r.kcnt = max(r.kcnt, o.kcnt);   // (only "some part's set overflowed" is read from it)
            Val<W> x, y;
            __builtin_memcpy(&x, &r.tmax, W);
            __builtin_memcpy(&y, &o.tmax, W);
            if (!is_float && int_lt<W>(x, y, nk)) r.tmax = o.tmax;
            unsigned long long vk0 = r.vote_k;
            vote_merge(vk0, r.vote_n, o.vote_k, o.vote_n);
            r.vote_k = vk0;
        }
    }
    const uint32_t flags = wg_or32(r.flags, s4);
    const uint32_t null_count = wg_sum32(r.nulls, s4);
    const uint32_t ksent = wg_or32(r.ksent, s4);
    const uint32_t over = wg_or32(has && r.kcnt > BIG_KCAP ? 1u : 0u, s4);
    const uint32_t run_total = wg_sum32(has ? r.runs : 0u, s4);
    unsigned long long* vk = (unsigned long long*)sample_mem;
    uint32_t* vn = s_misc;
    vk[t] = r.vote_k;
    vn[t] = has ? r.vote_n : 0u;
    __syncthreads();
    for (int stride = WG / 2; stride > 0; stride >>= 1) {
        if (t < stride) {
            unsigned long long kk = vk[t];
            uint32_t nn = vn[t];
            vote_merge(kk, nn, vk[t + stride], vn[t + stride]);
            vk[t] = kk;
            vn[t] = nn;
        }
        __syncthreads();
    }
    const uint64_t maj_k = vk[0];
    const uint32_t maj_n = vn[0];
    __syncthreads();
    uint64_t mx = 0;
    if (!is_float) {
        vk[t] = has ? r.tmax : big_sec_rec(slot, 0)->tmax;
        __syncthreads();
        for (int stride = WG / 2; stride > 0; stride >>= 1) {
            if (t < stride) {
                Val<W> x, y;
                const unsigned long long xa = vk[t], ya = vk[t + stride];
                __builtin_memcpy(&x, &xa, W);
                __builtin_memcpy(&y, &ya, W);
                if (int_lt<W>(x, y, nk)) vk[t] = ya;
            }
            __syncthreads();
        }
        mx = vk[0];
        __syncthreads();
    }
    // ---- the union of the sections' key sets
    bool set_ok = false;
    uint32_t set_unique = 0;
    const bool bin = big_is_bin(c);
    if (want_set && !over && !bin) {   // (the sections united their sets in the page's table: k_sel_big_sec)
        const uint32_t uc = big_union_cnt(slot, nsec)[0], uover = big_union_cnt(slot, nsec)[1];
        set_ok = !uover && uc <= BIG_KCAP;
        set_unique = uc + (ksent ? 1u : 0u);
    }
    // integers that never step down (sorted ids, timestamps): equal keys are neighbours, so the distinct keys are the runs —
    // no table, no count pass (a sorted 12 M-row Int64 page: k_sel_big_clear + k_sel_big_count were 0.4 of its 0.6 ms)
    const bool sorted_uq = want_set && !bin && !set_ok && !is_float && !(flags & 8u);
    // ---- what is missing for the decision?
    const bool all_equal = !(flags & 1u);
    const double tuple_count = (double)N;
    // (binary: set_ok is never true; choose_bin_impl looks at Dict's ratio on an all-equal page too unless OneValue took it)
    const bool need_uq = bin ? want_set && !(all_equal && !((forb >> SB_CODEC_ONEVALUE) & 1)) : want_set && !all_equal && !set_ok && !sorted_uq;
    const bool need_mc = want_vote && !all_equal && !((double)null_count / tuple_count >= 0.9) && ((double)maj_n + 1.0 >= 0.8 * tuple_count);
    if (t == 0) {
        BigPage b;
        __builtin_memset(&b, 0, sizeof b);
        b.flags = flags;
        b.nulls = null_count;
        b.maj_n = maj_n;
        b.maj_k = maj_k;
        b.tmax = mx;
        b.set_ok = set_ok ? 1u : 0u;
        b.set_unique = set_unique;
        b.ksent = ksent ? 1u : 0u;
        b.need_uq = need_uq ? 1u : 0u;
        b.need_mc = need_mc ? 1u : 0u;
        b.sorted_uq = sorted_uq ? 1u : 0u;
        b.run_uq = run_total;
        s_bp = b;
        *big_page_rec(slot) = b;
    }
    __syncthreads();
    if (need_uq || need_mc) return;   // k_sel_big_count, then k_sel_big_decide
    const BigPage bp = s_bp;
    const PrimCounts pc{bp.sorted_uq != 0, false, bp.run_uq, 0};
    __syncthreads();
    if (bin) big_decide_bin(a, c, p, page, bp);
    else big_decide<W>(a, c, p, page, bp, pc, lds_tab, s_misc, sample_mem);
}

__global__ void __launch_bounds__(WG) k_sel_big_clear(EncodeArgs a, const uint32_t* big, uint32_t voff) {
    const uint32_t page = big[blockIdx.y] + voff;
    if (a.codecs[page] != CODEC_PENDING) return;
    const EncPage p = get_page(a, page);
    const EncCol c = get_col(a, p.col);
    if (c.width > 8) return;
    const BigPage* bp = big_page_rec(page_slot(a, c, p));
    const uint64_t M = big_tab_slots(p.rows);
    if (!bp->need_uq || p.aux_bytes < M * 8) return;   // (the host sizes the aux area of a long page for this table)
    u32x4* tab = (u32x4*)(a.scratch + p.aux_off);   // (aux areas are 16-byte aligned)
    const u32x4 e = {SEL_EMPTY, SEL_EMPTY, SEL_EMPTY, SEL_EMPTY};
    for (uint64_t i = (uint64_t)blockIdx.x * WG + threadIdx.x; i < M / 2; i += (uint64_t)gridDim.x * WG) tab[i] = e;
}

template <int W>
__global__ void __launch_bounds__(WG, 4) k_sel_big_count(EncodeArgs a, const uint32_t* big, uint32_t voff) {
    __shared__ uint32_t s4[4];
    __shared__ uint32_t s_stop;
    // keys this section has looked up already: a column with a few frequent values (zipf text: the top ten words are 40 %
    // of the rows) would send every one of those rows to the SAME few words of the page's table — loads of one address from
    // all CUs queue up in one L2 channel (3 M zipf rows: 1.6 ms for the pass)
    constexpr uint32_t LSLOTS = 4096, LCAP = 3072;
    __shared__ unsigned long long lseen[LSLOTS];
    __shared__ uint32_t s_lcnt, s_use;
    uint32_t page;
    EncPage p;
    EncCol c;
    if (!big_page_of(a, big, W, &page, &p, &c, voff)) return;
    uint8_t* slot = page_slot(a, c, p);
    BigPage* bp = big_page_rec(slot);
    const bool need_uq = bp->need_uq && p.aux_bytes >= big_tab_slots(p.rows) * 8, need_mc = bp->need_mc;
    if (!need_uq && !need_mc) return;
    // (BIG_COUNT_SPLIT workgroups per section: the probes are latency chains, the more of them in flight the better)
    const uint64_t N = p.rows, SR = big_sec_rows(N) / BIG_COUNT_SPLIT;
    const uint64_t s0 = (uint64_t)blockIdx.x * SR;
    if (s0 >= N) return;
    const uint64_t s1 = min(N, s0 + SR);
    const int t = threadIdx.x;
    const uint8_t* vals = c.values + p.row0 * W;
    const uint32_t nk = c.nk;
    auto key = [=](uint64_t i) { return stat_key<W>(ld_val<W>(vals + i * W), nk); };
    auto k64 = [&](const Val<W>& k) {
        uint64_t x = 0;
        __builtin_memcpy(&x, &k, W);
        return x;
    };
    if (need_mc) {   // rows whose key is the vote's candidate (freq.rs:129-151 counts them exactly)
        const uint64_t mk = bp->maj_k;
        uint32_t mine = 0;
        for (uint64_t i0 = s0 + t; i0 < s1; i0 += (uint64_t)WG * 8) {   // (eight loads in flight: the loop is latency otherwise)
            unsigned long long kx[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const uint64_t i = i0 + (uint64_t)u * WG;
                kx[u] = k64(key(i < s1 ? i : s0));
            }
#pragma unroll
            for (int u = 0; u < 8; u++) mine += (i0 + (uint64_t)u * WG < s1 && kx[u] == mk) ? 1u : 0u;
        }
        const uint32_t tot = wg_sum32(mine, s4);
        if (t == 0 && tot) atomicAdd(&bp->mc, tot);
    }
    if (!need_uq) return;
    // distinct keys of the page: row indices in the page's HBM table (dict.rs:109-120 wants unique * 3 < N exactly)
    const uint64_t M = big_tab_slots(N);
    unsigned long long* tab = (unsigned long long*)(a.scratch + p.aux_off);
    constexpr unsigned long long EMPTY = ~0ull;
    const uint32_t mask = (uint32_t)(M - 1);
    const uint32_t limit = (uint32_t)((N - 1) / 3);
    uint32_t sent = 0;
    const bool bin = big_is_bin(c);
    for (uint32_t i = t; i < LSLOTS; i += WG) lseen[i] = EMPTY;
    if (t == 0) {
        s_lcnt = 0;
        s_use = 1;
    }
    uint32_t nbatch = 0;
    for (uint64_t base = s0; base < s1; base += WG * 8) {
        // (the page's count is looked at before the first batch and every eighth; in between the add below tells)
        __syncthreads();
        if (t == 0 && (nbatch & 7) == 0) s_stop = __hip_atomic_load(&bp->uq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > limit ? 1u : 0u;
        nbatch++;
        __syncthreads();
        if (s_stop) break;
        unsigned long long xu[8], cu[8];
        uint32_t hu[8];
        uint32_t pend = 0, newc = 0, neww = 0, nhit = 0;
        const bool use_lds = s_use != 0;
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint64_t i = base + (uint64_t)u * WG + t;
            const Val<W> kv = key(i < s1 ? i : s0);
            xu[u] = k64(kv);
            hu[u] = stat_hash<W>(kv) & mask;
            if (i < s1) {
                if (xu[u] == EMPTY) {
                    // (binary: the all-ones hash's string counts once — whoever flips the flag adds its bytes)
                    if (bin && !sent && atomicCAS(&bp->uq_sent, 0u, 1u) == 0u) neww += big_bin_weight(c, p, i);
                    sent = 1;
                } else {
                    // the first row of this section to meet a key looks it up in the page's table; the others know it is there
                    // (once the section's table is full and hardly anything hits it — distinct values — it is left alone)
                    uint32_t lh = (uint32_t)((xu[u] * 0x9E3779B97F4A7C15ull) >> 44) & (LSLOTS - 1);
                    bool seen = false;
                    const bool room = s_lcnt < LCAP;   // (racy count: LSLOTS - LCAP slots of slack)
                    for (uint32_t st = 0; st < (use_lds ? 16u : 0u); st++) {
                        const unsigned long long cur = lseen[lh];
                        if (cur == xu[u]) {
                            seen = true;
                            break;
                        }
                        if (cur == EMPTY) {
                            if (!room) break;
                            const unsigned long long old = atomicCAS(&lseen[lh], EMPTY, xu[u]);
                            if (old == EMPTY) {
                                atomicAdd(&s_lcnt, 1u);
                                break;
                            }
                            if (old == xu[u]) {
                                seen = true;
                                break;
                            }
                        }
                        lh = (lh + 1) & (LSLOTS - 1);
                    }
                    if (!seen) pend |= 1u << u;
                    else nhit++;
                }
            }
        }
        while (pend) {
#pragma unroll
            for (int u = 0; u < 8; u++)
                if ((pend >> u) & 1) cu[u] = __hip_atomic_load(tab + hu[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int u = 0; u < 8; u++) {
                if (!((pend >> u) & 1)) continue;
                if (cu[u] == EMPTY) {
                    unsigned long long e = EMPTY;
                    __hip_atomic_compare_exchange_strong(tab + hu[u], &e, xu[u], __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (e == EMPTY) {
                        newc++;
                        if (bin) neww += big_bin_weight(c, p, base + (uint64_t)u * WG + t);
                        pend &= ~(1u << u);
                        continue;
                    }
                    cu[u] = e;
                }
                if (cu[u] == xu[u]) pend &= ~(1u << u);
                else hu[u] = (hu[u] + 1) & mask;
            }
        }
        if (use_lds && s_lcnt >= LCAP) {   // is the full table still worth its probes?
            const uint32_t hits = wg_sum32(nhit, s4);
            if (t == 0 && hits * 8 < WG * 8) s_use = 0;
        }
        const uint32_t tot = wg_sum32(newc, s4);
        if (t == 0 && tot && atomicAdd(&bp->uq, tot) + tot > limit) s_stop = 1;
        if (bin) {   // (a batch is 2 048 rows of strings: their bytes fit 32 bits)
            const uint32_t wt = wg_sum32(neww, s4);
            if (t == 0 && wt) atomicAdd(&bp->tus, (unsigned long long)wt);
        }
    }
    if (sent) bp->uq_sent = 1;
}

template <int W>
__global__ void __launch_bounds__(WG, 2) k_sel_big_decide(EncodeArgs a, const uint32_t* big, uint32_t voff) {
    __shared__ uint32_t lds_tab[SEL_LDS_SLOTS];
    __shared__ uint32_t s_misc[2 * WG + 16];
    __shared__ __attribute__((aligned(16))) uint8_t sample_mem[SAMPLE_CAP * (W + 1) + 16];
    uint32_t page;
    EncPage p;
    EncCol c;
    if (!big_page_of(a, big, W, &page, &p, &c, voff)) return;   // (chosen by k_sel_big_merge already)
    const BigPage bp = *big_page_rec(page_slot(a, c, p));
    // without an aux area (cannot happen while Dict is a candidate) the count stays unknown: "more than the limit"
    const PrimCounts pc{bp.need_uq != 0 || bp.sorted_uq != 0, bp.need_mc != 0,
                        bp.sorted_uq ? bp.run_uq : p.aux_bytes >= big_tab_slots(p.rows) * 8 ? bp.uq + bp.uq_sent : 0xFFFFFFFEu, bp.mc};
    if (big_is_bin(c)) big_decide_bin(a, c, p, page, bp);
    else big_decide<W>(a, c, p, page, bp, pc, lds_tab, s_misc, sample_mem);
}

// ---------------------------------------------------------------------------------------------------- RLE of a long page
// integer/rle.rs:64-104: records (u32 count | value); a run boundary is a VALID row whose canonical key differs from the key
// of the previous valid row, nulls extend the current run, run 0 starts at row 0.  Everything a section needs from the
// rows before it is three words — is there a valid row before me, its key, the row where the current run began — so:
//   k_rle_big_count  (sections x pages)   the section walked as if it were the start of a page: boundaries found, first /
//                                         last valid key, last boundary row
//   k_rle_big_plan   (1 workgroup / page) scans over the sections: carry-in of every section (key, run start, records
//                                         before it); closes the last run, def levels, block header, page record
//   k_rle_big_emit   (sections x pages)   the same walk again from its carry-in, records written in place
// The walk is select_rle_page's (speculative RLE of the page selectors), restricted to a section.
struct RleSec {   // 64 bytes per section, at the tail of the page's slot (behind anything an RLE page can write)
    uint32_t nb, has, first_row_lo, pad0;        // boundaries inside the section (its first valid row not counted)
    uint64_t first_key, last_key;                // canonical keys of its first / last valid row
    uint64_t last_boundary;                      // row + 1 of its last boundary, 0 = none
    uint64_t first_row;                          // its first valid row
    // carry-in, written by k_rle_big_plan
    uint32_t in_has, in_nrec;
    uint64_t in_key, in_start;
};
static_assert(sizeof(RleSec) <= 96, "RleSec fits its stride");
constexpr uint32_t RLE_SEC_STRIDE = 96;
template <int W>
__device__ __forceinline__ RleSec* rle_sec_rec(uint8_t* slot, uint64_t N, bool nullable, uint32_t s) {
    // slot capacity >= 64 + def + 18 + N (W + 8) + 68 (slot_fixed_bytes); an RLE page ends below def + 9 + (N + 1) (4 + W)
    const uint64_t off = (64 + (nullable ? def_section_bytes(N) : 0) + 18 + N * (uint64_t)(W + 8) - SEL_BIG_SECTIONS * RLE_SEC_STRIDE) & ~(uint64_t)15;
    return (RleSec*)(slot + off + (uint64_t)s * RLE_SEC_STRIDE);
}

// rows [s0, s1) of the page: boundaries among its valid rows.  STORE: records are written (carry-in state given);
// otherwise only the state is tracked.  Returns through the references: have / last (key of the last valid row) / run_start
// (row where the current run began) / nrec (boundaries so far); first_* describe the first valid row met.
template <int W, bool STORE>
__device__ void rle_walk_section(const uint8_t* vals, const ValidView& vv, uint32_t nk, uint64_t N, uint64_t s0, uint64_t s1, uint8_t* dst,
                                 bool& have, Val<W>& last, uint64_t& run_start, uint32_t& nrec, bool& first_seen, Val<W>& first_key,
                                 uint64_t& first_row, uint32_t* sA, uint32_t* sB) {
    constexpr int K = 16;
    constexpr uint32_t CHUNK = WG * K;
    constexpr int REC = 4 + W;
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint64_t lt = (1ull << lane) - 1;
    auto getv = [=](uint64_t i) { return ld_val<W>(vals + i * W); };
    auto lds_barrier = []() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    };
    const uint64_t vtotal = vv.off + N;
    uint32_t par = 0;
    for (uint64_t cb = s0; cb < s1; cb += CHUNK, par ^= 1) {
        const uint32_t n = (uint32_t)min((uint64_t)CHUNK, s1 - cb);
        const uint32_t r0 = (uint32_t)t * K;
        const uint32_t mine = r0 < n ? min((uint32_t)K, n - r0) : 0u;
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
        const uint32_t m = vw.word();
        uint32_t bmask = 0;  // valid rows whose key differs from the previous VALID row of this thread: run starts
        Val<W> ek = val_zero<W>(), firstk = val_zero<W>(), firstv = val_zero<W>();
        bool seen = false;
#pragma unroll
        for (int j = 0; j < K; j++) {
            if ((m >> j) & 1) {
                const Val<W> kj = stat_key<W>(v[j], nk);
                if (!seen) {
                    firstk = kj;
                    firstv = v[j];
                } else if (!bits_eq<W>(kj, ek)) {
                    bmask |= 1u << j;
                }
                ek = kj;
                seen = true;
            }
        }
        uint32_t* s_has = sA + par * 16;
        uint32_t* s_cnt = sA + par * 16 + 4;
        uint32_t* s_blast = sA + par * 16 + 8;
        Val<W>* s_last = (Val<W>*)sB + par * 4;
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
            if (!pc) {   // the first valid row of everything walked so far (with the carry-in: of the page)
                if (STORE) st_val<W>(dst + 4, firstv);
                if (!first_seen) {   // (one thread: nobody before it has a valid row)
                    first_key = firstk;
                    first_row = cb + r0 + f;
                }
            } else if (!bits_eq<W>(pvk, firstk)) {
                bmask |= 1u << f;
            }
        }
        // who saw the first valid row?  the lowest lane of the lowest wave with one: broadcast through LDS below
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
        // the section's first valid row (key, row): held by one thread; pass it to everybody through LDS
        unsigned long long* s_first = (unsigned long long*)(sB + 64) + par * 2;   // [0] key bits, [1] row + 1
        if (t == 0) s_first[1] = 0;
        lds_barrier();
        if (!first_seen && m && !pc) {
            uint64_t kb = 0;
            __builtin_memcpy(&kb, &firstk, W);
            s_first[0] = kb;
            s_first[1] = cb + r0 + (uint32_t)(__ffs((int)m) - 1) + 1;
        }
        uint32_t base = nrec;
        uint64_t start_prev = run_start;
        for (int pw = 0; pw < 3; pw++)
            if (pw < w) {
                base += s_cnt[pw];
                if (s_blast[pw]) start_prev = cb + s_blast[pw] - 1;
            }
        if (STORE && bmask) {
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
        lds_barrier();
        if (!first_seen && s_first[1]) {
            first_seen = true;
            const unsigned long long kb = s_first[0];
            __builtin_memcpy(&first_key, &kb, W);
            first_row = s_first[1] - 1;
        }
    }
}

template <int W>
__device__ __forceinline__ bool rle_big_page_of(const EncodeArgs& a, const uint32_t* big, uint32_t* page, EncPage* p, EncCol* c, uint32_t voff) {
    *page = big[blockIdx.y] + voff;
    if (a.codecs[*page] != (int32_t)SB_CODEC_RLE || a.outs[*page].length != 0) return false;
    *p = get_page(a, *page);
    *c = get_col(a, p->col);
    return (int)c->width == W && p->rows >= (voff ? 65536u : SEL_BIG_ROWS);   // (virtual pages: VBIG_ROWS, sb_dict_big.h)
}

template <int W>
__global__ void __launch_bounds__(WG, 4) k_rle_big_count(EncodeArgs a, const uint32_t* big, uint32_t voff) {
    __shared__ uint32_t sA[32];
    __shared__ __attribute__((aligned(16))) uint32_t sB[64 + 8];
    if (a.use_counts && a.codec_counts[SB_CODEC_RLE] == 0) return;
    uint32_t page;
    EncPage p;
    EncCol c;
    if (!rle_big_page_of<W>(a, big, &page, &p, &c, voff)) return;
    const uint64_t N = p.rows, SR = big_sec_rows(N);
    const uint64_t s0 = (uint64_t)blockIdx.x * SR;
    if (s0 >= N) return;
    const uint64_t s1 = min(N, s0 + SR);
    const uint8_t* vals = c.values + p.row0 * W;
    const ValidView vv{c.validity, c.validity_bit_offset + p.row0};
    bool have = false, first_seen = false;
    Val<W> last = val_zero<W>(), first_key = val_zero<W>();
    uint64_t run_start = 0, first_row = 0;
    uint32_t nrec = 0;
    rle_walk_section<W, false>(vals, vv, c.nk, N, s0, s1, nullptr, have, last, run_start, nrec, first_seen, first_key, first_row, sA, sB);
    if (threadIdx.x == 0) {
        RleSec* r = rle_sec_rec<W>(page_slot(a, c, p), N, c.nullable != 0, blockIdx.x);
        RleSec o;
        __builtin_memset(&o, 0, sizeof o);
        o.nb = nrec;
        o.has = have ? 1u : 0u;
        __builtin_memcpy(&o.first_key, &first_key, W);
        __builtin_memcpy(&o.last_key, &last, W);
        o.last_boundary = nrec ? run_start + 1 : 0;
        o.first_row = first_row;
        *r = o;
    }
}

template <int W>
__global__ void __launch_bounds__(WG) k_rle_big_plan(EncodeArgs a, const uint32_t* big, uint32_t voff) {
    if (a.use_counts && a.codec_counts[SB_CODEC_RLE] == 0) return;
    uint32_t page;
    EncPage p;
    EncCol c;
    if (!rle_big_page_of<W>(a, big, &page, &p, &c, voff)) return;
    constexpr int REC = 4 + W;
    const uint64_t N = p.rows, SR = big_sec_rows(N);
    const uint32_t nsec = (uint32_t)((N + SR - 1) / SR);
    uint8_t* slot = page_slot(a, c, p);
    const uint64_t pos = c.nullable ? def_section_bytes(N) : 0;
    uint8_t* dst = slot + pos + 9;
    // <= 256 sections: every thread fetches one record, one thread walks them in LDS, every thread stores its carry-in
    __shared__ RleSec s_sec[SEL_BIG_SECTIONS];
    __shared__ uint32_t s_total;
    __shared__ unsigned long long s_last_start;
    __shared__ uint32_t s_any;
    if (threadIdx.x < nsec) s_sec[threadIdx.x] = *rle_sec_rec<W>(slot, N, c.nullable != 0, threadIdx.x);
    __syncthreads();
    if (threadIdx.x == 0) {
        bool has = false;
        uint64_t key = 0, start = 0;
        uint32_t nrec = 0;
        for (uint32_t s = 0; s < nsec; s++) {
            RleSec& o = s_sec[s];
            o.in_has = has ? 1u : 0u;
            o.in_key = key;
            o.in_start = start;
            o.in_nrec = nrec;
            if (o.has) {
                if (has && o.first_key != key) {   // the section's first valid row opens a run
                    nrec++;
                    start = o.first_row;
                }
                nrec += o.nb;
                if (o.last_boundary) start = o.last_boundary - 1;
                has = true;
                key = o.last_key;
            }
        }
        s_total = nrec;
        s_last_start = start;
        s_any = has ? 1u : 0u;
    }
    __syncthreads();
    if (threadIdx.x < nsec) *rle_sec_rec<W>(slot, N, c.nullable != 0, threadIdx.x) = s_sec[threadIdx.x];
    if (c.nullable) {
        uint8_t* bits = def_header(slot, N);
        def_bits_page(bits, ValidView{c.validity, c.validity_bit_offset}, p.row0, N, c.rows);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t nrec = s_total;
        uint8_t* r = dst + (uint64_t)nrec * REC;
        stu32(r, (uint32_t)(N - s_last_start));   // the last run
        if (!s_any) st_val<W>(r + 4, val_zero<W>());
        const uint64_t body = (uint64_t)(nrec + 1) * REC;
        put_hdr9(slot + pos, SB_CODEC_RLE, (uint32_t)body, (uint32_t)(N * W));
    }
}

template <int W>
__global__ void __launch_bounds__(WG, 4) k_rle_big_emit(EncodeArgs a, const uint32_t* big, uint32_t voff) {
    __shared__ uint32_t sA[32];
    __shared__ __attribute__((aligned(16))) uint32_t sB[64 + 8];
    if (a.use_counts && a.codec_counts[SB_CODEC_RLE] == 0) return;
    uint32_t page;
    EncPage p;
    EncCol c;
    if (!rle_big_page_of<W>(a, big, &page, &p, &c, voff)) return;
    const uint64_t N = p.rows, SR = big_sec_rows(N);
    const uint64_t s0 = (uint64_t)blockIdx.x * SR;
    if (s0 >= N) return;
    const uint64_t s1 = min(N, s0 + SR);
    const uint8_t* vals = c.values + p.row0 * W;
    const ValidView vv{c.validity, c.validity_bit_offset + p.row0};
    uint8_t* slot = page_slot(a, c, p);
    const uint64_t pos = c.nullable ? def_section_bytes(N) : 0;
    uint8_t* dst = slot + pos + 9;
    const RleSec in = *rle_sec_rec<W>(slot, N, c.nullable != 0, blockIdx.x);
    bool have = in.in_has != 0, first_seen = true;
    Val<W> last, first_key = val_zero<W>();
    __builtin_memcpy(&last, &in.in_key, W);
    uint64_t run_start = in.in_start, first_row = 0;
    uint32_t nrec = in.in_nrec;
    rle_walk_section<W, true>(vals, vv, c.nk, N, s0, s1, dst, have, last, run_start, nrec, first_seen, first_key, first_row, sA, sB);
}

// the page records of the RLE pages written above (after k_rle_big_emit: a.outs marks a page as emitted)
template <int W>
__global__ void k_rle_big_done(EncodeArgs a, const uint32_t* big, uint32_t voff) {
    if (a.use_counts && a.codec_counts[SB_CODEC_RLE] == 0) return;
    uint32_t page;
    EncPage p;
    EncCol c;
    if (threadIdx.x) return;
    if (!rle_big_page_of<W>(a, big, &page, &p, &c, voff)) return;
    uint8_t* slot = page_slot(a, c, p);
    const uint64_t pos = c.nullable ? def_section_bytes(p.rows) : 0;
    const uint32_t body = ldu32(slot + pos + 1);
    EncOut out;
    out.length = pos + 9 + body;
    out.out_off = 0;
    out.slot = slot;
    out.codec = SB_CODEC_RLE;
    out.pad = 1;   // emitted here: k_enc_emit_pages<., RLE> leaves the page alone
    a.outs[page] = out;
}
