// strawboat-hip: Dict pages of LONG pages written section-parallel (included by sb_encode.hip inside namespace sb, after
// sb_select_big.h).
//
// The reference's default paging makes a column ONE page (src/write/common.rs:54-58), and a low-cardinality column of
// millions of rows then chooses Dict (src/compression/integer/dict.rs:33-73): distinct values in first-occurrence order,
// one u32 index per row (a null row repeats the index before it, :46-55), the indices compressed as a nested
// compress_integer::<u32> block with Dict forbidden (:57-62), then `u32 n | n x value`.  k_enc_emit_pages<W, Dict> walks
// such a page with ONE workgroup (12 M rows: 78 ms).  Pages of SEL_BIG_ROWS rows or more that k_sel_big_* gave to Dict go
// through these kernels instead (bytes identical):
//
//   k_dict_big_clear   (grid x pages)      the page's key table (keys in the Dict aux area, first rows in the bigx area)
//   k_dict_big_insert  (sections x pages)  every keyed row's key -> (key, smallest row): a section first collects its own
//                                          (key, smallest row) pairs in an LDS table and hands them over at its end; keys
//                                          beyond the LDS table go to the HBM table row by row.  Float NaNs never equal
//                                          anything (dict.rs:208,225-229): every NaN row is an entry of its own
//   k_dict_big_mark    (grid x pages)      first rows -> a bitmap over the page's rows
//   k_dict_big_rank    (sections x pages)  prefix popcounts of the bitmap: id of an entry = rank of its first row
//   k_dict_big_ids     (grid x pages)      table slot -> id; firsts[id] = first row
//   k_dict_big_idx     (sections x pages)  row -> id of its key (nulls: forward fill, carried across waves / chunks /
//                                          sections), the u32 index array; the index array becomes a VIRTUAL page (entry
//                                          n_pages + page of the tables) of 4-byte unsigned values
//   k_sel_big_* <4>, k_rle_big_* <4>, k_bp_big_*, k_plain_big  (on the virtual pages) the nested block: the same
//                                          section-parallel selector, then the section- / tile-parallel writers
//   k_dict_big_finish  (1 workgroup / page) def levels, anything the parallel writers do not cover (Zstd / Snappy / exact
//                                          LZ4 / Freq-coded indices: the one-workgroup encoders), block header, page record
//   k_dict_big_values  (grid x pages)      the nested block moved behind the header, `u32 n`, the dictionary values
//
// The bigx area of a page (host-allocated for pages of >= SEL_BIG_ROWS rows when Dict is a candidate): record | idx[N] |
// firsts[N] | bitmap[N/32] | word prefixes[N/32] | first rows / ids of the table slots | the nested block's slot.
constexpr uint32_t DBIG_LDS_SLOTS = 4096, DBIG_LDS_CAP = 3072;
constexpr uint32_t DBIG_SPLIT = 4;   // workgroups per section in the row passes (insert, idx): 184 sections of a 12 M-row page leave CUs idle
constexpr unsigned long long DBIG_EMPTY = ~0ull;

struct DictBigRec {   // 4096 bytes
    uint32_t active, M3, D, sent_first, sent_id, nested_ok, bad, pad0;
    uint32_t sectot[SEL_BIG_SECTIONS];
    unsigned long long secbytes[SEL_BIG_SECTIONS];   // binary pages: entry bytes (8 + len of every first row) per section
    unsigned long long sent_eoff, etotal;
    uint32_t pad1[1024 - 8 - 3 * SEL_BIG_SECTIONS - 4];
};
static_assert(sizeof(DictBigRec) == 4096, "record of the long-page Dict writer");

__host__ __device__ __forceinline__ uint64_t dbig_slots(uint64_t unique) {
    uint64_t m = 1024;
    while (m < 2 * (unique + 8)) m <<= 1;
    return m;
}
__host__ __device__ __forceinline__ uint64_t dbig_slots_max(uint64_t N) { return dbig_slots((N - 1) / 3 + 2); }
__host__ __device__ __forceinline__ uint64_t dbig_nblk_cap(uint64_t N) { return (64 + 9 + 9 + N * 12 + 4 + 64 + 64 + 15) & ~15ull; }
constexpr uint64_t BIGX_FREQ_OFF = 4096, BIGX_FREQ_BYTES = 65536;   // the long-page Freq writer's record (sb_freq_big.h)
constexpr uint64_t BIGX_HEAD = BIGX_FREQ_OFF + BIGX_FREQ_BYTES;     // a page for which Dict is forbidden has only this much
constexpr uint32_t VBIG_ROWS = 65536;   // virtual pages (index arrays, Freq exceptions) of this many rows take the section-parallel path
constexpr uint32_t VPAD_ACTIVE = 7, VPAD_PLANNED = 6;   // EncOut.pad of a virtual page on that path: not written yet / sized, being written
struct DictBigLayout {
    uint64_t o_idx, o_firsts, o_bits, o_wpref, o_frow, o_nblk, total, nwords;
    uint64_t o_keys, o_wbytes, o_eoff;   // binary pages: the key table (the aux area holds the selector's), byte prefixes, entry offsets
};
__host__ __device__ __forceinline__ DictBigLayout dbig_layout(uint64_t N, bool binary = true) {
    DictBigLayout l;
    l.nwords = (N + 31) / 32;
    l.o_idx = BIGX_HEAD;
    l.o_firsts = l.o_idx + ((N * 4 + 63) & ~63ull);
    l.o_bits = l.o_firsts + ((N * 4 + 63) & ~63ull);
    l.o_wpref = l.o_bits + ((l.nwords * 4 + 63) & ~63ull);
    l.o_frow = l.o_wpref + ((l.nwords * 4 + 63) & ~63ull);
    l.o_nblk = l.o_frow + dbig_slots_max(N) * 4;
    l.o_keys = l.o_nblk + dbig_nblk_cap(N);
    l.total = l.o_keys;
    if (binary) {
        l.o_wbytes = l.o_keys + dbig_slots_max(N) * 8;
        l.o_eoff = l.o_wbytes + ((l.nwords * 8 + 63) & ~63ull);
        l.total = l.o_eoff + ((N - 1) / 3 + 64) * 8;
    }
    return l;
}

struct DictBigCtx {
    uint32_t page;
    EncPage p;
    EncCol c;
    uint8_t* bigx;
    DictBigRec* rec;
    DictBigLayout l;
    unsigned long long* keys;   // the aux area
    uint32_t *idx, *firsts, *bits, *wpref, *frow;
    uint8_t* nblk;
};
// the page of this block (blockIdx.y) if it is a long page of kind KIND (1 / 2 / 4 / 8: primitives of that width; -4 / -8:
// Binary / LargeBinary) that chose Dict and has not been written
template <int KIND>
__device__ __forceinline__ bool dbig_page_of(const EncodeArgs& a, const uint32_t* big, DictBigCtx* d, bool need_active = true) {
    d->page = big[blockIdx.y];
    if (a.codecs[d->page] != (int32_t)SB_CODEC_DICT || a.outs[d->page].length != 0) return false;
    d->p = a.pages[d->page];
    d->c = a.cols[d->p.col];
    if (!d->p.bigx_off) return false;
    if constexpr (KIND > 0) {
        if (big_is_bin(d->c) || (int)d->c.width != KIND) return false;
    } else {
        if (d->c.ptype != (KIND == -4 ? SB_TYPE_BINARY : SB_TYPE_LARGE_BINARY) || d->p.h64_off == ~0ull) return false;
    }
    d->bigx = a.scratch + d->p.bigx_off;
    d->rec = (DictBigRec*)d->bigx;
    d->l = dbig_layout(d->p.rows);
    d->keys = KIND > 0 ? (unsigned long long*)(a.scratch + d->p.aux_off) : (unsigned long long*)(d->bigx + d->l.o_keys);
    d->idx = (uint32_t*)(d->bigx + d->l.o_idx);
    d->firsts = (uint32_t*)(d->bigx + d->l.o_firsts);
    d->bits = (uint32_t*)(d->bigx + d->l.o_bits);
    d->wpref = (uint32_t*)(d->bigx + d->l.o_wpref);
    d->frow = (uint32_t*)(d->bigx + d->l.o_frow);
    d->nblk = d->bigx + d->l.o_nblk;
    return !need_active || d->rec->active != 0;
}
// the key of row i: the value's raw bits (a leading null interns T::default(), dict.rs:46-50) / the 64-bit hash of the
// row's string (binary/dict.rs:55-93 interns the slot's bytes whatever the validity says)
template <int KIND>
struct DbigKeys {
    const uint8_t* vals;   // page values / the page's h64 array
    __device__ __forceinline__ unsigned long long key(uint64_t i, bool valid) const {
        constexpr int W = KIND > 0 ? KIND : 8;
        unsigned long long x = 0;
        if constexpr (KIND > 0) {
            Val<W> v = ld_val<W>(vals + i * W);
            if (i == 0 && !valid) v = val_zero<W>();
            __builtin_memcpy(&x, &v, W);
        } else {
            x = ldu64(vals + i * 8);
        }
        return x;
    }
};
template <int KIND>
__device__ __forceinline__ DbigKeys<KIND> dbig_keys(const EncodeArgs& a, const DictBigCtx& d) {
    if constexpr (KIND > 0) return DbigKeys<KIND>{d.c.values + d.p.row0 * KIND};
    else return DbigKeys<KIND>{a.scratch + d.p.h64_off};
}
template <int W>
__device__ __forceinline__ bool dbig_is_nan(uint64_t x, uint32_t fkind) {
    if constexpr (W == 4) return fkind == 1 && (x & 0x7FFFFFFFu) > 0x7F800000u;
    if constexpr (W == 8) return fkind == 2 && (x & 0x7FFFFFFFFFFFFFFFull) > 0x7FF0000000000000ull;
    return false;
}
__device__ __forceinline__ uint32_t dbig_hash(uint64_t x) { return hash64(x + 0x9E3779B97F4A7C15ull); }

// slot of key x in the page's HBM table, inserting it when it is new
__device__ __forceinline__ uint32_t dbig_global_slot(unsigned long long* keys, uint32_t mask, unsigned long long x) {
    uint32_t h = dbig_hash(x) & mask;
    for (;;) {
        unsigned long long cur = __hip_atomic_load(keys + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == x) return h;
        if (cur == DBIG_EMPTY) {
            unsigned long long e = DBIG_EMPTY;
            __hip_atomic_compare_exchange_strong(keys + h, &e, x, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (e == DBIG_EMPTY || e == x) return h;
        }
        h = (h + 1) & mask;
    }
}
__device__ __forceinline__ void dbig_global_min(uint32_t* frow, uint32_t h, uint32_t row) {
    if (row < __hip_atomic_load(frow + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        __hip_atomic_fetch_min(frow + h, row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// slot of key x (present) — read only
__device__ __forceinline__ uint32_t dbig_global_find(const unsigned long long* keys, uint32_t mask, unsigned long long x) {
    uint32_t h = dbig_hash(x) & mask;
    for (;;) {
        const unsigned long long cur = keys[h];
        if (cur == x || cur == DBIG_EMPTY) return h;
        h = (h + 1) & mask;
    }
}

template <int KIND>
__global__ void __launch_bounds__(WG) k_dict_big_clear(EncodeArgs a, const uint32_t* big) {
    [[maybe_unused]] constexpr int W = KIND > 0 ? KIND : 8;   // bytes of a key
    if (a.use_counts && a.codec_counts[SB_CODEC_DICT] == 0) return;
    DictBigCtx d;
    if (!dbig_page_of<KIND>(a, big, &d, false)) return;
    const uint64_t N = d.p.rows;
    const BigPage bp = *big_page_rec(page_slot(a, d.c, d.p));
    const bool known = bp.set_ok || bp.need_uq;
    const uint64_t unique = bp.set_ok ? bp.set_unique : (uint64_t)bp.uq + bp.uq_sent;
    const uint64_t M3 = dbig_slots(unique);
    const bool ok = known && M3 <= dbig_slots_max(N) && (KIND < 0 || M3 * 8 <= d.p.aux_bytes) && N < 0xFFFFFFF0ull;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        d.rec->active = ok ? 1u : 0u;
        d.rec->M3 = (uint32_t)M3;
        d.rec->D = 0;
        d.rec->sent_first = 0xFFFFFFFFu;
        d.rec->sent_id = 0;
        d.rec->nested_ok = 0;
        d.rec->bad = 0;
        d.rec->sent_eoff = 0;
        d.rec->etotal = 0;
    }
    if (!ok) return;
    const uint64_t tid = (uint64_t)blockIdx.x * WG + threadIdx.x, nth = (uint64_t)gridDim.x * WG;
    for (uint64_t i = tid; i < M3; i += nth) {
        d.keys[i] = DBIG_EMPTY;
        d.frow[i] = 0xFFFFFFFFu;
    }
    for (uint64_t i = tid; i < d.l.nwords; i += nth) d.bits[i] = 0;
}

template <int KIND>
__global__ void __launch_bounds__(WG, 3) k_dict_big_insert(EncodeArgs a, const uint32_t* big) {
    [[maybe_unused]] constexpr int W = KIND > 0 ? KIND : 8;   // bytes of a key
    __shared__ unsigned long long lk[DBIG_LDS_SLOTS];
    __shared__ uint32_t lr[DBIG_LDS_SLOTS];
    __shared__ uint32_t s_cnt;
    if (a.use_counts && a.codec_counts[SB_CODEC_DICT] == 0) return;
    DictBigCtx d;
    if (!dbig_page_of<KIND>(a, big, &d)) return;
    const uint64_t N = d.p.rows, SR = big_sec_rows(N) / DBIG_SPLIT;
    const uint64_t s0 = (uint64_t)blockIdx.x * SR;
    if (s0 >= N) return;
    const uint64_t s1 = min(N, s0 + SR);
    const int t = threadIdx.x;
    const DbigKeys<KIND> kf = dbig_keys<KIND>(a, d);
    const ValidView vv{d.c.validity, d.c.validity_bit_offset + d.p.row0};
    const uint32_t fkind = d.c.fkind;
    const uint32_t mask = d.rec->M3 - 1;
    for (uint32_t i = t; i < DBIG_LDS_SLOTS; i += WG) {
        lk[i] = DBIG_EMPTY;
        lr[i] = 0xFFFFFFFFu;
    }
    if (t == 0) s_cnt = 0;
    __syncthreads();
    for (uint64_t base = s0; base < s1; base += (uint64_t)WG * 8) {
        unsigned long long xu[8];
        uint32_t keyed = 0;
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint64_t i = base + (uint64_t)u * WG + t;
            const bool in = i < s1;
            const bool valid = in && vv.get(i);
            xu[u] = kf.key(in ? i : s0, valid);
            if (valid || (in && i == 0)) keyed |= 1u << u;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (!((keyed >> u) & 1)) continue;
            const uint32_t row = (uint32_t)(base + (uint64_t)u * WG + t);
            const unsigned long long x = xu[u];
            if (dbig_is_nan<W>(x, fkind)) {   // an entry of its own: its row is a first row
                atomicOr(d.bits + (row >> 5), 1u << (row & 31));
                continue;
            }
            if (W == 8 && x == DBIG_EMPTY) {   // the table's "empty": kept beside the table
                if (row < __hip_atomic_load(&d.rec->sent_first, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                    __hip_atomic_fetch_min(&d.rec->sent_first, row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                continue;
            }
            uint32_t h = dbig_hash(x) & (DBIG_LDS_SLOTS - 1);
            for (;;) {
                const unsigned long long cur = lk[h];
                if (cur == x) {
                    if (row < lr[h]) atomicMin(&lr[h], row);
                    break;
                }
                if (cur == DBIG_EMPTY) {
                    if (s_cnt >= DBIG_LDS_CAP) {   // the section's table is full: straight to the page's table
                        dbig_global_min(d.frow, dbig_global_slot(d.keys, mask, x), row);
                        break;
                    }
                    const unsigned long long old = atomicCAS(&lk[h], DBIG_EMPTY, x);
                    if (old == DBIG_EMPTY) atomicAdd(&s_cnt, 1u);
                    if (old == DBIG_EMPTY || old == x) {
                        atomicMin(&lr[h], row);
                        break;
                    }
                }
                h = (h + 1) & (DBIG_LDS_SLOTS - 1);
            }
        }
    }
    __syncthreads();
    for (uint32_t i = t; i < DBIG_LDS_SLOTS; i += WG) {
        const unsigned long long x = lk[i];
        if (x != DBIG_EMPTY) dbig_global_min(d.frow, dbig_global_slot(d.keys, mask, x), lr[i]);
    }
}

template <int KIND>
__global__ void __launch_bounds__(WG) k_dict_big_mark(EncodeArgs a, const uint32_t* big) {
    [[maybe_unused]] constexpr int W = KIND > 0 ? KIND : 8;   // bytes of a key
    if (a.use_counts && a.codec_counts[SB_CODEC_DICT] == 0) return;
    DictBigCtx d;
    if (!dbig_page_of<KIND>(a, big, &d)) return;
    const uint64_t M3 = d.rec->M3;
    const uint64_t tid = (uint64_t)blockIdx.x * WG + threadIdx.x, nth = (uint64_t)gridDim.x * WG;
    for (uint64_t i = tid; i < M3; i += nth) {
        if (d.keys[i] == DBIG_EMPTY) continue;
        const uint32_t r = d.frow[i];
        atomicOr(d.bits + (r >> 5), 1u << (r & 31));
    }
    if (tid == 0 && d.rec->sent_first != 0xFFFFFFFFu) atomicOr(d.bits + (d.rec->sent_first >> 5), 1u << (d.rec->sent_first & 31));
}

template <int KIND>
__global__ void __launch_bounds__(WG) k_dict_big_rank(EncodeArgs a, const uint32_t* big) {
    [[maybe_unused]] constexpr int W = KIND > 0 ? KIND : 8;   // bytes of a key
    __shared__ uint32_t s4[4];
    if (a.use_counts && a.codec_counts[SB_CODEC_DICT] == 0) return;
    DictBigCtx d;
    if (!dbig_page_of<KIND>(a, big, &d)) return;
    const uint64_t N = d.p.rows, SR = big_sec_rows(N);
    const uint64_t s0 = (uint64_t)blockIdx.x * SR;
    if (s0 >= N) return;
    const uint64_t w0 = s0 / 32, w1 = min(d.l.nwords, (s0 + SR) / 32);
    const int t = threadIdx.x;
    uint32_t run = 0;
    unsigned long long brun = 0;
    __shared__ unsigned long long s8[4];
    unsigned long long* wbytes = (unsigned long long*)(d.bigx + d.l.o_wbytes);
    for (uint64_t wb = w0; wb < w1; wb += WG) {
        const uint64_t w = wb + t;
        const uint32_t word = w < w1 ? d.bits[w] : 0u;
        const uint32_t c = (uint32_t)__popc(word);
        const uint32_t incl = wave_incl_scan(c);
        unsigned long long by = 0, bincl = 0;
        if constexpr (KIND < 0) {   // entry bytes of the word's first rows: u64 len | bytes (binary/dict.rs:77-90)
            for (uint32_t m = word; m; m &= m - 1) by += big_bin_weight(d.c, d.p, w * 32 + (uint32_t)__ffs((int)m) - 1);
            bincl = wave_incl_scan64(by);
        }
        __syncthreads();
        if ((t & 63) == 63) {
            s4[t >> 6] = incl;
            s8[t >> 6] = bincl;
        }
        __syncthreads();
        uint32_t base = run;
        unsigned long long bbase = brun;
        for (int q = 0; q < (t >> 6); q++) {
            base += s4[q];
            bbase += s8[q];
        }
        if (w < w1) {
            d.wpref[w] = base + incl - c;
            if constexpr (KIND < 0) wbytes[w] = bbase + bincl - by;
        }
        run += s4[0] + s4[1] + s4[2] + s4[3];
        brun += s8[0] + s8[1] + s8[2] + s8[3];
    }
    if (t == 0) {
        d.rec->sectot[blockIdx.x] = run;
        d.rec->secbytes[blockIdx.x] = brun;
    }
}

// exclusive prefix of the sections' entry counts into s_base[0 .. nsec] (all threads call; WG >= SEL_BIG_SECTIONS)
__device__ __forceinline__ void dbig_sec_bases(const DictBigRec* rec, uint32_t nsec, uint32_t* s_base, uint32_t* s4) {
    const int t = threadIdx.x;
    const uint32_t c = (uint32_t)t < nsec ? rec->sectot[t] : 0u;
    const uint32_t incl = wave_incl_scan(c);
    if ((t & 63) == 63) s4[t >> 6] = incl;
    __syncthreads();
    uint32_t base = 0;
    for (int q = 0; q < (t >> 6); q++) base += s4[q];
    s_base[t + 1] = base + incl;
    if (t == 0) s_base[0] = 0;
    __syncthreads();
}
__device__ __forceinline__ uint32_t dbig_rank(const DictBigCtx& d, const uint32_t* s_base, uint64_t SR, uint32_t row) {
    return s_base[row / SR] + d.wpref[row >> 5] + (uint32_t)__popc(d.bits[row >> 5] & ((1u << (row & 31)) - 1u));
}

template <int KIND>
__global__ void __launch_bounds__(WG) k_dict_big_ids(EncodeArgs a, const uint32_t* big) {
    [[maybe_unused]] constexpr int W = KIND > 0 ? KIND : 8;   // bytes of a key
    __shared__ uint32_t s_base[WG + 1];
    __shared__ uint32_t s4[4];
    if (a.use_counts && a.codec_counts[SB_CODEC_DICT] == 0) return;
    DictBigCtx d;
    if (!dbig_page_of<KIND>(a, big, &d)) return;
    const uint64_t N = d.p.rows, SR = big_sec_rows(N);
    const uint32_t nsec = (uint32_t)((N + SR - 1) / SR);
    dbig_sec_bases(d.rec, nsec, s_base, s4);
    __shared__ unsigned long long s_bbase[SEL_BIG_SECTIONS + 1];
    if constexpr (KIND < 0) {
        if (threadIdx.x == 0) {
            unsigned long long r = 0;
            for (uint32_t q = 0; q < nsec; q++) {
                s_bbase[q] = r;
                r += d.rec->secbytes[q];
            }
            s_bbase[nsec] = r;
        }
        __syncthreads();
    }
    const unsigned long long* wbytes = (const unsigned long long*)(d.bigx + d.l.o_wbytes);
    unsigned long long* eoff = (unsigned long long*)(d.bigx + d.l.o_eoff);
    auto entry_off = [&](uint32_t r) -> unsigned long long {   // bytes of the entries in front of the one whose first row is r
        unsigned long long o = s_bbase[r / SR] + wbytes[r >> 5];
        for (uint32_t m = d.bits[r >> 5] & ((1u << (r & 31)) - 1u); m; m &= m - 1)
            o += big_bin_weight(d.c, d.p, (uint64_t)(r & ~31u) + (uint32_t)__ffs((int)m) - 1);
        return o;
    };
    const uint64_t M3 = d.rec->M3;
    const uint64_t tid = (uint64_t)blockIdx.x * WG + threadIdx.x, nth = (uint64_t)gridDim.x * WG;
    for (uint64_t i = tid; i < M3; i += nth) {
        if (d.keys[i] == DBIG_EMPTY) continue;
        const uint32_t r = d.frow[i];
        const uint32_t id = dbig_rank(d, s_base, SR, r);
        d.firsts[id] = r;
        d.frow[i] = id;
        if constexpr (KIND < 0) eoff[id] = entry_off(r);
    }
    if (tid == 0) {
        d.rec->D = s_base[nsec];
        if constexpr (KIND < 0) d.rec->etotal = s_bbase[nsec];
        const uint32_t sf = d.rec->sent_first;
        if (sf != 0xFFFFFFFFu) {
            const uint32_t id = dbig_rank(d, s_base, SR, sf);
            d.firsts[id] = sf;
            d.rec->sent_id = id;
            if constexpr (KIND < 0) eoff[id] = entry_off(sf);
        }
    }
}

template <int KIND>
__global__ void __launch_bounds__(WG, 3) k_dict_big_idx(EncodeArgs a, const uint32_t* big) {
    [[maybe_unused]] constexpr int W = KIND > 0 ? KIND : 8;   // bytes of a key
    __shared__ unsigned long long lk[DBIG_LDS_SLOTS];
    __shared__ uint32_t li[DBIG_LDS_SLOTS];
    __shared__ uint32_t s_base[WG + 1];
    __shared__ uint32_t s4[4];
    __shared__ uint32_t s_has[2][4], s_last[2][4];
    if (a.use_counts && a.codec_counts[SB_CODEC_DICT] == 0) return;
    DictBigCtx d;
    if (!dbig_page_of<KIND>(a, big, &d)) return;
    const uint64_t N = d.p.rows, SR = big_sec_rows(N);
    const uint32_t nsec = (uint32_t)((N + SR - 1) / SR);
    const uint64_t s0 = (uint64_t)blockIdx.x * (SR / DBIG_SPLIT);
    if (s0 >= N) return;
    const uint64_t s1 = min(N, s0 + SR / DBIG_SPLIT);
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const DbigKeys<KIND> kf = dbig_keys<KIND>(a, d);
    const ValidView vv{d.c.validity, d.c.validity_bit_offset + d.p.row0};
    const uint32_t fkind = d.c.fkind;
    const uint32_t M3 = d.rec->M3, mask = M3 - 1;
    const uint32_t sent_id = d.rec->sent_id;
    dbig_sec_bases(d.rec, nsec, s_base, s4);
    const bool in_lds = M3 <= DBIG_LDS_SLOTS;   // the whole table, slot for slot
    if (in_lds) {
        for (uint32_t i = t; i < DBIG_LDS_SLOTS; i += WG) {
            lk[i] = i < M3 ? d.keys[i] : DBIG_EMPTY;
            li[i] = i < M3 ? d.frow[i] : 0u;
        }
    }
    __syncthreads();
    auto id_of_key = [&](unsigned long long x, uint64_t row) -> uint32_t {   // row is keyed, x its key
        if (dbig_is_nan<W>(x, fkind)) {   // its own entry: id = rank of the row among the first rows
            const uint32_t id = dbig_rank(d, s_base, SR, (uint32_t)row);
            d.firsts[id] = (uint32_t)row;
            return id;
        }
        if (W == 8 && x == DBIG_EMPTY) return sent_id;
        if (in_lds) {
            uint32_t h = dbig_hash(x) & mask;
            while (lk[h] != x && lk[h] != DBIG_EMPTY) h = (h + 1) & mask;
            return li[h];
        }
        return d.frow[dbig_global_find(d.keys, mask, x)];
    };
    auto id_of = [&](uint64_t row) -> uint32_t { return id_of_key(kf.key(row, vv.get(row)), row); };
    // carry-in: the index of the last keyed row before the section (row 0 is always keyed)
    uint32_t carry = 0;
    const bool fill = vv.bits != nullptr;
    if (fill && s0 > 0) {
        // (every wave looks back 64 rows at a time)
        uint64_t hi = s0;
        uint32_t found = 0xFFFFFFFFu;
        while (found == 0xFFFFFFFFu) {
            const uint64_t lo = hi >= 64 ? hi - 64 : 0;
            const uint64_t r = lo + (uint64_t)lane;
            const bool k = r < hi && (r == 0 || vv.get(r));
            const uint64_t m = __ballot(k);
            if (m) found = (uint32_t)(lo + (uint64_t)top_bit(m));
            hi = lo;
        }
        carry = id_of(found);
    }
    uint32_t par = 0;
    // no validity bitmap: every row is keyed and nothing is carried — the keys of eight rows per thread in flight, then their ids
    // (one load per thread and iteration is a latency chain: 64 iterations of ~1.5 us for a 16 384-row part, 0.09 ms of a
    // 12 M-row page's 0.54)
    if (!fill) {
        for (uint64_t base = s0; base < s1; base += (uint64_t)WG * 8) {
            unsigned long long xu[8];
            uint32_t idv[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const uint64_t i = base + (uint64_t)u * WG + t;
                xu[u] = kf.key(i < s1 ? i : s0, true);
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const uint64_t i = base + (uint64_t)u * WG + t;
                idv[u] = i < s1 ? id_of_key(xu[u], i) : 0u;
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const uint64_t i = base + (uint64_t)u * WG + t;
                if (i < s1) d.idx[i] = idv[u];
            }
        }
    }
    for (uint64_t base = s0; fill && base < s1; base += (uint64_t)WG * 8) {
#pragma unroll 1
        for (int u = 0; u < 8; u++, par ^= 1) {
            const uint64_t g0 = base + (uint64_t)u * WG;   // this group's first row (uniform)
            if (g0 >= s1) break;
            const uint64_t i = g0 + t;
            const bool in = i < s1;
            const bool keyed = in && (!fill || i == 0 || vv.get(i));
            uint32_t id = keyed ? id_of(i) : 0u;
            if (fill) {
                const uint64_t km = __ballot(keyed);
                const uint64_t below = km & ((2ull << lane) - 1);   // keyed lanes up to and including mine
                const uint32_t src = __shfl(id, below ? top_bit(below) : 0, 64);
                const uint32_t lastw = __shfl(id, km ? top_bit(km) : 0, 64);
                if (lane == 0) {
                    s_has[par][w] = km != 0;
                    s_last[par][w] = lastw;
                }
                __syncthreads();
                uint32_t cin = carry;
                for (int q = 0; q < 4; q++) {
                    if (q < w && s_has[par][q]) cin = s_last[par][q];
                    if (s_has[par][q]) carry = s_last[par][q];
                }
                id = below ? src : cin;
            }
            if (in) d.idx[i] = id;
        }
    }
    if (blockIdx.x == 0 && t == 0) {   // the index array as a page of its own (compress_integer::<u32>, dict.rs:57-62)
        EncCol vc = d.c;
        vc.values = (const uint8_t*)d.idx;
        vc.validity = nullptr;
        vc.offsets = nullptr;
        vc.heads = nullptr;
        vc.out = nullptr;
        vc.values_bit_offset = 0;
        vc.validity_bit_offset = 0;
        vc.out_cap = 0;
        vc.rows = N;
        vc.nullable = 0;
        vc.width = 4;
        vc.ptype = SB_TYPE_UINT32;
        vc.fkind = 0;
        vc.nk = NK_UNSIGNED;
        vc.first_page = a.n_pages + d.page;
        vc.n_pages = 1;
        ((EncCol*)a.vcols)[d.page] = vc;
        EncPage vp;
        __builtin_memset(&vp, 0, sizeof vp);
        vp.rows = N;
        vp.slot_off = (uint64_t)(d.nblk - a.scratch);
        vp.slot_cap = dbig_nblk_cap(N);
        vp.seed = d.p.seed;
        vp.col = a.n_cols + d.page;
        vp.codec = CODEC_ON_DEVICE;
        vp.icodec = -1;
        vp.depth = d.p.depth + 1;
        vp.forb_extra = d.p.forb_extra | (1u << SB_CODEC_DICT);
        vp.h64_off = ~0ull;
        vp.zst_off = ~0ull;
        ((EncPage*)a.vpages)[d.page] = vp;
        int32_t ic = d.p.icodec >= 0 ? d.p.icodec : (a.has_ratio ? CODEC_PENDING : (int32_t)a.default_compression);
        a.codecs[a.n_pages + d.page] = ic;
        if (ic >= 0) atomicAdd(&a.codec_counts[ic & 31], 1u);
        EncOut vo;
        __builtin_memset(&vo, 0, sizeof vo);
        vo.pad = VPAD_ACTIVE;
        a.outs[a.n_pages + d.page] = vo;
    }
}

// ---------------------------------------------------------------------------------------------------- nested writers
// The virtual page of a list entry when it is on the section-parallel path (EncOut.pad says so: the codec words are zeroed
// per call, and 0 is a codec), its codec is one of `cmask` and it is in state `want_pad`.
__device__ __forceinline__ bool vbig_page_of(const EncodeArgs& a, const uint32_t* big, uint32_t cmask, uint32_t* page, EncPage* p, EncCol* c,
                                             int32_t* codec, uint32_t want_pad = VPAD_ACTIVE) {
    *page = big[blockIdx.y] + a.n_pages;
    const EncOut vo = a.outs[*page];
    if (vo.pad != want_pad || (want_pad == VPAD_ACTIVE && vo.length != 0)) return false;
    *codec = a.codecs[*page];
    if (*codec < 0 || *codec > 31 || !((cmask >> *codec) & 1)) return false;
    *p = get_page(a, *page);
    *c = get_col(a, p->col);
    return p->rows >= VBIG_ROWS;
}
__device__ __forceinline__ bool vbig_any(const EncodeArgs& a, uint32_t cmask) {   // (adaptive waves: nobody chose one of these codecs)
    if (!a.use_counts) return true;
    for (uint32_t k = 0; k < 31; k++)
        if (((cmask >> k) & 1) && a.codec_counts[k]) return true;
    return false;
}

// Bitpacking / DeltaBitpacking of a long u32 page (integer/bp.rs:48-62, delta_bp.rs:49-65), one workgroup per tile of
// TILE_ROWS rows = 32 blocks: pass 0 = bytes of every tile (1 + 16 * num_bits per block, num_bits from the RAW values),
// a scan over the tiles, pass 1 = the packed blocks at their offsets.  enc_bp's tile body.
template <bool STORE>
__device__ uint32_t bp_big_tile(const uint32_t* vals, uint64_t cb, uint32_t n, bool delta, uint8_t* dst, uint32_t* sA, uint32_t* sB) {
    const int t = threadIdx.x;
    constexpr int NB = TILE_ROWS / 128, K = TILE_ROWS / WG;
    __shared__ uint32_t s_nb[NB], s_off[NB + 1], s_woff[NB + 1];
    const uint32_t nblk = n / 128;
#pragma unroll
    for (int k = 0; k < K; k++) {
        const uint32_t i = (uint32_t)t + (uint32_t)k * WG;
        sA[sidx((int)i)] = i < n ? vals[cb + i] : 0u;
    }
    const uint32_t carry_prev = cb ? vals[cb - 1] : 0u;
    __syncthreads();
    const uint32_t* src = sA;
    if (STORE && delta) {
        for (uint32_t i = t; i < n; i += WG) sB[sidx((int)i)] = sA[sidx((int)i)] - (i ? sA[sidx((int)i - 1)] : carry_prev);
        src = sB;
    }
    {
        const uint32_t blk = t >> 3, sub = t & 7;
        uint32_t acc = 0;
        if (blk < nblk)
            for (uint32_t k = 0; k < 16; k++) acc |= sA[sidx((int)(blk * 128 + sub * 16 + k))];
        acc |= __shfl_xor(acc, 1, 64);
        acc |= __shfl_xor(acc, 2, 64);
        acc |= __shfl_xor(acc, 4, 64);
        if (sub == 0 && blk < nblk) s_nb[blk] = acc ? 32 - __clz(acc) : 0;
    }
    __syncthreads();
    if (t < 64) {
        const uint32_t nb = (uint32_t)t < nblk ? s_nb[t] : 0u;
        const uint32_t by = (uint32_t)t < nblk ? 1 + 16 * nb : 0u;
        const uint32_t ib = wave_incl_scan(by), iw = wave_incl_scan(4 * nb);
        if ((uint32_t)t < nblk) {
            s_off[t] = ib - by;
            s_woff[t] = iw - 4 * nb;
        }
        if ((uint32_t)t == nblk - 1) {
            s_off[nblk] = ib;
            s_woff[nblk] = iw;
        }
    }
    __syncthreads();
    const uint32_t bytes = s_off[nblk];
    if (!STORE) return bytes;
    const uint32_t total_words = s_woff[nblk];
    for (uint32_t j = t; j < total_words; j += WG) {
        uint32_t lo = 0, hi = nblk;
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (s_woff[mid] <= j) lo = mid;
            else hi = mid;
        }
        const uint32_t blk = lo, wi = j - s_woff[blk], nb = s_nb[blk];
        const uint32_t l = wi & 3, k = wi >> 2;
        const uint32_t lo_bit = 32 * k, hi_bit = 32 * k + 32;
        uint32_t word = 0;
        const uint32_t i0 = lo_bit / nb, i1 = min(31u, (hi_bit - 1) / nb);
        for (uint32_t i = i0; i <= i1; i++) {
            const uint32_t v = src[sidx((int)(blk * 128 + 4 * i + l))];
            const uint32_t bitpos = i * nb;
            if (bitpos >= lo_bit) word |= v << (bitpos - lo_bit);
            else if (bitpos + nb > lo_bit) word |= v >> (lo_bit - bitpos);
        }
        stu32(dst + s_off[blk] + 1 + 4 * wi, word);
    }
    if ((uint32_t)t < nblk) dst[s_off[t]] = (uint8_t)s_nb[t];
    return bytes;
}
// tile byte counts live behind the block (the slot's tail: 4 bytes per tile, far behind 4 N + 9)
__device__ __forceinline__ uint32_t* bp_big_tilebytes(uint8_t* slot, const EncPage& p) {
    const uint64_t ntiles = (p.rows + TILE_ROWS - 1) / TILE_ROWS;
    return (uint32_t*)(slot + ((p.slot_cap - 4 * (ntiles + 2)) & ~15ull));
}
template <int PASS>
__global__ void __launch_bounds__(WG) k_bp_big(EncodeArgs a, const uint32_t* big) {
    __shared__ uint32_t sA[SIDX_WORDS], sB[SIDX_WORDS];
    constexpr uint32_t CM = (1u << SB_CODEC_BITPACKING) | (1u << SB_CODEC_DELTA_BITPACKING);
    if (!vbig_any(a, CM)) return;
    uint32_t page;
    EncPage p;
    EncCol c;
    int32_t codec;
    if (!vbig_page_of(a, big, CM, &page, &p, &c, &codec, PASS == 2 ? VPAD_PLANNED : VPAD_ACTIVE) || p.rows % 128 != 0 || c.width != 4) return;
    const uint64_t N = p.rows, ntiles = (N + TILE_ROWS - 1) / TILE_ROWS;
    uint8_t* slot = page_slot(a, c, p);
    uint32_t* tb = bp_big_tilebytes(slot, p);
    const uint32_t* vals = (const uint32_t*)c.values + p.row0;
    const bool delta = codec == SB_CODEC_DELTA_BITPACKING;
    if (PASS == 1) {   // one workgroup per page: exclusive prefix of the tile bytes (u64 total in the two words behind)
        uint64_t run = 0;
        __shared__ uint32_t s4[4];
        for (uint64_t b = 0; b < ntiles; b += WG) {
            const uint64_t i = b + threadIdx.x;
            const uint32_t v = i < ntiles ? tb[i] : 0u;
            const uint32_t incl = wave_incl_scan(v);
            __syncthreads();
            if ((threadIdx.x & 63) == 63) s4[threadIdx.x >> 6] = incl;
            __syncthreads();
            uint32_t base = 0;
            for (int q = 0; q < (int)(threadIdx.x >> 6); q++) base += s4[q];
            if (i < ntiles) tb[i] = (uint32_t)(run + base + incl - v);   // (body < 2^32: checked below)
            run += (uint64_t)s4[0] + s4[1] + s4[2] + s4[3];
        }
        if (threadIdx.x == 0) {
            if (run > 0xFFFFFFF0ull) raise(a.status, SB_ERR_INVALID, page, 560);
            put_hdr9(slot, (uint32_t)codec, (uint32_t)run, (uint32_t)(N * 4));
            EncOut o;
            o.length = 9 + run;
            o.out_off = 0;
            o.slot = slot;
            o.codec = (uint32_t)codec;
            o.pad = VPAD_PLANNED;
            a.outs[page] = o;
        }
        return;
    }
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint64_t cb = tile * TILE_ROWS;
        const uint32_t n = (uint32_t)min((uint64_t)TILE_ROWS, N - cb);
        if (PASS == 0) {
            const uint32_t bytes = bp_big_tile<false>(vals, cb, n, delta, nullptr, sA, sB);
            if (threadIdx.x == 0) tb[tile] = bytes;
        } else {
            bp_big_tile<true>(vals, cb, n, delta, slot + 9 + tb[tile], sA, sB);
        }
        __syncthreads();
    }
}

// None, and Basic(LZ4) with the free parse = ONE literal run (enc_u32_block), of a long u32 page: header + copy
__global__ void __launch_bounds__(WG) k_plain_big(EncodeArgs a, const uint32_t* big) {
    const uint32_t CM = (1u << SB_CODEC_NONE) | ((a.flags & SB_WRITE_LZ4_EXACT) ? 0u : (1u << SB_CODEC_LZ4));
    if (!vbig_any(a, CM)) return;
    uint32_t page;
    EncPage p;
    EncCol c;
    int32_t codec;
    if (!vbig_page_of(a, big, CM, &page, &p, &c, &codec)) return;
    // (the one-literal-run LZ4 block is what enc_u32_block writes for the INDEX array of a Dict page; the exceptions block of
    // a Freq page goes through the matcher: enc_nested_block)
    if (codec == SB_CODEC_LZ4 && a.codecs[page - a.n_pages] != (int32_t)SB_CODEC_DICT) return;
    uint8_t* slot = page_slot(a, c, p);
    const uint64_t nbytes = p.rows * c.width;
    uint32_t hdr = 0;
    if (codec == SB_CODEC_LZ4) hdr = 1 + (nbytes >= 15 ? 1 + (uint32_t)((nbytes - 15) / 255) : 0);
    const uint64_t tid = (uint64_t)blockIdx.x * WG + threadIdx.x, nth = (uint64_t)gridDim.x * WG;
    if (codec == SB_CODEC_LZ4) {
        uint8_t* o = slot + 9;
        if (tid == 0) o[0] = (uint8_t)((nbytes >= 15 ? 15u : (uint32_t)nbytes) << 4);
        if (nbytes >= 15) {
            const uint64_t nff = (nbytes - 15) / 255;
            for (uint64_t k = tid; k < nff; k += nth) o[1 + k] = 255;
            if (tid == 0) o[1 + nff] = (uint8_t)((nbytes - 15) % 255);
        }
    }
    const uint8_t* src = c.values + p.row0 * c.width;
    uint8_t* dst = slot + 9 + hdr;
    // (dst is not 16-byte aligned: the copy is by dwords, source aligned)
    const uint64_t nd = nbytes / 4;
    for (uint64_t i = tid; i < nd; i += nth) stu32(dst + 4 * i, ldu32(src + 4 * i));
    if (tid == 0) {
        for (uint64_t i = nd * 4; i < nbytes; i++) dst[i] = src[i];
        put_hdr9(slot, (uint32_t)codec, (uint32_t)(hdr + nbytes), (uint32_t)nbytes);
        EncOut o;
        o.length = 9 + hdr + nbytes;
        o.out_off = 0;
        o.slot = slot;
        o.codec = (uint32_t)codec;
        o.pad = 1;
        a.outs[page] = o;
    }
}

// ---------------------------------------------------------------------------------------------------- finish
template <int KIND>
__global__ void __launch_bounds__(WG, 2) k_dict_big_finish(EncodeArgs a, const uint32_t* big) {
    [[maybe_unused]] constexpr int W = KIND > 0 ? KIND : 8;   // bytes of a key
    __shared__ __attribute__((aligned(16))) uint32_t lds[3 * SIDX_WORDS];
    __shared__ uint32_t s_w[4];
    if (a.use_counts && a.codec_counts[SB_CODEC_DICT] == 0) return;
    DictBigCtx d;
    if (!dbig_page_of<KIND>(a, big, &d)) return;
    if (KIND < 0 && d.rec->bad) return;   // two different strings with one hash: k_enc_emit_pages builds the page exactly
    uint32_t *sA = lds, *sB = lds + SIDX_WORDS, *sC = lds + 2 * SIDX_WORDS;
    const uint64_t N = d.p.rows;
    uint8_t* slot = page_slot(a, d.c, d.p);
    uint64_t pos = 0;
    if (d.c.nullable) {
        uint8_t* bits = def_header(slot, N);
        def_bits_page(bits, ValidView{d.c.validity, d.c.validity_bit_offset}, d.p.row0, N, d.c.rows);
        pos = def_section_bytes(N);
    }
    uint8_t* blk = slot + pos;
    const uint32_t vpage = a.n_pages + d.page;
    const int32_t ic = a.codecs[vpage];
    const uint32_t D = d.rec->D;
    const EncOut vout = a.outs[vpage];
    const bool par = vout.length != 0 && (vout.pad == 1 || vout.pad == VPAD_PLANNED);
    uint64_t ib;
    if (par) {   // written by the section- / tile-parallel kernels into the nested slot
        ib = vout.length;
        if (threadIdx.x == 0) d.rec->nested_ok = 1;
    } else if (ic == SB_CODEC_FREQ) {
        // mostly one index: the Freq kernels finish the page (see emit_prim_page<Dict>)
        if (d.p.vaux_bytes < 32 || !d.p.vslot_off) {
            if (threadIdx.x == 0) raise(a.status, SB_ERR_NYI, d.page, 502);
            return;
        }
        if (threadIdx.x == 0) {
            unsigned long long* rec = (unsigned long long*)(a.scratch + d.p.vaux_off);
            rec[0] = (unsigned long long)(uintptr_t)d.idx;
            rec[1] = (unsigned long long)(uintptr_t)d.firsts;
            rec[2] = D;
            atomicAdd(a.freq_count, 1u);
            EncOut o;
            o.length = 1;
            o.out_off = 0;
            o.slot = slot;
            o.codec = SB_CODEC_DICT;
            o.pad = 3;
            a.outs[d.page] = o;
        }
        return;
    } else {
        ib = enc_u32_block(d.idx, N, ic, blk + 9, sA, sB, sC, s_w, a.status, d.page, a.flags, nullptr, 0);
        if (ib == 0) return;
    }
    if (threadIdx.x == 0) {
        stu32(blk + 9 + ib, D);
        // entries: n x value / n x {u64 len | bytes}; a binary block's uncompressed_size = array.values().len() (binary/mod.rs:88)
        const uint64_t ebytes = KIND > 0 ? (uint64_t)D * W : d.rec->etotal;
        put_hdr9(blk, SB_CODEC_DICT, (uint32_t)(ib + 4 + ebytes), KIND > 0 ? (uint32_t)(N * W) : (uint32_t)d.c.values_len_total);
        EncOut o;
        o.length = pos + 9 + ib + 4 + ebytes;
        o.out_off = 0;
        o.slot = slot;
        o.codec = SB_CODEC_DICT;
        o.pad = 1;   // emitted here: k_enc_emit_pages<., Dict> leaves the page alone
        a.outs[d.page] = o;
    }
}

template <int KIND>
__global__ void __launch_bounds__(WG) k_dict_big_values(EncodeArgs a, const uint32_t* big) {
    [[maybe_unused]] constexpr int W = KIND > 0 ? KIND : 8;   // bytes of a key
    if (a.use_counts && a.codec_counts[SB_CODEC_DICT] == 0) return;
    const uint32_t page = big[blockIdx.y];
    const EncPage p = a.pages[page];
    const EncCol c = a.cols[p.col];
    // (after k_dict_big_finish: the page record exists, pad == 1 marks this path)
    if (!p.bigx_off || a.codecs[page] != (int32_t)SB_CODEC_DICT || a.outs[page].pad != 1 || a.outs[page].length == 0) return;
    if constexpr (KIND > 0) {
        if (big_is_bin(c) || (int)c.width != KIND) return;
    } else {
        if (c.ptype != (KIND == -4 ? SB_TYPE_BINARY : SB_TYPE_LARGE_BINARY)) return;
    }
    uint8_t* bigx = a.scratch + p.bigx_off;
    const DictBigRec* rec = (const DictBigRec*)bigx;
    if (!rec->active) return;
    const DictBigLayout l = dbig_layout(p.rows);
    const uint32_t* firsts = (const uint32_t*)(bigx + l.o_firsts);
    const uint8_t* nblk = bigx + l.o_nblk;
    uint8_t* slot = page_slot(a, c, p);
    const uint64_t pos = c.nullable ? def_section_bytes(p.rows) : 0;
    uint8_t* blk = slot + pos;
    const uint64_t tid = (uint64_t)blockIdx.x * WG + threadIdx.x, nth = (uint64_t)gridDim.x * WG;
    uint64_t ib;
    if (rec->nested_ok) {   // the nested block: from its own slot to its place behind the header
        ib = 9 + (uint64_t)ldu32(nblk + 1);
        uint8_t* dst = blk + 9;
        const uint64_t head = min(ib, (uint64_t)((16 - ((uintptr_t)dst & 15)) & 15));
        for (uint64_t i = tid; i < head; i += nth) dst[i] = nblk[i];
        const uint64_t nv = (ib - head) / 16;
        for (uint64_t i = tid; i < nv; i += nth) stu128(dst + head + 16 * i, ldu128(nblk + head + 16 * i));
        for (uint64_t i = head + 16 * nv + tid; i < ib; i += nth) dst[i] = nblk[i];
    } else {
        ib = 9 + (uint64_t)ldu32(blk + 9 + 1);
    }
    const uint32_t D = rec->D;
    uint8_t* q = blk + 9 + ib + 4;
    if constexpr (KIND > 0) {
        const uint8_t* vals = c.values + p.row0 * W;
        const bool lead_null = c.validity && !bit_at(c.validity, c.validity_bit_offset + p.row0);
        for (uint64_t k = tid; k < D; k += nth) {
            const uint32_t r = firsts[k];
            Val<W> v = ld_val<W>(vals + (uint64_t)r * W);
            if (r == 0 && lead_null) v = val_zero<W>();
            __builtin_memcpy(q + k * W, &v, W);
        }
    } else {   // u64 len | bytes per entry, at the offsets k_dict_big_ids found
        using O = typename std::conditional<KIND == -4, int32_t, int64_t>::type;
        const BinKeys<O> bk{c.offsets + p.row0 * sizeof(O), c.values, ValidView{nullptr, 0}};
        const unsigned long long* eoff = (const unsigned long long*)(bigx + l.o_eoff);
        for (uint64_t k = tid; k < D; k += nth) {
            const uint64_t r = firsts[k];
            const uint64_t b = bk.beg(r), e = bk.beg(r + 1);
            uint8_t* dd = q + eoff[k];
            stu64(dd, e - b);
            uint64_t j = 0;
            for (; j + 16 <= e - b; j += 16) stu128(dd + 8 + j, ldu128(c.values + b + j));   // (unaligned 16-byte moves)
            for (; j + 8 <= e - b; j += 8) stu64(dd + 8 + j, ldu64(c.values + b + j));
            for (; j < e - b; j++) *(gptr)(dd + 8 + j) = ldu8(c.values + b + j);
        }
    }
}

// binary pages: the dictionary was formed on 64-bit hashes — every keyed row's string against the string of its entry's
// first row (the index array names the entry).  A mismatch (two strings with one hash: never seen) sets the page's `bad`
// word: k_dict_big_finish leaves the page to k_enc_emit_pages, whose builder compares strings.
template <int KIND>
__global__ void __launch_bounds__(WG) k_dict_big_verify(EncodeArgs a, const uint32_t* big) {
    static_assert(KIND < 0, "binary pages only");
    if (a.use_counts && a.codec_counts[SB_CODEC_DICT] == 0) return;
    DictBigCtx d;
    if (!dbig_page_of<KIND>(a, big, &d)) return;
    using O = typename std::conditional<KIND == -4, int32_t, int64_t>::type;
    const uint64_t N = d.p.rows, SR = big_sec_rows(N) / DBIG_SPLIT;
    const uint64_t s0 = (uint64_t)blockIdx.x * SR;
    if (s0 >= N) return;
    const uint64_t s1 = min(N, s0 + SR);
    const ValidView vv{d.c.validity, d.c.validity_bit_offset + d.p.row0};
    const BinKeysHashed<O> kh{BinKeys<O>{d.c.offsets + d.p.row0 * sizeof(O), d.c.values, vv}, (const uint64_t*)(a.scratch + d.p.h64_off), d.c.values_len};
    bool ok = !(a.flags & SB_WRITE_DEBUG_VERIFY_FAIL_BIT);   // (tests: every page fails)
    constexpr int VU = 8;
    for (uint64_t base = s0 + threadIdx.x; base < s1 && ok; base += (uint64_t)WG * VU) {
        uint32_t f[VU];
        uint64_t row[VU];
#pragma unroll
        for (int u = 0; u < VU; u++) {
            row[u] = base + (uint64_t)u * WG;
            const bool keyed = row[u] < s1 && (row[u] == 0 || vv.get(row[u]));
            f[u] = keyed ? d.firsts[d.idx[row[u]]] : EMPTY;
            if (f[u] == (uint32_t)row[u]) f[u] = EMPTY;   // a first row needs no check
        }
        if (!kh.template exact_batch<VU>(f, row)) ok = false;
    }
    if (!ok) d.rec->bad = 1;
}
