// strawboat-hip: Freq pages of LONG pages prepared container-parallel (included by sb_encode.hip inside namespace sb, after
// the Freq page functions).
//
// integer/freq.rs:33-100: `top value | u32 rb_size | Roaring bitmap of the exception rows | BLOCK<T exceptions>`.  A sparse
// column written with the reference's default paging is ONE page of millions of rows; freq_prep_page walks it with one
// workgroup (vote, count, one container after the other: 80-150 ms for 12 M rows).  For pages of SEL_BIG_ROWS rows or more
// that the long-page selector gave to Freq (its record still holds the majority key and the null count):
//
//   k_freq_big_count  (containers x pages)   exceptions per 64 Ki-row Roaring container, first row of the top value
//   k_freq_big_plan   (1 workgroup / page)   Roaring header (non-empty containers, cardinalities, offsets), the top value,
//                                            def levels, where every container's body and exceptions go; the exceptions as
//                                            a virtual page — on the section-parallel path (k_sel_big_*, k_rle_big_*,
//                                            k_bp_big, k_plain_big with voff) when there are VBIG_ROWS of them or more
//   k_freq_big_emit   (containers x pages)   container bodies (sorted u16 arrays / 8 KiB bitmaps), exception values
//
// k_enc_freq_prep skips the pages prepared here; k_enc_nested writes what the parallel writers left (a Dict / OneValue /
// Zstd ... exceptions block); k_enc_freq_finish joins the blocks as before.
struct FreqBigRec {
    uint32_t active, n_ex, rb_size, ncne, top_is_null, pad[3];
    unsigned long long topk;
    uint32_t card[FREQ_MAX_CONTAINERS];
    uint32_t cmin[FREQ_MAX_CONTAINERS];      // first row of the top value inside the container (~0u: none)
    uint32_t part[FREQ_MAX_CONTAINERS * 4];  // k_freq_big_count: exceptions of a quarter container
    uint32_t pmin[FREQ_MAX_CONTAINERS * 4];
    uint32_t data_off[FREQ_MAX_CONTAINERS];  // container body, from the start of the Roaring bytes
    uint32_t ex_base[FREQ_MAX_CONTAINERS];
};
static_assert(sizeof(FreqBigRec) <= BIGX_FREQ_BYTES, "FreqBigRec fits its area");

template <int W>
__device__ __forceinline__ bool fbig_page_of(const EncodeArgs& a, const uint32_t* big, uint32_t* page, EncPage* p, EncCol* c, FreqBigRec** rec,
                                             bool prepared) {
    *page = big[blockIdx.y];
    if (a.codecs[*page] != (int32_t)SB_CODEC_FREQ) return false;
    *p = a.pages[*page];
    if (p->codec != CODEC_ON_DEVICE || !p->bigx_off || p->rows < SEL_BIG_ROWS || !p->vslot_off) return false;
    *c = a.cols[p->col];
    if ((int)c->width != W || (p->rows + 65535) / 65536 > FREQ_MAX_CONTAINERS) return false;
    *rec = (FreqBigRec*)(a.scratch + p->bigx_off + BIGX_FREQ_OFF);
    const EncOut o = a.outs[*page];
    return prepared ? (o.length != 0 && o.codec == SB_CODEC_FREQ && o.pad == 4) : o.length == 0;
}

template <int W>
__global__ void __launch_bounds__(WG) k_freq_big_count(EncodeArgs a, const uint32_t* big) {
    __shared__ uint32_t s4[4];
    __shared__ uint32_t s_min;
    if (*a.freq_count == 0) return;
    uint32_t page;
    EncPage p;
    EncCol c;
    FreqBigRec* rec;
    if (!fbig_page_of<W>(a, big, &page, &p, &c, &rec, false)) return;
    const uint64_t N = p.rows;
    const uint64_t b = (uint64_t)blockIdx.x * 16384;   // a quarter of a container per workgroup
    if (b >= N) return;
    const uint64_t e = min(N, b + 16384);
    const BigPage bp = *big_page_rec(page_slot(a, c, p));
    const bool top_is_null = (double)bp.nulls / (double)N >= 0.9;
    const uint8_t* vals = c.values + p.row0 * W;
    const ValidView vv{c.validity, c.validity_bit_offset + p.row0};
    const uint32_t nk = c.nk;
    const int t = threadIdx.x;
    if (t == 0) s_min = 0xFFFFFFFFu;
    __syncthreads();
    uint32_t cnt = 0, first = 0xFFFFFFFFu;
    for (uint64_t i0 = b + t; i0 < e; i0 += (uint64_t)WG * 8) {
        Val<W> v[8];
        bool ok[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint64_t i = i0 + (uint64_t)u * WG;
            v[u] = ld_val<W>(vals + (i < e ? i : b) * W);
            ok[u] = i < e && vv.get(i);
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint64_t i = i0 + (uint64_t)u * WG;
            const Val<W> k = stat_key<W>(v[u], nk);
            uint64_t x = 0;
            __builtin_memcpy(&x, &k, W);
            const bool is_top = i < e && !top_is_null && x == bp.maj_k;
            if (is_top && first == 0xFFFFFFFFu) first = (uint32_t)i;
            if (ok[u] && !is_top) cnt++;
        }
    }
    const uint32_t tot = wg_sum32(cnt, s4);
    if (first != 0xFFFFFFFFu) atomicMin(&s_min, first);
    __syncthreads();
    if (t == 0) {
        rec->part[blockIdx.x] = tot;
        rec->pmin[blockIdx.x] = s_min;
    }
}

template <int W>
__global__ void __launch_bounds__(WG) k_freq_big_plan(EncodeArgs a, const uint32_t* big, EncCol* cols_rw, EncPage* pages_rw) {
    __shared__ uint32_t s_card[FREQ_MAX_CONTAINERS], s_cmin[FREQ_MAX_CONTAINERS];
    if (*a.freq_count == 0) return;
    uint32_t page;
    EncPage p;
    EncCol c;
    FreqBigRec* rec;
    if (!fbig_page_of<W>(a, big, &page, &p, &c, &rec, false)) return;
    const uint64_t N = p.rows;
    const uint32_t nc_all = (uint32_t)((N + 65535) / 65536);
    const int t = threadIdx.x;
    const BigPage bp = *big_page_rec(page_slot(a, c, p));   // (read before the def levels overwrite the slot's head)
    const bool top_is_null = (double)bp.nulls / (double)N >= 0.9;
    for (uint32_t q = t; q < nc_all; q += WG) {
        uint32_t cs = 0, mn = 0xFFFFFFFFu;
        for (uint32_t u = 0; u < 4; u++)
            if (((uint64_t)q * 4 + u) * 16384 < N) {
                cs += rec->part[q * 4 + u];
                mn = min(mn, rec->pmin[q * 4 + u]);
            }
        s_card[q] = cs;
        s_cmin[q] = mn;
        rec->card[q] = cs;
        rec->cmin[q] = mn;
    }
    __syncthreads();
    uint8_t* slot = page_slot(a, c, p);
    uint64_t pos = 0;
    if (c.nullable) {
        uint8_t* bits = def_header(slot, N);
        def_bits_page(bits, ValidView{c.validity, c.validity_bit_offset}, p.row0, N, c.rows);
        pos = def_section_bytes(N);
    }
    uint8_t* blk = slot + pos;
    uint8_t* rb = blk + 9 + W + 4;
    if (t == 0) {
        uint32_t ncne = 0, n_ex = 0, top_row = 0xFFFFFFFFu;
        for (uint32_t cq = 0; cq < nc_all; cq++) {
            ncne += s_card[cq] ? 1u : 0u;
            n_ex += s_card[cq];
            if (top_row == 0xFFFFFFFFu) top_row = s_cmin[cq];
        }
        uint32_t rb_size = 8 + 8 * ncne;
        stu32(rb, 12346u);
        stu32(rb + 4, ncne);
        uint32_t k = 0, off = 8 + 8 * ncne, exb = 0;
        for (uint32_t cq = 0; cq < nc_all; cq++) {
            const uint32_t card = s_card[cq];
            rec->data_off[cq] = off;
            rec->ex_base[cq] = exb;
            if (!card) continue;
            *(gptr)(rb + 8 + 4 * k) = (uint8_t)cq;
            *(gptr)(rb + 8 + 4 * k + 1) = (uint8_t)(cq >> 8);
            *(gptr)(rb + 8 + 4 * k + 2) = (uint8_t)(card - 1);
            *(gptr)(rb + 8 + 4 * k + 3) = (uint8_t)((card - 1) >> 8);
            stu32(rb + 8 + 4 * ncne + 4 * k, off);
            const uint32_t body = card > 4096 ? 8192u : 2 * card;
            off += body;
            rb_size += body;
            exb += card;
            k++;
        }
        stu32(blk + 9 + W, rb_size);
        // the top value: the first slot's raw bits (null slots count), T::default() when >= 90 % of the rows are null
        Val<W> top = val_zero<W>();
        bool ok = true;
        if (!top_is_null) {
            if (top_row == 0xFFFFFFFFu) ok = false;   // (the selector's majority key does not occur: cannot happen)
            else top = ld_val<W>(c.values + (p.row0 + top_row) * W);
        }
        st_val<W>(blk + 9, top);
        rec->n_ex = n_ex;
        rec->rb_size = rb_size;
        rec->ncne = ncne;
        rec->top_is_null = top_is_null ? 1u : 0u;
        rec->topk = bp.maj_k;
        rec->active = ok ? 1u : 0u;
        if (!ok) {
            raise(a.status, SB_ERR_NYI, page, 548);
            return;
        }
        // ---- the virtual page that carries the exceptions (as freq_prep_page)
        EncCol vc = c;
        vc.values = a.scratch + p.ex_off;
        vc.validity = nullptr;
        vc.offsets = nullptr;
        vc.heads = nullptr;
        vc.out = nullptr;
        vc.values_bit_offset = 0;
        vc.validity_bit_offset = 0;
        vc.out_cap = 0;
        vc.rows = n_ex;
        vc.nullable = 0;
        vc.first_page = a.n_pages + page;
        vc.n_pages = 1;
        cols_rw[page] = vc;
        EncPage vp;
        __builtin_memset(&vp, 0, sizeof vp);
        vp.rows = n_ex;
        vp.slot_off = p.vslot_off;
        vp.slot_cap = 64 + 18 + N * (uint64_t)(W + 8) + 4 + 64;   // slot_fixed_bytes(type, not nullable, N)
        vp.aux_off = p.vaux_off;
        vp.aux_bytes = p.vaux_bytes;
        vp.seed = p.seed;
        vp.col = a.n_cols + page;
        vp.codec = a.nested_force >= 0 && !(((a.forbidden | (1u << SB_CODEC_FREQ)) >> a.nested_force) & 1)
                       ? a.nested_force
                       : (a.has_ratio ? CODEC_ON_DEVICE : (int32_t)a.default_compression);
        vp.icodec = -1;
        vp.depth = p.depth + 1;
        vp.forb_extra = p.forb_extra | (1u << SB_CODEC_FREQ);
        vp.h64_off = ~0ull;
        vp.zst_off = ~0ull;
        pages_rw[page] = vp;
        EncOut o;
        o.length = pos + 9 + W + 4 + rb_size;   // so far; k_enc_freq_finish adds the nested block
        o.out_off = 0;
        o.slot = slot;
        o.codec = SB_CODEC_FREQ;
        o.pad = 4;   // prepared here (k_enc_freq_prep leaves the page alone); k_freq_big_emit resets it to 0
        a.outs[page] = o;
        if (n_ex >= VBIG_ROWS) {   // the exceptions block on the section-parallel path
            const int32_t ic = vp.codec == CODEC_ON_DEVICE ? CODEC_PENDING : vp.codec;
            a.codecs[a.n_pages + page] = ic;
            if (ic >= 0) atomicAdd(&a.codec_counts[ic & 31], 1u);
            EncOut vo;
            __builtin_memset(&vo, 0, sizeof vo);
            vo.pad = VPAD_ACTIVE;
            a.outs[a.n_pages + page] = vo;
        }
    }
}

template <int W>
__global__ void __launch_bounds__(WG) k_freq_big_emit(EncodeArgs a, const uint32_t* big) {
    __shared__ uint32_t sA[SIDX_WORDS];
    __shared__ uint32_t s_w[4];
    if (*a.freq_count == 0) return;
    uint32_t page;
    EncPage p;
    EncCol c;
    FreqBigRec* rec;
    if (!fbig_page_of<W>(a, big, &page, &p, &c, &rec, true)) return;
    const uint64_t N = p.rows;
    const uint32_t cq = blockIdx.x;
    const uint64_t b = (uint64_t)cq * 65536;
    if (b >= N) return;
    const uint64_t e = min(N, b + 65536);
    const uint32_t card = rec->card[cq];
    if (!card) return;
    const int t = threadIdx.x, lane = t & 63;
    const bool top_is_null = rec->top_is_null != 0;
    const unsigned long long topk = rec->topk;
    const uint8_t* vals = c.values + p.row0 * W;
    const ValidView vv{c.validity, c.validity_bit_offset + p.row0};
    const uint32_t nk = c.nk;
    const uint64_t pos = c.nullable ? def_section_bytes(N) : 0;
    uint8_t* rb = page_slot(a, c, p) + pos + 9 + W + 4;
    uint8_t* ex = a.scratch + p.ex_off;
    const uint32_t data_off = rec->data_off[cq], ex_base = rec->ex_base[cq];
    const bool bitmap = card > 4096;
    auto is_exc = [&](uint64_t i) {
        if (!vv.get(i)) return false;
        if (top_is_null) return true;
        const Val<W> k = stat_key<W>(ld_val<W>(vals + i * W), nk);
        uint64_t x = 0;
        __builtin_memcpy(&x, &k, W);
        return x != topk;
    };
    if (bitmap)   // rows past the end of the page are not visited below
        for (uint32_t i = t; i < 2048; i += WG) stu32(rb + data_off + 4 * i, 0);
    __syncthreads();
    uint32_t carry = 0;
    for (uint64_t cb = b; cb < e; cb += TILE_ROWS) {
        const uint32_t n = (uint32_t)min((uint64_t)TILE_ROWS, e - cb);
        bool f[ROWS_PER_THREAD];
#pragma unroll
        for (int j = 0; j < ROWS_PER_THREAD; j++) {
            const uint32_t r = (uint32_t)t + (uint32_t)j * WG;
            f[j] = r < n && is_exc(cb + r);
            sA[sidx((int)r)] = f[j] ? 1u : 0u;
            if (bitmap) {   // 64 consecutive rows per wave step: one u64 word of the container
                const uint64_t m = __ballot(f[j]);
                if (lane == 0 && (r & ~63u) < n) stu64(rb + data_off + ((cb - b + (r & ~63u)) >> 6) * 8, m);
            }
        }
        __syncthreads();
        const uint32_t tot = tile_incl_scan(sA, s_w);
#pragma unroll
        for (int j = 0; j < ROWS_PER_THREAD; j++) {
            const uint32_t r = (uint32_t)t + (uint32_t)j * WG;
            if (!f[j]) continue;
            const uint32_t k = carry + sA[sidx((int)r)] - 1;
            if (!bitmap) {
                const uint32_t lo16 = (uint32_t)(cb - b) + r;
                *(gptr)(rb + data_off + 2 * k) = (uint8_t)lo16;
                *(gptr)(rb + data_off + 2 * k + 1) = (uint8_t)(lo16 >> 8);
            }
            st_val<W>(ex + (uint64_t)(ex_base + k) * W, ld_val<W>(vals + (cb + r) * W));
        }
        carry += tot;
        __syncthreads();
    }
}

// the page records back to what k_enc_nested / k_enc_freq_finish expect (pad 0) once every container is written
template <int W>
__global__ void k_freq_big_done(EncodeArgs a, const uint32_t* big) {
    if (*a.freq_count == 0 || threadIdx.x) return;
    uint32_t page;
    EncPage p;
    EncCol c;
    FreqBigRec* rec;
    if (!fbig_page_of<W>(a, big, &page, &p, &c, &rec, true)) return;
    a.outs[page].pad = 0;
}
