// strawboat-hip: Dict encoder fast path (included by sb_encode.hip inside namespace sb, after PrimKeys).
//
// For keys of <= 8 bytes and <= DL_CAP distinct values per page the hash table (keys + first rows)
// lives in LDS, so a probe costs an LDS read instead of an L2 round trip plus a gather of the
// candidate's key.  Two streaming passes:
//   A  insert every keyed row (CAS on the key slot, atomicMin on the first row only when it
//      improves), remember the row's slot as u16 in HBM;
//   B  per 4096-row chunk, thread = 16 consecutive rows: rows that are their key's first occurrence
//      get dictionary ids by a wave scan (first-occurrence order, integer/dict.rs:143-160); the slot's
//      "first row" word is then overwritten by TAG|id, and every row's index follows from its slot;
//      null rows repeat the previous index (dict.rs:46-54) through a carried "last id".
// Falls back (returns DICT_FALLBACK) when the page has more distinct keys, a float NaN (which never
// equals anything, integer/dict.rs:208,225-229) or too little aux space.
constexpr uint32_t DL_SLOTS = 4096, DL_CAP = 2048, DL_TAG = 0x80000000u, DICT_FALLBACK = 0xFFFFFFFEu;

template <int W>
__device__ uint32_t dict_build_lds(const PrimKeys<W>& ko, uint64_t N, uint32_t* aux, uint64_t aux_words,
                                   uint32_t** idx_out, uint32_t** firsts_out, uint32_t* lds /* 3 * SIDX_WORDS */) {
    static_assert(W <= 8, "LDS dictionary holds keys of at most 8 bytes");
    static_assert(3 * SIDX_WORDS >= 3 * (int)DL_SLOTS + 64, "LDS area too small");
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint64_t firsts_off = (N + 3) / 4 * 4, slots_off = firsts_off + (DL_CAP + 4);
    if (N >= DL_TAG || slots_off + (N + 1) / 2 + 8 > aux_words) return DICT_FALLBACK;
    constexpr unsigned long long SENT = ~0ull;
    unsigned long long* keys = (unsigned long long*)lds;
    uint32_t* rows = lds + 2 * DL_SLOTS;
    uint32_t* misc = rows + DL_SLOTS;  // [0] entries [1] abort [2] first row of the all-ones key [3] its TAG|id, [8..] wave records
    uint32_t* idx = aux;
    uint32_t* firsts = aux + firsts_off;
    uint16_t* slots = (uint16_t*)(aux + slots_off);
    for (uint32_t i = t; i < DL_SLOTS; i += WG) {
        keys[i] = SENT;
        rows[i] = EMPTY;
    }
    if (t < 64) misc[t] = t == 2 ? EMPTY : 0;
    __syncthreads();
    // ---- pass A
    constexpr int U = 8;
    for (uint64_t ib = t; ib < N; ib += (uint64_t)WG * U) {
        Val<W> vb[U];
        bool kd[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint64_t i = ib + (uint64_t)u * WG;
            const uint64_t ic = i < N ? i : N - 1;
            vb[u] = ld_val<W>(ko.vals + ic * W);
            kd[u] = i < N && ko.vv.get(ic);
        }
        if (misc[0] > DL_CAP) break;  // too many distinct keys: the caller falls back (checked again below)
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint64_t i = ib + (uint64_t)u * WG;
            if (i >= N) break;
            unsigned long long k = 0;
            __builtin_memcpy(&k, &vb[u], W);
            if (!kd[u]) {
                if (i != 0) {
                    *(__attribute__((address_space(1))) uint16_t*)(slots + i) = 0xFFFF;
                    continue;
                }
                k = 0;  // a leading null interns T::default()
            }
            if (ko.fkind == 1 && W == 4 && (k & 0x7FFFFFFFull) > 0x7F800000ull) misc[1] = 1;
            if (ko.fkind == 2 && W == 8 && (k & 0x7FFFFFFFFFFFFFFFull) > 0x7FF0000000000000ull) misc[1] = 1;
            uint32_t h;
            if (k == SENT) {
                if ((uint32_t)i < misc[2]) atomicMin(&misc[2], (uint32_t)i);
                h = DL_SLOTS;
            } else {
                h = hash64(k + 0x9E3779B97F4A7C15ull) & (DL_SLOTS - 1);
                for (;;) {
                    const unsigned long long cur = keys[h];
                    if (cur == k) break;
                    if (cur == SENT) {
                        const unsigned long long old = atomicCAS(&keys[h], SENT, k);
                        if (old == SENT) {
                            atomicAdd(&misc[0], 1u);
                            break;
                        }
                        if (old == k) break;
                    }
                    h = (h + 1) & (DL_SLOTS - 1);
                }
                if ((uint32_t)i < rows[h]) atomicMin(&rows[h], (uint32_t)i);
            }
            *(__attribute__((address_space(1))) uint16_t*)(slots + i) = (uint16_t)h;
        }
    }
    __syncthreads();
    if (misc[1] || misc[0] > DL_CAP) return DICT_FALLBACK;
    // ---- pass B
    constexpr int K = 16;
    constexpr uint32_t CHUNK = WG * K;
    const uint64_t lt = (1ull << lane) - 1;
    uint32_t nent = 0, carry_id = 0, par = 0;
    for (uint64_t cb = 0; cb < N; cb += CHUNK, par ^= 1) {
        uint32_t* s_cnt = misc + 8 + par * 16;      // [4] first occurrences per wave
        uint32_t* s_has = misc + 8 + par * 16 + 4;  // [4] wave has a keyed row
        uint32_t* s_lid = misc + 8 + par * 16 + 8;  // [4] id of the wave's last keyed row
        const uint64_t row0 = cb + (uint64_t)t * K;
        uint16_t sl[K];
        if (row0 + K <= N) {
            uint32_t raw[K / 2];
            const u32x4 a = ldu128((const uint8_t*)(slots + row0)), b = ldu128((const uint8_t*)(slots + row0 + 8));
            raw[0] = a.x; raw[1] = a.y; raw[2] = a.z; raw[3] = a.w;
            raw[4] = b.x; raw[5] = b.y; raw[6] = b.z; raw[7] = b.w;
#pragma unroll
            for (int j = 0; j < K; j++) sl[j] = (uint16_t)(raw[j >> 1] >> (16 * (j & 1)));
        } else {
#pragma unroll
            for (int j = 0; j < K; j++) sl[j] = row0 + j < N ? ldu16((const uint8_t*)(slots + row0 + j)) : (uint16_t)0xFFFF;
        }
        uint32_t km = 0, fm = 0;
#pragma unroll
        for (int j = 0; j < K; j++) {
            if (sl[j] == 0xFFFF) continue;
            km |= 1u << j;
            const uint32_t r = sl[j] == DL_SLOTS ? misc[2] : rows[sl[j]];
            if (r == (uint32_t)(row0 + j)) fm |= 1u << j;
        }
        const uint32_t cnt = (uint32_t)__popc(fm);
        const uint32_t incl = wave_incl_scan(cnt);
        if (lane == 63) s_cnt[w] = incl;
        __syncthreads();
        uint32_t id = nent + incl - cnt;
        for (int pw = 0; pw < 3; pw++)
            if (pw < w) id += s_cnt[pw];
        nent += s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
#pragma unroll
        for (int j = 0; j < K; j++)
            if ((fm >> j) & 1) {
                if (sl[j] == DL_SLOTS)
                    misc[3] = DL_TAG | id;
                else
                    rows[sl[j]] = DL_TAG | id;
                *(__attribute__((address_space(1))) uint32_t*)(firsts + id) = (uint32_t)(row0 + j);
                id++;
            }
        __syncthreads();
        // indices: a keyed row takes its key's id, a null row repeats the previous index
        uint32_t out[K];
        uint32_t lastid = 0;
#pragma unroll
        for (int j = 0; j < K; j++) {
            if ((km >> j) & 1) lastid = (sl[j] == DL_SLOTS ? misc[3] : rows[sl[j]]) & ~DL_TAG;
            out[j] = lastid;
        }
        const uint64_t hm = __ballot(km != 0);
        const uint32_t last_w = (uint32_t)__builtin_amdgcn_readlane((int)lastid, hm ? top_bit(hm) : 0);
        if (lane == 0) {
            s_has[w] = hm != 0;
            s_lid[w] = last_w;
        }
        __syncthreads();
        uint32_t cin = carry_id;  // id of the last keyed row before this wave
        for (int pw = 0; pw < 3; pw++)
            if (pw < w && s_has[pw]) cin = s_lid[pw];
        const uint64_t pm = hm & lt;
        const uint32_t pl = __shfl(lastid, pm ? top_bit(pm) : 0, 64);
        const uint32_t tin = pm ? pl : cin;  // ... before this thread
        const int f = km ? __ffs((int)km) - 1 : K;
#pragma unroll
        for (int j = 0; j < K; j++)
            if (j < f) out[j] = tin;
        if (row0 + K <= N) {
#pragma unroll
            for (int q = 0; q < K / 4; q++)
                stu128((uint8_t*)(idx + row0 + 4 * q), u32x4{out[4 * q], out[4 * q + 1], out[4 * q + 2], out[4 * q + 3]});
        } else {
#pragma unroll
            for (int j = 0; j < K; j++)
                if (row0 + j < N) *(__attribute__((address_space(1))) uint32_t*)(idx + row0 + j) = out[j];
        }
        for (int pw = 0; pw < 4; pw++)
            if (s_has[pw]) carry_id = s_lid[pw];
    }
    __syncthreads();
    *idx_out = idx;
    *firsts_out = firsts;
    return nent;
}
