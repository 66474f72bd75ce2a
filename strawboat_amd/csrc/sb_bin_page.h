// strawboat-hip: the codec choice AND the dictionary of an adaptive Binary / LargeBinary page in ONE pass over its strings
// (included by sb_encode.hip inside namespace sb, behind k_enc_select).
//
// Replaces, for pages of BP_MIN_ROWS .. 65 536 rows, the chain k_enc_bin_hash -> k_enc_select<-4 / -8> -> k_enc_bin_verify
// (-> redo) of rounds 3-5: that chain wrote an 8-byte hash per row to HBM, read it back three times, and read the strings
// twice (hash, verify) — 6.8 GB of traffic for the 1.37 GB of C3 — in four latency-bound kernels of 256-thread workgroups,
// two per CU.  Here ONE workgroup of 1024 threads per page (one per CU: the 32 Ki-slot table is 128 KB of LDS) streams the
// rows once:
//   * row -> (offset pair, <= 32 bytes of the string in registers) -> 64-bit hash -> slot of an LDS table whose word is
//     tag15 | unkeyed | row16; a probe that meets its tag compares the STRINGS (the row's bytes are in registers, the
//     representative's come from L1 / L2: the frequent strings' first rows are hot) — the table is exact, so there is no
//     verify pass and no redo; equal strings always meet in one slot (linear probing, no deletions), atomicMin keeps the
//     smallest KEYED row (row 0 and the valid rows intern keys, binary/dict.rs:55-93), else the smallest row;
//   * the slot of every row goes to HBM as a u16 (L2-resident: 128 KB per page), the only per-row product;
//   * the statistics of binary/mod.rs:265-348 fall out of the same pass: distinct strings = inserts (the pass stops once
//     Dict's limit (N - 1) / 3 is passed: the table can never fill), their bytes = the inserters' lengths, all_equal =
//     one insert, nulls from the validity words, the Freq majority by a Boyer-Moore vote over slot numbers + one count
//     pass over the u16 slots;
//   * a page that chooses Dict gets its ids in first-occurrence order from a bitmap of the first keyed rows + prefix
//     popcounts (id = rank), the table is overwritten with slot -> id, and the u32 index array (a null row repeats the
//     index before it) is written for the page's emitter: k_enc_emit_pages<-4 / -8, Dict> starts at the nested selection.
// reference: src/compression/binary/mod.rs:293-348 (choose_compressor), binary/dict.rs:38-100, binary/freq.rs:147-169,
// binary/one_value.rs:42-48.

__device__ __forceinline__ uint32_t bp_sum(uint32_t v, uint32_t* s16) {   // sum over the 1024 threads
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s16[threadIdx.x >> 6] = v;
    __syncthreads();
    uint32_t r = 0;
#pragma unroll
    for (int k = 0; k < BP_WG / 64; k++) r += s16[k];
    return r;
}
// The hash of a string: a function of its bytes alone (dwords little-endian, the last one zero-padded), in two forms — from
// the 8 masked dwords of a string of <= 32 bytes in registers, and bytewise from memory (long strings, and the last rows of a
// column, whose 32-byte loads would leave the buffer) — so that the copies of a string meet in one probe sequence wherever they sit.
constexpr uint32_t BP_REGW = 6;   // dwords of a string held in registers by the row loop (longer strings: the second loop)
__device__ __forceinline__ uint32_t bp_hash_step(uint32_t h, uint32_t d) {   // rotate + add: v_mul_lo_u32 is quarter rate, six of them
    return ((h << 5) | (h >> 27)) + d;                                           // per row were a third of the row loop's issue time
}
__device__ __forceinline__ uint32_t bp_hash_fin(uint32_t h) {
    h ^= h >> 16;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    return h ^ (h >> 16);
}
__device__ __forceinline__ uint32_t bp_hash_init(uint32_t n) { return n + 0x85EBCA6Bu; }
__device__ uint32_t bp_hash_mem(const uint8_t* values, uint64_t b, uint32_t n) {
    uint32_t h = bp_hash_init(n);
    const uint32_t nd = n <= 4 * BP_REGW ? 4 * BP_REGW : n;   // (the register form runs over all its dwords, zeros included)
    for (uint32_t k = 0; k < nd; k += 4) {
        uint32_t d = 0;
        for (uint32_t q = 0; q < 4 && k + q < n; q++) d |= (uint32_t)ldu8(values + b + k + q) << (8 * q);
        h = bp_hash_step(h, d);
    }
    return bp_hash_fin(h);
}
// ---- ids in first-occurrence order out of a settled table (words: tag | BP_UNKEYED | row16, BP_EMPTY), then the u32 index
// array.  firsts[id] at aux + BP_W_FIRSTS, the indices in the last N words of the aux area; the table is left as slot -> id.
// Returns the number of entries.  Shared by k_enc_bin_page and k_enc_prim_dict.
__device__ __forceinline__ uint32_t bp_ids_and_index(uint32_t* tab, const uint32_t nslots, uint32_t* s_x, uint32_t* s_y, uint32_t* s_w, uint32_t* aux,
                                                     const uint16_t* slot16, const ValidView& vv, const uint32_t N) {
    const uint32_t t = threadIdx.x, lane = t & 63, wv = t >> 6;
    uint32_t D = 0;
    {
        STL(66);
        // ---- ids in first-occurrence order: bitmap of the first KEYED rows, prefix popcounts, id = rank
        s_x[t] = 0;
        s_x[t + BP_WG] = 0;
        __syncthreads();
        for (uint32_t sl = t; sl < nslots; sl += BP_WG) {
            const uint32_t wd = tab[sl];
            if (wd != BP_EMPTY && !(wd & BP_UNKEYED)) atomicOr(&s_x[(wd & 0xFFFFu) >> 5], 1u << (wd & 31u));
        }
        __syncthreads();
        {
            const uint32_t m0 = (uint32_t)__popc(s_x[2 * t]), m1 = (uint32_t)__popc(s_x[2 * t + 1]);
            const uint32_t incl = wave_incl_scan(m0 + m1);
            if (lane == 63) s_w[wv] = incl;
            __syncthreads();
            uint32_t run = incl - (m0 + m1);
            for (uint32_t pw = 0; pw < wv; pw++) run += s_w[pw];
#pragma unroll
            for (int k = 0; k < BP_WG / 64; k++) D += s_w[k];
            s_y[2 * t] = run;
            s_y[2 * t + 1] = run + m0;
        }
        __syncthreads();
        uint32_t* firsts = aux + BP_W_FIRSTS;
        for (uint32_t sl = t; sl < nslots; sl += BP_WG) {
            const uint32_t wd = tab[sl];
            if (wd == BP_EMPTY) continue;
            uint32_t id = 0xFFFFFFFFu;
            if (!(wd & BP_UNKEYED)) {
                const uint32_t r = wd & 0xFFFFu;
                id = s_y[r >> 5] + (uint32_t)__popc(s_x[r >> 5] & ((1u << (r & 31)) - 1u));
                gst32(firsts + id, r);
            }
            tab[sl] = id;
        }
        __syncthreads();
        STL(67);
        // ---- the u32 index array (the last N words of the aux area, where the emitter's LZ4 scratch does not reach)
        uint32_t* idx = aux + bh_table_slots(N) + 2 * (uint64_t)N;
        if (!vv.bits) {
            const uint32_t npair = (N + 1) / 2;
            for (uint32_t k0 = t; k0 < npair; k0 += BP_WG * 8) {
                uint32_t pr[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const uint32_t k = k0 + (uint32_t)u * BP_WG;
                    pr[u] = gld32((const uint32_t*)slot16 + (k < npair ? k : 0));
                }
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const uint32_t k = k0 + (uint32_t)u * BP_WG;
                    if (k >= npair) continue;
                    const uint32_t i0 = tab[pr[u] & 0xFFFFu];
                    if (2 * k + 1 < N)
                        gst64((uint64_t*)(idx + 2 * k), (uint64_t)i0 | ((uint64_t)tab[pr[u] >> 16] << 32));
                    else
                        gst32(idx + 2 * k, i0);
                }
            }
        } else {
            // a null row repeats the index before it (dict.rs:46-55): per 64-row chunk (a wave's step) the last keyed id, an
            // inclusive "last one that has any" scan over the <= 1024 chunks, then the rows with the carry of the chunk before
            const uint32_t nch = (N + 63) / 64;
            constexpr int V = 4;   // (the loads of four steps together)
            for (uint32_t base = 0; base < N; base += BP_WG * V) {
                uint32_t sl[V], vb[V];
#pragma unroll
                for (int u = 0; u < V; u++) {
                    const uint32_t i = base + (uint32_t)u * BP_WG + t, ic = i < N ? i : N - 1;
                    sl[u] = ldu16((const uint8_t*)(slot16 + ic));
                    vb[u] = ldu8(vv.bits + ((vv.off + ic) >> 3));
                }
#pragma unroll
                for (int u = 0; u < V; u++) {
                    const uint32_t i = base + (uint32_t)u * BP_WG + t, ic = i < N ? i : N - 1;
                    const bool kd = i < N && (i == 0 || ((vb[u] >> ((vv.off + ic) & 7)) & 1));
                    const uint32_t sid = kd ? tab[sl[u]] : 0u;
                    const uint64_t km = __ballot(kd);
                    const uint32_t last = __shfl(sid, km ? 63 - __clzll((long long)km) : 0, 64);
                    const uint32_t ch = (base + (uint32_t)u * BP_WG) / 64 + wv;
                    if (lane == 0 && ch < nch) s_x[ch] = km ? last : 0xFFFFFFFFu;
                }
            }
            __syncthreads();
            {
                const uint32_t v = t < nch ? s_x[t] : 0xFFFFFFFFu;
                const uint64_t hm = __ballot(v != 0xFFFFFFFFu);
                const uint64_t below = hm & ((lane == 63) ? ~0ull : ((2ull << lane) - 1));
                const uint32_t src = below ? 63 - (uint32_t)__clzll((long long)below) : 0;
                const uint32_t got = __shfl(v, (int)src, 64);
                uint32_t incl = below ? got : 0xFFFFFFFFu;
                if (lane == 63) s_w[wv] = incl;
                __syncthreads();
                if (incl == 0xFFFFFFFFu)
                    for (int pw = (int)wv - 1; pw >= 0; pw--)
                        if (s_w[pw] != 0xFFFFFFFFu) {
                            incl = s_w[pw];
                            break;
                        }
                s_y[t] = incl;
            }
            __syncthreads();
            for (uint32_t base = 0; base < N; base += BP_WG * V) {
                uint32_t sl[V], vb[V];
#pragma unroll
                for (int u = 0; u < V; u++) {
                    const uint32_t i = base + (uint32_t)u * BP_WG + t, ic = i < N ? i : N - 1;
                    sl[u] = ldu16((const uint8_t*)(slot16 + ic));
                    vb[u] = ldu8(vv.bits + ((vv.off + ic) >> 3));
                }
#pragma unroll
                for (int u = 0; u < V; u++) {
                    const uint32_t i = base + (uint32_t)u * BP_WG + t, ic = i < N ? i : N - 1;
                    const bool kd = i < N && (i == 0 || ((vb[u] >> ((vv.off + ic) & 7)) & 1));
                    const uint32_t sid = kd ? tab[sl[u]] : 0u;
                    const uint64_t km = __ballot(kd);
                    const uint64_t below = km & ((lane == 63) ? ~0ull : ((2ull << lane) - 1));
                    const uint32_t src = below ? 63 - (uint32_t)__clzll((long long)below) : 0;
                    const uint32_t got = __shfl(sid, (int)src, 64);
                    const uint32_t ch = (base + (uint32_t)u * BP_WG) / 64 + wv;
                    const uint32_t carry = ch && ch <= nch ? s_y[ch - 1] : 0u;
                    if (i < N) gst32(idx + i, below ? got : carry);
                }
            }
        }
        }
    return D;
}

// ---- the codec of a Dict page's index block: compress_integer::<u32> with Dict forbidden (binary/dict.rs:60-62, integer/
// dict.rs:57-62 -> integer/mod.rs:231-347), the arithmetic of choose_prim / decide_prim on what the fused kernels know.
__device__ __forceinline__ uint32_t bp_index_codec(const EncodeArgs& a, const EncPage& p, const uint32_t* idx, const uint32_t N, const uint32_t D,
                                                   const uint32_t forb_n, uint32_t* s_x, uint32_t* s_y, uint32_t* s_w) {
    const uint32_t t = threadIdx.x;
    {
            const uint32_t i00 = gld32(idx);
            uint32_t f_neq0 = 0, f_uns = 0, vc = 0, vn = 0;
            for (uint32_t i0 = t * 4; i0 < N; i0 += BP_WG * 4 * 4) {   // (four 16-byte loads per thread and step)
                u32x4 q[4];
                uint32_t pv[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const uint32_t i = i0 + (uint32_t)u * BP_WG * 4;
                    q[u] = i + 4 <= N ? *(const __attribute__((address_space(1))) u32x4*)(idx + i) : u32x4{i00, i00, i00, i00};
                    pv[u] = (i && i < N) ? gld32(idx + i - 1) : 0u;
                    if (i < N && i + 4 > N) {   // (N % 4 != 0: the last values one by one)
                        uint32_t w4[4] = {0, 0, 0, 0};
                        for (uint32_t b = 0; i + b < N; b++) w4[b] = gld32(idx + i + b);
                        for (uint32_t b = N - i; b < 4; b++) w4[b] = w4[N - i - 1];
                        q[u] = u32x4{w4[0], w4[1], w4[2], w4[3]};
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const uint32_t i = i0 + (uint32_t)u * BP_WG * 4;
                    if (i >= N) continue;
                    const uint32_t w4[4] = {q[u].x, q[u].y, q[u].z, q[u].w};
                    uint32_t prev = pv[u];
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                        if (i + b >= N) break;
                        if (w4[b] != i00) f_neq0 = 1;
                        if ((i + b) && w4[b] < prev) f_uns = 1;
                        prev = w4[b];
                        if (vn == 0) {
                            vc = w4[b];
                            vn = 1;
                        } else if (w4[b] == vc) {
                            vn++;
                        } else {
                            vn--;
                        }
                    }
                }
            }
            const uint32_t flags = bp_sum(f_neq0 | (f_uns << 16), s_w);
            const bool n_all_equal = !(flags & 0xFFFFu), n_sorted = !(flags >> 16);
            uint32_t nmc = 0;
            if (!((forb_n >> SB_CODEC_FREQ) & 1) && !n_all_equal && D - 1 >= 256) {
                s_x[t] = vc;
                s_y[t] = vn;
                __syncthreads();
                for (uint32_t stride = BP_WG / 2; stride > 0; stride >>= 1) {
                    if (t < stride) {
                        const uint32_t c0 = s_x[t], n0 = s_y[t], c1 = s_x[t + stride], n1 = s_y[t + stride];
                        uint32_t cc = c0, nn = n0;
                        if (n1) {
                            if (n0 == 0) { cc = c1; nn = n1; }
                            else if (c0 == c1) nn = n0 + n1;
                            else if (n1 > n0) { cc = c1; nn = n1 - n0; }
                            else nn = n0 - n1;
                        }
                        s_x[t] = cc;
                        s_y[t] = nn;
                    }
                    __syncthreads();
                }
                const uint32_t cand = s_x[0], cn = s_y[0];
                __syncthreads();
                uint32_t mine = 0;
                if (cn && (double)cn + 1.0 >= 0.0)   // (count the candidate: a 90 % majority survives any merge order)
                    for (uint32_t i0 = t * 4; i0 < N; i0 += BP_WG * 4) {
                        if (i0 + 4 <= N) {
                            const u32x4 v = *(const __attribute__((address_space(1))) u32x4*)(idx + i0);
                            mine += (v.x == cand) + (v.y == cand) + (v.z == cand) + (v.w == cand);
                        } else {
                            for (uint32_t b = 0; i0 + b < N; b++) mine += gld32(idx + i0 + b) == cand;
                        }
                    }
                nmc = bp_sum(mine, s_w);
            }
            // the three trials: 640 sampled rows each (or the whole page when N / 10 <= 64), one row per thread
            const bool whole = N / SAMPLE_COUNT <= SAMPLE_SIZE;
            const uint32_t sn = whole ? N : SAMPLE_ROWS;
            auto trial = [&](uint32_t cd, bool rle) -> uint32_t {   // RLE: runs of the sample; else its bit-packed size
                __syncthreads();
                if (t < sn) {
                    uint64_t row = 0;
                    sample_row(N, p.seed, p.depth + 1, cd, t, row);
                    s_x[t] = gld32(idx + row);
                }
                if (t < 8) s_y[t] = 0;
                __syncthreads();
                uint32_t v = 0;
                if (rle) {
                    v = bp_sum(t > 0 && t < sn && s_x[t] != s_x[t - 1] ? 1u : 0u, s_w);
                    return sn ? v + 1 : 0;
                }
                const uint32_t nblk = sn / 128;
                if (t < nblk * 128) atomicOr(&s_y[t >> 7], s_x[t]);
                __syncthreads();
                for (uint32_t b = 0; b < nblk; b++) v += 1 + 16 * (s_y[b] ? 32 - __clz(s_y[b]) : 0);
                return v;
            };
            const double n_tuple = (double)N;
            double n_max = a.ratio;
            uint32_t n_res = a.default_compression;
            static const uint8_t NORD[5] = {SB_CODEC_ONEVALUE, SB_CODEC_FREQ, SB_CODEC_RLE, SB_CODEC_BITPACKING, SB_CODEC_DELTA_BITPACKING};
            for (int oi = 0; oi < 5; oi++) {
                const uint32_t cd = NORD[oi];
                if ((forb_n >> cd) & 1) continue;
                double r = 0.0;
                if (cd == SB_CODEC_ONEVALUE) {
                    r = n_all_equal ? n_tuple : 0.0;
                } else if (cd == SB_CODEC_FREQ) {
                    if (!n_all_equal && (double)nmc / n_tuple >= 0.9 && (int64_t)(D - 1) >= 256) r = (double)(N - 1);
                } else if (cd == SB_CODEC_RLE) {
                    const uint32_t runs = trial(cd, true);
                    r = (double)((uint64_t)sn * 4) / (double)((uint64_t)runs * 8);
                } else {
                    if (N % 128 != 0) continue;
                    if (cd == SB_CODEC_DELTA_BITPACKING && !n_sorted) continue;
                    const uint32_t size = trial(cd, false);
                    r = (double)((uint64_t)sn * 4) / (double)size;
                    if (cd == SB_CODEC_DELTA_BITPACKING) r *= 1.5;
                }
                if (r > n_max) {
                    n_max = r;
                    n_res = cd;
                    if (r == n_tuple) break;
                }
            }
            __syncthreads();
        return n_res;
    }
}

// ---- the index array bit-packed at dst (integer/bp.rs:45-61; BitPacker4x: a width byte, then 4 interleaved lanes per block
// of 128), 32 768 indices per step through `tab` (the table's LDS, free by then); returns the bytes written.
__device__ __forceinline__ uint32_t bp_pack_indices(uint8_t* dst, const uint32_t* idx, const uint32_t N, uint32_t* tab, uint32_t* s_x, uint32_t* s_w) {
    const uint32_t t = threadIdx.x, lane = t & 63, wv = t >> 6;
    {
            constexpr uint32_t CH = 32768, NBLK = CH / 128, TPB = BP_WG / NBLK;   // 256 blocks per step, four threads per block
            uint32_t* s_nb = s_x;              // [NBLK] widths
            uint32_t* s_off = s_x + NBLK;      // [NBLK + 1] byte offsets inside the step
            uint32_t out_pos = 0;
            for (uint32_t cb = 0; cb < N; cb += CH) {
                const uint32_t n = min(CH, N - cb), nblk = n / 128;
                __syncthreads();
                uint32_t acc = 0;
                {   // 32 consecutive values per thread: staged, and OR-ed for the block's width
                    const uint32_t v0 = t * (128 / TPB);
                    if (v0 < n) {
#pragma unroll
                        for (int q = 0; q < 8; q++) {
                            const u32x4 v = *(const __attribute__((address_space(1))) u32x4*)(idx + cb + v0 + 4 * q);
                            *(__attribute__((address_space(3))) u32x4*)((l32p)tab + v0 + 4 * q) = v;
                            acc |= v.x | v.y | v.z | v.w;
                        }
                    }
                    acc |= __shfl_xor(acc, 1, 64);
                    acc |= __shfl_xor(acc, 2, 64);
                    if ((t & (TPB - 1)) == 0 && t / TPB < nblk) s_nb[t / TPB] = acc ? 32 - __clz(acc) : 0;
                }
                __syncthreads();
                if (t < NBLK) {   // byte offsets of the step's blocks (four waves)
                    const uint32_t nb = t < nblk ? s_nb[t] : 0u;
                    const uint32_t by = t < nblk ? 1 + 16 * nb : 0u;
                    const uint32_t ib = wave_incl_scan(by);
                    if (lane == 63) s_w[wv] = ib;
                    s_off[t] = ib - by;
                }
                __syncthreads();
                if (t < NBLK) {
                    uint32_t add = 0;
                    for (uint32_t pw = 0; pw < wv; pw++) add += s_w[pw];
                    s_off[t] += add;
                }
                const uint32_t step_bytes = s_w[0] + s_w[1] + s_w[2] + s_w[3];
                __syncthreads();
                {   // pack: the step's blocks side by side, a word per thread and turn
                    const uint32_t blk = t / TPB;
                    if (blk < nblk) {
                        const uint32_t nb = s_nb[blk], nw = 4 * nb;
                        uint8_t* bo = dst + out_pos + s_off[blk] + 1;
                        if ((t & (TPB - 1)) == 0) *(gptr)(bo - 1) = (uint8_t)nb;
                        for (uint32_t wi = t & (TPB - 1); wi < nw; wi += TPB) {
                            const uint32_t l = wi & 3, k = wi >> 2;  // word k of lane l
                            const uint32_t lo_bit = 32 * k, hi_bit = 32 * k + 32;
                            uint32_t word = 0;
                            const uint32_t i0 = lo_bit / nb, i1 = min(31u, (hi_bit - 1) / nb);
                            for (uint32_t i = i0; i <= i1; i++) {
                                const uint32_t v = tab[blk * 128 + 4 * i + l];
                                const uint32_t bitpos = i * nb;
                                if (bitpos >= lo_bit) word |= v << (bitpos - lo_bit);
                                else if (bitpos + nb > lo_bit) word |= v >> (lo_bit - bitpos);
                            }
                            stu32(bo + 4 * wi, word);
                        }
                    }
                }
                out_pos += step_bytes;
            }
            return out_pos;
    }
}

template <class O>
__global__ void __launch_bounds__(BP_WG) k_enc_bin_page(EncodeArgs a) {
    __shared__ __attribute__((aligned(16))) uint32_t tab[BP_SLOTS];
    __shared__ uint32_t s_x[2048];   // vote candidates | bitmap of first rows | last keyed id per 64-row chunk
    __shared__ uint32_t s_y[2048];   // vote counts | word prefixes | carried id per chunk
    __shared__ uint32_t s_w[BP_WG / 64];
    __shared__ uint32_t s_cnt;
    __shared__ unsigned long long s_tus;
    // byte masks of the six dwords of a string of n <= 24 bytes (row n, eight words): two LDS reads per use instead of ~30
    // compare / select instructions (each v_cmp -> v_cndmask pair costs a wait state on top)
    __shared__ __attribute__((aligned(32))) uint32_t s_msk[(4 * BP_REGW + 1) * 8];
    const uint32_t t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const uint32_t page = spread_block(blockIdx.x, gridDim.x) + a.page_base;
    const EncPage p = get_page(a, page);
    if (!bp_page_ok(a, p, page)) return;
    const EncCol c = get_col(a, p.col);
    if (c.ptype != (sizeof(O) == 4 ? SB_TYPE_BINARY : SB_TYPE_LARGE_BINARY)) return;
    const uint32_t N = (uint32_t)p.rows;
    uint32_t* aux = (uint32_t*)(a.scratch + p.aux_off);
    uint16_t* slot16 = (uint16_t*)(aux + bp_w_slot16(N));
    const uint8_t* offs = c.offsets + p.row0 * sizeof(O);
    const uint8_t* values = c.values;
    const uint64_t vlen = c.values_len;
    const ValidView vv{c.validity, c.validity_bit_offset + p.row0};
    const BinKeys<O> bk{offs, values, vv};
    STL(60);
    for (uint32_t i = t; i < BP_SLOTS; i += BP_WG) tab[i] = BP_EMPTY;
    if (t < (4 * BP_REGW + 1) * 8) {
        const uint32_t n = t >> 3, k = t & 7, have = n > 4 * k ? n - 4 * k : 0;
        s_msk[t] = have >= 4 ? 0xFFFFFFFFu : (1u << (8 * have)) - 1u;
    }
    if (t == 0) {
        s_cnt = 0;
        s_tus = 0;
    }
    __syncthreads();
    STL(61);
    const uint32_t limit = (N - 1) / 3;
    uint32_t bm_c = 0, bm_n = 0;
    unsigned long long my_tus = 0, slow_rows = 0;   // (bit k: row k * BP_WG + t)
    constexpr int U = 4;
    const uint8_t* dummy = (const uint8_t*)aux;   // 64 readable bytes for the loads of lanes that have nothing to load
    const uint8_t* vsafe = vlen >= 32 ? values : dummy;   // (uniform) base of the 32-byte loads: idle lanes read its first bytes
    using Off = typename std::conditional<sizeof(O) == 4, uint32_t, uint64_t>::type;   // (Arrow offsets are not negative)
    // The table is 8192 BUCKETS of four slots (one ds_read_b128 per probe): a key takes the first free slot of its bucket —
    // slots fill in order and are never freed, so a string's copies find it there — and moves on to the next bucket only
    // when all four hold other keys.  With one slot per probe a tenth of the rows needed a second round, and a round costs
    // the whole wave ~500 instructions whatever the number of lanes that still look.
    constexpr uint32_t NBUCK = BP_SLOTS / 4;
    // offsets and validity bytes of a step are requested one step ahead (they depend on nothing)
    uint64_t po0[U], po1[U];
    uint32_t pvb[U];
    auto request = [&](uint32_t base) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t i = base + (uint32_t)u * BP_WG + t;
            const uint32_t ic = i < N ? i : N - 1;
            if constexpr (sizeof(O) == 4) {
                po0[u] = ldu64(offs + (uint64_t)ic * 4);
                po1[u] = 0;
            } else {
                po0[u] = ldu64(offs + (uint64_t)ic * 8);
                po1[u] = ldu64(offs + (uint64_t)ic * 8 + 8);
            }
            pvb[u] = vv.bits ? (uint32_t)ldu8(vv.bits + ((vv.off + ic) >> 3)) : 0xFFu;
        }
    };
    request(0);
    for (uint32_t base = 0; base < N; base += BP_WG * U) {
        STL(80 + base / (BP_WG * U));
        // (no barrier in the loop: every wave adds its inserts before it looks again, so the count overshoots the limit by at
        // most one step of the 16 waves = 4096 keys)
        if (__hip_atomic_load(&s_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) > limit) break;
        // Every phase issues the loads of its U rows together, unconditionally (idle lanes read the buffer's first bytes): a
        // step is three round trips — bytes, the representatives' offsets, their bytes — whatever the rows hold.
        Off b[U];
        uint32_t len[U], hb[U], fin[U], word[U], w[U][BP_REGW];   // hb: the bucket a row looks at; fin: the slot it settled in
        uint32_t pend = 0, fastm = 0, keyedm = 0, skip = 0;   // skip: 4 bits per row, slots of the bucket whose string differs
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t i = base + (uint32_t)u * BP_WG + t;
            const uint32_t ic = i < N ? i : N - 1;
            Off e;
            if constexpr (sizeof(O) == 4) {
                b[u] = (uint32_t)po0[u];
                e = (uint32_t)(po0[u] >> 32);
            } else {
                b[u] = po0[u];
                e = po1[u];
            }
            len[u] = (uint32_t)(e - b[u]);
            if (i < N && (Off)(e - b[u]) <= 4 * BP_REGW && (uint64_t)b[u] + 32 <= vlen) fastm |= 1u << u;
            else if (i < N) slow_rows |= 1ull << (base / BP_WG + (uint32_t)u);   // (long strings, the column's last rows: second loop)
            if (i == 0 || ((pvb[u] >> ((vv.off + ic) & 7)) & 1)) keyedm |= 1u << u;
        }
        if (base + BP_WG * U < N) request(base + BP_WG * U);
        {
            u32x4 q0[U];
            uint64_t q1[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                // (a gather costs the CU's address unit a cycle per ACTIVE lane: idle lanes and short strings stay out)
                q0[u] = u32x4{0, 0, 0, 0};
                q1[u] = 0;
                if ((fastm >> u) & 1) q0[u] = ldu128(values + b[u]);
                if (((fastm >> u) & 1) && len[u] > 16) q1[u] = ldu64(values + b[u] + 16);
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint32_t n = ((fastm >> u) & 1) ? len[u] : 0;
                const l32p mp = (l32p)s_msk + n * 8;
                const uint32_t qq[6] = {q0[u].x, q0[u].y, q0[u].z, q0[u].w, (uint32_t)q1[u], (uint32_t)(q1[u] >> 32)};
#pragma unroll
                for (uint32_t k = 0; k < BP_REGW; k++) w[u][k] = qq[k] & mp[k];
            }
        }
        pend = fastm;
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t i = base + (uint32_t)u * BP_WG + t;
            uint32_t h = bp_hash_init(len[u]);
#pragma unroll
            for (uint32_t k = 0; k < BP_REGW; k++) h = bp_hash_step(h, w[u][k]);   // (all six dwords, zeros included: see bp_hash_mem)
            h = bp_hash_fin(h);
            hb[u] = h & (NBUCK - 1);
            fin[u] = 0;
            uint32_t tag = (h >> 13) & 0x7FFFu;
            if (tag == 0x7FFFu) tag = 0x7FFEu;
            word[u] = (tag << 17) | (((keyedm >> u) & 1) ? 0u : BP_UNKEYED) | (i & 0xFFFFu);
        }
        uint32_t newk = 0;
#ifdef SB_BP_DEBUG
        uint32_t dbg_it = 0;
        const uint32_t dbg_f0 = fastm;
#endif
        while (pend) {
#ifdef SB_BP_DEBUG
            dbg_it++;
#endif
            u32x4 bk4[U];
            uint32_t rep[U], cur[U];
            uint32_t cmp = 0, selm = 0;   // selm: two bits per row, the slot of its bucket a row compares with
#pragma unroll
            for (int u = 0; u < U; u++) {   // (four u32 reads of one aligned bucket: the compiler makes them a ds_read_b128)
                const l32p bp = (l32p)tab + hb[u] * 4;
                bk4[u] = u32x4{bp[0], bp[1], bp[2], bp[3]};
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                rep[u] = 0;
                cur[u] = 0;
                if (!((pend >> u) & 1)) continue;
                // the first slot of the bucket that is free or carries my tag (and was not found to hold another string)
                const uint32_t sk = (skip >> (4 * u)) & 15u, tg = word[u] >> 17;
                const uint32_t s4[4] = {bk4[u].x, bk4[u].y, bk4[u].z, bk4[u].w};
                uint32_t sel = 4, cw = 0;
#pragma unroll
                for (int k = 3; k >= 0; k--) {
                    const bool ok = (s4[k] == BP_EMPTY || (s4[k] >> 17) == tg) && !((sk >> k) & 1);
                    sel = ok ? (uint32_t)k : sel;
                    cw = ok ? s4[k] : cw;
                }
                if (sel == 4) {   // four other keys: the next bucket
                    hb[u] = (hb[u] + 1) & (NBUCK - 1);
                    skip &= ~(15u << (4 * u));
                    continue;
                }
                const uint32_t si = hb[u] * 4 + sel;
                if (cw == BP_EMPTY) {
                    cw = atomicCAS(&tab[si], BP_EMPTY, word[u]);
                    if (cw == BP_EMPTY) {
                        newk++;
                        my_tus += (unsigned long long)len[u] + 8;
                        pend &= ~(1u << u);
                        fin[u] = si;
                        continue;
                    }
                    if ((cw >> 17) != tg) continue;   // (another key took the slot: look at the bucket again)
                }
                cmp |= 1u << u;
                rep[u] = cw & 0xFFFFu;
                cur[u] = cw;
                selm |= sel << (2 * u);
            }
            if (cmp) {
                Off rb[U];
                uint32_t cf = 0;
                {
                    uint64_t o0[U], o1[U];
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        o0[u] = o1[u] = 0;
                        if (!((cmp >> u) & 1)) continue;
                        if constexpr (sizeof(O) == 4) {
                            o0[u] = ldu64(offs + (uint64_t)rep[u] * 4);
                        } else {
                            o0[u] = ldu64(offs + (uint64_t)rep[u] * 8);
                            o1[u] = ldu64(offs + (uint64_t)rep[u] * 8 + 8);
                        }
                    }
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        uint32_t rl;
                        if constexpr (sizeof(O) == 4) {
                            rb[u] = (uint32_t)o0[u];
                            rl = (uint32_t)(o0[u] >> 32) - (uint32_t)o0[u];
                        } else {
                            rb[u] = o0[u];
                            rl = (uint32_t)(o1[u] - o0[u]);
                        }
                        if (((cmp >> u) & 1) && rl == len[u]) cf |= 1u << u;   // (a representative in here is a row of this loop: 32 readable bytes)
                    }
                }
                uint32_t eq = 0;
                {
                    u32x4 q0[U];
                    uint64_t q1[U];
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        q0[u] = u32x4{0, 0, 0, 0};
                        q1[u] = 0;
                        if ((cf >> u) & 1) q0[u] = ldu128(values + rb[u]);
                        if (((cf >> u) & 1) && len[u] > 16) q1[u] = ldu64(values + rb[u] + 16);
                    }
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        const l32p mp = (l32p)s_msk + (((cf >> u) & 1) ? len[u] : 0u) * 8;
                        const uint32_t qq[6] = {q0[u].x, q0[u].y, q0[u].z, q0[u].w, (uint32_t)q1[u], (uint32_t)(q1[u] >> 32)};
                        uint32_t d = 0;
#pragma unroll
                        for (uint32_t k = 0; k < BP_REGW; k++) d |= (qq[k] & mp[k]) ^ w[u][k];
                        if (((cf >> u) & 1) && d == 0) eq |= 1u << u;
                    }
                }
#pragma unroll
                for (int u = 0; u < U; u++) {
                    if (!((cmp >> u) & 1)) continue;
                    const uint32_t sel = (selm >> (2 * u)) & 3u, si = hb[u] * 4 + sel;
                    if ((eq >> u) & 1) {
                        if (word[u] < cur[u]) atomicMin(&tab[si], word[u]);   // (later rows of a class: nothing to do)
                        pend &= ~(1u << u);
                        fin[u] = si;
                    } else {
                        skip |= 1u << (4 * u + sel);   // (one tag, another string)
                    }
                }
            }
        }
        if (newk) atomicAdd(&s_cnt, newk);
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t i = base + (uint32_t)u * BP_WG + t;
            if (!((fastm >> u) & 1)) continue;
            const uint32_t sl = fin[u];
#ifdef SB_BP_DEBUG
            gst32(aux + bh_table_slots(N) + 2 * (uint64_t)N + i, 0xD0000000u | (dbg_it << 16) | (dbg_f0 << 8) | (fastm << 4) | pend);
#endif
            *(__attribute__((address_space(1))) uint16_t*)(slot16 + i) = (uint16_t)sl;
            if (bm_n == 0) {
                bm_c = sl;
                bm_n = 1;
            } else if (sl == bm_c) {
                bm_n++;
            } else {
                bm_n--;
            }
        }
    }
    // ---- the rows the loop above left: strings of more than 4 * BP_REGW bytes and the last rows of the column (a 32-byte load
    // would leave the buffer), one at a time, hashed and compared from memory.  A class's smallest row does not depend on the
    // order of the inserts, and the representatives the first loop compared with were rows of the first loop.
    STL(62);
    __syncthreads();   // (from here on a representative may sit at the buffer's end)
    STL(63);
    while (slow_rows) {
        if (__hip_atomic_load(&s_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) > limit) break;
        const uint32_t k = (uint32_t)__ffsll((long long)slow_rows) - 1;
        slow_rows &= slow_rows - 1;
        const uint32_t i = k * BP_WG + t;
        const uint64_t b0 = bk.beg(i);
        const uint32_t n = (uint32_t)(bk.beg((uint64_t)i + 1) - b0);
        const uint32_t h = bp_hash_mem(values, b0, n);
        uint32_t hbk = h & (NBUCK - 1);
        uint32_t tag = (h >> 13) & 0x7FFFu;
        if (tag == 0x7FFFu) tag = 0x7FFEu;
        const uint32_t wd = (tag << 17) | ((i == 0 || vv.get(i)) ? 0u : BP_UNKEYED) | (i & 0xFFFFu);
        uint32_t si = 0;
        for (uint32_t q = 0;; q++) {   // slot q & 3 of the bucket; slots fill in order
            si = hbk * 4 + (q & 3);
            uint32_t cw = tab[si];
            if (cw == BP_EMPTY) {
                cw = atomicCAS(&tab[si], BP_EMPTY, wd);
                if (cw == BP_EMPTY) {
                    atomicAdd(&s_cnt, 1u);
                    my_tus += (unsigned long long)n + 8;
                    break;
                }
            }
            if ((cw >> 17) == tag && bk.eq(cw & 0xFFFFu, i)) {
                if (wd < cw) atomicMin(&tab[si], wd);
                break;
            }
            if ((q & 3) == 3) hbk = (hbk + 1) & (NBUCK - 1);
        }
        *(__attribute__((address_space(1))) uint16_t*)(slot16 + i) = (uint16_t)si;
        if (bm_n == 0) {
            bm_c = si;
            bm_n = 1;
        } else if (si == bm_c) {
            bm_n++;
        } else {
            bm_n--;
        }
    }
    if (my_tus) atomicAdd(&s_tus, my_tus);
    uint32_t nulls = 0;
    if (vv.bits)
        for (uint32_t w = t; w * 32 < N; w += BP_WG) {
            const uint32_t nb = min(32u, N - w * 32);
            const uint32_t wd = bits32(vv.bits, vv.off + (uint64_t)w * 32, vv.off + N) & (nb >= 32 ? 0xFFFFFFFFu : (1u << nb) - 1);
            nulls += nb - (uint32_t)__popc(wd);
        }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    STL(64);
#ifdef SB_BP_DEBUG
    if (s_cnt <= limit) {   // every row's slot holds a key
        uint32_t bad = 0, badrow = 0, badslot = 0;
        for (uint32_t i = t; i < N; i += BP_WG) {
            const uint32_t sl = ldu16((const uint8_t*)(slot16 + i));
            if (tab[sl] == BP_EMPTY) {
                if (!bad) { badrow = i; badslot = sl; }
                bad++;
            }
        }
        uint32_t neq = 0, neqrow = 0;
        for (uint32_t i = t; i < N; i += BP_WG) {
            const uint32_t sl = ldu16((const uint8_t*)(slot16 + i));
            const uint32_t wd = tab[sl];
            if (wd != BP_EMPTY && !bk.eq(wd & 0xFFFFu, i)) {
                if (!neq) neqrow = i;
                neq++;
            }
        }
        if (neq && page == 0 && t < 400) {
            const uint32_t i = neqrow;
            const uint64_t b0 = bk.beg(i);
            const uint32_t n = (uint32_t)(bk.beg((uint64_t)i + 1) - b0);
            const uint32_t h = bp_hash_mem(values, b0, n);
            const uint32_t sl = ldu16((const uint8_t*)(slot16 + i));
            uint32_t found = 0xFFFFFFFFu;
            for (uint32_t q = 0; q < BP_SLOTS; q++) {
                const uint32_t wd = tab[q];
                if (wd != BP_EMPTY && bk.eq(wd & 0xFFFFu, i)) { found = q; break; }
            }
            printf("dbg %08x ", gld32(aux + bh_table_slots(N) + 2 * (uint64_t)N + i));
            printf("page %u thread %u row %u (+%u more) len %u: slot16 %u holds %08x; hash %08x bucket %u tag %04x; string found in slot %u (%08x)\n", page, t, i, neq - 1, n, sl,
                   tab[sl], h, h & 8191, (h >> 13) & 0x7FFF, found, found != 0xFFFFFFFFu ? tab[found] : 0u);
        }
        if (t == 0) printf("page %u: s_cnt %u\n", page, s_cnt);
        if (bad) {
            raise(a.status, SB_ERR_INVALID, page, 900000 + bad);
            printf("page %u thread %u: %u rows in empty slots, first row %u slot %u\n", page, t, bad, badrow, badslot);
        }
    }
    __syncthreads();
#endif
    const uint32_t null_count = bp_sum(nulls, s_w);
    const uint32_t uq = s_cnt;
    const uint64_t tus = s_tus;
    const bool over = uq > limit;   // Dict is out; more than a third of the rows are distinct, so no value has a majority either
    const bool all_equal = !over && uq == 1;
    const uint32_t forb = a.forbidden | p.forb_extra;
    auto forbidden = [&](uint32_t cd) { return (forb >> cd) & 1u; };
    const double tuple_count = (double)N;
    // ---- the Freq majority (binary/freq.rs:147-169): Boyer-Moore vote over slot numbers, then the candidate's exact count
    uint32_t mc = 0;
    const bool want_mc = !forbidden(SB_CODEC_FREQ) && !all_equal && !over && !((double)null_count / tuple_count >= 0.9) &&
                         (uint64_t)uq * 10 <= (uint64_t)N + 10;   // (a 90 % majority leaves room for N / 10 other strings)
    if (want_mc) {
        s_x[t] = bm_c;
        s_y[t] = bm_n;
        __syncthreads();
        for (uint32_t stride = BP_WG / 2; stride > 0; stride >>= 1) {
            if (t < stride) {
                const uint32_t c0 = s_x[t], n0 = s_y[t], c1 = s_x[t + stride], n1 = s_y[t + stride];
                uint32_t cc = c0, nn = n0;
                if (n1) {
                    if (n0 == 0) {
                        cc = c1;
                        nn = n1;
                    } else if (c0 == c1) {
                        nn = n0 + n1;
                    } else if (n1 > n0) {
                        cc = c1;
                        nn = n1 - n0;
                    } else {
                        nn = n0 - n1;
                    }
                }
                s_x[t] = cc;
                s_y[t] = nn;
            }
            __syncthreads();
        }
        const uint32_t cand = s_x[0], cn = s_y[0];
        __syncthreads();
        uint32_t mine = 0;
        if (cn) {   // (eight independent loads per thread and step; the word behind an odd N's last row belongs to the area)
            const uint32_t npair = (N + 1) / 2;
            for (uint32_t k0 = t; k0 < npair; k0 += BP_WG * 8) {
                uint32_t pr[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const uint32_t k = k0 + (uint32_t)u * BP_WG;
                    pr[u] = gld32((const uint32_t*)slot16 + (k < npair ? k : 0));
                }
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const uint32_t k = k0 + (uint32_t)u * BP_WG;
                    if (k < npair) mine += ((pr[u] & 0xFFFFu) == cand) + ((pr[u] >> 16) == cand && 2 * k + 1 < N);
                }
            }
        }
        mc = bp_sum(mine, s_w);
    }
    STL(65);
    // ---- choose_compressor (binary/mod.rs:293-348), the arithmetic of choose_bin_impl
    const double total_bytes = (double)(c.values_len_total + ((uint64_t)N + 1) * sizeof(O));
    double max_ratio = a.ratio;
    uint32_t result = a.default_compression;
    static const uint8_t ORDER[3] = {SB_CODEC_ONEVALUE, SB_CODEC_FREQ, SB_CODEC_DICT};
    for (int oi = 0; oi < 3; oi++) {
        const uint32_t cd = ORDER[oi];
        if (forbidden(cd)) continue;
        double r = 0.0;
        if (cd == SB_CODEC_ONEVALUE) {
            r = all_equal ? tuple_count : 0.0;
        } else if (cd == SB_CODEC_FREQ) {
            if (!all_equal) {
                if ((double)null_count / tuple_count >= 0.9)
                    r = (double)(N - 1);
                else if ((double)mc / tuple_count >= 0.9)
                    r = (double)(N - 1);
            }
        } else {
            if (!over && (uint64_t)uq * 3 < N) {
                uint64_t after = tus + (uint64_t)N * (uint64_t)(bits_needed(uq) / 8);
                after += (uint64_t)N * 2 / 128;
                r = total_bytes / (double)after;
            }
        }
        if (r > max_ratio) {
            max_ratio = r;
            result = cd;
            if (r == tuple_count) break;
        }
    }
    const uint32_t codec = result;
    uint32_t D = 0;
    if (codec == SB_CODEC_DICT) {
        D = bp_ids_and_index(tab, BP_SLOTS, s_x, s_y, s_w, aux, slot16, vv, N);
    }
    uint32_t bp_bytes = 0, ent_bytes = 0, ent_word = 0, icodec = 0xFFFFFFFFu;   // icodec: the index block's codec when this kernel chose it
    if (codec == SB_CODEC_DICT) {
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");   // (firsts and idx, written above, are read below)
        const uint32_t* firsts = aux + BP_W_FIRSTS;
        const uint32_t* idx = aux + bh_table_slots(N) + 2 * (uint64_t)N;
        const uint32_t forb_n = forb | (1u << SB_CODEC_DICT);
        // ---- the codec of the index block: compress_integer::<u32> with Dict forbidden (binary/dict.rs:60-62 ->
        // integer/mod.rs:231-347), the arithmetic of choose_prim / decide_prim on what this kernel knows: every id occurs
        // (maximum D - 1, never negative, no nulls), all_equal and sortedness from one pass over the index array, the Freq
        // majority by a vote + count, the RLE / Bitpacking / DeltaBitpacking trials on the seeded samples (sb_select.h:
        // sample_row, sample_rle_runs, sample_bp_size) gathered out of the index array.
        STL(69);
        if (p.icodec < 0) icodec = bp_index_codec(a, p, idx, N, D, forb_n, s_x, s_y, s_w);
        STL(70);
        // ---- the index array bit-packed where the emitter's nested block will stand (integer/bp.rs:45-61; BitPacker4x: a
        // width byte, then 4 interleaved lanes per block of 128) when Bitpacking won: 32 768 indices per step through the
        // table's LDS.
        uint8_t* const blk = page_slot(a, c, p) + (c.nullable ? def_section_bytes(N) : 0);
        if (N % 128 == 0 && icodec == SB_CODEC_BITPACKING) bp_bytes = bp_pack_indices(blk + 18, idx, N, tab, s_x, s_w);
        __syncthreads();
        STL(71);
        // With the body in place everything behind it has its place too: the entries go straight into the page and this
        // kernel finishes it (headers, def levels, EncOut.pad = 1 like a page of the fused RLE selectors) —
        // k_enc_emit_pages<-4 / -8, Dict> is then not needed for the page (a fifth of C3's encode went there once, 8 % still
        // with its handed-over form).
        const bool whole = bp_bytes != 0;
        uint8_t* const ent_dst = blk + 9 + 9 + bp_bytes + 4;
        // ---- the dictionary's entries (u64 len | bytes, dictionary order: binary/dict.rs:84-93) in a staging area of the aux
        // words: the emitter places them behind the index block with one coalesced copy (its own pass — first row -> offsets
        // -> bytes for 20 entries per thread of a 256-thread workgroup — was a fifth of its time)
        {
            const uint64_t w0 = (bp_w_slot16(N) + ((uint64_t)N + 1) / 2 + 4 + 3) & ~3ull, M = bh_table_slots(N);
            if (D && (whole || w0 * 4 + tus + 64 <= M * 4)) {
                // (the table's LDS is free by now: entry sizes, then their offsets, one word per entry; entries are taken
                // lane = entry with the loads of four of them in flight)
                uint8_t* stage = whole ? ent_dst : (uint8_t*)(aux + w0);
                // (first row -> its offsets, eight entries in flight; with 32-bit offsets and at most 16 384 entries the entry's
                // source offset stays in the upper half of the table's LDS: the second pass then goes straight to the bytes —
                // three dependent gathers into scattered rows instead of five, 47 -> ~20 us of a page's 370)
                constexpr int EA = 8;
                const bool keep_b = sizeof(O) == 4 && D <= BP_SLOTS / 2;
                for (uint32_t i0 = t; i0 < D; i0 += BP_WG * EA) {
                    uint32_t r[EA];
                    uint64_t pr[EA], pb[EA];
#pragma unroll
                    for (int u = 0; u < EA; u++) r[u] = gld32(firsts + min(i0 + (uint32_t)u * BP_WG, D - 1));
#pragma unroll
                    for (int u = 0; u < EA; u++) {
                        if constexpr (sizeof(O) == 4) {
                            pr[u] = ldu64(offs + (uint64_t)r[u] * 4);
                            pb[u] = 0;
                        } else {
                            pb[u] = ldu64(offs + (uint64_t)r[u] * 8);
                            pr[u] = ldu64(offs + (uint64_t)r[u] * 8 + 8) - pb[u];
                        }
                    }
#pragma unroll
                    for (int u = 0; u < EA; u++) {
                        const uint32_t id = i0 + (uint32_t)u * BP_WG;
                        if (id >= D) continue;
                        const uint32_t L = sizeof(O) == 4 ? (uint32_t)(pr[u] >> 32) - (uint32_t)pr[u] : (uint32_t)pr[u];
                        tab[id] = L + 8;
                        if (keep_b) tab[BP_SLOTS / 2 + id] = (uint32_t)pr[u];
                    }
                }
                __syncthreads();
                const uint32_t per = (D + BP_WG - 1) / BP_WG, e0 = min(D, t * per), e1 = min(D, e0 + per);
                uint32_t mine = 0;
                for (uint32_t id = e0; id < e1; id++) mine += tab[id];
                const uint32_t incl = wave_incl_scan(mine);
                if (lane == 63) s_w[wv] = incl;
                __syncthreads();
                uint32_t at = incl - mine;
                for (uint32_t pw = 0; pw < wv; pw++) at += s_w[pw];
                uint32_t total = 0;
#pragma unroll
                for (int k = 0; k < BP_WG / 64; k++) total += s_w[k];
                for (uint32_t id = e0; id < e1; id++) {   // exclusive offsets in place
                    const uint32_t el = tab[id];
                    tab[id] = at;
                    at += el;
                }
                __syncthreads();
                constexpr int EU = 4;
                for (uint32_t i0 = t; i0 < D; i0 += BP_WG * EU) {
                    uint32_t r[EU], eo[EU], L[EU];
                    uint64_t b0[EU];
                    u32x4 v0[EU];
                    uint64_t v1[EU];
#pragma unroll
                    for (int u = 0; u < EU; u++) {
                        const uint32_t id = min(i0 + (uint32_t)u * BP_WG, D - 1);
                        r[u] = keep_b ? 0u : gld32(firsts + id);
                        eo[u] = tab[id];
                        L[u] = (id + 1 < D ? tab[id + 1] : total) - eo[u] - 8;
                        b0[u] = keep_b ? (uint64_t)tab[BP_SLOTS / 2 + id] : 0;
                    }
                    if (!keep_b) {
#pragma unroll
                        for (int u = 0; u < EU; u++) b0[u] = bk.beg(r[u]);
                    }
#pragma unroll
                    for (int u = 0; u < EU; u++) {   // (24 bytes when they are all there: the common short string)
                        v0[u] = u32x4{0, 0, 0, 0};
                        v1[u] = 0;
                        if (L[u] <= 24 && b0[u] + 24 <= vlen) {
                            v0[u] = ldu128(values + b0[u]);
                            v1[u] = ldu64(values + b0[u] + 16);
                        }
                    }
#pragma unroll
                    for (int u = 0; u < EU; u++) {
                        const uint32_t id = i0 + (uint32_t)u * BP_WG;
                        if (id >= D) continue;
                        uint8_t* d = stage + eo[u];
                        stu64(d, (uint64_t)L[u]);
                        uint32_t k = 0;
                        if (L[u] <= 24 && b0[u] + 24 <= vlen) {   // exact stores out of the registers (the next entry belongs to another lane)
                            const uint32_t w6[6] = {v0[u].x, v0[u].y, v0[u].z, v0[u].w, (uint32_t)v1[u], (uint32_t)(v1[u] >> 32)};
#pragma unroll
                            for (uint32_t q = 0; q < 6; q++) {
                                if (4 * q + 4 <= L[u]) stu32(d + 8 + 4 * q, w6[q]);
                                else if (4 * q < L[u])
                                    for (uint32_t bb = 0; 4 * q + bb < L[u]; bb++) *(gptr)(d + 8 + 4 * q + bb) = (uint8_t)(w6[q] >> (8 * bb));
                            }
                            continue;
                        }
                        for (; k + 16 <= L[u]; k += 16) stu128(d + 8 + k, ldu128(values + b0[u] + k));
                        for (; k < L[u]; k++) *(gptr)(d + 8 + k) = ldu8(values + b0[u] + k);
                    }
                }
                ent_bytes = total;
                ent_word = (uint32_t)w0;
                __syncthreads();
            }
        }
        STL(72);
        if (whole && ent_bytes) {
            const uint64_t pos = c.nullable ? def_section_bytes(N) : 0;
            if (c.nullable && t < WG) {   // (def_bits_page walks with the page kernels' 256 threads)
                uint8_t* bits = def_header(blk - pos, N);
                def_bits_page(bits, ValidView{c.validity, c.validity_bit_offset}, p.row0, N, c.rows);
            }
            if (t == 0) {
                const uint64_t ib = 9 + (uint64_t)bp_bytes, body = ib + 4 + ent_bytes;
                put_hdr9(blk + 9, SB_CODEC_BITPACKING, bp_bytes, (uint32_t)(N * 4));
                stu32(blk + 9 + ib, D);
                put_hdr9(blk, SB_CODEC_DICT, (uint32_t)body, (uint32_t)c.values_len_total);   // (binary/mod.rs:88: the whole shared buffer)
                EncOut o;
                o.length = pos + 9 + body;
                o.out_off = 0;
                o.slot = blk - pos;
                o.codec = SB_CODEC_DICT;
                o.pad = 1;
                a.outs[page] = o;
            }
        }
    }
    STL(68);
    if (t == 0) {
        gst32(aux + BH_W_ICODEC, icodec + 1u);
        gst32(aux + BH_W_BPBYTES, bp_bytes);
        gst32(aux + BH_W_ENTBYTES, a.outs[page].pad == 1 ? 0u : ent_bytes);   // (a page finished here has nothing staged)
        gst32(aux + BH_W_ENTWORD, ent_word);
        gst32(aux + BH_W_D, D);
        gst32(aux + BH_W_BAD, 0u);
        gst32(aux + BH_W_MAGIC, codec == SB_CODEC_DICT ? BH_MAGIC2 : 0u);
        gst32(aux + BH_W_FUSED, BP_DONE);
        a.codecs[page] = (int32_t)codec;
        atomicAdd(&a.codec_counts[codec & 31], 1u);
        if (!has_device_encoder(codec))
            raise(a.status, SB_ERR_NYI, page, 700 + codec);
        else if (codec == SB_CODEC_FREQ)
            atomicAdd(a.freq_count, 1u);
    }
}

// ---- Dict pages of 1- / 2- / 4-byte INTEGERS (the selector chose Dict; integer/dict.rs:33-73): the dictionary, the index
// array, the index block's codec and its bit-packed body by one workgroup of 1024 threads, as for binary pages — the page
// kernel (k_enc_emit_pages<W, Dict>: LDS / HBM table, choose_prim over the indices, enc_bp — one 256-thread workgroup, two
// per CU, 0.5 ms for C4's 306 Int32 pages whatever the load) then writes headers and the dictionary's values.  The table's
// slot holds the key itself (key32 | row16 in a u64: exact, no gathers); only keyed rows are entered (row 0 and the valid
// rows: a leading null interns T::default(), dict.rs:46-50).  A page with more than PD_CAP distinct values is left to the
// page kernel.  Floats stay there too (a NaN equals nothing: every NaN row is an entry of its own).
template <int W>
__global__ void __launch_bounds__(BP_WG) k_enc_prim_dict(EncodeArgs a) {
    static_assert(W == 1 || W == 2 || W == 4, "keys of up to 32 bits");
    __shared__ __attribute__((aligned(16))) uint32_t tab[BP_SLOTS];   // PD_SLOTS u64 slots, later PD_SLOTS u32 words (bp_ids_and_index)
    __shared__ uint32_t s_x[2048];
    __shared__ uint32_t s_y[2048];
    __shared__ uint32_t s_w[BP_WG / 64];
    __shared__ uint32_t s_cnt;
    if (a.use_counts && a.codec_counts[SB_CODEC_DICT] == 0) return;
    const uint32_t t = threadIdx.x;
    const uint32_t page = spread_block(blockIdx.x, gridDim.x) + a.page_base;
    const EncPage p = get_page(a, page);
    if (codec_of(a, p, page) != SB_CODEC_DICT) return;
    const EncCol c = get_col(a, p.col);
    if (!pd_page_ok(a, p, page, c, W)) return;
    if (a.outs[page].pad == 1 && a.outs[page].length != 0) return;   // (already written)
    const uint32_t N = (uint32_t)p.rows;
    uint32_t* aux = (uint32_t*)(a.scratch + p.aux_off);
    uint16_t* slot16 = (uint16_t*)(aux + bp_w_slot16(N));
    const uint8_t* vals = c.values + p.row0 * W;
    const ValidView vv{c.validity, c.validity_bit_offset + p.row0};
    unsigned long long* t64 = (unsigned long long*)tab;
    for (uint32_t i = t; i < PD_SLOTS; i += BP_WG) t64[i] = ~0ull;
    if (t == 0) {
        s_cnt = 0;
        gst32(aux + BH_W_FUSED, 0u);   // (a stale mark of an earlier call must not survive a page this kernel gives up on)
        gst32(aux + BH_W_MAGIC, 0u);
    }
    __syncthreads();
    constexpr int U = 8;
    for (uint32_t base = 0; base < N; base += BP_WG * U) {
        if (__hip_atomic_load(&s_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) > PD_CAP) break;
        uint32_t key[U], vb[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t i = base + (uint32_t)u * BP_WG + t, ic = i < N ? i : N - 1;
            if constexpr (W == 4) key[u] = ldu32(vals + (uint64_t)ic * 4);
            else if constexpr (W == 2) key[u] = ldu16(vals + (uint64_t)ic * 2);
            else key[u] = ldu8(vals + ic);
            vb[u] = vv.bits ? (uint32_t)ldu8(vv.bits + ((vv.off + ic) >> 3)) : 0xFFu;
        }
        uint32_t newk = 0;
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t i = base + (uint32_t)u * BP_WG + t, ic = i < N ? i : N - 1;
            if (i >= N) continue;
            const bool valid = (vb[u] >> ((vv.off + ic) & 7)) & 1;
            uint32_t sl = 0;
            if (valid || i == 0) {
                const uint32_t k = valid ? key[u] : 0u;   // (a leading null interns the default value)
                uint32_t h = k * 0x9E3779B1u;
                h ^= h >> 15;
                h *= 0x85EBCA6Bu;
                h ^= h >> 13;
                sl = h & (PD_SLOTS - 1);
                const unsigned long long wd = ((unsigned long long)k << 32) | (i & 0xFFFFu);
                for (;;) {
                    unsigned long long cur = t64[sl];
                    if (cur == ~0ull) {
                        cur = atomicCAS(&t64[sl], ~0ull, wd);
                        if (cur == ~0ull) {
                            newk++;
                            break;
                        }
                    }
                    if ((uint32_t)(cur >> 32) == k && (uint32_t)(cur >> 16 & 0xFFFFu) == 0) {
                        if (wd < cur) atomicMin(&t64[sl], wd);
                        break;
                    }
                    sl = (sl + 1) & (PD_SLOTS - 1);
                }
            }
            *(__attribute__((address_space(1))) uint16_t*)(slot16 + i) = (uint16_t)sl;
        }
        if (newk) atomicAdd(&s_cnt, newk);
    }
    __syncthreads();
    if (s_cnt > PD_CAP) return;   // (the page kernel builds this one)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    {   // the u64 slots -> the u32 words bp_ids_and_index reads (keyed, row16), in place: everything is read first
        unsigned long long cur[PD_SLOTS / BP_WG];
#pragma unroll
        for (uint32_t q = 0; q < PD_SLOTS / BP_WG; q++) cur[q] = t64[t + q * BP_WG];
        __syncthreads();
#pragma unroll
        for (uint32_t q = 0; q < PD_SLOTS / BP_WG; q++) tab[t + q * BP_WG] = cur[q] == ~0ull ? BP_EMPTY : (uint32_t)(cur[q] & 0xFFFFu);
        __syncthreads();
    }
    const uint32_t D = bp_ids_and_index(tab, PD_SLOTS, s_x, s_y, s_w, aux, slot16, vv, N);
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const uint32_t* idx = aux + bh_table_slots(N) + 2 * (uint64_t)N;
    const uint32_t forb_n = a.forbidden | p.forb_extra | (1u << SB_CODEC_DICT);
    const uint32_t icodec = bp_index_codec(a, p, idx, N, D, forb_n, s_x, s_y, s_w);
    uint32_t bp_bytes = 0;
    if (N % 128 == 0 && icodec == SB_CODEC_BITPACKING)
        bp_bytes = bp_pack_indices(page_slot(a, c, p) + (c.nullable ? def_section_bytes(N) : 0) + 18, idx, N, tab, s_x, s_w);
    if (t == 0) {
        gst32(aux + BH_W_ICODEC, icodec + 1u);
        gst32(aux + BH_W_BPBYTES, bp_bytes);
        gst32(aux + BH_W_ENTBYTES, 0u);
        gst32(aux + BH_W_D, D);
        gst32(aux + BH_W_BAD, 0u);
        gst32(aux + BH_W_MAGIC, BH_MAGIC2);
        gst32(aux + BH_W_FUSED, PD_DONE);
    }
}
