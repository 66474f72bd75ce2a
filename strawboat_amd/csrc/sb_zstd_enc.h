// strawboat-hip: Zstandard frame ENCODER on the device (RFC 8878), codec id 2 — replaces zstd::bulk::compress_to_buffer
// (libzstd, level 3) at the reference call site src/compression/basic.rs:122-135.  Format-valid, not libzstd's bytes
// (BASELINE.md §6): any Zstd decoder — the reference's, libzstd's, sb_zstd.h — reads the frame back to the input.
//
// One wave64 per page sub-buffer, one frame, blocks of up to 128 KiB of content:
//   * sequences from the LDS matcher the LZ4 encoder uses (sb_lz4.h LzMatcher: 64 probe positions per step, lazy
//     selection, history across blocks), recorded as (literal length, match length, offset) + a literal buffer in HBM;
//   * literals: Huffman coded when the block's literals use byte values 0..128 only (the tree then fits the direct
//     4-bit weight description; text, small integers) — lengths from a two-queue Huffman build limited to 11 bits,
//     canonical codes as libzstd assigns them, four streams whose bits are placed lane-parallel (16 symbols per lane,
//     prefix sums of code lengths, ds_or into an LDS window); raw literals otherwise;
//   * sequences: FSE with the predefined distributions (RFC 8878 §3.1.1.3.2.2) — the encoding tables are built once
//     per wave in LDS; the state chain is serial by construction and runs on lane 0;
//   * a block that does not get smaller is stored raw.
#pragma once
#include "sb_lz4.h"

namespace sb {

constexpr uint32_t ZE_BLOCK = 128 * 1024;
#ifndef ZE_PROBE
#define ZE_PROBE 2048u   // bytes of a piece parsed before deciding whether sequences pay (ze_block); 1024: C5 write 157 -> 109 GB/s (pieces that should be literals-only take the full parse)
#endif
constexpr uint32_t ZE_HUF_MAXBITS = 11;
constexpr uint32_t ZE_LITONLY_MAXBITS = 9;   // == ZH_MAXBITS (sb_zstd.h): literals-only pieces are read lane per stream

struct ZeSeq {       // 8 bytes per sequence in HBM
    uint32_t ll;
    uint16_t ml;     // match length (>= 4)
    uint16_t off;    // distance (1 .. 65535)
};
struct ZeSymTT {
    int32_t delta_nb_bits;
    int32_t delta_find_state;
};
// Matcher geometry: 2^11 table entries, 8 KiB ring.  LDS per wave is the encoder's occupancy (one wave per 16 KiB piece):
// 28 KB allowed 5 waves per CU, 20 KB allow 8 (arrays that are never live together share their bytes).
constexpr int ZE_HB = 11, ZE_RB = 13;
struct ZEncLds {
    Lz4EncLds<ZE_HB, ZE_RB> lz;      // matcher; lz.out doubles as the bit window of the Huffman streams, tab + ring as the tree builder's scratch
    union {
        uint32_t hist[256];          // byte histogram of the literals ...
        uint32_t h_cnt[256];         // ... which is dead once the tree is described: code | length << 16 per symbol for the stream packer
    };
    uint16_t hcode[256];
    uint8_t hlen[256];
    uint16_t ll_st[64], ml_st[64], of_st[32];   // FSE state tables (predefined distributions)
    ZeSymTT ll_tt[36], ml_tt[53], of_tt[29];
    uint32_t misc[8];
    union {
        // one batch of sequences, prepared by all lanes for lane 0: codes + extra-bit values + the FSE transforms of the codes
        struct {
            uint32_t sq_code[64], sq_ll[64], sq_ml[64], sq_of[64];
            ZeSymTT sq_tt[64][3];
        };
        // the literals section comes before the sequences section: its scratch shares the batch area (behind sq_code / sq_ll /
        // sq_ml, which the FSE coder of the weights borrows as spread table, cumulative counts and weight histogram)
        struct {
            uint32_t lit_pad_[192];
            uint16_t h_sorted[256];          // symbols by (count, symbol)
            uint16_t w_st[64];               // FSE coding of the Huffman weights (alphabets of more than 128 symbols: RFC 8878 4.2.1.2)
            ZeSymTT w_tt[13];
            int16_t w_norm[13];
            uint8_t w_val[256];
        };
    };
};

__device__ const uint8_t ZE_LL_BITS[36] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1,
                                           1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
__device__ const uint8_t ZE_ML_BITS[53] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                           0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
__device__ const uint8_t ZE_LL_CODE[64] = {0,  1,  2,  3,  4,  5,  6,  7,  8,  9,  10, 11, 12, 13, 14, 15, 16, 16, 17, 17, 18, 18,
                                           19, 19, 20, 20, 20, 20, 21, 21, 21, 21, 22, 22, 22, 22, 22, 22, 22, 22, 23, 23, 23, 23,
                                           23, 23, 23, 23, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24};
__device__ const uint8_t ZE_ML_CODE[128] = {
    0,  1,  2,  3,  4,  5,  6,  7,  8,  9,  10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31,
    32, 32, 33, 33, 34, 34, 35, 35, 36, 36, 36, 36, 37, 37, 37, 37, 38, 38, 38, 38, 38, 38, 38, 38, 39, 39, 39, 39, 39, 39, 39, 39,
    40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 41, 41, 41, 41, 41, 41, 41, 41, 41, 41, 41, 41, 41, 41, 41, 41,
    42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42};

__device__ __forceinline__ uint32_t ze_highbit(uint32_t v) { return 31u - (uint32_t)__clz((int)v); }
__device__ __forceinline__ uint32_t ze_ll_code(uint32_t ll) { return ll > 63 ? ze_highbit(ll) + 19 : ZE_LL_CODE[ll]; }
__device__ __forceinline__ uint32_t ze_ml_code(uint32_t mlbase) { return mlbase > 127 ? ze_highbit(mlbase) + 36 : ZE_ML_CODE[mlbase]; }

// FSE_buildCTable for a normalized distribution (executed by one lane)
__device__ inline void ze_build_ctable(const int16_t* norm, int nsym, int log, uint16_t* state_table, ZeSymTT* tt, uint8_t* spread /* 1 << log */,
                                       int* cumul /* nsym + 1 words of LDS: a private array indexed at run time would live in scratch */) {
    const int size = 1 << log, mask = size - 1, step = (size >> 1) + (size >> 3) + 3;
    int high = size - 1;
    cumul[0] = 0;
    for (int s = 0; s < nsym; s++) {
        if (norm[s] == -1) {
            cumul[s + 1] = cumul[s] + 1;
            spread[high--] = (uint8_t)s;
        } else {
            cumul[s + 1] = cumul[s] + norm[s];
        }
    }
    int pos = 0;
    for (int s = 0; s < nsym; s++) {
        for (int k = 0; k < norm[s]; k++) {
            spread[pos] = (uint8_t)s;
            pos = (pos + step) & mask;
            while (pos > high) pos = (pos + step) & mask;
        }
    }
    for (int u = 0; u < size; u++) {
        const int s = spread[u];
        state_table[cumul[s]++] = (uint16_t)(size + u);
    }
    int total = 0;
    for (int s = 0; s < nsym; s++) {
        const int c = norm[s];
        if (c == 0) {
            tt[s].delta_nb_bits = ((log + 1) << 16) - (1 << log);
            tt[s].delta_find_state = 0;
        } else if (c == -1 || c == 1) {
            tt[s].delta_nb_bits = (log << 16) - (1 << log);
            tt[s].delta_find_state = total - 1;
            total++;
        } else {
            const int max_bits_out = log - (int)ze_highbit((uint32_t)(c - 1));
            const int min_state_plus = c << max_bits_out;
            tt[s].delta_nb_bits = (max_bits_out << 16) - min_state_plus;
            tt[s].delta_find_state = total - c;
            total += c;
        }
    }
}

// forward-growing bit writer used by lane 0 (sequences section): bits are added low to high, bytes flushed to HBM
struct ZeBits {
    uint8_t* p;
    uint64_t acc;
    uint32_t nb;
    __device__ __forceinline__ void add(uint32_t v, uint32_t n) {
        acc |= (uint64_t)(v & ((1u << n) - 1)) << nb;   // (n <= 25)
        nb += n;
    }
    __device__ __forceinline__ void flush() {   // one unaligned 8-byte store; the bytes above the whole ones are rewritten later
        stu64(p, acc);
        p += nb >> 3;
        acc = (nb & ~7u) >= 64 ? 0 : acc >> (nb & ~7u);
        nb &= 7;
    }
    __device__ __forceinline__ uint8_t* close() {   // end mark, then the partial byte
        add(1, 1);
        flush();
        if (nb) *p++ = (uint8_t)acc;
        return p;
    }
};

// Huffman codes of the literals: a complete prefix code of at most max_bits bits for the symbols with hist[s] > 0, canonical
// codes (libzstd's order) in Z.hcode, lengths in Z.hlen; returns the longest length (0: no tree, the caller stores raw
// literals).  The lengths come from PACKAGE-MERGE run by the whole wave (optimal for the length limit; the round-2 builder
// — two-queue merge, depth walk and Kraft repair on one lane — cost ~1 M cycles for a 256-symbol alphabet, this ~40 k).
//   lists   level max_bits holds the leaves (weights ascending); level l = merge(leaves, packages of level l + 1), a package
//           being the sum of two neighbours.  Every leaf / package finds its place by a binary search in the other list
//           (9 steps, the four leaves and four packages of a lane side by side); PK[l][i] keeps the packages in front of leaf i.
//   lengths top-down: the first 2n - 2 items of level 1 are taken; the packages among them open twice as many items of the
//           level below; a leaf's code length is the number of levels in which it is among the items taken.
// `sc`: 1536 + 64 * (max_bits + 1) words of LDS scratch (the matcher's table and ring once the block's parse is over).
static_assert(sizeof(((Lz4EncLds<ZE_HB, ZE_RB>*)nullptr)->tab) + sizeof(((Lz4EncLds<ZE_HB, ZE_RB>*)nullptr)->ring) >= 4 * (1536 + 64 * (ZE_HUF_MAXBITS + 1)),
              "sb_zstd_enc.h: the tree builder's scratch must fit the matcher's table + ring");
__device__ inline uint32_t ze_huf_build_pm(ZEncLds& Z, uint32_t max_sym, uint32_t max_bits, uint32_t* sc) {
    const uint32_t lane = threadIdx.x & 63;
    uint32_t* W = sc;            // [256] leaf weights, ascending
    uint32_t* P = sc + 256;      // [256] packages of the current level
    uint32_t* LA = sc + 512;     // [512] lists (ping-pong)
    uint32_t* LB = sc + 1024;
    uint8_t* PK = (uint8_t*)(sc + 1536);   // [max_bits + 1][256]
    // order of the present symbols by (count, symbol): rank = keys below mine
    uint32_t mykey[4], myrank[4] = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t sy = lane + 64 * j;
        const uint32_t c = sy <= max_sym ? Z.hist[sy] : 0u;
        mykey[j] = c ? (c << 8) | sy : 0xFFFFFFFFu;
        LA[sy] = mykey[j];
        Z.hlen[sy] = 0;
    }
    uint32_t ns = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) ns += (uint32_t)__popcll(__ballot(mykey[j] != 0xFFFFFFFFu));
    wave_sync();
    if (ns < 2) return 0;
    for (uint32_t o = 0; o <= max_sym; o += 4) {   // (uniform addresses: broadcast reads)
        const uint32_t k0 = LA[o], k1 = LA[o + 1], k2 = LA[o + 2], k3 = LA[o + 3];
#pragma unroll
        for (int j = 0; j < 4; j++) myrank[j] += (k0 < mykey[j]) + (k1 < mykey[j]) + (k2 < mykey[j]) + (k3 < mykey[j]);
    }
    wave_sync();
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (mykey[j] != 0xFFFFFFFFu) {
            Z.h_sorted[myrank[j]] = (uint16_t)(lane + 64 * j);
            W[myrank[j]] = mykey[j] >> 8;
        }
    }
    wave_sync();
    uint32_t w[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t i = lane + 64 * j;
        w[j] = i < ns ? W[i] : 0xFFFFFFFFu;
        if (i < ns) {
            LA[i] = w[j];
            PK[max_bits * 256 + i] = 0;
        }
    }
    wave_sync();
    uint32_t* cur = LA;
    uint32_t* nxt = LB;
    uint32_t len = ns;
    const uint32_t keep = 2 * ns - 2;
    for (uint32_t l = max_bits - 1; l >= 1; l--) {
        const uint32_t np = min(len / 2, ns - 1);
        uint32_t pj[4], bl[4] = {0, 0, 0, 0}, bp[4] = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t jj = lane + 64 * j;
            pj[j] = jj < np ? cur[2 * jj] + cur[2 * jj + 1] : 0xFFFFFFFFu;
            if (jj < np) P[jj] = pj[j];
        }
        wave_sync();
        // bl = packages below my leaf (ties: the leaf first), bp = leaves at or below my package
        for (uint32_t step = 256; step; step >>= 1) {
            uint32_t vl[4], vp[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                vl[j] = bl[j] + step <= np ? P[bl[j] + step - 1] : 0xFFFFFFFFu;
                vp[j] = bp[j] + step <= ns ? W[bp[j] + step - 1] : 0xFFFFFFFFu;
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (bl[j] + step <= np && vl[j] < w[j]) bl[j] += step;
                if (bp[j] + step <= ns && vp[j] <= pj[j]) bp[j] += step;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t i = lane + 64 * j;
            if (i < ns) {
                PK[l * 256 + i] = (uint8_t)bl[j];
                if (i + bl[j] < keep) nxt[i + bl[j]] = w[j];
            }
            if (i < np && i + bp[j] < keep) nxt[i + bp[j]] = pj[j];
        }
        wave_sync();
        len = min(ns + np, keep);
        uint32_t* tsw = cur;
        cur = nxt;
        nxt = tsw;
    }
    uint32_t k = keep, mylen[4] = {0, 0, 0, 0};
    for (uint32_t l = 1; l <= max_bits && k; l++) {
        uint32_t a = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t i = lane + 64 * j;
            const bool in = i < ns && i + PK[l * 256 + i] < k;
            a += (uint32_t)__popcll(__ballot(in));
            if (in) mylen[j]++;
        }
        k = 2 * (k - a);
    }
    uint32_t maxd = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t i = lane + 64 * j;
        if (i < ns) {
            Z.hlen[Z.h_sorted[i]] = (uint8_t)mylen[j];
            maxd = max(maxd, mylen[j]);
        }
    }
    for (int d = 32; d > 0; d >>= 1) maxd = max(maxd, (uint32_t)__shfl_xor((int)maxd, d, 64));
    wave_sync();
    if (!maxd || maxd > max_bits) return 0;
    // canonical codes: longest codes first, symbols of one length in symbol order
    uint32_t ml[4];
#pragma unroll
    for (int j = 0; j < 4; j++) ml[j] = Z.hlen[(lane + 64 * j) & 255];
    uint32_t code = 0;
    for (uint32_t ln = maxd; ln >= 1; ln--) {
        uint32_t before = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint64_t m = __ballot(ml[j] == ln && lane + 64 * j <= max_sym);
            if (ml[j] == ln && lane + 64 * j <= max_sym) Z.hcode[lane + 64 * j] = (uint16_t)(code + before + lane_rank(m));
            before += (uint32_t)__popcll(m);
        }
        code = (code + before) >> 1;
    }
    wave_sync();
    return maxd;
}

// weights of the symbols 0 .. max_sym - 1 into Z.w_val and their histogram into Z.sq_ml[0 .. 12] (the whole wave)
__device__ inline void ze_huf_weights_prep(ZEncLds& Z, uint32_t max_sym, uint32_t hbits) {
    const uint32_t lane = threadIdx.x & 63;
    uint32_t wv[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t sy = lane + 64 * j;
        wv[j] = 0xFFu;
        if (sy < max_sym) {
            wv[j] = Z.hlen[sy] ? hbits + 1 - Z.hlen[sy] : 0u;
            Z.w_val[sy] = (uint8_t)wv[j];
        }
    }
    for (uint32_t w = 0; w < 13; w++) {
        uint32_t c = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) c += (uint32_t)__popcll(__ballot(wv[j] == w));
        if (lane == 0) Z.sq_ml[w] = c;
    }
    wave_sync();
}
// The tree description for more than 128 symbols: the weights of symbols 0 .. max_sym - 1, FSE-coded with two interleaved
// states (RFC 8878 4.2.1.2; HUF_compressWeights upstream).  Lane 0 only; returns the bytes written at dst (header byte =
// compressed size < 128, table description, stream), 0 when the weights do not compress into that.
__device__ inline uint32_t ze_huf_weights_fse(ZEncLds& Z, uint32_t max_sym, uint32_t hbits, uint8_t* dst) {
    const uint32_t n = max_sym;   // weights written (the last symbol's weight is implied)
    if (n < 2) return 0;
    // (Z.w_val and the counts of the 13 possible weights, Z.sq_ml[0 .. 12], come from ze_huf_weights_prep, run by the wave)
    const uint32_t* cnt = Z.sq_ml;
    uint32_t maxw = 0;
    for (uint32_t w = 0; w < 13; w++)
        if (cnt[w]) maxw = w;
    for (uint32_t w = 0; w <= maxw; w++)
        if (cnt[w] == n) return 0;     // one weight only: not representable this way
    // table log: at most 6, and small enough for the number of weights (FSE_optimalTableLog)
    uint32_t log = 6;
    while (log > 5 && (1u << log) > n) log--;
    const uint32_t size = 1u << log;
    // normalisation: proportional, every present weight at least 1, the largest takes the rounding
    uint32_t tot = 0, big = 0;
    for (uint32_t w = 0; w <= maxw; w++) {
        uint32_t v = 0;
        if (cnt[w]) v = max(1u, (uint32_t)(((uint64_t)cnt[w] * size + n / 2) / n));
        Z.w_norm[w] = (int16_t)v;
        tot += v;
        if (cnt[w] > cnt[big]) big = w;
    }
    if (tot != size) {
        const int32_t fix = (int32_t)size - (int32_t)tot;
        if ((int32_t)Z.w_norm[big] + fix < 1) return 0;
        Z.w_norm[big] = (int16_t)((int32_t)Z.w_norm[big] + fix);
    }
    // table description (FSE_writeNCount; the inverse of z_fse_header)
    uint8_t* out = dst + 1;
    {
        uint64_t acc = 0;
        uint32_t nb = 0;
        uint8_t* q = out;
        auto put = [&](uint32_t v, uint32_t bits) {
            acc |= (uint64_t)v << nb;
            nb += bits;
            while (nb >= 8) {
                *q++ = (uint8_t)acc;
                acc >>= 8;
                nb -= 8;
            }
        };
        put(log - 5, 4);
        int32_t remaining = (int32_t)size + 1, threshold = (int32_t)size, nbits = (int32_t)log + 1;
        uint32_t sy = 0;
        bool prev0 = false;
        while (remaining > 1 && sy <= maxw) {
            if (prev0) {   // run of zero probabilities: 2-bit repeat counts
                uint32_t z = 0;
                while (sy + z <= maxw && Z.w_norm[sy + z] == 0) z++;
                uint32_t r = z;
                while (r >= 3) {
                    put(3, 2);
                    r -= 3;
                }
                put(r, 2);
                sy += z;
                prev0 = false;
                continue;
            }
            const int32_t count = Z.w_norm[sy];
            const int32_t mx = (2 * threshold - 1) - remaining;
            remaining -= count < 0 ? -count : count;
            int32_t c = count + 1;
            if (c >= threshold) c += mx;
            put((uint32_t)c, (uint32_t)(nbits - (c < mx ? 1 : 0)));
            sy++;
            prev0 = count == 0;
            while (remaining < threshold) {
                nbits--;
                threshold >>= 1;
            }
        }
        if (remaining != 1) return 0;
        if (nb) *q++ = (uint8_t)acc;
        out = q;
    }
    // encoding table, then the weights from the last to the first (FSE_compress_usingCTable)
    ze_build_ctable(Z.w_norm, (int)maxw + 1, (int)log, Z.w_st, Z.w_tt, (uint8_t*)Z.sq_code, (int*)Z.sq_ll);
    auto init_state = [&](uint32_t sym) -> uint32_t {
        const uint32_t nbo = (uint32_t)(Z.w_tt[sym].delta_nb_bits + (1 << 15)) >> 16;
        const uint32_t value = (nbo << 16) - (uint32_t)Z.w_tt[sym].delta_nb_bits;
        return Z.w_st[(value >> nbo) + Z.w_tt[sym].delta_find_state];
    };
    ZeBits bs{out, 0, 0};
    auto enc = [&](uint32_t& st, uint32_t sym) {
        const uint32_t nbo = (uint32_t)(st + Z.w_tt[sym].delta_nb_bits) >> 16;
        bs.add(st, nbo);
        bs.flush();
        st = Z.w_st[(st >> nbo) + Z.w_tt[sym].delta_find_state];
    };
    uint32_t ip = n, s1, s2;
    if (n & 1) {
        s1 = init_state(Z.w_val[--ip]);
        s2 = init_state(Z.w_val[--ip]);
        enc(s1, Z.w_val[--ip]);
    } else {
        s2 = init_state(Z.w_val[--ip]);
        s1 = init_state(Z.w_val[--ip]);
    }
    // An even number of weights is left: pairs (one for each state).  A weight's table entry does not depend on the states,
    // so the next pair's entries are read while this pair's two next-state reads are in flight — the chain per pair is one
    // LDS round trip, not three per weight (the loop was 40 % of a 16 KiB piece's time).
    ZeSymTT ta = {0, 0}, tb = {0, 0};
    if (ip >= 2) {
        ta = Z.w_tt[Z.w_val[ip - 1]];
        tb = Z.w_tt[Z.w_val[ip - 2]];
    }
    while (ip >= 2) {
        ZeSymTT tc = {0, 0}, td = {0, 0};
        if (ip >= 4) {
            tc = Z.w_tt[Z.w_val[ip - 3]];
            td = Z.w_tt[Z.w_val[ip - 4]];
        }
        const uint32_t na = (uint32_t)(s2 + ta.delta_nb_bits) >> 16, nb2 = (uint32_t)(s1 + tb.delta_nb_bits) >> 16;
        const uint32_t n2 = Z.w_st[(s2 >> na) + ta.delta_find_state], n1 = Z.w_st[(s1 >> nb2) + tb.delta_find_state];
        bs.add(s2, na);
        bs.add(s1, nb2);
        bs.flush();
        s2 = n2;
        s1 = n1;
        ta = tc;
        tb = td;
        ip -= 2;
    }
    bs.add(s2, log);
    bs.flush();
    bs.add(s1, log);
    bs.flush();
    uint8_t* e = bs.close();
    const uint32_t csize = (uint32_t)(e - (dst + 1));
    if (csize >= 128 || csize < 2) return 0;
    dst[0] = (uint8_t)csize;
    return 1 + csize;
}

// One Huffman stream: symbols lits[0, m) in REVERSE order into a forward bit stream + end mark.  16 symbols per lane,
// positions from prefix sums of the code lengths, bits OR-ed into an LDS window (Z.lz.out) tile by tile.
// Returns the stream's size in bytes.
__device__ inline uint32_t ze_huf_stream(ZEncLds& Z, const uint8_t* lits, uint32_t m, uint8_t* dst) {
    const uint32_t lane = threadIdx.x & 63;
    uint32_t* win = (uint32_t*)Z.lz.out;        // 2048 bytes = 512 words; a tile of 1024 symbols needs <= 11264 bits = 352 words
    const uint32_t* hct = Z.h_cnt;               // code | length << 16 per symbol (ze_huf_pack_table)
    uint32_t out = 0;                            // whole bytes already written to dst
    uint32_t carry_bits = 0, carry = 0;          // bits of the unfinished byte (< 8), kept in `carry`
    for (uint32_t done = 0; done <= m; done += 1024) {   // (one extra round when m % 1024 == 0 writes the end mark)
        const uint32_t tile = min(1024u, m - done);
        for (uint32_t k = lane; k < 384; k += 64) win[k] = 0;
        wave_sync();
        // my symbols: reverse positions [done + 16 lane, done + 16 lane + 16) -> lits[m - 1 - r]: the 16 bytes in front of
        // lits + m - done - 16 lane, last byte first
        uint32_t sym[16];
        const uint32_t mine = 16 * lane < tile ? min(16u, tile - 16 * lane) : 0u;
        if (mine == 16) {
            const u32x4 v = ldu128(lits + (m - done - 16 * lane - 16));
            const uint32_t wd[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 16; j++) sym[j] = (wd[3 - (j >> 2)] >> (8 * (3 - (j & 3)))) & 0xFF;
        } else {
#pragma unroll
            for (int j = 0; j < 16; j++) sym[j] = (uint32_t)j < mine ? (uint32_t)ldu8(lits + (m - 1 - (done + 16 * lane + j))) : 0u;
        }
        // the lane's codes back to back in a 192-bit register buffer (16 codes of <= 11 bits)
        uint64_t a0 = 0, a1 = 0, a2 = 0;
        uint32_t bits = 0;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const uint32_t e = (uint32_t)j < mine ? hct[sym[j]] : 0u;
            const uint64_t c = e & 0xFFFFu;
            const uint32_t l = e >> 16, sh = bits & 63, wd = bits >> 6;
            const uint64_t lo = c << sh, hi = sh ? c >> (64 - sh) : 0ull;   // (hi != 0 only when the code crosses a 64-bit border)
            if (wd == 0) {
                a0 |= lo;
                a1 |= hi;
            } else if (wd == 1) {
                a1 |= lo;
                a2 |= hi;
            } else {
                a2 |= lo;
            }
            bits += l;
        }
        const uint32_t incl = wave_scan_dpp(bits);
        const uint32_t pos = carry_bits + incl - bits;
        if (lane == 0 && carry_bits) atomicOr(&win[0], carry);
        if (bits) {   // the buffer, shifted to its bit position, word by word (<= 7 words)
            const uint32_t w0 = pos >> 5, sh = pos & 31;
            const uint32_t b32[6] = {(uint32_t)a0, (uint32_t)(a0 >> 32), (uint32_t)a1, (uint32_t)(a1 >> 32), (uint32_t)a2, (uint32_t)(a2 >> 32)};
            const uint32_t nw = (sh + bits + 31) >> 5;
            uint32_t prev = 0;
#pragma unroll
            for (int k = 0; k < 7; k++) {
                const uint32_t cur = k < 6 ? b32[k] : 0u;
                const uint32_t v = sh ? (cur << sh) | (prev >> (32 - sh)) : cur;
                if ((uint32_t)k < nw && v) atomicOr(&win[w0 + k], v);
                prev = cur;
            }
        }
        uint32_t total = carry_bits + rdlane(incl, 63);
        const bool last = done + tile >= m;
        if (last) {
            if (lane == 0) atomicOr(&win[total >> 5], 1u << (total & 31));   // end mark
            total += 1;
        }
        wave_sync();
        const uint32_t nbytes = last ? (total + 7) >> 3 : total >> 3;
        const uint8_t* wb = (const uint8_t*)win;
        for (uint32_t k = lane; 4 * k + 4 <= nbytes; k += 64) stu32(dst + out + 4 * k, win[k]);
        if (lane < (nbytes & 3)) dst[out + (nbytes & ~3u) + lane] = wb[(nbytes & ~3u) + lane];
        out += nbytes;
        carry_bits = last ? 0 : total & 7;
        carry = carry_bits ? (uint32_t)wb[nbytes] & ((1u << carry_bits) - 1) : 0;
        wave_sync();
        if (last) break;
    }
    return out;
}
// code | length << 16 of every symbol in Z.h_cnt (free once the tree is built): one lookup per literal in ze_huf_stream
__device__ inline void ze_huf_pack_table(ZEncLds& Z) {
    const uint32_t lane = threadIdx.x & 63;
    for (uint32_t k = lane; k < 256; k += 64) Z.h_cnt[k] = (uint32_t)Z.hcode[k] | ((uint32_t)Z.hlen[k] << 16);
    wave_sync();
}

// HBM scratch of one wave: literal buffer (1x), sequence records (2x) and the compressed block under construction (3x: a
// block that turns out larger than its content is dropped for a raw block, so it is not built in the caller's buffer)
__host__ __device__ __forceinline__ uint64_t zstd_scratch_bytes(uint64_t n) {
    const uint64_t b = n < ZE_BLOCK ? n : ZE_BLOCK;
    return 6 * ((b + 15) & ~15ull) + 256;
}
// byte histogram of p[0, n) into Z.hist (all lanes; 16 bytes per lane and load) and the largest byte value present
__device__ inline uint32_t ze_histogram(ZEncLds& Z, const uint8_t* p, uint32_t n) {
    const uint32_t lane = threadIdx.x & 63;
    for (uint32_t k = lane; k < 256; k += 64) Z.hist[k] = 0;
    wave_sync();
    const uint32_t nvec = n >> 4;
    for (uint32_t i = lane; i < nvec; i += 64) {
        const u32x4 v = ldu128(p + 16 * (size_t)i);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            atomicAdd(&Z.hist[w[k] & 255], 1u);
            atomicAdd(&Z.hist[(w[k] >> 8) & 255], 1u);
            atomicAdd(&Z.hist[(w[k] >> 16) & 255], 1u);
            atomicAdd(&Z.hist[w[k] >> 24], 1u);
        }
    }
    for (uint32_t i = (nvec << 4) + lane; i < n; i += 64) atomicAdd(&Z.hist[ldu8(p + i)], 1u);
    wave_sync();
    uint32_t maxs = 0;
    for (uint32_t k = lane; k < 256; k += 64)
        if (Z.hist[k]) maxs = max(maxs, k);
    for (int d = 32; d > 0; d >>= 1) maxs = max(maxs, (uint32_t)__shfl_xor((int)maxs, d, 64));
    return maxs;
}

// One block of a frame: src[c0, c1) of the input src[0, n) -> 3-byte header + content at bh; returns its size.  The matcher
// carries its history from the blocks before (ALONE = false, one wave walks the frame) or starts without any (ALONE =
// true: the block is the only one of a frame of its own, see zstd_compress_block_alone).  `scratch`: zstd_scratch_bytes(cap).
template <bool ALONE, class MT>
__device__ uint32_t ze_block(const uint8_t* src, uint32_t n, uint32_t c0, uint32_t c1, bool last_block, uint8_t* bh, ZEncLds& Z,
                             uint8_t* scratch, uint32_t blk_cap, MT& mt) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t blk = c1 - c0;
    uint8_t* lits = scratch;                                   // <= blk_cap bytes
    const uint32_t cap16 = (blk_cap + 15) & ~15u;
    ZeSeq* seqs = (ZeSeq*)(scratch + cap16);                    // <= blk_cap / 4 records of 8 bytes
    uint8_t* attempt = scratch + 3 * (size_t)cap16;              // the compressed block is built here
    uint32_t total = 0;
    LZP_BEGIN
    uint32_t nseq = 0, nlit = 0;
    uint32_t tail_from = c0;
    bool probed = false, lit_only = false;
    uint32_t lit_maxs = 0;
    if (blk >= 32) {
        if (ALONE)
            mt.begin_alone(c0, c1 - 12, c1 - 5, false);
        else
            mt.begin_chunk(c0, c1 - 12, c1 - 5);
        LZP(8);
        while (mt.next()) {
            LZP(9);
            // ---- record the chosen sequences of this step: literals [lit_start, p) + match
            const bool chosen = (mt.C >> lane) & 1;
            uint32_t lit_start = mt.anchor;
            {
                const uint64_t below = mt.C & ((1ull << lane) - 1);
                const uint32_t prevl = below ? 63u - (uint32_t)__builtin_clzll(below) : 0u;
                const uint32_t pe = __shfl(mt.p + mt.mlen, prevl, 64);
                if (below) lit_start = pe;
            }
            const uint32_t ll = chosen ? mt.p - lit_start : 0u;
            // a match longer than 65535 is cut (the rest is found again as the next match)
            uint32_t ml = mt.mlen;
            const uint64_t longm = __ballot(chosen && ml > 65535u);
            if (longm) {   // only the first such match of the step is kept; the step ends there
                const uint32_t l0 = (uint32_t)__builtin_ctzll(longm);
                mt.C &= (2ull << l0) - 1;
                if (lane == l0) ml = 65535u;
                mt.covered = rdlane(mt.p, l0) + 65535u;
            }
            const bool keep = (mt.C >> lane) & 1;
            const uint32_t myll = keep ? ll : 0u;
            const uint32_t incl = wave_scan_dpp(myll);
            const uint32_t k = nseq + lane_rank(mt.C);
            if (keep) {
                ZeSeq r;
                r.ll = myll;
                r.ml = (uint16_t)ml;
                r.off = (uint16_t)(mt.p - mt.cand);
                seqs[k] = r;
                uint8_t* w = lits + nlit + (incl - myll);
                if (myll <= 64) {   // (at most 64 bytes back: in the matcher's ring, no HBM load)
                    for (uint32_t i = 0; i < myll; i += 8) {
                        const uint64_t v = lds_rd8_ring(Z.lz.ring, (lit_start + i) & (LzMatcher<ZE_HB, ZE_RB>::R - 1), LzMatcher<ZE_HB, ZE_RB>::RWM);
                        if (myll - i >= 8) {
                            stu64(w + i, v);
                        } else {
                            for (uint32_t b = 0; b < myll - i; b++) w[i + b] = (uint8_t)(v >> (8 * b));
                        }
                    }
                }
            }
            uint64_t bigl = __ballot(keep && myll > 64);
            while (bigl) {
                const uint32_t l = (uint32_t)__builtin_ctzll(bigl);
                bigl &= bigl - 1;
                wave_copy_g2g(lits + nlit + rdlane(incl - myll, l), src + rdlane(lit_start, l), rdlane(myll, l));
            }
            nlit += rdlane(incl, 63);
            nseq += (uint32_t)__popcll(mt.C);
            mt.advance();
            LZP(10);
            if (ALONE && !probed && blk >= 4096 && mt.anchor - c0 >= ZE_PROBE) {
                // After the first 2 KiB of a piece that is a frame of its own: are sequences worth their bits here?  A sequence
                // costs ~20 bits of FSE codes and offset; entropy-coded as a literal, a byte costs h = (Huffman bits of the
                // piece's byte histogram) / 8 bytes.  Matches of avg_ml bytes pay off only if avg_ml * h > 2.5 bytes — they do
                // not for columns of small records (64-bit values whose upper bytes repeat: 3-byte matches; short codes
                // like "s123"): those pieces become ONE Huffman-coded literals section with no sequences, which is smaller
                // (C5: 13.6 -> 8.x MB per array; libzstd level 3 writes 9.7) and skips the matcher, the sequence coding and,
                // on the read side, the sequence execution.
                probed = true;
                const uint32_t seen = mt.anchor - c0, matched = seen - min(seen, nlit);
                const uint32_t maxs0 = ze_histogram(Z, src + c0, blk);
                if (maxs0 >= 1) {
                    // order-0 cost of the piece: entropy + 2 % (what a Huffman code of these counts comes to), per symbol
                    // in 1/64 bits in Z.hcode (free until the tree is built)
                    uint32_t est64 = 0;
                    for (uint32_t k2 = lane; k2 < 256; k2 += 64) {
                        const uint32_t c = k2 <= maxs0 ? Z.hist[k2] : 0u;
                        const uint32_t b64 = c ? min(65535u, (uint32_t)(__log2f((float)blk / (float)c) * 65.28f + 0.5f)) : 0u;
                        Z.hcode[k2] = (uint16_t)max(b64, 64u);   // (no code of a tree is shorter than one bit)
                        est64 += c * max(b64, 64u);
                    }
                    for (int d = 32; d > 0; d >>= 1) est64 += (uint32_t)__shfl_xor((int)est64, d, 64);
                    const uint32_t est = est64 / 64;
                    const bool helps = est / 8 + 160 < blk - blk / 32;
                    // what the LZ parse would cost, extrapolated from the part seen: its literals at the piece's code lengths
                    // (the bytes matches take away are often the cheap ones: the literals left over cost MORE per byte than
                    // the average) + ~20 bits per sequence
                    wave_stores_visible();
                    uint32_t lbits = 0;
                    for (uint32_t i2 = lane; i2 < nlit; i2 += 64) lbits += Z.hcode[ldu8(lits + i2)];
                    for (int d = 32; d > 0; d >>= 1) lbits += (uint32_t)__shfl_xor((int)lbits, d, 64);
                    lbits /= 64;
                    const uint64_t lz_bits = ((uint64_t)lbits + 20ull * nseq) * blk / max(seen, 1u);
                    (void)matched;
                    const bool seq_poor = nseq == 0 || (uint64_t)est <= lz_bits;
                    if (helps && seq_poor) {
                        lit_only = true;
                        lit_maxs = maxs0;
                        break;
                    }
                }
            }
        }
        tail_from = mt.anchor;
    }
    if (lit_only) {   // the piece's own bytes are the literals
        lits = (uint8_t*)(src + c0);
        nlit = blk;
        nseq = 0;
    } else {
        // trailing literals of the block
        wave_copy_g2g(lits + nlit, src + tail_from, c1 - tail_from);
        nlit += c1 - tail_from;
    }
    wave_stores_visible();   // lits / seqs are read back below
    LZP(11);
    LZP_CNT(16, nseq);
    LZP_CNT(17, nlit);
    // ---- the block: compressed if that is smaller, raw otherwise
    uint8_t* body = attempt;
    uint32_t csize = 0;
    bool ok = (nseq > 0 || nlit > 64) && nlit / 2 + 2 * nseq < blk;   // (cheap lower bound of the compressed size)
    if (ok) {
        uint32_t q = 0;
        // ---- literals section
        uint32_t maxs, hbits = 0;
        maxs = lit_only ? lit_maxs : ze_histogram(Z, lits, nlit);   // (lit_only: the probe's histogram is of exactly these bytes)
        if (nlit >= 64 && maxs >= 1) {
            // (a block without sequences is read lane per stream when its codes have at most ZE_LITONLY_MAXBITS bits)
            const uint32_t bits = nseq == 0 ? ZE_LITONLY_MAXBITS : ZE_HUF_MAXBITS;
            // the parse of the frame's last block is over: the matcher's table and ring are the tree builder's scratch
            // (a block that is not its frame's last keeps the matcher's history: its literals stay raw — one wave never
            // walks a frame of several blocks in this library, buffers of more than one piece go through the chunk kernel)
            hbits = (ALONE || last_block) ? ze_huf_build_pm(Z, maxs, bits, (uint32_t*)Z.lz.tab) : 0u;
        }
        LZP(15);
        bool huf = hbits != 0;
        if (huf) {
            // size estimate: the tree + the coded bits must beat raw
            uint32_t est = 0;
            for (uint32_t k = lane; k <= maxs; k += 64) est += Z.hist[k] * Z.hlen[k];
            for (int d = 32; d > 0; d >>= 1) est += (uint32_t)__shfl_xor((int)est, d, 64);
            const uint32_t coded = (est + 7) / 8 + 1 + (maxs <= 128 ? (maxs + 1) / 2 : 100u) + 16;
            if (coded >= nlit || nlit > 262143u) huf = false;
        }
        if (huf) {
            const bool four = nlit >= 256;
            // header size: 3 bytes (sizes < 1024), 4 (< 16384), 5 (< 262144); single stream only in the 3-byte form
            const uint32_t hsz = !four ? 3u : 5u;   // (the compressed size is not known yet: the widest form always fits)
            uint8_t* lh = body + q;
            uint32_t w = hsz;
            // tree description: direct 4-bit weights of symbols 0 .. maxs - 1 (up to 128 of them), else FSE-coded weights
            if (maxs <= 128) {
                if (lane == 0) {
                    lh[w] = (uint8_t)(127 + maxs);
                    for (uint32_t s = 0; s < maxs; s += 2) {
                        const uint32_t w0 = Z.hlen[s] ? hbits + 1 - Z.hlen[s] : 0;
                        const uint32_t w1 = (s + 1 < maxs && Z.hlen[s + 1]) ? hbits + 1 - Z.hlen[s + 1] : 0;
                        lh[w + 1 + s / 2] = (uint8_t)((w0 << 4) | w1);
                    }
                }
                w += 1 + (maxs + 1) / 2;
            } else {
                ze_huf_weights_prep(Z, maxs, hbits);
                if (lane == 0) Z.misc[2] = ze_huf_weights_fse(Z, maxs, hbits, lh + w);
                wave_sync();
                __builtin_amdgcn_s_waitcnt(0);
                if (Z.misc[2] == 0) huf = false;
                w += Z.misc[2];
            }
            const uint32_t tree_end = w;
            LZP(7);
            if (huf) ze_huf_pack_table(Z);   // (after the tree description: ze_huf_weights_fse is done with its scratch)
            if (!huf) {
            } else if (four) {
                const uint32_t seg = (nlit + 3) / 4;
                uint32_t ssz[4];
                w += 6;   // jump table
                for (int j = 0; j < 4; j++) {
                    const uint32_t b0 = min(nlit, seg * j), b1 = min(nlit, seg * (j + 1));
                    ssz[j] = ze_huf_stream(Z, lits + b0, b1 - b0, lh + w);
                    w += ssz[j];
                }
                if (lane == 0)
                    for (int j = 0; j < 3; j++) {
                        lh[tree_end + 2 * j] = (uint8_t)ssz[j];
                        lh[tree_end + 2 * j + 1] = (uint8_t)(ssz[j] >> 8);
                    }
                if (ssz[0] > 65535u || ssz[1] > 65535u || ssz[2] > 65535u) huf = false;
            } else {
                w += ze_huf_stream(Z, lits, nlit, lh + w);
            }
            const uint32_t comp = w - hsz;   // tree + jump table + streams
            if (huf && comp < nlit && comp < 262144u) {
                if (lane == 0) {
                    if (!four) {            // type 2, size format 0: 10-bit sizes, single stream
                        const uint32_t v = 2u | (0u << 2) | (nlit << 4) | (comp << 14);
                        lh[0] = (uint8_t)v; lh[1] = (uint8_t)(v >> 8); lh[2] = (uint8_t)(v >> 16);
                    } else {                // type 2, size format 3: 18-bit sizes, four streams
                        const uint64_t v = 2ull | (3ull << 2) | ((uint64_t)nlit << 4) | ((uint64_t)comp << 22);
                        for (int k = 0; k < 5; k++) lh[k] = (uint8_t)(v >> (8 * k));
                    }
                }
                q += w;
            } else {
                huf = false;
            }
        }
        if (!huf) {   // raw literals
            uint8_t* lh = body + q;
            uint32_t hsz;
            if (nlit < 32) {
                hsz = 1;
                if (lane == 0) lh[0] = (uint8_t)(nlit << 3);
            } else if (nlit < 4096) {
                hsz = 2;
                if (lane == 0) { const uint32_t v = (nlit << 4) | (1u << 2); lh[0] = (uint8_t)v; lh[1] = (uint8_t)(v >> 8); }
            } else {
                hsz = 3;
                if (lane == 0) { const uint32_t v = (nlit << 4) | (3u << 2); lh[0] = (uint8_t)v; lh[1] = (uint8_t)(v >> 8); lh[2] = (uint8_t)(v >> 16); }
            }
            wave_copy_g2g(lh + hsz, lits, nlit);
            q += hsz + nlit;
        }
        LZP(12);
        // ---- sequences section
        uint8_t* sh = body + q;
        uint32_t shdr;
        if (nseq < 128) {
            shdr = 1;
            if (lane == 0) sh[0] = (uint8_t)nseq;
        } else if (nseq < 0x7F00) {
            shdr = 2;
            if (lane == 0) { sh[0] = (uint8_t)((nseq >> 8) + 0x80); sh[1] = (uint8_t)nseq; }
        } else {
            shdr = 3;
            if (lane == 0) { sh[0] = 0xFF; sh[1] = (uint8_t)(nseq - 0x7F00); sh[2] = (uint8_t)((nseq - 0x7F00) >> 8); }
        }
        q += shdr;
        if (nseq) {
            // The FSE state chain is serial (lane 0).  Everything else about a sequence — its three codes and the values of
            // its extra bits — is prepared 64 sequences at a time by all lanes (one HBM round trip per batch instead of one
            // per sequence), from the LAST sequence backwards.
            ZeBits bs{sh + shdr + 1, 0, 0};
            uint32_t s_ll = 0, s_of = 0, s_ml = 0;
            if (lane == 0) sh[shdr] = 0;   // Symbol_Compression_Modes: predefined LL / OF / ML
            auto init_state = [&](const uint16_t* st, const ZeSymTT* tt, uint32_t sym) -> uint32_t {
                const uint32_t nb = (uint32_t)(tt[sym].delta_nb_bits + (1 << 15)) >> 16;
                const uint32_t value = (nb << 16) - (uint32_t)tt[sym].delta_nb_bits;
                return st[(value >> nb) + tt[sym].delta_find_state];
            };
            for (uint32_t hi_k = nseq; hi_k > 0; hi_k -= min(hi_k, 64u)) {
                const uint32_t cnt = min(hi_k, 64u);
                if (lane < cnt) {
                    const ZeSeq r = seqs[hi_k - 1 - lane];
                    const uint32_t ofb = (uint32_t)r.off + 3;
                    const uint32_t llc = ze_ll_code(r.ll), mlc = ze_ml_code((uint32_t)r.ml - 3);
                    // (the code / bit-count tables live in HBM: looked up here, 64 at a time, not inside the serial loop)
                    Z.sq_code[lane] = llc | (mlc << 6) | (ze_highbit(ofb) << 12) | ((uint32_t)ZE_LL_BITS[llc] << 17) | ((uint32_t)ZE_ML_BITS[mlc] << 22);
                    Z.sq_ll[lane] = r.ll;
                    Z.sq_ml[lane] = (uint32_t)r.ml - 3;
                    Z.sq_of[lane] = ofb;
                    Z.sq_tt[lane][0] = Z.of_tt[ze_highbit(ofb)];
                    Z.sq_tt[lane][1] = Z.ml_tt[mlc];
                    Z.sq_tt[lane][2] = Z.ll_tt[llc];
                }
                wave_sync();
                if (lane == 0) {
                    for (uint32_t j = 0; j < cnt; j++) {
                        const uint32_t code = Z.sq_code[j], llc = code & 63, mlc = (code >> 6) & 63, ofc = (code >> 12) & 31;
                        const uint32_t llb = (code >> 17) & 31, mlb = code >> 22;
                        if (hi_k == nseq && j == 0) {   // the last sequence initialises the three states
                            s_ll = init_state(Z.ll_st, Z.ll_tt, llc);
                            s_of = init_state(Z.of_st, Z.of_tt, ofc);
                            s_ml = init_state(Z.ml_st, Z.ml_tt, mlc);
                        } else {
                            // the three chains are independent: their transforms arrive with one LDS round trip, their new
                            // states with a second one
                            const ZeSymTT t_of = Z.sq_tt[j][0], t_ml = Z.sq_tt[j][1], t_ll = Z.sq_tt[j][2];
                            const uint32_t nb_of = (uint32_t)(s_of + t_of.delta_nb_bits) >> 16;
                            const uint32_t nb_ml = (uint32_t)(s_ml + t_ml.delta_nb_bits) >> 16;
                            const uint32_t nb_ll = (uint32_t)(s_ll + t_ll.delta_nb_bits) >> 16;
                            const uint32_t n_of = Z.of_st[(s_of >> nb_of) + t_of.delta_find_state];
                            const uint32_t n_ml = Z.ml_st[(s_ml >> nb_ml) + t_ml.delta_find_state];
                            const uint32_t n_ll = Z.ll_st[(s_ll >> nb_ll) + t_ll.delta_find_state];
                            bs.add(s_of, nb_of);
                            bs.add(s_ml, nb_ml);
                            bs.add(s_ll, nb_ll);   // (<= 5 + 6 + 6 bits on top of < 8 pending)
                            bs.flush();
                            s_of = n_of;
                            s_ml = n_ml;
                            s_ll = n_ll;
                        }
                        bs.add(Z.sq_ll[j], llb);
                        bs.add(Z.sq_ml[j], mlb);   // (<= 16 + 16 bits on top of < 8 pending)
                        bs.flush();
                        bs.add(Z.sq_of[j], ofc);
                        bs.flush();
                    }
                }
                wave_sync();
            }
            if (lane == 0) {
                bs.add(s_ml, 6);
                bs.flush();
                bs.add(s_of, 5);
                bs.flush();
                bs.add(s_ll, 6);
                uint8_t* e = bs.close();
                Z.misc[1] = (uint32_t)(e - (sh + shdr));
            }
            wave_sync();
            __builtin_amdgcn_s_waitcnt(0);
            q += Z.misc[1];
        }
        csize = q;
        ok = csize < blk;
        LZP(13);
    }
    if (ok) {
        if (lane == 0) {
            const uint32_t v = (last_block ? 1u : 0u) | (2u << 1) | (csize << 3);
            bh[0] = (uint8_t)v; bh[1] = (uint8_t)(v >> 8); bh[2] = (uint8_t)(v >> 16);
        }
        wave_stores_visible();
        wave_copy_g2g(bh + 3, attempt, csize);
        total = 3 + csize;
    } else {   // raw block
        body = bh + 3;
        if (lane == 0) {
            const uint32_t v = (last_block ? 1u : 0u) | (0u << 1) | (blk << 3);
            bh[0] = (uint8_t)v; bh[1] = (uint8_t)(v >> 8); bh[2] = (uint8_t)(v >> 16);
        }
        wave_stores_visible();
        wave_copy_g2g(body, src + c0, blk);
        total = 3 + blk;
    }
    wave_stores_visible();
    LZP(14);
    LZP_END;
    return total;
}

__device__ __forceinline__ uint32_t ze_frame_header(uint8_t* dst, uint32_t n) {   // by one lane; returns the header size
    dst[0] = 0x28; dst[1] = 0xB5; dst[2] = 0x2F; dst[3] = 0xFD;
    if (n < 256) {
        dst[4] = 0x20;
        dst[5] = (uint8_t)n;
    } else if (n < 65536 + 256) {
        dst[4] = 0x60;
        dst[5] = (uint8_t)(n - 256);
        dst[6] = (uint8_t)((n - 256) >> 8);
    } else {
        dst[4] = 0xA0;
        for (int k = 0; k < 4; k++) dst[5 + k] = (uint8_t)(n >> (8 * k));
    }
    return n < 256 ? 6u : n < 65536 + 256 ? 7u : 9u;
}
__device__ __forceinline__ uint32_t ze_frame_header_bytes(uint32_t n) { return n < 256 ? 6u : n < 65536 + 256 ? 7u : 9u; }
// FSE encoding tables of the predefined distributions, built by the COMPILER (constexpr restatement of ze_build_ctable)
// and copied into LDS by the wave: every piece is a frame of its own, and three table builds by one lane were a fixed
// cost per 16 KiB piece.
struct ZePreTables {
    uint16_t ll_st[64], ml_st[64], of_st[32];
    ZeSymTT ll_tt[36], ml_tt[53], of_tt[29];
};
constexpr int zec_highbit(uint32_t v) {
    int r = 0;
    while (v >>= 1) r++;
    return r;
}
template <int NSYM, int LOG>
constexpr void zec_ctable(const int16_t (&norm)[NSYM], uint16_t* state_table, ZeSymTT* tt) {
    constexpr int size = 1 << LOG, mask = size - 1, step = (size >> 1) + (size >> 3) + 3;
    int high = size - 1;
    int cumul[NSYM + 1] = {};
    uint8_t spread[size] = {};
    for (int s = 0; s < NSYM; s++) {
        if (norm[s] == -1) {
            cumul[s + 1] = cumul[s] + 1;
            spread[high--] = (uint8_t)s;
        } else {
            cumul[s + 1] = cumul[s] + norm[s];
        }
    }
    int pos = 0;
    for (int s = 0; s < NSYM; s++) {
        for (int k = 0; k < norm[s]; k++) {
            spread[pos] = (uint8_t)s;
            pos = (pos + step) & mask;
            while (pos > high) pos = (pos + step) & mask;
        }
    }
    for (int u = 0; u < size; u++) {
        const int s = spread[u];
        state_table[cumul[s]++] = (uint16_t)(size + u);
    }
    int total = 0;
    for (int s = 0; s < NSYM; s++) {
        const int c = norm[s];
        if (c == 0) {
            tt[s].delta_nb_bits = ((LOG + 1) << 16) - (1 << LOG);
            tt[s].delta_find_state = 0;
        } else if (c == -1 || c == 1) {
            tt[s].delta_nb_bits = (LOG << 16) - (1 << LOG);
            tt[s].delta_find_state = total - 1;
            total++;
        } else {
            const int max_bits_out = LOG - zec_highbit((uint32_t)(c - 1));
            const int min_state_plus = c << max_bits_out;
            tt[s].delta_nb_bits = (max_bits_out << 16) - min_state_plus;
            tt[s].delta_find_state = total - c;
            total += c;
        }
    }
}
constexpr int16_t ZEC_LL[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
constexpr int16_t ZEC_ML[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};
constexpr int16_t ZEC_OF[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};
constexpr ZePreTables zec_make_pre() {
    ZePreTables t = {};
    zec_ctable<36, 6>(ZEC_LL, t.ll_st, t.ll_tt);
    zec_ctable<53, 6>(ZEC_ML, t.ml_st, t.ml_tt);
    zec_ctable<29, 5>(ZEC_OF, t.of_st, t.of_tt);
    return t;
}
__device__ const ZePreTables g_zepre = zec_make_pre();
__device__ __forceinline__ void ze_tables(ZEncLds& Z) {   // (once per wave)
    const uint32_t lane = threadIdx.x & 63;
    Z.ll_st[lane] = g_zepre.ll_st[lane];
    Z.ml_st[lane] = g_zepre.ml_st[lane];
    if (lane < 32) Z.of_st[lane] = g_zepre.of_st[lane];
    if (lane < 36) Z.ll_tt[lane] = g_zepre.ll_tt[lane];
    if (lane < 53) Z.ml_tt[lane] = g_zepre.ml_tt[lane];
    if (lane < 29) Z.of_tt[lane] = g_zepre.of_tt[lane];
    wave_sync();
}

// Compress src[0, n) into dst as one Zstd frame (capacity >= n + 3 * ceil(n / 128 KiB) + 16); executed by ONE wave64.
// `scratch` (HBM): zstd_scratch_bytes(n).  Returns the frame size.
__device__ uint32_t zstd_compress_wave(const uint8_t* src, uint32_t n, uint8_t* dst, ZEncLds& Z, uint8_t* scratch) {
    const uint32_t lane = threadIdx.x & 63;
    // ---- frame header: magic, single segment + frame content size (1 / 2 / 4 bytes as libzstd sizes it), no checksum
    if (lane == 0) ze_frame_header(dst, n);
    uint32_t o = ze_frame_header_bytes(n);
    if (n == 0) {   // one empty raw block, last
        if (lane == 0) { dst[o] = 1; dst[o + 1] = 0; dst[o + 2] = 0; }
        return o + 3;
    }
    ze_tables(Z);
    const uint32_t blk_cap = min(n, ZE_BLOCK);
    LzMatcher<ZE_HB, ZE_RB> mt(Z.lz, src, n);
    mt.init();
    for (uint32_t c0 = 0; c0 < n; c0 += ZE_BLOCK) {
        const uint32_t c1 = min(n, c0 + ZE_BLOCK);
        o += ze_block<false>(src, n, c0, c1, c1 == n, dst + o, Z, scratch, blk_cap, mt);
    }
    return o;
}

// The piece [c0, c1) of the buffer src[0, n) as a FRAME of its own (header with the piece's size + one block), compressed
// by a wave of its own into `out`; returns the frame's size.  A buffer's frames, back to back, are what ZSTD_decompress
// — and with it zstd::bulk::decompress_to_buffer, src/compression/basic.rs:93-97 — reads as one buffer; frames share no
// state (no window, tables or repeat offsets), so they are compressed AND decompressed independently, one wave each.
__device__ uint32_t zstd_compress_block_alone(const uint8_t* src, uint32_t n, uint32_t c0, uint32_t c1, uint8_t* out, ZEncLds& Z,
                                              uint8_t* scratch, uint32_t blk_cap) {
    if ((threadIdx.x & 63) == 0) ze_frame_header(out, c1 - c0);
    const uint32_t h = ze_frame_header_bytes(c1 - c0);
    ze_tables(Z);
    LzMatcher<ZE_HB, ZE_RB> mt(Z.lz, src, n);
    mt.init();
    return h + ze_block<true>(src, n, c0, c1, true, out + h, Z, scratch, blk_cap, mt);
}

}  // namespace sb
