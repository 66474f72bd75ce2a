// strawboat-hip: Zstandard frame decoder on the device (RFC 8878), codec id 2.
//
// Replaces zstd::bulk::decompress_to_buffer (libzstd) at the reference call site
// src/compression/basic.rs:93-97.  One wave per frame.  The entropy stages are serial by
// construction (FSE / Huffman bit streams are consumed backwards, every symbol depends on the
// state left by the previous one), so lane 0 walks them with the tables in LDS; the four Huffman
// literal streams decode on four lanes; sequence *execution* (literal and match copies into the
// page's output) is spread over all 64 lanes in batches of 64 decoded sequences.  Blocks of a
// frame and frames of different pages are independent of each other only at frame granularity:
// parallelism comes from the number of pages in flight.
#pragma once
#include "sb_common.h"
#include "sb_lz4.h"

namespace sb {

struct ZFse {
    uint8_t symbol, nbits;
    uint16_t base;
};
struct ZSeq {
    uint32_t ll, ml, off;
};
struct ZWork {  // per-wave LDS workspace (~12 KB)
    ZFse ll[512], of[256], ml[512];
    uint16_t htab[2048];   // Huffman decoding table: symbol | code length << 8 per code of huf_bits bits
    uint8_t wts[256];
    int16_t norm[64];
    uint16_t next[64];
    ZSeq seq[64];
    uint32_t ll_log, of_log, ml_log, have_ll, have_of, have_ml, have_huf, huf_bits;
    uint32_t rep[3];
    int32_t err;
    uint32_t nbatch;
    // window of the sequences bit stream (read backwards): bytes [bw_lo, bw_lo + ZBW) of it, refilled by the wave before a
    // batch of 64 sequences (a batch consumes < 64 * 12 bytes)
    uint32_t bw_lo;
    __attribute__((aligned(16))) uint8_t bw[2048 + 16];
    LzSeqLds ring;   // output ring of the sequence executor (sb_lz4.h LzSeqExec)
    uint32_t t_llbase[36], t_mlbase[53];   // baseline / extra-bit tables (copies of the __constant__ ones: the serial
    uint8_t t_llbits[36], t_mlbits[53];    // sequence loop reads them with LDS latency, not a scalar-cache miss each)
    // per FSE state: extra-bit count | baseline << 8 of the state's length code, so that the serial loop gets them with the
    // state's entry in ONE LDS round trip (code -> table was a dependent second one)
    uint32_t xll[512], xml[512];
    // the tables of the predefined distributions, built once per wave (pre_built: set to 0 by the kernel before its first job):
    // frames of 16 KiB pieces use them in every block, and building one costs lane 0 ~50 us
    ZFse pre_ll[64], pre_of[32], pre_ml[64];
    uint32_t pre_built;
};
constexpr uint32_t ZBW = 2048;

#define ZFAIL(code)         \
    do {                    \
        wk->err = (code);   \
        return 0;           \
    } while (0)

// bits [bitpos-nb, bitpos) of the stream as a number whose MSB is bit bitpos-1; bits below 0 read as 0
__device__ __forceinline__ uint32_t z_peek(const uint8_t* p, int64_t bitpos, int nb) {
    if (nb <= 0) return 0;
    int64_t lo = bitpos - nb;
    if (lo >= 0) {
        const int64_t byte0 = lo >> 3;
        const int sh = (int)(lo & 7);
        const int need = (sh + nb + 7) >> 3;
        uint64_t w = 0;
        for (int k = 0; k < need; k++) w |= (uint64_t)p[byte0 + k] << (8 * k);
        return (uint32_t)((w >> sh) & ((1ull << nb) - 1));
    }
    if (bitpos <= 0) return 0;
    const int avail = (int)bitpos;
    const int need = (avail + 7) >> 3;
    uint64_t w = 0;
    for (int k = 0; k < need; k++) w |= (uint64_t)p[k] << (8 * k);
    return (uint32_t)((w & ((1ull << avail) - 1)) << (int)(-lo));
}

__device__ inline bool z_fse_build2(ZFse* t, uint16_t* next, const int16_t* norm, int nsym, int log);
__device__ inline bool z_fse_build(ZFse* t, ZWork* wk, const int16_t* norm, int nsym, int log) {
    return z_fse_build2(t, wk->next, norm, nsym, log);
}
__device__ inline bool z_fse_build2(ZFse* t, uint16_t* next, const int16_t* norm, int nsym, int log) {
    const int size = 1 << log;
    int high = size - 1;
    for (int s = 0; s < nsym; s++) {
        if (norm[s] == -1) {
            t[high--].symbol = (uint8_t)s;
            next[s] = 1;
        } else {
            next[s] = (uint16_t)norm[s];
        }
    }
    const int step = (size >> 1) + (size >> 3) + 3, mask = size - 1;
    int pos = 0;
    for (int s = 0; s < nsym; s++)
        for (int i = 0; i < norm[s]; i++) {
            t[pos].symbol = (uint8_t)s;
            do {
                pos = (pos + step) & mask;
            } while (pos > high);
        }
    if (pos != 0) return false;
    for (int i = 0; i < size; i++) {
        const uint8_t s = t[i].symbol;
        const uint16_t ns = next[s]++;
        const int nb = log - (31 - __clz((int)ns));
        t[i].nbits = (uint8_t)nb;
        t[i].base = (uint16_t)(((uint32_t)ns << nb) - (uint32_t)size);
    }
    return true;
}

// FSE table description (forward bit stream); returns bytes consumed or 0 on error
__device__ inline uint32_t z_fse_header(const uint8_t* src, uint32_t n, int max_sym, int max_log, int16_t* norm,
                                        int* nsym_out, int* log_out) {
    uint64_t bitpos = 0;
    auto peek = [&](int nb) -> uint32_t {
        uint64_t v = 0;
        const uint64_t byte = bitpos >> 3;
        for (int i = 0; i < 6 && byte + i < n; i++) v |= (uint64_t)src[byte + i] << (8 * i);
        v >>= (bitpos & 7);
        return (uint32_t)(v & ((1ull << nb) - 1));
    };
    const int log = (int)peek(4) + 5;
    bitpos += 4;
    if (log > max_log) return 0;
    int remaining = (1 << log) + 1, threshold = 1 << log, nbits = log + 1, sym = 0;
    bool prev0 = false;
    for (int i = 0; i <= max_sym; i++) norm[i] = 0;
    while (remaining > 1 && sym <= max_sym) {
        if (prev0) {
            for (;;) {
                const uint32_t r = peek(2);
                bitpos += 2;
                sym += (int)r;
                if (r != 3) break;
            }
            prev0 = false;
            if (sym > max_sym) return 0;
            continue;
        }
        const int mx = (2 * threshold - 1) - remaining;
        int count;
        const uint32_t v = peek(nbits);
        if ((int)(v & (uint32_t)(threshold - 1)) < mx) {
            count = (int)(v & (uint32_t)(threshold - 1));
            bitpos += nbits - 1;
        } else {
            count = (int)(v & (uint32_t)(2 * threshold - 1));
            if (count >= threshold) count -= mx;
            bitpos += nbits;
        }
        count--;
        remaining -= count < 0 ? -count : count;
        norm[sym++] = (int16_t)count;
        prev0 = count == 0;
        while (remaining < threshold) {
            nbits--;
            threshold >>= 1;
        }
    }
    if (remaining != 1) return 0;
    const uint32_t used = (uint32_t)((bitpos + 7) >> 3);
    if (used > n) return 0;
    *nsym_out = sym;
    *log_out = log;
    return used;
}

__device__ inline bool z_huf_build(ZWork* wk, int nw) {
    uint32_t total = 0;
    for (int i = 0; i < nw; i++) {
        if (wk->wts[i] > 11) return false;
        total += wk->wts[i] ? (1u << (wk->wts[i] - 1)) : 0;
    }
    if (total == 0) return false;
    const int max_bits = 32 - __clz((int)total);
    const uint32_t left = (1u << max_bits) - total;
    if (left == 0 || (left & (left - 1)) || max_bits > 11) return false;
    wk->wts[nw] = (uint8_t)(32 - __clz((int)left));
    const int n = nw + 1;
    wk->huf_bits = (uint32_t)max_bits;
    uint32_t code = 0;
    for (int wt = 1; wt <= max_bits; wt++)
        for (int s = 0; s < n; s++) {
            if (wk->wts[s] != wt) continue;
            const uint32_t span = 1u << (wt - 1);
            for (uint32_t k = 0; k < span; k++) {
                wk->htab[code + k] = (uint16_t)((uint32_t)s | ((uint32_t)(max_bits + 1 - wt) << 8));
            }
            code += span;
        }
    return true;
}

// Huffman tree description; returns bytes consumed or 0
__device__ inline uint32_t z_huf_read(ZWork* wk, const uint8_t* src, uint32_t n) {
    if (n < 1) return 0;
    const uint8_t hb = src[0];
    int nw;
    uint32_t used;
    if (hb >= 128) {
        nw = hb - 127;
        const uint32_t bytes = (uint32_t)(nw + 1) / 2;
        if (n < 1 + bytes) return 0;
        for (int i = 0; i < nw; i++) {
            const uint8_t b = src[1 + i / 2];
            wk->wts[i] = (i & 1) ? (b & 15) : (b >> 4);
        }
        used = 1 + bytes;
    } else {
        const uint32_t clen = hb;
        if (n < 1 + clen || clen < 2) return 0;
        int nsym, log;
        const uint32_t hsz = z_fse_header(src + 1, clen, 12, 6, wk->norm, &nsym, &log);
        if (!hsz || hsz >= clen) return 0;
        ZFse* t = wk->ll;  // scratch: the literal-length table is rebuilt later
        if (!z_fse_build(t, wk, wk->norm, nsym, log)) return 0;
        const uint8_t* bs = src + 1 + hsz;
        const uint32_t bn = clen - hsz;
        if (bs[bn - 1] == 0) return 0;
        int64_t bitpos = (int64_t)(bn - 1) * 8 + (31 - __clz((int)bs[bn - 1]));
        uint32_t s1 = z_peek(bs, bitpos, log);
        bitpos -= log;
        uint32_t s2 = z_peek(bs, bitpos, log);
        bitpos -= log;
        nw = 0;
        for (;;) {  // two interleaved FSE states; when the stream runs dry the other state is flushed
            if (nw >= 254) return 0;
            wk->wts[nw++] = t[s1].symbol;
            if (bitpos < t[s1].nbits) {
                wk->wts[nw++] = t[s2].symbol;
                break;
            }
            {
                const uint32_t nb = t[s1].nbits;
                s1 = t[s1].base + z_peek(bs, bitpos, (int)nb);
                bitpos -= nb;
            }
            wk->wts[nw++] = t[s2].symbol;
            if (bitpos < t[s2].nbits) {
                wk->wts[nw++] = t[s1].symbol;
                break;
            }
            {
                const uint32_t nb = t[s2].nbits;
                s2 = t[s2].base + z_peek(bs, bitpos, (int)nb);
                bitpos -= nb;
            }
        }
        used = 1 + clen;
    }
    if (!z_huf_build(wk, nw)) return 0;
    wk->have_huf = 1;
    return used;
}

// one Huffman stream (executed by one lane); returns false on error: z_lane_huf_stream (below) with the wave's table —
// one table read per symbol, a register window refilled one load ahead, eight output bytes per store
__device__ inline bool z_lane_huf_stream(const uint16_t* tab, uint32_t mb, const uint8_t* sb_, uint32_t sn, uint8_t* dst, uint32_t outn);
__device__ inline bool z_huf_stream(const ZWork* wk, const uint8_t* src, uint32_t n, uint8_t* dst, uint32_t out) {
    return z_lane_huf_stream(wk->htab, wk->huf_bits, src, n, dst, out);
}

__constant__ int16_t Z_LL_DEFAULT[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2,
                                         2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
__constant__ int16_t Z_ML_DEFAULT[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                         1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};
__constant__ int16_t Z_OF_DEFAULT[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1,
                                         1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};
__constant__ uint32_t Z_LL_BASE[36] = {0,  1,  2,  3,  4,  5,  6,  7,  8,  9,   10,  11,  12,  13,   14,   15,   16,   18,
                                       20, 22, 24, 28, 32, 40, 48, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536};
__constant__ uint8_t Z_LL_BITS[36] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1,
                                      1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
__constant__ uint32_t Z_ML_BASE[53] = {3,  4,  5,  6,  7,  8,  9,  10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20,
                                       21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 37, 39, 41,
                                       43, 47, 51, 59, 67, 83, 99, 131, 259, 515, 1027, 2051, 4099, 8195, 16387, 32771, 65539};
__constant__ uint8_t Z_ML_BITS[53] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                      0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};

// ---- the decoding tables of the predefined distributions (RFC 8878 3.1.1.3.2.2), built at COMPILE time: entries packed
// like ZFse (symbol | nbits << 8 | base << 16); xll / xml = per state the extra-bit count | baseline << 8 of its length code
struct ZPreTables {
    uint32_t ll[64], of[32], ml[64], xll[64], xml[64];
};
constexpr uint32_t zc_highbit(uint32_t v) {
    uint32_t r = 0;
    while (v >>= 1) r++;
    return r;
}
template <int NSYM, int LOG>
constexpr void zc_build(const int (&norm)[NSYM], uint32_t* out) {
    constexpr int size = 1 << LOG;
    int high = size - 1;
    uint32_t next[NSYM] = {};
    uint32_t sym[size] = {};
    for (int s = 0; s < NSYM; s++) {
        if (norm[s] == -1) {
            sym[high--] = (uint32_t)s;
            next[s] = 1;
        } else {
            next[s] = (uint32_t)norm[s];
        }
    }
    const int step = (size >> 1) + (size >> 3) + 3, mask = size - 1;
    int pos = 0;
    for (int s = 0; s < NSYM; s++)
        for (int i = 0; i < norm[s]; i++) {
            sym[pos] = (uint32_t)s;
            do {
                pos = (pos + step) & mask;
            } while (pos > high);
        }
    for (int i = 0; i < size; i++) {
        const uint32_t sy = sym[i];
        const uint32_t ns = next[sy]++;
        const uint32_t nb = (uint32_t)LOG - zc_highbit(ns);
        const uint32_t base = (ns << nb) - (uint32_t)size;
        out[i] = sy | (nb << 8) | (base << 16);
    }
}
constexpr ZPreTables zc_make_pre() {
    ZPreTables t = {};
    constexpr int ll_norm[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
    constexpr int ml_norm[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};
    constexpr int of_norm[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};
    constexpr uint32_t ll_base[36] = {0,  1,  2,  3,  4,  5,  6,  7,  8,  9,   10,  11,  12,  13,   14,   15,   16,   18,
                                      20, 22, 24, 28, 32, 40, 48, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536};
    constexpr uint32_t ll_bits[36] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
    constexpr uint32_t ml_base[53] = {3,  4,  5,  6,  7,  8,  9,  10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20,
                                      21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 37, 39, 41,
                                      43, 47, 51, 59, 67, 83, 99, 131, 259, 515, 1027, 2051, 4099, 8195, 16387, 32771, 65539};
    constexpr uint32_t ml_bits[53] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                      0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
    zc_build<36, 6>(ll_norm, t.ll);
    zc_build<29, 5>(of_norm, t.of);
    zc_build<53, 6>(ml_norm, t.ml);
    for (int i = 0; i < 64; i++) {
        const uint32_t cl = t.ll[i] & 255, cm = t.ml[i] & 255;
        t.xll[i] = ll_bits[cl] | (ll_base[cl] << 8);
        t.xml[i] = ml_bits[cm] | (ml_base[cm] << 8);
    }
    return t;
}
__device__ const ZPreTables g_zpre = zc_make_pre();

// ---- sequences decoded LANE PER FRAME (k_zstd_seq_lanes): the FSE state chain of a frame is serial, so a wave that owns
// one frame spends ~900 cycles per sequence on it; 64 lanes that each own a frame run 64 chains in the same instructions.
// A frame qualifies when every compressed block codes its sequences with the predefined tables or single-symbol (RLE)
// tables — what this library's encoder writes for its 16 KiB frames; frames with transmitted tables (libzstd's usual
// choice) keep the one-wave path.  Output: one 64-bit record per sequence, repeat offsets resolved,
//     literal length (18 bits) | match length << 18 (18 bits) | offset << 36 (28 bits)
// which zstd_inflate_wave executes instead of decoding the stream itself.
struct ZPre {
    uint32_t rec_off;   // first record of the frame in the record area (ZPRE_NONE: not pre-decoded)
    uint32_t nrec;
};
constexpr uint32_t ZPRE_NONE = 0xFFFFFFFFu;
struct ZLaneTabs {      // LDS copy of g_zpre
    uint32_t ll[64], of[32], ml[64], xll[64], xml[64];
};
// 64 stream bits whose top bit is stream bit `bitpos - 1` (bits below the start of the stream read as 0)
__device__ __forceinline__ uint64_t z_lane_top(const uint8_t* sb_, int32_t bitpos) {
    const int32_t lo = bitpos - 64;
    if (lo >= 0) {
        const uint32_t byte0 = (uint32_t)lo >> 3, sh = (uint32_t)lo & 7;
        uint64_t v = ldu64(sb_ + byte0) >> sh;
        if (sh) v |= (uint64_t)ldu8(sb_ + byte0 + 8) << (64 - sh);
        return v;
    }
    if (bitpos <= 0) return 0ull;
    uint64_t v = 0;
    const uint32_t nbytes = ((uint32_t)bitpos + 7) >> 3;
    for (uint32_t k = 0; k < nbytes; k++) v |= (uint64_t)ldu8(sb_ + k) << (8 * k);
    return v << (uint32_t)(-lo);
}
// One frame (src[0, n) must be exactly one frame with a content size of out_len) walked by ONE LANE.  recs == nullptr:
// count the sequences and check that the frame qualifies; else write the records.  Returns the number of sequences, or
// ZPRE_NONE when the frame does not qualify / is malformed (the one-wave decoder then reports the error).
__device__ uint32_t z_lane_frame(const uint8_t* src, uint32_t n, uint32_t out_len, const ZLaneTabs& T, uint64_t* recs, uint32_t nrec_cap) {
    if (n < 6 || ldu32(src) != 0xFD2FB528u) return ZPRE_NONE;
    uint32_t ip = 4;
    const uint8_t fhd = ldu8(src + ip++);
    const int fcs_flag = fhd >> 6, single = (fhd >> 5) & 1, checksum = (fhd >> 2) & 1, did = fhd & 3;
    if ((fhd & 0x08) || did) return ZPRE_NONE;
    if (!single) ip += 1;
    const uint32_t fcs_bytes = fcs_flag == 0 ? (single ? 1u : 0u) : (1u << fcs_flag);
    if (n - ip < fcs_bytes + 3) return ZPRE_NONE;
    ip += fcs_bytes;
    uint32_t total = 0, op = 0;
    uint32_t r0 = 1, r1 = 4, r2 = 8;
    // current tables: mode 0 = predefined (LDS), 1 = one symbol (kept here); `have` as in the one-wave decoder
    uint32_t kind_ll = 2, kind_of = 2, kind_ml = 2, rle_ll = 0, rle_of = 0, rle_ml = 0;
    for (;;) {
        if (n - ip < 3) return ZPRE_NONE;
        const uint32_t bh = (uint32_t)ldu8(src + ip) | ((uint32_t)ldu8(src + ip + 1) << 8) | ((uint32_t)ldu8(src + ip + 2) << 16);
        ip += 3;
        const uint32_t btype = (bh >> 1) & 3, bsize = bh >> 3;
        if (btype == 3) return ZPRE_NONE;
        if (btype == 0 || btype == 1) {
            const uint32_t body = btype == 1 ? 1u : bsize;
            if (n - ip < body) return ZPRE_NONE;
            ip += body;
            op += bsize;
        } else {
            if (bsize > 128 * 1024 || n - ip < bsize || bsize < 2) return ZPRE_NONE;
            const uint8_t* bs = src + ip;
            uint32_t bp = 0;
            const uint8_t b0 = ldu8(bs + bp++);
            const uint32_t ltype = b0 & 3, sf = (b0 >> 2) & 3;
            uint32_t regen = 0;
            if (ltype == 0 || ltype == 1) {
                if (sf == 0 || sf == 2) {
                    regen = b0 >> 3;
                } else if (sf == 1) {
                    if (bsize - bp < 1) return ZPRE_NONE;
                    regen = (b0 >> 4) | ((uint32_t)ldu8(bs + bp) << 4);
                    bp += 1;
                } else {
                    if (bsize - bp < 2) return ZPRE_NONE;
                    regen = (b0 >> 4) | ((uint32_t)ldu8(bs + bp) << 4) | ((uint32_t)ldu8(bs + bp + 1) << 12);
                    bp += 2;
                }
                const uint32_t body = ltype == 0 ? regen : 1u;
                if (bsize - bp < body) return ZPRE_NONE;
                bp += body;
            } else {
                uint32_t csize;
                if (bsize - bp < 4) return ZPRE_NONE;
                if (sf == 0 || sf == 1) {
                    const uint32_t v = (b0 >> 4) | ((uint32_t)ldu8(bs + bp) << 4) | ((uint32_t)ldu8(bs + bp + 1) << 12);
                    bp += 2;
                    regen = v & 0x3FF;
                    csize = v >> 10;
                } else if (sf == 2) {
                    const uint32_t v = (b0 >> 4) | ((uint32_t)ldu8(bs + bp) << 4) | ((uint32_t)ldu8(bs + bp + 1) << 12) | ((uint32_t)ldu8(bs + bp + 2) << 20);
                    bp += 3;
                    regen = v & 0x3FFF;
                    csize = v >> 14;
                } else {
                    const uint64_t v = (b0 >> 4) | ((uint64_t)ldu8(bs + bp) << 4) | ((uint64_t)ldu8(bs + bp + 1) << 12) |
                                       ((uint64_t)ldu8(bs + bp + 2) << 20) | ((uint64_t)ldu8(bs + bp + 3) << 28);
                    bp += 4;
                    regen = (uint32_t)(v & 0x3FFFF);
                    csize = (uint32_t)(v >> 18);
                }
                if (bsize - bp < csize) return ZPRE_NONE;
                bp += csize;
            }
            if (bsize - bp < 1) return ZPRE_NONE;
            uint32_t nseq;
            {
                const uint8_t s0 = ldu8(bs + bp++);
                if (s0 < 128) {
                    nseq = s0;
                } else if (s0 < 255) {
                    if (bsize - bp < 1) return ZPRE_NONE;
                    nseq = ((uint32_t)(s0 - 128) << 8) + ldu8(bs + bp);
                    bp += 1;
                } else {
                    if (bsize - bp < 2) return ZPRE_NONE;
                    nseq = (uint32_t)ldu8(bs + bp) + ((uint32_t)ldu8(bs + bp + 1) << 8) + 0x7F00;
                    bp += 2;
                }
            }
            uint32_t block_out = regen;
            if (nseq) {
                if (bsize - bp < 1) return ZPRE_NONE;
                const uint8_t modes = ldu8(bs + bp++);
                if (modes & 3) return ZPRE_NONE;
                const uint32_t m_ll = (modes >> 6) & 3, m_of = (modes >> 4) & 3, m_ml = (modes >> 2) & 3;
                if (m_ll == 2 || m_of == 2 || m_ml == 2) return ZPRE_NONE;   // transmitted tables: the one-wave path
                if (m_ll == 1) {
                    if (bsize - bp < 1) return ZPRE_NONE;
                    rle_ll = ldu8(bs + bp++);
                    if (rle_ll > 35) return ZPRE_NONE;
                    kind_ll = 1;
                } else if (m_ll == 0) {
                    kind_ll = 0;
                } else if (kind_ll == 2) {
                    return ZPRE_NONE;
                }
                if (m_of == 1) {
                    if (bsize - bp < 1) return ZPRE_NONE;
                    rle_of = ldu8(bs + bp++);
                    if (rle_of > 31) return ZPRE_NONE;
                    kind_of = 1;
                } else if (m_of == 0) {
                    kind_of = 0;
                } else if (kind_of == 2) {
                    return ZPRE_NONE;
                }
                if (m_ml == 1) {
                    if (bsize - bp < 1) return ZPRE_NONE;
                    rle_ml = ldu8(bs + bp++);
                    if (rle_ml > 52) return ZPRE_NONE;
                    kind_ml = 1;
                } else if (m_ml == 0) {
                    kind_ml = 0;
                } else if (kind_ml == 2) {
                    return ZPRE_NONE;
                }
                if (recs) {
                    if (total + nseq > nrec_cap) return ZPRE_NONE;
                    const uint8_t* sb_ = bs + bp;
                    const uint32_t sn_ = bsize - bp;
                    if (sn_ == 0) return ZPRE_NONE;
                    const uint8_t lastb = ldu8(sb_ + sn_ - 1);
                    if (lastb == 0) return ZPRE_NONE;
                    int32_t bitpos = (int32_t)(sn_ - 1) * 8 + (31 - __clz((int)lastb));
                    const uint32_t lll = kind_ll ? 0u : 6u, ofl = kind_of ? 0u : 5u, mll = kind_ml ? 0u : 6u;
                    uint64_t C = z_lane_top(sb_, bitpos);
                    uint32_t cbits = 64;   // valid bits left in C (top-aligned)
                    auto take = [&](uint32_t nb) -> uint32_t {
                        if (nb == 0) return 0u;
                        const uint32_t v = (uint32_t)(C >> (64 - nb));
                        C <<= nb;
                        cbits -= nb;
                        bitpos -= (int32_t)nb;
                        return v;
                    };
                    uint32_t sl = take(lll), so = take(ofl), sm = take(mll);
                    // single-symbol tables: the state stays 0 and the entry is the symbol with 0 bits
                    const uint32_t fx_ll = rle_ll, fx_of = rle_of, fx_ml = rle_ml;
                    const uint32_t fx_xll = (uint32_t)Z_LL_BITS[rle_ll] | (Z_LL_BASE[rle_ll] << 8);
                    const uint32_t fx_xml = (uint32_t)Z_ML_BITS[rle_ml] | (Z_ML_BASE[rle_ml] << 8);
                    for (uint32_t k = 0; k < nseq; k++) {
                        const uint32_t e_of = kind_of ? fx_of : T.of[so], e_ml = kind_ml ? fx_ml : T.ml[sm], e_ll = kind_ll ? fx_ll : T.ll[sl];
                        const uint32_t x_ml = kind_ml ? fx_xml : T.xml[sm], x_ll = kind_ll ? fx_xll : T.xll[sl];
                        const uint32_t ofc = e_of & 255;
                        if (ofc > 27) return ZPRE_NONE;    // (offsets beyond 2^28 do not fit a record)
                        const uint32_t mlbits = x_ml & 255, llbits = x_ll & 255;
                        if (cbits < ofc + mlbits + llbits) {
                            C = z_lane_top(sb_, bitpos);
                            cbits = 64;
                        }
                        const uint32_t ofv = (1u << ofc) + take(ofc);
                        const uint32_t mlen = (x_ml >> 8) + take(mlbits);
                        const uint32_t llen = (x_ll >> 8) + take(llbits);
                        uint32_t offset;
                        if (ofv > 3) {
                            offset = ofv - 3;
                            r2 = r1;
                            r1 = r0;
                            r0 = offset;
                        } else {
                            uint32_t idx = ofv - 1;
                            if (llen == 0) idx++;
                            if (idx == 0) {
                                offset = r0;
                            } else {
                                offset = idx == 1 ? r1 : idx == 2 ? r2 : r0 - 1;
                                if (offset == 0) return ZPRE_NONE;
                                if (idx > 1) r2 = r1;
                                r1 = r0;
                                r0 = offset;
                            }
                        }
                        if (llen >= (1u << 18) || mlen >= (1u << 18) || offset >= (1u << 28)) return ZPRE_NONE;
                        if (k + 1 < nseq) {
                            const uint32_t nl = (e_ll >> 8) & 255, nm = (e_ml >> 8) & 255, no = (e_of >> 8) & 255;
                            if (cbits < nl + nm + no) {
                                C = z_lane_top(sb_, bitpos);
                                cbits = 64;
                            }
                            sl = (e_ll >> 16) + take(nl);
                            sm = (e_ml >> 16) + take(nm);
                            so = (e_of >> 16) + take(no);
                        }
                        if (bitpos < 0) return ZPRE_NONE;
                        block_out += mlen;
                        recs[total + k] = (uint64_t)llen | ((uint64_t)mlen << 18) | ((uint64_t)offset << 36);
                    }
                    if (bitpos != 0) return ZPRE_NONE;
                }
                total += nseq;
            }
            (void)block_out;
            (void)op;
            ip += bsize;
        }
        if (bh & 1) break;
    }
    if (checksum) {
        if (n - ip < 4) return ZPRE_NONE;
        ip += 4;
    }
    if (ip != n) return ZPRE_NONE;     // more than one frame: the one-wave path walks them
    (void)out_len;
    return total;
}

// ---- literals-only frames decoded LANE PER STREAM (k_inflate, batches): a frame whose one block is a Huffman-coded literals
// section of four streams and no sequences — what this library's encoder writes for pieces where sequences do not pay
// (sb_zstd_enc.h) — is four serial bit streams; on the one-wave path they occupy 4 lanes of 64 and a lane pays ~800 cycles
// per symbol (two table reads, a byte store and an exposed refill).  Here 16 such frames are decoded together: lane 4 g + j
// takes stream j of frame g, with the frame's table in LDS (one u16 per code: symbol | length << 8, codes of up to ZH_MAXBITS
// bits), a 128-bit register window refilled one load ahead, and eight output bytes per store.
constexpr uint32_t ZH_MAXBITS = 9, ZH_GROUP = 16;
struct ZHufFrame {   // a literals-only frame (filled by the lane that owns the job)
    const uint8_t* ls;   // literals section payload: tree description, jump table, streams
    uint8_t* dst;
    uint32_t lleft;      // bytes of that payload
    uint32_t regen;      // bytes it regenerates (= the frame's content)
};
struct ZHufBuild {   // table construction, lane per frame
    uint8_t wts[ZH_GROUP][256];
    ZFse fse[ZH_GROUP][64];
    int16_t norm[ZH_GROUP][16];
    uint16_t next[ZH_GROUP][16];
};
struct ZHufStage {   // stream decoding, lane per stream: a 128-byte ring per stream, filled by the wave together
    uint32_t ring[64][32];   // row = stream; the stream's 32-bit word w (bytes [4 w, 4 w + 4) of it) sits in slot (w + stream) & 31
    uint64_t ptr[64];        // stream base, then (after the loads of a round are issued) where the round's bytes go
    int32_t lo[64];          // lowest byte offset of the stream that is in its ring (a multiple of 16)
    uint8_t nch[64];         // 16-byte chunks below `lo` to bring in this round (0 .. 4)
    uint8_t cnt[64];         // bytes decoded in the round
};
struct ZHufLanes {   // phase workspace (shares its LDS with ZWork)
    uint16_t tab[ZH_GROUP][1 << ZH_MAXBITS];
    union {
        ZHufBuild b;
        ZHufStage s;
    };
    uint64_t ob[64][8];          // the round's output, 64 bytes per stream (u64 m of stream s in slot (m + s) & 7)
    ZHufFrame fr[ZH_GROUP];
    uint32_t bits[ZH_GROUP];     // code bits of frame g's table (0: not decodable here)
    uint32_t str0[ZH_GROUP];     // offset of the jump table inside ls
};
static_assert(sizeof(ZHufStage) <= sizeof(ZHufBuild), "sb_zstd.h: the staging area lives in the table builder's scratch");
// One frame src[0, n) with content out_len: is it a literals-only frame of the shape above?  (lane per frame)
__device__ inline bool z_lane_litonly(const uint8_t* src, uint32_t n, uint32_t out_len, const uint8_t** ls, uint32_t* lleft, uint32_t* regen_out) {
    if (n < 12 || ldu32(src) != 0xFD2FB528u) return false;
    uint32_t ip = 4;
    const uint8_t fhd = ldu8(src + ip++);
    const int fcs_flag = fhd >> 6, single = (fhd >> 5) & 1, checksum = (fhd >> 2) & 1, did = fhd & 3;
    if ((fhd & 0x08) || did) return false;
    if (!single) ip += 1;
    const uint32_t fcs_bytes = fcs_flag == 0 ? (single ? 1u : 0u) : (1u << fcs_flag);
    if (fcs_bytes == 0 || fcs_bytes == 8 || n - ip < fcs_bytes + 3) return false;
    uint32_t fcs = 0;
    for (uint32_t i = 0; i < fcs_bytes; i++) fcs |= (uint32_t)ldu8(src + ip + i) << (8 * i);
    if (fcs_bytes == 2) fcs += 256;
    if (fcs != out_len) return false;
    ip += fcs_bytes;
    const uint32_t bh = (uint32_t)ldu8(src + ip) | ((uint32_t)ldu8(src + ip + 1) << 8) | ((uint32_t)ldu8(src + ip + 2) << 16);
    ip += 3;
    if (!(bh & 1) || ((bh >> 1) & 3) != 2) return false;       // one block, the last, compressed
    const uint32_t bsize = bh >> 3;
    if (bsize > 128 * 1024 || n - ip < bsize || bsize < 8) return false;
    if (ip + bsize + (checksum ? 4u : 0u) != n) return false;
    const uint8_t* bs = src + ip;
    const uint8_t b0 = ldu8(bs);
    const uint32_t ltype = b0 & 3, sf = (b0 >> 2) & 3;
    if (ltype != 2 || sf == 0) return false;                     // Huffman with its tree, four streams
    uint32_t bp = 1, regen, csize;
    if (sf == 1) {
        const uint32_t v = (b0 >> 4) | ((uint32_t)ldu8(bs + bp) << 4) | ((uint32_t)ldu8(bs + bp + 1) << 12);
        bp += 2;
        regen = v & 0x3FF;
        csize = v >> 10;
    } else if (sf == 2) {
        const uint32_t v = (b0 >> 4) | ((uint32_t)ldu8(bs + bp) << 4) | ((uint32_t)ldu8(bs + bp + 1) << 12) | ((uint32_t)ldu8(bs + bp + 2) << 20);
        bp += 3;
        regen = v & 0x3FFF;
        csize = v >> 14;
    } else {
        const uint64_t v = (b0 >> 4) | ((uint64_t)ldu8(bs + bp) << 4) | ((uint64_t)ldu8(bs + bp + 1) << 12) |
                           ((uint64_t)ldu8(bs + bp + 2) << 20) | ((uint64_t)ldu8(bs + bp + 3) << 28);
        bp += 4;
        regen = (uint32_t)(v & 0x3FFFF);
        csize = (uint32_t)(v >> 18);
    }
    if (bsize - bp < csize + 1 || bp + csize + 1 != bsize) return false;   // the sequences section is its one byte ...
    if (ldu8(bs + bp + csize) != 0) return false;                          // ... "no sequences"
    if (regen != out_len || regen < 64 || csize < 16) return false;
    *ls = bs + bp;
    *lleft = csize;
    *regen_out = regen;
    return true;
}
// The tree of frame g (lane g < ZH_GROUP): weights (direct or FSE-coded) -> H.tab[g]; H.bits[g] = code bits or 0.
__device__ inline void z_lane_huf_table(ZHufLanes& H, uint32_t g) {
    const uint8_t* src = H.fr[g].ls;
    const uint32_t n = H.fr[g].lleft;
    H.bits[g] = 0;
    uint8_t* wts = H.b.wts[g];
    const uint8_t hb = ldu8(src);
    int nw;
    uint32_t used;
    if (hb >= 128) {
        nw = hb - 127;
        const uint32_t bytes = (uint32_t)(nw + 1) / 2;
        if (n < 1 + bytes) return;
        for (int i = 0; i < nw; i++) {
            const uint8_t b = ldu8(src + 1 + i / 2);
            wts[i] = (i & 1) ? (b & 15) : (b >> 4);
        }
        used = 1 + bytes;
    } else {
        const uint32_t clen = hb;
        if (n < 1 + clen || clen < 2) return;
        int nsym, log;
        const uint32_t hsz = z_fse_header(src + 1, clen, 12, 6, H.b.norm[g], &nsym, &log);
        if (!hsz || hsz >= clen) return;
        ZFse* t = H.b.fse[g];
        if (!z_fse_build2(t, H.b.next[g], H.b.norm[g], nsym, log)) return;
        const uint8_t* bs = src + 1 + hsz;
        const uint32_t bn = clen - hsz;
        if (ldu8(bs + bn - 1) == 0) return;
        int64_t bitpos = (int64_t)(bn - 1) * 8 + (31 - __clz((int)ldu8(bs + bn - 1)));
        uint32_t s1 = z_peek(bs, bitpos, log);
        bitpos -= log;
        uint32_t s2 = z_peek(bs, bitpos, log);
        bitpos -= log;
        nw = 0;
        for (;;) {
            if (nw >= 254) return;
            wts[nw++] = t[s1].symbol;
            if (bitpos < t[s1].nbits) {
                wts[nw++] = t[s2].symbol;
                break;
            }
            {
                const uint32_t nb = t[s1].nbits;
                s1 = t[s1].base + z_peek(bs, bitpos, (int)nb);
                bitpos -= nb;
            }
            wts[nw++] = t[s2].symbol;
            if (bitpos < t[s2].nbits) {
                wts[nw++] = t[s1].symbol;
                break;
            }
            {
                const uint32_t nb = t[s2].nbits;
                s2 = t[s2].base + z_peek(bs, bitpos, (int)nb);
                bitpos -= nb;
            }
        }
        used = 1 + clen;
    }
    if (n < used + 6 || nw < 1) return;
    H.str0[g] = used;
    H.bits[g] = (uint32_t)nw;   // (weights decoded: z_wave_huf_fill turns this into the table's code bits)
}
// The tables of the group's frames from their weights, by the whole wave, frame after frame: class sizes by ballots, the
// symbols in (weight, symbol) order into a 256-byte list, then every lane fills 8 of the table's entries (class of the
// code by a compare chain over <= 11 class starts, symbol from the list).  One lane filling its frame's table weight by
// weight (11 passes over 256 symbols + 512 stores) took 0.9 ms per group.
__device__ inline void z_wave_huf_fill(ZHufLanes& H, uint32_t ng) {
    const uint32_t lane = threadIdx.x & 63;
    for (uint32_t g = 0; g < ng; g++) {
        const uint32_t nw = H.bits[g];   // (uniform)
        if (!nw) continue;
        wave_sync();
        uint32_t w[4];
        bool bad = false;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t sy = lane + 64 * j;
            w[j] = sy < nw ? (uint32_t)H.b.wts[g][sy] : 0u;
            bad = bad || w[j] > 11;
        }
        uint32_t cw[12], total = 0;
#pragma unroll
        for (uint32_t wv = 1; wv < 12; wv++) {
            uint32_t c = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) c += (uint32_t)__popcll(__ballot(w[j] == wv));
            cw[wv] = c;
            total += c << (wv - 1);
        }
        uint32_t max_bits = 0, lw = 0;
        bool okf = !__ballot(bad) && total != 0;
        if (okf) {
            max_bits = 32u - (uint32_t)__clz((int)total);
            const uint32_t left = (1u << max_bits) - total;
            okf = left != 0 && !(left & (left - 1)) && max_bits <= ZH_MAXBITS;
            lw = okf ? 32u - (uint32_t)__clz((int)left) : 0u;   // weight of the last symbol (implied)
        }
        if (!okf) {
            if (lane == 0) H.bits[g] = 0;
            continue;
        }
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (lane + 64 * j == nw) w[j] = lw;
        uint32_t start[12], before[12];
        {
            uint32_t code = 0, nb = 0;
#pragma unroll
            for (uint32_t wv = 1; wv < 12; wv++) {
                const uint32_t c = cw[wv] + (wv == lw ? 1u : 0u);
                cw[wv] = c;
                start[wv] = code;
                before[wv] = nb;
                code += c << (wv - 1);
                nb += c;
            }
        }
        uint8_t* sl = (uint8_t*)H.b.fse[g];   // (the frame's FSE table is done with)
#pragma unroll
        for (uint32_t wv = 1; wv < 12; wv++) {
            if (!cw[wv]) continue;   // (uniform)
            uint32_t base = before[wv];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint64_t m = __ballot(w[j] == wv);
                if (w[j] == wv) sl[base + lane_rank(m)] = (uint8_t)(lane + 64 * j);
                base += (uint32_t)__popcll(m);
            }
        }
        wave_sync();
        uint16_t* tab = H.tab[g];
#pragma unroll
        for (uint32_t k = 0; k < (1u << ZH_MAXBITS) / 64; k++) {
            const uint32_t c = lane + 64 * k;
            if (c >= (1u << max_bits)) break;
            uint32_t cls = 1, st = 0, bf = 0;
#pragma unroll
            for (uint32_t wv = 1; wv < 12; wv++) {
                if (cw[wv] && c >= start[wv]) {
                    cls = wv;
                    st = start[wv];
                    bf = before[wv];
                }
            }
            tab[c] = (uint16_t)((uint32_t)sl[bf + ((c - st) >> (cls - 1))] | ((max_bits + 1 - cls) << 8));
        }
        if (lane == 0) H.bits[g] = max_bits;
    }
    wave_sync();
}

#ifndef ZTL   // phase stamps of z_wave_huf_streams (scripts/micro/inflate_timeline.hip defines them)
#define ZTL_BEGIN
#define ZTL(p)
#endif
// The streams of up to 16 staged frames, LANE PER STREAM (lane 4 g + j = stream j of frame g), in rounds of 32 symbols.
// A lane reading its own stream from HBM is one cache line per lane and load, and the same again for its 8-byte stores:
// 2600 cycles per symbol on C5's leaves.  Here the wave moves the data.  Every stream has a 128-byte RING in LDS, addressed
// by the byte offset inside the stream (streams are read from their end downwards); a round reads at most 32 x 9 bits + the
// 64 bits of look-ahead = 44 bytes of it.  At the START of a round the wave issues the loads of the 16-byte chunks the round
// AFTER it may need (four lanes per stream, at most four chunks each), every lane decodes its 32 symbols into a 32-byte
// row of `ob`, and only then the chunks are written into the rings and the rows stored (two lanes per stream): the HBM
// latency of the input hides behind the symbol loop.  `act`: the lane has a stream; returns its verdict (all bits consumed,
// nothing beyond).
constexpr uint32_t ZH_ROUND = 32;
__device__ inline bool z_wave_huf_streams(ZHufLanes& H, uint32_t g, uint32_t mb, const uint8_t* sb_, uint32_t sn, uint8_t* dst, uint32_t outn, bool act) {
    const uint32_t lane = threadIdx.x & 63;
    ZHufStage& S = H.s;
    uint32_t* ob32 = (uint32_t*)H.ob;          // [64][8]: the round's output, u32 m of stream s in slot (m + s) & 7
    uint32_t* sn_pub = ob32 + 64 * 8;          // (the second half of ob: the streams' sizes while the rings are primed)
    int32_t left = 0;
    if (act) {
        const uint8_t lastb = sn ? ldu8(sb_ + sn - 1) : (uint8_t)0;
        if (lastb == 0) {
            act = false;
        } else {
            left = (int32_t)sn * 8 - (int32_t)(8 - (31 - (uint32_t)__clz((int)lastb)));   // bits above the end mark, and the mark, are gone
        }
    }
    const bool had = act;
    bool ok = true;
    uint32_t done = 0;
    // ---- prime the rings: the 128 bytes below the 16-byte border at or above the stream's top
    const int32_t hi0 = (left + 7) >> 3;                   // bytes that still hold unread bits
    const int32_t top = (hi0 + 15) & ~15;
    S.ptr[lane] = (uint64_t)(uintptr_t)sb_;
    S.lo[lane] = act ? top - 128 : (int32_t)0x80000000;
    sn_pub[lane] = sn;
    wave_sync();
#pragma unroll
    for (uint32_t p = 0; p < 8; p++) {
        const uint32_t s = 8 * p + (lane >> 3), c = lane & 7;
        const int32_t lo = S.lo[s];
        if (lo == (int32_t)0x80000000) continue;
        const uint8_t* src = (const uint8_t*)(uintptr_t)S.ptr[s];
        const int32_t b = lo + 16 * (int32_t)c, sns = (int32_t)sn_pub[s];
        if (b + 16 <= 0 || b >= sns) continue;
        uint32_t w4[4] = {0, 0, 0, 0};
        if (b >= 0 && b + 16 <= sns) {
            const u32x4 v = ldu128(src + b);
            w4[0] = v.x; w4[1] = v.y; w4[2] = v.z; w4[3] = v.w;
        } else {   // the stream starts or ends inside this chunk: the bytes outside read as zeros
            for (int32_t k = 0; k < 16; k++)
                if (b + k >= 0 && b + k < sns) w4[k >> 2] |= (uint32_t)ldu8(src + b + k) << (8 * (k & 3));
        }
        const uint32_t w0 = (uint32_t)(b >> 2);   // (b >= -15: a chunk that starts in front of the stream holds its words 0 ..)
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (b + 4 * q >= 0) S.ring[s][(w0 + q + s) & 31] = w4[q];
    }
    wave_sync();
    int32_t lo_loaded = top - 128 > 0 ? top - 128 : 0;
    const uint32_t* ring = S.ring[lane];
    auto word = [&](int32_t w) -> uint32_t { return w >= 0 ? ring[((uint32_t)w + lane) & 31] : 0u; };
    // the bit buffer: the next bits of the stream top-aligned, cnt of them valid (> 32 after every refill); wi = next word
    uint64_t bb = 0;
    uint32_t cnt = 0;
    int32_t wi = -1;
    if (act) {
        const uint32_t kb = (uint32_t)((hi0 - 1) & 3) + 1;   // bytes of the top word that belong to the stream
        const int32_t wt = (hi0 - 1) >> 2;
        const uint32_t u0 = (uint32_t)(8 * hi0 - left);      // unused bits of the top byte (0 .. 7)
        bb = ((uint64_t)word(wt) << (32 + 8 * (4 - kb))) << u0;
        cnt = 8 * kb - u0;
        wi = wt - 1;
    }
    const uint16_t* tab = H.tab[g];
    const uint32_t sh = 32 - mb;
    ZTL_BEGIN
    while (__ballot(act && ok && done < outn)) {
        const bool run = act && ok && done < outn;
        ZTL(0);
        // ---- the chunks the NEXT round may read: down to word wi - 20 (this round reads down to wi - 10 at most)
        int32_t lo_new = lo_loaded;
        if (run) {
            const int32_t need = (4 * (wi - 20)) & ~15;
            lo_new = need < 0 ? 0 : need;
            if (lo_new > lo_loaded) lo_new = lo_loaded;
        }
        S.lo[lane] = lo_loaded;
        S.nch[lane] = (uint8_t)((lo_loaded - lo_new) >> 4);
        wave_sync();
        u32x4 pre[4];
        uint32_t pre_m = 0;
#pragma unroll
        for (uint32_t p = 0; p < 4; p++) {
            const uint32_t s = 16 * p + (lane >> 2), c = lane & 3;
            pre[p] = u32x4{0, 0, 0, 0};
            if (c < S.nch[s]) {   // (below what is loaded: inside the stream)
                pre[p] = ldu128((const uint8_t*)(uintptr_t)S.ptr[s] + (S.lo[s] - 16 * (int32_t)(c + 1)));
                pre_m |= 1u << p;
            }
        }
        ZTL(1);
        // ---- 32 symbols
        uint32_t n_this = 0;
        if (run) {
            n_this = min(ZH_ROUND, outn - done);
            uint32_t out32 = 0;
            uint32_t* orow = ob32 + 8 * lane;
            for (uint32_t i = 0; i < n_this; i++) {
                if (cnt <= 32) {
                    bb |= (uint64_t)word(wi--) << (32 - cnt);
                    cnt += 32;
                }
                const uint32_t e = tab[(uint32_t)(bb >> 32) >> sh];
                const uint32_t len = e >> 8;
                bb <<= len;
                cnt -= len;
                left -= (int32_t)len;
                out32 = (out32 >> 8) | ((e & 255u) << 24);
                if ((i & 3) == 3) orow[((i >> 2) + lane) & 7] = out32;
            }
            if (n_this & 3) orow[((n_this >> 2) + lane) & 7] = out32 >> (8 * (4 - (n_this & 3)));
            if (left < 0) ok = false;
        }
        ZTL(2);
        wave_sync();   // (every lane is done with the ring words the new chunks replace: they lie 128 bytes higher)
        // ---- the chunks into the rings, the rows to HBM
#pragma unroll
        for (uint32_t p = 0; p < 4; p++) {
            if (!((pre_m >> p) & 1)) continue;
            const uint32_t s = 16 * p + (lane >> 2), c = lane & 3;
            const uint32_t w0 = (uint32_t)((S.lo[s] - 16 * (int32_t)(c + 1)) >> 2);
            S.ring[s][(w0 + s) & 31] = pre[p].x;
            S.ring[s][(w0 + 1 + s) & 31] = pre[p].y;
            S.ring[s][(w0 + 2 + s) & 31] = pre[p].z;
            S.ring[s][(w0 + 3 + s) & 31] = pre[p].w;
        }
        lo_loaded = lo_new;
        S.cnt[lane] = (uint8_t)(ok ? n_this : 0u);
        wave_sync();
        S.ptr[lane] = (uint64_t)(uintptr_t)(dst + done);   // (the loads through ptr are done: it now says where the rows go)
        wave_sync();
#pragma unroll
        for (uint32_t p = 0; p < 2; p++) {
            const uint32_t s = 32 * p + (lane >> 1), c = lane & 1;
            const uint32_t cs = S.cnt[s];
            if (16 * c >= cs) continue;
            uint8_t* d = (uint8_t*)(uintptr_t)S.ptr[s] + 16 * c;
            const uint32_t* orow = ob32 + 8 * s;
            const uint32_t v0 = orow[(4 * c + s) & 7], v1 = orow[(4 * c + 1 + s) & 7], v2 = orow[(4 * c + 2 + s) & 7], v3 = orow[(4 * c + 3 + s) & 7];
            if (16 * c + 16 <= cs) {
                stu128(d, u32x4{v0, v1, v2, v3});
            } else {
                const uint32_t vv[4] = {v0, v1, v2, v3};
                for (uint32_t k = 0; k < cs - 16 * c; k++) *(gptr)(d + k) = (uint8_t)(vv[k >> 2] >> (8 * (k & 3)));
            }
        }
        done += n_this;
        wave_sync();
        S.ptr[lane] = (uint64_t)(uintptr_t)sb_;
        ZTL(3);
    }
    return had && ok && done == outn && left == 0;
}
// One stream sb_[0, sn) -> outn bytes at dst, by one lane; table: 2^mb entries.  Returns false on a malformed stream.
__device__ inline bool z_lane_huf_stream(const uint16_t* tab, uint32_t mb, const uint8_t* sb_, uint32_t sn, uint8_t* dst, uint32_t outn) {
    if (sn == 0) return false;
    const uint8_t lastb = ldu8(sb_ + sn - 1);
    if (lastb == 0) return false;
    // window: W0 = stream bits [top - 64, top), W1 = the 64 below; `used` bits of W0 are consumed (the end mark first)
    auto load8 = [&](int32_t byte_off) -> uint64_t {   // bytes [byte_off, byte_off + 8) of the stream; below its start: 0
        if (byte_off >= 0) return ldu64(sb_ + byte_off);
        uint64_t v = 0;
        for (int32_t k = byte_off < -8 ? 8 : -byte_off; k < 8; k++) v |= (uint64_t)ldu8(sb_ + byte_off + k) << (8 * k);
        return v;
    };
    int32_t off = (int32_t)sn - 8;             // byte offset of W0
    uint64_t W0 = load8(off), W1 = load8(off - 8), W2 = load8(off - 16);
    uint32_t used = 8 - (31 - (uint32_t)__clz((int)lastb));   // bits above the end mark, and the mark itself
    int32_t left = (int32_t)sn * 8 - (int32_t)used;           // stream bits not consumed yet
    uint64_t acc = 0;
    uint32_t i = 0;
    for (; i < outn; i++) {
        if (used >= 64) {
            W0 = W1;
            W1 = W2;
            off -= 8;
            W2 = load8(off - 16);            // (needed 128 bits from now: its latency hides behind ~20 symbols)
            used -= 64;
        }
        const uint64_t top = used ? (W0 << used) | (W1 >> (64 - used)) : W0;
        const uint32_t e = tab[(uint32_t)(top >> (64 - mb))];
        const uint32_t len = e >> 8;
        used += len;
        left -= (int32_t)len;
        acc |= (uint64_t)(e & 255) << (8 * (i & 7));
        if ((i & 7) == 7) {
            stu64(dst + i - 7, acc);
            acc = 0;
        }
    }
    for (uint32_t k = outn & ~7u; k < outn; k++) *(gptr)(dst + k) = (uint8_t)(acc >> (8 * (k & 7)));
    return left == 0;
}

// sets one of the three sequence tables according to its compression mode; returns bytes consumed
// from `src` (0 is a valid answer), or 0xFFFFFFFF on error
__device__ inline uint32_t z_seq_table(ZWork* wk, int mode, ZFse* t, uint32_t* log_out, uint32_t* have,
                                       const int16_t* def, int def_n, int def_log, int max_sym, int max_log,
                                       const uint8_t* src, uint32_t n) {
    if (mode == 0) {
        ZFse* pre = def_n == 36 ? wk->pre_ll : def_n == 29 ? wk->pre_of : wk->pre_ml;
        const uint32_t bit = def_n == 36 ? 1u : def_n == 29 ? 2u : 4u;
        if (!(wk->pre_built & bit)) {
            for (int i = 0; i < def_n; i++) wk->norm[i] = def[i];
            if (!z_fse_build(pre, wk, wk->norm, def_n, def_log)) return 0xFFFFFFFFu;
            wk->pre_built |= bit;
        }
        for (int i = 0; i < (1 << def_log); i++) t[i] = pre[i];
        *log_out = (uint32_t)def_log;
        *have = 1;
        return 0;
    }
    if (mode == 1) {
        if (n < 1 || src[0] > max_sym) return 0xFFFFFFFFu;
        t[0].symbol = src[0];
        t[0].nbits = 0;
        t[0].base = 0;
        *log_out = 0;
        *have = 1;
        return 1;
    }
    if (mode == 2) {
        int nsym, log;
        const uint32_t used = z_fse_header(src, n, max_sym, max_log, wk->norm, &nsym, &log);
        if (!used) return 0xFFFFFFFFu;
        if (!z_fse_build(t, wk, wk->norm, nsym, log)) return 0xFFFFFFFFu;
        *log_out = (uint32_t)log;
        *have = 1;
        return used;
    }
    return *have ? 0 : 0xFFFFFFFFu;
}

// Decodes one frame.  All 64 lanes call it with identical arguments; returns bytes produced
// (wk->err != 0 on failure).  `lit` = a 128 KiB + 32 literal buffer owned by this wave.
__device__ uint32_t zstd_inflate_wave(const uint8_t* src, uint32_t n, uint8_t* dst, uint32_t out_len, ZWork* wk,
                                      uint8_t* lit, const uint64_t* recs = nullptr /* k_zstd_seq_lanes' records of this frame */) {
    const int lane = threadIdx.x & 63;
    auto wsync = []() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    if (lane == 0) wk->err = 0;
    for (int i = lane; i < 53; i += 64) {
        wk->t_mlbase[i] = Z_ML_BASE[i];
        wk->t_mlbits[i] = Z_ML_BITS[i];
        if (i < 36) {
            wk->t_llbase[i] = Z_LL_BASE[i];
            wk->t_llbits[i] = Z_LL_BITS[i];
        }
    }
    wsync();
    uint32_t ip = 0, op = 0;
    LZP_BEGIN
#define ZERR(c)                        \
    do {                               \
        if (lane == 0) wk->err = (c);  \
        wsync();                       \
        return 0;                      \
    } while (0)
    LzSeqExec ex(wk->ring, dst);   // sequences run through an LDS output ring; everything else writes HBM directly
    // One or more frames back to back — ZSTD_decompress, which zstd::bulk::decompress_to_buffer (basic.rs:93-97) ends in,
    // decodes "any number of frames concatenated"; this library's encoder writes the 16 KiB pieces of a large buffer as
    // frames of their own so that they decode in parallel (k_parse queues one job per frame; a buffer whose frames were
    // not split arrives here whole).
    bool first_frame = true;
    for (;;) {   // frames
    if (lane == 0) {   // entropy tables and repeat offsets do not carry over between frames
        wk->have_ll = wk->have_of = wk->have_ml = wk->have_huf = 0;
        wk->rep[0] = 1;
        wk->rep[1] = 4;
        wk->rep[2] = 8;
    }
    wsync();
    if (!first_frame) ex.restart(op);
    first_frame = false;
    const uint32_t frame_op0 = op;
    // frame header (all lanes, uniform)
    for (;;) {  // skippable frames
        if (n - ip < 4) ZERR(1);
        const uint32_t magic = ldu32(src + ip);
        if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) {
            if (n - ip < 8) ZERR(2);
            const uint32_t sz = ldu32(src + ip + 4);
            if (n - ip - 8 < sz) ZERR(3);
            ip += 8 + sz;
            continue;
        }
        if (magic != 0xFD2FB528u) ZERR(4);
        ip += 4;
        break;
    }
    if (ip >= n) ZERR(5);
    const uint8_t fhd = src[ip++];
    const int fcs_flag = fhd >> 6, single = (fhd >> 5) & 1, checksum = (fhd >> 2) & 1, did = fhd & 3;
    if (fhd & 0x08) ZERR(6);
    if (!single) ip += 1;
    const int did_bytes = did == 0 ? 0 : did == 1 ? 1 : did == 2 ? 2 : 4;
    if (did) {
        uint32_t id = 0;
        for (int i = 0; i < did_bytes; i++) id |= (uint32_t)src[ip + i] << (8 * i);
        if (id) ZERR(7);  // dictionaries are never used by the reference
        ip += did_bytes;
    }
    const int fcs_bytes = fcs_flag == 0 ? (single ? 1 : 0) : (1 << fcs_flag);
    uint64_t fcs = 0;
    if (fcs_bytes) {
        if (n - ip < (uint32_t)fcs_bytes) ZERR(8);
        for (int i = 0; i < fcs_bytes; i++) fcs |= (uint64_t)src[ip + i] << (8 * i);
        if (fcs_bytes == 2) fcs += 256;
        ip += fcs_bytes;
    }
    for (;;) {  // blocks
        if (n - ip < 3) ZERR(9);
        const uint32_t bh = (uint32_t)src[ip] | ((uint32_t)src[ip + 1] << 8) | ((uint32_t)src[ip + 2] << 16);
        ip += 3;
        const bool last = bh & 1;
        const int btype = (bh >> 1) & 3;
        const uint32_t bsize = bh >> 3;
        if (btype == 0) {
            if (n - ip < bsize || out_len - op < bsize) ZERR(10);
            wave_copy_g2g(dst + op, src + ip, bsize);
            ip += bsize;
            op += bsize;
            ex.restart(op);
        } else if (btype == 1) {
            if (n - ip < 1 || out_len - op < bsize) ZERR(11);
            const uint8_t v = src[ip++];
            for (uint32_t i = lane; i < bsize; i += 64) dst[op + i] = v;
            op += bsize;
            ex.restart(op);
        } else if (btype == 2) {
            if (bsize > 128 * 1024 || n - ip < bsize) ZERR(12);
            const uint8_t* bs = src + ip;
            uint32_t bp = 0;
            // ---- literals section
            if (bsize < 1) ZERR(13);
            const uint8_t b0 = bs[bp++];
            const int ltype = b0 & 3, sf = (b0 >> 2) & 3;
            uint32_t regen = 0, csize = 0;
            int streams = 1;
            const uint8_t* litp = lit;
            if (ltype == 0 || ltype == 1) {
                if (sf == 0 || sf == 2) {
                    regen = b0 >> 3;
                } else if (sf == 1) {
                    regen = (b0 >> 4) | ((uint32_t)bs[bp] << 4);
                    bp += 1;
                } else {
                    regen = (b0 >> 4) | ((uint32_t)bs[bp] << 4) | ((uint32_t)bs[bp + 1] << 12);
                    bp += 2;
                }
                if (ltype == 0) {
                    if (bsize - bp < regen) ZERR(14);
                    litp = bs + bp;  // raw literals are used in place
                    bp += regen;
                } else {
                    if (regen > 128 * 1024) ZERR(15);
                    const uint8_t v = bs[bp++];
                    for (uint32_t i = lane; i < regen; i += 64) lit[i] = v;
                    wsync();
                }
            } else {
                if (sf == 0 || sf == 1) {
                    const uint32_t v = (b0 >> 4) | ((uint32_t)bs[bp] << 4) | ((uint32_t)bs[bp + 1] << 12);
                    bp += 2;
                    regen = v & 0x3FF;
                    csize = v >> 10;
                    streams = sf == 0 ? 1 : 4;
                } else if (sf == 2) {
                    const uint32_t v = (b0 >> 4) | ((uint32_t)bs[bp] << 4) | ((uint32_t)bs[bp + 1] << 12) |
                                       ((uint32_t)bs[bp + 2] << 20);
                    bp += 3;
                    regen = v & 0x3FFF;
                    csize = v >> 14;
                    streams = 4;
                } else {
                    const uint64_t v = (b0 >> 4) | ((uint64_t)bs[bp] << 4) | ((uint64_t)bs[bp + 1] << 12) |
                                       ((uint64_t)bs[bp + 2] << 20) | ((uint64_t)bs[bp + 3] << 28);
                    bp += 4;
                    regen = (uint32_t)(v & 0x3FFFF);
                    csize = (uint32_t)(v >> 18);
                    streams = 4;
                }
                if (bsize - bp < csize || regen > 128 * 1024) ZERR(16);
                const uint8_t* ls = bs + bp;
                uint32_t lleft = csize;
                bp += csize;
                if (ltype == 2) {
                    if (lane == 0) {
                        const uint32_t used = z_huf_read(wk, ls, lleft);
                        if (!used) wk->err = 17;
                        wk->nbatch = used;
                    }
                    wsync();
                    if (wk->err) return 0;
                    ls += wk->nbatch;
                    lleft -= wk->nbatch;
                } else if (!wk->have_huf) {
                    ZERR(18);
                }
                if (streams == 1) {
                    if (lane == 0 && !z_huf_stream(wk, ls, lleft, lit, regen)) wk->err = 19;
                } else {
                    if (lleft < 6) ZERR(20);
                    const uint32_t s1 = ls[0] | ((uint32_t)ls[1] << 8), s2 = ls[2] | ((uint32_t)ls[3] << 8),
                                   s3 = ls[4] | ((uint32_t)ls[5] << 8);
                    if (6 + s1 + s2 + s3 > lleft) ZERR(21);
                    const uint32_t s4 = lleft - 6 - s1 - s2 - s3;
                    const uint32_t per = (regen + 3) / 4;
                    if (per * 3 > regen) ZERR(22);
                    const uint8_t* q = ls + 6;
                    if (lane < 4) {  // the four streams decode on four lanes
                        const uint32_t so[4] = {0, s1, s1 + s2, s1 + s2 + s3};
                        const uint32_t sn[4] = {s1, s2, s3, s4};
                        const uint32_t outn = lane < 3 ? per : regen - 3 * per;
                        if (!z_huf_stream(wk, q + so[lane], sn[lane], lit + (uint32_t)lane * per, outn)) wk->err = 23;
                    }
                }
                wsync();
                if (wk->err) return 0;
            }
            // ---- sequences section
            if (bsize - bp < 1) ZERR(24);
            uint32_t nseq;
            {
                const uint8_t s0 = bs[bp++];
                if (s0 < 128) {
                    nseq = s0;
                } else if (s0 < 255) {
                    nseq = ((uint32_t)(s0 - 128) << 8) + bs[bp];
                    bp += 1;
                } else {
                    nseq = (uint32_t)bs[bp] + ((uint32_t)bs[bp + 1] << 8) + 0x7F00;
                    bp += 2;
                }
            }
            uint32_t lit_pos = 0;
            LZP(20);
            if (nseq > 0 && recs) {
                // the sequences were decoded by k_zstd_seq_lanes (lane per frame): lane k takes record k of the batch
                for (uint32_t done = 0; done < nseq; done += 64) {
                    const uint32_t nb = min(64u, nseq - done);
                    const bool have = (uint32_t)lane < nb;
                    const uint64_t r = have ? gld64(recs + done + lane) : (1ull << 36);
                    const uint32_t llen = have ? (uint32_t)(r & 0x3FFFF) : 0u, mlen = have ? (uint32_t)((r >> 18) & 0x3FFFF) : 0u;
                    const uint32_t off = (uint32_t)(r >> 36);
                    const uint32_t lsum = wave_scan_dpp(llen), osum = wave_scan_dpp(llen + mlen);
                    const bool bad = have && ((uint64_t)lit_pos + lsum > regen || (uint64_t)op + osum > out_len || off == 0 ||
                                              off > op + osum - mlen);
                    if (__ballot(bad)) ZERR(30);
                    const uint32_t lit_total = rdlane(lsum, 63);
                    op += ex.run(nb, llen, mlen, off, litp + lit_pos, op);
                    lit_pos += lit_total;
                    wsync();
                }
                recs += nseq;
            } else if (nseq > 0) {
                if (lane == 0) {
                    const uint8_t modes = bs[bp];
                    uint32_t q = bp + 1;
                    uint32_t u = (modes & 3) ? 0xFFFFFFFFu : 0;
                    if (u == 0) u = z_seq_table(wk, (modes >> 6) & 3, wk->ll, &wk->ll_log, &wk->have_ll, Z_LL_DEFAULT, 36, 6, 35, 9, bs + q, bsize - q);
                    if (u != 0xFFFFFFFFu) {
                        q += u;
                        u = z_seq_table(wk, (modes >> 4) & 3, wk->of, &wk->of_log, &wk->have_of, Z_OF_DEFAULT, 29, 5, 31, 8, bs + q, bsize - q);
                    }
                    if (u != 0xFFFFFFFFu) {
                        q += u;
                        u = z_seq_table(wk, (modes >> 2) & 3, wk->ml, &wk->ml_log, &wk->have_ml, Z_ML_DEFAULT, 53, 6, 52, 9, bs + q, bsize - q);
                    }
                    if (u == 0xFFFFFFFFu)
                        wk->err = 25;
                    else
                        wk->nbatch = q + u;
                }
                wsync();
                if (wk->err) return 0;
                {   // extras per state (all lanes); a state whose symbol is outside the code range makes the block invalid
                    bool bad = false;
                    for (uint32_t st = lane; st < (1u << wk->ll_log); st += 64) {
                        const uint32_t c = wk->ll[st].symbol;
                        if (c > 35) { bad = true; continue; }
                        wk->xll[st] = (uint32_t)wk->t_llbits[c] | (wk->t_llbase[c] << 8);
                    }
                    for (uint32_t st = lane; st < (1u << wk->ml_log); st += 64) {
                        const uint32_t c = wk->ml[st].symbol;
                        if (c > 52) { bad = true; continue; }
                        wk->xml[st] = (uint32_t)wk->t_mlbits[c] | (wk->t_mlbase[c] << 8);
                    }
                    for (uint32_t st = lane; st < (1u << wk->of_log); st += 64)
                        if (wk->of[st].symbol > 31) bad = true;
                    if (__ballot(bad)) ZERR(27);
                    wsync();
                }
                bp = wk->nbatch;
                const uint8_t* sb_ = bs + bp;
                const uint32_t sn_ = bsize - bp;
                if (sn_ == 0 || sb_[sn_ - 1] == 0) ZERR(26);
                // lane 0 keeps the bit reader and the three FSE states; batches of 64 sequences
                int64_t bitpos = (int64_t)(sn_ - 1) * 8 + (31 - __clz((int)sb_[sn_ - 1]));
                uint32_t sl = 0, so = 0, sm = 0;
                // the stream is read through an LDS window: `wp` points at the window so that wp[x] == sb_[x] for the bytes in it
                auto refill_window = [&](int64_t bp_) {
                    const uint32_t hi_byte = (uint32_t)min((int64_t)sn_, (bp_ >> 3) + 9);
                    const uint32_t lo_byte = hi_byte > ZBW ? (hi_byte - ZBW) & ~15u : 0u;
                    wsync();
                    for (uint32_t k = lane * 16; k < ZBW + 16 && lo_byte + k < sn_; k += 64 * 16)
                        for (uint32_t b = 0; b < 16 && lo_byte + k + b < sn_; b++) wk->bw[k + b] = sb_[lo_byte + k + b];
                    if (lane == 0) wk->bw_lo = lo_byte;
                    wsync();
                };
                refill_window(bitpos);
                LZP(21);
                // The sequence loop is a serial chain (three FSE states and one bit position), so it runs WAVE-UNIFORM: every
                // lane computes the same values, the LDS reads go through v_readfirstlane, and the whole chain is scalar
                // instructions (a lane-0-only loop is the same chain on the vector unit: ~4x the cycles per step).  Lane k
                // keeps sequence k of the batch in registers: no LDS staging between decode and execute.
                auto rfl = [](uint32_t v) -> uint32_t { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
                {
                    const uint8_t* wp = wk->bw - wk->bw_lo;
                    const uint32_t lll = rfl(wk->ll_log), ofl = rfl(wk->of_log), mll = rfl(wk->ml_log);
                    sl = rfl(z_peek(wp, bitpos, (int)lll));
                    bitpos -= lll;
                    so = rfl(z_peek(wp, bitpos, (int)ofl));
                    bitpos -= ofl;
                    sm = rfl(z_peek(wp, bitpos, (int)mll));
                    bitpos -= mll;
                }
                uint32_t r0 = rfl(wk->rep[0]), r1 = rfl(wk->rep[1]), r2 = rfl(wk->rep[2]);
                for (uint32_t done = 0; done < nseq; done += 64) {
                    const uint32_t nb = min(64u, nseq - done);
                    {   // a batch reads at most 64 * (16 + 16 + 32 + 27) bits = 728 bytes below the current position
                        const int64_t need_lo = (bitpos >> 3) - 760;
                        if (need_lo < (int64_t)wk->bw_lo && wk->bw_lo > 0) refill_window(bitpos);
                    }
                    uint32_t my_ll = 0, my_ml = 0, my_off = 1;
                    int32_t err = 0;
                    {
                        const uint32_t* w32 = (const uint32_t*)wk->bw;
                        const int64_t win_lo_bits = (int64_t)rfl(wk->bw_lo) * 8;
                        for (uint32_t k = 0; k < nb; k++) {
                            // ONE LDS round trip per sequence: the three FSE entries of the current states, the extras of the two
                            // length codes (per state: xll / xml) and the 64 stream bits below bitpos (three aligned dwords of
                            // the window + a funnel shift; a take() is then two shifts)
                            const uint32_t e_of = rfl(*(const uint32_t*)&wk->of[so]), e_ml = rfl(*(const uint32_t*)&wk->ml[sm]),
                                           e_ll = rfl(*(const uint32_t*)&wk->ll[sl]);
                            const uint32_t x_ml = rfl(wk->xml[sm]), x_ll = rfl(wk->xll[sl]);
                            auto load_top = [&]() -> uint64_t {
                                const int64_t lo = bitpos - 64;
                                const int64_t rel = (lo < 0 ? 0 : lo) - win_lo_bits;
                                if (rel < 0) {           // (malformed stream / stale window: never on a valid one)
                                    err = 29;
                                    return 0ull;
                                }
                                const uint32_t idx = (uint32_t)rel >> 5, sh = (uint32_t)rel & 31;
                                const uint32_t d0 = rfl(w32[idx]), d1 = rfl(w32[idx + 1]), d2 = rfl(w32[idx + 2]);
                                uint64_t v = (((uint64_t)d1 << 32) | d0) >> sh;
                                if (sh) v |= (uint64_t)d2 << (64 - sh);
                                if (lo < 0) v = bitpos > 0 ? v << (uint32_t)(-lo) : 0ull;   // fewer than 64 bits left: low bits read as 0
                                return v;
                            };
                            uint64_t C = load_top();
                            auto take = [&](uint32_t nbits) -> uint32_t {
                                if (nbits == 0) return 0u;
                                const uint32_t v = (uint32_t)(C >> (64 - nbits));
                                C <<= nbits;
                                bitpos -= nbits;
                                return v;
                            };
                            const uint32_t ofc = e_of & 255;
                            const uint32_t mlbits = x_ml & 255, llbits = x_ll & 255;
                            const uint32_t mlbase = x_ml >> 8, llbase = x_ll >> 8;
                            const uint64_t ofv = ((uint64_t)1 << ofc) + take(ofc);
                            const uint32_t mlen = mlbase + take(mlbits);
                            const uint32_t llen = llbase + take(llbits);
                            uint32_t offset;
                            if (ofv > 3) {
                                offset = (uint32_t)(ofv - 3);
                                r2 = r1;
                                r1 = r0;
                                r0 = offset;
                            } else {
                                uint32_t idx = (uint32_t)ofv - 1;
                                if (llen == 0) idx++;
                                if (idx == 0) {
                                    offset = r0;
                                } else {
                                    offset = idx == 1 ? r1 : idx == 2 ? r2 : r0 - 1;
                                    if (offset == 0) {
                                        err = 28;
                                        break;
                                    }
                                    if (idx > 1) r2 = r1;
                                    r1 = r0;
                                    r0 = offset;
                                }
                            }
                            if (done + k + 1 < nseq) {   // new states: LL, ML, OF (entry = symbol | nbits << 8 | base << 16)
                                const uint32_t nl = (e_ll >> 8) & 255, nm = (e_ml >> 8) & 255, no = (e_of >> 8) & 255;
                                // the container still holds 64 - (offset + length bits) stream bits: reload only when the state bits
                                // do not fit any more (offsets beyond 2^20 or long extra-bit runs)
                                if (ofc + mlbits + llbits + nl + nm + no > 64) C = load_top();
                                sl = (e_ll >> 16) + take(nl);
                                sm = (e_ml >> 16) + take(nm);
                                so = (e_of >> 16) + take(no);
                            }
                            if (bitpos < 0) err = 29;
                            if (err) break;
                            if ((uint32_t)lane == k) {   // lane k keeps sequence k
                                my_ll = llen;
                                my_ml = mlen;
                                my_off = offset;
                            }
                        }
                        if (done + nb == nseq && !err && bitpos != 0) err = 29;
                    }
                    LZP(22);
                    if (err) ZERR(err);
                    // execute the batch through the LDS output ring (sb_lz4.h LzSeqExec): lane k holds sequence k
                    {
                        const bool have = (uint32_t)lane < nb;
                        const uint32_t llen = have ? my_ll : 0u, mlen = have ? my_ml : 0u;
                        const uint32_t off = have ? my_off : 1u;
                        const uint32_t lsum = wave_scan_dpp(llen), osum = wave_scan_dpp(llen + mlen);
                        const bool bad = have && ((uint64_t)lit_pos + lsum > regen || (uint64_t)op + osum > out_len || off == 0 ||
                                                  off > op + osum - mlen);
                        if (__ballot(bad)) ZERR(30);
                        const uint32_t lit_total = rdlane(lsum, 63);
                        op += ex.run(nb, llen, mlen, off, litp + lit_pos, op);
                        lit_pos += lit_total;
                        LZP(23);
                    }
                    wsync();
                }
                if (lane == 0) {   // the repeat offsets carry into the next block
                    wk->rep[0] = r0;
                    wk->rep[1] = r1;
                    wk->rep[2] = r2;
                }
                wsync();
            }
            const uint32_t rest = regen - lit_pos;
            if (out_len - op < rest) ZERR(31);
            ex.finish(op);
            for (uint32_t i = lane; i < rest; i += 64) dst[op + i] = litp[lit_pos + i];
            op += rest;
            ex.restart(op);
            ip += bsize;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_s_waitcnt(0);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        } else {
            ZERR(32);
        }
        if (last) break;
    }
    if (checksum) {
        if (n - ip < 4) ZERR(35);
        ip += 4;
    }
    if (fcs_bytes && fcs != op - frame_op0) ZERR(33);
    if (ip >= n) break;
    }   // frames
    LZP(19);
    LZP_END;
    if (op != out_len) ZERR(34);
#undef ZERR
    return op;
}

}  // namespace sb
