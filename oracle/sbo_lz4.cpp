// ORACLE — TEST INFRASTRUCTURE ONLY (see sbo.h).
//
// LZ4 *block* format, the payload of codec id 1 (reference call sites:
// src/compression/basic.rs:87-91 decompress_lz4 -> lz4::block::decompress_to_buffer,
// :108-120 compress_lz4 -> lz4::block::compress_to_buffer(src, None, false, dst)
// = LZ4_compress_default, no size prefix).  The algorithm itself lives in liblz4 (C,
// bundled by lz4-sys; `lz4 = "1.23.1"`, Cargo.toml:23), which is not under
// /root/reference; this file restates the published block format and liblz4 1.9.x's
// greedy single-probe hash parser from its documentation.  tests/test_oracle_blocks.py
// pins both directions against bytes liblz4 1.9.3 produced (tests/golden/blocks/, generator committed).
#include <cstring>

#include "sbo.h"

namespace sbo {

static const int MINMATCH = 4;
static const int MFLIMIT = 12;
static const int LASTLITERALS = 5;
static const unsigned ML_BITS = 4, ML_MASK = 15, RUN_MASK = 15;
static const uint32_t LZ4_DISTANCE_MAX = 65535;
static const size_t LZ4_64Klimit = 65536 + (MFLIMIT - 1);

size_t lz4_compress_bound(size_t n) { return n + n / 255 + 16; }

static inline uint32_t rd32(const uint8_t* p) {
    uint32_t v;
    memcpy(&v, p, 4);
    return v;
}
static inline uint64_t rd64(const uint8_t* p) {
    uint64_t v;
    memcpy(&v, p, 8);
    return v;
}

// LZ4_hash4 / LZ4_hash5 (64-bit little-endian build)
static inline uint32_t hash_pos(const uint8_t* p, bool by_u16) {
    if (by_u16) return (rd32(p) * 2654435761u) >> (32 - 13);
    return (uint32_t)(((rd64(p) << 24) * 889523592379ull) >> (64 - 12));
}

static inline unsigned count_match(const uint8_t* ip, const uint8_t* match, const uint8_t* limit) {
    const uint8_t* s = ip;
    while (ip < limit && *ip == *match) {
        ++ip;
        ++match;
    }
    return (unsigned)(ip - s);
}

size_t lz4_compress(const uint8_t* src, size_t n, uint8_t* dst, size_t cap) {
    if (cap < lz4_compress_bound(n)) throw Error(-2, "lz4_compress: dst too small");
    const bool by_u16 = n < LZ4_64Klimit;
    std::vector<uint32_t> table(by_u16 ? 8192 : 4096, 0u);
    const uint8_t* ip = src;
    const uint8_t* base = src;
    const uint8_t* anchor = src;
    const uint8_t* iend = src + n;
    const uint8_t* mflimitPlusOne = iend - MFLIMIT + 1;
    const uint8_t* matchlimit = iend - LASTLITERALS;
    uint8_t* op = dst;

    if (n >= (size_t)MFLIMIT + 1) {
        table[hash_pos(ip, by_u16)] = 0;
        ip++;
        uint32_t forwardH = hash_pos(ip, by_u16);
        for (;;) {
            const uint8_t* match;
            uint8_t* token;
            // find a match
            {
                const uint8_t* forwardIp = ip;
                int step = 1;
                int searchMatchNb = 1 << 6;
                bool done = false;
                for (;;) {
                    uint32_t h = forwardH;
                    uint32_t current = (uint32_t)(forwardIp - base);
                    uint32_t matchIndex = table[h];
                    ip = forwardIp;
                    forwardIp += step;
                    step = (searchMatchNb++ >> 6);
                    if (forwardIp > mflimitPlusOne) {
                        done = true;
                        break;
                    }
                    match = base + matchIndex;
                    forwardH = hash_pos(forwardIp, by_u16);
                    table[h] = current;
                    if (!by_u16 && matchIndex + LZ4_DISTANCE_MAX < current) continue;  // too far
                    if (rd32(match) == rd32(ip)) break;
                }
                if (done) break;  // -> last literals
            }
            // catch up
            while (ip > anchor && match > src && ip[-1] == match[-1]) {
                ip--;
                match--;
            }
            // encode literals
            {
                unsigned litLength = (unsigned)(ip - anchor);
                token = op++;
                if (litLength >= RUN_MASK) {
                    int len = (int)(litLength - RUN_MASK);
                    *token = (uint8_t)(RUN_MASK << ML_BITS);
                    for (; len >= 255; len -= 255) *op++ = 255;
                    *op++ = (uint8_t)len;
                } else {
                    *token = (uint8_t)(litLength << ML_BITS);
                }
                memcpy(op, anchor, litLength);
                op += litLength;
            }
            bool end_of_chunk = false;
            for (;;) {  // _next_match
                uint16_t off = (uint16_t)(ip - match);
                op[0] = (uint8_t)off;
                op[1] = (uint8_t)(off >> 8);
                op += 2;
                unsigned matchCode = count_match(ip + MINMATCH, match + MINMATCH, matchlimit);
                ip += (size_t)matchCode + MINMATCH;
                if (matchCode >= ML_MASK) {
                    *token += ML_MASK;
                    matchCode -= ML_MASK;
                    while (matchCode >= 255) {
                        *op++ = 255;
                        matchCode -= 255;
                    }
                    *op++ = (uint8_t)matchCode;
                } else {
                    *token += (uint8_t)matchCode;
                }
                anchor = ip;
                if (ip >= mflimitPlusOne) {
                    end_of_chunk = true;
                    break;
                }
                table[hash_pos(ip - 2, by_u16)] = (uint32_t)(ip - 2 - base);
                // test next position
                uint32_t h = hash_pos(ip, by_u16);
                uint32_t current = (uint32_t)(ip - base);
                uint32_t matchIndex = table[h];
                match = base + matchIndex;
                table[h] = current;
                if ((by_u16 || matchIndex + LZ4_DISTANCE_MAX >= current) && rd32(match) == rd32(ip)) {
                    token = op++;
                    *token = 0;
                    continue;
                }
                break;
            }
            if (end_of_chunk) break;
            forwardH = hash_pos(++ip, by_u16);
        }
    }
    // last literals
    {
        size_t lastRun = (size_t)(iend - anchor);
        if (lastRun >= RUN_MASK) {
            size_t acc = lastRun - RUN_MASK;
            *op++ = (uint8_t)(RUN_MASK << ML_BITS);
            for (; acc >= 255; acc -= 255) *op++ = 255;
            *op++ = (uint8_t)acc;
        } else {
            *op++ = (uint8_t)(lastRun << ML_BITS);
        }
        memcpy(op, anchor, lastRun);
        op += lastRun;
    }
    return (size_t)(op - dst);
}

// LZ4_decompress_safe semantics: sequences of token | [litlen ext] | literals |
// offset(le16) | [matchlen ext]; last sequence has literals only.
void lz4_decompress(const uint8_t* src, size_t n, uint8_t* dst, size_t out_len) {
    const uint8_t* ip = src;
    const uint8_t* iend = src + n;
    uint8_t* op = dst;
    uint8_t* oend = dst + out_len;
    if (n == 0) {
        if (out_len == 0) return;
        throw Error(-2, "lz4: empty input");
    }
    for (;;) {
        if (ip >= iend) throw Error(-2, "lz4: truncated (token)");
        unsigned token = *ip++;
        size_t lit = token >> 4;
        if (lit == 15) {
            unsigned s;
            do {
                if (ip >= iend) throw Error(-2, "lz4: truncated (litlen)");
                s = *ip++;
                lit += s;
            } while (s == 255);
        }
        if ((size_t)(iend - ip) < lit || (size_t)(oend - op) < lit) throw Error(-2, "lz4: literal overrun");
        memcpy(op, ip, lit);
        op += lit;
        ip += lit;
        if (ip == iend) break;  // last sequence: literals only
        if (iend - ip < 2) throw Error(-2, "lz4: truncated (offset)");
        size_t off = (size_t)ip[0] | ((size_t)ip[1] << 8);
        ip += 2;
        if (off == 0 || off > (size_t)(op - dst)) throw Error(-2, "lz4: bad offset");
        size_t ml = token & 15;
        if (ml == 15) {
            unsigned s;
            do {
                if (ip >= iend) throw Error(-2, "lz4: truncated (matchlen)");
                s = *ip++;
                ml += s;
            } while (s == 255);
        }
        ml += MINMATCH;
        if ((size_t)(oend - op) < ml) throw Error(-2, "lz4: match overrun");
        const uint8_t* m = op - off;
        for (size_t i = 0; i < ml; i++) op[i] = m[i];  // overlapping copy is byte-serial
        op += ml;
    }
    if (op != oend) throw Error(-2, "lz4: output size mismatch");
}

}  // namespace sbo
