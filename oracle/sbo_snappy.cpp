// ORACLE — TEST INFRASTRUCTURE ONLY (see sbo.h).
//
// Snappy *raw* format, codec id 3 (reference call sites: src/compression/basic.rs:99-106
// snap::raw::Decoder::decompress, :137-152 snap::raw::Encoder::compress; `snap = "1.1.0"`,
// Cargo.toml:25 — not under /root/reference).  Restates the published format
// (varint length preamble, then literal / copy-1 / copy-2 / copy-4 elements).  The encoder
// is a plain greedy 4-byte-hash matcher: its bytes are valid Snappy but not claimed
// identical to the snap crate's.  Cross-checked against pyarrow's snappy codec in tests.
#include <cstring>

#include "sbo.h"

namespace sbo {

size_t snappy_compress_bound(size_t n) { return 32 + n + n / 6; }

static inline uint32_t rd32(const uint8_t* p) {
    uint32_t v;
    memcpy(&v, p, 4);
    return v;
}

static uint8_t* emit_literal(uint8_t* op, const uint8_t* lit, size_t len) {
    if (len == 0) return op;
    size_t n = len - 1;
    if (n < 60) {
        *op++ = (uint8_t)(n << 2);
    } else {
        int bytes = n < (1u << 8) ? 1 : n < (1u << 16) ? 2 : n < (1u << 24) ? 3 : 4;
        *op++ = (uint8_t)((59 + bytes) << 2);
        for (int i = 0; i < bytes; i++) *op++ = (uint8_t)(n >> (8 * i));
    }
    memcpy(op, lit, len);
    return op + len;
}
static uint8_t* emit_copy(uint8_t* op, size_t offset, size_t len) {
    while (len > 0) {
        size_t l = len > 64 ? 64 : len;
        if (len > 64 && len - 64 < 4) l = 60;  // keep the tail >= 4
        if (l >= 4 && l <= 11 && offset < 2048) {
            *op++ = (uint8_t)(1 | ((l - 4) << 2) | ((offset >> 8) << 5));
            *op++ = (uint8_t)offset;
        } else if (offset < 65536) {
            *op++ = (uint8_t)(2 | ((l - 1) << 2));
            *op++ = (uint8_t)offset;
            *op++ = (uint8_t)(offset >> 8);
        } else {
            *op++ = (uint8_t)(3 | ((l - 1) << 2));
            for (int i = 0; i < 4; i++) *op++ = (uint8_t)(offset >> (8 * i));
        }
        len -= l;
    }
    return op;
}

size_t snappy_compress(const uint8_t* src, size_t n, uint8_t* dst, size_t cap) {
    if (cap < snappy_compress_bound(n)) throw Error(-2, "snappy_compress: dst too small");
    uint8_t* op = dst;
    size_t v = n;
    while (v >= 0x80) {
        *op++ = (uint8_t)(v | 0x80);
        v >>= 7;
    }
    *op++ = (uint8_t)v;
    std::vector<uint32_t> table(1 << 14, 0xFFFFFFFFu);
    size_t ip = 0, anchor = 0;
    while (n >= 4 && ip + 4 <= n) {
        uint32_t h = (rd32(src + ip) * 0x1e35a7bdu) >> (32 - 14);
        uint32_t cand = table[h];
        table[h] = (uint32_t)ip;
        if (cand != 0xFFFFFFFFu && rd32(src + cand) == rd32(src + ip)) {
            size_t ml = 4;
            while (ip + ml < n && src[cand + ml] == src[ip + ml]) ml++;
            op = emit_literal(op, src + anchor, ip - anchor);
            op = emit_copy(op, ip - cand, ml);
            ip += ml;
            anchor = ip;
        } else {
            ip++;
        }
    }
    op = emit_literal(op, src + anchor, n - anchor);
    return (size_t)(op - dst);
}

void snappy_decompress(const uint8_t* src, size_t n, uint8_t* dst, size_t out_len) {
    size_t ip = 0;
    uint64_t ulen = 0;
    unsigned shift = 0;
    for (;;) {
        if (ip >= n) throw Error(-2, "snappy: truncated preamble");
        uint8_t b = src[ip++];
        ulen |= (uint64_t)(b & 0x7F) << shift;
        shift += 7;
        if (!(b & 0x80)) break;
    }
    if (ulen != out_len) throw Error(-2, "snappy: length mismatch");
    size_t op = 0;
    while (ip < n) {
        uint8_t tag = src[ip++];
        size_t len, off;
        switch (tag & 3) {
            case 0: {
                len = (tag >> 2) + 1;
                if (len > 60) {
                    int bytes = (int)len - 60;
                    if (ip + bytes > n) throw Error(-2, "snappy: truncated literal length");
                    len = 0;
                    for (int i = 0; i < bytes; i++) len |= (size_t)src[ip + i] << (8 * i);
                    len += 1;
                    ip += bytes;
                }
                if (ip + len > n || op + len > out_len) throw Error(-2, "snappy: literal overrun");
                memcpy(dst + op, src + ip, len);
                ip += len;
                op += len;
                continue;
            }
            case 1:
                if (ip + 1 > n) throw Error(-2, "snappy: truncated copy1");
                len = ((tag >> 2) & 7) + 4;
                off = ((size_t)(tag >> 5) << 8) | src[ip];
                ip += 1;
                break;
            case 2:
                if (ip + 2 > n) throw Error(-2, "snappy: truncated copy2");
                len = (tag >> 2) + 1;
                off = (size_t)src[ip] | ((size_t)src[ip + 1] << 8);
                ip += 2;
                break;
            default:
                if (ip + 4 > n) throw Error(-2, "snappy: truncated copy4");
                len = (tag >> 2) + 1;
                off = rd32(src + ip);
                ip += 4;
                break;
        }
        if (off == 0 || off > op || op + len > out_len) throw Error(-2, "snappy: bad copy");
        for (size_t i = 0; i < len; i++) dst[op + i] = dst[op - off + i];
        op += len;
    }
    if (op != out_len) throw Error(-2, "snappy: output size mismatch");
}

}  // namespace sbo
