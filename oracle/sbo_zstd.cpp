// ORACLE — TEST INFRASTRUCTURE ONLY (see sbo.h).
//
// Zstandard frame format (RFC 8878), codec id 2.  Reference call sites:
// src/compression/basic.rs:93-97 zstd::bulk::decompress_to_buffer,
// :122-135 zstd::bulk::compress_to_buffer(src, dst, 0)  (level 0 = library default 3).
// The algorithm lives in libzstd (C, via zstd-sys; `zstd = "0.11"`, Cargo.toml:24),
// which is not under /root/reference; this file restates the published format.
//
// Decoder: complete single-frame decoder — raw / RLE / compressed blocks, Huffman literals
// (1 and 4 streams, treeless), FSE sequences (predefined / RLE / compressed / repeat),
// repeat offsets, optional content checksum skipped.  No dictionaries (the reference
// never uses one).
// Encoder: emits a valid frame made of raw and RLE blocks.  Compressed bytes of libzstd
// level 3 are library-version dependent upstream (no Cargo.lock) and are not reproduced.
#include <cstring>

#include "sbo.h"

namespace sbo {

static const uint32_t ZSTD_MAGIC = 0xFD2FB528u;
static const size_t BLOCK_MAX = 128 * 1024;

size_t zstd_compress_bound(size_t n) { return n + 3 * (n / BLOCK_MAX + 1) + 18; }

size_t zstd_compress(const uint8_t* src, size_t n, uint8_t* dst, size_t cap) {
    if (cap < zstd_compress_bound(n)) throw Error(-2, "zstd_compress: dst too small");
    uint8_t* op = dst;
    memcpy(op, &ZSTD_MAGIC, 4);
    op += 4;
    // frame header: single segment, 8-byte frame content size, no checksum, no dict
    *op++ = (uint8_t)((3u << 6) | (1u << 5));
    uint64_t fcs = n;
    memcpy(op, &fcs, 8);
    op += 8;
    size_t pos = 0;
    do {
        size_t len = n - pos > BLOCK_MAX ? BLOCK_MAX : n - pos;
        bool last = pos + len == n;
        bool rle = len > 0;
        for (size_t i = 1; i < len && rle; i++) rle = src[pos + i] == src[pos];
        uint32_t hdr = (uint32_t)(last ? 1 : 0) | ((rle ? 1u : 0u) << 1) | ((uint32_t)len << 3);
        op[0] = (uint8_t)hdr;
        op[1] = (uint8_t)(hdr >> 8);
        op[2] = (uint8_t)(hdr >> 16);
        op += 3;
        if (rle) {
            *op++ = src[pos];
        } else {
            memcpy(op, src + pos, len);
            op += len;
        }
        pos += len;
    } while (pos < n);
    return (size_t)(op - dst);
}

// ---------------------------------------------------------------- decoder
namespace {

struct In {
    const uint8_t* p;
    const uint8_t* end;
    size_t left() const { return (size_t)(end - p); }
    const uint8_t* take(size_t n) {
        if (left() < n) throw Error(-2, "zstd: truncated input");
        const uint8_t* r = p;
        p += n;
        return r;
    }
    uint32_t le(int bytes) {
        const uint8_t* b = take((size_t)bytes);
        uint32_t v = 0;
        for (int i = 0; i < bytes; i++) v |= (uint32_t)b[i] << (8 * i);
        return v;
    }
};

// forward bit reader (FSE table descriptions)
struct FwdBits {
    const uint8_t* p;
    size_t n;
    size_t bitpos = 0;
    uint32_t peek(int nb) const {
        uint64_t v = 0;
        size_t byte = bitpos >> 3;
        for (int i = 0; i < 8 && byte + i < n; i++) v |= (uint64_t)p[byte + i] << (8 * i);
        v >>= (bitpos & 7);
        return (uint32_t)(v & ((1ull << nb) - 1));
    }
    void skip(int nb) { bitpos += nb; }
    size_t bytes_used() const { return (bitpos + 7) >> 3; }
};

// backward bit reader (Huffman / sequence bitstreams): bits are consumed from the end
struct BackBits {
    const uint8_t* p;
    int64_t bitpos;  // index of the next bit to read + 1 counted from the stream start
    void init(const uint8_t* s, size_t n) {
        if (n == 0) throw Error(-2, "zstd: empty bitstream");
        p = s;
        uint8_t last = s[n - 1];
        if (last == 0) throw Error(-2, "zstd: bitstream end mark missing");
        int hb = 31 - __builtin_clz((unsigned)last);
        bitpos = (int64_t)(n - 1) * 8 + hb;  // bits [0, bitpos) remain
    }
    // read nb bits (nb <= 32); reading past the start yields zeros (allowed by the format)
    uint32_t read(int nb) {
        if (nb == 0) return 0;
        uint32_t v = 0;
        for (int i = 0; i < nb; i++) {
            int64_t b = bitpos - 1 - i;
            uint32_t bit = b >= 0 ? (p[b >> 3] >> (b & 7)) & 1u : 0u;
            v = (v << 1) | bit;
        }
        bitpos -= nb;
        return v;
    }
    uint32_t peek(int nb) const {
        BackBits c = *this;
        return c.read(nb);
    }
};

struct FseEntry {
    uint8_t symbol, nbits;
    uint16_t base;
};
struct FseTable {
    int log = 0;
    std::vector<FseEntry> t;
};

static void fse_build(FseTable& tb, const int16_t* norm, int nsym, int log) {
    int size = 1 << log;
    tb.log = log;
    tb.t.assign((size_t)size, FseEntry{0, 0, 0});
    std::vector<uint16_t> next((size_t)nsym);
    int high = size - 1;
    for (int s = 0; s < nsym; s++) {
        if (norm[s] == -1) {
            tb.t[(size_t)high--].symbol = (uint8_t)s;
            next[(size_t)s] = 1;
        } else {
            next[(size_t)s] = (uint16_t)norm[s];
        }
    }
    int step = (size >> 1) + (size >> 3) + 3, mask = size - 1, pos = 0;
    for (int s = 0; s < nsym; s++) {
        for (int i = 0; i < norm[s]; i++) {
            tb.t[(size_t)pos].symbol = (uint8_t)s;
            do {
                pos = (pos + step) & mask;
            } while (pos > high);
        }
    }
    if (pos != 0) throw Error(-2, "zstd: corrupt FSE distribution");
    for (int i = 0; i < size; i++) {
        uint8_t s = tb.t[(size_t)i].symbol;
        uint16_t ns = next[s]++;
        int nb = log - (31 - __builtin_clz((unsigned)ns));
        tb.t[(size_t)i].nbits = (uint8_t)nb;
        tb.t[(size_t)i].base = (uint16_t)(((unsigned)ns << nb) - (unsigned)size);
    }
}

// returns bytes consumed
static size_t fse_read_header(const uint8_t* src, size_t n, int max_sym, int max_log, int16_t* norm, int& nsym,
                              int& log) {
    FwdBits b{src, n};
    log = (int)b.peek(4) + 5;
    b.skip(4);
    if (log > max_log) throw Error(-2, "zstd: FSE accuracy log too large");
    int remaining = (1 << log) + 1, threshold = 1 << log, nbits = log + 1;
    int sym = 0;
    bool prev0 = false;
    memset(norm, 0, sizeof(int16_t) * (size_t)(max_sym + 1));
    while (remaining > 1 && sym <= max_sym) {
        if (prev0) {
            // repeat flags: 2-bit values, 3 means "continue"
            for (;;) {
                uint32_t r = b.peek(2);
                b.skip(2);
                sym += (int)r;
                if (r != 3) break;
            }
            prev0 = false;
            if (sym > max_sym) throw Error(-2, "zstd: FSE too many symbols");
            continue;
        }
        int max = (2 * threshold - 1) - remaining;
        int count;
        uint32_t v = b.peek(nbits);
        if ((int)(v & (uint32_t)(threshold - 1)) < max) {
            count = (int)(v & (uint32_t)(threshold - 1));
            b.skip(nbits - 1);
        } else {
            count = (int)(v & (uint32_t)(2 * threshold - 1));
            if (count >= threshold) count -= max;
            b.skip(nbits);
        }
        count--;  // -1 = "less than 1" probability
        remaining -= count < 0 ? -count : count;
        norm[sym++] = (int16_t)count;
        prev0 = count == 0;
        while (remaining < threshold) {
            nbits--;
            threshold >>= 1;
        }
    }
    if (remaining != 1) throw Error(-2, "zstd: corrupt FSE header");
    nsym = sym;
    if (b.bytes_used() > n) throw Error(-2, "zstd: truncated FSE header");
    return b.bytes_used();
}

// ---- Huffman
struct HufTable {
    int max_bits = 0;
    std::vector<uint8_t> sym, len;  // indexed by max_bits-bit prefix
};

static void huf_build(HufTable& h, const uint8_t* weights, int nw) {
    // last weight is implied
    uint32_t total = 0;
    for (int i = 0; i < nw; i++) {
        if (weights[i] > 11) throw Error(-2, "zstd: huffman weight too large");
        total += weights[i] ? (1u << (weights[i] - 1)) : 0;
    }
    if (total == 0) throw Error(-2, "zstd: huffman weights all zero");
    int max_bits = 32 - __builtin_clz(total);  // floor(log2(total)) + 1
    uint32_t left = (1u << max_bits) - total;
    if (left == 0 || (left & (left - 1))) throw Error(-2, "zstd: huffman weights not a power of two");
    uint8_t w[256];
    memcpy(w, weights, (size_t)nw);
    w[nw] = (uint8_t)(32 - __builtin_clz(left));  // log2(left) + 1
    int n = nw + 1;
    if (max_bits > 11) throw Error(-2, "zstd: huffman table too deep");
    h.max_bits = max_bits;
    h.sym.assign((size_t)1 << max_bits, 0);
    h.len.assign((size_t)1 << max_bits, 0);
    // symbols sorted by weight ascending then symbol value; lowest weights get the lowest codes
    uint32_t code = 0;
    for (int wt = 1; wt <= max_bits; wt++) {
        for (int s = 0; s < n; s++) {
            if (w[s] != wt) continue;
            uint32_t span = 1u << (wt - 1);
            int nbits = max_bits + 1 - wt;
            for (uint32_t k = 0; k < span; k++) {
                h.sym[code + k] = (uint8_t)s;
                h.len[code + k] = (uint8_t)nbits;
            }
            code += span;
        }
    }
}

static size_t huf_read_table(HufTable& h, const uint8_t* src, size_t n) {
    if (n < 1) throw Error(-2, "zstd: truncated huffman header");
    uint8_t hb = src[0];
    uint8_t weights[256];
    int nw;
    size_t used;
    if (hb >= 128) {  // direct 4-bit weights
        nw = hb - 127;
        size_t bytes = (size_t)(nw + 1) / 2;
        if (n < 1 + bytes) throw Error(-2, "zstd: truncated huffman weights");
        for (int i = 0; i < nw; i++) {
            uint8_t b = src[1 + i / 2];
            weights[i] = (i & 1) ? (b & 15) : (b >> 4);
        }
        used = 1 + bytes;
    } else {  // FSE-compressed weights
        size_t clen = hb;
        if (n < 1 + clen) throw Error(-2, "zstd: truncated huffman weights");
        int16_t norm[13];
        int nsym, log;
        size_t hsz = fse_read_header(src + 1, clen, 12, 6, norm, nsym, log);
        FseTable tb;
        fse_build(tb, norm, nsym, log);
        BackBits bb;
        bb.init(src + 1 + hsz, clen - hsz);
        uint32_t s1 = bb.read(log), s2 = bb.read(log);
        nw = 0;
        for (;;) {
            if (nw >= 255) throw Error(-2, "zstd: too many huffman weights");
            weights[nw++] = tb.t[s1].symbol;
            if (bb.bitpos < tb.t[s1].nbits) {  // not enough bits to update: flush the other state
                weights[nw++] = tb.t[s2].symbol;
                break;
            }
            s1 = tb.t[s1].base + bb.read(tb.t[s1].nbits);
            weights[nw++] = tb.t[s2].symbol;
            if (bb.bitpos < tb.t[s2].nbits) {
                weights[nw++] = tb.t[s1].symbol;
                break;
            }
            s2 = tb.t[s2].base + bb.read(tb.t[s2].nbits);
        }
        used = 1 + clen;
    }
    huf_build(h, weights, nw);
    return used;
}

static void huf_decode_stream(const HufTable& h, const uint8_t* src, size_t n, uint8_t* dst, size_t out) {
    BackBits bb;
    bb.init(src, n);
    for (size_t i = 0; i < out; i++) {
        uint32_t idx = bb.peek(h.max_bits);
        dst[i] = h.sym[idx];
        bb.bitpos -= h.len[idx];
    }
    if (bb.bitpos != 0) throw Error(-2, "zstd: huffman stream not fully consumed");
}

// ---- sequences
static const int16_t LL_DEFAULT[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2,
                                       2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
static const int16_t ML_DEFAULT[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                       1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};
static const int16_t OF_DEFAULT[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1,
                                       1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};
static const uint32_t LL_BASE[36] = {0,  1,  2,  3,  4,  5,  6,  7,  8,  9,   10,  11,  12,  13,   14,   15,   16,   18,
                                     20, 22, 24, 28, 32, 40, 48, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536};
static const uint8_t LL_BITS[36] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1,
                                    1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
static const uint32_t ML_BASE[53] = {3,  4,  5,  6,  7,  8,  9,  10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20,
                                     21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 37, 39, 41,
                                     43, 47, 51, 59, 67, 83, 99, 131, 259, 515, 1027, 2051, 4099, 8195, 16387, 32771, 65539};
static const uint8_t ML_BITS[53] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                    0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};

struct FrameState {
    HufTable huf;
    bool have_huf = false;
    FseTable ll, of, ml;
    bool have_ll = false, have_of = false, have_ml = false;
    uint64_t rep[3] = {1, 4, 8};
};

static void seq_table(In& in, int mode, FseTable& tb, bool& have, const int16_t* def, int def_n, int def_log,
                      int max_sym, int max_log) {
    switch (mode) {
        case 0:
            fse_build(tb, def, def_n, def_log);
            have = true;
            break;
        case 1: {  // RLE: one symbol, zero bits
            uint8_t s = *in.take(1);
            if (s > max_sym) throw Error(-2, "zstd: RLE symbol out of range");
            tb.log = 0;
            tb.t.assign(1, FseEntry{s, 0, 0});
            have = true;
            break;
        }
        case 2: {
            int16_t norm[64];
            int nsym, log;
            size_t used = fse_read_header(in.p, in.left(), max_sym, max_log, norm, nsym, log);
            in.take(used);
            fse_build(tb, norm, nsym, log);
            have = true;
            break;
        }
        default:
            if (!have) throw Error(-2, "zstd: repeat mode without a previous table");
    }
}

static void decode_block(FrameState& fs, const uint8_t* src, size_t n, uint8_t* dst_base, uint8_t*& op,
                         uint8_t* oend) {
    In in{src, src + n};
    // ---- literals section
    uint8_t b0 = *in.take(1);
    int ltype = b0 & 3, sf = (b0 >> 2) & 3;
    size_t regen, csize = 0;
    int streams = 1;
    std::vector<uint8_t> lits;
    if (ltype == 0 || ltype == 1) {
        if (sf == 0 || sf == 2)
            regen = b0 >> 3;
        else if (sf == 1)
            regen = (b0 >> 4) | ((size_t)*in.take(1) << 4);
        else {
            const uint8_t* e = in.take(2);
            regen = (b0 >> 4) | ((size_t)e[0] << 4) | ((size_t)e[1] << 12);
        }
        lits.resize(regen);
        if (ltype == 0) {
            if (regen) memcpy(lits.data(), in.take(regen), regen);
        } else {
            uint8_t v = *in.take(1);
            if (regen) memset(lits.data(), v, regen);
        }
    } else {
        if (sf == 0 || sf == 1) {
            const uint8_t* e = in.take(2);
            uint32_t v = (b0 >> 4) | ((uint32_t)e[0] << 4) | ((uint32_t)e[1] << 12);
            regen = v & 0x3FF;
            csize = v >> 10;
            streams = sf == 0 ? 1 : 4;
        } else if (sf == 2) {
            const uint8_t* e = in.take(3);
            uint32_t v = (b0 >> 4) | ((uint32_t)e[0] << 4) | ((uint32_t)e[1] << 12) | ((uint32_t)e[2] << 20);
            regen = v & 0x3FFF;
            csize = v >> 14;
            streams = 4;
        } else {
            const uint8_t* e = in.take(4);
            uint64_t v = (b0 >> 4) | ((uint64_t)e[0] << 4) | ((uint64_t)e[1] << 12) | ((uint64_t)e[2] << 20) |
                         ((uint64_t)e[3] << 28);
            regen = v & 0x3FFFF;
            csize = (size_t)(v >> 18);
            streams = 4;
        }
        const uint8_t* lsrc = in.take(csize);
        size_t lleft = csize;
        if (ltype == 2) {
            size_t used = huf_read_table(fs.huf, lsrc, lleft);
            fs.have_huf = true;
            lsrc += used;
            lleft -= used;
        } else if (!fs.have_huf) {
            throw Error(-2, "zstd: treeless literals without a previous table");
        }
        lits.resize(regen);
        if (streams == 1) {
            huf_decode_stream(fs.huf, lsrc, lleft, lits.data(), regen);
        } else {
            if (lleft < 6) throw Error(-2, "zstd: truncated jump table");
            size_t s1 = lsrc[0] | ((size_t)lsrc[1] << 8), s2 = lsrc[2] | ((size_t)lsrc[3] << 8),
                   s3 = lsrc[4] | ((size_t)lsrc[5] << 8);
            if (6 + s1 + s2 + s3 > lleft) throw Error(-2, "zstd: bad jump table");
            size_t s4 = lleft - 6 - s1 - s2 - s3;
            size_t per = (regen + 3) / 4;
            if (per * 3 > regen) throw Error(-2, "zstd: bad 4-stream literal size");
            const uint8_t* q = lsrc + 6;
            huf_decode_stream(fs.huf, q, s1, lits.data(), per);
            huf_decode_stream(fs.huf, q + s1, s2, lits.data() + per, per);
            huf_decode_stream(fs.huf, q + s1 + s2, s3, lits.data() + 2 * per, per);
            huf_decode_stream(fs.huf, q + s1 + s2 + s3, s4, lits.data() + 3 * per, regen - 3 * per);
        }
    }
    // ---- sequences section
    size_t nseq;
    {
        uint8_t s0 = *in.take(1);
        if (s0 < 128)
            nseq = s0;
        else if (s0 < 255)
            nseq = ((size_t)(s0 - 128) << 8) + *in.take(1);
        else {
            const uint8_t* e = in.take(2);
            nseq = (size_t)e[0] + ((size_t)e[1] << 8) + 0x7F00;
        }
    }
    size_t lit_pos = 0;
    if (nseq > 0) {
        uint8_t modes = *in.take(1);
        if (modes & 3) throw Error(-2, "zstd: reserved bits set in sequence modes");
        seq_table(in, (modes >> 6) & 3, fs.ll, fs.have_ll, LL_DEFAULT, 36, 6, 35, 9);
        seq_table(in, (modes >> 4) & 3, fs.of, fs.have_of, OF_DEFAULT, 29, 5, 31, 8);
        seq_table(in, (modes >> 2) & 3, fs.ml, fs.have_ml, ML_DEFAULT, 53, 6, 52, 9);
        BackBits bb;
        bb.init(in.p, in.left());
        uint32_t sl = bb.read(fs.ll.log), so = bb.read(fs.of.log), sm = bb.read(fs.ml.log);
        for (size_t i = 0; i < nseq; i++) {
            uint8_t ofc = fs.of.t[so].symbol, mlc = fs.ml.t[sm].symbol, llc = fs.ll.t[sl].symbol;
            if (ofc > 31 || mlc > 52 || llc > 35) throw Error(-2, "zstd: sequence code out of range");
            uint64_t ofv = ((uint64_t)1 << ofc) + bb.read(ofc);
            uint64_t mlen = ML_BASE[mlc] + bb.read(ML_BITS[mlc]);
            uint64_t llen = LL_BASE[llc] + bb.read(LL_BITS[llc]);
            uint64_t offset;
            if (ofv > 3) {
                offset = ofv - 3;
                fs.rep[2] = fs.rep[1];
                fs.rep[1] = fs.rep[0];
                fs.rep[0] = offset;
            } else {
                uint64_t idx = ofv - 1;  // 0..2
                if (llen == 0) idx++;
                if (idx == 0) {
                    offset = fs.rep[0];
                } else {
                    offset = idx < 3 ? fs.rep[idx] : fs.rep[0] - 1;
                    if (offset == 0) throw Error(-2, "zstd: zero repeat offset");
                    if (idx > 1) fs.rep[2] = fs.rep[1];
                    fs.rep[1] = fs.rep[0];
                    fs.rep[0] = offset;
                }
            }
            if (i + 1 < nseq) {  // update order: LL, ML, OF
                sl = fs.ll.t[sl].base + bb.read(fs.ll.t[sl].nbits);
                sm = fs.ml.t[sm].base + bb.read(fs.ml.t[sm].nbits);
                so = fs.of.t[so].base + bb.read(fs.of.t[so].nbits);
            }
            if (lit_pos + llen > lits.size()) throw Error(-2, "zstd: literal overrun");
            if ((uint64_t)(oend - op) < llen + mlen) throw Error(-2, "zstd: output overrun");
            memcpy(op, lits.data() + lit_pos, (size_t)llen);
            op += llen;
            lit_pos += (size_t)llen;
            if (offset > (uint64_t)(op - dst_base)) throw Error(-2, "zstd: offset beyond window start");
            const uint8_t* m = op - offset;
            for (uint64_t k = 0; k < mlen; k++) op[k] = m[k];
            op += mlen;
        }
        if (bb.bitpos != 0) throw Error(-2, "zstd: sequence bitstream not fully consumed");
    }
    size_t rest = lits.size() - lit_pos;
    if ((size_t)(oend - op) < rest) throw Error(-2, "zstd: output overrun (trailing literals)");
    memcpy(op, lits.data() + lit_pos, rest);
    op += rest;
}

}  // namespace

void zstd_decompress(const uint8_t* src, size_t n, uint8_t* dst, size_t out_len) {
    In in{src, src + n};
    uint8_t* op = dst;
    uint8_t* oend = dst + out_len;
    // zstd::bulk::decompress_to_buffer ends in ZSTD_decompress: any number of frames (and skippable frames) back to back
    while (in.left() >= 4) {
        uint32_t magic = in.le(4);
        if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) {  // skippable frame
            uint32_t sz = in.le(4);
            in.take(sz);
            continue;
        }
        if (magic != ZSTD_MAGIC) throw Error(-2, "zstd: bad magic");
        uint8_t fhd = *in.take(1);
        int fcs_flag = fhd >> 6, single = (fhd >> 5) & 1, checksum = (fhd >> 2) & 1, did = fhd & 3;
        if (fhd & 0x08) throw Error(-2, "zstd: reserved bit set");
        if (!single) in.take(1);  // window descriptor
        static const int DID_BYTES[4] = {0, 1, 2, 4};
        if (did) {
            uint32_t id = in.le(DID_BYTES[did]);
            if (id != 0) throw Error(-2, "zstd: dictionaries are not supported");
        }
        int fcs_bytes = fcs_flag == 0 ? (single ? 1 : 0) : (1 << fcs_flag);
        uint64_t fcs = 0;
        if (fcs_bytes) {
            const uint8_t* b = in.take((size_t)fcs_bytes);
            for (int i = 0; i < fcs_bytes; i++) fcs |= (uint64_t)b[i] << (8 * i);
            if (fcs_bytes == 2) fcs += 256;
        }
        FrameState fs;
        uint8_t* frame_start = op;
        for (;;) {
            uint32_t bh = in.le(3);
            bool last = bh & 1;
            int btype = (bh >> 1) & 3;
            size_t bsize = bh >> 3;
            if (btype == 0) {
                if ((size_t)(oend - op) < bsize) throw Error(-2, "zstd: output overrun (raw block)");
                const uint8_t* b = in.take(bsize);
                if (bsize) memcpy(op, b, bsize);
                op += bsize;
            } else if (btype == 1) {
                if ((size_t)(oend - op) < bsize) throw Error(-2, "zstd: output overrun (rle block)");
                uint8_t v = *in.take(1);
                if (bsize) memset(op, v, bsize);
                op += bsize;
            } else if (btype == 2) {
                if (bsize > BLOCK_MAX) throw Error(-2, "zstd: block too large");
                decode_block(fs, in.take(bsize), bsize, dst, op, oend);
            } else {
                throw Error(-2, "zstd: reserved block type");
            }
            if (last) break;
        }
        if (checksum) in.take(4);
        if (fcs_bytes && (uint64_t)(op - frame_start) != fcs) throw Error(-2, "zstd: frame content size mismatch");
    }
    if (op != oend) throw Error(-2, "zstd: output size mismatch");
}

}  // namespace sbo
