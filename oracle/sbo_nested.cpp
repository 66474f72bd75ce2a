// ORACLE — TEST INFRASTRUCTURE ONLY (see sbo.h).
//
// Nested (Dremel) level section of a page: `u32 page_rows | u32 rep_len | u32 def_len | rep | def`
// (reference: write_nested_validity, src/write/serialize.rs:217-232; read_validity_nested,
// src/read/read_basic.rs:65-173).  The level arithmetic itself lives in arrow2 0.17
// (`io::parquet::write::{write_rep_and_def, num_values, slice_parquet_array}`,
// `io::parquet::read::{init_nested, NestedState}`) and parquet2 0.17 (hybrid-RLE), which are not
// under /root/reference: this file restates their published behaviour (Dremel levels: the
// definition level grows by one for every optional node and every repeated node on the path, the
// repetition level is the depth of the innermost list that continues).  Parity with arrow2's exact
// bytes is therefore UNPINNED; what is pinned is the reference's own decoder state machine
// (read_basic.rs:93-164, restated line by line below) and encode->decode round trips.
#include <algorithm>
#include <cstdio>
#include <cstring>

#include "sbo_util.h"

namespace sbo {

struct NestedLevel {
    int32_t kind;         // 0 primitive, 1 list (i32 offsets), 2 large list (i64 offsets), 3 struct
    int32_t is_optional;
    const uint8_t* validity;  // may be null
    uint64_t validity_off;
    const uint8_t* offsets;   // lists: length + 1 entries
    uint64_t length;
};

static inline uint64_t list_off(const NestedLevel& l, uint64_t i) {
    if (l.kind == 1) {
        int32_t v;
        memcpy(&v, l.offsets + i * 4, 4);
        return (uint64_t)v;
    }
    int64_t v;
    memcpy(&v, l.offsets + i * 8, 8);
    return (uint64_t)v;
}
static inline uint32_t level_valid(const NestedLevel& l, uint64_t i) {
    if (!l.is_optional) return 0;
    if (!l.validity) return 1;
    uint64_t k = l.validity_off + i;
    return (l.validity[k >> 3] >> (k & 7)) & 1;
}

// get_bit_width (parquet2 read::levels): 16 - leading_zeros(max as u16)
static inline uint32_t level_bit_width(uint32_t max_level) {
    uint32_t b = 0;
    while (max_level >> b) b++;
    return b;
}

static void gen_entries(const NestedLevel* lv, int D, int k, uint64_t i, uint32_t rep_in, uint32_t def_base,
                        uint32_t rep_depth, std::vector<uint32_t>& rep, std::vector<uint32_t>& def) {
    const NestedLevel& l = lv[k];
    const uint32_t v = level_valid(l, i);
    if (l.kind == 0) {
        rep.push_back(rep_in);
        def.push_back(def_base + v);
    } else if (l.kind == 3) {
        gen_entries(lv, D, k + 1, i, rep_in, def_base + v, rep_depth, rep, def);
    } else {
        const uint64_t b = list_off(l, i), e = list_off(l, i + 1);
        if (e == b) {  // empty (or null) list: one entry that stops here
            rep.push_back(rep_in);
            def.push_back(def_base + v);
        } else {
            for (uint64_t c = b; c < e; c++)
                gen_entries(lv, D, k + 1, c, c == b ? rep_in : rep_depth + 1, def_base + v + 1, rep_depth + 1, rep, def);
        }
    }
}

// parquet2 0.17 encode_u32: one bit-packed run; values packed LSB first in chunks of 32; the last
// partial chunk writes ceil(rem * bits / 8) bytes out of a buffer that still holds the previous
// chunk's values in the unused slots
static void encode_levels(const std::vector<uint32_t>& v, uint32_t bits, std::vector<uint8_t>& out) {
    uint64_t header = (((uint64_t)v.size() + 7) / 8) << 1 | 1;
    while (header >= 0x80) {
        out.push_back((uint8_t)(header | 0x80));
        header >>= 7;
    }
    out.push_back((uint8_t)header);
    uint32_t buffer[32];
    memset(buffer, 0, sizeof buffer);
    const size_t chunks = v.size() / 32, rem = v.size() % 32;
    auto pack = [&](size_t nbytes) {
        uint8_t packed[128];
        memset(packed, 0, sizeof packed);
        for (int j = 0; j < 32; j++) {
            const uint64_t bitpos = (uint64_t)j * bits;
            uint64_t val = (uint64_t)(buffer[j] & (bits >= 32 ? 0xFFFFFFFFu : ((1u << bits) - 1))) << (bitpos & 7);
            for (int b = 0; b < 5 && (bitpos >> 3) + b < 128; b++) packed[(bitpos >> 3) + b] |= (uint8_t)(val >> (8 * b));
        }
        out.insert(out.end(), packed, packed + nbytes);
    };
    for (size_t c = 0; c < chunks; c++) {
        memcpy(buffer, v.data() + 32 * c, 128);
        pack(4 * bits);
    }
    if (rem) {
        memcpy(buffer, v.data() + 32 * chunks, rem * 4);
        pack((rem * bits + 7) / 8);
    }
}

// parquet2 HybridRleDecoder: runs of bit-packed groups or RLE values
static void decode_levels(const uint8_t* p, size_t n, uint32_t bits, size_t count, std::vector<uint32_t>& out) {
    out.clear();
    if (bits == 0) {
        out.assign(count, 0);
        return;
    }
    Reader r(p, n);
    while (out.size() < count) {
        if (r.left() == 0) io_eof("levels");
        uint64_t ind = 0;
        unsigned sh = 0;
        for (;;) {
            uint8_t b = r.u8("levels uleb");
            ind |= (uint64_t)(b & 0x7F) << sh;
            sh += 7;
            if (!(b & 0x80)) break;
        }
        if (ind & 1) {
            size_t bytes = std::min((size_t)(ind >> 1) * bits, r.left());
            const uint8_t* run = r.take(bytes);
            const size_t avail = bytes * 8 / bits;
            for (size_t j = 0; j < avail && out.size() < count; j++) {
                const uint64_t bitpos = (uint64_t)j * bits;
                uint64_t w = 0;
                for (int b = 0; b < 5 && (bitpos >> 3) + b < bytes; b++) w |= (uint64_t)run[(bitpos >> 3) + b] << (8 * b);
                out.push_back((uint32_t)((w >> (bitpos & 7)) & ((1ull << bits) - 1)));
            }
        } else {
            const size_t len = (size_t)(ind >> 1), vb = (bits + 7) / 8;
            const uint8_t* pv = r.take(vb, "levels rle value");
            uint32_t val = 0;
            for (size_t b = 0; b < vb; b++) val |= (uint32_t)pv[b] << (8 * b);
            for (size_t j = 0; j < len && out.size() < count; j++) out.push_back(val);
        }
    }
}

struct NestedWritten {
    std::vector<uint8_t> bytes;  // the level section
    uint64_t num_values, leaf_start, leaf_count;
};

// write_nested_validity for top-level rows [r0, r0 + len)  (serialize.rs:217-232)
void nested_write_levels(const NestedLevel* lv, int D, uint64_t r0, uint64_t len, NestedWritten& w) {
    uint32_t max_rep = 0, max_def = 0;
    for (int k = 0; k < D; k++) {
        if (lv[k].kind == 1 || lv[k].kind == 2) {
            max_rep++;
            max_def++;
        }
        if (lv[k].is_optional) max_def++;
    }
    std::vector<uint32_t> rep, def;
    for (uint64_t i = r0; i < r0 + len; i++) gen_entries(lv, D, 0, i, 0, 0, 0, rep, def);
    uint64_t s = r0, e = r0 + len;  // slice_parquet_array: the leaf range of the row range
    for (int k = 0; k < D; k++)
        if (lv[k].kind == 1 || lv[k].kind == 2) {
            s = list_off(lv[k], s);
            e = list_off(lv[k], e);
        }
    w.leaf_start = s;
    w.leaf_count = e - s;
    w.num_values = rep.size();
    std::vector<uint8_t> rb, db;
    if (max_rep > 0) encode_levels(rep, level_bit_width(max_rep), rb);
    if (max_def > 0) encode_levels(def, level_bit_width(max_def), db);
    w.bytes.clear();
    put_u32(w.bytes, (uint32_t)len);
    put_u32(w.bytes, (uint32_t)rb.size());
    put_u32(w.bytes, (uint32_t)db.size());
    put_bytes(w.bytes, rb.data(), rb.size());
    put_bytes(w.bytes, db.data(), db.size());
}

struct NestedRead {
    // per level: list -> start offsets (one per element) ; optional levels -> validity bits
    std::vector<std::vector<int64_t>> offsets;
    std::vector<std::vector<uint8_t>> validity;  // one byte per element (0/1), empty if not nullable
    std::vector<uint64_t> lengths;
    std::vector<uint8_t> leaf_validity;          // one byte per leaf slot, empty if the leaf is not nullable
    uint64_t consumed = 0;
};

// read_validity_nested (read_basic.rs:65-173).  kinds/nullable describe init: Vec<InitNested>.
void nested_read_levels(const uint8_t* page, size_t n, uint64_t num_values, const int32_t* kinds,
                        const int32_t* nullable, int D, NestedRead& out) {
    Reader r(page, n);
    const uint32_t additional = r.u32("page rows");
    const uint32_t rep_len = r.u32("rep len"), def_len = r.u32("def len");
    uint32_t max_rep = 0, max_def = 0;
    for (int k = 0; k < D; k++) {
        if (kinds[k] == 1 || kinds[k] == 2) {
            max_rep++;
            max_def++;
        }
        if (nullable[k]) max_def++;
    }
    const uint8_t* rp = r.take(rep_len, "rep levels");
    const uint8_t* dp = r.take(def_len, "def levels");
    std::vector<uint32_t> reps, defs;
    decode_levels(rp, rep_len, level_bit_width(max_rep), (size_t)num_values, reps);
    decode_levels(dp, def_len, level_bit_width(max_def), (size_t)num_values, defs);
    out.offsets.assign(D, {});
    out.validity.assign(D, {});
    out.lengths.assign(D, 0);
    std::vector<uint32_t> cum_sum(D + 1, 0), cum_rep(D + 1, 0);
    for (int k = 0; k < D; k++) {
        const bool repeated = kinds[k] == 1 || kinds[k] == 2;
        cum_sum[k + 1] = cum_sum[k] + (nullable[k] ? 1 : 0) + (repeated ? 1 : 0);
        cum_rep[k + 1] = cum_rep[k] + (repeated ? 1 : 0);
    }
    uint32_t rows = 0;
    for (size_t e = 0; e < reps.size(); e++) {
        const uint32_t rep = reps[e], def = defs[e];
        if (rep == 0) rows++;
        bool is_required = false;
        for (int depth = 0; depth < D; depth++) {
            const bool right_level = rep <= cum_rep[depth] && def >= cum_sum[depth];
            if (is_required || right_level) {
                const int64_t length = depth + 1 < D ? (int64_t)out.lengths[depth + 1] : 1;
                const bool is_valid = nullable[depth] && def > cum_sum[depth];
                // nest.push(length, is_valid)
                if (kinds[depth] == 1 || kinds[depth] == 2) out.offsets[depth].push_back(length);
                if (nullable[depth] && kinds[depth] != 0) out.validity[depth].push_back(is_valid ? 1 : 0);
                out.lengths[depth]++;
                const bool nest_is_required = kinds[depth] == 3 && !nullable[depth];
                is_required = nest_is_required && !is_valid;
                if (depth == D - 1 && nullable[depth]) {
                    const bool lv = def != cum_sum[depth];
                    out.leaf_validity.push_back(right_level && lv ? 1 : 0);
                }
            }
        }
        const uint32_t next_rep = e + 1 < reps.size() ? reps[e + 1] : 0;
        if (next_rep == 0 && rows == additional) break;
    }
    out.consumed = 12 + (uint64_t)rep_len + def_len;
}

}  // namespace sbo

// ---------------------------------------------------------------- C ABI for ctypes
using namespace sbo;
extern "C" {

struct sbo_nested_level {
    int32_t kind, is_optional;
    const uint8_t* validity;
    uint64_t validity_off;
    const uint8_t* offsets;
    uint64_t length;
};

void* sbo_nested_write(const sbo_nested_level* lv, int32_t D, uint64_t r0, uint64_t len, char* err, size_t cap) {
    NestedWritten* w = new NestedWritten();
    try {
        std::vector<NestedLevel> v(D);
        for (int k = 0; k < D; k++) v[k] = NestedLevel{lv[k].kind, lv[k].is_optional, lv[k].validity, lv[k].validity_off, lv[k].offsets, lv[k].length};
        nested_write_levels(v.data(), D, r0, len, *w);
        return w;
    } catch (const std::exception& e) {
        if (err && cap) snprintf(err, cap, "%s", e.what());
        delete w;
        return nullptr;
    }
}
uint64_t sbo_nested_written_len(void* h) { return ((NestedWritten*)h)->bytes.size(); }
const uint8_t* sbo_nested_written_data(void* h) { return ((NestedWritten*)h)->bytes.data(); }
void sbo_nested_written_info(void* h, uint64_t* out3) {
    NestedWritten* w = (NestedWritten*)h;
    out3[0] = w->num_values;
    out3[1] = w->leaf_start;
    out3[2] = w->leaf_count;
}
void sbo_nested_written_free(void* h) { delete (NestedWritten*)h; }

void* sbo_nested_read(const uint8_t* page, uint64_t n, uint64_t num_values, const int32_t* kinds, const int32_t* nullable,
                      int32_t D, char* err, size_t cap) {
    NestedRead* r = new NestedRead();
    try {
        nested_read_levels(page, (size_t)n, num_values, kinds, nullable, D, *r);
        return r;
    } catch (const std::exception& e) {
        if (err && cap) snprintf(err, cap, "%s", e.what());
        delete r;
        return nullptr;
    }
}
uint64_t sbo_nested_read_consumed(void* h) { return ((NestedRead*)h)->consumed; }
uint64_t sbo_nested_read_length(void* h, int32_t k) { return ((NestedRead*)h)->lengths[k]; }
uint64_t sbo_nested_read_noffsets(void* h, int32_t k) { return ((NestedRead*)h)->offsets[k].size(); }
const int64_t* sbo_nested_read_offsets(void* h, int32_t k) { return ((NestedRead*)h)->offsets[k].data(); }
uint64_t sbo_nested_read_nvalidity(void* h, int32_t k) { return ((NestedRead*)h)->validity[k].size(); }
const uint8_t* sbo_nested_read_validity(void* h, int32_t k) { return ((NestedRead*)h)->validity[k].data(); }
uint64_t sbo_nested_read_nleaf_validity(void* h) { return ((NestedRead*)h)->leaf_validity.size(); }
const uint8_t* sbo_nested_read_leaf_validity(void* h) { return ((NestedRead*)h)->leaf_validity.data(); }
void sbo_nested_read_free(void* h) { delete (NestedRead*)h; }

}  // extern "C"
