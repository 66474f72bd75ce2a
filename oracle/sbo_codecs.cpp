// ORACLE — TEST INFRASTRUCTURE ONLY (see sbo.h).
//
// Scalar CPU restatement of strawboat's page codecs, selectors and page drivers.
// Reference files followed (all under /root/reference/src):
//   compression/mod.rs, compression/basic.rs,
//   compression/integer/{mod,rle,dict,bp,delta_bp,one_value,freq}.rs,
//   compression/double/{mod,rle,dict,one_value,freq,patas}.rs,
//   compression/binary/{mod,dict,one_value,freq}.rs,
//   compression/boolean/{mod,rle,one_value}.rs,
//   write/{serialize,primitive,binary,boolean,common}.rs,
//   read/read_basic.rs, read/array/{integer,double,boolean,binary}.rs.
#include <algorithm>
#include <string>
#include <string_view>

#include "sbo_util.h"

#include <dlfcn.h>

namespace sbo {

// ============================================================ system block codecs (CPU baseline leg only)
// the CPU leg links the box's liblz4 / libzstd when present — the libraries the reference's `lz4`
// and `zstd` crates wrap (src/compression/basic.rs:87-135: LZ4_compress_default / LZ4_decompress_safe,
// ZSTD_compress at level 0 = the default level 3 / ZSTD_decompress).  Off by default: the tests run the restatement's own
// codecs (pinned against these very libraries in tests/test_oracle_blocks.py); bench.py's cpu_baseline switches them on.
namespace syscodec {
typedef int (*lz4_compress_fn)(const char*, char*, int, int);
typedef int (*lz4_decompress_fn)(const char*, char*, int, int);
typedef int (*lz4_version_fn)();
typedef size_t (*zstd_compress_fn)(void*, size_t, const void*, size_t, int);
typedef size_t (*zstd_decompress_fn)(void*, size_t, const void*, size_t);
typedef unsigned (*zstd_iserror_fn)(size_t);
typedef unsigned (*zstd_version_fn)();
static lz4_compress_fn lz4_c = nullptr;
static lz4_decompress_fn lz4_d = nullptr;
static zstd_compress_fn zstd_c = nullptr;
static zstd_decompress_fn zstd_d = nullptr;
static zstd_iserror_fn zstd_err = nullptr;
static int lz4_ver = 0, zstd_ver = 0;
static bool use_lz4 = false, use_zstd = false;
// returns bit 0: liblz4 in use, bit 1: libzstd in use
static int enable(bool on) {
    use_lz4 = use_zstd = false;
    if (!on) return 0;
    if (!lz4_c) {
        void* h = dlopen("liblz4.so.1", RTLD_NOW | RTLD_LOCAL);
        if (h) {
            lz4_c = (lz4_compress_fn)dlsym(h, "LZ4_compress_default");
            lz4_d = (lz4_decompress_fn)dlsym(h, "LZ4_decompress_safe");
            lz4_version_fn v = (lz4_version_fn)dlsym(h, "LZ4_versionNumber");
            if (v) lz4_ver = v();
        }
    }
    if (!zstd_c) {
        void* h = dlopen("libzstd.so.1", RTLD_NOW | RTLD_LOCAL);
        if (h) {
            zstd_c = (zstd_compress_fn)dlsym(h, "ZSTD_compress");
            zstd_d = (zstd_decompress_fn)dlsym(h, "ZSTD_decompress");
            zstd_err = (zstd_iserror_fn)dlsym(h, "ZSTD_isError");
            zstd_version_fn v = (zstd_version_fn)dlsym(h, "ZSTD_versionNumber");
            if (v) zstd_ver = (int)v();
        }
    }
    use_lz4 = lz4_c && lz4_d;
    use_zstd = zstd_c && zstd_d && zstd_err;
    return (use_lz4 ? 1 : 0) | (use_zstd ? 2 : 0);
}
}  // namespace syscodec
int system_codecs_enable(int on) { return syscodec::enable(on != 0); }
int system_codec_version(int which) { return which == 0 ? syscodec::lz4_ver : syscodec::zstd_ver; }

// ============================================================ CommonCompression
// src/compression/basic.rs:62-84
static size_t common_compress(uint8_t codec, const uint8_t* in, size_t n, std::vector<uint8_t>& out) {
    size_t start = out.size();
    switch (codec) {
        case C_NONE:  // basic.rs:79-82
            put_bytes(out, in, n);
            return n;
        case C_LZ4: {  // basic.rs:108-120
            size_t bound = lz4_compress_bound(n);
            out.resize(start + bound);
            size_t sz;
            if (syscodec::use_lz4 && n < 0x7E000000u) {
                const int r = syscodec::lz4_c((const char*)in, (char*)out.data() + start, (int)n, (int)bound);
                if (r <= 0 && n) out_of_spec("LZ4_compress_default failed");
                sz = (size_t)r;
            } else {
                sz = lz4_compress(in, n, out.data() + start, bound);
            }
            out.resize(start + sz);
            return sz;
        }
        case C_ZSTD: {  // basic.rs:122-135
            size_t bound = zstd_compress_bound(n) + n / 128 + 64;
            out.resize(start + bound);
            size_t sz;
            if (syscodec::use_zstd) {
                sz = syscodec::zstd_c(out.data() + start, bound, in, n, 0);   // level 0 = the default level, as basic.rs:131
                if (syscodec::zstd_err(sz)) out_of_spec("ZSTD_compress failed");
            } else {
                sz = zstd_compress(in, n, out.data() + start, bound);
            }
            out.resize(start + sz);
            return sz;
        }
        case C_SNAPPY: {  // basic.rs:137-152
            size_t bound = snappy_compress_bound(n);
            out.resize(start + bound);
            size_t sz = snappy_compress(in, n, out.data() + start, bound);
            out.resize(start + sz);
            return sz;
        }
    }
    out_of_spec("Unknown compression codec " + std::to_string(codec));
}

// src/compression/basic.rs:62-72
static void common_decompress(uint8_t codec, const uint8_t* in, size_t n, uint8_t* dst, size_t out_len) {
    switch (codec) {
        case C_NONE:  // copy_from_slice panics on length mismatch
            if (n != out_len) out_of_spec("None block: compressed_size != expected size");
            if (n) memcpy(dst, in, n);
            return;
        case C_LZ4:
            if (syscodec::use_lz4 && n < 0x7E000000u && out_len < 0x7E000000u) {
                if (syscodec::lz4_d((const char*)in, (char*)dst, (int)n, (int)out_len) != (int)out_len) out_of_spec("LZ4_decompress_safe failed");
                return;
            }
            lz4_decompress(in, n, dst, out_len);
            return;
        case C_ZSTD:
            if (syscodec::use_zstd) {
                const size_t r = syscodec::zstd_d(dst, out_len, in, n);
                if (syscodec::zstd_err(r) || r != out_len) out_of_spec("ZSTD_decompress failed");
                return;
            }
            zstd_decompress(in, n, dst, out_len);
            return;
        case C_SNAPPY:
            snappy_decompress(in, n, dst, out_len);
            return;
    }
    out_of_spec("Unknown compression codec " + std::to_string(codec));
}

static inline bool is_basic(uint8_t c) { return c <= 3; }

// src/compression/mod.rs:64-82 from_codec
static void check_codec(uint8_t c) {
    if (c <= 3 || (c >= 10 && c <= 16)) return;
    out_of_spec("Unknown compression codec " + std::to_string(c));
}

// src/read/read_basic.rs:181-189 read_compress_header
struct Hdr9 {
    uint8_t codec;
    size_t csize, usize;
};
static Hdr9 read_hdr9(Reader& r) {
    r.need(9, "hdr9");
    Hdr9 h;
    h.codec = r.u8();
    h.csize = r.u32();
    h.usize = r.u32();
    return h;
}
static size_t begin_hdr9(std::vector<uint8_t>& out, uint8_t codec) {  // integer/mod.rs:49-52
    put_u8(out, codec);
    size_t pos = out.size();
    for (int i = 0; i < 8; i++) out.push_back(0);
    return pos;
}
static void end_hdr9(std::vector<uint8_t>& out, size_t pos, size_t csize, size_t usize) {
    patch_u32(out, pos, (uint32_t)csize);
    patch_u32(out, pos + 4, (uint32_t)usize);
}

// ============================================================ BitPacker4x  [3P bitpacking 0.8]
// 128 values seen as 32 vectors of 4 lanes; lane l packs values l, l+4, l+8 … LSB first
// into num_bits 32-bit words; word k of lane l lives at u32 index 4k+l.
uint8_t bitpack4x_num_bits(const uint32_t* in) {
    uint32_t acc = 0;
    for (int i = 0; i < 128; i++) acc |= in[i];
    return acc == 0 ? 0 : (uint8_t)(32 - __builtin_clz(acc));
}
void bitpack4x_pack(const uint32_t* in, uint8_t nb, uint8_t* out, bool delta, uint32_t initial) {
    if (nb == 0) return;
    uint32_t w[128];
    memset(w, 0, sizeof(uint32_t) * 4 * nb);
    for (int j = 0; j < 128; j++) {
        uint32_t v = in[j];
        if (delta) v = in[j] - (j ? in[j - 1] : initial);  // consecutive, wrapping
        int l = j & 3, i = j >> 2;
        unsigned bitpos = (unsigned)i * nb, word = bitpos >> 5, sh = bitpos & 31;
        w[4 * word + l] |= v << sh;  // no masking: the crate ORs raw registers
        if (sh + nb > 32) w[4 * (word + 1) + l] |= v >> (32 - sh);
    }
    memcpy(out, w, (size_t)16 * nb);
}
void bitpack4x_unpack(const uint8_t* in, uint8_t nb, uint32_t* out, bool delta, uint32_t initial) {
    uint32_t w[128];
    if (nb) memcpy(w, in, (size_t)16 * nb);
    uint32_t mask = nb >= 32 ? 0xFFFFFFFFu : ((1u << nb) - 1);
    uint32_t prev = initial;
    for (int j = 0; j < 128; j++) {
        uint32_t v = 0;
        if (nb) {
            int l = j & 3, i = j >> 2;
            unsigned bitpos = (unsigned)i * nb, word = bitpos >> 5, sh = bitpos & 31;
            v = w[4 * word + l] >> sh;
            if (sh + nb > 32) v |= w[4 * (word + 1) + l] << (32 - sh);
            v &= mask;
        }
        if (delta) {
            v += prev;
            prev = v;
        }
        out[j] = v;
    }
}

// ============================================================ Roaring portable format [3P roaring 0.10.1]
static void roaring_serialize(const std::vector<uint32_t>& sorted, std::vector<uint8_t>& out) {
    struct C {
        uint16_t key;
        size_t b, e;
    };
    std::vector<C> cs;
    for (size_t i = 0; i < sorted.size();) {
        size_t j = i;
        uint16_t key = (uint16_t)(sorted[i] >> 16);
        while (j < sorted.size() && (uint16_t)(sorted[j] >> 16) == key) j++;
        cs.push_back({key, i, j});
        i = j;
    }
    put_u32(out, 12346);  // SERIAL_COOKIE_NO_RUNCONTAINER
    put_u32(out, (uint32_t)cs.size());
    for (auto& c : cs) {
        put_u16(out, c.key);
        put_u16(out, (uint16_t)(c.e - c.b - 1));
    }
    uint32_t off = 8 + 8 * (uint32_t)cs.size();
    for (auto& c : cs) {
        put_u32(out, off);
        size_t card = c.e - c.b;
        off += card > 4096 ? 8192 : (uint32_t)(2 * card);
    }
    for (auto& c : cs) {
        size_t card = c.e - c.b;
        if (card > 4096) {
            uint64_t bm[1024];
            memset(bm, 0, sizeof bm);
            for (size_t i = c.b; i < c.e; i++) {
                uint32_t lo = sorted[i] & 0xFFFF;
                bm[lo >> 6] |= 1ull << (lo & 63);
            }
            put_bytes(out, bm, sizeof bm);
        } else {
            for (size_t i = c.b; i < c.e; i++) put_u16(out, (uint16_t)(sorted[i] & 0xFFFF));
        }
    }
}
static std::vector<uint32_t> roaring_deserialize(const uint8_t* p, size_t n) {
    Reader r(p, n);
    uint32_t cookie = r.u32("roaring cookie");
    size_t nc;
    std::vector<uint8_t> is_run;
    bool has_run = false;
    if ((cookie & 0xFFFF) == 12347) {
        has_run = true;
        nc = (cookie >> 16) + 1;
        const uint8_t* rb = r.take((nc + 7) / 8, "roaring run bitmap");
        is_run.assign(rb, rb + (nc + 7) / 8);
    } else if (cookie == 12346) {
        nc = r.u32("roaring size");
    } else {
        out_of_spec("roaring: unknown cookie");
    }
    std::vector<std::pair<uint16_t, uint32_t>> desc(nc);
    for (size_t i = 0; i < nc; i++) {
        desc[i].first = r.u16();
        desc[i].second = (uint32_t)r.u16() + 1;
    }
    if (!has_run || nc >= 4) r.take(4 * nc, "roaring offsets");
    std::vector<uint32_t> res;
    for (size_t i = 0; i < nc; i++) {
        uint32_t hi = (uint32_t)desc[i].first << 16;
        uint32_t card = desc[i].second;
        bool run = has_run && ((is_run[i >> 3] >> (i & 7)) & 1);
        if (run) {
            uint16_t nr = r.u16();
            for (uint16_t k = 0; k < nr; k++) {
                uint32_t s = r.u16(), l = r.u16();
                for (uint32_t v = s; v <= s + l; v++) res.push_back(hi | v);
            }
        } else if (card > 4096) {
            const uint8_t* b = r.take(8192, "roaring bitmap container");
            for (uint32_t v = 0; v < 65536; v++)
                if ((b[v >> 3] >> (v & 7)) & 1) res.push_back(hi | v);
        } else {
            for (uint32_t k = 0; k < card; k++) res.push_back(hi | r.u16());
        }
    }
    return res;
}

// ============================================================ primitive arrays
template <class T>
struct PrimArr {
    const T* v;
    size_t n;
    Validity val;
    T at(size_t i) const {
        T x;
        memcpy(&x, (const uint8_t*)v + i * sizeof(T), sizeof(T));
        return x;
    }
};

// IntegerStats / DoubleStats (integer/mod.rs:164-177, double/mod.rs:162-176)
template <class T>
struct Stats {
    PrimArr<T> src;
    size_t tuple_count = 0, total_bytes = 0, null_count = 0, unique_count = 0, set_count = 0;
    double average_run_length = 0;
    bool is_sorted = true;
    T min, max;
    std::unordered_map<typename Ops<T>::Key, size_t, KeyHash<typename Ops<T>::Key>> distinct;
};

// gen_stats: integer/mod.rs:179-229, double/mod.rs:178-229
template <class T>
static void gen_stats(const PrimArr<T>& a, Stats<T>& s) {
    typedef Ops<T> O;
    s.src = a;
    s.tuple_count = a.n;
    s.total_bytes = a.n * sizeof(T);
    s.null_count = a.val.null_count(a.n);
    s.set_count = a.n - s.null_count;
    s.is_sorted = true;
    s.min = zero_value<T>();
    s.max = zero_value<T>();
    bool init = false;
    T last = zero_value<T>();
    size_t run_count = 0;
    s.distinct.reserve(1024);
    for (size_t i = 0; i < a.n; i++) {
        T cur = a.at(i);
        if (a.val.get(i)) {
            if (O::lt(cur, last)) s.is_sorted = false;
            if (!O::eq(last, cur)) {
                run_count++;
                last = cur;
            }
        }
        s.distinct[O::key(cur)] += 1;  // null slots are counted too (mod.rs:211)
        if (!init) {
            init = true;
            s.min = cur;
            s.max = cur;
        }
        if (O::lt(s.max, cur))
            s.max = cur;
        else if (O::lt(cur, s.min))
            s.min = cur;
    }
    s.unique_count = s.distinct.size();
    s.average_run_length = (double)a.n / (double)run_count;
}

template <class T>
static void compress_prim(const PrimArr<T>& a, const WriteOptions& opts, uint32_t depth,
                          std::vector<uint8_t>& out);
template <class T>
static void decompress_prim(Reader& r, size_t length, std::vector<uint8_t>& out);

template <class T>
static inline T* grow(std::vector<uint8_t>& out, size_t n) {
    size_t old = out.size();
    out.resize(old + n * sizeof(T));
    return (T*)(out.data() + old);
}

// ------------------------------------------------------------ RLE
// integer/rle.rs:64-104, double/rle.rs:61-103
template <class T>
static size_t rle_compress(const PrimArr<T>& a, std::vector<uint8_t>& out) {
    typedef Ops<T> O;
    size_t start = out.size();
    uint32_t seen = 0;
    T last = zero_value<T>();
    bool all_null = true;
    for (size_t i = 0; i < a.n; i++) {
        T item = a.at(i);
        if (a.val.get(i)) {
            if (all_null) {
                all_null = false;
                last = item;
                seen += 1;
            } else if (!O::eq(last, item)) {
                put_u32(out, seen);
                put_bytes(out, &last, sizeof(T));
                last = item;
                seen = 1;
            } else {
                seen += 1;
            }
        } else {
            seen += 1;  // nulls extend the current run
        }
    }
    if (seen != 0) {
        put_u32(out, seen);
        put_bytes(out, &last, sizeof(T));
    }
    return out.size() - start;
}
// integer/rle.rs:106-134
template <class T>
static void rle_decompress(Reader& in, size_t length, std::vector<uint8_t>& out) {
    size_t num = 0;
    for (;;) {
        uint32_t len = in.u32("rle count");
        const uint8_t* vb = in.take(sizeof(T), "rle value");
        T* dst = grow<T>(out, len);
        for (uint32_t i = 0; i < len; i++) memcpy((uint8_t*)dst + (size_t)i * sizeof(T), vb, sizeof(T));
        num += len;
        if (num >= length) break;
    }
}

// ------------------------------------------------------------ OneValue
// integer/one_value.rs:63-94
template <class T>
static size_t onevalue_compress(const PrimArr<T>& a, std::vector<uint8_t>& out) {
    T val = zero_value<T>();
    for (size_t i = 0; i < a.n; i++)
        if (a.val.get(i)) {
            val = a.at(i);
            break;
        }
    put_bytes(out, &val, sizeof(T));
    return sizeof(T);
}
template <class T>
static void onevalue_decompress(Reader& in, size_t length, std::vector<uint8_t>& out) {
    const uint8_t* vb = in.take(sizeof(T), "onevalue");
    T* dst = grow<T>(out, length);
    for (size_t i = 0; i < length; i++) memcpy((uint8_t*)dst + i * sizeof(T), vb, sizeof(T));
}

// ------------------------------------------------------------ Dict
// integer/dict.rs:33-73 + DictEncoder :133-223; double/dict.rs:38-77
template <class T>
static size_t dict_compress(const PrimArr<T>& a, const WriteOptions& opts, uint32_t depth,
                            std::vector<uint8_t>& out) {
    size_t start = out.size();
    std::vector<uint32_t> indices;
    indices.reserve(a.n);
    std::vector<T> sets;
    // interner keyed on the raw value bytes (RawNative::as_bytes, dict.rs:231-239);
    // equality is T's == so a float NaN never matches an earlier entry (dict.rs:208).
    std::unordered_map<typename Ops<T>::Key, uint32_t, KeyHash<typename Ops<T>::Key>> map;
    auto raw_key = [](T v) {
        if constexpr (Ops<T>::is_float) {
            uint64_t k = 0;
            memcpy(&k, &v, sizeof(T));
            return k;
        } else {
            return Ops<T>::key(v);
        }
    };
    auto intern = [&](T v) -> uint32_t {
        bool never_equal = false;
        if constexpr (Ops<T>::is_float) never_equal = std::isnan(v);
        if (!never_equal) {
            auto it = map.find(raw_key(v));
            if (it != map.end()) return it->second;
        }
        uint32_t k = (uint32_t)sets.size();
        sets.push_back(v);
        if (!never_equal) map.emplace(raw_key(v), k);
        return k;
    };
    for (size_t i = 0; i < a.n; i++) {
        if (a.val.get(i)) {
            indices.push_back(intern(a.at(i)));
        } else if (indices.empty()) {
            indices.push_back(intern(zero_value<T>()));
        } else {
            indices.push_back(indices.back());
        }
    }
    WriteOptions nested = opts;
    nested.forbidden_mask |= 1u << C_DICT;
    nested.force_codec = opts.force_index_codec;
    nested.force_index_codec = -1;
    PrimArr<uint32_t> ia{indices.data(), indices.size(), Validity{}};
    compress_prim<uint32_t>(ia, nested, depth + 1, out);
    put_u32(out, (uint32_t)sets.size());
    for (auto& v : sets) put_bytes(out, &v, sizeof(T));
    return out.size() - start;
}
// integer/dict.rs:75-103
template <class T>
static void dict_decompress(Reader& in, size_t length, std::vector<uint8_t>& out) {
    std::vector<uint8_t> ibytes;
    decompress_prim<uint32_t>(in, length, ibytes);
    size_t data_size = (size_t)in.u32("dict size") * sizeof(T);
    if (in.left() < data_size) out_of_spec("Invalid data size");
    const uint8_t* data = in.take(data_size);
    size_t ni = ibytes.size() / 4;
    T* dst = grow<T>(out, ni);
    for (size_t i = 0; i < ni; i++) {
        uint32_t idx;
        memcpy(&idx, ibytes.data() + 4 * i, 4);
        if ((size_t)idx * sizeof(T) >= data_size) out_of_spec("dict index out of bounds");
        memcpy((uint8_t*)dst + i * sizeof(T), data + (size_t)idx * sizeof(T), sizeof(T));
    }
}

// ------------------------------------------------------------ Freq
// top value with a deterministic tie-break (reference: HashMap iteration order,
// integer/freq.rs:50-55 — SURVEY App. B#5): highest count, earliest first occurrence.
template <class T>
static void freq_top(const PrimArr<T>& a, const Stats<T>& s, T& top, size_t& max_count) {
    typedef Ops<T> O;
    max_count = 0;
    top = zero_value<T>();
    std::unordered_map<typename O::Key, bool, KeyHash<typename O::Key>> seen;
    for (size_t i = 0; i < a.n; i++) {
        T v = a.at(i);
        auto k = O::key(v);
        if (seen.count(k)) continue;
        seen[k] = true;
        size_t c = s.distinct.at(k);
        if (c > max_count) {
            max_count = c;
            top = v;
        }
    }
}
// integer/freq.rs:34-88, double/freq.rs
template <class T>
static size_t freq_compress(const PrimArr<T>& a, const Stats<T>& s, const WriteOptions& opts, uint32_t depth,
                            std::vector<uint8_t>& out) {
    typedef Ops<T> O;
    size_t start = out.size();
    bool top_is_null = false;
    T top = zero_value<T>();
    size_t max_count = 0;
    if ((double)s.null_count / (double)s.tuple_count >= 0.9)
        top_is_null = true;
    else
        freq_top(a, s, top, max_count);
    std::vector<uint32_t> ex_idx;
    std::vector<T> ex;
    for (size_t i = 0; i < a.n; i++) {
        if (a.val.get(i)) {
            T v = a.at(i);
            if (top_is_null || !O::eq(v, top)) {
                ex_idx.push_back((uint32_t)i);
                ex.push_back(v);
            }
        }
    }
    put_bytes(out, &top, sizeof(T));
    std::vector<uint8_t> rb;
    roaring_serialize(ex_idx, rb);
    put_u32(out, (uint32_t)rb.size());
    put_bytes(out, rb.data(), rb.size());
    WriteOptions nested = opts;
    nested.forbidden_mask |= 1u << C_FREQ;
    nested.force_codec = opts.force_index_codec;
    nested.force_index_codec = -1;
    PrimArr<T> ea{ex.data(), ex.size(), Validity{}};
    compress_prim<T>(ea, nested, depth + 1, out);
    return out.size() - start;
}
// integer/freq.rs:90-127
template <class T>
static void freq_decompress(Reader& in, size_t length, std::vector<uint8_t>& out) {
    size_t begin = out.size();
    const uint8_t* tb = in.take(sizeof(T), "freq top");
    T* dst = grow<T>(out, length);
    for (size_t i = 0; i < length; i++) memcpy((uint8_t*)dst + i * sizeof(T), tb, sizeof(T));
    uint32_t rbsize = in.u32("freq bitmap size");
    const uint8_t* rbp = in.take(rbsize, "freq bitmap");
    std::vector<uint32_t> idx = roaring_deserialize(rbp, rbsize);
    std::vector<uint8_t> ex;
    decompress_prim<T>(in, idx.size(), ex);
    if (ex.size() / sizeof(T) != idx.size()) out_of_spec("freq: exceptions length mismatch");
    for (size_t i = 0; i < idx.size(); i++) {
        if (idx[i] >= length) out_of_spec("freq: exception index out of bounds");
        memcpy(out.data() + begin + (size_t)idx[i] * sizeof(T), ex.data() + i * sizeof(T), sizeof(T));
    }
}

// ------------------------------------------------------------ Bitpacking / DeltaBitpacking (4-byte ints)
// integer/bp.rs:36-64, integer/delta_bp.rs:36-68
template <class T>
static size_t bp_compress(const PrimArr<T>& a, bool delta, std::vector<uint8_t>& out) {
    if (sizeof(T) != 4) throw Error(-4, "Bitpacking needs a 4-byte integer type");
    if (a.n % 128 != 0) throw Error(-4, "Bitpacking needs len % 128 == 0 (the crate asserts BLOCK_LEN)");
    size_t start = out.size();
    uint32_t initial = 0;
    uint32_t blk[128];
    for (size_t b = 0; b < a.n; b += 128) {
        memcpy(blk, (const uint8_t*)a.v + b * 4, 512);
        uint8_t nb = bitpack4x_num_bits(blk);  // from RAW values, also for delta (delta_bp.rs:50)
        out.push_back(nb);
        size_t pos = out.size();
        out.resize(pos + 16 * (size_t)nb);
        bitpack4x_pack(blk, nb, out.data() + pos, delta, initial);
        initial = blk[127];
    }
    return out.size() - start;
}
// integer/bp.rs:66-86, integer/delta_bp.rs:70-92 — always whole 128-blocks
template <class T>
static void bp_decompress(Reader& in, size_t length, bool delta, std::vector<uint8_t>& out) {
    if (sizeof(T) != 4) throw Error(-4, "Bitpacking needs a 4-byte integer type");
    uint32_t initial = 0;
    for (size_t i = 0; i < length; i += 128) {
        uint8_t nb = in.u8("bitpack num_bits");
        if (nb > 32) out_of_spec("bitpack: num_bits > 32");
        const uint8_t* p = in.take(16 * (size_t)nb, "bitpack block");
        uint32_t* dst = grow<uint32_t>(out, 128);
        uint32_t tmp[128];
        bitpack4x_unpack(p, nb, tmp, delta, initial);
        memcpy(dst, tmp, 512);
        initial = tmp[127];
    }
}

// ------------------------------------------------------------ Patas (floats)
uint32_t patas_pack(uint32_t ref_diff, uint32_t sig_bytes, uint32_t tz) {  // patas.rs:145-149
    return ((ref_diff & 0xFF) << 9 | ((sig_bytes & 7) << 6) | (tz & 0xFF)) & 0xFFFF;
}
template <class T, class B>
static size_t patas_compress(const PrimArr<T>& a, std::vector<uint8_t>& out) {  // patas.rs:36-104
    size_t start = out.size();
    const size_t BLOCK = 128;
    const unsigned W = sizeof(T) * 8;
    std::unordered_map<uint64_t, size_t> last_index;
    std::vector<B> bits(a.n);
    for (size_t i = 0; i < a.n; i++) {
        T v = a.at(i);
        B b;
        memcpy(&b, &v, sizeof(T));
        bits[i] = b;
        if (i == 0) {
            put_bytes(out, &b, sizeof(T));
        } else {
            size_t ref = 0;
            auto it = last_index.find((uint64_t)b);
            if (it != last_index.end()) ref = it->second;
            if (ref > i || (i - ref) >= BLOCK) ref = i - 1;
            size_t diff = i - ref;
            B x = b ^ bits[i - diff];
            unsigned tz = x == 0 ? W : (sizeof(T) == 8 ? (unsigned)__builtin_ctzll((uint64_t)x) : (unsigned)__builtin_ctz((uint32_t)x));
            unsigned lz = x == 0 ? W : (sizeof(T) == 8 ? (unsigned)__builtin_clzll((uint64_t)x) : (unsigned)__builtin_clz((uint32_t)x));
            unsigned is_equal = tz == W ? 1 : 0;
            unsigned sig_bits = is_equal ? 0 : W - tz - lz;
            unsigned sig_bytes = (sig_bits >> 3) + ((sig_bits & 7) != 0);
            put_u16(out, (uint16_t)patas_pack((uint32_t)diff, sig_bytes, tz - is_equal));
            B sh = (B)(x >> (tz - is_equal));
            put_bytes(out, &sh, sig_bytes);
        }
        last_index[(uint64_t)b] = i;
    }
    return out.size() - start;
}
template <class T, class B>
static void patas_decompress(Reader& in, size_t length, std::vector<uint8_t>& out) {  // patas.rs:106-133
    if (sizeof(T) != 8) throw Error(-4, "Patas f32 decode is broken upstream (SURVEY App. B#10)");
    if (length == 0) out_of_spec("patas: zero length");
    size_t begin = out.size();
    const uint8_t* fb = in.take(sizeof(T), "patas first");
    grow<T>(out, length);
    memcpy(out.data() + begin, fb, sizeof(T));
    for (size_t i = 1; i < length; i++) {
        uint16_t pk = in.u16("patas packed");
        unsigned diff = (pk >> 9) & 0x7F, sb = (pk >> 6) & 7, tz = pk & 0x3F;
        if (tz < 63 && sb == 0) sb = 8;  // unpack, patas.rs:152-163
        B val = 0;
        memcpy(&val, in.take(sb, "patas value"), sb);
        if (diff == 0 || diff > i) out_of_spec("patas: bad reference");
        B prev;
        memcpy(&prev, out.data() + begin + (i - diff) * sizeof(T), sizeof(T));
        B x = (B)((val << tz) ^ prev);
        memcpy(out.data() + begin + i * sizeof(T), &x, sizeof(T));
    }
}

// ------------------------------------------------------------ sampling (compress_sample_ratio)
// integer/mod.rs:310-347, double/mod.rs:309-347 with an injectable RNG (sbo.h sample_rand)
template <class T>
struct OwnedArr {
    std::vector<T> v;
    std::vector<uint8_t> vb;
    bool has_val = false;
    PrimArr<T> view() const {
        Validity val;
        val.present = has_val;
        val.bits = Bits{vb.data(), 0};
        return PrimArr<T>{v.data(), v.size(), val};
    }
};
static const size_t SAMPLE_COUNT = 10, SAMPLE_SIZE = 64;  // compression/mod.rs:30-33

template <class T>
static bool build_sample(const PrimArr<T>& src, uint64_t seed, uint32_t depth, uint32_t codec, OwnedArr<T>& s) {
    if (src.n / SAMPLE_COUNT <= SAMPLE_SIZE) return false;
    size_t separator = src.n / SAMPLE_COUNT, remainder = src.n % SAMPLE_COUNT;
    s.has_val = src.val.present;
    s.v.reserve(SAMPLE_COUNT * SAMPLE_SIZE);
    s.vb.assign((SAMPLE_COUNT * SAMPLE_SIZE + 7) / 8, 0);
    size_t k = 0;
    for (size_t si = 0; si < SAMPLE_COUNT; si++) {
        size_t range_end = (si == SAMPLE_COUNT - 1 ? separator + remainder : separator) - SAMPLE_SIZE;
        size_t begin = si * separator + (size_t)sample_rand(seed, depth, codec, (uint32_t)si, range_end);
        for (size_t j = 0; j < SAMPLE_SIZE; j++, k++) {
            bool valid = src.val.get(begin + j);
            // MutablePrimitiveArray pushes T::default() for None
            s.v.push_back(valid ? src.at(begin + j) : zero_value<T>());
            if (valid) s.vb[k >> 3] |= (uint8_t)(1u << (k & 7));
        }
    }
    return true;
}

template <class T>
static size_t ext_compress(uint8_t codec, const PrimArr<T>& a, const Stats<T>& st, const WriteOptions& opts,
                           uint32_t depth, std::vector<uint8_t>& out);

template <class T>
static double sample_ratio(uint8_t compress_with, uint8_t trial_id, const Stats<T>& stats, const WriteOptions& opts,
                           uint32_t depth) {
    OwnedArr<T> owned;
    Stats<T> sstats;
    const Stats<T>* st = &stats;
    if (build_sample(stats.src, opts.rng_seed, depth, trial_id, owned)) {
        gen_stats(owned.view(), sstats);
        st = &sstats;
    }
    std::vector<uint8_t> tmp;
    size_t size;
    try {
        WriteOptions def;  // WriteOptions::default()
        size = ext_compress<T>(compress_with, st->src, *st, def, depth, tmp);
    } catch (const Error&) {
        size = st->total_bytes;  // .unwrap_or(stats.total_bytes)
    }
    return (double)st->total_bytes / (double)size;
}

// per-codec compress_ratio
template <class T>
static double codec_ratio(uint8_t c, const Stats<T>& s, const WriteOptions& opts, uint32_t depth) {
    typedef Ops<T> O;
    switch (c) {
        case C_ONEVALUE:  // one_value.rs:53-59
            return s.unique_count <= 1 ? (double)s.tuple_count : 0.0;
        case C_FREQ: {  // freq.rs:129-151
            if (s.unique_count <= 1) return 0.0;
            if ((double)s.null_count / (double)s.tuple_count >= 0.9) return (double)(s.tuple_count - 1);
            size_t max_count = 0;
            for (auto& kv : s.distinct) max_count = std::max(max_count, kv.second);
            bool big = O::is_float ? true : (O::as_i64(s.max) >= (1 << 8));
            if ((double)max_count / (double)s.tuple_count >= 0.9 && big) return (double)(s.tuple_count - 1);
            return 0.0;
        }
        case C_DICT: {  // dict.rs:109-120
            if (s.unique_count * 3 >= s.tuple_count) return 0.0;
            size_t after = s.unique_count * sizeof(T) +
                           s.tuple_count * (size_t)(get_bits_needed((uint64_t)s.unique_count) / 8);
            after += s.tuple_count * 2 / 128;
            return (double)s.total_bytes / (double)after;
        }
        case C_RLE:  // rle.rs:58-60
            return sample_ratio<T>(C_RLE, C_RLE, s, opts, depth);
        case C_BITPACK:  // bp.rs:92-100
            if (O::is_float || O::as_i64(s.min) < 0 || sizeof(T) != 4 || s.src.n % 128 != 0) return 0.0;
            return sample_ratio<T>(C_BITPACK, C_BITPACK, s, opts, depth);
        case C_DELTABP:  // delta_bp.rs:97-109
            if (O::is_float || O::as_i64(s.min) < 0 || sizeof(T) != 4 || s.src.n % 128 != 0 || !s.is_sorted ||
                s.null_count > 0)
                return 0.0;
            return sample_ratio<T>(C_BITPACK, C_DELTABP, s, opts, depth) * 1.5;
        case C_PATAS:  // patas.rs:139-141
            return sample_ratio<T>(C_PATAS, C_PATAS, s, opts, depth);
    }
    return 0.0;
}

// choose_compressor: integer/mod.rs:231-308, double/mod.rs:231-307
template <class T>
static uint8_t choose_prim(const Stats<T>& s, const WriteOptions& opts, uint32_t depth) {
    if (opts.force_codec >= 0 && !opts.forbidden((uint8_t)opts.force_codec)) return (uint8_t)opts.force_codec;
    uint8_t result = opts.default_compression;
    if (!opts.has_ratio) return result;
    double max_ratio = opts.ratio;
    static const uint8_t INT_ORDER[] = {C_ONEVALUE, C_FREQ, C_DICT, C_RLE, C_BITPACK, C_DELTABP};
    static const uint8_t DBL_ORDER[] = {C_ONEVALUE, C_FREQ, C_DICT, C_PATAS, C_RLE};
    const uint8_t* order = Ops<T>::is_float ? DBL_ORDER : INT_ORDER;
    size_t norder = Ops<T>::is_float ? 5 : 6;
    for (size_t i = 0; i < norder; i++) {
        uint8_t c = order[i];
        if (opts.forbidden(c)) continue;
        double r = codec_ratio<T>(c, s, opts, depth);
        if (r > max_ratio) {
            max_ratio = r;
            result = c;
            if (r == (double)s.tuple_count) break;
        }
    }
    return result;
}

template <class T>
static size_t ext_compress(uint8_t codec, const PrimArr<T>& a, const Stats<T>& st, const WriteOptions& opts,
                           uint32_t depth, std::vector<uint8_t>& out) {
    switch (codec) {
        case C_RLE:
            return rle_compress<T>(a, out);
        case C_DICT:
            return dict_compress<T>(a, opts, depth, out);
        case C_ONEVALUE:
            return onevalue_compress<T>(a, out);
        case C_FREQ:
            return freq_compress<T>(a, st, opts, depth, out);
        case C_BITPACK:
            if (Ops<T>::is_float) break;
            return bp_compress<T>(a, false, out);
        case C_DELTABP:
            if (Ops<T>::is_float) break;
            return bp_compress<T>(a, true, out);
        case C_PATAS:
            if constexpr (std::is_same<T, double>::value) return patas_compress<double, uint64_t>(a, out);
            if constexpr (std::is_same<T, float>::value) return patas_compress<float, uint32_t>(a, out);
            break;
    }
    out_of_spec("Unknown compression codec " + std::to_string(codec) + " for this type");
}

// compress_integer / compress_double: integer/mod.rs:35-70, double/mod.rs:32-67
template <class T>
static void compress_prim(const PrimArr<T>& a, const WriteOptions& opts, uint32_t depth,
                          std::vector<uint8_t>& out) {
    Stats<T> stats;
    gen_stats(a, stats);  // runs unconditionally (mod.rs:41)
    uint8_t codec = choose_prim<T>(stats, opts, depth);
    size_t pos = begin_hdr9(out, codec);
    size_t csize;
    if (is_basic(codec))
        csize = common_compress(codec, (const uint8_t*)a.v, a.n * sizeof(T), out);
    else
        csize = ext_compress<T>(codec, a, stats, opts, depth, out);
    end_hdr9(out, pos, csize, a.n * sizeof(T));
}

// decompress_integer / decompress_double: integer/mod.rs:72-117
template <class T>
static void decompress_prim(Reader& r, size_t length, std::vector<uint8_t>& out) {
    Hdr9 h = read_hdr9(r);
    check_codec(h.codec);
    r.need(h.csize, "compressed block");
    if (is_basic(h.codec)) {
        T* dst = grow<T>(out, length);
        common_decompress(h.codec, r.p, h.csize, (uint8_t*)dst, length * sizeof(T));
    } else {
        // Extend codecs see the rest of the buffer (mod.rs:108-110), then csize is consumed
        Reader in(r.p, r.left());
        switch (h.codec) {
            case C_RLE:
                rle_decompress<T>(in, length, out);
                break;
            case C_DICT:
                dict_decompress<T>(in, length, out);
                break;
            case C_ONEVALUE:
                onevalue_decompress<T>(in, length, out);
                break;
            case C_FREQ:
                freq_decompress<T>(in, length, out);
                break;
            case C_BITPACK:
            case C_DELTABP:
                if (Ops<T>::is_float) out_of_spec("Unknown compression codec for double");
                bp_decompress<T>(in, length, h.codec == C_DELTABP, out);
                break;
            case C_PATAS:
                if constexpr (std::is_same<T, double>::value)
                    patas_decompress<double, uint64_t>(in, length, out);
                else if constexpr (std::is_same<T, float>::value)
                    patas_decompress<float, uint32_t>(in, length, out);
                else
                    out_of_spec("Unknown compression codec Patas for integer");
                break;
            default:
                out_of_spec("Unknown compression codec " + std::to_string(h.codec));
        }
    }
    r.p += h.csize;
}

// ============================================================ binary
template <class O>
struct BinArr {
    const O* offsets;  // n+1
    size_t n;
    const uint8_t* values;
    uint64_t values_len_total;  // array.values().len(): the WHOLE shared buffer (slices keep it)
    Validity val;
    O off(size_t i) const {
        O x;
        memcpy(&x, (const uint8_t*)offsets + i * sizeof(O), sizeof(O));
        return x;
    }
    std::string_view get(size_t i) const {
        return std::string_view((const char*)values + off(i), (size_t)(off(i + 1) - off(i)));
    }
};
struct BinStats {  // binary/mod.rs:255-291
    size_t tuple_count = 0, total_bytes = 0, unique_count = 0, total_unique_size = 0, null_count = 0;
    std::unordered_map<std::string_view, size_t> distinct;
};
template <class O>
static void gen_bin_stats(const BinArr<O>& a, BinStats& s) {
    s.tuple_count = a.n;
    s.total_bytes = a.values_len_total + (a.n + 1) * sizeof(O);
    s.null_count = a.val.null_count(a.n);
    for (size_t i = 0; i < a.n; i++) s.distinct[a.get(i)] += 1;
    for (auto& kv : s.distinct) s.total_unique_size += kv.first.size() + 8;
    s.unique_count = s.distinct.size();
}
static double bin_ratio(uint8_t c, const BinStats& s) {
    switch (c) {
        case C_ONEVALUE:  // binary/one_value.rs:42-48
            return s.unique_count <= 1 ? (double)s.tuple_count : 0.0;
        case C_FREQ: {  // binary/freq.rs:147-169
            if (s.unique_count <= 1) return 0.0;
            if ((double)s.null_count / (double)s.tuple_count >= 0.9) return (double)(s.tuple_count - 1);
            size_t max_count = 0;
            for (auto& kv : s.distinct) max_count = std::max(max_count, kv.second);
            if ((double)max_count / (double)s.tuple_count >= 0.9) return (double)(s.tuple_count - 1);
            return 0.0;
        }
        case C_DICT: {  // binary/dict.rs:43-53
            if (s.unique_count * 3 >= s.tuple_count) return 0.0;
            size_t after = s.total_unique_size +
                           s.tuple_count * (size_t)(get_bits_needed((uint64_t)s.unique_count) / 8);
            after += s.tuple_count * 2 / 128;
            return (double)s.total_bytes / (double)after;
        }
    }
    return 0.0;
}
static uint8_t choose_bin(const BinStats& s, const WriteOptions& opts) {  // binary/mod.rs:293-348
    if (opts.force_codec >= 0 && !opts.forbidden((uint8_t)opts.force_codec)) return (uint8_t)opts.force_codec;
    uint8_t result = opts.default_compression;
    if (!opts.has_ratio) return result;
    double max_ratio = opts.ratio;
    static const uint8_t ORDER[] = {C_ONEVALUE, C_FREQ, C_DICT};
    for (uint8_t c : ORDER) {
        if (opts.forbidden(c)) continue;
        double r = bin_ratio(c, s);
        if (r > max_ratio) {
            max_ratio = r;
            result = c;
            if (r == (double)s.tuple_count) break;
        }
    }
    return result;
}

template <class O>
static void compress_binary(const BinArr<O>& a, const WriteOptions& opts, std::vector<uint8_t>& out) {
    BinStats stats;
    gen_bin_stats(a, stats);
    uint8_t codec = choose_bin(stats, opts);
    if (is_basic(codec)) {  // binary/mod.rs:42-81
        std::vector<O> zero_offsets;
        const uint8_t* obuf = (const uint8_t*)a.offsets;
        O first = a.off(0);
        if (first != 0) {
            zero_offsets.resize(a.n + 1);
            for (size_t i = 0; i <= a.n; i++) zero_offsets[i] = a.off(i) - first;
            obuf = (const uint8_t*)zero_offsets.data();
        }
        size_t olen = (a.n + 1) * sizeof(O);
        size_t pos = begin_hdr9(out, codec);
        size_t cs = common_compress(codec, obuf, olen, out);
        end_hdr9(out, pos, cs, olen);
        size_t vlen = (size_t)(a.off(a.n) - first);
        pos = begin_hdr9(out, codec);
        cs = common_compress(codec, a.values + first, vlen, out);
        end_hdr9(out, pos, cs, vlen);
        return;
    }
    size_t pos = begin_hdr9(out, codec);
    size_t start = out.size();
    switch (codec) {
        case C_ONEVALUE: {  // binary/one_value.rs:50-68
            std::string_view v;
            for (size_t i = 0; i < a.n; i++)
                if (a.val.get(i)) {
                    v = a.get(i);
                    break;
                }
            put_u32(out, (uint32_t)v.size());
            put_bytes(out, v.data(), v.size());
            break;
        }
        case C_DICT: {  // binary/dict.rs:55-93
            std::vector<uint32_t> indices;
            indices.reserve(a.n);
            std::vector<std::string_view> sets;
            std::unordered_map<std::string_view, uint32_t> map;
            for (size_t i = 0; i < a.n; i++) {
                if (!a.val.get(i) && !indices.empty()) {
                    indices.push_back(indices.back());
                } else {
                    std::string_view v = a.get(i);  // a leading null interns its slot bytes
                    auto it = map.find(v);
                    if (it == map.end()) {
                        uint32_t k = (uint32_t)sets.size();
                        sets.push_back(v);
                        map.emplace(v, k);
                        indices.push_back(k);
                    } else {
                        indices.push_back(it->second);
                    }
                }
            }
            WriteOptions nested = opts;
            nested.forbidden_mask |= 1u << C_DICT;
            nested.force_codec = opts.force_index_codec;
            nested.force_index_codec = -1;
            PrimArr<uint32_t> ia{indices.data(), indices.size(), Validity{}};
            compress_prim<uint32_t>(ia, nested, 1, out);
            put_u32(out, (uint32_t)sets.size());
            for (auto& v : sets) {
                put_u64(out, (uint64_t)v.size());
                put_bytes(out, v.data(), v.size());
            }
            break;
        }
        case C_FREQ: {  // binary/freq.rs:44-101
            bool top_is_null = false;
            std::string_view top;
            size_t max_count = 0;
            if ((double)stats.null_count / (double)stats.tuple_count >= 0.9) {
                top_is_null = true;
            } else {  // deterministic tie-break: earliest first occurrence
                std::unordered_map<std::string_view, bool> seen;
                for (size_t i = 0; i < a.n; i++) {
                    std::string_view v = a.get(i);
                    if (seen.count(v)) continue;
                    seen[v] = true;
                    size_t c = stats.distinct.at(v);
                    if (c > max_count) {
                        max_count = c;
                        top = v;
                    }
                }
            }
            std::vector<uint32_t> ex_idx;
            for (size_t i = 0; i < a.n; i++)
                if (a.val.get(i) && (top_is_null || a.get(i) != top)) ex_idx.push_back((uint32_t)i);
            put_u64(out, (uint64_t)top.size());
            put_bytes(out, top.data(), top.size());
            std::vector<uint8_t> rb;
            roaring_serialize(ex_idx, rb);
            put_u32(out, (uint32_t)rb.size());
            put_bytes(out, rb.data(), rb.size());
            for (uint32_t i : ex_idx) {
                std::string_view v = a.get(i);
                put_u64(out, (uint64_t)v.size());
                put_bytes(out, v.data(), v.size());
            }
            break;
        }
        default:
            out_of_spec("Unknown compression codec " + std::to_string(codec) + " for binary");
    }
    end_hdr9(out, pos, out.size() - start, (size_t)a.values_len_total);  // binary/mod.rs:83-88
}

template <class O>
static inline void push_off(std::vector<uint8_t>& offs, uint64_t v) {
    O o = (O)v;
    put_bytes(offs, &o, sizeof(O));
}
template <class O>
static inline O last_off(const std::vector<uint8_t>& offs) {
    O o;
    memcpy(&o, offs.data() + offs.size() - sizeof(O), sizeof(O));
    return o;
}

// binary/mod.rs:95-183
template <class O>
static void decompress_binary(Reader& r, size_t length, std::vector<uint8_t>& offs, std::vector<uint8_t>& values) {
    Hdr9 h = read_hdr9(r);
    check_codec(h.codec);
    r.need(h.csize, "compressed block");
    if (is_basic(h.codec)) {
        bool have_last = !offs.empty();
        O last = have_last ? last_off<O>(offs) : 0;
        size_t old = offs.size();
        offs.resize(old + (length + 1) * sizeof(O));
        common_decompress(h.codec, r.p, h.csize, offs.data() + old, (length + 1) * sizeof(O));
        r.p += h.csize;
        if (have_last) {  // fix offset (mod.rs:136-144): drop the duplicate leading 0, rebase
            O* base = (O*)(offs.data() + old);
            for (size_t i = 0; i < length; i++) {
                O next;
                memcpy(&next, (uint8_t*)base + (i + 1) * sizeof(O), sizeof(O));
                O v = last + next;
                memcpy((uint8_t*)base + i * sizeof(O), &v, sizeof(O));
            }
            offs.resize(offs.size() - sizeof(O));
        }
        Hdr9 h2 = read_hdr9(r);
        r.need(h2.csize, "compressed values block");
        size_t vold = values.size();
        values.resize(vold + h2.usize);
        common_decompress(h.codec, r.p, h2.csize, values.data() + vold, h2.usize);  // codec of block 1 (mod.rs:168)
        r.p += h2.csize;
        return;
    }
    Reader in(r.p, r.left());
    switch (h.codec) {
        case C_ONEVALUE: {  // binary/one_value.rs:70-97
            size_t len = in.u32("onevalue len");
            if (in.left() < len) out_of_spec("data size is less than " + std::to_string(len));
            const uint8_t* v = in.take(len);
            if (offs.empty()) push_off<O>(offs, 0);
            for (size_t i = 0; i < length; i++) {
                put_bytes(values, v, len);
                push_off<O>(offs, values.size());
            }
            break;
        }
        case C_DICT: {  // binary/dict.rs:95-140
            std::vector<uint8_t> ibytes;
            decompress_prim<uint32_t>(in, length, ibytes);
            std::vector<size_t> doff{0};
            std::vector<uint8_t> data;
            size_t nsets = in.u32("dict size");
            for (size_t k = 0; k < nsets; k++) {
                size_t len = (size_t)in.u64("dict entry len");
                if (in.left() < len) out_of_spec("data size is less than " + std::to_string(len));
                put_bytes(data, in.take(len), len);
                doff.push_back(data.size());
            }
            uint64_t last;
            if (offs.empty()) {
                push_off<O>(offs, 0);
                last = 0;
            } else {
                last = (uint64_t)last_off<O>(offs);
            }
            size_t ni = ibytes.size() / 4;
            for (size_t i = 0; i < ni; i++) {
                uint32_t idx;
                memcpy(&idx, ibytes.data() + 4 * i, 4);
                if ((size_t)idx + 1 >= doff.size()) out_of_spec("dict index out of bounds");
                size_t b = doff[idx], e = doff[idx + 1];
                put_bytes(values, data.data() + b, e - b);
                last += e - b;
                push_off<O>(offs, last);
            }
            break;
        }
        case C_FREQ: {  // binary/freq.rs:103-145
            size_t len = (size_t)in.u64("freq top len");
            if (in.left() < len) out_of_spec("data size is less than " + std::to_string(len));
            const uint8_t* top = in.take(len);
            uint32_t rbsize = in.u32("freq bitmap size");
            std::vector<uint32_t> idx = roaring_deserialize(in.take(rbsize, "freq bitmap"), rbsize);
            if (offs.empty()) push_off<O>(offs, 0);
            size_t e = 0;
            for (size_t i = 0; i < length; i++) {
                if (e < idx.size() && idx[e] == i) {
                    e++;
                    size_t l = (size_t)in.u64("freq exception len");
                    if (in.left() < l) out_of_spec("data size is less than " + std::to_string(l));
                    put_bytes(values, in.take(l), l);
                } else {
                    put_bytes(values, top, len);
                }
                push_off<O>(offs, values.size());
            }
            break;
        }
        default:
            out_of_spec("Unknown compression codec " + std::to_string(h.codec) + " for binary");
    }
    r.p += h.csize;
}

// ============================================================ boolean
struct BoolArr {
    Bits values;
    size_t n;
    Validity val;
};
struct BoolStats {  // boolean/mod.rs:140-192
    size_t rows = 0, total_bytes = 0, null_count = 0, false_count = 0, true_count = 0;
    double average_run_length = 0;
};
static void gen_bool_stats(const BoolArr& a, BoolStats& s) {
    bool init = false, last = false;
    size_t run_count = 0;
    for (size_t i = 0; i < a.n; i++) {
        bool valid = a.val.get(i), v = a.values.get(i);
        if (!init) {
            init = true;
            last = valid ? v : false;
        }
        if (valid) {
            if (v)
                s.true_count++;
            else
                s.false_count++;
            if (last != v) {
                run_count++;
                last = v;
            }
        } else {
            s.null_count++;
        }
    }
    s.rows = a.n;
    s.total_bytes = a.n / 8;
    s.average_run_length = (double)a.n / 8.0 / (double)run_count;
}
static size_t bool_rle_compress(const BoolArr& a, std::vector<uint8_t>& out) {  // boolean/rle.rs:31-39
    size_t start = out.size();
    uint32_t seen = 0;
    uint8_t last = 0;
    bool all_null = true;
    for (size_t i = 0; i < a.n; i++) {
        uint8_t item = a.values.get(i) ? 1 : 0;
        if (a.val.get(i)) {
            if (all_null) {
                all_null = false;
                last = item;
                seen += 1;
            } else if (last != item) {
                put_u32(out, seen);
                put_u8(out, last);
                last = item;
                seen = 1;
            } else {
                seen += 1;
            }
        } else {
            seen += 1;
        }
    }
    if (seen != 0) {
        put_u32(out, seen);
        put_u8(out, last);
    }
    return out.size() - start;
}
static double bool_sample_ratio(const BoolArr& a, const BoolStats& s, const WriteOptions& opts) {  // boolean/mod.rs:240-278
    std::vector<uint8_t> tmp;
    if (a.n / SAMPLE_COUNT <= SAMPLE_SIZE) {
        size_t size = bool_rle_compress(a, tmp);
        return (double)s.total_bytes / (double)size;
    }
    size_t separator = a.n / SAMPLE_COUNT, remainder = a.n % SAMPLE_COUNT;
    std::vector<uint8_t> vb((SAMPLE_COUNT * SAMPLE_SIZE + 7) / 8, 0), nb(vb.size(), 0);
    size_t k = 0;
    for (size_t si = 0; si < SAMPLE_COUNT; si++) {
        size_t range_end = (si == SAMPLE_COUNT - 1 ? separator + remainder : separator) - SAMPLE_SIZE;
        size_t begin = si * separator + (size_t)sample_rand(opts.rng_seed, 0, C_RLE, (uint32_t)si, range_end);
        for (size_t j = 0; j < SAMPLE_SIZE; j++, k++) {
            bool valid = a.val.get(begin + j);
            if (valid && a.values.get(begin + j)) vb[k >> 3] |= (uint8_t)(1u << (k & 7));
            if (valid) nb[k >> 3] |= (uint8_t)(1u << (k & 7));
        }
    }
    BoolArr sa{Bits{vb.data(), 0}, SAMPLE_COUNT * SAMPLE_SIZE, Validity{a.val.present, Bits{nb.data(), 0}}};
    size_t size = bool_rle_compress(sa, tmp);
    return (double)((SAMPLE_COUNT * SAMPLE_SIZE) / 8) / (double)size;
}
static void compress_boolean(const BoolArr& a, const WriteOptions& opts, std::vector<uint8_t>& out) {  // boolean/mod.rs:23-61
    BoolStats s;
    gen_bool_stats(a, s);
    uint8_t codec = opts.default_compression;
    if (opts.force_codec >= 0 && !opts.forbidden((uint8_t)opts.force_codec)) {
        codec = (uint8_t)opts.force_codec;
    } else if (opts.has_ratio) {  // boolean/mod.rs:194-238
        double max_ratio = opts.ratio;
        static const uint8_t ORDER[] = {C_ONEVALUE, C_RLE};
        for (uint8_t c : ORDER) {
            if (opts.forbidden(c)) continue;
            double r = c == C_ONEVALUE ? ((s.true_count == 0 || s.false_count == 0) ? (double)s.rows : 0.0)
                                       : bool_sample_ratio(a, s, opts);
            if (r > max_ratio) {
                max_ratio = r;
                codec = c;
                if (r == (double)s.rows) break;
            }
        }
    }
    size_t pos = begin_hdr9(out, codec);
    size_t cs;
    if (is_basic(codec)) {  // boolean/mod.rs:44-54
        size_t nbytes = (a.n + 7) / 8;
        if ((a.values.off & 7) != 0) {
            std::vector<uint8_t> packed(nbytes, 0);
            for (size_t i = 0; i < a.n; i++)
                if (a.values.get(i)) packed[i >> 3] |= (uint8_t)(1u << (i & 7));
            cs = common_compress(codec, packed.data(), nbytes, out);
        } else {  // raw bytes of the shared buffer, trailing bits included
            cs = common_compress(codec, a.values.p + (a.values.off >> 3), nbytes, out);
        }
    } else if (codec == C_RLE) {
        cs = bool_rle_compress(a, out);
    } else if (codec == C_ONEVALUE) {  // boolean/one_value.rs:44-52
        uint8_t v = 0;
        for (size_t i = 0; i < a.n; i++)
            if (a.val.get(i)) {
                v = a.values.get(i);
                break;
            }
        put_u8(out, v);
        cs = 1;
    } else {
        out_of_spec("Unknown compression codec " + std::to_string(codec) + " for boolean");
    }
    end_hdr9(out, pos, cs, a.n);  // uncompressed_size = row count (boolean/mod.rs:59)
}
static void decompress_boolean(Reader& r, size_t length, BitBuilder& out) {  // boolean/mod.rs:63-102
    Hdr9 h = read_hdr9(r);
    check_codec(h.codec);
    r.need(h.csize, "compressed block");
    if (is_basic(h.codec)) {
        size_t bytes = (length + 7) / 8;
        std::vector<uint8_t> buf(bytes, 0);
        common_decompress(h.codec, r.p, h.csize, buf.data(), bytes);
        for (size_t i = 0; i < length; i++) out.push((buf[i >> 3] >> (i & 7)) & 1);
    } else if (h.codec == C_RLE) {  // boolean/rle.rs:41-55
        Reader in(r.p, r.left());
        size_t num = 0;
        while (in.left() != 0) {
            uint32_t len = in.u32("bool rle count");
            bool t = in.u8("bool rle value") != 0;
            for (uint32_t i = 0; i < len; i++) out.push(t);
            num += len;
            if (num >= length) break;
        }
    } else if (h.codec == C_ONEVALUE) {  // boolean/one_value.rs:54-61
        if (r.left() == 0) out_of_spec("data size is less than 1");
        out.extend_constant(length, r.p[0] > 0);
    } else {
        out_of_spec("Unknown compression codec " + std::to_string(h.codec) + " for boolean");
    }
    r.p += h.csize;
}

// ============================================================ def levels (validity section)
static void put_uleb(std::vector<uint8_t>& o, uint64_t v) {
    while (v >= 0x80) {
        o.push_back((uint8_t)(v | 0x80));
        v >>= 7;
    }
    o.push_back((uint8_t)v);
}
// write_validity: write/serialize.rs:200-215 -> arrow2 write_def_levels(V2) -> parquet2 encode_bool [3P]
static void write_validity(const Validity& val, size_t n, std::vector<uint8_t>& out) {
    std::vector<uint8_t> lv;
    put_uleb(lv, (uint64_t)(((n + 7) / 8) << 1) | 1);
    size_t base = lv.size();
    lv.resize(base + (n + 7) / 8, 0);
    for (size_t i = 0; i < n; i++)
        if (val.get(i)) lv[base + (i >> 3)] |= (uint8_t)(1u << (i & 7));
    put_u32(out, (uint32_t)lv.size());
    put_bytes(out, lv.data(), lv.size());
}
// read_validity: read/read_basic.rs:36-63 (parquet2 hybrid_rle::Decoder with num_bits = 1 [3P])
static void read_validity(Reader& r, size_t length, BitBuilder& b) {
    uint32_t def_len = r.u32("def_levels_len");
    if (def_len == 0) return;
    Reader d(r.take(def_len, "def levels"), def_len);
    while (d.left() != 0) {
        uint64_t indicator = 0;
        unsigned shift = 0;
        for (;;) {
            uint8_t byte = d.u8("uleb");
            indicator |= (uint64_t)(byte & 0x7F) << shift;
            shift += 7;
            if (!(byte & 0x80)) break;
        }
        if (d.left() == 0) break;
        if (indicator & 1) {
            size_t bytes = std::min((size_t)(indicator >> 1), d.left());
            const uint8_t* run = d.take(bytes);
            if (bytes * 8 < length) out_of_spec("def levels: bit-packed run shorter than the page");
            for (size_t i = 0; i < length; i++) b.push((run[i >> 3] >> (i & 7)) & 1);
        } else {
            out_of_spec("def levels: RLE run (unreachable!() upstream, read_basic.rs:59)");
        }
    }
}

// ============================================================ page / column drivers
size_t type_width(int32_t t) {
    switch (t) {
        case T_I8:
        case T_U8:
            return 1;
        case T_I16:
        case T_U16:
            return 2;
        case T_I32:
        case T_U32:
        case T_F32:
            return 4;
        case T_I64:
        case T_U64:
        case T_F64:
            return 8;
        case T_I128:
            return 16;
        case T_I256:
            return 32;
    }
    return 0;
}

template <class T>
static void write_prim_page(const ColumnIn& c, const WriteOptions& opts, std::vector<uint8_t>& out) {
    Validity val{c.validity != nullptr, Bits{c.validity, c.validity_bit_offset}};
    PrimArr<T> a{(const T*)c.values, (size_t)c.rows, val};
    compress_prim<T>(a, opts, 0, out);
}

#define SBO_DISPATCH_PRIM(ptype, CALL)                      \
    switch (ptype) {                                        \
        case T_I8: { typedef int8_t T; CALL; } break;       \
        case T_I16: { typedef int16_t T; CALL; } break;     \
        case T_I32: { typedef int32_t T; CALL; } break;     \
        case T_I64: { typedef int64_t T; CALL; } break;     \
        case T_U8: { typedef uint8_t T; CALL; } break;      \
        case T_U16: { typedef uint16_t T; CALL; } break;    \
        case T_U32: { typedef uint32_t T; CALL; } break;    \
        case T_U64: { typedef uint64_t T; CALL; } break;    \
        case T_I128: { typedef I128 T; CALL; } break;       \
        case T_I256: { typedef I256 T; CALL; } break;       \
        case T_F32: { typedef float T; CALL; } break;       \
        case T_F64: { typedef double T; CALL; } break;      \
        default: out_of_spec("not a primitive type");       \
    }

// write::write_simple (write/serialize.rs:52-132)
void write_page(const ColumnIn& c, const WriteOptions& opts, std::vector<uint8_t>& out) {
    if (c.ptype == T_NULL) return;  // serialize.rs:63
    Validity val{c.validity != nullptr, Bits{c.validity, c.validity_bit_offset}};
    if (c.nullable) write_validity(val, (size_t)c.rows, out);
    if (c.ptype == T_BOOL) {
        BoolArr a{Bits{c.values, c.values_bit_offset}, (size_t)c.rows, val};
        compress_boolean(a, opts, out);
    } else if (c.ptype == T_BIN32) {
        BinArr<int32_t> a{(const int32_t*)c.offsets, (size_t)c.rows, c.values, c.values_len, val};
        compress_binary<int32_t>(a, opts, out);
    } else if (c.ptype == T_BIN64) {
        BinArr<int64_t> a{(const int64_t*)c.offsets, (size_t)c.rows, c.values, c.values_len, val};
        compress_binary<int64_t>(a, opts, out);
    } else {
        SBO_DISPATCH_PRIM(c.ptype, write_prim_page<T>(c, opts, out));
    }
}

// NativeWriter::encode_chunk page loop (write/common.rs:54-109) for one flat leaf.
void write_column(const ColumnIn& col, const WriteOptions& opts, std::vector<uint8_t>& out,
                  std::vector<PageMeta>& metas) {
    if (col.rows == 0) throw Error(-1, "encode_chunk on an empty chunk panics upstream (step_by(0))");
    uint64_t page_size = opts.max_page_size ? std::min<uint64_t>(opts.max_page_size, col.rows) : col.rows;
    size_t w = type_width(col.ptype);
    uint64_t page_index = 0;
    for (uint64_t offset = 0; offset < col.rows; offset += page_size, page_index++) {
        uint64_t length = offset + page_size > col.rows ? col.rows - offset : page_size;
        ColumnIn page = col;
        page.rows = length;
        if (col.validity) page.validity_bit_offset = col.validity_bit_offset + offset;
        if (col.ptype == T_BOOL)
            page.values_bit_offset = col.values_bit_offset + offset;
        else if (col.ptype == T_BIN32)
            page.offsets = col.offsets + offset * 4;
        else if (col.ptype == T_BIN64)
            page.offsets = col.offsets + offset * 8;
        else
            page.values = col.values + offset * w;
        WriteOptions popts = opts;
        popts.rng_seed = mix64(opts.rng_seed ^ ((opts.page_index0 + page_index) * 0xD6E8FEB86659FD93ull));
        size_t start = out.size();
        write_page(page, popts, out);
        metas.push_back(PageMeta{(uint64_t)(out.size() - start), length});
    }
}

// read_integer / read_double (read/array/integer.rs:210-238), read_boolean
// (read/array/boolean.rs:191-219), read_binary (read/array/binary.rs:223-265)
void read_column(int32_t ptype, bool nullable, const uint8_t* pages, uint64_t pages_len, const PageMeta* metas,
                 uint64_t n_pages, ColumnOut& out) {
    Reader r(pages, (size_t)pages_len);
    BitBuilder vb, bb;
    out.rows = 0;
    for (uint64_t p = 0; p < n_pages; p++) {
        size_t length = (size_t)metas[p].num_values;
        const uint8_t* page_start = r.p;
        out.rows += length;
        if (ptype == T_NULL) continue;  // read/array/null.rs:48-52
        if (nullable) read_validity(r, length, vb);
        if (ptype == T_BOOL) {
            decompress_boolean(r, length, bb);
        } else if (ptype == T_BIN32) {
            decompress_binary<int32_t>(r, length, out.offsets, out.values);
        } else if (ptype == T_BIN64) {
            decompress_binary<int64_t>(r, length, out.offsets, out.values);
        } else {
            SBO_DISPATCH_PRIM(ptype, decompress_prim<T>(r, length, out.values));
        }
        // The reference never compares what a page's decoders consumed with PageMeta.length: the iterator API hands every page
        // its own buffer and drops what is left of it (src/read/array/integer.rs:69-81: `reader.into_inner()` goes back to the
        // page iterator), the batch API reads page after page from one reader (src/read/array/integer.rs:210-238), so a page
        // that consumes LESS than its length makes the next page start early — garbage or an error.  Restated as: consuming more
        // than the page is refused, less only on the column's last page (nothing behind it depends on where it ended).
        const uint64_t used = (uint64_t)(r.p - page_start);
        if (used > metas[p].length || (used < metas[p].length && p + 1 < n_pages))
            out_of_spec("page " + std::to_string(p) + ": consumed " + std::to_string(used) +
                        " bytes, PageMeta.length = " + std::to_string(metas[p].length));
        r.p = page_start + metas[p].length;
    }
    if (ptype == T_BOOL) {
        out.values = std::move(bb.bytes);
        if (bb.len != out.rows) out_of_spec("boolean values length mismatch");
    } else if (ptype == T_BIN32 || ptype == T_BIN64) {
        size_t ow = ptype == T_BIN32 ? 4 : 8;
        if (out.offsets.empty()) out.offsets.assign(ow, 0);  // empty column: single 0 offset
        if (out.offsets.size() / ow != out.rows + 1) out_of_spec("offsets length mismatch");
    } else if (ptype != T_NULL) {
        size_t w = type_width(ptype);
        // Bitpacking emits whole 128-blocks (bp.rs:72-84); PrimitiveArray::try_new would then fail
        if (out.values.size() != out.rows * w) out_of_spec("values length mismatch");
    }
    if (nullable && ptype != T_NULL) {
        if (vb.len != out.rows) out_of_spec("validity mask length must match the number of values");
        out.validity = std::move(vb.bytes);
        out.validity_bits = vb.len;
    }
}

// a minimal `stat_simple` (src/stat.rs:63-152): top-level codec id per page and, for
// Dict/Freq pages of primitives and Dict pages of binary, the nested block's codec id
void stat_column(int32_t ptype, bool nullable, const uint8_t* pages, uint64_t pages_len, const PageMeta* metas,
                 uint64_t n_pages, std::vector<uint8_t>& codecs, std::vector<uint8_t>& inner) {
    size_t off = 0;
    for (uint64_t p = 0; p < n_pages; p++) {
        if (off + metas[p].length > pages_len) io_eof("stat page");
        Reader r(pages + off, (size_t)metas[p].length);
        off += (size_t)metas[p].length;
        if (ptype == T_NULL) {
            codecs.push_back(255);
            inner.push_back(255);
            continue;
        }
        if (nullable) {
            uint32_t dl = r.u32();
            r.take(dl);
        }
        Hdr9 h = read_hdr9(r);
        codecs.push_back(h.codec);
        uint8_t in = 255;
        if (h.codec == C_DICT) {
            in = r.u8();
        } else if (h.codec == C_FREQ && ptype != T_BIN32 && ptype != T_BIN64 && ptype != T_BOOL) {
            r.take(type_width(ptype));
            uint32_t rb = r.u32();
            r.take(rb);
            in = r.u8();
        }
        inner.push_back(in);
    }
}

}  // namespace sbo
