// ORACLE — TEST INFRASTRUCTURE ONLY (see sbo.h).  Small shared helpers.
#pragma once
#include <cmath>
#include <cstring>
#include <functional>
#include <unordered_map>

#include "sbo.h"

namespace sbo {

// ---------------------------------------------------------------- byte cursor
struct Reader {
    const uint8_t* p;
    const uint8_t* end;
    Reader(const uint8_t* b, size_t n) : p(b), end(b + n) {}
    size_t left() const { return (size_t)(end - p); }
    void need(size_t n, const char* what) const {
        if (left() < n) io_eof(what);
    }
    uint8_t u8(const char* w = "u8") {
        need(1, w);
        return *p++;
    }
    uint16_t u16(const char* w = "u16") {
        need(2, w);
        uint16_t v;
        memcpy(&v, p, 2);
        p += 2;
        return v;
    }
    uint32_t u32(const char* w = "u32") {
        need(4, w);
        uint32_t v;
        memcpy(&v, p, 4);
        p += 4;
        return v;
    }
    uint64_t u64(const char* w = "u64") {
        need(8, w);
        uint64_t v;
        memcpy(&v, p, 8);
        p += 8;
        return v;
    }
    const uint8_t* take(size_t n, const char* w = "bytes") {
        need(n, w);
        const uint8_t* r = p;
        p += n;
        return r;
    }
};

inline void put_u8(std::vector<uint8_t>& o, uint8_t v) { o.push_back(v); }
inline void put_u16(std::vector<uint8_t>& o, uint16_t v) {
    o.push_back((uint8_t)v);
    o.push_back((uint8_t)(v >> 8));
}
inline void put_u32(std::vector<uint8_t>& o, uint32_t v) {
    for (int i = 0; i < 4; i++) o.push_back((uint8_t)(v >> (8 * i)));
}
inline void put_u64(std::vector<uint8_t>& o, uint64_t v) {
    for (int i = 0; i < 8; i++) o.push_back((uint8_t)(v >> (8 * i)));
}
inline void put_bytes(std::vector<uint8_t>& o, const void* p, size_t n) {
    const uint8_t* b = (const uint8_t*)p;
    o.insert(o.end(), b, b + n);
}
inline void patch_u32(std::vector<uint8_t>& o, size_t pos, uint32_t v) {
    for (int i = 0; i < 4; i++) o[pos + i] = (uint8_t)(v >> (8 * i));
}

// ---------------------------------------------------------------- bitmaps (Arrow LSB-first)
struct Bits {
    const uint8_t* p = nullptr;
    uint64_t off = 0;
    bool get(uint64_t i) const {
        uint64_t k = off + i;
        return (p[k >> 3] >> (k & 7)) & 1;
    }
    Bits slice(uint64_t start) const { return Bits{p, off + start}; }
};

// MutableBitmap restatement (append-only)
struct BitBuilder {
    std::vector<uint8_t> bytes;
    uint64_t len = 0;
    void push(bool v) {
        if ((len & 7) == 0) bytes.push_back(0);
        if (v) bytes.back() |= (uint8_t)(1u << (len & 7));
        len++;
    }
    void extend_constant(uint64_t n, bool v) {
        for (uint64_t i = 0; i < n; i++) push(v);
    }
};

// src/compression/mod.rs:111-116 is_valid
struct Validity {
    bool present = false;
    Bits bits;
    bool get(uint64_t i) const { return !present || bits.get(i); }
    Validity slice(uint64_t start) const { return Validity{present, bits.slice(start)}; }
    uint64_t null_count(uint64_t n) const {
        if (!present) return 0;
        uint64_t c = 0;
        for (uint64_t i = 0; i < n; i++) c += !bits.get(i);
        return c;
    }
};

// src/compression/mod.rs:119-121 get_bits_needed
inline uint32_t get_bits_needed(uint64_t v) { return v == 0 ? 0 : 64 - (uint32_t)__builtin_clzll(v); }

// ---------------------------------------------------------------- i256 (arrow2 types::i256)
struct I256 {
    uint64_t w[4];  // little-endian limbs
    bool operator==(const I256& o) const { return memcmp(w, o.w, 32) == 0; }
    bool operator!=(const I256& o) const { return !(*this == o); }
    bool operator<(const I256& o) const {
        if ((int64_t)w[3] != (int64_t)o.w[3]) return (int64_t)w[3] < (int64_t)o.w[3];
        for (int i = 2; i >= 0; i--)
            if (w[i] != o.w[i]) return w[i] < o.w[i];
        return false;
    }
    bool operator>(const I256& o) const { return o < *this; }
};
typedef __int128 I128;

// ---------------------------------------------------------------- per-type ops
// integer family: native ==, <  (src/compression/integer/traits.rs:5-39)
// float family: OrderedFloat semantics for stats/RLE/Freq (src/compression/double/traits.rs:51):
//   NaN == NaN, NaN greatest, -0.0 == +0.0
template <class T>
struct Ops {
    static constexpr bool is_float = false;
    static bool eq(T a, T b) { return a == b; }
    static bool lt(T a, T b) { return a < b; }
    static int64_t as_i64(T a) { return (int64_t)a; }
    typedef uint64_t Key;
    static Key key(T a) {
        uint64_t k = 0;
        memcpy(&k, &a, sizeof(T));
        return k;
    }
};
struct WideKey {
    uint64_t w[4];
    bool operator==(const WideKey& o) const { return memcmp(w, o.w, 32) == 0; }
};
struct WideKeyHash {
    size_t operator()(const WideKey& k) const {
        return (size_t)mix64(k.w[0] ^ mix64(k.w[1] ^ mix64(k.w[2] ^ mix64(k.w[3]))));
    }
};
template <>
struct Ops<I128> {
    static constexpr bool is_float = false;
    static bool eq(I128 a, I128 b) { return a == b; }
    static bool lt(I128 a, I128 b) { return a < b; }
    static int64_t as_i64(I128 a) { return (int64_t)a; }
    typedef WideKey Key;
    static Key key(I128 a) {
        WideKey k{{0, 0, 0, 0}};
        memcpy(k.w, &a, 16);
        return k;
    }
};
template <>
struct Ops<I256> {
    static constexpr bool is_float = false;
    static bool eq(I256 a, I256 b) { return a == b; }
    static bool lt(I256 a, I256 b) { return a < b; }
    static int64_t as_i64(I256 a) { return (int64_t)a.w[0]; }
    typedef WideKey Key;
    static Key key(I256 a) {
        WideKey k;
        memcpy(k.w, a.w, 32);
        return k;
    }
};
template <class F, class B>
struct FloatOps {
    static constexpr bool is_float = true;
    static bool eq(F a, F b) { return (std::isnan(a) && std::isnan(b)) || a == b; }
    static bool lt(F a, F b) {  // OrderedFloat: NaN is the greatest value
        if (std::isnan(a)) return false;
        if (std::isnan(b)) return true;
        return a < b;
    }
    static int64_t as_i64(F) { return 0; }
    typedef uint64_t Key;
    static Key key(F a) {  // canonical bits (OrderedFloat's Hash canonicalises NaN and zero)
        if (std::isnan(a)) return ~0ull;
        if (a == 0) return 0;
        B b;
        memcpy(&b, &a, sizeof(F));
        return (uint64_t)b;
    }
};
template <>
struct Ops<float> : FloatOps<float, uint32_t> {};
template <>
struct Ops<double> : FloatOps<double, uint64_t> {};

template <class K>
struct KeyHash {
    size_t operator()(const K& k) const { return std::hash<K>()(k); }
};
template <>
struct KeyHash<WideKey> : WideKeyHash {};

template <class T>
inline T zero_value() {
    T z;
    memset(&z, 0, sizeof(T));
    return z;
}

}  // namespace sbo
