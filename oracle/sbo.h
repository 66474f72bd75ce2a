// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// CPU restatement of sundy-li/strawboat's per-page encode/decode + adaptive
// compression path (reference: /root/reference/src/compression/**,
// src/write/{serialize,primitive,binary,boolean,common}.rs, src/read/read_basic.rs,
// src/read/array/{integer,double,boolean,binary}.rs).  Written from scratch; every
// function cites the reference file:line it follows.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this
// library, and only as the checker / CPU baseline.  The product (strawboat_amd/) never
// links, imports or calls it.
//
// Parity pinning: the Rust reference cannot be built in this image (no rustc/cargo,
// needs nightly-2023-03-10 + un-vendored crates).  This restatement is pinned against
//   * the reference's own known-answer tests for this path (patas pack/unpack,
//     src/compression/double/patas.rs:191-202; stat.rs:228-269 codec choice),
//   * the hand-derived byte vectors of SURVEY.md Appendix C (tests/golden/),
//   * liblz4 1.9.3 / libzstd / pyarrow as independent third-party cross-checks for the
//     LZ4 / Zstd / Snappy block formats (fixtures generated in-container, committed),
//   * round-trip properties on the shapes of tests/it/io.rs.
// Byte parity with the third-party crates the reference calls (bitpacking 0.8
// BitPacker4x, roaring 0.10 portable format, parquet2 0.17 hybrid-RLE) follows their
// published formats; those crates are not under /root/reference.
#pragma once
#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

namespace sbo {

// src/compression/mod.rs:37-51, :92-108 (on-disk codec ids)
enum Codec : uint8_t {
    C_NONE = 0,
    C_LZ4 = 1,
    C_ZSTD = 2,
    C_SNAPPY = 3,
    C_RLE = 10,
    C_DICT = 11,
    C_ONEVALUE = 12,
    C_FREQ = 13,
    C_BITPACK = 14,
    C_DELTABP = 15,
    C_PATAS = 16,
};

// physical kinds: the dispatch key of src/read/batch_read.rs:37-63
enum PhysType : int32_t {
    T_BOOL = 0,
    T_I8 = 1,
    T_I16 = 2,
    T_I32 = 3,
    T_I64 = 4,
    T_U8 = 5,
    T_U16 = 6,
    T_U32 = 7,
    T_U64 = 8,
    T_I128 = 9,
    T_I256 = 10,
    T_F32 = 11,
    T_F64 = 12,
    T_BIN32 = 13,  // Binary / Utf8 (i32 offsets)
    T_BIN64 = 14,  // LargeBinary / LargeUtf8 (i64 offsets)
    T_NULL = 15,
};

struct Error : std::runtime_error {
    int code;  // -1 OutOfSpec, -2 External, -3 Io, -4 NotYetImplemented
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};
[[noreturn]] inline void out_of_spec(const std::string& m) { throw Error(-1, m); }
[[noreturn]] inline void io_eof(const std::string& m) { throw Error(-3, "unexpected eof: " + m); }

// src/write/common.rs:37-45 WriteOptions, plus the two knobs the reference lacks but a
// deterministic build needs: force_codec (like the debug-only env
// switches, src/util/env.rs:20-24) and an injectable sampling seed.
struct WriteOptions {
    uint8_t default_compression = C_NONE;  // CommonCompression
    bool has_ratio = false;                // default_compress_ratio: Option<f64>
    double ratio = 0.0;
    uint64_t max_page_size = 0;            // 0 = None
    uint32_t forbidden_mask = 0;           // bit (codec id) set => forbidden
    int32_t force_codec = -1;              // top-level page codec, -1 = choose
    int32_t force_index_codec = -1;        // nested (Dict indices / Freq exceptions) codec
    uint64_t rng_seed = 42;                // page-level sampling seed
    uint64_t page_index0 = 0;              // index of the call's first page inside its column (page-range work items)
    bool forbidden(uint8_t c) const { return (forbidden_mask >> c) & 1u; }
};

// deterministic replacement for thread_rng() in compress_sample_ratio
// (src/compression/integer/mod.rs:316,332).  splitmix64 finaliser; shared verbatim
// with the HIP selector so codec choice is reproducible on both sides.
inline uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// gen_range(0..n) for sample `sample_i` of the trial of `codec` at nesting `depth`
inline uint64_t sample_rand(uint64_t page_seed, uint32_t depth, uint32_t codec, uint32_t sample_i,
                            uint64_t n) {
    uint64_t r = mix64(page_seed ^ mix64(((uint64_t)depth << 40) | ((uint64_t)codec << 32) | sample_i));
    return (uint64_t)(((unsigned __int128)r * n) >> 64);
}

// ---- LZ4 block format (sbo_lz4.cpp) — third-party algorithm: liblz4 via the `lz4`
// crate (Cargo.toml:23), call sites src/compression/basic.rs:87-91,108-120.
// the box's liblz4 / libzstd for Basic(LZ4 / Zstd) blocks (CPU baseline leg; sbo_codecs.cpp)
int system_codecs_enable(int on);
int system_codec_version(int which);
size_t lz4_compress_bound(size_t n);
// restatement of LZ4_compress_default (greedy single-probe hash parse, liblz4 1.9.x)
size_t lz4_compress(const uint8_t* src, size_t n, uint8_t* dst, size_t cap);
// LZ4_decompress_safe into exactly out_len bytes; throws on malformed input
void lz4_decompress(const uint8_t* src, size_t n, uint8_t* dst, size_t out_len);

// ---- Zstd frame format (sbo_zstd.cpp) — third-party: libzstd via `zstd` crate
// (Cargo.toml:24), call sites src/compression/basic.rs:93-97,122-135.
void zstd_decompress(const uint8_t* src, size_t n, uint8_t* dst, size_t out_len);
// valid single-frame encoder (raw/RLE blocks; byte parity with libzstd level 3 is
// library-version dependent upstream and not claimed)
size_t zstd_compress_bound(size_t n);
size_t zstd_compress(const uint8_t* src, size_t n, uint8_t* dst, size_t cap);

// ---- Snappy raw (sbo_snappy.cpp) — `snap` crate (Cargo.toml:25), basic.rs:99-106,137-152
void snappy_decompress(const uint8_t* src, size_t n, uint8_t* dst, size_t out_len);
size_t snappy_compress_bound(size_t n);
size_t snappy_compress(const uint8_t* src, size_t n, uint8_t* dst, size_t cap);

// ---- column level API (sbo_codecs.cpp) ------------------------------------------

struct PageMeta {  // src/lib.rs:75-80
    uint64_t length;
    uint64_t num_values;
};

struct ColumnIn {  // one flat leaf column (Arrow buffers, host)
    int32_t ptype = T_I64;
    bool nullable = false;            // schema field nullable (serialize.rs:234-240)
    uint64_t rows = 0;
    const uint8_t* values = nullptr;  // primitive values / boolean bitmap / binary values
    uint64_t values_bit_offset = 0;   // boolean only
    uint64_t values_len = 0;          // binary only: byte length of the WHOLE values buffer
                                      // (array.values().len(); slices share it, binary/mod.rs:88,268)
    const uint8_t* validity = nullptr;  // may be null even if nullable
    uint64_t validity_bit_offset = 0;
    const uint8_t* offsets = nullptr;  // binary: (rows+1) offsets of i32/i64
};

struct ColumnOut {  // what read_integer/read_binary/read_boolean materialise
    uint64_t rows = 0;
    std::vector<uint8_t> values;    // primitive bytes / boolean bitmap / binary values
    std::vector<uint8_t> validity;  // bitmap bytes (empty if not nullable)
    uint64_t validity_bits = 0;
    std::vector<uint8_t> offsets;   // binary offsets bytes
};

// NativeWriter::encode_chunk page loop for one flat leaf (src/write/common.rs:54-109)
void write_column(const ColumnIn& col, const WriteOptions& opts, std::vector<uint8_t>& out,
                  std::vector<PageMeta>& metas);
// write::write for one already-sliced page (src/write/serialize.rs:36-132)
void write_page(const ColumnIn& page, const WriteOptions& opts, std::vector<uint8_t>& out);
size_t type_width(int32_t ptype);
// batch_read::read_simple (src/read/batch_read.rs:27-64) → read_integer & co
void read_column(int32_t ptype, bool nullable, const uint8_t* pages, uint64_t pages_len,
                 const PageMeta* metas, uint64_t n_pages, ColumnOut& out);

// helpers exposed for unit tests
uint32_t patas_pack(uint32_t ref_diff, uint32_t sig_bytes, uint32_t trailing_zeros);  // patas.rs:145-149
void bitpack4x_pack(const uint32_t* in128, uint8_t num_bits, uint8_t* out, bool delta, uint32_t initial);
void bitpack4x_unpack(const uint8_t* in, uint8_t num_bits, uint32_t* out128, bool delta, uint32_t initial);
uint8_t bitpack4x_num_bits(const uint32_t* in128);
// codec ids of each top-level page block of a column (a small `stat.rs`, src/stat.rs:63-152)
void stat_column(int32_t ptype, bool nullable, const uint8_t* pages, uint64_t pages_len,
                 const PageMeta* metas, uint64_t n_pages, std::vector<uint8_t>& codecs,
                 std::vector<uint8_t>& inner_codecs);

}  // namespace sbo
