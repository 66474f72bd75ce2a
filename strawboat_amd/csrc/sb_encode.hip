// strawboat-hip: page encode kernels for gfx950 (MI355X) and the sb_write_columns entry point.
//
// Replaces, on the device, the per-page work of the reference's write path:
//   write_validity           src/write/serialize.rs:200-215 (def levels, parquet V2 bit-packed run)
//   compress_integer/double  src/compression/integer/mod.rs:35-70, double/mod.rs:32-67
//        RLE                 integer/rle.rs:64-104, double/rle.rs:61-103
//        Dict                integer/dict.rs:33-73 (+ nested compress_integer::<u32>)
//        Bitpacking          integer/bp.rs:36-64; DeltaBitpacking integer/delta_bp.rs:36-68
//        OneValue            integer/one_value.rs:63-75
//   compress_boolean         src/compression/boolean/mod.rs:23-61 (+ rle.rs, one_value.rs)
//   compress_binary          src/compression/binary/mod.rs:26-93 (+ dict.rs, one_value.rs)
// and the page loop / PageMeta bookkeeping of NativeWriter::encode_chunk
// (src/write/common.rs:54-109).
//
// Kernel sequence of one call:
//   k_enc_emit_tiles  1 workgroup / (page, tile)  pages whose codec is None: def bits + plain copy
//   k_enc_emit_pages  1 workgroup / page          RLE / Dict / bit-packing / OneValue pages,
//                                                  sequential over TILE_ROWS chunks with carries
//   k_enc_layout      1 thread / column           page lengths -> offsets in the output, PageMeta
//   k_enc_compact     1 workgroup / (page, 64 KiB) slot -> final position
// Pages whose size is known up front (None / OneValue of fixed-width types) are written straight
// to their final position ("direct"), everything else goes through a worst-case sized slot.
#include <cstring>
#include <type_traits>
#include <utility>
#include <vector>

#include "sb_host.h"
#include "sb_lz4.h"
#include "sb_zstd_enc.h"

namespace sb {

struct EncCol {
    const uint8_t* values;
    const uint8_t* validity;
    const uint8_t* offsets;
    const uint8_t* heads;  // optional per-page heads (nested level sections), back to back
    uint8_t* out;
    uint64_t values_bit_offset;
    uint64_t values_len;
    uint64_t values_len_total;  // array.values().len() of the column (== values_len unless the call holds a page range)
    uint64_t validity_bit_offset;
    uint64_t out_cap;
    uint64_t rows;
    int32_t ptype;
    int32_t nullable;
    uint32_t width;
    uint32_t first_page;
    uint32_t n_pages;
    uint32_t fkind;  // 0 = bitwise equality, 1 = f32, 2 = f64 (OrderedFloat equality for RLE)
    uint32_t nk;     // NumKind: signed / unsigned / f32 / f64 (typed order of the selector)
    uint32_t pad;
};

struct EncPage {
    uint64_t row0;
    uint64_t rows;
    uint64_t slot_off;    // byte offset of the slot in scratch (fixed part; binary adds its value share)
    uint64_t aux_off;     // Dict: hash table + F/R/idx arrays
    uint64_t aux_bytes;
    uint64_t direct_off;  // direct pages: byte offset in the column's output
    uint64_t seed;
    uint32_t col;
    int32_t codec;        // codec decided by the host (-1: decided on the device)
    int32_t icodec;       // nested codec for Dict indices
    uint32_t direct;
    uint64_t head_off;    // offset of the page's head (level section) in EncCol.heads
    uint64_t head_bytes;  // bytes reserved in front of the page's block
    // Freq: the page's exceptions become a "virtual page" (entry n_pages + page of the tables) that runs
    // through the same select / emit kernels in a second wave (integer/freq.rs:76-83)
    uint64_t ex_off;      // scratch: exception values (rows * width bytes)
    uint64_t vslot_off;   // scratch: slot of the virtual page
    uint64_t vaux_off;    // scratch: Dict aux of the virtual page
    uint64_t vaux_bytes;
    uint64_t slot_cap;    // bytes available in the page's slot (Freq pages append a block of unknown size)
    uint32_t depth;       // nesting depth of this block (sampling RNG stream; 0 = page)
    uint32_t forb_extra;  // codecs forbidden for this block in addition to the options'
    uint64_t h64_off;     // binary pages: scratch offset of one u64 hash per row (~0: none), written by bin_hash_rows
    uint64_t zst_off;     // Basic(Zstd) pages: scratch of the Zstd encoder (zstd_scratch_bytes; ~0: none)
    uint64_t bigx_off;    // long pages (>= SEL_BIG_ROWS rows) that may become Dict pages: work area of sb_dict_big.h (0: none)
};

struct EncOut {
    uint64_t length;    // bytes of the page
    uint64_t out_off;   // offset in the column's output (k_enc_layout)
    uint8_t* slot;      // where the page was emitted
    uint32_t codec;
    uint32_t pad;
};

struct EncodeArgs {
    const EncCol* cols;
    const EncPage* pages;
    EncOut* outs;
    uint8_t* scratch;
    Status* status;
    uint64_t* results;  // per column: [n_pages lengths][n_pages num_values][total]
    int32_t* codecs;    // per page: codec chosen on the device (adaptive mode)
    double ratio;       // default_compress_ratio
    uint32_t has_ratio;
    uint32_t forbidden;
    uint32_t n_pages;
    uint32_t n_cols;
    uint32_t default_compression;
    const EncCol* vcols;  // virtual columns / pages of Freq exceptions: entries n_cols.. / n_pages.. (device-written)
    const EncPage* vpages;
    uint32_t* freq_count;  // pages that chose Freq in this call (the Freq kernels return at once when 0)
    uint32_t* codec_counts;  // [32] adaptive mode: pages per chosen codec ([31]: left to the row-level selector); an emit
                             // kernel whose codec nobody chose returns before it looks at a page
    uint32_t use_counts;     // this launch may trust codec_counts (adaptive wave over real pages)
    uint32_t null_cols;      // the batch holds Null columns (their empty pages are recorded by k_enc_emit_tiles)
    uint32_t page_base;   // first table entry this launch works on (0: pages, n_pages: virtual pages)
    int32_t nested_force; // force_index_codec: the codec forced on nested blocks (-1: none)
    uint32_t flags;       // sb_write_options.flags (SB_WRITE_LZ4_EXACT)
    // LZ4 blocks longer than one chunk are compressed chunk by chunk, one wave each (k_enc_lz4_plan / _chunks / _stitch)
    struct LzChunkPlan* lzc_plan;   // per page (nullptr: every LZ4 page goes through k_enc_emit_lz4)
    struct LzChunkDesc* lzc_list;   // the chunks of this call, in the order the pages reserved them
    uint32_t* lzc_count;
    uint8_t* lzc_pool;              // one slot of LZC_SLOT bytes per chunk
    uint32_t lzc_cap;
    uint32_t lzc_chunk;             // chunk bytes of this call: LZC_CH (LZ4, Snappy) or ZPAR_CH (Zstd blocks, one wave each)
    int32_t lzc_codec;              // the Basic codec whose big blocks go chunk by chunk in this call (LZ4 / Zstd / Snappy)
    uint8_t* zpar_scratch;          // Zstd: encoder scratch of the chunk waves (ZPAR_WAVES x zstd_scratch_bytes(ZPAR_CH))
    uint32_t pre_hashed;            // k_enc_bin_hash ran before the selector: the h64 arrays of adaptive binary pages are filled
    uint32_t redo;                  // k_enc_select: second pass over the binary pages k_enc_bin_verify failed (no tags, exact count)
    uint32_t bin_fused;             // k_enc_bin_page (sb_bin_page.h) ran in front of the binary chain: the pages it decided are skipped
    uint32_t skips;                 // SKIP_* bits: kernels this call left out because the last call with the same plan did not need them; a page
                                    // that needs one after all is left unwritten, and k_enc_layout asks for the replay (KIND_REPLAY)
};
constexpr uint32_t SKIP_DICT_BIG = 1u, SKIP_FREQ_BIG = 2u, SKIP_EMIT = 4u;
constexpr uint32_t ZPAR_CH = 32768;      // a Zstd frame's blocks when they are compressed by waves of their own (measured on C5, write / read GB/s: 16 KiB 129 / 171, 32 KiB 147 / 201, 64 KiB 90 / 175)
constexpr uint32_t ZPAR_CH_SMALL = 16384; // ... in calls with fewer pieces than chunk waves
constexpr uint32_t ZPAR_WAVES = 2048;     // 8 per CU: what 20 KB of LDS per wave (and 221 VGPRs) keep resident; a larger pool runs a second, thin round
constexpr uint32_t LZC_CH = 65536;                                         // chunk bytes (a multiple of 1024)
// A block of LZC_LONG bytes and more is cut into LZC_CH_LONG-byte chunks: a one-page column of 96 MB is 1 465 chunks of 64 KiB on
// 2 816 resident chunk waves — one round whose length is a chunk's (2.3 ms) — but 5 860 of 16 KiB, which keep the chip full
// (1.6 ms).  A function of the BLOCK, not of the call: a page's bytes do not depend on what else the call writes.
constexpr uint32_t LZC_LONG = 8u << 20, LZC_CH_LONG = 16384;
__host__ __device__ __forceinline__ uint32_t lz4_chunk_bytes(uint64_t block_bytes) { return block_bytes >= LZC_LONG ? LZC_CH_LONG : 65536u; }
constexpr uint32_t LZC_SLOT = (16 + LZC_CH + LZC_CH / 255 + 16 + 15) / 16 * 16;   // u32 size | u32 tail anchor | 8 pad | sequences
struct LzChunkPlan { uint32_t base, n_a, n_b, pad; };   // chunks [base, base + n_a) = first block, then n_b of a binary page's values block
struct LzChunkDesc { uint32_t page, idx; };             // idx: chunk of its block; bit 31: the values block

__device__ __forceinline__ EncPage get_page(const EncodeArgs& a, uint32_t i) {
    return i < a.n_pages ? a.pages[i] : a.vpages[i - a.n_pages];
}
__device__ __forceinline__ EncCol get_col(const EncodeArgs& a, uint32_t i) {
    return i < a.n_cols ? a.cols[i] : a.vcols[i - a.n_cols];
}
constexpr int32_t CODEC_ON_DEVICE = -1;  // EncPage.codec: chosen by k_enc_select; < -1: table entry not in use
__device__ __forceinline__ int32_t codec_of(const EncodeArgs& a, const EncPage& p, uint32_t page) {
    return p.codec >= 0 ? p.codec : (p.codec == CODEC_ON_DEVICE ? a.codecs[page] : -2);
}
__device__ __forceinline__ bool has_device_encoder(uint32_t codec) {
    return codec == SB_CODEC_NONE || codec == SB_CODEC_LZ4 || codec == SB_CODEC_ZSTD || codec == SB_CODEC_SNAPPY || codec == SB_CODEC_RLE ||
           codec == SB_CODEC_DICT ||
           codec == SB_CODEC_ONEVALUE || codec == SB_CODEC_BITPACKING || codec == SB_CODEC_DELTA_BITPACKING || codec == SB_CODEC_PATAS ||
           codec == SB_CODEC_FREQ;  // (Freq: primitives only, checked where the codec is chosen)
}

constexpr uint32_t EMPTY = 0xFFFFFFFFu;
constexpr uint64_t DICT_FREQ_PENDING = ~0ull;  // emit_prim_page<Dict>: the page waits for the Freq kernels (EncOut.pad = 3)
constexpr uint32_t COMPACT_CHUNK = 64 * 1024;

#ifdef SB_RLE_TIMELINE  // scripts/micro/rle_timeline.hip: s_memtime stamps of one workgroup's phases
__device__ unsigned long long* g_tl;
#define TL_INIT unsigned long long* tl_ = blockIdx.x == 100 ? g_tl : nullptr; uint32_t tl_c = 0;
#define TL(p)                                                                                        \
    do {                                                                                             \
        if (tl_ && lane == 0 && tl_c < 16) tl_[(w * 16 + tl_c) * 8 + (p)] = __builtin_readcyclecounter(); \
    } while (0)
#define TL_NEXT tl_c++;
#define STL(p)                                                                                       \
    do {                                                                                             \
        if (g_tl && blockIdx.x == 100 && threadIdx.x == 0) g_tl[512 + (p)] = __builtin_readcyclecounter(); \
    } while (0)
// per-chunk stamps of chunk XTL_CHUNK (workgroup 100, thread 0): g_tl[600 + p]
#define XTL(p)                                                                                                        \
    do {                                                                                                              \
        if (g_tl && blockIdx.x == 100 && threadIdx.x == 0 && cb == 8 * 4096) g_tl[600 + (p)] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define TL_INIT
#define TL(p)
#define TL_NEXT
#define STL(p)
#define XTL(p)
#endif

// ------------------------------------------------------------------------------ small helpers
__device__ __forceinline__ bool bit_at(const uint8_t* p, uint64_t i) { return (ldu8(p + (i >> 3)) >> (i & 7)) & 1; }

struct ValidView {
    const uint8_t* bits;  // NULL = all valid
    uint64_t off;
    __device__ __forceinline__ bool get(uint64_t i) const { return !bits || bit_at(bits, off + i); }
};

// 32 bits starting at bit `pos` of an LSB-first bitmap holding `total_bits` bits (reads stay in range)
__device__ __forceinline__ uint32_t bits32(const uint8_t* p, uint64_t pos, uint64_t total_bits) {
    const uint64_t nbytes = (total_bits + 7) >> 3;
    const uint64_t b0 = pos >> 3;
    const uint32_t sh = (uint32_t)(pos & 7);
    uint64_t v = 0;
    if (nbytes >= 8) {  // one 8-byte load, pulled back inside the bitmap near its end (no divergent tail loop)
        const uint64_t b = b0 + 8 <= nbytes ? b0 : nbytes - 8;
        v = ldu64(p + b);
        const uint64_t skip = (b0 - b) * 8;
        v = skip < 64 ? v >> skip : 0;
    } else {
        for (uint32_t k = 0; k < 8 && b0 + k < nbytes; k++) v |= (uint64_t)ldu8(p + b0 + k) << (8 * k);
    }
    return (uint32_t)(v >> sh);
}

// Split form of bits32 for software pipelining: issue() only starts the 8-byte load, word() does the
// shifting when the bits are consumed (one loop iteration later), so the load's latency is not
// exposed in front of the loads that follow it.  `want` = number of bits needed (0..32).
struct VWord {
    uint64_t raw;
    uint32_t shift, mask;
    __device__ __forceinline__ uint32_t word() const { return (uint32_t)(raw >> shift) & mask; }
};
__device__ __forceinline__ VWord vword_issue(const uint8_t* p, uint64_t pos, uint64_t total_bits, uint32_t want) {
    VWord r;
    r.mask = want >= 32 ? 0xFFFFFFFFu : (1u << want) - 1;
    r.shift = 0;
    r.raw = ~0ull;
    if (!p || want == 0) return r;
    const uint64_t nbytes = (total_bits + 7) >> 3;
    const uint64_t b0 = pos >> 3;
    if (nbytes >= 8) {
        const uint64_t b = b0 + 8 <= nbytes ? b0 : nbytes - 8;
        r.raw = ldu64(p + b);
        r.shift = (uint32_t)((b0 - b) * 8 + (pos & 7));  // <= 56 + 7
    } else {
        r.raw = bits32(p, pos, total_bits);
    }
    return r;
}

template <int W>
__device__ __forceinline__ bool val_eq(const Val<W>& a, const Val<W>& b, uint32_t fkind) {
    if constexpr (W == 4) {
        if (fkind == 1) {  // OrderedFloat<f32>: NaN == NaN, -0 == +0
            float x = __uint_as_float(a.x), y = __uint_as_float(b.x);
            return (x != x && y != y) || x == y;
        }
        return a.x == b.x;
    } else if constexpr (W == 8) {
        if (fkind == 2) {
            double x = __longlong_as_double((long long)a.x), y = __longlong_as_double((long long)b.x);
            return (x != x && y != y) || x == y;
        }
        return a.x == b.x;
    } else if constexpr (W == 16) {
        return a.x.x == b.x.x && a.x.y == b.x.y && a.x.z == b.x.z && a.x.w == b.x.w;
    } else if constexpr (W == 32) {
        return a.x.x == b.x.x && a.x.y == b.x.y && a.x.z == b.x.z && a.x.w == b.x.w && a.y.x == b.y.x &&
               a.y.y == b.y.y && a.y.z == b.y.z && a.y.w == b.y.w;
    } else {
        return a.x == b.x;
    }
}
template <int W>
__device__ __forceinline__ Val<W> val_zero() {
    Val<W> v;
    __builtin_memset(&v, 0, sizeof(v));
    return v;
}


__device__ __forceinline__ uint32_t uleb_len(uint64_t v) {
    uint32_t n = 1;
    while (v >= 0x80) {
        v >>= 7;
        n++;
    }
    return n;
}

// bytes of the def-level section: u32 def_len | ULEB((ceil(N/8)<<1)|1) | ceil(N/8) bytes
__host__ __device__ __forceinline__ uint64_t def_section_bytes(uint64_t N) {
    uint64_t h = (((N + 7) / 8) << 1) | 1;
    uint32_t n = 1;
    while (h >= 0x80) {
        h >>= 7;
        n++;
    }
    return 4 + n + (N + 7) / 8;
}

// write the def-level section header (thread 0) and return where the bits go
__device__ __forceinline__ uint8_t* def_header(uint8_t* dst, uint64_t N) {
    const uint64_t nbytes = (N + 7) / 8;
    uint64_t h = (nbytes << 1) | 1;
    const uint32_t ul = uleb_len(h);
    if (threadIdx.x == 0) {
        stu32(dst, (uint32_t)(ul + nbytes));
        uint8_t* q = dst + 4;
        while (h >= 0x80) {
            *q++ = (uint8_t)(h | 0x80);
            h >>= 7;
        }
        *q = (uint8_t)h;
    }
    return dst + 4 + ul;
}

// def-level bits of rows [r0, r0+rows) of the page (validity slice re-packed from bit 0, pad bits 0;
// no validity bitmap => all ones: arrow2 write_def_levels (true, None) => repeat(true))
__device__ __forceinline__ void def_bits_tile(uint8_t* bits_dst, const ValidView& v, uint64_t page_row0, uint64_t N,
                                              uint64_t total_rows, uint64_t r0, uint32_t rows) {
    const int t = threadIdx.x;
    const uint32_t nbytes = (rows + 7) / 8;  // r0 is a multiple of TILE_ROWS: byte aligned in the page
    uint8_t* d = bits_dst + (r0 >> 3);
    for (uint32_t b4 = t * 4; b4 < nbytes; b4 += WG * 4) {
        const uint32_t bit0 = b4 * 8;
        uint32_t w = v.bits ? bits32(v.bits, v.off + page_row0 + r0 + bit0, v.off + total_rows) : 0xFFFFFFFFu;
        const uint32_t nb = min(32u, rows - bit0);
        if (nb < 32) w &= (1u << nb) - 1;
        const uint32_t nby = min(4u, nbytes - b4);
        if (nby == 4) {
            stu32(d + b4, w);
        } else {
            for (uint32_t k = 0; k < nby; k++) *(gptr)(d + b4 + k) = (uint8_t)(w >> (8 * k));
        }
    }
    (void)N;
}

// the whole def-level bit section of a page by one workgroup: U independent bitmap loads in flight per
// thread, unaligned dword stores (the section starts at an odd offset behind the ULEB header)
__device__ __forceinline__ void def_bits_page(uint8_t* bits_dst, const ValidView& v, uint64_t page_row0, uint64_t N,
                                              uint64_t total_rows) {
    constexpr int U = 4;
    const uint32_t nbytes = (uint32_t)((N + 7) / 8), nwords = (nbytes + 3) / 4;
    for (uint32_t w0 = threadIdx.x; w0 < nwords; w0 += WG * U) {
        VWord r[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint64_t bit0 = (uint64_t)(w0 + u * WG) * 32;
            const uint32_t want = bit0 < N ? (uint32_t)min((uint64_t)32, N - bit0) : 0u;
            r[u] = vword_issue(v.bits, v.off + page_row0 + bit0, v.off + total_rows, want);
            if (!want) r[u].mask = 0;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t wi = w0 + u * WG;
            if (wi >= nwords) break;
            const uint32_t w = r[u].word();
            uint8_t* d = bits_dst + (uint64_t)wi * 4;
            if (wi * 4 + 4 <= nbytes) {
                stu32(d, w);
            } else {
                for (uint32_t k = 0; wi * 4 + k < nbytes; k++) *(gptr)(d + k) = (uint8_t)(w >> (8 * k));
            }
        }
    }
}

__device__ __forceinline__ void put_hdr9(uint8_t* p, uint32_t codec, uint32_t csize, uint32_t usize) {
    p[0] = (uint8_t)codec;
    stu32(p + 1, csize);
    stu32(p + 5, usize);
}

// copy n bytes, both sides arbitrarily aligned (slot-relative headers make dst odd)
__device__ __forceinline__ void wg_copy(uint8_t* dst, const uint8_t* src, uint64_t n) {
    const int t = threadIdx.x;
    uint64_t head = (16 - ((uintptr_t)dst & 15)) & 15;
    if (head > n) head = n;
    if ((uint64_t)t < head) dst[t] = src[t];
    const uint64_t nvec = (n - head) >> 4;
    for (uint64_t i = t; i < nvec; i += WG) *(u32x4*)(dst + head + i * 16) = ldu128(src + head + i * 16);
    const uint64_t tail0 = head + (nvec << 4);
    if (tail0 + t < n) dst[tail0 + t] = src[tail0 + t];
}

// ------------------------------------------------------------------------------ RLE
// Run boundaries are valid rows whose value differs from the previous valid row's value; nulls
// extend the current run; leading nulls join the first run (integer/rle.rs:75-101).  Records:
// u32 count | value.  A run's value is the value of the row that opened it (the reference keeps
// `last_value` from the run's first row, so a float run carries the FIRST value's bits,
// rle.rs:79,87).
//
// gfx950 shape: one workgroup walks the page in chunks of 256*R rows; wave w owns R consecutive
// groups of 64 rows (lane = row).  "Previous valid row", "rank among boundaries" and "previous
// boundary" are wave64 ballots / v_mbcnt / readlane on 64-bit lane masks held in SGPRs, the
// carries between a wave's groups stay scalar, and the only cross-wave state is one small LDS
// record per wave (two LDS-only barriers per chunk, double-buffered).  Values are loaded once
// (coalesced) and stay in registers.
template <int W>
struct RleShape {
    static constexpr int R = W <= 8 ? 8 : (W == 16 ? 4 : 2);  // 64-row groups per wave per chunk
};
template <int W>
__device__ __forceinline__ Val<W> shfl_val(const Val<W>& v, int src) {
    Val<W> r;
    if constexpr (W <= 4) {
        r.x = (decltype(r.x))__shfl((uint32_t)v.x, src, 64);
    } else {
        uint32_t a[W / 4], b[W / 4];
        __builtin_memcpy(a, &v, W);
#pragma unroll
        for (int k = 0; k < W / 4; k++) b[k] = __shfl(a[k], src, 64);
        __builtin_memcpy(&r, b, W);
    }
    return r;
}
// value held by lane `src` (wave-uniform index) as a wave-uniform value (v_readlane -> SGPRs)
template <int W>
__device__ __forceinline__ Val<W> readlane_val(const Val<W>& v, int src) {
    Val<W> r;
    if constexpr (W <= 4) {
        r.x = (decltype(r.x))__builtin_amdgcn_readlane((int)(uint32_t)v.x, src);
    } else {
        int a[W / 4], b[W / 4];
        __builtin_memcpy(a, &v, W);
#pragma unroll
        for (int k = 0; k < W / 4; k++) b[k] = __builtin_amdgcn_readlane(a[k], src);
        __builtin_memcpy(&r, b, W);
    }
    return r;
}
__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, int src) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, src);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), src);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ int top_bit(uint64_t m) { return 63 - __clzll((long long)m); }
__device__ __forceinline__ uint32_t mbcnt64(uint64_t m) {  // set bits of m below my lane
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
}

// equality used for run detection: FK 0 = bit pattern, 1 = OrderedFloat<f32>, 2 = OrderedFloat<f64>
template <int W, int FK>
__device__ __forceinline__ bool rle_eq(const Val<W>& a, const Val<W>& b) {
    if constexpr (FK == 1 && W == 4) {
        if (a.x == b.x) return true;
        const uint32_t ax = a.x & 0x7FFFFFFFu, bx = b.x & 0x7FFFFFFFu;
        return ((ax | bx) == 0) || (ax > 0x7F800000u && bx > 0x7F800000u);  // +-0, NaN == NaN
    } else if constexpr (FK == 2 && W == 8) {
        if (a.x == b.x) return true;
        const uint64_t ax = a.x & 0x7FFFFFFFFFFFFFFFull, bx = b.x & 0x7FFFFFFFFFFFFFFFull;
        return ((ax | bx) == 0) || (ax > 0x7FF0000000000000ull && bx > 0x7FF0000000000000ull);
    } else {
        return val_eq<W>(a, b, 0);
    }
}

// validity mask of the 64-row group g of the chunk starting at page row cb (n rows in the chunk)
__device__ __forceinline__ uint64_t group_mask(const ValidView& vv, uint64_t vtotal, uint64_t cb, uint32_t g,
                                               uint32_t n) {
    if (64 * g >= n) return 0;
    uint64_t m = ~0ull;
    if (vv.bits) {
        const uint64_t p = vv.off + cb + 64 * g, byte = p >> 3, nbytes = (vtotal + 7) >> 3;
        const uint32_t sh = (uint32_t)(p & 7);
        if (byte + 9 <= nbytes) {
            const uint64_t lo = ldu64(vv.bits + byte);
            m = sh ? (lo >> sh) | ((uint64_t)ldu8(vv.bits + byte + 8) << (64 - sh)) : lo;
        } else {
            m = (uint64_t)bits32(vv.bits, p, vtotal) | ((uint64_t)bits32(vv.bits, p + 32, vtotal) << 32);
        }
    }
    if (n - 64 * g < 64) m &= (1ull << (n - 64 * g)) - 1;
    return m;
}

// LDS use: sA >= 64 words, sB >= 8 * W bytes
template <int W, int FK, class GetVal>
__device__ uint64_t enc_rle(GetVal getv, const ValidView& vv, uint64_t N, uint8_t* dst, uint32_t* sA, uint32_t* sB) {
    constexpr int REC = 4 + W;
    constexpr int R = RleShape<W>::R;
    constexpr uint32_t WROWS = 64 * R;       // rows per wave per chunk
    constexpr uint32_t CHUNK = 4 * WROWS;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint64_t lt = (1ull << lane) - 1;
    uint64_t run_start = 0;  // row where the open run started
    uint32_t nrec = 0;       // records closed so far (the open run is record `nrec`)
    bool have = false;       // a valid row has been seen
    Val<W> last = val_zero<W>();  // value of the last valid row seen so far
    const uint64_t vtotal = vv.off + N;
    auto lds_barrier = []() {  // LDS-only hand-off between the four waves
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    // software pipeline: the loads of chunk k+1 are issued before chunk k is processed
    Val<W> vn[R];
    uint64_t mn = 0;
    auto fetch = [&](uint64_t cb) {
        const uint32_t n = (uint32_t)min((uint64_t)CHUNK, N - cb);
        const uint32_t wrow = (uint32_t)w * WROWS;
#pragma unroll
        for (int j = 0; j < R; j++) {
            const uint32_t row = wrow + 64 * j + lane;
            vn[j] = getv(cb + (row < n ? row : n - 1));
        }
        mn = lane < R ? group_mask(vv, vtotal, cb, (uint32_t)w * R + lane, n) : 0;
    };
    if (N) fetch(0);
    uint32_t par = 0;
    for (uint64_t cb = 0; cb < N; cb += CHUNK, par ^= 1) {
        uint32_t* s_has = sA + par * 16;          // [4] wave has a valid row
        uint32_t* s_cnt = sA + par * 16 + 4;      // [4] boundaries found by the wave
        uint32_t* s_blast = sA + par * 16 + 8;    // [4] (last boundary row in chunk)+1, 0 = none
        Val<W>* s_last = (Val<W>*)sB + par * 4;   // [4] value of the wave's last valid row
        const uint32_t wrow = (uint32_t)w * WROWS;
        Val<W> v[R];
        uint64_t vm[R];
#pragma unroll
        for (int j = 0; j < R; j++) {
            v[j] = vn[j];
            vm[j] = readlane_u64(mn, j);
        }
        if (cb + CHUNK < N) fetch(cb + CHUNK);
        // ---- phase 1: wave summary (has a valid row, value of the last one)
        bool has_w = false;
        Val<W> last_w = val_zero<W>();
#pragma unroll
        for (int j = 0; j < R; j++)
            if (vm[j]) {
                has_w = true;
                last_w = readlane_val<W>(v[j], top_bit(vm[j]));
            }
        if (lane == 0) {
            s_has[w] = has_w;
            s_last[w] = last_w;
        }
        lds_barrier();
        // carry-in of this wave: nearest earlier wave with a valid row, else the chunk carry
        bool chas = have;
        Val<W> cval = last;
        bool earlier_has = false;
        for (int pw = 0; pw < 3; pw++)
            if (pw < w && s_has[pw]) {
                chas = true;
                earlier_has = true;
                cval = s_last[pw];
            }
        // the first run takes the first valid row's value
        if (!have && !earlier_has && has_w) {
            Val<W> fv = val_zero<W>();
            bool got = false;
#pragma unroll
            for (int j = 0; j < R; j++)
                if (!got && vm[j]) {
                    got = true;
                    fv = readlane_val<W>(v[j], (int)__ffsll((long long)vm[j]) - 1);
                }
            if (lane == 0) __builtin_memcpy(dst + 4, &fv, W);
        }
        // ---- phase 2: boundaries.  One compare per row: against the previous valid row of the
        // group if there is one, else against the carried value (lanes before the group's first
        // valid row).  Scalar work is kept minimal: the CU has ONE scalar ALU for its 16 waves.
        uint64_t bm[R];
        uint32_t cnt_w = 0, blast_w = 0;
#pragma unroll
        for (int j = 0; j < R; j++) {
            const uint64_t pm = vm[j] & lt;
            const bool has_prev = pm != 0;
            const Val<W> pv = shfl_val<W>(v[j], has_prev ? top_bit(pm) : 0);
            Val<W> other = has_prev ? pv : cval;
            const bool ne = !rle_eq<W, FK>(other, v[j]);
            const uint64_t nem = __ballot(ne && (has_prev || chas));
            bm[j] = nem & vm[j];
            const int tv = vm[j] ? top_bit(vm[j]) : 0;
            const Val<W> lv = readlane_val<W>(v[j], tv);
            chas = chas || (vm[j] != 0);
            cval = vm[j] ? lv : cval;
            cnt_w += (uint32_t)__popcll(bm[j]);
            blast_w = bm[j] ? wrow + 64 * j + (uint32_t)top_bit(bm[j]) + 1 : blast_w;
        }
        if (lane == 0) {
            s_cnt[w] = cnt_w;
            s_blast[w] = blast_w;
        }
        lds_barrier();
        // ---- phase 3: ranks and records
        uint32_t base = nrec;
        uint64_t start_prev = run_start;
        for (int pw = 0; pw < 3; pw++)
            if (pw < w) {
                base += s_cnt[pw];
                if (s_blast[pw]) start_prev = cb + s_blast[pw] - 1;
            }
#pragma unroll
        for (int j = 0; j < R; j++) {
            if (bm[j] == 0) continue;
            const bool b = __builtin_amdgcn_inverse_ballot_w64(bm[j]);
            const uint64_t grow = cb + wrow + 64 * j;
            const uint64_t pbm = bm[j] & lt;
            const uint64_t start = pbm ? grow + top_bit(pbm) : start_prev;
            if (b) {
                const uint32_t rk = base + mbcnt64(bm[j]);  // boundaries before this one
                uint8_t* closed = dst + (uint64_t)rk * REC;
                stu32(closed, (uint32_t)(grow + lane - start));  // count of the run that ends here
                __builtin_memcpy(closed + REC + 4, &v[j], W);    // value of the run that starts here
            }
            base += (uint32_t)__popcll(bm[j]);
            start_prev = grow + top_bit(bm[j]);
        }
        // ---- chunk carries (wave-uniform, identical in every wave)
        for (int pw = 0; pw < 4; pw++) {
            if (s_has[pw]) {
                have = true;
                last = s_last[pw];
            }
            nrec += s_cnt[pw];
            if (s_blast[pw]) run_start = cb + s_blast[pw] - 1;
        }
    }
    if (N == 0) return 0;
    if (threadIdx.x == 0) {  // close the final run; an all-null page is one run of T::default() (rle.rs:98-101)
        uint8_t* r = dst + (uint64_t)nrec * REC;
        stu32(r, (uint32_t)(N - run_start));
        if (!have) {
            const Val<W> z = val_zero<W>();
            __builtin_memcpy(r + 4, &z, W);
        }
    }
    __syncthreads();
    return (uint64_t)(nrec + 1) * REC;
}

// ---- RLE, row-segment shape (used for whole pages; the ballot shape above stays for the u32
// index streams nested in Dict pages, which only have a few LDS words to spare).
//
// lane = row costs ~1.7 VALU + ~1.7 SALU wave-instructions per row, because every 64-row group pays
// for its own 64-bit mask bookkeeping and the CU has a single scalar unit.  Here a thread owns K
// CONSECUTIVE rows instead: the chunk (256*K rows) is loaded coalesced, transposed through LDS
// (one pad element per thread segment -> conflict-free), and each thread walks its rows carrying
// "previous valid value".  Cross-thread state is one ballot + one permute per wave per chunk
// (carry of the last valid value, rank of the first boundary, previous boundary row); cross-wave
// state is four small LDS records.  The next chunk's global loads are issued one iteration ahead
// and stay in flight across the LDS-only barriers.
template <int W>
struct RleRows {
    static constexpr int K = W <= 8 ? 16 : (W == 16 ? 8 : 4);       // rows per thread per chunk
    static constexpr uint32_t CHUNK = WG * K;
    static constexpr uint32_t VAL_BYTES = (CHUNK + WG) * W;            // padded value tile
    static constexpr uint32_t WORDS = (VAL_BYTES + 15) / 16 * 4 + CHUNK / 32 + 64 + 4 * W;  // + s_vb + sA + sB
};

template <int W, int FK, class GetVal>
__device__ uint64_t enc_rle_rows(GetVal getv, const ValidView& vv, uint64_t N, uint8_t* dst, uint32_t* lds) {
    constexpr int REC = 4 + W;
    constexpr int K = RleRows<W>::K;
    constexpr uint32_t CHUNK = RleRows<W>::CHUNK;
    Val<W>* s_val = (Val<W>*)lds;
    uint32_t* s_vb = lds + (RleRows<W>::VAL_BYTES + 15) / 16 * 4;
    uint32_t* sA = s_vb + CHUNK / 32;
    uint32_t* sB = sA + 64;
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint64_t lt = (1ull << lane) - 1;
    uint64_t run_start = 0;  // row where the open run started
    uint32_t nrec = 0;       // records closed so far (the open run is record `nrec`)
    bool have = false;       // a valid row has been seen
    Val<W> last = val_zero<W>();
    const uint64_t vtotal = vv.off + N;
    auto lds_barrier = []() {  // LDS-only hand-off: global loads / stores stay in flight across it
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    };
    Val<W> vn[K];
    VWord wvn;  // validity word t of the chunk (threads < CHUNK/32), finished when it is staged
    auto fetch = [&](uint64_t cb) {
        const uint32_t n = (uint32_t)min((uint64_t)CHUNK, N - cb);
#pragma unroll
        for (int u = 0; u < K; u++) {
            const uint32_t row = (uint32_t)t + (uint32_t)u * WG;
            vn[u] = getv(cb + (row < n ? row : n - 1));
        }
        const uint32_t vb0 = (uint32_t)t * 32;
        wvn = vword_issue(vv.bits, vv.off + cb + vb0, vtotal, vb0 < n ? min(32u, n - vb0) : 0u);
        if (vb0 >= n) wvn.mask = 0;
    };
    if (N) fetch(0);
    uint32_t par = 0;
    TL_INIT
    for (uint64_t cb = 0; cb < N; cb += CHUNK, par ^= 1) {
        uint32_t* s_has = sA + par * 16;          // [4] wave has a valid row
        uint32_t* s_cnt = sA + par * 16 + 4;      // [4] boundaries found by the wave
        uint32_t* s_blast = sA + par * 16 + 8;    // [4] (last boundary row in chunk)+1, 0 = none
        Val<W>* s_last = (Val<W>*)sB + par * 4;   // [4] value of the wave's last valid row
        // ---- stage the chunk: values transposed through LDS, validity words
#pragma unroll
        for (int u = 0; u < K; u++) {
            const uint32_t i = (uint32_t)t + (uint32_t)u * WG;
            s_val[i + i / K] = vn[u];
        }
        TL(0);
        if (t < (int)(CHUNK / 32)) s_vb[t] = wvn.word();
        lds_barrier();
        TL(1);
        if (cb + CHUNK < N) fetch(cb + CHUNK);  // next chunk's loads fly while this one is processed
        TL(2);
        const uint32_t bit0 = (uint32_t)t * K;
        const uint32_t m = (s_vb[bit0 >> 5] >> (bit0 & 31)) & ((1u << K) - 1);
        // ---- phase 1: last valid value per thread -> per wave
        const Val<W> lastv = s_val[t * K + t + (m ? 31 - __clz((int)m) : 0)];
        const uint64_t hm = __ballot(m != 0);
        const bool has_w = hm != 0;
        const Val<W> last_w = readlane_val<W>(lastv, has_w ? top_bit(hm) : 0);
        if (lane == 0) {
            s_has[w] = has_w;
            s_last[w] = last_w;
        }
        lds_barrier();
        TL(3);
        bool chas = have;
        Val<W> cval = last;
        for (int pw = 0; pw < 3; pw++)
            if (pw < w && s_has[pw]) {
                chas = true;
                cval = s_last[pw];
            }
        // carry into this thread: nearest earlier lane of the wave with a valid row, else the wave's carry
        const uint64_t pm = hm & lt;
        const Val<W> pvs = shfl_val<W>(lastv, pm ? top_bit(pm) : 0);
        const bool pc = pm ? true : chas;
        const Val<W> pv = pm ? pvs : cval;
        if (!pc && m) {  // the very first valid row of the page gives the first run its value
            const Val<W> firstv = s_val[t * K + t + __ffs((int)m) - 1];
            st_val<W>(dst + 4, firstv);
        }
        // ---- phase 2: boundaries of my rows.  Forward-fill the previous valid value over null rows;
        // a row whose filled value differs BITWISE from its predecessor's is a candidate (null rows
        // never are).  Floats then drop the candidates OrderedFloat calls equal (+-0, NaN == NaN).
        uint32_t bmask = 0;
        {
            Val<W> v[K];  // all K LDS reads are issued back to back (one wait), then the dependent chain runs
#pragma unroll
            for (int j = 0; j < K; j++) v[j] = s_val[t * K + t + j];
            Val<W> e = pv;
#pragma unroll
            for (int j = 0; j < K; j++) {
                const Val<W> ej = ((m >> j) & 1) ? v[j] : e;
                if (!val_eq<W>(ej, e, 0)) bmask |= 1u << j;
                e = ej;
            }
        }
        if (!pc && m) bmask &= ~(1u << (__ffs((int)m) - 1));  // nothing before it: not a boundary
        if constexpr (FK != 0) {
            uint32_t cc = bmask;
            while (cc) {
                const int j = __ffs((int)cc) - 1;
                cc &= cc - 1;
                const uint32_t below = m & ((1u << j) - 1);
                const Val<W> cur = s_val[t * K + t + j];
                const Val<W> prev = below ? s_val[t * K + t + (31 - __clz((int)below))] : pv;
                if (rle_eq<W, FK>(prev, cur)) bmask &= ~(1u << j);
            }
        }
        TL(4);
        const uint32_t cnt = (uint32_t)__popc(bmask);
        const uint32_t blast = bmask ? (uint32_t)t * K + (31u - (uint32_t)__clz((int)bmask)) + 1 : 0;
        const uint32_t incl = wave_incl_scan(cnt);
        const uint64_t bmk = __ballot(bmask != 0);
        const uint64_t pb = bmk & lt;
        const uint32_t prev_blast = __shfl(blast, pb ? top_bit(pb) : 0, 64);
        const uint32_t cnt_w = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        const uint32_t blast_w = (uint32_t)__builtin_amdgcn_readlane((int)blast, bmk ? top_bit(bmk) : 0);
        if (lane == 0) {
            s_cnt[w] = cnt_w;
            s_blast[w] = bmk ? blast_w : 0;
        }
        lds_barrier();
        TL(5);
        // ---- phase 3: ranks and records
        uint32_t base = nrec;
        uint64_t start_prev = run_start;
        for (int pw = 0; pw < 3; pw++)
            if (pw < w) {
                base += s_cnt[pw];
                if (s_blast[pw]) start_prev = cb + s_blast[pw] - 1;
            }
        uint32_t rk = base + incl - cnt;  // boundaries before my first one
        uint64_t start = pb ? cb + prev_blast - 1 : start_prev;
        uint32_t bmm = bmask;
        while (bmm) {
            const int j = __ffs((int)bmm) - 1;
            bmm &= bmm - 1;
            const uint64_t row = cb + (uint64_t)t * K + j;
            const Val<W> bv = s_val[t * K + t + j];
            uint8_t* closed = dst + (uint64_t)rk * REC;
            stu32(closed, (uint32_t)(row - start));        // count of the run that ends here
            st_val<W>(closed + REC + 4, bv);               // value of the run that starts here
            start = row;
            rk++;
        }
        // ---- chunk carries (wave-uniform, identical in every wave)
        for (int pw = 0; pw < 4; pw++) {
            if (s_has[pw]) {
                have = true;
                last = s_last[pw];
            }
            nrec += s_cnt[pw];
            if (s_blast[pw]) run_start = cb + s_blast[pw] - 1;
        }
        TL(6);
        // the next chunk's staging overwrites s_val / s_vb: wait until every thread is past its reads
        lds_barrier();
        TL(7);
        TL_NEXT
    }
    if (N == 0) return 0;
    if (threadIdx.x == 0) {  // close the final run; an all-null page is one run of T::default() (rle.rs:98-101)
        uint8_t* r = dst + (uint64_t)nrec * REC;
        stu32(r, (uint32_t)(N - run_start));
        if (!have) {
            const Val<W> z = val_zero<W>();
            __builtin_memcpy(r + 4, &z, W);
        }
    }
    __syncthreads();
    return (uint64_t)(nrec + 1) * REC;
}

// ------------------------------------------------------------------------------ Patas (floats)
// double/patas.rs:36-104: first value raw, then per value `u16 packed | significant bytes` of
// bits ^ bits[ref], where ref = the last earlier row with the same bit pattern if it is less than 128
// rows back, else the previous row — and row 0 for a pattern not seen before while i < 128
// (`indices.get(..).unwrap_or(0)`, patas.rs:59-65).  Validity is ignored (null slots are values).
// A chunk of 4096 rows is staged in LDS behind a 128-row halo; every row searches its window
// backwards (one step inside runs), record lengths are scanned into byte positions, records are
// written with byte stores (they are 2..10 bytes at arbitrary offsets).
template <int W, class GetVal>
__device__ uint64_t enc_patas(GetVal getv, uint64_t N, uint8_t* dst, uint32_t* sA, uint32_t* s_w, uint8_t* s_raw) {
    static_assert(W == 4 || W == 8, "Patas is defined for f32 / f64");
    using B = typename std::conditional<(W == 8), unsigned long long, uint32_t>::type;
    constexpr int BITS = W * 8;
    constexpr uint32_t HALO = 128;
    B* s_v = (B*)s_raw;  // HALO + TILE_ROWS entries
    const int t = threadIdx.x;
    if (N == 0) return 0;
    auto bits_of = [&](uint64_t i) {
        const Val<W> v = getv(i);
        B b = 0;
        __builtin_memcpy(&b, &v, W);
        return b;
    };
    if (t == 0) {
        const B b0 = bits_of(0);
        for (int k = 0; k < W; k++) *(gptr)(dst + k) = (uint8_t)(b0 >> (8 * k));
    }
    uint64_t out_pos = W;
    for (uint64_t cb = 0; cb < N; cb += TILE_ROWS) {
        const uint32_t n = (uint32_t)min((uint64_t)TILE_ROWS, N - cb);
        for (uint32_t r = t; r < HALO + n; r += WG) {  // halo rows cb-128 .. cb-1, then the chunk
            const int64_t i = (int64_t)cb - HALO + r;
            s_v[r] = i >= 0 ? bits_of((uint64_t)i) : (B)0;
        }
        __syncthreads();
        uint32_t pk[ROWS_PER_THREAD], len[ROWS_PER_THREAD];
        B sv[ROWS_PER_THREAD];
#pragma unroll
        for (int j = 0; j < ROWS_PER_THREAD; j++) {
            const uint32_t r = (uint32_t)t + (uint32_t)j * WG;
            const uint64_t i = cb + r;
            pk[j] = 0;
            len[j] = 0;
            sv[j] = 0;
            if (r < n && i > 0) {
                const B b = s_v[HALO + r];
                const uint32_t back = (uint32_t)min((uint64_t)127, i);
                uint32_t diff = 0;
                for (uint32_t d0 = 1; d0 <= back && diff == 0; d0 += 8) {  // 8 LDS reads in flight per step
                    B cnd[8];
#pragma unroll
                    for (uint32_t q = 0; q < 8; q++) cnd[q] = s_v[HALO + r - (d0 + q <= back ? d0 + q : 0)];
#pragma unroll
                    for (int q = 7; q >= 0; q--)
                        if (d0 + q <= back && cnd[q] == b) diff = d0 + q;
                }
                if (diff == 0) diff = i < 128 ? (uint32_t)i : 1u;  // unseen: row 0 while i < 128, else the previous row
                const B x = b ^ s_v[HALO + r - diff];
                uint32_t tz, lz;
                if constexpr (W == 8) {
                    tz = x ? (uint32_t)__builtin_ctzll(x) : 64u;
                    lz = x ? (uint32_t)__builtin_clzll(x) : 64u;
                } else {
                    tz = x ? (uint32_t)__builtin_ctz(x) : 32u;
                    lz = x ? (uint32_t)__builtin_clz(x) : 32u;
                }
                const uint32_t is_equal = x == 0 ? 1u : 0u;
                const uint32_t sig_bits = is_equal ? 0u : BITS - tz - lz;
                const uint32_t sig_bytes = (sig_bits >> 3) + ((sig_bits & 7) != 0);
                const uint32_t sh = tz - is_equal;
                pk[j] = (((diff & 0xFF) << 9) | ((sig_bytes & 7) << 6) | (sh & 0xFF)) & 0xFFFF;  // patas.rs:145-149
                sv[j] = sh >= (uint32_t)BITS ? (B)0 : (B)(x >> sh);
                len[j] = 2 + sig_bytes;
            }
            sA[sidx((int)r)] = len[j];
        }
        __syncthreads();
        const uint32_t total = tile_incl_scan(sA, s_w);
#pragma unroll
        for (int j = 0; j < ROWS_PER_THREAD; j++) {
            const uint32_t r = (uint32_t)t + (uint32_t)j * WG;
            if (len[j] == 0) continue;
            uint8_t* q = dst + out_pos + sA[sidx((int)r)] - len[j];
            *(gptr)q = (uint8_t)pk[j];
            *(gptr)(q + 1) = (uint8_t)(pk[j] >> 8);
            for (uint32_t k = 0; k + 2 < len[j]; k++) *(gptr)(q + 2 + k) = (uint8_t)(sv[j] >> (8 * k));
        }
        out_pos += total;
        __syncthreads();
    }
    return out_pos;
}

// ------------------------------------------------------------------------------ bit-packing
// u8 num_bits | 16*num_bits bytes per 128 values (integer/bp.rs:48-62); num_bits from the RAW
// values also for the delta variant (delta_bp.rs:50); packing ORs unmasked values like the
// bitpacking crate does.
template <class GetU32>
__device__ uint64_t enc_bp(GetU32 getv, uint64_t N, bool delta, uint8_t* dst, uint32_t* sA, uint32_t* sB,
                           uint32_t* s_w) {
    const int t = threadIdx.x;
    constexpr int NB = TILE_ROWS / 128, K = TILE_ROWS / WG;
    __shared__ uint32_t s_nb[NB], s_off[NB + 1], s_woff[NB + 1];
    uint64_t out_pos = 0;
    uint32_t carry_prev = 0;            // the value before the tile (delta variant)
    // the next tile's values are requested before the current tile is packed: one HBM round trip per tile is hidden
    uint32_t nxt[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
        const uint64_t i = (uint64_t)t + (uint64_t)k * WG;
        nxt[k] = i < N ? getv(i) : 0u;
    }
    for (uint64_t cb = 0; cb < N; cb += TILE_ROWS) {
        const uint32_t n = (uint32_t)min((uint64_t)TILE_ROWS, N - cb);  // multiple of 128
        const uint32_t nblk = n / 128;
#pragma unroll
        for (int k = 0; k < K; k++) sA[sidx(t + k * (int)WG)] = nxt[k];
        if (cb + TILE_ROWS < N) {
#pragma unroll
            for (int k = 0; k < K; k++) {
                const uint64_t i = cb + TILE_ROWS + (uint64_t)t + (uint64_t)k * WG;
                nxt[k] = i < N ? getv(i) : 0u;
            }
        }
        __syncthreads();
        const uint32_t* src = sA;
        if (delta) {
            for (uint32_t i = t; i < n; i += WG) sB[sidx((int)i)] = sA[sidx((int)i)] - (i ? sA[sidx((int)i - 1)] : carry_prev);
            src = sB;
        }
        // num_bits per block: 8 threads per block
        {
            const uint32_t blk = t >> 3, sub = t & 7;
            uint32_t acc = 0;
            if (blk < nblk)
                for (uint32_t k = 0; k < 16; k++) acc |= sA[sidx((int)(blk * 128 + sub * 16 + k))];
            acc |= __shfl_xor(acc, 1, 64);
            acc |= __shfl_xor(acc, 2, 64);
            acc |= __shfl_xor(acc, 4, 64);
            if (sub == 0 && blk < nblk) s_nb[blk] = acc ? 32 - __clz(acc) : 0;
        }
        __syncthreads();
        if (t < 64) {   // byte offset and first output word of every block (one wave scan)
            const uint32_t nb = (uint32_t)t < nblk ? s_nb[t] : 0u;
            const uint32_t by = (uint32_t)t < nblk ? 1 + 16 * nb : 0u;
            const uint32_t ib = wave_incl_scan(by), iw = wave_incl_scan(4 * nb);
            if ((uint32_t)t < nblk) {
                s_off[t] = ib - by;
                s_woff[t] = iw - 4 * nb;
            }
            if ((uint32_t)t == nblk - 1) {
                s_off[nblk] = ib;
                s_woff[nblk] = iw;
            }
        }
        __syncthreads();
        // pack: one output word per thread and step over ALL words of the tile (a block of nb bits has 4 * nb words)
        const uint32_t total_words = s_woff[nblk];
        for (uint32_t j = t; j < total_words; j += WG) {
            uint32_t lo = 0, hi = nblk;   // the block of word j: last b with s_woff[b] <= j
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (s_woff[mid] <= j)
                    lo = mid;
                else
                    hi = mid;
            }
            const uint32_t blk = lo, wi = j - s_woff[blk], nb = s_nb[blk];
            const uint32_t l = wi & 3, k = wi >> 2;  // word k of lane l
            const uint32_t lo_bit = 32 * k, hi_bit = 32 * k + 32;
            uint32_t word = 0;
            const uint32_t i0 = lo_bit / nb, i1 = min(31u, (hi_bit - 1) / nb);
            for (uint32_t i = i0; i <= i1; i++) {
                const uint32_t v = src[sidx((int)(blk * 128 + 4 * i + l))];
                const uint32_t bitpos = i * nb;
                if (bitpos >= lo_bit)
                    word |= v << (bitpos - lo_bit);
                else if (bitpos + nb > lo_bit)  // slot straddles in from the previous word
                    word |= v >> (lo_bit - bitpos);
            }
            stu32(dst + out_pos + s_off[blk] + 1 + 4 * wi, word);
        }
        if ((uint32_t)t < nblk) dst[out_pos + s_off[t]] = (uint8_t)s_nb[t];
        carry_prev = sA[sidx((int)n - 1)];
        out_pos += s_off[nblk];
        __syncthreads();
    }
    (void)s_w;
    return out_pos;
}

// ------------------------------------------------------------------------------ LZ4 block encode
// Byte-exact restatement of LZ4_compress_default (liblz4 1.9.x: greedy parse, one hash probe per
// position, skip acceleration 1) — what lz4::block::compress_to_buffer(src, None, false, dst) runs
// (reference call site src/compression/basic.rs:108-120).  The 16 KB hash table lives in LDS.
// One wave compresses one block with that parse, 64 probe positions at a time.
// The greedy search examines positions q_0, q_1, ... whose spacing follows the skip schedule
// (step = searchMatchNb++ >> 6) until one of them finds a 4-byte match through the hash table; every
// examined position is inserted.  Nothing in that depends on the outcome of earlier probes of the same
// search except the table contents, so lane l takes probe 64 b + l of batch b: it hashes its position,
// reads the table, and sees the insertions of the lower lanes of its own batch through a match-any over
// the hash bits (the latest lower lane with the same hash supersedes the table value).  The first lane
// with a match wins; lanes up to it commit their insertions (per hash the highest such lane), the
// others commit nothing.  The position tested right after a match (`ip` with literal length 0,
// LZ4_compress_generic's _next_match tail) is probe 0 of the next search's first batch.
// Match extension and literal copies are wave-wide.
__device__ __forceinline__ uint32_t lz4_hash_bits(uint64_t bytes, bool by_u16) {
    if (by_u16) return ((uint32_t)bytes * 2654435761u) >> (32 - 13);
    return (uint32_t)(((bytes << 24) * 889523592379ull) >> (64 - 12));
}
__device__ __forceinline__ uint64_t lz4_ld8_safe(const uint8_t* src, uint32_t pos, uint32_t n) {
    if (pos + 8 <= n) return ldu64(src + pos);
    uint64_t v = 0;
    for (uint32_t b = 0; pos + b < n && b < 8; b++) v |= (uint64_t)src[pos + b] << (8 * b);
    return v;
}
__device__ __forceinline__ void wave_copy_bytes(uint8_t* dst, const uint8_t* src, uint32_t len) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t body = len & ~7u;
    for (uint32_t i = lane * 8; i < body; i += 512) stu64(dst + i, ldu64(src + i));
    if (lane < (len & 7u)) dst[body + lane] = src[body + lane];
}
// probe number -> position for a search whose first regular probe is at p0 (q_0 = p0, q_{k+1} = q_k + step_k,
// step_0 = 1, step_k = (63 + k) >> 6)
__device__ __forceinline__ uint32_t lz4_probe_pos(uint32_t p0, uint32_t k) {
    if (k == 0) return p0;
    const uint32_t f = (k - 1) >> 6, r = (k - 1) & 63;
    return p0 + 1 + 32 * f * (f + 1) + r * (f + 1);
}
__device__ uint32_t lz4_compress_wave(const uint8_t* src, uint32_t n, uint8_t* dst, uint32_t* tab4096) {
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t below = lane ? (~0ull >> (64 - lane)) : 0ull;
    const bool by_u16 = n < 65536u + 11u;  // LZ4_64Klimit
    const int hbits = by_u16 ? 13 : 12;
    uint16_t* tab16 = (uint16_t*)tab4096;
    if (by_u16)
        for (uint32_t i = lane; i < 8192; i += 64) tab16[i] = 0;
    else
        for (uint32_t i = lane; i < 4096; i += 64) tab4096[i] = 0;
    auto tab_get = [&](uint32_t h) { return by_u16 ? (uint32_t)tab16[h] : tab4096[h]; };
    auto tab_put = [&](uint32_t h, uint32_t v) {
        if (by_u16)
            tab16[h] = (uint16_t)v;
        else
            tab4096[h] = v;
    };
    uint32_t op = 0, anchor = 0;
    if (n >= 13) {
        const uint32_t mfl1 = n - 11;        // mflimitPlusOne
        const uint32_t matchlimit = n - 5;
        uint32_t ip = 1;                     // first position to examine
        uint32_t preput = 0;                 // position inserted before the search starts
        bool special = false;                // probe 0 = the position right after a match
        for (;;) {
            // ------------------------------------------------ search
            uint32_t mpos = 0;
            bool have = false;
            for (uint32_t b = 0;; b++) {
                const uint32_t idx = 64 * b + lane;
                uint32_t q, qn;
                bool valid;
                if (special) {
                    if (idx == 0) {
                        q = ip;
                        qn = ip;
                        valid = true;
                    } else {
                        q = lz4_probe_pos(ip + 1, idx - 1);
                        qn = lz4_probe_pos(ip + 1, idx);
                        valid = qn <= mfl1;
                    }
                } else {
                    q = lz4_probe_pos(ip, idx);
                    qn = lz4_probe_pos(ip, idx + 1);
                    valid = qn <= mfl1;
                }
                const uint64_t vmask = __ballot(valid);  // a prefix of the lanes
                const uint64_t data = valid ? ldu64(src + q) : 0;
                if (b == 0) {  // every lane computes the same insertion; one writes it
                    const uint32_t hp = lz4_hash_bits(ldu64(src + preput), by_u16);
                    if (lane == 0) tab_put(hp, preput);
                }
                const uint32_t h = lz4_hash_bits(data, by_u16);
                const uint32_t v = valid ? tab_get(h) : 0;
                // lanes with my hash (valid ones only)
                uint64_t peers = vmask;
                for (int bit = 0; bit < hbits; bit++) {
                    const bool mine = (h >> bit) & 1;
                    const uint64_t m = __ballot(mine);
                    peers &= mine ? m : ~m;
                }
                const uint64_t lower = peers & below;
                const int prev = lower ? 63 - __clzll((long long)lower) : 0;
                const uint32_t qprev = (uint32_t)__shfl((int)q, prev, 64);
                const uint32_t m = lower ? qprev : v;
                const bool cand = valid && (by_u16 || m + 65535u >= q);
                const bool found = cand && ldu32(src + m) == (uint32_t)data;
                const uint64_t fm = __ballot(found);
                const int kf = fm ? __ffsll((long long)fm) - 1 : 63;
                const uint64_t upto = kf == 63 ? ~0ull : ((2ull << kf) - 1);
                // commit: per hash the highest probing lane that really ran
                if (valid && (lane <= (uint32_t)kf) && !(peers & ~below & ~(1ull << lane) & upto)) tab_put(h, q);
                if (fm) {
                    ip = (uint32_t)__shfl((int)q, kf, 64);
                    mpos = (uint32_t)__shfl((int)m, kf, 64);
                    have = true;
                    break;
                }
                if (vmask != ~0ull) break;  // ran into the end of the block
            }
            if (!have) break;
            // ------------------------------------------------ catch up (ip > anchor only for regular probes)
            while (ip > anchor && mpos > 0) {
                const uint32_t room = min(ip - anchor, mpos);
                const bool eq = lane < room && src[ip - 1 - lane] == src[mpos - 1 - lane];
                const uint64_t ne = ~__ballot(eq);
                const uint32_t back = ne ? (uint32_t)(__ffsll((long long)ne) - 1) : 64;
                ip -= back;
                mpos -= back;
                if (back < 64) break;
            }
            // ------------------------------------------------ literals
            const uint32_t lit = ip - anchor;
            const uint32_t tok = op++;
            uint32_t tokv;
            if (lit >= 15) {
                tokv = 15u << 4;
                const uint32_t rest = lit - 15, n255 = rest / 255;
                for (uint32_t i = lane; i < n255; i += 64) dst[op + i] = 255;
                if (lane == 0) dst[op + n255] = (uint8_t)(rest - n255 * 255);
                op += n255 + 1;
            } else {
                tokv = lit << 4;
            }
            wave_copy_bytes(dst + op, src + anchor, lit);
            op += lit;
            // ------------------------------------------------ match
            {
                const uint32_t off = ip - mpos;
                if (lane == 0) {
                    dst[op] = (uint8_t)off;
                    dst[op + 1] = (uint8_t)(off >> 8);
                }
                op += 2;
                uint32_t mc = 0;
                for (;;) {  // LZ4_count(ip + 4, match + 4, matchlimit), 512 bytes per round
                    const uint32_t a0 = ip + 4 + mc + 8 * lane;
                    uint64_t d = ~0ull;
                    if (a0 < matchlimit) {
                        d = lz4_ld8_safe(src, a0, n) ^ lz4_ld8_safe(src, mpos + 4 + mc + 8 * lane, n);
                        const uint32_t room = matchlimit - a0;
                        if (room < 8) d |= ~0ull << (8 * room);
                    }
                    const uint64_t nz = __ballot(d != 0);
                    if (nz) {
                        const int fl = __ffsll((long long)nz) - 1;
                        const uint64_t dl = (uint64_t)__shfl((long long)d, fl, 64);
                        mc += 8 * fl + (uint32_t)((__ffsll((long long)dl) - 1) >> 3);
                        break;
                    }
                    mc += 512;
                }
                ip += mc + 4;
                if (mc >= 15) {
                    tokv += 15;
                    const uint32_t rest = mc - 15, n255 = rest / 255;
                    for (uint32_t i = lane; i < n255; i += 64) dst[op + i] = 255;
                    if (lane == 0) dst[op + n255] = (uint8_t)(rest - n255 * 255);
                    op += n255 + 1;
                } else {
                    tokv += mc;
                }
                if (lane == 0) dst[tok] = (uint8_t)tokv;
            }
            anchor = ip;
            if (ip >= mfl1) break;
            preput = ip - 2;
            special = true;
        }
    }
    {  // last literals
        const uint32_t last = n - anchor;
        if (last >= 15) {
            const uint32_t rest = last - 15, n255 = rest / 255;
            if (lane == 0) dst[op] = 15u << 4;
            op++;
            for (uint32_t i = lane; i < n255; i += 64) dst[op + i] = 255;
            if (lane == 0) dst[op + n255] = (uint8_t)(rest - n255 * 255);
            op += n255 + 1;
        } else {
            if (lane == 0) dst[op] = (uint8_t)(last << 4);
            op++;
        }
        wave_copy_bytes(dst + op, src + anchor, last);
        op += last;
    }
    return op;
}

// LZ4 block of one page sub-buffer, executed by one wave.  Default: the parallel format-valid encoder (sb_lz4.h);
// SB_WRITE_LZ4_EXACT: the serial greedy parse above, byte-identical to LZ4_compress_default.
__device__ __forceinline__ uint32_t lz4_compress_block(const uint8_t* src, uint32_t n, uint8_t* dst, uint32_t* ws16k, uint32_t flags) {
    if (flags & SB_WRITE_LZ4_EXACT) return lz4_compress_wave(src, n, dst, ws16k);
    return lz4_compress_wave_fast<11, 13>(src, n, dst, *reinterpret_cast<Lz4EncLds<11, 13>*>(ws16k));
}

// Snappy stream of one page sub-buffer: the LZ4 matcher with Snappy's element syntax (sb_lz4.h), one wave of the
// workgroup; ws16k: 16 KiB of LDS; s_out: one LDS word for the size.  All threads of the workgroup call it.
__device__ __forceinline__ uint32_t snappy_compress_block_wg(const uint8_t* src, uint32_t n, uint8_t* dst, uint32_t* ws16k, uint32_t* s_out) {
    __syncthreads();
    uint32_t sz = 0;
    if (threadIdx.x < 64) sz = snappy_compress_wave<11, 13>(src, n, dst, *reinterpret_cast<Lz4EncLds<11, 13>*>(ws16k));
    if (threadIdx.x == 0) *s_out = sz;
    __syncthreads();
    return *s_out;
}
__device__ uint32_t zstd_store_frame_wg(const uint8_t* src, uint32_t n, uint8_t* dst, uint32_t* s4);

__device__ uint32_t lz4_stitch_block(const uint8_t* pool, uint32_t slot_stride, uint32_t chunk_bytes, const uint8_t* src, uint32_t n,
                                     uint32_t nch, uint8_t* dst, uint32_t* sh, uint32_t part = 0, uint32_t parts = 1);
// An LZ4 block inside a page kernel (the u32 indices of a Dict page whose row count is not a multiple of 128: 68 KB on the
// 16 960-row last page of a 1 M-row column, 2 ms through one wave while the other three wait): three waves compress a
// third of the block each into slots of `tmp` (HBM), then the workgroup joins them like k_enc_lz4_stitch does.
// lds: 3 * sizeof(Lz4EncLds<11, 13>) bytes (reused by the join); all threads call it.  0: block too small / no room.
constexpr uint32_t LZ3_MIN = 12 * 1024;
__device__ uint32_t lz4_compress_block_wg3(const uint8_t* src, uint32_t n, uint8_t* dst, uint32_t* lds, uint32_t lds_bytes, uint8_t* tmp,
                                          uint64_t tmp_bytes) {
    typedef Lz4EncLds<11, 13> L;
    const uint32_t chunk = ((n + 2) / 3 + 1023) / 1024 * 1024;
    const uint32_t stride = (16 + chunk + chunk / 255 + 64 + 15) / 16 * 16;
    const uint32_t nch = (n + chunk - 1) / chunk;
    if (n < LZ3_MIN || lds_bytes < 3 * sizeof(L) || lds_bytes < (5 * WG + 8) * 4 || tmp_bytes < (uint64_t)stride * nch) return 0;
    __syncthreads();
    const uint32_t w = threadIdx.x >> 6;
    if (w < nch) {
        L* l = reinterpret_cast<L*>(lds) + w;
        const uint32_t c0 = w * chunk, c1 = min(n, c0 + chunk);
        uint8_t* slot = tmp + (uint64_t)w * stride;
        uint32_t anchor = c0;
        const uint32_t len = lz4_compress_range<11, 13, true>(src, n, c0, c1, slot + 16, *l, &anchor);
        if ((threadIdx.x & 63) == 0) {
            stu32(slot, len);
            stu32(slot + 4, anchor);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const uint32_t sz = lz4_stitch_block(tmp, stride, chunk, src, n, nch, dst, lds);
    __syncthreads();
    return sz;
}

// ------------------------------------------------------------------------------ u32 blocks (nested)
// compress_integer::<u32> of an index array without validity: hdr9 + body.  Returns bytes written.
__device__ uint64_t enc_u32_block(const uint32_t* idx, uint64_t N, int32_t codec, uint8_t* dst, uint32_t* sA,
                                  uint32_t* sB, uint32_t* sC, uint32_t* s_w, Status* st, uint32_t page, uint32_t flags,
                                  uint8_t* tmp = nullptr, uint64_t tmp_bytes = 0 /* HBM scratch for the three-wave LZ4 path */) {
    auto getu = [=](uint64_t i) { return idx[i]; };
    auto getv = [=](uint64_t i) {
        Val<4> v;
        v.x = idx[i];
        return v;
    };
    uint64_t body = 0;
    ValidView none{nullptr, 0};
    switch (codec) {
        case SB_CODEC_NONE:
            wg_copy(dst + 9, (const uint8_t*)idx, N * 4);
            body = N * 4;
            break;
        case SB_CODEC_RLE:
            body = enc_rle<4, 0>(getv, none, N, dst + 9, sA, sB);
            break;
        case SB_CODEC_ONEVALUE:
            if (threadIdx.x == 0) stu32(dst + 9, N ? idx[0] : 0);
            body = 4;
            break;
        case SB_CODEC_ZSTD:
            __syncthreads();
            body = zstd_store_frame_wg((const uint8_t*)idx, (uint32_t)(N * 4), dst + 9, s_w);
            break;
        case SB_CODEC_SNAPPY:
            body = snappy_compress_block_wg((const uint8_t*)idx, (uint32_t)(N * 4), dst + 9, sA, s_w);
            break;
        case SB_CODEC_LZ4: {  // Basic(Lz4) over the raw index bytes (integer/mod.rs:55-58)
            __syncthreads();
            if (!(flags & SB_WRITE_LZ4_EXACT)) {
                // The u32 indices of a page whose row count is no multiple of 128 (bit-packing is not allowed then: every
                // column's last page).  The default parse is free to choose, and a block like this
                // is read by ONE wave whatever else the call holds — 17 000 sequences took 2 ms on both sides, longer than
                // all other pages of a 64-column batch together.  So it is written as ONE literal run: a valid LZ4 block
                // every decoder copies out (here: 16-byte wave copies), ~2x its compressed size on a page that is 1 / 60 of
                // its column (C3: +1 % page bytes).  SB_WRITE_LZ4_EXACT keeps liblz4's parse.
                const uint32_t nbytes = (uint32_t)(N * 4);
                uint8_t* o = dst + 9;
                uint32_t hdr = 1;
                if (nbytes >= 15) hdr += 1 + (nbytes - 15) / 255;
                if (threadIdx.x == 0) {
                    o[0] = (uint8_t)((nbytes >= 15 ? 15u : nbytes) << 4);
                    if (nbytes >= 15) {
                        uint32_t rest = nbytes - 15, k = 1;
                        for (; rest >= 255; rest -= 255) o[k++] = 255;
                        o[k] = (uint8_t)rest;
                    }
                }
                wg_copy(o + hdr, (const uint8_t*)idx, nbytes);
                body = hdr + nbytes;
                break;
            }
            if (!(flags & SB_WRITE_LZ4_EXACT) && tmp && sB > sA && sC - sB == sB - sA) {
                // (sA, sB, sC are one contiguous LDS area in the page kernels: 3 * LW words)
                const uint32_t z = lz4_compress_block_wg3((const uint8_t*)idx, (uint32_t)(N * 4), dst + 9, sA,
                                                          (uint32_t)((sC - sA) + (sC - sB)) * 4, tmp, tmp_bytes);
                if (z) {
                    body = z;
                    break;
                }
            }
            uint32_t sz = 0;
            if (threadIdx.x < 64) sz = lz4_compress_block((const uint8_t*)idx, (uint32_t)(N * 4), dst + 9, sA, flags);
            __syncthreads();
            if (threadIdx.x == 0) s_w[0] = sz;
            __syncthreads();
            body = s_w[0];
            break;
        }
        case SB_CODEC_BITPACKING:
        case SB_CODEC_DELTA_BITPACKING:
            if (N % 128 != 0) {
                if (threadIdx.x == 0) raise(st, SB_ERR_NYI, page, 500);  // the crate asserts BLOCK_LEN
                return 0;
            }
            body = enc_bp(getu, N, codec == SB_CODEC_DELTA_BITPACKING, dst + 9, sA, sB, s_w);
            break;
        default:
            if (threadIdx.x == 0) raise(st, SB_ERR_NYI, page, 501);
            return 0;
    }
    __syncthreads();
    if (threadIdx.x == 0) put_hdr9(dst, (uint32_t)codec, (uint32_t)body, (uint32_t)(N * 4));
    return 9 + body;
}

// ------------------------------------------------------------------------------ Dict
__device__ __forceinline__ uint32_t hash64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return (uint32_t)(z ^ (z >> 31));
}
template <int W>
__device__ __forceinline__ uint32_t hash_val(const Val<W>& v) {
    if constexpr (W <= 8) {
        uint64_t k = 0;
        __builtin_memcpy(&k, &v, W);
        return hash64(k + 0x9E3779B97F4A7C15ull);
    } else {
        uint64_t w[W / 8];
        __builtin_memcpy(w, &v, W);
        uint64_t h = 0x9E3779B97F4A7C15ull;
        for (int i = 0; i < W / 8; i++) h = (uint64_t)hash64(h ^ w[i]) * 0x9E3779B97F4A7C15ull + w[i];
        return hash64(h);
    }
}
// Dict equality: raw bytes, except that a float NaN never equals anything (RawNative derives
// PartialEq from T, integer/dict.rs:208,225-229)
template <int W>
__device__ __forceinline__ bool dict_eq(const Val<W>& a, const Val<W>& b, uint32_t fkind) {
    if constexpr (W == 4) {
        if (fkind == 1) {
            float x = __uint_as_float(a.x);
            if (x != x) return false;
        }
    }
    if constexpr (W == 8) {
        if (fkind == 2) {
            double x = __longlong_as_double((long long)a.x);
            if (x != x) return false;
        }
    }
    return val_eq<W>(a, b, 0);
}

__device__ __forceinline__ uint32_t table_load(uint32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Generic first-occurrence dictionary builder.  KeyOps supplies hash(i), eq(i, j) for rows that
// carry a key; keyed(i) says whether row i interns a key (valid rows, and row 0 even if null).
// Outputs idx[N] (u32) in aux and the first-occurrence rows in `firsts` (dict order).
// Returns the number of entries.
// key classes with a hash-only fast mode (BinKeysHashed)
template <class T, class = void>
struct is_hashed_keys { static constexpr bool value = false; };
template <class T>
struct is_hashed_keys<T, decltype((void)T::HASHED)> { static constexpr bool value = true; };
template <class KeyOps>
__device__ uint32_t dict_build(KeyOps ko, uint64_t N, uint32_t* aux, uint64_t aux_words, uint32_t** idx_out,
                               uint32_t** firsts_out, uint32_t* sA, uint32_t* sB, uint32_t* s_w, Status* st,
                               uint32_t page, uint32_t* lds_table = nullptr, uint32_t lds_slots = 0) {
    const int t = threadIdx.x;
    uint64_t M = 64;
    while (M < 2 * N) M <<= 1;
    if (M + 3 * N > aux_words) {
        if (t == 0) raise(st, SB_ERR_INVALID, page, 510);
        return EMPTY;
    }
    // (workspace arrays in HBM: gld32 / gst32 — see sb_common.h — keep them off the LDS counter)
    uint32_t* F = aux + M;       // row -> first row with the same key
    uint32_t* R = F + N;         // first row -> dictionary id ; later: dict id -> first row (firsts)
    uint32_t* idx = R + N;
    __shared__ uint32_t s_keys, s_collide;
    constexpr bool HASHED = is_hashed_keys<KeyOps>::value;
    bool exact_mode = !HASHED;
    // The table of first rows: in LDS while the page has few distinct keys (tier 0: every probe is an LDS access; string
    // keys are compared through L2), in HBM otherwise (tier 1: pow2 >= 2N slots in the aux area).
    for (int tier = (lds_table && lds_slots) ? 0 : 1; tier < 2; tier++) {
        const SlotTable table{tier == 0 ? lds_table : aux, tier == 0};
        const uint64_t slots = tier == 0 ? lds_slots : M;
        const uint32_t cap = tier == 0 ? min(lds_slots / 8 * 5, lds_slots - (uint32_t)WG * 16 - 1) : 0xFFFFFFFFu;   // (see distinct_count)
        for (uint64_t i = t; i < slots; i += WG) table.st((uint32_t)i, EMPTY);
        if (t == 0) s_keys = 0;
        __syncthreads();
        // (fences between the phases of a page kernel are WORKGROUP scope: one workgroup owns the page and runs on one CU,
        // whose waves share the vector L1; an agent-scope release / acquire writes back and invalidates L2 — ~100 us per
        // fence once the chip is busy, measured on k_enc_bin_hash: 5.8 ms with, 0.5 ms without)
        if (tier == 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        const uint32_t mask = (uint32_t)(slots - 1);
        bool overflow = false;
        // phase 1: insert; the slot of a key ends up holding the smallest row that carries it.  The slot a row
        // landed in is remembered (in F), so that phase 2 does not hash and compare the key a second time.
        for (uint64_t base = 0; base < N; base += (uint64_t)WG * 16) {
            // eight rows in flight per thread, probed in rounds (see distinct_count)
            for (int j0 = 0; j0 < 16; j0 += 8) {
                uint64_t iu[8];
                uint32_t hu[8], cu[8];
                uint32_t pend = 0;
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    iu[u] = base + (uint64_t)(j0 + u) * WG + t;
                    const bool in = iu[u] < N;
                    if (!in) iu[u] = 0;
                    if (in) {
                        if (ko.keyed(iu[u])) pend |= 1u << u;
                        else gst32(F + iu[u], EMPTY);
                    }
                    hu[u] = ko.hash(iu[u]) & mask;
                }
                while (pend) {
#pragma unroll
                    for (int u = 0; u < 8; u++)
                        if ((pend >> u) & 1) cu[u] = table.ld(hu[u]);
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        if (!((pend >> u) & 1) || cu[u] != EMPTY) continue;
                        const uint32_t old = table.cas(hu[u], EMPTY, (uint32_t)iu[u]);
                        if (old == EMPTY) {
                            if (tier == 0) atomicAdd(&s_keys, 1u);
                            gst32(F + iu[u], hu[u]);
                            pend &= ~(1u << u);
                        } else {
                            cu[u] = old;
                        }
                    }
                    bool same[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const uint32_t cr = ((pend >> u) & 1) ? cu[u] : (uint32_t)iu[u];
                        if constexpr (HASHED) same[u] = cr == (uint32_t)iu[u] || (exact_mode ? ko.eq(cr, iu[u]) : ko.heq(cr, iu[u]));
                        else same[u] = cr == (uint32_t)iu[u] || ko.eq(cr, iu[u]);
                    }
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        if (!((pend >> u) & 1)) continue;
                        if (same[u]) {
                            if ((uint32_t)iu[u] < cu[u]) table.amin(hu[u], (uint32_t)iu[u]);  // rows arrive roughly in order: rarely needed
                            gst32(F + iu[u], hu[u]);
                            pend &= ~(1u << u);
                        } else {
                            hu[u] = (hu[u] + 1) & mask;
                        }
                    }
                }
            }
            if (tier == 0) {   // too many distinct keys for the LDS table: start over on the HBM one
                __syncthreads();
                if (s_keys > cap) {
                    overflow = true;
                    break;
                }
            }
        }
        __syncthreads();
        STL(21);
        if (overflow) continue;
        if (tier == 1) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        // phase 2: F[i] = first row of row i's key (a key never leaves its slot)
        for (uint64_t i = t; i < N; i += WG) {
            const uint32_t h = gld32(F + i);
            if (h != EMPTY) gst32(F + i, table.ld(h));
        }
        __syncthreads();
        STL(22);
        if constexpr (HASHED) {
            if (!exact_mode) {   // every row against the first row of its hash class
                if (t == 0) s_collide = 0;
                __syncthreads();
                uint32_t bad = 0;
                constexpr int VU = 8;    // rows in flight per thread: the pass is a chain of dependent random accesses
                for (uint64_t base = t; base < N; base += (uint64_t)WG * VU) {
                    uint32_t f[VU];
                    uint64_t row[VU];
#pragma unroll
                    for (int u = 0; u < VU; u++) {
                        row[u] = base + (uint64_t)u * WG;
                        f[u] = row[u] < N ? gld32(F + row[u]) : EMPTY;
                        if (f[u] == (uint32_t)row[u]) f[u] = EMPTY;   // a first row needs no check
                    }
                    if (!ko.template exact_batch<VU>(f, row)) bad = 1;
                }
                if (bad) atomicOr(&s_collide, 1u);
                __syncthreads();
                if (s_collide) {     // two different strings with one hash: the exact build on the HBM table
                    exact_mode = true;
                    tier = 0;        // (tier++ -> 1)
                    __syncthreads();
                    continue;
                }
            }
        }
        break;
    }
    STL(23);
    // phase 3: dictionary ids in first-occurrence order (rank of first rows), chunked scan; firsts[id] = the first row
    // (the HBM table area: not needed any more, or never used)
    uint32_t* firsts = aux;
    uint32_t nent = 0;
    for (uint64_t cb = 0; cb < N; cb += TILE_ROWS) {
        const uint32_t n = (uint32_t)min((uint64_t)TILE_ROWS, N - cb);
        for (uint32_t i = t; i < TILE_ROWS; i += WG) sA[sidx((int)i)] = (i < n && gld32(F + cb + i) == (uint32_t)(cb + i)) ? 1 : 0;
        __syncthreads();
        const uint32_t tot = tile_incl_scan(sA, s_w);
        for (uint32_t i = t; i < n; i += WG) {
            const uint32_t inc = sA[sidx((int)i)], prev = i ? sA[sidx((int)i - 1)] : 0;
            if (inc != prev) {   // a first row
                gst32(R + cb + i, nent + inc - 1);
                gst32(firsts + nent + inc - 1, (uint32_t)(cb + i));
            }
        }
        nent += tot;
        __syncthreads();
    }
    __syncthreads();
    STL(24);
    // phase 4: idx[i] = id of the key of the last keyed row <= i (nulls repeat the previous index)
    uint32_t carry_last = 0;  // (last keyed row)+1 seen in earlier chunks; row 0 is always keyed
    for (uint64_t cb = 0; cb < N; cb += TILE_ROWS) {
        const uint32_t n = (uint32_t)min((uint64_t)TILE_ROWS, N - cb);
        for (uint32_t i = t; i < TILE_ROWS; i += WG) sB[sidx((int)i)] = (i < n && gld32(F + cb + i) != EMPTY) ? i + 1 : 0;
        __syncthreads();
        tile_incl_scan_max(sB, s_w);
        for (uint32_t i = t; i < n; i += WG) {
            const uint32_t lk = sB[sidx((int)i)];
            const uint64_t row = lk ? cb + lk - 1 : (uint64_t)carry_last - 1;
            gst32(idx + cb + i, gld32(R + gld32(F + row)));
        }
        const uint32_t lk = sB[sidx((int)n - 1)];
        if (lk) carry_last = (uint32_t)(cb + lk);
        __syncthreads();
    }
    STL(25);
    __syncthreads();
    STL(26);
    *idx_out = idx;
    *firsts_out = firsts;
    return nent;
}

template <int W>
struct PrimKeys {
    const uint8_t* vals;  // page values
    ValidView vv;
    uint32_t fkind;
    __device__ __forceinline__ bool keyed(uint64_t i) const { return i == 0 || vv.get(i); }
    __device__ __forceinline__ Val<W> key(uint64_t i) const {
        if (i == 0 && !vv.get(0)) return val_zero<W>();  // a leading null interns T::default() (dict.rs:46-50)
        return ld_val<W>(vals + i * W);
    }
    __device__ __forceinline__ uint32_t hash(uint64_t i) const { return hash_val<W>(key(i)); }
    __device__ __forceinline__ bool eq(uint64_t a, uint64_t b) const { return dict_eq<W>(key(a), key(b), fkind); }
};

#include "sb_dict_lds.h"

template <class O>
struct BinKeys {
    const uint8_t* offs;  // page offsets (N+1), absolute into values
    const uint8_t* values;
    ValidView vv;
    __device__ __forceinline__ bool keyed(uint64_t i) const { return i == 0 || vv.get(i); }
    __device__ __forceinline__ uint64_t beg(uint64_t i) const {
        if constexpr (sizeof(O) == 4) return (uint64_t)(O)ldu32(offs + i * 4);
        else return (uint64_t)(O)ldu64(offs + i * 8);
    }
    __device__ __forceinline__ uint32_t hash(uint64_t i) const {
        const uint64_t b = beg(i), e = beg(i + 1);
        uint64_t h = 0xcbf29ce484222325ull ^ (e - b);
        uint64_t p = b;
        for (; p + 8 <= e; p += 8) h = (h ^ ldu64(values + p)) * 0x100000001b3ull + (h >> 29);
        for (; p < e; p++) h = (h ^ ldu8(values + p)) * 0x100000001b3ull;
        return hash64(h);
    }
    __device__ __forceinline__ bool eq(uint64_t a, uint64_t b) const {
        const uint64_t ab = beg(a), ae = beg(a + 1), bb = beg(b), be = beg(b + 1);
        if (ae - ab != be - bb) return false;
        const uint64_t n = ae - ab;
        uint64_t k = 0;
        for (; k + 8 <= n; k += 8)
            if (ldu64(values + ab + k) != ldu64(values + bb + k)) return false;
        for (; k < n; k++)
            if (ldu8(values + ab + k) != ldu8(values + bb + k)) return false;
        return true;
    }
};

// ---- binary pages: one 64-bit hash per row, computed ONCE per page in a streaming pass (four rows in flight per
// thread), so that the selector's statistics and the dictionary build probe with 8-byte keys instead of chasing
// offsets -> bytes for both strings of every comparison.  Statistics treat equal hashes (the length is mixed in) as
// equal strings; the dictionary build, which decides what bytes a page holds, still compares the strings.
__device__ __forceinline__ uint64_t bin_hash_bytes(const uint8_t* values, uint64_t b, uint64_t e, uint64_t values_len) {
    uint64_t h = 0x9E3779B97F4A7C15ull ^ ((e - b) * 0xD6E8FEB86659FD93ull);
    uint64_t p = b;
    for (; p + 8 <= e; p += 8) {
        h = (h ^ ldu64(values + p)) * 0xFF51AFD7ED558CCDull;
        h ^= h >> 32;
    }
    if (p < e) {
        uint64_t w = 0;
        if (p + 8 <= values_len) {
            w = ldu64(values + p) & ((1ull << (8 * (e - p))) - 1);
        } else {
            for (uint64_t k = 0; p + k < e; k++) w |= (uint64_t)ldu8(values + p + k) << (8 * k);
        }
        h = (h ^ w) * 0xC4CEB9FE1A85EC53ull;
        h ^= h >> 29;
    }
    h *= 0x94D049BB133111EBull;
    return h ^ (h >> 31);
}
// the same hash over a string staged in LDS (`l`: dword array with 8 bytes of slack) at byte b .. e
__device__ __forceinline__ uint64_t lds3_rd8(l32p a, uint32_t x) {   // 8 bytes at byte x of an LDS dword array (8 bytes of slack)
    const uint32_t w = x >> 2;
    const uint32_t d0 = a[w], d1 = a[w + 1], d2 = a[w + 2];
    return (uint64_t)__builtin_amdgcn_alignbyte(d1, d0, x & 3) | ((uint64_t)__builtin_amdgcn_alignbyte(d2, d1, x & 3) << 32);
}
__device__ __forceinline__ uint64_t bin_hash_lds(l32p l, uint32_t b, uint32_t e) {
    uint64_t h = 0x9E3779B97F4A7C15ull ^ ((uint64_t)(e - b) * 0xD6E8FEB86659FD93ull);
    uint32_t p = b;
    for (; p + 8 <= e; p += 8) {
        h = (h ^ lds3_rd8(l, p)) * 0xFF51AFD7ED558CCDull;
        h ^= h >> 32;
    }
    if (p < e) {
        const uint64_t w = lds3_rd8(l, p) & ((1ull << (8 * (e - p))) - 1);
        h = (h ^ w) * 0xC4CEB9FE1A85EC53ull;
        h ^= h >> 29;
    }
    h *= 0x94D049BB133111EBull;
    return h ^ (h >> 31);
}
// bin_hash_rows through LDS: the value bytes of a page are one contiguous range, so instead of every thread chasing
// offsets -> bytes (3-4 dependent HBM round trips per 4 rows, 0.49 ms per 64 Ki-row page) the workgroup streams the
// bytes of a tile of rows into LDS with coalesced 16-byte loads and hashes from there (two round trips per ~3000 rows).
template <class O>
__device__ void bin_hash_rows_staged(const BinKeys<O>& bk, uint64_t N, uint64_t values_len, uint64_t* h64, uint32_t* lds, uint32_t lds_bytes,
                                     uint64_t row_begin = 0, bool read_back = true) {   // rows [row_begin, N); read_back: the caller reads h64 in this kernel
    const uint32_t t = threadIdx.x;
    const uint32_t cap = lds_bytes - 16;
    uint64_t r0 = row_begin;
    while (r0 < N) {
        uint64_t R = min((uint64_t)4096, N - r0);
        const uint64_t b0 = bk.beg(r0);
        uint64_t nb = bk.beg(r0 + R) - b0;
        while (nb > cap && R > 1) {      // (uniform: rows of more than ~15 bytes on average)
            R >>= 1;
            nb = bk.beg(r0 + R) - b0;
        }
        if (nb > cap) {                  // one row larger than the staging area: straight from HBM
            if (t == 0) gst64(h64 + r0, bin_hash_bytes(bk.values, b0, b0 + nb, values_len));
            r0 += 1;
            continue;
        }
        __syncthreads();
        for (uint64_t x = (uint64_t)t * 16; x < nb; x += (uint64_t)WG * 16) {
            u32x4 v;
            if (b0 + x + 16 <= values_len) {
                v = ldu128(bk.values + b0 + x);
            } else {
                uint32_t w[4] = {0, 0, 0, 0};
                for (uint32_t q = 0; q < 16 && b0 + x + q < values_len; q++) w[q >> 2] |= (uint32_t)ldu8(bk.values + b0 + x + q) << (8 * (q & 3));
                v = u32x4{w[0], w[1], w[2], w[3]};
            }
            l32p d = (l32p)lds + (x >> 2);
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
        __syncthreads();
        for (uint64_t r = r0 + t; r < r0 + R; r += WG) {
            const uint32_t b = (uint32_t)(bk.beg(r) - b0), e = (uint32_t)(bk.beg(r + 1) - b0);
            gst64(h64 + r, bin_hash_lds((l32p)lds, b, e));
        }
        r0 += R;
    }
    __syncthreads();
    if (read_back) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
}
template <class O>
__device__ void bin_hash_rows(const BinKeys<O>& bk, uint64_t N, uint64_t values_len, uint64_t* h64) {
    constexpr int U = 4;
    for (uint64_t base = threadIdx.x; base < N; base += (uint64_t)WG * U) {
        uint64_t b[U], e[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint64_t i = base + (uint64_t)u * WG;
            b[u] = i < N ? bk.beg(i) : 0;
            e[u] = i < N ? bk.beg(i + 1) : 0;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint64_t i = base + (uint64_t)u * WG;
            if (i < N) gst64(h64 + i, bin_hash_bytes(bk.values, b[u], e[u], values_len));
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
// statistics over hashed rows (choose_bin): key equality = hash equality
template <class O>
struct KeyOpsBinHashed {
    BinKeys<O> k;
    const uint64_t* h64;
    __device__ __forceinline__ uint32_t hash(uint64_t i) const { return (uint32_t)(gld64(h64 + i) >> 17); }
    __device__ __forceinline__ bool eq(uint64_t a, uint64_t b) const { return gld64(h64 + a) == gld64(h64 + b); }
    __device__ __forceinline__ uint32_t weight(uint64_t i) const { return (uint32_t)(k.beg(i + 1) - k.beg(i)) + 8; }
    // eq(a, b) == (key64(a) == key64(b)): lets a pass keep the key it compares against in a register
    __device__ __forceinline__ uint64_t key64(uint64_t i) const { return gld64(h64 + i); }
};
template <class T, class = void>
struct has_key64 { static constexpr bool value = false; };
template <class T>
struct has_key64<T, decltype((void)&T::key64)> { static constexpr bool value = true; };
// dictionary build over hashed rows: the hash finds the slot and rejects different strings, the strings decide
template <class O>
struct BinKeysHashed {
    BinKeys<O> k;
    const uint64_t* h64;
    __device__ __forceinline__ bool keyed(uint64_t i) const { return k.keyed(i); }
    __device__ __forceinline__ uint64_t beg(uint64_t i) const { return k.beg(i); }
    __device__ __forceinline__ uint32_t hash(uint64_t i) const { return (uint32_t)(gld64(h64 + i) >> 17); }
    __device__ __forceinline__ bool eq(uint64_t a, uint64_t b) const { return gld64(h64 + a) == gld64(h64 + b) && k.eq(a, b); }
    // dict_build's fast mode: classes by hash alone (no offsets -> bytes chase inside the probe loop), then ONE streaming
    // pass checks every row against the first row of its class, four rows in flight per thread; a 64-bit collision (never
    // seen, but it would merge two different strings) sends the page through the exact build
    static constexpr bool HASHED = true;
    __device__ __forceinline__ bool heq(uint64_t a, uint64_t b) const { return gld64(h64 + a) == gld64(h64 + b); }
    __device__ __forceinline__ bool exact(uint64_t a, uint64_t b) const { return k.eq(a, b); }
    uint64_t values_len = 0;   // bytes readable behind k.values (0: exact_batch takes the slow path)
    // row[u] against row f[u] (EMPTY: nothing to check) for U rows at once: three rounds of independent accesses —
    // offsets, then the first 24 bytes of both strings with byte masks — instead of U chains of eq()'s early-exit loop.
    // Strings longer than 24 bytes or within 24 bytes of the buffer's end take eq().
    template <int U>
    __device__ __forceinline__ bool exact_batch(const uint32_t (&f)[U], const uint64_t (&row)[U]) const {
        uint64_t bi[U], bf[U], n[U];
        bool ok = true, fast[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const bool need = f[u] != 0xFFFFFFFFu;
            const uint64_t r = need ? row[u] : 0, g = need ? f[u] : 0;
            bi[u] = k.beg(r);
            bf[u] = k.beg(g);
            n[u] = k.beg(r + 1) - bi[u];
            const uint64_t nf = k.beg(g + 1) - bf[u];
            if (need && nf != n[u]) ok = false;
            fast[u] = need && n[u] <= 24 && bi[u] + 24 <= values_len && bf[u] + 24 <= values_len;
            if (need && !fast[u] && nf == n[u] && !k.eq(g, r)) ok = false;
        }
        uint64_t x = 0;
        if (values_len < 24) return ok;   // (no fast row: nothing below may touch the buffer)
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint8_t* pi = k.values + (fast[u] ? bi[u] : 0);
            const uint8_t* pf = k.values + (fast[u] ? bf[u] : 0);
#pragma unroll
            for (int w = 0; w < 3; w++) {
                const uint64_t d = ldu64(pi + 8 * w) ^ ldu64(pf + 8 * w);
                const uint64_t have = n[u] > (uint64_t)(8 * w) ? n[u] - 8 * w : 0;   // bytes of this word that belong to the string
                const uint64_t m = have >= 8 ? ~0ull : ((1ull << (8 * have)) - 1);
                if (fast[u]) x |= d & m;
            }
        }
        return ok && x == 0;
    }
};

// ------------------------------------------------------------------------------ adaptive selection
}  // namespace sb
#include "sb_select.h"
namespace sb {

struct SelScratch {
    uint32_t* lds_tab;   // SEL_LDS_SLOTS words
    uint32_t* s_misc;    // >= 2*WG + 16 words
    uint8_t* sample_mem; // SAMPLE_ROWS * (W + 1) bytes, 16-byte aligned
    uint32_t* gtab;      // HBM table for the distinct count (may be null)
    uint64_t gslots;
    uint32_t lds_slots = SEL_LDS_SLOTS;  // slots of lds_tab (a power of two)
    // binary pages (hashed rows): the distinct count keeps 32-bit tags in the LDS slots (no HBM access per probe) and notes
    // the slot of every row in slot16 (HBM) — the dictionary builder starts from there (bin_dict_handover); *tagged = 1 when
    // the count was taken that way and the table holds the smallest row of every class
    uint16_t* slot16 = nullptr;
    uint32_t* tagged = nullptr;
};

// ---- hand-over of a binary page's dictionary from the selector to the builder (aux area of the page, words) -------------
constexpr uint32_t BH_MAGIC = 0x48444231u;       // a dictionary was handed over (and the tag table counted the page's keys)
constexpr uint32_t BH_TAGS_USED = 0x48444230u;   // the tag table counted the page's keys, the page is not a Dict page
constexpr uint32_t SB_WRITE_DEBUG_VERIFY_FAIL_BIT = 1u << 30;   // (sb_write_options.flags, tests) k_enc_bin_verify fails every page
constexpr uint32_t BH_W_MAGIC = 0, BH_W_D = 1, BH_W_BAD = 2, BH_W_ID16 = 16;
constexpr uint32_t BH_SLOTS = 12288;                                  // tag slots (3/4 of BIN_LDS_SLOTS: the last quarter is the
                                                                      // hand-over's bitmap + prefix); up to 5/8 of them hold keys
constexpr uint32_t BH_W_REP16 = BH_W_ID16 + BH_SLOTS / 2;             // id16[BH_SLOTS] as u16
constexpr uint32_t BH_W_FIRSTS = BH_W_REP16 + BH_SLOTS / 2;           // rep16[BH_SLOTS] as u16: smallest row of the class
constexpr uint32_t BH_W_SLOT16 = BH_W_FIRSTS + BH_SLOTS;              // firsts[<= BH_SLOTS]
__host__ __device__ __forceinline__ uint64_t bh_table_slots(uint64_t N) {
    uint64_t M = 64;
    while (M < 2 * N) M <<= 1;
    return M;
}
// the fixed tables + slot16 sit in [0, M) of the aux area, the LZ4 scratch of the index block in [M, M + 2N), idx in the last N words
__host__ __device__ __forceinline__ bool bh_fits(uint64_t N, uint64_t aux_bytes) {
    const uint64_t M = bh_table_slots(N);
    return N >= 2048 && N <= 65536 && BH_W_SLOT16 + (N + 1) / 2 + 4 <= M && aux_bytes / 4 >= M + 3 * N;
}
constexpr uint32_t SEL_TAG_FAIL = 0xFFFFFFFEu;

// row-index hash-set operations over canonical primitive keys
template <int W, class Key>
struct KeyOpsPrim {
    Key key;
    __device__ __forceinline__ uint32_t hash(uint64_t i) const { return stat_hash<W>(key(i)); }
    __device__ __forceinline__ bool eq(uint64_t a, uint64_t b) const { return bits_eq<W>(key(a), key(b)); }
    __device__ __forceinline__ uint32_t weight(uint64_t) const { return 0; }
};

// Exact number of distinct keys among rows [0,N), or a value > limit once the count is known to
// exceed it.  LDS tier first; restart on the HBM table when more than SEL_LDS_SLOTS/2 keys show
// up.  *weight_sum receives the sum of ops.weight(i) over the first row of every distinct key.
template <class Ops>
__device__ uint32_t distinct_count(Ops ops, uint64_t N, uint32_t limit, const SelScratch& sc, uint64_t* weight_sum) {
    const int t = threadIdx.x;
    __shared__ uint32_t s_cnt;
    __shared__ unsigned long long s_wsum;
    for (int tier = 0; tier < 2; tier++) {
        const SlotTable tab{tier == 0 ? sc.lds_tab : sc.gtab, tier == 0};
        const uint64_t slots = tier == 0 ? sc.lds_slots : sc.gslots;
        if (tier == 1 && (!sc.gtab || sc.gslots < 2 * N)) return limit + 1;
        // LDS tier up to a load of 5/8 — and never so full that the WG * 16 rows inserted between two checks could
        // occupy every slot (a probe for one more key would then never find an empty one)
        const uint32_t cap = tier == 0 ? min(sc.lds_slots / 8 * 5, sc.lds_slots - (uint32_t)WG * 16 - 1) : 0xFFFFFFFFu;
        for (uint64_t i = t; i < slots; i += WG) tab.st((uint32_t)i, SEL_EMPTY);
        if (t == 0) {
            s_cnt = 0;
            s_wsum = 0;
        }
        __syncthreads();
        if (tier == 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        const uint32_t mask = (uint32_t)(slots - 1);
        bool overflow = false;
        for (uint64_t base = 0; base < N; base += WG * 16) {
            // Eight rows in flight per thread, probed in rounds: a key comparison is a dependent chain (table slot -> the
            // slot's row -> its key in HBM), so every round issues the slot reads of all unsettled rows together, then their
            // key comparisons together; a row leaves when it finds its key or claims an empty slot, the others step to
            // their next slot.  (One row at a time costs the chain's latency per row: 0.6 ms per 64 Ki-row page.)
            for (int j0 = 0; j0 < 16; j0 += 8) {
                uint64_t iu[8];
                uint32_t hu[8], cu[8];
                uint32_t pend = 0;
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    iu[u] = base + (uint64_t)(j0 + u) * WG + t;
                    if (iu[u] < N) pend |= 1u << u;
                    else iu[u] = 0;
                    hu[u] = ops.hash(iu[u]) & mask;
                }
                while (pend) {
#pragma unroll
                    for (int u = 0; u < 8; u++)
                        if ((pend >> u) & 1) cu[u] = tab.ld(hu[u]);
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        if (!((pend >> u) & 1) || cu[u] != SEL_EMPTY) continue;
                        const uint32_t old = tab.cas(hu[u], SEL_EMPTY, (uint32_t)iu[u]);
                        if (old == SEL_EMPTY) {
                            atomicAdd(&s_cnt, 1u);
                            const uint32_t wgt = ops.weight(iu[u]);
                            if (wgt) atomicAdd(&s_wsum, (unsigned long long)wgt);
                            pend &= ~(1u << u);
                        } else {
                            cu[u] = old;
                        }
                    }
                    bool same[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) same[u] = ops.eq(((pend >> u) & 1) ? cu[u] : (uint32_t)iu[u], iu[u]);
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        if (!((pend >> u) & 1)) continue;
                        if (same[u]) pend &= ~(1u << u);
                        else hu[u] = (hu[u] + 1) & mask;
                    }
                }
            }
            __syncthreads();
            const uint32_t c = s_cnt;
            if (c > limit) return c;
            if (c > cap) {
                overflow = true;
                break;
            }
        }
        __syncthreads();
        if (!overflow) {
            if (weight_sum) *weight_sum = s_wsum;
            return s_cnt;
        }
    }
    return limit + 1;
}

// The distinct count of a binary page over its 64-bit row hashes with NO HBM access inside the probe loop: a slot holds a
// 32-bit tag (hash bits 32..63; the home slot comes from bits 0..31), a probe compares tags, and the slot of every row goes
// to sc.slot16.  Equal strings always meet in one slot (linear probing without deletions); two DIFFERENT hashes with one
// tag that meet in a probe sequence would share a slot — about 4e-7 per 64 Ki-row page — and so would two different
// strings with one 64-bit hash: k_enc_bin_verify compares the string of EVERY row with the smallest row of its slot
// afterwards, and a page that fails is selected again without tags (k_enc_select's redo pass), so the count a decision
// rests on is exact.  On return the table holds the smallest row of every class (rep16 in the aux area as well).
__device__ __forceinline__ uint32_t bh_home(uint64_t k) { return (uint32_t)(((uint64_t)(uint32_t)k * BH_SLOTS) >> 32); }
template <class Ops>
__device__ uint32_t distinct_count_tags(Ops ops, uint64_t N, uint32_t limit, const SelScratch& sc, uint64_t* weight_sum, uint32_t* aux) {
    const int t = threadIdx.x;
    __shared__ uint32_t s_tcnt;
    __shared__ unsigned long long s_twsum;
    uint32_t* tab = sc.lds_tab;
    constexpr uint32_t cap = BH_SLOTS / 8 * 5;   // (WG * 16 rows between two checks never fill the other 3/8)
    static_assert(BH_SLOTS - BH_SLOTS / 8 * 5 > WG * 16 + 1, "a step could fill the table");
    for (uint32_t i = t; i < BH_SLOTS; i += WG) tab[i] = SEL_EMPTY;
    if (t == 0) {
        s_tcnt = 0;
        s_twsum = 0;
    }
    __syncthreads();
    uint16_t* slot16 = sc.slot16;
    STL(47);
    constexpr int U = 16;   // rows per thread and step; the keys of the next step are requested before this step's probes
    uint64_t kn[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        const uint64_t i = (uint64_t)u * WG + t;
        kn[u] = ops.key64(i < N ? i : 0);
    }
    for (uint64_t base = 0; base < N; base += (uint64_t)WG * U) {
        uint32_t hh[U], tg[U];
        uint32_t pend = 0, newk = 0;
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint64_t i = base + (uint64_t)u * WG + t;
            if (i < N) pend |= 1u << u;
            hh[u] = bh_home(kn[u]);
            tg[u] = (uint32_t)(kn[u] >> 32);
            if (tg[u] == SEL_EMPTY) tg[u] = 0xFFFFFFFEu;
        }
        const uint64_t nbase = base + (uint64_t)WG * U;
        if (nbase < N) {
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint64_t i = nbase + (uint64_t)u * WG + t;
                kn[u] = ops.key64(i < N ? i : 0);
            }
        }
        // probes in rounds: the slot reads of all unsettled rows are issued together
        while (pend) {
            uint32_t cur[U];
#pragma unroll
            for (int u = 0; u < U; u++)
                if ((pend >> u) & 1) cur[u] = tab[hh[u]];
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (!((pend >> u) & 1)) continue;
                uint32_t c = cur[u];
                if (c == SEL_EMPTY) {
                    c = atomicCAS(&tab[hh[u]], SEL_EMPTY, tg[u]);
                    if (c == SEL_EMPTY) {
                        newk++;
                        c = tg[u];
                    }
                }
                if (c == tg[u])
                    pend &= ~(1u << u);
                else
                    hh[u] = hh[u] + 1 == BH_SLOTS ? 0u : hh[u] + 1;
            }
        }
        if (newk) atomicAdd(&s_tcnt, newk);
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint64_t i = base + (uint64_t)u * WG + t;
            if (i < N) *(__attribute__((address_space(1))) uint16_t*)(slot16 + i) = (uint16_t)hh[u];
        }
        __syncthreads();
        const uint32_t c = s_tcnt;
        if (c > limit) return c;
        if (c > cap) return SEL_TAG_FAIL;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    STL(48);
    // the smallest row of every class (the tags are not needed any more): its length is the class's weight, the verify
    // kernel compares every row with it, the hand-over starts from it
    for (uint32_t i = t; i < BH_SLOTS; i += WG) tab[i] = SEL_EMPTY;
    __syncthreads();
    for (uint64_t base = (uint64_t)t * 2; base < N; base += (uint64_t)WG * 2 * 8) {
        uint32_t sl[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint64_t i = base + (uint64_t)u * WG * 2;
            sl[u] = i + 1 < N ? gld32((const uint32_t*)(slot16 + i)) : (i < N ? (uint32_t)ldu16((const uint8_t*)(slot16 + i)) | 0xFFFF0000u : 0xFFFFFFFFu);
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint32_t i = (uint32_t)(base + (uint64_t)u * WG * 2);
            const uint32_t s0 = sl[u] & 0xFFFFu, s1 = sl[u] >> 16;
            if (s0 != 0xFFFFu && i < tab[s0]) atomicMin(&tab[s0], i);
            if (s1 != 0xFFFFu && i + 1 < tab[s1]) atomicMin(&tab[s1], i + 1);
        }
    }
    __syncthreads();
    unsigned long long wsum = 0;
    uint16_t* rep16 = (uint16_t*)(aux + BH_W_REP16);
    for (uint32_t sl = t; sl < BH_SLOTS; sl += WG * 8) {   // (eight independent offset pairs in flight)
        uint32_t wv[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint32_t q = sl + (uint32_t)u * WG;
            const uint32_t cur = q < BH_SLOTS ? tab[q] : SEL_EMPTY;
            wv[u] = cur != SEL_EMPTY ? ops.weight(cur) : 0u;
            if (q < BH_SLOTS) *(__attribute__((address_space(1))) uint16_t*)(rep16 + q) = (uint16_t)cur;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) wsum += wv[u];
    }
    if (wsum) atomicAdd(&s_twsum, wsum);
    __syncthreads();
    STL(49);
    if (weight_sum) *weight_sum = s_twsum;
    return s_tcnt;
}

// Dictionary ids in first-occurrence order from the selector's table (see distinct_count_tags), left in the page's aux area
// for the builder: id16[slot], firsts[id], D.  tab: BH_SLOTS words (smallest row per class); rank: 4096 words of LDS.
template <class O>
__device__ void bin_dict_handover(const BinKeys<O>& bk, uint64_t N, uint32_t* tab, uint32_t* rank, uint32_t* s4, uint32_t* aux) {
    const uint32_t t = threadIdx.x, lane = t & 63, w = t >> 6;
    const uint16_t* slot16 = (const uint16_t*)(aux + BH_W_SLOT16);
    // first KEYED row per slot (row 0 and the valid rows intern keys, binary/dict.rs:55-93); without a validity bitmap
    // that is the smallest row the table already holds
    if (bk.vv.bits) {
        for (uint32_t sl = t; sl < BH_SLOTS; sl += WG) tab[sl] = SEL_EMPTY;
        __syncthreads();
        for (uint64_t base = t; base < N; base += (uint64_t)WG * 8) {
            uint32_t sl[8];
            bool kd[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const uint64_t i = base + (uint64_t)u * WG;
                sl[u] = i < N ? ldu16((const uint8_t*)(slot16 + i)) : 0u;
                kd[u] = i < N && bk.keyed(i);
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const uint32_t i = (uint32_t)(base + (uint64_t)u * WG);
                if (kd[u] && i < tab[sl[u]]) atomicMin(&tab[sl[u]], i);
            }
        }
    }
    for (uint32_t k = t; k < 4096; k += WG) rank[k] = 0;
    __syncthreads();
    // bitmap of the first rows, then the exclusive popcount prefix of its 2048 words (8 words per thread)
    for (uint32_t sl = t; sl < BH_SLOTS; sl += WG) {
        const uint32_t r = tab[sl];
        if (r != SEL_EMPTY) atomicOr(&rank[r >> 5], 1u << (r & 31));
    }
    __syncthreads();
    uint32_t mine = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) mine += (uint32_t)__popc(rank[t * 8 + k]);
    const uint32_t incl = wave_incl_scan(mine);
    if (lane == 63) s4[w] = incl;
    __syncthreads();
    uint32_t run = incl - mine;
    for (uint32_t pw = 0; pw < w; pw++) run += s4[pw];
    const uint32_t D = s4[0] + s4[1] + s4[2] + s4[3];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        rank[2048 + t * 8 + k] = run;
        run += (uint32_t)__popc(rank[t * 8 + k]);
    }
    __syncthreads();
    uint16_t* id16 = (uint16_t*)(aux + BH_W_ID16);
    uint32_t* firsts = aux + BH_W_FIRSTS;
    for (uint32_t sl = t; sl < BH_SLOTS; sl += WG) {
        const uint32_t r = tab[sl];
        uint32_t id = 0xFFFFu;
        if (r != SEL_EMPTY) {
            id = rank[2048 + (r >> 5)] + (uint32_t)__popc(rank[r >> 5] & ((1u << (r & 31)) - 1u));
            gst32(firsts + id, r);
        }
        *(__attribute__((address_space(1))) uint16_t*)(id16 + sl) = (uint16_t)id;
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (t == 0) {
        gst32(aux + BH_W_D, D);
        gst32(aux + BH_W_BAD, 0u);
        gst32(aux + BH_W_MAGIC, BH_MAGIC);
    }
}

// Boyer-Moore majority candidate (row index or SEL_EMPTY) with an exact count of its key.
template <class Ops>
__device__ uint32_t majority_count(Ops ops, uint64_t N, uint32_t* s_a /* 2*WG + 4 words */) {
    const int t = threadIdx.x;
    // (candidates are always rows of the page — row 0 while a thread has none — so that a comparison the
    // compiler hoists above the count checks cannot read outside the column)
    uint32_t cand = 0, cnt = 0;
    if constexpr (has_key64<Ops>::value) {
        // the candidate's key stays in a register and the rows' keys stream in, eight loads in flight (with eq() every
        // step is a dependent load of the candidate's key: 0.24 ms per 64 Ki-row page)
        uint64_t ck = 0;
        for (uint64_t base = 0; base < N; base += (uint64_t)WG * 8) {
            uint64_t kv[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const uint64_t i = base + (uint64_t)u * WG + t;
                kv[u] = ops.key64(i < N ? i : 0);
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const uint64_t i = base + (uint64_t)u * WG + t;
                if (i >= N) continue;
                if (cnt == 0) {
                    cand = (uint32_t)i;
                    ck = kv[u];
                    cnt = 1;
                } else if (kv[u] == ck) {
                    cnt++;
                } else {
                    cnt--;
                }
            }
        }
    } else {
        for (uint64_t base = 0; base < N; base += WG) {  // (uniform trip count, rows guarded inside)
            const uint64_t i = base + (uint64_t)t;
            if (i >= N) continue;
            if (cnt == 0) {
                cand = (uint32_t)i;
                cnt = 1;
            } else if (ops.eq(cand, i)) {
                cnt++;
            } else {
                cnt--;
            }
        }
    }
    s_a[t] = cand;
    s_a[WG + t] = cnt;
    __syncthreads();
    for (int stride = WG / 2; stride > 0; stride >>= 1) {
        if (t < stride) {
            const uint32_t c0 = s_a[t], n0 = s_a[WG + t], c1 = s_a[t + stride], n1 = s_a[WG + t + stride];
            uint32_t c = c0, n = n0;
            if (n1) {
                if (n0 == 0) {
                    c = c1;
                    n = n1;
                } else if (ops.eq(c0, c1)) {
                    n = n0 + n1;
                } else if (n1 > n0) {
                    c = c1;
                    n = n1 - n0;
                } else {
                    n = n0 - n1;
                }
            }
            s_a[t] = c;
            s_a[WG + t] = n;
        }
        __syncthreads();
    }
    const uint32_t c = s_a[WG] ? s_a[0] : SEL_EMPTY;
    __syncthreads();
    if (c == SEL_EMPTY) return 0;
    uint32_t mine = 0;
    if constexpr (has_key64<Ops>::value) {
        const uint64_t ck = ops.key64(c);
        for (uint64_t base = 0; base < N; base += (uint64_t)WG * 8) {
            uint64_t kv[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const uint64_t i = base + (uint64_t)u * WG + t;
                kv[u] = ops.key64(i < N ? i : c);
            }
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (base + (uint64_t)u * WG + t < N && kv[u] == ck) mine++;
        }
    } else {
        for (uint64_t base = 0; base < N; base += WG) {
            const uint64_t i = base + (uint64_t)t;
            if (i < N && ops.eq(c, i)) mine++;
        }
    }
    return wg_sum32(mine, s_a + 2 * WG);
}

// choose_compressor for primitives (integer/mod.rs:231-308, double/mod.rs:231-307)
// thread-local partial statistics of one page, produced by a streaming pass (choose_prim's lane = row
// pass, or the row-segment pass of the fused select + RLE kernel) and reduced by decide_prim
template <int W>
struct PrimPartials {
    uint32_t f_neq0, f_unsorted, f_neg, nulls;
    Val<W> tmax;
    uint64_t vote_k;
    uint32_t vote_n;
};

// numbers decide_prim would otherwise obtain by walking the page with one workgroup (long pages: sb_select_big.h)
struct PrimCounts {
    bool have_uq, have_mc;
    uint32_t uq, mc;   // distinct keys (or any value above Dict's limit), rows equal to the vote's candidate
};

// gen_stats' reductions + choose_compressor (integer/mod.rs:231-308, double/mod.rs:231-307): the part of
// the selector that does not depend on how the page was streamed
template <int W, class GetVal>
__device__ uint32_t decide_prim(GetVal getv, const ValidView& vv, uint64_t N, uint32_t nk, const SelectOpts& o,
                                const SelScratch& sc, const PrimPartials<W>& pp, bool want_set, bool want_vote, uint32_t s_kcnt,
                                uint32_t s_ksent, const SamplePre<W>& pre_rle, const SamplePre<W>& pre_bp,
                                const SamplePre<W>& pre_dbp, const SamplePre<W>& pre_patas, bool prefetched = true,
                                const PrimCounts* pc = nullptr) {
    auto valid = [&](uint64_t i) { return vv.get(i); };
    (void)valid;
    const int t = threadIdx.x;
    auto forbidden = [&](uint32_t c) { return (o.forbidden >> c) & 1u; };
    const bool is_float = nk >= NK_F32;
    auto key = [&](uint64_t i) { return stat_key<W>(getv(i), nk); };
    uint32_t* s4 = sc.s_misc + 2 * WG;
    constexpr bool SMALL = W <= 8;
    constexpr uint32_t KSLOTS = SEL_LDS_SLOTS / 2, KCAP = KSLOTS / 2;
    auto k64 = [&](const Val<W>& k) {
        uint64_t x = 0;
        if constexpr (SMALL) __builtin_memcpy(&x, &k, W);
        return x;
    };
    const uint32_t f_neq0 = pp.f_neq0, f_unsorted = pp.f_unsorted, f_neg = pp.f_neg, nulls = pp.nulls;
    const Val<W> tmax = pp.tmax;
    const uint64_t vote_k = pp.vote_k;
    const uint32_t vote_n = pp.vote_n;
    STL(1);
    const uint32_t flags = wg_or32(f_neq0 | (f_unsorted << 1) | (f_neg << 2), s4);
    const uint32_t null_count = wg_sum32(nulls, s4);
    const bool all_equal = !(flags & 1);
    bool is_sorted = !(flags & 2);
    const bool any_neg = flags & 4;
    if (!is_float && W == 4 && nk == NK_SIGNED && (int32_t)as_i64<W>(getv(0), nk) < 0) is_sorted = false;  // vs last_value = 0
    // exact distinct count from the LDS set (valid unless it overflowed)
    uint32_t set_unique = 0;
    bool set_ok = false;
    if (want_set) {
        set_unique = s_kcnt + s_ksent;
        set_ok = s_kcnt <= KCAP;
    }
    // merged Boyer-Moore vote: (key, margin); a key with >= 90 % of the rows leaves margin >= 0.8 N
    uint64_t maj_k = 0;
    uint32_t maj_n = 0;
    if (want_vote) {
        unsigned long long* vk = (unsigned long long*)sc.sample_mem;  // 256 * 8 B, free until the samples are drawn
        uint32_t* vn = sc.s_misc;
        vk[t] = vote_k;
        vn[t] = vote_n;
        __syncthreads();
        for (int stride = WG / 2; stride > 0; stride >>= 1) {
            if (t < stride) {
                const unsigned long long c0 = vk[t], c1 = vk[t + stride];
                const uint32_t n0 = vn[t], n1 = vn[t + stride];
                if (n1) {
                    if (n0 == 0) {
                        vk[t] = c1;
                        vn[t] = n1;
                    } else if (c0 == c1) {
                        vn[t] = n0 + n1;
                    } else if (n1 > n0) {
                        vk[t] = c1;
                        vn[t] = n1 - n0;
                    } else {
                        vn[t] = n0 - n1;
                    }
                }
            }
            __syncthreads();
        }
        maj_k = vk[0];
        maj_n = vn[0];
        __syncthreads();
    }
    // typed maximum (only Freq for integers looks at it: max.as_i64() >= 256, freq.rs:146)
    int64_t max_i64 = 0;
    if (!is_float && !forbidden(SB_CODEC_FREQ)) {
        Val<W>* red = (Val<W>*)sc.sample_mem;  // 256 * W bytes <= 8 KB, free until the samples are drawn
        red[t] = tmax;
        __syncthreads();
        for (int stride = WG / 2; stride > 0; stride >>= 1) {
            if (t < stride && int_lt<W>(red[t], red[t + stride], nk)) red[t] = red[t + stride];
            __syncthreads();
        }
        max_i64 = as_i64<W>(red[0], nk);
        __syncthreads();
    }
    STL(2);
    const double tuple_count = (double)N;
    const double total_bytes = (double)(N * W);
    double max_ratio = o.ratio;
    uint32_t result = o.default_codec;
    static const uint8_t INT_ORDER[6] = {SB_CODEC_ONEVALUE, SB_CODEC_FREQ, SB_CODEC_DICT,
                                         SB_CODEC_RLE, SB_CODEC_BITPACKING, SB_CODEC_DELTA_BITPACKING};
    static const uint8_t DBL_ORDER[5] = {SB_CODEC_ONEVALUE, SB_CODEC_FREQ, SB_CODEC_DICT, SB_CODEC_PATAS, SB_CODEC_RLE};
    const int norder = is_float ? 5 : 6;
    KeyOpsPrim<W, decltype(key)> kops{key};
    Sample<W> smp;
    smp.val = (Val<W>*)sc.sample_mem;
    smp.valid = sc.sample_mem + SAMPLE_CAP * W;
    for (int oi = 0; oi < norder; oi++) {
        const uint32_t c = is_float ? DBL_ORDER[oi] : INT_ORDER[oi];
        if (forbidden(c)) continue;
        double r = 0.0;
        switch (c) {
            case SB_CODEC_ONEVALUE:  // one_value.rs:53-59
                r = all_equal ? tuple_count : 0.0;
                break;
            case SB_CODEC_FREQ: {  // freq.rs:129-151
                if (all_equal) break;
                if ((double)null_count / tuple_count >= 0.9) {
                    r = (double)(N - 1);
                    break;
                }
                uint32_t mc;
                if constexpr (SMALL) {
                    mc = 0;
                    if ((double)maj_n + 1.0 >= 0.8 * tuple_count) {  // only then can a key hold >= 90 % of the rows
                        if (pc && pc->have_mc) {
                            mc = pc->mc;
                        } else {
                            uint32_t mine = 0;
                            for (uint64_t i = t; i < N; i += WG) mine += k64(key(i)) == maj_k ? 1 : 0;
                            mc = wg_sum32(mine, s4);
                        }
                    }
                } else {
                    mc = majority_count(kops, N, sc.s_misc);
                }
                const bool big = is_float ? true : (max_i64 >= 256);
                if ((double)mc / tuple_count >= 0.9 && big) r = (double)(N - 1);
                break;
            }
            case SB_CODEC_DICT: {  // dict.rs:109-120
                if (N < 3) break;
                const uint32_t limit = (uint32_t)((N - 1) / 3);  // largest unique with unique*3 < N
                const uint32_t uq = all_equal ? 1u : set_ok ? set_unique : (pc && pc->have_uq) ? pc->uq : distinct_count(kops, N, limit, sc, nullptr);
                if ((uint64_t)uq * 3 >= N) break;
                uint64_t after = (uint64_t)uq * W + N * (uint64_t)(bits_needed(uq) / 8);
                after += N * 2 / 128;
                r = total_bytes / (double)after;
                break;
            }
            case SB_CODEC_RLE: {  // rle.rs:58-60
                if (prefetched)
                    commit_sample<W>(pre_rle, N, smp);
                else
                    load_sample<W>(getv, valid, N, o.seed, o.depth, SB_CODEC_RLE, smp);
                uint32_t runs;
                if (nk == NK_F32)
                    runs = sample_rle_runs<W, (W == 4 ? 1 : 0)>(smp, s4);
                else if (nk == NK_F64)
                    runs = sample_rle_runs<W, (W == 8 ? 2 : 0)>(smp, s4);
                else
                    runs = sample_rle_runs<W, 0>(smp, s4);
                r = (double)((uint64_t)smp.n * W) / (double)((uint64_t)runs * (4 + W));
                __syncthreads();
                break;
            }
            case SB_CODEC_BITPACKING:  // bp.rs:92-100
            case SB_CODEC_DELTA_BITPACKING: {  // delta_bp.rs:97-109
                if constexpr (W == 4) {
                    if (any_neg || N % 128 != 0) break;
                    if (c == SB_CODEC_DELTA_BITPACKING && (!is_sorted || null_count > 0)) break;
                    if (prefetched)
                        commit_sample<4>(c == SB_CODEC_BITPACKING ? pre_bp : pre_dbp, N, smp);
                    else
                        load_sample<4>(getv, valid, N, o.seed, o.depth, c, smp);
                    const uint32_t size = sample_bp_size(smp, s4, sc.s_misc);
                    r = (double)((uint64_t)smp.n * 4) / (double)size;
                    if (c == SB_CODEC_DELTA_BITPACKING) r *= 1.5;
                    __syncthreads();
                }
                break;
            }
            case SB_CODEC_PATAS: {  // patas.rs:139-141
                if constexpr (W == 4 || W == 8) {
                    // every value costs at least 2 bytes: the ratio stays below W / 2, so the trial cannot
                    // change the outcome once another codec is at or above that (same choice, no work)
                    if (max_ratio >= (double)W / 2) break;
                    if (prefetched)
                        commit_sample<W>(pre_patas, N, smp);
                    else
                        load_sample<W>(getv, valid, N, o.seed, o.depth, SB_CODEC_PATAS, smp);
                    const uint32_t size = sample_patas_size<W>(smp, s4);
                    r = (double)((uint64_t)smp.n * W) / (double)size;
                    __syncthreads();
                }
                break;
            }
        }
        STL(3 + oi);
        if (r > max_ratio) {
            max_ratio = r;
            result = c;
            if (r == tuple_count) break;
        }
    }
    STL(10);
    return result;
}

template <int W, class GetVal>
__device__ uint32_t choose_prim(GetVal getv, const ValidView& vv, uint64_t N, uint32_t nk, const SelectOpts& o,
                                const SelScratch& sc) {
    auto valid = [&](uint64_t i) { return vv.get(i); };
    const int t = threadIdx.x;
    auto forbidden = [&](uint32_t c) { return (o.forbidden >> c) & 1u; };
    if (o.force >= 0 && !forbidden((uint32_t)o.force)) return (uint32_t)o.force;
    if (!o.has_ratio || N == 0) return o.default_codec;
    const bool is_float = nk >= NK_F32;
    auto key = [&](uint64_t i) { return stat_key<W>(getv(i), nk); };
    // ---- one streaming pass: flags, null count, typed max, Boyer-Moore vote, and (W <= 8) an LDS
    // hash set of the canonical keys for the exact distinct count Dict needs
    STL(0);
    // sample rows of the trials that can run, fetched now (see SamplePre)
    SamplePre<W> pre_rle, pre_bp, pre_dbp, pre_patas;
    if (N) {
        if (!forbidden(SB_CODEC_RLE)) pre_rle = prefetch_sample<W>(getv, valid, N, o.seed, o.depth, SB_CODEC_RLE);
        if constexpr (W == 4) {
            if (!is_float && !forbidden(SB_CODEC_BITPACKING) && N % 128 == 0)
                pre_bp = prefetch_sample<W>(getv, valid, N, o.seed, o.depth, SB_CODEC_BITPACKING);
            if (!is_float && !forbidden(SB_CODEC_DELTA_BITPACKING) && N % 128 == 0)
                pre_dbp = prefetch_sample<W>(getv, valid, N, o.seed, o.depth, SB_CODEC_DELTA_BITPACKING);
        }
        if constexpr (W == 4 || W == 8) {
            if (is_float && !forbidden(SB_CODEC_PATAS)) pre_patas = prefetch_sample<W>(getv, valid, N, o.seed, o.depth, SB_CODEC_PATAS);
        }
    }
    const Val<W> k0 = key(0);
    uint32_t f_neq0 = 0, f_unsorted = 0, f_neg = 0, nulls = 0;
    Val<W> tmax = getv(0);
    constexpr bool SMALL = W <= 8;
    constexpr uint64_t SENT = ~0ull;
    constexpr uint32_t KSLOTS = SEL_LDS_SLOTS / 2, KCAP = KSLOTS / 2;  // 4096 u64 slots, 2048 keys
    unsigned long long* kset = (unsigned long long*)sc.lds_tab;
    __shared__ uint32_t s_kcnt, s_ksent;
    const bool want_set = SMALL && !forbidden(SB_CODEC_DICT) && N >= 3;
    const bool want_vote = SMALL && !forbidden(SB_CODEC_FREQ);
    if (want_set) {
        for (uint32_t i = t; i < KSLOTS; i += WG) kset[i] = SENT;
        if (t == 0) {
            s_kcnt = 0;
            s_ksent = 0;
        }
        __syncthreads();
    }
    auto k64 = [&](const Val<W>& k) {
        uint64_t x = 0;
        if constexpr (SMALL) __builtin_memcpy(&x, &k, W);
        return x;
    };
    uint64_t vote_k = 0;
    uint32_t vote_n = 0;
    // The page is walked in chunks of CH rows.  The chunk's validity bits are staged in LDS first
    // (the only place that touches the bitmap's byte/bit offsets), so the row loop is branch-free:
    // U independent value loads per thread are issued back to back, then processed.
    constexpr int U = W <= 8 ? 8 : 2;
    constexpr uint32_t CH = 8192;
    uint32_t* s_vb = (uint32_t*)sc.sample_mem;  // CH/32 words (1 KB): free until the samples are drawn
    auto vword = [&](uint64_t cb) {  // validity word t of the chunk at cb (threads < CH/32): load only
        const uint32_t cn = (uint32_t)min((uint64_t)CH, N - cb);
        const uint32_t bit0 = (uint32_t)t * 32;
        const bool mine = t < (int)(CH / 32) && bit0 < cn;
        VWord r = vword_issue(vv.bits, vv.off + cb + bit0, vv.off + N, mine ? min(32u, cn - bit0) : 0u);
        if (!mine) r.mask = 0;
        return r;
    };
    VWord wv_next = vword(0);
    // Distinct keys (W <= 8): rows are not probed one by one.  Lane = row, so a run of equal values
    // sits in neighbouring lanes: only the lanes whose raw bits differ from the lane before them
    // (ballot) append their key to a small per-wave LDS buffer, and the buffer is probed into the
    // hash set 64 keys at a time when it fills.  For run-heavy pages that is one probe pass per
    // ~1000 rows instead of one per 64; the comparison with row 0's key (all_equal) rides along.
    using KE = typename std::conditional<(W == 8), unsigned long long, uint32_t>::type;
    constexpr uint32_t CBUF = 96;  // entries per wave (a wave appends at most 64 at a time)
    const int lane = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    KE* cbuf = (KE*)(sc.sample_mem + CH / 8) + wv * CBUF;
    // run length of every buffered key inside its 64-row group: the weight of the Freq vote
    uint8_t* cwgt = sc.sample_mem + CH / 8 + 4 * CBUF * sizeof(KE) + wv * CBUF;
    uint32_t ccount = 0;
    auto flush = [&]() {
        for (uint32_t base = 0; base < ccount; base += 64) {
            const bool act = base + lane < ccount;
            uint64_t raw = act ? (uint64_t)cbuf[base + lane] : 0;
            Val<W> rv;
            if constexpr (SMALL) __builtin_memcpy(&rv, &raw, W);
            const Val<W> kk = stat_key<W>(rv, nk);
            if (act && !bits_eq<W>(kk, k0)) f_neq0 = 1;
            if (want_vote && act) {  // Boyer-Moore with run-length weights (== feeding the run's rows one by one)
                const uint64_t x = k64(kk);
                const uint32_t wgt = cwgt[base + lane];
                if (vote_k == x) {
                    vote_n += wgt;
                } else if (vote_n >= wgt) {
                    vote_n -= wgt;
                } else {
                    vote_k = x;
                    vote_n = wgt - vote_n;
                }
            }
            if (want_set && act && s_kcnt <= KCAP) {
                const uint64_t x = k64(kk);
                if (x == SENT) {
                    s_ksent = 1;
                } else {
                    uint32_t h = (((uint32_t)x ^ (uint32_t)(x >> 32) * 0x85EBCA6Bu) * 0x9E3779B1u >> 15) & (KSLOTS - 1);
                    for (;;) {
                        unsigned long long cur = kset[h];
                        if (cur == x) break;
                        if (cur == SENT) {
                            const unsigned long long old = atomicCAS(&kset[h], (unsigned long long)SENT, (unsigned long long)x);
                            if (old == SENT) {
                                atomicAdd(&s_kcnt, 1u);
                                break;
                            }
                            if (old == x) break;
                        }
                        h = (h + 1) & (KSLOTS - 1);
                    }
                }
            }
        }
        ccount = 0;
    };
    for (uint64_t cb = 0; cb < N; cb += CH) {
     const uint32_t cn = (uint32_t)min((uint64_t)CH, N - cb);
     __syncthreads();
     if (t < (int)(CH / 32)) s_vb[t] = wv_next.word();
     __syncthreads();
     if (cb + CH < N) wv_next = vword(cb + CH);  // in flight while this chunk is processed
     for (uint64_t ib0 = cb; ib0 < cb + cn; ib0 += (uint64_t)WG * U) {  // uniform trip count: wave-wide ops inside
      const uint64_t ib = ib0 + t;
      Val<W> vbuf[U], pbuf[U];
      bool okb[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
          const uint64_t i = ib + (uint64_t)u * WG;
          const uint64_t ic = i < N ? i : N - 1;
          vbuf[u] = getv(ic);
          if (!is_float && W == 4) pbuf[u] = getv(ic ? ic - 1 : 0);
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
          const uint32_t r = (uint32_t)(ib - cb) + (uint32_t)u * WG;
          okb[u] = r < cn ? (s_vb[r >> 5] >> (r & 31)) & 1 : false;
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        const uint64_t i = ib + (uint64_t)u * WG;
        const bool in = i < N;
        const Val<W> v = vbuf[u];
        if (in && !okb[u]) nulls++;
        if (!is_float && in) {
            if (int_lt<W>(tmax, v, nk)) tmax = v;
            if (W == 4 && nk == NK_SIGNED && (int32_t)as_i64<W>(v, nk) < 0) f_neg = 1;
            if (W == 4 && i > 0 && int_lt<W>(v, pbuf[u], nk)) f_unsorted = 1;
        }
        if constexpr (SMALL) {
            const KE raw = (KE)k64(v);
            const KE prev = (KE)__shfl_up(raw, 1, 64);
            const bool bnd = in && (lane == 0 || raw != prev);
            const uint64_t bm = __ballot(bnd);
            const uint32_t nb = (uint32_t)__popcll(bm);
            const uint32_t nin = (uint32_t)__popcll(__ballot(in));  // rows in range are a lane prefix
            if (ccount + nb > CBUF) flush();
            if (bnd) {
                const uint32_t slot = ccount + mbcnt64(bm);
                cbuf[slot] = raw;
                if (want_vote) {  // rows up to the next boundary (or the end of the rows in range)
                    const uint64_t later = lane == 63 ? 0ull : bm & (~0ull << (lane + 1));
                    const uint32_t end = later ? (uint32_t)(__ffsll((long long)later) - 1) : nin;
                    cwgt[slot] = (uint8_t)(end - (uint32_t)lane);
                }
            }
            ccount += nb;
        } else {
            if (in && !bits_eq<W>(stat_key<W>(v, nk), k0)) f_neq0 = 1;
        }
      }
     }
    }
    if constexpr (SMALL) flush();
    __syncthreads();  // every wave's keys are in the set before its size is read
    PrimPartials<W> pp{f_neq0, f_unsorted, f_neg, nulls, tmax, vote_k, vote_n};
    return decide_prim<W>(getv, vv, N, nk, o, sc, pp, want_set, want_vote, want_set ? s_kcnt : 0u, want_set ? s_ksent : 0u, pre_rle,
                          pre_bp, pre_dbp, pre_patas);
}

// choose_compressor for booleans (boolean/mod.rs:194-238) with gen_stats (:151-192)
__device__ uint32_t choose_bool(const uint8_t* bits, uint64_t boff, const ValidView& vv, uint64_t N, const SelectOpts& o,
                                const SelScratch& sc) {
    auto forbidden = [&](uint32_t c) { return (o.forbidden >> c) & 1u; };
    if (o.force >= 0 && !forbidden((uint32_t)o.force)) return (uint32_t)o.force;
    if (!o.has_ratio || N == 0) return o.default_codec;
    uint32_t* s4 = sc.s_misc + 2 * WG;
    // valid trues / falses, 32 rows per step (a bit per step was 10 ms for a 12 M-row page)
    uint32_t nt = 0, nf = 0;
    const uint64_t nwords = (N + 31) / 32;
    for (uint64_t g = threadIdx.x; g < nwords; g += WG) {
        const uint64_t left = N - g * 32;
        const uint32_t mask = left >= 32 ? 0xFFFFFFFFu : (1u << left) - 1;
        const uint32_t v = bits32(bits, boff + g * 32, boff + N);
        const uint32_t m = vv.bits ? bits32(vv.bits, vv.off + g * 32, vv.off + N) & mask : mask;
        nt += (uint32_t)__popc(v & m);
        nf += (uint32_t)__popc(~v & m);
    }
    const uint32_t true_count = wg_sum32(nt, s4), false_count = wg_sum32(nf, s4);
    double max_ratio = o.ratio;
    uint32_t result = o.default_codec;
    if (!forbidden(SB_CODEC_ONEVALUE)) {  // one_value.rs:36-42
        const double r = (true_count == 0 || false_count == 0) ? (double)N : 0.0;
        if (r > max_ratio) {
            max_ratio = r;
            result = SB_CODEC_ONEVALUE;
            if (r == (double)N) return result;
        }
    }
    if (!forbidden(SB_CODEC_RLE)) {  // boolean/rle.rs:61-63 -> compress_sample_ratio (:240-278)
        Sample<1> smp;
        smp.val = (Val<1>*)sc.sample_mem;
        smp.valid = sc.sample_mem + SAMPLE_CAP;
        auto getv = [&](uint64_t i) {
            Val<1> v;
            v.x = bit_at(bits, boff + i) ? 1 : 0;
            return v;
        };
        auto valid = [&](uint64_t i) { return vv.get(i); };
        // total_bytes = values().len() / 8 of the (sample) array; sizes: 5 bytes per run
        double r;
        if (N / SAMPLE_COUNT <= SAMPLE_SIZE) {
            smp.whole = true;
            // whole array: count runs over all N rows (may exceed the LDS sample): stream it
            uint32_t cnt = 0;
            for (uint64_t k = threadIdx.x; k < N; k += WG) {
                if (!vv.get(k)) continue;
                int64_t p = (int64_t)k - 1;
                while (p >= 0 && !vv.get((uint64_t)p)) p--;
                if (p >= 0 && bit_at(bits, boff + (uint64_t)p) != bit_at(bits, boff + k)) cnt++;
            }
            const uint32_t runs = wg_sum32(cnt, s4) + 1;
            r = (double)(N / 8) / (double)((uint64_t)runs * 5);
        } else {
            load_sample<1>(getv, valid, N, o.seed, 0, SB_CODEC_RLE, smp);
            const uint32_t runs = sample_rle_runs<1, 0>(smp, s4);
            r = (double)(SAMPLE_ROWS / 8) / (double)((uint64_t)runs * 5);
            __syncthreads();
        }
        if (r > max_ratio) {
            max_ratio = r;
            result = SB_CODEC_RLE;
        }
    }
    return result;
}

// row-index hash-set operations over all rows of a binary page (gen_stats, binary/mod.rs:265-291)
template <class O>
struct KeyOpsBin {
    BinKeys<O> k;
    __device__ __forceinline__ uint32_t hash(uint64_t i) const { return k.hash(i); }
    __device__ __forceinline__ bool eq(uint64_t a, uint64_t b) const { return k.eq(a, b); }
    __device__ __forceinline__ uint32_t weight(uint64_t i) const { return (uint32_t)(k.beg(i + 1) - k.beg(i)) + 8; }
};

// choose_compressor for binary (binary/mod.rs:293-348)
template <class O, class Ops>
__device__ uint32_t choose_bin_impl(const Ops& ops, const BinKeys<O>& bk, uint64_t N, uint64_t values_len_total, const SelectOpts& o,
                                    const SelScratch& sc) {
    auto forbidden = [&](uint32_t c) { return (o.forbidden >> c) & 1u; };
    uint32_t* s4 = sc.s_misc + 2 * WG;
    uint32_t f_neq0 = 0, nulls = 0;
    if constexpr (has_key64<Ops>::value) {   // keys streamed eight loads at a time; nulls from the validity words
        const uint64_t k0 = ops.key64(0);
        for (uint64_t base = threadIdx.x; base < N; base += (uint64_t)WG * 8) {
            uint64_t kv[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const uint64_t i = base + (uint64_t)u * WG;
                kv[u] = ops.key64(i < N ? i : 0);
            }
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (kv[u] != k0) f_neq0 = 1;
        }
        if (bk.vv.bits)
            for (uint64_t w = threadIdx.x; w * 32 < N; w += WG) {
                const uint32_t nb = (uint32_t)min((uint64_t)32, N - w * 32);
                const uint32_t word = bits32(bk.vv.bits, bk.vv.off + w * 32, bk.vv.off + N) & (nb >= 32 ? 0xFFFFFFFFu : (1u << nb) - 1);
                nulls += nb - (uint32_t)__popc(word);
            }
    } else {
        for (uint64_t i = threadIdx.x; i < N; i += WG) {
            if (!ops.eq(0, i)) f_neq0 = 1;
            if (!bk.vv.get(i)) nulls++;
        }
    }
    const bool all_equal = !wg_or32(f_neq0, s4);
    const uint32_t null_count = wg_sum32(nulls, s4);
    STL(42);
    const double tuple_count = (double)N;
    const double total_bytes = (double)(values_len_total + (N + 1) * sizeof(O));
    double max_ratio = o.ratio;
    uint32_t result = o.default_codec;
    static const uint8_t ORDER[3] = {SB_CODEC_ONEVALUE, SB_CODEC_FREQ, SB_CODEC_DICT};
    for (int oi = 0; oi < 3; oi++) {
        const uint32_t c = ORDER[oi];
        if (forbidden(c)) continue;
        double r = 0.0;
        if (c == SB_CODEC_ONEVALUE) {  // binary/one_value.rs:42-48
            r = all_equal ? tuple_count : 0.0;
        } else if (c == SB_CODEC_FREQ) {  // binary/freq.rs:147-169
            if (!all_equal) {
                if ((double)null_count / tuple_count >= 0.9) {
                    r = (double)(N - 1);
                } else {
                    const uint32_t mc = majority_count(ops, N, sc.s_misc);
                    if ((double)mc / tuple_count >= 0.9) r = (double)(N - 1);
                }
            }
            STL(43);
        } else {  // binary/dict.rs:43-53
            if (N >= 3) {
                const uint32_t limit = (uint32_t)((N - 1) / 3);
                uint64_t tus = 0;
                uint32_t uq = SEL_TAG_FAIL;
                if constexpr (has_key64<Ops>::value) {
                    if (sc.slot16 && sc.tagged && sc.lds_slots >= BH_SLOTS + 4096 && N <= 65536) {
                        uq = distinct_count_tags(ops, N, limit, sc, &tus, sc.gtab);
                        if (threadIdx.x == 0) *sc.tagged = uq != SEL_TAG_FAIL && uq <= limit ? 1u : 0u;
                        __syncthreads();
                    }
                }
                if (uq == SEL_TAG_FAIL) uq = distinct_count(ops, N, limit, sc, &tus);
                STL(44);
                if ((uint64_t)uq * 3 < N) {
                    uint64_t after = tus + N * (uint64_t)(bits_needed(uq) / 8);
                    after += N * 2 / 128;
                    r = total_bytes / (double)after;
                }
            }
        }
        if (r > max_ratio) {
            max_ratio = r;
            result = c;
            if (r == tuple_count) break;
        }
    }
    return result;
}
// h64 (one u64 per row, or null): when present the statistics run over hashed rows (bin_hash_rows)
template <class O>
__device__ uint32_t choose_bin(const BinKeys<O>& bk, uint64_t N, uint64_t values_len_total, const SelectOpts& o,
                               const SelScratch& sc, uint64_t* h64 = nullptr, uint64_t values_len = 0, bool pre_hashed = false) {
    auto forbidden = [&](uint32_t c) { return (o.forbidden >> c) & 1u; };
    if (o.force >= 0 && !forbidden((uint32_t)o.force)) return (uint32_t)o.force;
    if (!o.has_ratio || N == 0) return o.default_codec;
    if (h64) {
        STL(40);
        if (pre_hashed) {   // k_enc_bin_hash filled h64 for this launch
        } else if (sc.lds_slots >= 4096)   // (the table is not in use yet: its LDS stages the value bytes)
            bin_hash_rows_staged<O>(bk, N, values_len, h64, sc.lds_tab, sc.lds_slots * 4);
        else
            bin_hash_rows<O>(bk, N, values_len, h64);
        STL(41);
        const uint32_t r = choose_bin_impl<O>(KeyOpsBinHashed<O>{bk, h64}, bk, N, values_len_total, o, sc);
        STL(45);
        return r;
    }
    return choose_bin_impl<O>(KeyOpsBin<O>{bk}, bk, N, values_len_total, o, sc);
}

// Blocks go to XCD blockIdx % 8.  Pages at the same position of their column (every column's short last page,
// whose u32 indices fall to LZ4 and take 100x longer than a bit-packed page) are a fixed stride apart: with a
// power-of-two page count per column they would all run on ONE XCD.  Rotate each group of 8 blocks by a hash
// of the group number.
__device__ __forceinline__ uint32_t spread_block(uint32_t b, uint32_t nblocks) {
    const uint32_t g = b >> 3;
    if (((g + 1) << 3) > nblocks) return b;
    return (g << 3) | ((b + ((g * 0x9E3779B1u) >> 29)) & 7);
}

// ---- k_enc_bin_page / k_enc_prim_dict (sb_bin_page.h): what the page kernels below need to know about them
#ifndef SB_BIN_BIG_ROWS
#define SB_BIN_BIG_ROWS (1ull << 18)
#endif
constexpr uint64_t BP_BIG_ROWS = SB_BIN_BIG_ROWS;   // (= BIN_BIG_ROWS of sb_select_big.h: such pages take the section-parallel path)
constexpr int BP_WG = 1024;
constexpr uint32_t BP_SLOTS = 32768;          // limit (<= 21 845) + one step of the whole workgroup (4096) never fills it
constexpr uint32_t BP_EMPTY = 0xFFFFFFFFu;
constexpr uint32_t BP_UNKEYED = 0x10000u;
constexpr uint64_t BP_MIN_ROWS = 512;
constexpr uint32_t BH_W_FUSED = 3;            // aux word: BP_DONE when this kernel decided the page in this call
constexpr uint32_t BP_DONE = 0x46555345u;
constexpr uint32_t BH_MAGIC2 = 0x48444232u;   // aux[BH_W_MAGIC]: a dictionary in the layout below was handed over
constexpr uint32_t BH_W_BPBYTES = 4, BH_W_ENTBYTES = 5, BH_W_ENTWORD = 6, BH_W_ICODEC = 7;   // aux words: the speculative bit-packed index block / the staged entries
constexpr uint32_t BP_W_FIRSTS = 16;          // firsts[<= (N - 1) / 3], then slot16[N]; idx in the last N words of the aux area
__host__ __device__ __forceinline__ uint64_t bp_w_slot16(uint64_t N) { return (BP_W_FIRSTS + N / 3 + 2 + 3) & ~3ull; }
__host__ __device__ __forceinline__ bool bp_fits(uint64_t N, uint64_t aux_bytes) {
    const uint64_t M = bh_table_slots(N);
    return N >= BP_MIN_ROWS && N <= 65536 && bp_w_slot16(N) + (N + 1) / 2 + 4 <= M && aux_bytes / 4 >= M + 3 * N;
}
// the pages this kernel takes: a pure function of the launch and the page (every later kernel asks the same question)
__device__ __forceinline__ bool bp_page_ok(const EncodeArgs& a, const EncPage& p, uint32_t page) {
    return a.bin_fused && a.use_counts && a.page_base == 0 && page < a.n_pages && a.has_ratio && p.codec == CODEC_ON_DEVICE &&
           p.h64_off != ~0ull && p.aux_bytes && bp_fits(p.rows, p.aux_bytes) && !(p.rows >= BP_BIG_ROWS && p.bigx_off) &&
           !(((a.forbidden | p.forb_extra) >> SB_CODEC_DICT) & 1);
}
__device__ __forceinline__ bool bp_page_done(const EncodeArgs& a, const EncPage& p, uint32_t page) {
    return bp_page_ok(a, p, page) && gld32((const uint32_t*)(a.scratch + p.aux_off) + BH_W_FUSED) == BP_DONE;
}
constexpr uint32_t PD_SLOTS = BP_SLOTS / 2, PD_CAP = 10240;
constexpr uint32_t PD_DONE = 0x50444943u;   // aux[BH_W_FUSED]
__device__ __forceinline__ bool pd_page_ok(const EncodeArgs& a, const EncPage& p, uint32_t page, const EncCol& c, uint32_t W) {
    return a.bin_fused && a.use_counts && a.page_base == 0 && page < a.n_pages && a.has_ratio && p.codec == CODEC_ON_DEVICE && p.icodec < 0 &&
           p.aux_bytes && bp_fits(p.rows, p.aux_bytes) && !(p.rows >= BP_BIG_ROWS && p.bigx_off) && c.fkind == 0 && c.width == W &&
           c.ptype != SB_TYPE_BOOLEAN && c.ptype != SB_TYPE_BINARY && c.ptype != SB_TYPE_LARGE_BINARY && c.ptype != SB_TYPE_NULL;
}
__device__ __forceinline__ bool pd_page_done(const EncodeArgs& a, const EncPage& p, uint32_t page, const EncCol& c, uint32_t W) {
    if (!pd_page_ok(a, p, page, c, W)) return false;
    const uint32_t* aux = (const uint32_t*)(a.scratch + p.aux_off);
    return gld32(aux + BH_W_FUSED) == PD_DONE && gld32(aux + BH_W_MAGIC) == BH_MAGIC2;
}

// ------------------------------------------------------------------------------ page kernels
struct PageCtx {
    const EncCol* c;
    const EncPage* p;
    uint8_t* slot;
    ValidView vv;  // page-relative validity
};

__device__ __forceinline__ uint8_t* page_slot(const EncodeArgs& a, const EncCol& c, const EncPage& p) {
    if (p.direct) return c.out + p.direct_off + p.head_bytes;
    uint64_t off = p.slot_off;
    if (c.ptype == SB_TYPE_BINARY || c.ptype == SB_TYPE_LARGE_BINARY) {
        // value-dependent share of the slot space: bytes of this column's values before the page
        uint64_t o0, or0;
        if (c.ptype == SB_TYPE_BINARY) {
            o0 = ldu32(c.offsets);
            or0 = ldu32(c.offsets + p.row0 * 4);
        } else {
            o0 = ldu64(c.offsets);
            or0 = ldu64(c.offsets + p.row0 * 8);
        }
        const uint64_t before = or0 - o0;
        off += before + before / 128;
    }
    return a.scratch + off;
}

template <int W, int CODEC>
__device__ uint64_t emit_prim_page(const EncodeArgs& a, const EncCol& c, const EncPage& p, uint32_t page, uint8_t* blk,
                                   const ValidView& vv, uint32_t* sA, uint32_t* sB, uint32_t* sC, uint32_t* s_w) {
    const uint64_t N = p.rows;
    const uint8_t* vals = c.values + p.row0 * W;
    auto getv = [=](uint64_t i) { return ld_val<W>(vals + i * W); };
    uint64_t body = 0;
    if constexpr (CODEC == SB_CODEC_RLE) {
        if (c.fkind == 1 && W == 4)
            body = enc_rle_rows<W, (W == 4 ? 1 : 0)>(getv, vv, N, blk + 9, sA);
        else if (c.fkind == 2 && W == 8)
            body = enc_rle_rows<W, (W == 8 ? 2 : 0)>(getv, vv, N, blk + 9, sA);
        else
            body = enc_rle_rows<W, 0>(getv, vv, N, blk + 9, sA);
    } else if constexpr (CODEC == SB_CODEC_ONEVALUE) {  // first valid value or T::default() (one_value.rs:63-75)
        __shared__ unsigned long long s_first;
        if (threadIdx.x == 0) s_first = ~0ull;
        __syncthreads();
        unsigned long long mine = ~0ull;
        for (uint64_t i = threadIdx.x; i < N && mine == ~0ull; i += WG)
            if (vv.get(i)) mine = i;
        if (mine != ~0ull) atomicMin(&s_first, mine);
        __syncthreads();
        if (threadIdx.x == 0) {
            Val<W> v = s_first == ~0ull ? val_zero<W>() : getv(s_first);
            __builtin_memcpy(blk + 9, &v, W);
        }
        body = W;
    } else if constexpr (CODEC == SB_CODEC_BITPACKING || CODEC == SB_CODEC_DELTA_BITPACKING) {
        if constexpr (W == 4) {
            if (N % 128 != 0 || c.ptype == SB_TYPE_FLOAT32) {
                if (threadIdx.x == 0) raise(a.status, SB_ERR_NYI, page, 520);
                return 0;
            }
            auto getu = [=](uint64_t i) { return ldu32(vals + i * 4); };
            body = enc_bp(getu, N, CODEC == SB_CODEC_DELTA_BITPACKING, blk + 9, sA, sB, s_w);
        } else {
            if (threadIdx.x == 0) raise(a.status, SB_ERR_NYI, page, 521);
            return 0;
        }
    } else if constexpr (CODEC == SB_CODEC_PATAS) {
        if constexpr (W == 4 || W == 8) {
            if (c.fkind == 0) {  // "Unknown compression codec Patas for integer"
                if (threadIdx.x == 0) raise(a.status, SB_ERR_OUT_OF_SPEC, page, 523);
                return 0;
            }
            body = enc_patas<W>(getv, N, blk + 9, sA, s_w, (uint8_t*)sB);
        } else {
            if (threadIdx.x == 0) raise(a.status, SB_ERR_OUT_OF_SPEC, page, 523);
            return 0;
        }
    } else if constexpr (CODEC == SB_CODEC_DICT) {
        PrimKeys<W> ko{vals, vv, c.fkind};
        uint32_t *idx, *firsts;
        uint32_t* aux = (uint32_t*)(a.scratch + p.aux_off);
        uint32_t D = DICT_FALLBACK;
        STL(50);
        uint32_t pre_ic = 0, pre_bp = 0;   // k_enc_prim_dict's index codec (+ 1) / bit-packed block
        bool pre = false;
        if constexpr (W <= 4) pre = pd_page_done(a, p, page, c, (uint32_t)W);
        if (pre) {   // dictionary, indices, index codec (and its bit-packed body) by k_enc_prim_dict
            idx = aux + bh_table_slots(N) + 2 * N;
            firsts = aux + BP_W_FIRSTS;
            D = gld32(aux + BH_W_D);
            pre_ic = gld32(aux + BH_W_ICODEC);
            pre_bp = gld32(aux + BH_W_BPBYTES);
        } else {
            if constexpr (W <= 8) D = dict_build_lds<W>(ko, N, aux, p.aux_bytes / 4, &idx, &firsts, sA);
            if (D == DICT_FALLBACK) D = dict_build(ko, N, aux, p.aux_bytes / 4, &idx, &firsts, sA, sB, s_w, a.status, page);
        }
        if (D == EMPTY) return 0;
        STL(51);
        int32_t ic = p.icodec >= 0 ? p.icodec : (int32_t)a.default_compression;
        if (p.icodec < 0 && a.has_ratio && pre_ic) {
            ic = (int32_t)pre_ic - 1;
        } else if (p.icodec < 0 && a.has_ratio) {  // nested compress_integer::<u32>: same selector, Dict forbidden (dict.rs:60-62)
            SelectOpts so{a.ratio, 1u, a.forbidden | p.forb_extra | (1u << SB_CODEC_DICT), a.default_compression, -1, p.seed, p.depth + 1};
            SelScratch sc{sA, sA + SEL_LDS_SLOTS, (uint8_t*)(sA + SEL_LDS_SLOTS + 2 * WG + 16), nullptr, 0};
            const uint32_t* ip = idx;
            ic = (int32_t)choose_prim<4>([=](uint64_t i) { Val<4> v; v.x = ip[i]; return v; },
                                         ValidView{nullptr, 0}, N, NK_UNSIGNED, so, sc);
            __syncthreads();
        }
        STL(52);
        if (ic == SB_CODEC_FREQ) {
            // u32 indices that are mostly one value (the column may not use Freq itself: integers with a maximum
            // below 256, freq.rs:146): the index block is a Freq block with its own nested exceptions block.  The
            // Freq kernels finish the page (k_enc_freq_prep -> k_enc_nested<4> -> k_enc_freq_finish); this kernel
            // leaves them the index array and the first rows of the dictionary entries.
            if (p.vaux_bytes < 32 || !p.vslot_off) {
                if (threadIdx.x == 0) raise(a.status, SB_ERR_NYI, page, 502);
                return 0;
            }
            if (threadIdx.x == 0) {
                unsigned long long* rec = (unsigned long long*)(a.scratch + p.vaux_off);
                rec[0] = (unsigned long long)(uintptr_t)idx;
                rec[1] = (unsigned long long)(uintptr_t)firsts;
                rec[2] = D;
                atomicAdd(a.freq_count, 1u);
            }
            return DICT_FREQ_PENDING;
        }
        // (HBM scratch for the three-wave LZ4 path: 2 N words of the aux area that are dead by now — the F / R arrays of
        // dict_build, or the tail of the area when dict_build_lds put the indices at its start)
        uint8_t* lz_tmp = nullptr;
        {
            uint64_t M = 64;
            while (M < 2 * N) M <<= 1;
            const uint64_t aux_words = p.aux_bytes / 4;
            if (aux_words >= M + 3 * N) lz_tmp = (uint8_t*)(idx == aux ? aux + (aux_words - 2 * N) : aux + M);
        }
        uint64_t ib;
        if (ic == SB_CODEC_BITPACKING && pre_bp) {   // the body stands there already: the block's header
            if (threadIdx.x == 0) put_hdr9(blk + 9, SB_CODEC_BITPACKING, pre_bp, (uint32_t)(N * 4));
            ib = 9 + (uint64_t)pre_bp;
        } else {
            ib = enc_u32_block(idx, N, ic, blk + 9, sA, sB, sC, s_w, a.status, page, a.flags, lz_tmp, 8 * N);
        }
        if (ib == 0) return 0;
        STL(53);
        uint8_t* q = blk + 9 + ib;
        if (threadIdx.x == 0) stu32(q, D);
        for (uint32_t k = threadIdx.x; k < D; k += WG) {
            Val<W> v = ko.key(firsts[k]);
            __builtin_memcpy(q + 4 + (uint64_t)k * W, &v, W);
        }
        body = ib + 4 + (uint64_t)D * W;
        STL(54);
    } else {
        if (threadIdx.x == 0) raise(a.status, SB_ERR_NYI, page, 522);  // LZ4/Zstd/Freq/Patas encode: host path
        return 0;
    }
    __syncthreads();
    if (threadIdx.x == 0) put_hdr9(blk, (uint32_t)CODEC, (uint32_t)body, (uint32_t)(N * W));
    return 9 + body;
}

template <int CODEC>
__device__ uint64_t emit_bool_page(const EncodeArgs& a, const EncCol& c, const EncPage& p, uint32_t page, uint8_t* blk,
                                   const ValidView& vv, uint32_t* sA, uint32_t* sB, uint32_t* sC, uint32_t* s_w) {
    const uint64_t N = p.rows;
    const uint8_t* bits = c.values;
    const uint64_t boff = c.values_bit_offset + p.row0;
    uint64_t body = 0;
    if constexpr (CODEC == SB_CODEC_RLE) {  // values as u8 0/1 (boolean/rle.rs:31-39)
        auto getv = [=](uint64_t i) {
            Val<1> v;
            v.x = bit_at(bits, boff + i) ? 1 : 0;
            return v;
        };
        body = enc_rle_rows<1, 0>(getv, vv, N, blk + 9, sA);
    } else if constexpr (CODEC == SB_CODEC_ONEVALUE) {  // boolean/one_value.rs:44-52
        __shared__ unsigned long long s_firstb;
        if (threadIdx.x == 0) s_firstb = ~0ull;
        __syncthreads();
        unsigned long long mine = ~0ull;
        for (uint64_t i = threadIdx.x; i < N && mine == ~0ull; i += WG)
            if (vv.get(i)) mine = i;
        if (mine != ~0ull) atomicMin(&s_firstb, mine);
        __syncthreads();
        if (threadIdx.x == 0) blk[9] = s_firstb == ~0ull ? 0 : (bit_at(bits, boff + s_firstb) ? 1 : 0);
        body = 1;
    } else {
        if (threadIdx.x == 0) raise(a.status, SB_ERR_NYI, page, 530);
        return 0;
    }
    __syncthreads();
    if (threadIdx.x == 0) put_hdr9(blk, (uint32_t)CODEC, (uint32_t)body, (uint32_t)N);  // rows, not bytes (mod.rs:59)
    return 9 + body;
}

// The builder's side of bin_dict_handover: the selector of this launch left id16[slot], firsts[id], D and the slot of every
// row in the aux area; what remains is the index of every row — a keyed row takes its key's id, a null row repeats the
// index before it (binary/dict.rs:55-93) — one streaming pass with the id table in LDS.
template <class O>
__device__ uint32_t bin_dict_from_handover(const BinKeys<O>& bk, uint64_t N, uint32_t* aux, uint32_t** idx_out, uint32_t** firsts_out,
                                           uint32_t* sA, uint32_t* sB, uint32_t* lds_id16 /* BH_SLOTS / 2 words */, uint32_t* s_w) {
    const uint32_t t = threadIdx.x;
    const uint64_t M = bh_table_slots(N);
    for (uint32_t k = t; k < BH_SLOTS / 2; k += WG) lds_id16[k] = gld32(aux + BH_W_ID16 + k);
    __syncthreads();
    const uint16_t* id16 = (const uint16_t*)lds_id16;
    const uint16_t* slot16 = (const uint16_t*)(aux + BH_W_SLOT16);
    uint32_t* idx = aux + M + 2 * N;
    if (!bk.vv.bits) {   // no validity bitmap: every row is keyed, the pass is one stream (16 rows per thread in flight)
        for (uint64_t base = (uint64_t)t * 2; base < N; base += (uint64_t)WG * 2 * 8) {
            uint32_t sl[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const uint64_t i = base + (uint64_t)u * WG * 2;
                sl[u] = i + 1 < N ? gld32((const uint32_t*)(slot16 + i)) : (i < N ? (uint32_t)ldu16((const uint8_t*)(slot16 + i)) : 0u);
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const uint64_t i = base + (uint64_t)u * WG * 2;
                if (i + 1 < N) {
                    gst64((uint64_t*)(idx + i), (uint64_t)id16[sl[u] & 0xFFFFu] | ((uint64_t)id16[sl[u] >> 16] << 32));
                } else if (i < N) {
                    gst32(idx + i, (uint32_t)id16[sl[u] & 0xFFFFu]);
                }
            }
        }
        __syncthreads();
        *idx_out = idx;
        *firsts_out = aux + BH_W_FIRSTS;
        return gld32(aux + BH_W_D);
    }
    uint32_t carry_id = 0;
    for (uint64_t cb = 0; cb < N; cb += TILE_ROWS) {
        const uint32_t n = (uint32_t)min((uint64_t)TILE_ROWS, N - cb);
        for (uint32_t i = t; i < TILE_ROWS; i += WG) {
            const bool kd = i < n && bk.keyed(cb + i);
            sB[sidx((int)i)] = kd ? i + 1 : 0;
            sA[sidx((int)i)] = kd ? (uint32_t)id16[ldu16((const uint8_t*)(slot16 + cb + i))] : 0u;
        }
        __syncthreads();
        tile_incl_scan_max(sB, s_w);
        for (uint32_t i = t; i < n; i += WG) {
            const uint32_t lk = sB[sidx((int)i)];
            gst32(idx + cb + i, lk ? sA[sidx((int)lk - 1)] : carry_id);
        }
        const uint32_t lkl = sB[sidx((int)n - 1)];
        if (lkl) carry_id = sA[sidx((int)lkl - 1)];
        __syncthreads();
    }
    *idx_out = idx;
    *firsts_out = aux + BH_W_FIRSTS;
    return gld32(aux + BH_W_D);
}

template <class O, int CODEC>
__device__ uint64_t emit_binary_page(const EncodeArgs& a, const EncCol& c, const EncPage& p, uint32_t page,
                                     uint8_t* blk, const ValidView& vv, uint32_t* sA, uint32_t* sB, uint32_t* sC,
                                     uint32_t* s_w, uint32_t* lds_table = nullptr, uint32_t lds_slots = 0) {
    const uint64_t N = p.rows;
    const uint8_t* offs = c.offsets + p.row0 * sizeof(O);
    auto off_at = [=](uint64_t i) {
        O o;
        __builtin_memcpy(&o, offs + i * sizeof(O), sizeof(O));
        return (uint64_t)o;
    };
    uint64_t body = 0;
    if constexpr (CODEC == SB_CODEC_ONEVALUE) {  // u32 len | bytes of the first valid row (binary/one_value.rs:50-68)
        __shared__ unsigned long long s_firstv;
        if (threadIdx.x == 0) s_firstv = ~0ull;
        __syncthreads();
        unsigned long long mine = ~0ull;
        for (uint64_t i = threadIdx.x; i < N && mine == ~0ull; i += WG)
            if (vv.get(i)) mine = i;
        if (mine != ~0ull) atomicMin(&s_firstv, mine);
        __syncthreads();
        uint64_t b = 0, e = 0;
        if (s_firstv != ~0ull) {
            b = off_at(s_firstv);
            e = off_at(s_firstv + 1);
        }
        if (threadIdx.x == 0) stu32(blk + 9, (uint32_t)(e - b));
        wg_copy(blk + 13, c.values + b, e - b);
        body = 4 + (e - b);
    } else if constexpr (CODEC == SB_CODEC_DICT) {  // binary/dict.rs:55-93
        BinKeys<O> ko{offs, c.values, vv};
        uint32_t *idx, *firsts;
        uint32_t* aux = (uint32_t*)(a.scratch + p.aux_off);
        uint32_t D;
        STL(20);
        // the selector of this launch built the dictionary (tables in the aux area) and k_enc_bin_verify compared the strings
        const bool handed = p.codec == CODEC_ON_DEVICE && p.h64_off != ~0ull && lds_slots >= BH_SLOTS && bh_fits(N, p.aux_bytes) &&
                            gld32(aux + BH_W_MAGIC) == BH_MAGIC && gld32(aux + BH_W_BAD) == 0;
        uint32_t pre_bp = 0, pre_ent = 0, pre_ent_word = 0, pre_ic = 0;   // k_enc_bin_page's index codec (+ 1) / bit-packed block / staged entries
        if (bp_page_done(a, p, page) && gld32(aux + BH_W_MAGIC) == BH_MAGIC2) {   // ids, first rows and the index array by k_enc_bin_page
            idx = aux + bh_table_slots(N) + 2 * N;
            firsts = aux + BP_W_FIRSTS;
            D = gld32(aux + BH_W_D);
            pre_ic = gld32(aux + BH_W_ICODEC);
            pre_bp = gld32(aux + BH_W_BPBYTES);
            pre_ent = gld32(aux + BH_W_ENTBYTES);
            pre_ent_word = gld32(aux + BH_W_ENTWORD);
        } else if (handed) {
            D = bin_dict_from_handover<O>(ko, N, aux, &idx, &firsts, sA, sB, sC, s_w);
            STL(27);
        } else if (p.h64_off != ~0ull) {   // hashed rows: computed by the selector of this call, or here when the codec was forced
            uint64_t* h64 = (uint64_t*)(a.scratch + p.h64_off);
            if (p.codec != CODEC_ON_DEVICE) {
                if (lds_slots >= 4096)
                    bin_hash_rows_staged<O>(ko, N, c.values_len, h64, lds_table, lds_slots * 4);
                else
                    bin_hash_rows<O>(ko, N, c.values_len, h64);
            }
            D = dict_build(BinKeysHashed<O>{ko, h64, c.values_len}, N, aux, p.aux_bytes / 4, &idx, &firsts, sA, sB, s_w, a.status, page,
                           lds_slots > 1 ? lds_table : nullptr, lds_slots > 1 ? lds_slots : 0);
        } else {
            D = dict_build(ko, N, aux, p.aux_bytes / 4, &idx, &firsts, sA, sB, s_w, a.status, page,
                           lds_slots > 1 ? lds_table : nullptr, lds_slots > 1 ? lds_slots : 0);
        }
        if (D == EMPTY) return 0;
        int32_t ic = p.icodec >= 0 ? p.icodec : (int32_t)a.default_compression;
        if (p.icodec < 0 && a.has_ratio && pre_ic) {   // chosen by k_enc_bin_page
            ic = (int32_t)pre_ic - 1;
        } else if (p.icodec < 0 && a.has_ratio) {  // nested compress_integer::<u32>: same selector, Dict forbidden (dict.rs:60-62)
            SelectOpts so{a.ratio, 1u, a.forbidden | p.forb_extra | (1u << SB_CODEC_DICT), a.default_compression, -1, p.seed, p.depth + 1};
            SelScratch sc{sA, sA + SEL_LDS_SLOTS, (uint8_t*)(sA + SEL_LDS_SLOTS + 2 * WG + 16), nullptr, 0};
            const uint32_t* ip = idx;
            ic = (int32_t)choose_prim<4>([=](uint64_t i) { Val<4> v; v.x = gld32(ip + i); return v; },
                                         ValidView{nullptr, 0}, N, NK_UNSIGNED, so, sc);
            __syncthreads();
        }
        STL(30);
        if (ic == SB_CODEC_FREQ) {  // as in emit_prim_page<Dict>: the Freq kernels write the index block and the entries
            if (p.vaux_bytes < 32 || !p.vslot_off) {
                if (threadIdx.x == 0) raise(a.status, SB_ERR_NYI, page, 502);
                return 0;
            }
            if (threadIdx.x == 0) {
                unsigned long long* rec = (unsigned long long*)(a.scratch + p.vaux_off);
                rec[0] = (unsigned long long)(uintptr_t)idx;
                rec[1] = (unsigned long long)(uintptr_t)firsts;
                rec[2] = D;
                atomicAdd(a.freq_count, 1u);
            }
            return DICT_FREQ_PENDING;
        }
        // (HBM scratch for the three-wave LZ4 path: 2 N words of the aux area that are dead by now — the F / R arrays of
        // dict_build, or the tail of the area when dict_build_lds put the indices at its start)
        uint8_t* lz_tmp = nullptr;
        {
            uint64_t M = 64;
            while (M < 2 * N) M <<= 1;
            const uint64_t aux_words = p.aux_bytes / 4;
            if (aux_words >= M + 3 * N) lz_tmp = (uint8_t*)(idx == aux ? aux + (aux_words - 2 * N) : aux + M);
        }
        uint64_t ib;
        if (ic == SB_CODEC_BITPACKING && pre_bp) {   // the body stands there already: the block's header
            if (threadIdx.x == 0) put_hdr9(blk + 9, SB_CODEC_BITPACKING, pre_bp, (uint32_t)(N * 4));
            ib = 9 + (uint64_t)pre_bp;
        } else {
            ib = enc_u32_block(idx, N, ic, blk + 9, sA, sB, sC, s_w, a.status, page, a.flags, lz_tmp, 8 * N);
        }
        if (ib == 0) return 0;
        STL(31);
        uint8_t* q = blk + 9 + ib;
        if (threadIdx.x == 0) stu32(q, D);
        q += 4;
        // entries: u64 len | bytes, in dictionary order; positions = scan of (8 + len)
        uint64_t pos = 0;
        if (pre_ent) {   // staged by k_enc_bin_page
            wg_copy(q, (const uint8_t*)(aux + pre_ent_word), pre_ent);
            pos = pre_ent;
        }
        for (uint32_t kb = pre_ent ? D : 0u; kb < D; kb += TILE_ROWS) {
            const uint32_t n = min((uint32_t)TILE_ROWS, D - kb);
            for (uint32_t i = threadIdx.x; i < TILE_ROWS; i += WG) {
                uint32_t len = 0;
                if (i < n) {
                    const uint64_t r = gld32(firsts + kb + i);
                    len = (uint32_t)(ko.beg(r + 1) - ko.beg(r)) + 8;
                }
                sA[sidx((int)i)] = len;
            }
            __syncthreads();
            const uint32_t tot = tile_incl_scan(sA, s_w);
            for (uint32_t i0 = threadIdx.x; i0 < n; i0 += WG * 4) {   // four entries per thread in flight (first row -> offsets -> bytes)
                uint64_t rr[4], bb[4], ee[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const uint32_t i = i0 + (uint32_t)u * WG;
                    rr[u] = i < n ? gld32(firsts + kb + i) : 0u;
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    bb[u] = ko.beg(rr[u]);
                    ee[u] = ko.beg(rr[u] + 1);
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const uint32_t i = i0 + (uint32_t)u * WG;
                    if (i >= n) continue;
                    const uint64_t b = bb[u], e = ee[u];
                    uint8_t* d = q + pos + sA[sidx((int)i)] - (e - b) - 8;
                    stu64(d, e - b);
                    if (e - b <= 32 && b + 32 <= c.values_len) {   // the common short string: two unconditional 16-byte loads
                        const u32x4 v0 = ldu128(c.values + b), v1 = ldu128(c.values + b + 16);
                        uint64_t w[4] = {(uint64_t)v0.x | ((uint64_t)v0.y << 32), (uint64_t)v0.z | ((uint64_t)v0.w << 32),
                                         (uint64_t)v1.x | ((uint64_t)v1.y << 32), (uint64_t)v1.z | ((uint64_t)v1.w << 32)};
                        const uint32_t len = (uint32_t)(e - b), full = len >> 3;
#pragma unroll
                        for (uint32_t q8 = 0; q8 < 4; q8++)
                            if (q8 < full) stu64(d + 8 + 8 * q8, w[q8]);
                        uint64_t rest = full == 0 ? w[0] : full == 1 ? w[1] : full == 2 ? w[2] : full == 3 ? w[3] : 0ull;
                        for (uint32_t k = full * 8; k < len; k++) {
                            *(gptr)(d + 8 + k) = (uint8_t)rest;
                            rest >>= 8;
                        }
                        continue;
                    }
                    uint64_t k = 0;
                    for (; k + 16 <= e - b; k += 16) stu128(d + 8 + k, ldu128(c.values + b + k));   // (unaligned 16-byte moves)
                    for (; k + 8 <= e - b; k += 8) stu64(d + 8 + k, ldu64(c.values + b + k));
                    for (; k < e - b; k++) *(gptr)(d + 8 + k) = ldu8(c.values + b + k);
                }
            }
            pos += tot;
            __syncthreads();
        }
        STL(32);
        body = ib + 4 + pos;
    } else {
        if (threadIdx.x == 0) raise(a.status, SB_ERR_NYI, page, 540);
        return 0;
    }
    __syncthreads();
    // uncompressed_size = array.values().len(): the whole shared buffer (binary/mod.rs:88)
    if (threadIdx.x == 0) put_hdr9(blk, (uint32_t)CODEC, (uint32_t)body, (uint32_t)c.values_len_total);
    return 9 + body;
}

// codec choice per page (adaptive mode): one workgroup per page, one instance per KIND
#include "sb_select_rle.h"
#include "sb_select_runs.h"
#include "sb_select_big.h"
#include "sb_dict_big.h"
#include "sb_bin_page.h"

// Binary pages hash strings: a probe that misses the LDS tier costs a random HBM access per row (13 GB of traffic for
// 1.15 GB of C3 input when the table sat in HBM), so their LDS table is 16 Ki slots (~10 000 distinct strings per page).
constexpr uint32_t BIN_LDS_SLOTS = 16384;

// The 64-bit row hashes of adaptive binary pages, TILE-parallel: hashing a page is independent work per row, and inside the
// page kernels it was a serial chain of ~20 staged tiles per page (0.37 ms of a page's 0.93 ms in the selector even alone on
// a CU).  One workgroup per (page, BH_ROWS rows): the tile's bytes are staged in LDS with coalesced 16-byte loads and hashed
// from there; the selector and the dictionary builder then start from h64.
constexpr uint32_t BH_ROWS = 2048, BH_LDS_WORDS = 10240;   // 40 KiB of staging: 4 workgroups per CU
#ifndef BV_ROWS_
#define BV_ROWS_ 4096
#endif
constexpr uint32_t BV_ROWS = BV_ROWS_;   // rows per workgroup of k_enc_bin_verify
__global__ void __launch_bounds__(WG) k_enc_bin_hash(EncodeArgs a) {
    const uint32_t page = blockIdx.x + a.page_base;
    const EncPage p = get_page(a, page);
    if (p.codec != CODEC_ON_DEVICE || p.h64_off == ~0ull) return;
    const uint64_t r0 = (uint64_t)blockIdx.y * BH_ROWS;
    if (r0 >= p.rows) return;
    if (bp_page_done(a, p, page)) return;   // decided by k_enc_bin_page
    const EncCol c = get_col(a, p.col);
    const uint64_t r1 = min(p.rows, r0 + BH_ROWS);
    uint64_t* h64 = (uint64_t*)(a.scratch + p.h64_off);
    // Thread = row, eight rows in flight: the offsets of neighbouring rows are neighbours and so are their bytes (a wave's
    // 64 strings span about a kilobyte), so the loads are near-coalesced without staging the tile in LDS — and with
    // thousands of tiles in flight nobody waits for a round trip (staged through LDS the kernel took 0.50 ms on C3).
    auto body = [&](auto bk) {
        constexpr int U = 8;
        for (uint64_t base = r0 + threadIdx.x; base < r1; base += (uint64_t)WG * U) {
            uint64_t b[U], e[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint64_t i = base + (uint64_t)u * WG;
                const uint64_t ic = i < r1 ? i : r0;
                b[u] = bk.beg(ic);
                e[u] = bk.beg(ic + 1);
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint64_t i = base + (uint64_t)u * WG;
                if (i < r1) gst64(h64 + i, bin_hash_bytes(bk.values, b[u], e[u], c.values_len));
            }
        }
    };
    const ValidView vv{c.validity, c.validity_bit_offset + p.row0};
    if (c.ptype == SB_TYPE_BINARY)
        body(BinKeys<int32_t>{c.offsets + p.row0 * 4, c.values, vv});
    else if (c.ptype == SB_TYPE_LARGE_BINARY)
        body(BinKeys<int64_t>{c.offsets + p.row0 * 8, c.values, vv});
}

// Binary pages whose keys the selector counted with its tag table (distinct_count_tags): the string of EVERY row against
// the smallest row of its slot, TILE-parallel (one workgroup per (page, BH_ROWS rows)).  Slots were formed by 32-bit tags
// of 64-bit hashes; a slot that holds two different strings (a tag or hash collision: never seen) sets the page's BAD
// word, and the page is selected again without tags (k_enc_select, redo pass) and built by the exact builder.
__global__ void __launch_bounds__(WG) k_enc_bin_verify(EncodeArgs a, uint32_t tiles_per_page) {
    // A row is compared with the first row of its class — somewhere earlier in the page — so the tiles of a page share what
    // they gather (the strings of the frequent classes).  Workgroups go to XCD (linear id mod 8), each with an L2 of its own:
    // the tiles of a page are handed out one after the other ON ONE XCD (8 pages side by side) instead of 1024 pages apart
    // (PMC: 4.3 GB fetched per launch for 1.15 GB of strings — every tile fetched its classes' first rows again).
    const uint32_t lin = blockIdx.x, xcd = lin & 7, j = lin >> 3;
    const uint32_t pidx = (j / tiles_per_page) * 8 + xcd, tile = j % tiles_per_page;
    if (pidx >= a.n_pages) return;
    const uint32_t page = pidx + a.page_base;
    const EncPage p = get_page(a, page);
    if (p.codec != CODEC_ON_DEVICE || p.h64_off == ~0ull || !p.aux_bytes) return;
    const uint64_t r0 = (uint64_t)tile * BV_ROWS;
    if (r0 >= p.rows) return;
    if (bp_page_done(a, p, page)) return;   // decided by k_enc_bin_page: its table compares the strings
    uint32_t* aux = (uint32_t*)(a.scratch + p.aux_off);
    if (!bh_fits(p.rows, p.aux_bytes)) return;
    const uint32_t magic = gld32(aux + BH_W_MAGIC);
    if (magic != BH_MAGIC && magic != BH_TAGS_USED) return;
    if (a.flags & SB_WRITE_DEBUG_VERIFY_FAIL_BIT) {
        if (threadIdx.x == 0 && tile == 0) atomicOr(aux + BH_W_BAD, 1u);
        return;
    }
    const EncCol c = get_col(a, p.col);
    const uint64_t r1 = min(p.rows, r0 + BV_ROWS);
    const ValidView vv{c.validity, c.validity_bit_offset + p.row0};
    const uint16_t* slot16 = (const uint16_t*)(aux + BH_W_SLOT16);
    const uint16_t* rep16 = (const uint16_t*)(aux + BH_W_REP16);
    const uint64_t* h64 = (const uint64_t*)(a.scratch + p.h64_off);
    bool ok = true;
    // A divergent load costs the CU's address unit one cycle per lane, so the pass is sized in GATHERS per row: the slot's
    // representative row (1), its offset pair with one 8-byte load (1), its first 16 bytes (1, strings of up to 16 bytes) —
    // the row's own offsets and bytes are near-contiguous across lanes.  Longer strings take eq().
    auto body = [&](auto keys) {
        using KT = decltype(keys);
        constexpr int VU = 8;
        constexpr bool O32 = sizeof(keys.k.beg(0)) == 8 && std::is_same<KT, BinKeysHashed<int32_t>>::value;
        const uint8_t* offs = keys.k.offs;
        const uint8_t* values = keys.k.values;
        const uint64_t vlen = c.values_len;
        for (uint64_t base = r0 + threadIdx.x; base < r1; base += (uint64_t)WG * VU) {
            uint32_t g[VU];
            uint64_t br[VU], er[VU], bg[VU], eg[VU];
            bool need[VU];
#pragma unroll
            for (int u = 0; u < VU; u++) {
                const uint64_t row = base + (uint64_t)u * WG;
                const uint64_t rc = row < r1 ? row : r0;
                g[u] = ldu16((const uint8_t*)(rep16 + ldu16((const uint8_t*)(slot16 + rc))));
                need[u] = row < r1 && g[u] != (uint32_t)rc;
                if (!need[u]) g[u] = (uint32_t)rc;
            }
#pragma unroll
            for (int u = 0; u < VU; u++) {
                const uint64_t row = base + (uint64_t)u * WG;
                const uint64_t rc = row < r1 ? row : r0;
                if constexpr (O32) {
                    const uint64_t pr = ldu64(offs + rc * 4), pg = ldu64(offs + (uint64_t)g[u] * 4);
                    br[u] = (uint32_t)pr; er[u] = pr >> 32; bg[u] = (uint32_t)pg; eg[u] = pg >> 32;
                } else {
                    br[u] = keys.k.beg(rc); er[u] = keys.k.beg(rc + 1); bg[u] = keys.k.beg(g[u]); eg[u] = keys.k.beg((uint64_t)g[u] + 1);
                }
            }
#pragma unroll
            for (int u = 0; u < VU; u++) {
                if (!need[u]) continue;
                const uint64_t n = er[u] - br[u];
                if (n != eg[u] - bg[u]) {
                    ok = false;
                    continue;
                }
                if (n <= 32 && br[u] + 32 <= vlen && bg[u] + 32 <= vlen) {   // two unconditional 16-byte pairs, byte masks
                    const u32x4 a4 = ldu128(values + br[u]), b4 = ldu128(values + bg[u]);
                    const u32x4 a5 = ldu128(values + br[u] + 16), b5 = ldu128(values + bg[u] + 16);
                    auto msk = [](uint64_t have) -> uint64_t { return have >= 8 ? ~0ull : ((1ull << (8 * have)) - 1); };
                    const uint64_t d0 = ((uint64_t)(a4.x ^ b4.x)) | ((uint64_t)(a4.y ^ b4.y) << 32), d1 = ((uint64_t)(a4.z ^ b4.z)) | ((uint64_t)(a4.w ^ b4.w) << 32);
                    const uint64_t d2 = ((uint64_t)(a5.x ^ b5.x)) | ((uint64_t)(a5.y ^ b5.y) << 32), d3 = ((uint64_t)(a5.z ^ b5.z)) | ((uint64_t)(a5.w ^ b5.w) << 32);
                    const uint64_t x = (d0 & msk(n)) | (d1 & msk(n > 8 ? n - 8 : 0)) | (d2 & msk(n > 16 ? n - 16 : 0)) | (d3 & msk(n > 24 ? n - 24 : 0));
                    if (x) ok = false;
                } else {
                    const uint64_t row = base + (uint64_t)u * WG;
                    if (!keys.k.eq(g[u], row)) ok = false;
                }
            }
        }
    };
    if (c.ptype == SB_TYPE_BINARY)
        body(BinKeysHashed<int32_t>{BinKeys<int32_t>{c.offsets + p.row0 * 4, c.values, vv}, h64, c.values_len});
    else
        body(BinKeysHashed<int64_t>{BinKeys<int64_t>{c.offsets + p.row0 * 8, c.values, vv}, h64, c.values_len});
    if (!ok) atomicOr(aux + BH_W_BAD, 1u);
}

template <int KIND>
__global__ void __launch_bounds__(WG) k_enc_select(EncodeArgs a) {
    constexpr uint32_t LDS_SLOTS = KIND < 0 ? BIN_LDS_SLOTS : SEL_LDS_SLOTS;
    __shared__ uint32_t lds_tab[LDS_SLOTS];
    __shared__ uint32_t s_misc[2 * WG + 16];
    // sample area, also the streaming scratch of choose_prim: 1 KB validity words + 4 x 128 keys
    constexpr int SMP = SAMPLE_CAP * ((KIND > 0 ? KIND : 1) + 1) + 16, STR = 1024 + 4 * 96 * (KIND == 8 ? 8 : 4) + 4 * 96;
    __shared__ __attribute__((aligned(16))) uint8_t sample_mem[SMP > STR ? SMP : STR];
    static_assert(KIND >= 0 || LDS_SLOTS >= BH_SLOTS + 4096, "tag slots + the hand-over's bitmap and prefix share lds_tab");
    __shared__ uint32_t s_tagged;
    const uint32_t page = spread_block(blockIdx.x, gridDim.x) + a.page_base;
    const EncPage p = get_page(a, page);
    if (p.codec != CODEC_ON_DEVICE) return;
    const EncCol c = get_col(a, p.col);
    if (c.ptype == SB_TYPE_NULL) return;
    const bool is_bool = c.ptype == SB_TYPE_BOOLEAN;
    const bool is_bin = c.ptype == SB_TYPE_BINARY || c.ptype == SB_TYPE_LARGE_BINARY;
    if constexpr (KIND == 0) {
        if (!is_bool) return;
    } else if constexpr (KIND == -4) {
        if (c.ptype != SB_TYPE_BINARY) return;
    } else if constexpr (KIND == -8) {
        if (c.ptype != SB_TYPE_LARGE_BINARY) return;
    } else {
        if (is_bool || is_bin || c.width != (uint32_t)KIND) return;
    }
    const uint64_t N = p.rows;
    if constexpr (KIND == 1 || KIND == 2) {
        // long pages in the adaptive wave of a flat call: left to the section-parallel selector (launched after this kernel)
        if (a.use_counts && !a.redo && a.page_base == 0 && page < a.n_pages && N >= SEL_BIG_ROWS) {
            if (threadIdx.x == 0) a.codecs[page] = CODEC_PENDING;
            return;
        }
    }
    if constexpr (KIND < 0) {
        if (bp_page_done(a, p, page)) return;   // codec and dictionary by k_enc_bin_page (sb_bin_page.h)
        // long binary pages: the statistics over their row hashes section-parallel, the Dict pages among them too
        // (sb_select_big.h / sb_dict_big.h, launched after this kernel)
        if (a.use_counts && !a.redo && a.page_base == 0 && page < a.n_pages && N >= BIN_BIG_ROWS && p.bigx_off && p.h64_off != ~0ull &&
            a.pre_hashed) {
            if (threadIdx.x == 0) a.codecs[page] = CODEC_PENDING;
            return;
        }
    }
    const ValidView vv{c.validity, c.validity_bit_offset + p.row0};
    SelectOpts so{a.ratio, a.has_ratio, a.forbidden | p.forb_extra, a.default_compression, -1, p.seed, p.depth};
    SelScratch sc{lds_tab, s_misc, sample_mem, p.aux_bytes ? (uint32_t*)(a.scratch + p.aux_off) : nullptr, 0};
    sc.lds_slots = LDS_SLOTS;
    if (sc.gtab) {  // the Dict aux area starts with a table of pow2 >= 2N slots
        uint64_t M = 64;
        while (M < 2 * N) M <<= 1;
        sc.gslots = M;
    }
    uint32_t codec;
    bool handover = false;
    if constexpr (KIND < 0) {   // the dictionary of a Dict page is handed to the builder (bin_dict_handover) when its tables fit
        if (threadIdx.x == 0) s_tagged = 0;
        const bool fits = p.h64_off != ~0ull && page < a.n_pages && sc.gtab && bh_fits(N, p.aux_bytes) && !((a.forbidden >> SB_CODEC_DICT) & 1);
        if (a.redo) {   // second pass: only the pages k_enc_bin_verify failed, counted exactly this time
            if (!fits) return;
            const uint32_t magic = gld32(sc.gtab + BH_W_MAGIC);
            if ((magic != BH_MAGIC && magic != BH_TAGS_USED) || gld32(sc.gtab + BH_W_BAD) == 0) return;
            if (threadIdx.x == 0) {   // the first pass's choice is withdrawn
                const uint32_t old = (uint32_t)a.codecs[page];
                atomicSub(&a.codec_counts[old & 31], 1u);
                if (old == SB_CODEC_FREQ) atomicSub(a.freq_count, 1u);
            }
        } else if (fits) {
            handover = true;
            sc.slot16 = (uint16_t*)(sc.gtab + BH_W_SLOT16);
            sc.tagged = &s_tagged;
        }
        __syncthreads();
        if (threadIdx.x == 0 && sc.gtab && p.aux_bytes >= 64) {
            gst32(sc.gtab + BH_W_MAGIC, 0u);
            gst32(sc.gtab + BH_W_BAD, 0u);
        }
        __syncthreads();
    } else {
        if (a.redo) return;
    }
    if constexpr (KIND == 0) {
        codec = choose_bool(c.values, c.values_bit_offset + p.row0, vv, N, so, sc);
    } else if constexpr (KIND == -4) {
        BinKeys<int32_t> bk{c.offsets + p.row0 * 4, c.values, vv};
        codec = choose_bin<int32_t>(bk, N, c.values_len_total, so, sc, p.h64_off != ~0ull ? (uint64_t*)(a.scratch + p.h64_off) : nullptr, c.values_len,
                                    a.pre_hashed != 0);
        __syncthreads();
        if (handover && s_tagged) {
            if (codec == SB_CODEC_DICT)
                bin_dict_handover<int32_t>(bk, N, lds_tab, lds_tab + BH_SLOTS, s_misc, sc.gtab);
            else if (threadIdx.x == 0)
                gst32(sc.gtab + BH_W_MAGIC, BH_TAGS_USED);
        }
        STL(46);
    } else if constexpr (KIND == -8) {
        BinKeys<int64_t> bk{c.offsets + p.row0 * 8, c.values, vv};
        codec = choose_bin<int64_t>(bk, N, c.values_len_total, so, sc, p.h64_off != ~0ull ? (uint64_t*)(a.scratch + p.h64_off) : nullptr, c.values_len,
                                    a.pre_hashed != 0);
        __syncthreads();
        if (handover && s_tagged) {
            if (codec == SB_CODEC_DICT)
                bin_dict_handover<int64_t>(bk, N, lds_tab, lds_tab + BH_SLOTS, s_misc, sc.gtab);
            else if (threadIdx.x == 0)
                gst32(sc.gtab + BH_W_MAGIC, BH_TAGS_USED);
        }
    } else {
        const uint8_t* vals = c.values + p.row0 * KIND;
        codec = choose_prim<KIND>([=](uint64_t i) { return ld_val<KIND>(vals + i * KIND); }, vv, N, c.nk, so, sc);
    }
    if (threadIdx.x == 0) {
        a.codecs[page] = (int32_t)codec;
        atomicAdd(&a.codec_counts[codec & 31], 1u);
        if (!has_device_encoder(codec))
            raise(a.status, SB_ERR_NYI, page, 700 + codec);
        else if (codec == SB_CODEC_FREQ && page < a.n_pages)
            atomicAdd(a.freq_count, 1u);
    }
}

// Adaptive pages of 4- / 8-byte values: statistics and speculative RLE in one pass (sb_select_rle.h).
// One instance per (width, float-ness): the RLE equality differs, and keeping each instance a kernel of
// its own keeps it at 4 waves / SIMD (several big inlined paths in one kernel double the registers).
template <int KIND, int FK>
__global__ void __launch_bounds__(WG, 4) k_enc_select_rle(EncodeArgs a) {
    __shared__ uint32_t lds_tab[SEL_LDS_SLOTS];
    __shared__ uint32_t s_misc[2 * WG + 16];
    __shared__ __attribute__((aligned(16))) uint8_t sample_mem[SAMPLE_CAP * (KIND + 1) + 16];
    __shared__ uint32_t s_cnt2[2];
    if (a.codec_counts[31] == 0) return;  // k_enc_select_runs (launched before) took every page
    const uint32_t page = blockIdx.x;
    const EncPage p = a.pages[page];
    if (p.codec != CODEC_ON_DEVICE) return;
    const EncCol c = a.cols[p.col];
    if (c.ptype == SB_TYPE_NULL || c.ptype == SB_TYPE_BOOLEAN || c.ptype == SB_TYPE_BINARY || c.ptype == SB_TYPE_LARGE_BINARY) return;
    if (c.width != (uint32_t)KIND || (c.fkind != 0) != (FK != 0)) return;
    if (a.codecs[page] != CODEC_PENDING) return;  // k_enc_select_runs (launched before) took the page
    const uint64_t N = p.rows;
    if (N >= SEL_BIG_ROWS) return;   // long pages: section-parallel (sb_select_big.h, launched after this kernel)
    SelectOpts so{a.ratio, a.has_ratio, a.forbidden | p.forb_extra, a.default_compression, -1, p.seed, p.depth};
    SelScratch sc{lds_tab, s_misc, sample_mem, p.aux_bytes ? (uint32_t*)(a.scratch + p.aux_off) : nullptr, 0};
    if (sc.gtab) {  // the Dict aux area starts with a table of pow2 >= 2N slots
        uint64_t M = 64;
        while (M < 2 * N) M <<= 1;
        sc.gslots = M;
    }
    uint32_t codec = so.default_codec;
    bool kept = false;
    if (N > 0) codec = select_rle_page<KIND, FK>(a, c, p, page, so, sc, s_cnt2, &kept);
    if (threadIdx.x == 0) {
        a.codecs[page] = (int32_t)codec;
        if (!kept) atomicAdd(&a.codec_counts[codec & 31], 1u);  // (kept: the RLE page is already written)
        if (!has_device_encoder(codec))
            raise(a.status, SB_ERR_NYI, page, 700 + codec);
        else if (codec == SB_CODEC_FREQ)
            atomicAdd(a.freq_count, 1u);
    }
}

// The same selection + speculative RLE with one raw run per lane (sb_select_runs.h); pages whose runs are
// too short are left to k_enc_select_rle (CODEC_PENDING).
#ifndef SB_RUNS_OCC
#define SB_RUNS_OCC 4
#endif
template <int KIND, int FK>
__global__ void __launch_bounds__(WG, SB_RUNS_OCC) k_enc_select_runs(EncodeArgs a) {
    __shared__ uint32_t lds_tab[SEL_LDS_SLOTS];
    // the sample area and the misc words are one pool: the run list and the run values of the streaming loop use
    // both (they are free until decide_prim runs)
    constexpr int SMP_BYTES = (SAMPLE_CAP * (KIND + 1) + 16 + 15) / 16 * 16, MISC_BYTES = (2 * WG + 16) * 4;
    __shared__ __attribute__((aligned(16))) uint8_t pool[SMP_BYTES + MISC_BYTES];
    uint8_t* sample_mem = pool;
    uint32_t* s_misc = (uint32_t*)(pool + SMP_BYTES);
    __shared__ uint32_t s_cnt2[2];
    static_assert(SMP_BYTES + MISC_BYTES >= 2 * (RUNS_CAP + 8) + (WG * 16 / 32 + 32) * 4 + 8 * KIND + (RUNS_CAP + 1) * KIND,
                  "run list and run values must fit the pool");
    const uint32_t page = blockIdx.x;
    const EncPage p = a.pages[page];
    if (p.codec != CODEC_ON_DEVICE) return;
    const EncCol c = a.cols[p.col];
    if (c.ptype == SB_TYPE_NULL || c.ptype == SB_TYPE_BOOLEAN || c.ptype == SB_TYPE_BINARY || c.ptype == SB_TYPE_LARGE_BINARY) return;
    if (c.width != (uint32_t)KIND || (c.fkind != 0) != (FK != 0)) return;
    const uint64_t N = p.rows;
    if (N >= SEL_BIG_ROWS) {   // long pages: section-parallel selection and RLE (sb_select_big.h), not one workgroup's walk
        if (threadIdx.x == 0) {
            a.codecs[page] = CODEC_PENDING;
            atomicAdd(&a.codec_counts[31], 1u);
        }
        return;
    }
    SelectOpts so{a.ratio, a.has_ratio, a.forbidden | p.forb_extra, a.default_compression, -1, p.seed, p.depth};
    SelScratch sc{lds_tab, s_misc, sample_mem, p.aux_bytes ? (uint32_t*)(a.scratch + p.aux_off) : nullptr, 0};
    if (sc.gtab) {  // the Dict aux area starts with a table of pow2 >= 2N slots
        uint64_t M = 64;
        while (M < 2 * N) M <<= 1;
        sc.gslots = M;
    }
    uint32_t codec = so.default_codec;
    bool kept = false, fallback = false;
    if (N > 0) codec = select_runs_page<KIND, FK>(a, c, p, page, so, sc, s_cnt2, &kept, &fallback);
    if (threadIdx.x == 0) {
        if (fallback) {
            a.codecs[page] = CODEC_PENDING;
            atomicAdd(&a.codec_counts[31], 1u);
            return;
        }
        a.codecs[page] = (int32_t)codec;
        if (!kept) atomicAdd(&a.codec_counts[codec & 31], 1u);  // (kept: the RLE page is already written)
        if (!has_device_encoder(codec))
            raise(a.status, SB_ERR_NYI, page, 700 + codec);
        else if (codec == SB_CODEC_FREQ)
            atomicAdd(a.freq_count, 1u);
    }
}

typedef void (*EncSelectKernel)(EncodeArgs);
static EncSelectKernel enc_select_kernel(int kind) {
    switch (kind) {
        case 0:
            return k_enc_select<0>;
        case 1:
            return k_enc_select<1>;
        case 2:
            return k_enc_select<2>;
        case 4:
            return k_enc_select<4>;
        case 8:
            return k_enc_select<8>;
        case 16:
            return k_enc_select<16>;
        case 32:
            return k_enc_select<32>;
        case -4:
            return k_enc_select<-4>;
        case -8:
            return k_enc_select<-8>;
    }
    return nullptr;
}

// Pages with an extended codec: one workgroup per page.  One kernel instance per (KIND, CODEC):
// KIND = value width 1..32 for primitives, 0 = boolean, -4 / -8 = binary with i32 / i64 offsets.
// A monolithic kernel over all kinds needs 300 VGPRs (1 wave/SIMD); the split instances stay
// within 4 waves/SIMD.  Pages of another kind/codec exit at once.
// workgroups per CU the instances are compiled for (what their registers and LDS really allow: Dict pages hold a 64 KiB
// table or the hand-over's id table, Patas needs ~200 VGPRs)
template <int KIND, int CODEC>
constexpr int emit_pages_occupancy() {
    return CODEC == SB_CODEC_DICT ? 2 : CODEC == SB_CODEC_PATAS ? 1 : ((CODEC == SB_CODEC_ONEVALUE || CODEC == SB_CODEC_RLE) && KIND == 32) ? 3
           : (CODEC == SB_CODEC_RLE || CODEC == SB_CODEC_ONEVALUE) ? 4 : 3;
}
template <int KIND, int CODEC>
__global__ void __launch_bounds__(WG, (emit_pages_occupancy<KIND, CODEC>()))
    k_enc_emit_pages(EncodeArgs a) {
    // RLE / OneValue only need the small per-group records; Dict and bit-packing use full tile arrays
    constexpr int LW = CODEC == SB_CODEC_ONEVALUE ? 256
                       : CODEC == SB_CODEC_RLE ? (int)(RleRows<(KIND > 0 ? KIND : 1)>::WORDS + 2) / 3
                                               : SIDX_WORDS;
    // binary Dict pages: the table of first rows sits in LDS while the page has at most ~10 000 distinct strings.  It is
    // dead before the scans of the build start, so it shares the 3 * LW words of sA / sB / sC.
    constexpr uint32_t BIN_TABLE = (KIND < 0 && CODEC == SB_CODEC_DICT) ? BIN_LDS_SLOTS : 1;
    // (a handed-over dictionary: sA, sB and behind them the id table of BH_SLOTS u16 entries, see bin_dict_from_handover)
    constexpr int LDS_MIN = (KIND < 0 && CODEC == SB_CODEC_DICT) ? 2 * LW + (int)BH_SLOTS / 2 : 0;
    constexpr int LDS_WORDS0 = 3 * LW > (int)BIN_TABLE ? 3 * LW : (int)BIN_TABLE;
    constexpr int LDS_WORDS = LDS_WORDS0 > LDS_MIN ? LDS_WORDS0 : LDS_MIN;
    __shared__ __attribute__((aligned(16))) uint32_t lds[LDS_WORDS];
    uint32_t *sA = lds, *sB = lds + LW, *sC = lds + 2 * LW;
    uint32_t* const s_bin_table = lds;
    __shared__ uint32_t s_w[4];
    if (a.use_counts && a.codec_counts[CODEC] == 0) return;  // adaptive batch without a page of this codec
    const uint32_t page = spread_block(blockIdx.x, gridDim.x) + a.page_base;
    const EncPage p = get_page(a, page);
    if (codec_of(a, p, page) != CODEC) return;
    if constexpr (CODEC == SB_CODEC_RLE || CODEC == SB_CODEC_DICT) {
        // already emitted: by the fused select + RLE pass / by the section-parallel writers of long pages (sb_dict_big.h)
        if (a.outs[page].pad == 1 && a.outs[page].length != 0) return;
    }
    if constexpr (CODEC == SB_CODEC_RLE) {
        if (threadIdx.x == 0 && a.use_counts) atomicAdd(&a.codec_counts[29], 1u);   // (RLE pages the fused selectors did not write: the hint for the next call)
    }
    if constexpr (CODEC == SB_CODEC_DICT) {
        // a long page whose section-parallel writers were skipped on a hint: not one workgroup's walk over millions of rows —
        // the page stays unwritten and the call is replayed (k_enc_layout)
        if ((a.skips & SKIP_DICT_BIG) && p.bigx_off && page < a.n_pages) return;
        // (binary Dict pages k_enc_bin_page did not finish itself: the hint for the next call)
        if (KIND < 0 && threadIdx.x == 0 && a.use_counts) atomicAdd(&a.codec_counts[30], 1u);
    }
    const EncCol c = get_col(a, p.col);
    if (c.ptype == SB_TYPE_NULL) return;
    const bool is_bool = c.ptype == SB_TYPE_BOOLEAN;
    const bool is_bin = c.ptype == SB_TYPE_BINARY || c.ptype == SB_TYPE_LARGE_BINARY;
    if constexpr (KIND == 0) {
        if (!is_bool) return;
    } else if constexpr (KIND == -4) {
        if (c.ptype != SB_TYPE_BINARY) return;
    } else if constexpr (KIND == -8) {
        if (c.ptype != SB_TYPE_LARGE_BINARY) return;
    } else {
        if (is_bool || is_bin || c.width != (uint32_t)KIND) return;
    }
    uint8_t* slot = page_slot(a, c, p);
    const uint64_t N = p.rows;
    uint64_t pos = 0;
    ValidView vv{c.validity, c.validity_bit_offset + p.row0};
    if (c.nullable) {
        uint8_t* bits = def_header(slot, N);
        def_bits_page(bits, ValidView{c.validity, c.validity_bit_offset}, p.row0, N, c.rows);
        pos = def_section_bytes(N);
    }
    uint64_t blen = 0;
    uint8_t* blk = slot + pos;
    if constexpr (KIND == 0)
        blen = emit_bool_page<CODEC>(a, c, p, page, blk, vv, sA, sB, sC, s_w);
    else if constexpr (KIND == -4)
        blen = emit_binary_page<int32_t, CODEC>(a, c, p, page, blk, vv, sA, sB, sC, s_w, s_bin_table, BIN_TABLE);
    else if constexpr (KIND == -8)
        blen = emit_binary_page<int64_t, CODEC>(a, c, p, page, blk, vv, sA, sB, sC, s_w, s_bin_table, BIN_TABLE);
    else
        blen = emit_prim_page<KIND, CODEC>(a, c, p, page, blk, vv, sA, sB, sC, s_w);
    if (threadIdx.x == 0) {
        EncOut o;
        o.length = blen == DICT_FREQ_PENDING ? 1 : (blen ? pos + blen : 0);
        o.out_off = 0;
        o.slot = slot;
        o.codec = (uint32_t)CODEC;
        o.pad = blen == DICT_FREQ_PENDING ? 3 : 0;
        a.outs[page] = o;
    }
}

typedef void (*EncPageKernel)(EncodeArgs);
template <int KIND>
static EncPageKernel enc_page_kernel_for_codec(int32_t codec) {
    switch (codec) {
        case SB_CODEC_RLE:
            return k_enc_emit_pages<KIND, SB_CODEC_RLE>;
        case SB_CODEC_DICT:
            return k_enc_emit_pages<KIND, SB_CODEC_DICT>;
        case SB_CODEC_ONEVALUE:
            return k_enc_emit_pages<KIND, SB_CODEC_ONEVALUE>;
        case SB_CODEC_BITPACKING:
            return k_enc_emit_pages<KIND, SB_CODEC_BITPACKING>;
        case SB_CODEC_DELTA_BITPACKING:
            return k_enc_emit_pages<KIND, SB_CODEC_DELTA_BITPACKING>;
        case SB_CODEC_PATAS:
            if constexpr (KIND == 4 || KIND == 8) return k_enc_emit_pages<KIND, SB_CODEC_PATAS>;
            return nullptr;
    }
    return nullptr;
}
// kind: 1,2,4,8,16,32 / 0 / -4 / -8 (see k_enc_emit_pages)
static EncPageKernel enc_page_kernel(int kind, int32_t codec) {
    switch (kind) {
        case 0:
            return enc_page_kernel_for_codec<0>(codec);
        case 1:
            return enc_page_kernel_for_codec<1>(codec);
        case 2:
            return enc_page_kernel_for_codec<2>(codec);
        case 4:
            return enc_page_kernel_for_codec<4>(codec);
        case 8:
            return enc_page_kernel_for_codec<8>(codec);
        case 16:
            return enc_page_kernel_for_codec<16>(codec);
        case 32:
            return enc_page_kernel_for_codec<32>(codec);
        case -4:
            return enc_page_kernel_for_codec<-4>(codec);
        case -8:
            return enc_page_kernel_for_codec<-8>(codec);
    }
    return nullptr;
}

// Zstd frame made of raw / RLE blocks (valid RFC 8878, accepted by libzstd and hence by the
// reference's decompress_zstd, basic.rs:93-97; no entropy stage yet, so no size reduction
// unless a 128 KiB block is constant).  Byte-identical to oracle/sbo_zstd.cpp zstd_compress.
// Executed by the whole workgroup; returns the frame size.
__device__ uint32_t zstd_store_frame_wg(const uint8_t* src, uint32_t n, uint8_t* dst, uint32_t* s4) {
    if (threadIdx.x == 0) {
        stu32(dst, 0xFD2FB528u);
        dst[4] = (uint8_t)((3u << 6) | (1u << 5));  // single segment, 8-byte frame content size
        stu64(dst + 5, (uint64_t)n);
    }
    uint32_t op = 13, pos = 0;
    do {
        const uint32_t len = min(128u * 1024u, n - pos);
        const bool last = pos + len == n;
        uint32_t diff = 0;
        for (uint32_t i = threadIdx.x; i < len; i += WG) diff |= src[pos + i] != src[pos];
        const bool rle = len > 0 && wg_or32(diff, s4) == 0;
        if (threadIdx.x == 0) {
            const uint32_t hdr = (last ? 1u : 0u) | ((rle ? 1u : 0u) << 1) | (len << 3);
            dst[op] = (uint8_t)hdr;
            dst[op + 1] = (uint8_t)(hdr >> 8);
            dst[op + 2] = (uint8_t)(hdr >> 16);
            if (rle) dst[op + 3] = src[pos];
        }
        op += 3;
        if (rle) {
            op += 1;
        } else {
            wg_copy(dst + op, src + pos, len);
            op += len;
        }
        pos += len;
    } while (pos < n);
    __syncthreads();
    return op;
}

// ------------------------------------------------------------------------------ Freq (primitives)
// integer/freq.rs:34-88, double/freq.rs: top[w] | u32 rb_size | RoaringBitmap of the exception rows |
// compress_integer(exceptions) with Freq forbidden.  k_enc_freq_prep writes everything up to the
// bitmap and turns the exceptions into a virtual page (table entry n_pages + page, its own column
// entry n_cols + page) that goes through k_enc_select / the emit kernels in a second wave;
// k_enc_freq_finish appends the block that wave produced and closes the header.
//   top value: T::default() when >= 90 % of the rows are null (every valid row is an exception), else
//   the most frequent value over ALL slots (null slots count, integer/mod.rs:211) — found by a
//   Boyer-Moore vote, which is exact whenever Freq can be chosen (>= 90 % equal); a page without a
//   majority value (only reachable with force_codec) raises NYI.
//   Roaring portable format [3P roaring 0.10.1 serialize_into]: cookie 12346, container count,
//   (key, cardinality-1) pairs, offsets, then per non-empty 64 Ki-row container a sorted u16 array
//   (cardinality <= 4096) or a 1024 x u64 bitmap.
// Roaring portable serialization of the exception rows of one page (format note above):
// header + containers at `rb`, its byte size also stored at `size_field`; on_exc(row, k) is called for
// the k-th exception.  All threads of the workgroup call this.
constexpr uint32_t FREQ_MAX_CONTAINERS = 1024;  // pages of up to 64 Mi rows (4 KiB of LDS for the cardinalities)
template <class IsExc, class OnExc>
__device__ void freq_roaring(uint64_t N, uint8_t* rb, uint8_t* size_field, uint32_t* sA, uint32_t* s_w, uint32_t* s_card,
                             IsExc is_exc, OnExc on_exc, uint32_t& rb_size_out, uint32_t& n_ex_out) {
    const int t = threadIdx.x, lane = t & 63;
    const uint32_t nc_all = (uint32_t)((N + 65535) / 65536);
    // ---- pass A: cardinality per 64 Ki-row container
    for (uint32_t q = t; q < FREQ_MAX_CONTAINERS; q += WG) s_card[q] = 0;
    __syncthreads();
    for (uint32_t cq = 0; cq < nc_all; cq++) {
        const uint64_t b = (uint64_t)cq * 65536, e = min(N, b + 65536);
        uint32_t cnt = 0;
        for (uint64_t i = b + t; i < e; i += WG) cnt += is_exc(i) ? 1u : 0u;
        const uint32_t tot = wg_sum32(cnt, s_w);
        if (t == 0) s_card[cq] = tot;
    }
    __syncthreads();
    // header
    uint32_t ncne = 0, n_ex = 0;
    for (uint32_t cq = 0; cq < nc_all; cq++) {
        ncne += s_card[cq] ? 1u : 0u;
        n_ex += s_card[cq];
    }
    uint32_t rb_size = 8 + 8 * ncne;
    for (uint32_t cq = 0; cq < nc_all; cq++)
        if (s_card[cq]) rb_size += s_card[cq] > 4096 ? 8192u : 2 * s_card[cq];
    if (t == 0) {
        stu32(size_field, rb_size);
        stu32(rb, 12346u);
        stu32(rb + 4, ncne);
        uint32_t k = 0, off = 8 + 8 * ncne;
        for (uint32_t cq = 0; cq < nc_all; cq++) {
            if (!s_card[cq]) continue;
            *(gptr)(rb + 8 + 4 * k) = (uint8_t)cq;
            *(gptr)(rb + 8 + 4 * k + 1) = (uint8_t)(cq >> 8);
            *(gptr)(rb + 8 + 4 * k + 2) = (uint8_t)(s_card[cq] - 1);
            *(gptr)(rb + 8 + 4 * k + 3) = (uint8_t)((s_card[cq] - 1) >> 8);
            stu32(rb + 8 + 4 * ncne + 4 * k, off);
            off += s_card[cq] > 4096 ? 8192u : 2 * s_card[cq];
            k++;
        }
    }
    // ---- pass B: containers and the exception values
    uint32_t data_off = 8 + 8 * ncne, ex_base = 0;
    for (uint32_t cq = 0; cq < nc_all; cq++) {
        const uint32_t card = s_card[cq];
        if (!card) continue;
        const bool bitmap = card > 4096;
        const uint64_t b = (uint64_t)cq * 65536, e = min(N, b + 65536);
        if (bitmap)  // rows past the end of the page are not visited below
            for (uint32_t i = t; i < 2048; i += WG) stu32(rb + data_off + 4 * i, 0);
        __syncthreads();
        uint32_t carry = 0;
        for (uint64_t cb = b; cb < e; cb += TILE_ROWS) {
            const uint32_t n = (uint32_t)min((uint64_t)TILE_ROWS, e - cb);
            bool f[ROWS_PER_THREAD];
#pragma unroll
            for (int j = 0; j < ROWS_PER_THREAD; j++) {
                const uint32_t r = (uint32_t)t + (uint32_t)j * WG;
                f[j] = r < n && is_exc(cb + r);
                sA[sidx((int)r)] = f[j] ? 1u : 0u;
                if (bitmap) {  // 64 consecutive rows per wave step: one u64 word of the container
                    const uint64_t m = __ballot(f[j]);
                    if (lane == 0 && (r & ~63u) < n) stu64(rb + data_off + ((cb - b + (r & ~63u)) >> 6) * 8, m);
                }
            }
            __syncthreads();
            const uint32_t tot = tile_incl_scan(sA, s_w);
#pragma unroll
            for (int j = 0; j < ROWS_PER_THREAD; j++) {
                const uint32_t r = (uint32_t)t + (uint32_t)j * WG;
                if (!f[j]) continue;
                const uint32_t k = carry + sA[sidx((int)r)] - 1;
                if (!bitmap) {
                    const uint32_t lo16 = (uint32_t)(cb - b) + r;
                    *(gptr)(rb + data_off + 2 * k) = (uint8_t)lo16;
                    *(gptr)(rb + data_off + 2 * k + 1) = (uint8_t)(lo16 >> 8);
                }
                on_exc(cb + r, ex_base + k);
            }
            carry += tot;
            __syncthreads();
        }
        data_off += bitmap ? 8192u : 2 * card;
        ex_base += card;
    }
    rb_size_out = rb_size;
    n_ex_out = n_ex;
}

// The exact top value of a Freq page that has no majority value (only reachable with force_codec; the vote above
// covers every page choose_compressor can send here): first-occurrence dictionary ids for ALL slots (null slots
// count, integer/mod.rs:211), a count per id, arg-max with ties going to the earliest first occurrence (the
// oracle's deterministic stand-in for the reference's HashMap order).  Returns the row of the top value's first
// occurrence, EMPTY if the page has no Dict work area.
template <int W>
struct FreqKeys {  // every slot carries a key; equality of the statistics (canonical float keys)
    const uint8_t* vals;
    uint32_t nk;
    __device__ __forceinline__ bool keyed(uint64_t) const { return true; }
    __device__ __forceinline__ Val<W> key(uint64_t i) const { return stat_key<W>(ld_val<W>(vals + i * W), nk); }
    __device__ __forceinline__ uint32_t hash(uint64_t i) const { return stat_hash<W>(key(i)); }
    __device__ __forceinline__ bool eq(uint64_t a, uint64_t b) const { return bits_eq<W>(key(a), key(b)); }
};
template <class KeyOps>
__device__ uint32_t freq_exact_top(KeyOps ko, uint64_t N, uint32_t* aux, uint64_t aux_words, uint32_t* sA, uint32_t* sB, uint32_t* s_w,
                                   Status* st, uint32_t page) {
    const int t = threadIdx.x;
    if (!aux || !aux_words) return EMPTY;
    uint32_t *idx, *firsts;
    const uint32_t D = dict_build(ko, N, aux, aux_words, &idx, &firsts, sA, sB, s_w, st, page);
    if (D == EMPTY) return EMPTY;
    uint64_t M = 64;
    while (M < 2 * N) M <<= 1;
    uint32_t* cnt = aux + M;  // the builder's F array (one word per row) is free again
    for (uint32_t i = t; i < D; i += WG) cnt[i] = 0;
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    for (uint64_t i = t; i < N; i += WG) atomicAdd(&cnt[idx[i]], 1u);
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    unsigned long long best = 0;  // count << 32 | ~id: highest count, then the smallest id (= earliest first occurrence)
    for (uint32_t i = t; i < D; i += WG) {
        const unsigned long long c = ((unsigned long long)table_load(&cnt[i]) << 32) | (0xFFFFFFFFu - i);
        if (c > best) best = c;
    }
    unsigned long long* red = (unsigned long long*)sA;
    red[t] = best;
    __syncthreads();
    for (int stride = WG / 2; stride > 0; stride >>= 1) {
        if (t < stride && red[t + stride] > red[t]) red[t] = red[t + stride];
        __syncthreads();
    }
    const uint32_t id = 0xFFFFFFFFu - (uint32_t)red[0];
    __syncthreads();
    return firsts[id];
}

template <int W>
__device__ void freq_prep_page(const EncodeArgs& a, EncCol* cols_rw, EncPage* pages_rw, const EncCol& c, const EncPage& p,
                               uint32_t page, uint32_t* lds, uint8_t* dict_slot = nullptr, uint64_t dict_pos = 0,
                               uint64_t ex_cap = ~0ull) {
    // dict_slot != nullptr: `c` / `p` describe the u32 index array of a Dict page whose real slot is dict_slot and
    // whose Dict block starts at dict_pos; the Freq block written here is the index block of that page
    const int t = threadIdx.x;
    const uint64_t N = p.rows;
    const uint8_t* vals = c.values + p.row0 * W;
    const ValidView vv{c.validity, c.validity_bit_offset + p.row0};
    uint32_t* sA = lds;                       // SIDX_WORDS: tile scans
    uint32_t* s_w = lds + SIDX_WORDS;         // 4
    uint32_t* s_card = s_w + 8;               // FREQ_MAX_CONTAINERS
    unsigned long long* s_first = (unsigned long long*)(s_card + FREQ_MAX_CONTAINERS);  // first row of the top value
    uint32_t* s_cnt = (uint32_t*)(s_first + 2);                                          // vote counts (WG)
    Val<W>* s_key = (Val<W>*)(((uintptr_t)(s_cnt + WG) + 15) & ~(uintptr_t)15);            // vote keys (WG)
    uint8_t* slot = page_slot(a, c, p);
    uint64_t pos = 0;
    if (c.nullable) {
        uint8_t* bits = def_header(slot, N);
        def_bits_page(bits, ValidView{c.validity, c.validity_bit_offset}, p.row0, N, c.rows);
        pos = def_section_bytes(N);
    }
    uint8_t* blk = slot + pos;
    const uint32_t nc_all = (uint32_t)((N + 65535) / 65536);
    if (nc_all > FREQ_MAX_CONTAINERS) {
        if (t == 0) raise(a.status, SB_ERR_NYI, page, 540);
        return;
    }
    // canonical key of slot i (the equality of distinct_values / `*val != top_value`)
    auto keyof = [&](uint64_t i) { return stat_key<W>(ld_val<W>(vals + i * W), c.nk); };
    // ---- null share and the vote
    uint32_t nulls = 0;
    Val<W> vk = val_zero<W>();
    uint32_t vn = 0;
    for (uint64_t base = 0; base < N; base += WG) {
        const uint64_t i = base + (uint64_t)t;
        if (i >= N) continue;
        if (!vv.get(i)) nulls++;
        const Val<W> x = keyof(i);
        if (vn == 0) {
            vk = x;
            vn = 1;
        } else if (bits_eq<W>(vk, x)) {
            vn++;
        } else {
            vn--;
        }
    }
    const uint32_t null_count = wg_sum32(nulls, s_w);
    const bool top_is_null = (double)null_count / (double)N >= 0.9;
    Val<W> topk = val_zero<W>();
    Val<W> top = val_zero<W>();
    if (!top_is_null) {
        s_key[t] = vk;
        s_cnt[t] = vn;
        __syncthreads();
        for (int stride = WG / 2; stride > 0; stride >>= 1) {
            if (t < stride) {
                const Val<W> c0 = s_key[t], c1 = s_key[t + stride];
                const uint32_t n0 = s_cnt[t], n1 = s_cnt[t + stride];
                if (n1) {
                    if (n0 == 0) {
                        s_key[t] = c1;
                        s_cnt[t] = n1;
                    } else if (bits_eq<W>(c0, c1)) {
                        s_cnt[t] = n0 + n1;
                    } else if (n1 > n0) {
                        s_key[t] = c1;
                        s_cnt[t] = n1 - n0;
                    } else {
                        s_cnt[t] = n0 - n1;
                    }
                }
            }
            __syncthreads();
        }
        topk = s_key[0];
        __syncthreads();
        // its count and first occurrence (the value written is the first slot's raw bits)
        uint32_t mine = 0;
        unsigned long long first = ~0ull;
        for (uint64_t base = 0; base < N; base += WG) {
            const uint64_t i = base + (uint64_t)t;
            if (i < N && bits_eq<W>(keyof(i), topk)) {
                mine++;
                if (first == ~0ull) first = i;
            }
        }
        const uint32_t mc = wg_sum32(mine, s_w);
        if (t == 0) s_first[0] = ~0ull;
        __syncthreads();
        if (first != ~0ull) atomicMin(&s_first[0], first);
        __syncthreads();
        uint64_t top_row = s_first[0];
        __syncthreads();
        if ((uint64_t)mc * 2 <= N) {  // no majority (host-forced Freq only): exact counts
            uint32_t* sB = s_card + FREQ_MAX_CONTAINERS;  // a second tile array over the vote scratch (no longer needed)
            const uint32_t r = freq_exact_top(FreqKeys<W>{vals, c.nk}, N, p.aux_bytes ? (uint32_t*)(a.scratch + p.aux_off) : nullptr,
                                              p.aux_bytes / 4, sA, sB, s_w, a.status, page);
            if (r == EMPTY) {
                if (t == 0) raise(a.status, SB_ERR_NYI, page, 541);
                return;
            }
            top_row = r;
            topk = keyof(top_row);
        }
        top = ld_val<W>(vals + top_row * W);
        __syncthreads();
    }
    auto is_exc = [&](uint64_t i) { return vv.get(i) && (top_is_null || !bits_eq<W>(keyof(i), topk)); };
    uint8_t* ex = a.scratch + p.ex_off;
    uint32_t rb_size, n_ex;
    if (t == 0) st_val<W>(blk + 9, top);
    freq_roaring(N, blk + 9 + W + 4, blk + 9 + W, sA, s_w, s_card, is_exc,
                 [&](uint64_t row, uint32_t k) {
                     if (((uint64_t)k + 1) * W <= ex_cap) st_val<W>(ex + (uint64_t)k * W, ld_val<W>(vals + row * W));
                 }, rb_size, n_ex);
    if ((uint64_t)n_ex * W > ex_cap) {  // (only a forced nested Freq on data without a dominant index gets here)
        if (t == 0) raise(a.status, SB_ERR_NYI, page, 546);
        return;
    }
    // ---- the virtual page that carries the exceptions through the second wave
    if (t == 0) {
        EncCol vc = c;
        vc.values = ex;
        vc.validity = nullptr;
        vc.offsets = nullptr;
        vc.heads = nullptr;
        vc.out = nullptr;
        vc.values_bit_offset = 0;
        vc.validity_bit_offset = 0;
        vc.out_cap = 0;
        vc.rows = n_ex;
        vc.nullable = 0;
        vc.first_page = a.n_pages + page;
        vc.n_pages = 1;
        cols_rw[page] = vc;
        EncPage vp;
        __builtin_memset(&vp, 0, sizeof vp);
        vp.rows = n_ex;
        vp.slot_off = p.vslot_off;
        vp.aux_off = p.vaux_off;
        vp.aux_bytes = p.vaux_bytes;
        vp.seed = p.seed;
        vp.col = a.n_cols + page;
        vp.codec = a.nested_force >= 0 && !(((a.forbidden | (1u << SB_CODEC_FREQ)) >> a.nested_force) & 1)
                       ? a.nested_force
                       : (a.has_ratio ? CODEC_ON_DEVICE : (int32_t)a.default_compression);
        vp.icodec = -1;
        vp.depth = p.depth + 1;
        vp.forb_extra = p.forb_extra | (1u << SB_CODEC_FREQ);
        pages_rw[page] = vp;
        EncOut o;
        o.length = pos + 9 + W + 4 + rb_size;  // so far; k_enc_freq_finish adds the nested block
        o.out_off = 0;
        o.slot = slot;
        o.codec = SB_CODEC_FREQ;
        o.pad = 0;
        if (dict_slot) {  // the page stays a Dict page in waiting; its length so far counts from the real slot
            o.length += dict_pos + 9;
            o.slot = dict_slot;
            o.codec = SB_CODEC_DICT;
            o.pad = 3;
        }
        a.outs[page] = o;
    }
}

// Binary / Utf8 Freq page (binary/freq.rs:44-101): u64 top_len | top | u32 rb_size | Roaring | per exception
// `u64 len | bytes` (plain, no nested block) — written completely here.  The top value is the majority value
// over ALL slots (binary/mod.rs:265-291 counts null slots too), empty when >= 90 % of the rows are null.
template <class O>
__device__ void freq_prep_bin(const EncodeArgs& a, const EncCol& c, const EncPage& p, uint32_t page, uint32_t* lds) {
    const int t = threadIdx.x;
    const uint64_t N = p.rows;
    const ValidView vv{c.validity, c.validity_bit_offset + p.row0};
    uint32_t* sA = lds;
    uint32_t* s_w = lds + SIDX_WORDS;
    uint32_t* s_card = s_w + 8;
    uint32_t* s_vote = s_card + FREQ_MAX_CONTAINERS;  // 2 * WG + 4
    uint8_t* slot = page_slot(a, c, p);
    uint64_t pos = 0;
    if (c.nullable) {
        uint8_t* bits = def_header(slot, N);
        def_bits_page(bits, ValidView{c.validity, c.validity_bit_offset}, p.row0, N, c.rows);
        pos = def_section_bytes(N);
    }
    uint8_t* blk = slot + pos;
    if ((N + 65535) / 65536 > FREQ_MAX_CONTAINERS) {
        if (t == 0) raise(a.status, SB_ERR_NYI, page, 540);
        return;
    }
    const BinKeys<O> bk{c.offsets + p.row0 * sizeof(O), c.values, vv};
    uint32_t nulls = 0;
    for (uint64_t i = t; i < N; i += WG) nulls += vv.get(i) ? 0u : 1u;
    const uint32_t null_count = wg_sum32(nulls, s_w);
    const bool top_is_null = (double)null_count / (double)N >= 0.9;
    uint32_t cand = 0;
    uint64_t tb = 0, te = 0;
    if (!top_is_null) {
        uint32_t cd = 0, cnt = 0;  // (always a row of the page, see majority_count)
        for (uint64_t base = 0; base < N; base += WG) {
            const uint64_t i = base + (uint64_t)t;
            if (i >= N) continue;
            if (cnt == 0) {
                cd = (uint32_t)i;
                cnt = 1;
            } else if (bk.eq(cd, i)) {
                cnt++;
            } else {
                cnt--;
            }
        }
        s_vote[t] = cd;
        s_vote[WG + t] = cnt;
        __syncthreads();
        for (int stride = WG / 2; stride > 0; stride >>= 1) {
            if (t < stride) {
                const uint32_t c0 = s_vote[t], n0 = s_vote[WG + t], c1 = s_vote[t + stride], n1 = s_vote[WG + t + stride];
                uint32_t cc = c0, n = n0;
                if (n1) {
                    if (n0 == 0) {
                        cc = c1;
                        n = n1;
                    } else if (bk.eq(c0, c1)) {
                        n = n0 + n1;
                    } else if (n1 > n0) {
                        cc = c1;
                        n = n1 - n0;
                    } else {
                        n = n0 - n1;
                    }
                }
                s_vote[t] = cc;
                s_vote[WG + t] = n;
            }
            __syncthreads();
        }
        const bool have = s_vote[WG] != 0;
        cand = s_vote[0];
        __syncthreads();
        uint32_t mine = 0;
        if (have)
            for (uint64_t base = 0; base < N; base += WG) {
                const uint64_t i = base + (uint64_t)t;
                if (i < N && bk.eq(cand, i)) mine++;
            }
        const uint32_t mc = wg_sum32(mine, s_w);
        if ((uint64_t)mc * 2 <= N) {  // no majority (host-forced Freq only): exact counts over all slots
            uint32_t* sB = s_card + FREQ_MAX_CONTAINERS;
            const BinKeys<O> all{c.offsets + p.row0 * sizeof(O), c.values, ValidView{nullptr, 0}};
            const uint32_t r = freq_exact_top(all, N, p.aux_bytes ? (uint32_t*)(a.scratch + p.aux_off) : nullptr, p.aux_bytes / 4, sA, sB,
                                              s_w, a.status, page);
            if (r == EMPTY) {
                if (t == 0) raise(a.status, SB_ERR_NYI, page, 541);
                return;
            }
            cand = r;
        }
        tb = bk.beg(cand);
        te = bk.beg((uint64_t)cand + 1);
    }
    const uint64_t top_len = te - tb;
    if (t == 0) stu64(blk + 9, top_len);
    wg_copy(blk + 17, c.values + tb, top_len);
    auto is_exc = [&](uint64_t i) { return vv.get(i) && (top_is_null || !bk.eq(cand, i)); };
    uint32_t rb_size, n_ex;
    uint8_t* rb = blk + 9 + 8 + top_len + 4;
    freq_roaring(N, rb, rb - 4, sA, s_w, s_card, is_exc, [](uint64_t, uint32_t) {}, rb_size, n_ex);
    // exception records, in row order: position = scan of (8 + len) over the exception rows
    uint8_t* rec = rb + rb_size;
    uint64_t carry = 0;
    for (uint64_t cb = 0; cb < N; cb += TILE_ROWS) {
        const uint32_t n = (uint32_t)min((uint64_t)TILE_ROWS, N - cb);
        __syncthreads();
        for (uint32_t r = t; r < TILE_ROWS; r += WG) {
            uint32_t len = 0;
            if (r < n && is_exc(cb + r)) len = (uint32_t)(bk.beg(cb + r + 1) - bk.beg(cb + r)) + 8;
            sA[sidx((int)r)] = len;
        }
        __syncthreads();
        const uint32_t tot = tile_incl_scan(sA, s_w);
        for (uint32_t r = t; r < n; r += WG) {
            const uint32_t endb = sA[sidx((int)r)];
            const uint32_t prev = r ? sA[sidx((int)r - 1)] : 0;
            if (endb == prev) continue;
            const uint64_t b = bk.beg(cb + r), len = endb - prev - 8;
            uint8_t* d = rec + carry + prev;
            stu64(d, len);
            for (uint64_t q = 0; q < len; q++) d[8 + q] = c.values[b + q];
        }
        carry += tot;
    }
    __syncthreads();
    if (t == 0) {
        const uint64_t body = 8 + top_len + 4 + rb_size + carry;
        put_hdr9(blk, SB_CODEC_FREQ, (uint32_t)body, (uint32_t)c.values_len_total);  // binary/mod.rs:83-88
        EncOut o;
        o.length = pos + 9 + body;
        o.out_off = 0;
        o.slot = slot;
        o.codec = SB_CODEC_FREQ;
        o.pad = 2;  // complete: k_enc_freq_finish has nothing to append
        a.outs[page] = o;
    }
}

#include "sb_freq_big.h"

__global__ void __launch_bounds__(WG, 2) k_enc_freq_prep(EncodeArgs a_in, EncCol* cols_rw, EncPage* pages_rw) {
    // tile array | s_w | container counts | vote scratch (a second tile array in the exact-count path)
    __shared__ __attribute__((aligned(16))) uint32_t lds[SIDX_WORDS + 8 + FREQ_MAX_CONTAINERS + SIDX_WORDS + 16];
    static_assert(SIDX_WORDS + 16 >= 4 + WG + 8 + 8 * WG, "vote scratch");
    if (*a_in.freq_count == 0) return;
    // The page functions below are real calls that take the arguments by reference, so the kernel keeps a copy of them
    // in private memory.  Made from `a` itself that copy is written in the prologue, before the return above: 192 bytes
    // per lane, 49 MB per launch of a batch without a Freq page (profiles/r04_c2_pmc_traffic.json of the build before).
    const EncodeArgs a = a_in;
  for (uint32_t page = blockIdx.x; page < a.n_pages; page += gridDim.x) {
    __syncthreads();
    const EncPage p = get_page(a, page);
    const int32_t pcodec = codec_of(a, p, page);
    if (pcodec == SB_CODEC_DICT) {  // a Dict page whose u32 indices are to be a Freq block (emit_prim_page<Dict>)
        const EncOut o0 = a.outs[page];
        if (o0.codec != SB_CODEC_DICT || o0.pad != 3) continue;
        const EncCol c = get_col(a, p.col);
        const unsigned long long* rec = (const unsigned long long*)(a.scratch + p.vaux_off);
        EncCol ic = c;   // the index array as a column of its own
        ic.values = (const uint8_t*)(uintptr_t)rec[0];
        ic.validity = nullptr;
        ic.validity_bit_offset = 0;
        ic.nullable = 0;
        ic.width = 4;
        ic.ptype = SB_TYPE_UINT32;
        ic.fkind = 0;
        ic.nk = NK_UNSIGNED;
        ic.rows = p.rows;
        EncPage ip = p;
        const uint64_t dpos = c.nullable ? def_section_bytes(p.rows) : 0;
        uint8_t* dslot = page_slot(a, c, p);
        ip.row0 = 0;
        ip.direct = 0;
        ip.slot_off = (uint64_t)(dslot + dpos + 9 - a.scratch);
        ip.depth = p.depth + 1;                              // compress_integer::<u32> inside Dict (dict.rs:60-62)
        ip.forb_extra = p.forb_extra | (1u << SB_CODEC_DICT);
        ip.aux_bytes = 0;
        ip.vaux_bytes = 0;                                   // (the area holds the record read above)
        const bool cbin = c.ptype == SB_TYPE_BINARY || c.ptype == SB_TYPE_LARGE_BINARY;  // (exception area: rows/2 + 1 indices)
        freq_prep_page<4>(a, cols_rw, pages_rw, ic, ip, page, lds, dslot, dpos, cbin ? (p.rows / 2 + 1) * 4 : p.rows * c.width);
        continue;
    }
    if (pcodec != SB_CODEC_FREQ) continue;
    if (a.outs[page].length != 0 && a.outs[page].codec == SB_CODEC_FREQ) continue;   // a long page, prepared container-parallel (sb_freq_big.h)
    if ((a.skips & SKIP_FREQ_BIG) && p.bigx_off && p.rows >= SEL_BIG_ROWS) continue;   // ... whose kernels were skipped on a hint: replay (k_enc_layout)
    const EncCol c = get_col(a, p.col);
    if (c.ptype == SB_TYPE_BOOLEAN || c.ptype == SB_TYPE_NULL || p.rows == 0) {
        if (threadIdx.x == 0) raise(a.status, SB_ERR_OUT_OF_SPEC, page, 542);  // no Freq for booleans upstream
        continue;
    }
    if (c.ptype == SB_TYPE_BINARY) {
        freq_prep_bin<int32_t>(a, c, p, page, lds);
        continue;
    }
    if (c.ptype == SB_TYPE_LARGE_BINARY) {
        freq_prep_bin<int64_t>(a, c, p, page, lds);
        continue;
    }
    switch (c.width) {
        case 1:
            freq_prep_page<1>(a, cols_rw, pages_rw, c, p, page, lds);
            break;
        case 2:
            freq_prep_page<2>(a, cols_rw, pages_rw, c, p, page, lds);
            break;
        case 4:
            freq_prep_page<4>(a, cols_rw, pages_rw, c, p, page, lds);
            break;
        case 8:
            freq_prep_page<8>(a, cols_rw, pages_rw, c, p, page, lds);
            break;
        case 16:
            freq_prep_page<16>(a, cols_rw, pages_rw, c, p, page, lds);
            break;
        default:
            freq_prep_page<32>(a, cols_rw, pages_rw, c, p, page, lds);
            break;
    }
  }
}

// The exceptions block of a Freq page (a virtual page): selection and encoding in ONE kernel, every
// codec behind a runtime switch.  These blocks are small (<= 10 % of a page when Freq was chosen), so
// the register cost of the monolithic shape does not matter, and a batch without Freq pages pays for
// one near-empty launch per value width instead of one per (width, codec).
template <int W>
__device__ void enc_nested_block(const EncodeArgs& a, uint32_t page, uint32_t* lds, uint32_t* s_w, uint32_t* s_misc2) {
    uint32_t& s_sz = s_misc2[0];
    uint32_t& s_codec = s_misc2[1];
    uint32_t *sA = lds, *sB = lds + SIDX_WORDS, *sC = lds + 2 * SIDX_WORDS;
    {  // the table entry of a virtual page is only valid when k_enc_freq_prep wrote it for its real page
        const uint32_t q = page - a.n_pages;
        const EncPage rp = a.pages[q];
        const int32_t rc = codec_of(a, rp, q);
        const EncOut ro = a.outs[q];
        if (rc == SB_CODEC_DICT) {  // Freq-coded indices of a Dict page: valid once prep has written past the headers
            if (ro.codec != SB_CODEC_DICT || ro.pad != 3 || ro.length <= 1) return;
        } else {
            if (rc != SB_CODEC_FREQ) return;
            if (ro.length == 0 || ro.codec != SB_CODEC_FREQ || ro.pad == 2) return;  // prep raised / binary page (no nested block)
        }
    }
    {   // written by the section- / tile-parallel kernels already (sb_dict_big.h: virtual pages of >= VBIG_ROWS rows)
        const EncOut vo = a.outs[page];
        if (vo.length != 0 && (vo.pad == 1 || vo.pad == VPAD_PLANNED)) return;
    }
    const EncPage p = get_page(a, page);
    const EncCol c = get_col(a, p.col);
    if (c.width != (uint32_t)W) return;
    const uint64_t N = p.rows;
    const uint8_t* vals = c.values;
    const ValidView vv{nullptr, 0};
    int32_t codec = p.codec;
    if (codec == CODEC_ON_DEVICE) {
        SelectOpts so{a.ratio, a.has_ratio, a.forbidden | p.forb_extra, a.default_compression, -1, p.seed, p.depth};
        SelScratch sc{sA, sA + SEL_LDS_SLOTS, (uint8_t*)(sA + SEL_LDS_SLOTS + 2 * WG + 16),
                      p.aux_bytes ? (uint32_t*)(a.scratch + p.aux_off) : nullptr, 0};
        if (sc.gtab) {
            uint64_t M = 64;
            while (M < 2 * N) M <<= 1;
            sc.gslots = M;
        }
        const uint32_t ch = choose_prim<W>([=](uint64_t i) { return ld_val<W>(vals + i * W); }, vv, N, c.nk, so, sc);
        __syncthreads();
        if (threadIdx.x == 0) s_codec = ch;
        __syncthreads();
        codec = (int32_t)s_codec;
        __syncthreads();
    }
    uint8_t* blk = a.scratch + p.slot_off;
    uint64_t blen = 0;
    switch (codec) {
        case SB_CODEC_NONE:
            wg_copy(blk + 9, vals, N * W);
            if (threadIdx.x == 0) put_hdr9(blk, SB_CODEC_NONE, (uint32_t)(N * W), (uint32_t)(N * W));
            blen = 9 + N * W;
            break;
        case SB_CODEC_LZ4:
        case SB_CODEC_ZSTD:
        case SB_CODEC_SNAPPY: {
            uint32_t sz;
            __syncthreads();
            if (codec == SB_CODEC_ZSTD) {
                sz = zstd_store_frame_wg(vals, (uint32_t)(N * W), blk + 9, s_w);
            } else if (codec == SB_CODEC_SNAPPY) {
                sz = snappy_compress_block_wg(vals, (uint32_t)(N * W), blk + 9, sA, &s_sz);
            } else {
                uint32_t z = 0;
                if (threadIdx.x < 64) z = lz4_compress_block(vals, (uint32_t)(N * W), blk + 9, sA, a.flags);
                if (threadIdx.x == 0) s_sz = z;
                __syncthreads();
                sz = s_sz;
            }
            if (threadIdx.x == 0) put_hdr9(blk, (uint32_t)codec, sz, (uint32_t)(N * W));
            blen = 9 + sz;
            break;
        }
        case SB_CODEC_RLE:
            blen = emit_prim_page<W, SB_CODEC_RLE>(a, c, p, page, blk, vv, sA, sB, sC, s_w);
            break;
        case SB_CODEC_DICT:
            blen = emit_prim_page<W, SB_CODEC_DICT>(a, c, p, page, blk, vv, sA, sB, sC, s_w);
            break;
        case SB_CODEC_ONEVALUE:
            blen = emit_prim_page<W, SB_CODEC_ONEVALUE>(a, c, p, page, blk, vv, sA, sB, sC, s_w);
            break;
        case SB_CODEC_BITPACKING:
            if constexpr (W == 4) blen = emit_prim_page<W, SB_CODEC_BITPACKING>(a, c, p, page, blk, vv, sA, sB, sC, s_w);
            break;
        case SB_CODEC_DELTA_BITPACKING:
            if constexpr (W == 4) blen = emit_prim_page<W, SB_CODEC_DELTA_BITPACKING>(a, c, p, page, blk, vv, sA, sB, sC, s_w);
            break;
        case SB_CODEC_PATAS:
            if constexpr (W == 4 || W == 8) blen = emit_prim_page<W, SB_CODEC_PATAS>(a, c, p, page, blk, vv, sA, sB, sC, s_w);
            break;
        default:
            if (threadIdx.x == 0) raise(a.status, SB_ERR_NYI, page, 545);
            break;
    }
    if (threadIdx.x == 0) {
        EncOut o{blen, 0, blk, (uint32_t)codec, 0};
        a.outs[page] = o;
    }
}
// a few hundred workgroups walk all virtual pages: a batch without Freq pages costs a short launch
template <int W>
__global__ void __launch_bounds__(WG) k_enc_nested(EncodeArgs a_in) {
    // three tile arrays for the emitters; the selector lays its hash set, misc words and sample area over them
    constexpr int SEL_WORDS = SEL_LDS_SLOTS + 2 * WG + 16 + (SAMPLE_CAP * (W + 1) + 16 + 3) / 4;
    __shared__ __attribute__((aligned(16))) uint32_t lds[SEL_WORDS > 3 * SIDX_WORDS ? SEL_WORDS : 3 * SIDX_WORDS];
    __shared__ uint32_t s_w[4];
    __shared__ uint32_t s_misc2[2];
    if (*a_in.freq_count == 0) return;
    const EncodeArgs a = a_in;   // (the private copy the block functions are called with: after the return, see k_enc_freq_prep)
    for (uint32_t q = blockIdx.x; q < a.n_pages; q += gridDim.x) {
        enc_nested_block<W>(a, a.n_pages + q, lds, s_w, s_misc2);
        __syncthreads();
    }
}

__global__ void __launch_bounds__(WG) k_enc_freq_finish(EncodeArgs a) {
    __shared__ uint32_t s_ent[SIDX_WORDS];  // entry positions of a binary dictionary
    __shared__ uint32_t s_w[4];
    if (*a.freq_count == 0) return;
    const uint32_t page = blockIdx.x;
    const EncPage p = get_page(a, page);
    const int32_t pcodec = codec_of(a, p, page);
    if (pcodec == SB_CODEC_DICT) {  // Dict page with Freq-coded indices: exceptions block, then `u32 D | entries`, then both headers
        const EncOut o = a.outs[page];
        if (o.codec != SB_CODEC_DICT || o.pad != 3) return;
        const EncOut vo = a.outs[a.n_pages + page];
        const EncCol c = get_col(a, p.col);
        const unsigned long long* rec = (const unsigned long long*)(a.scratch + p.vaux_off);
        const uint32_t* firsts = (const uint32_t*)(uintptr_t)rec[1];
        const uint32_t D = (uint32_t)rec[2];
        const uint32_t W = c.width;
        const uint64_t total = o.length + vo.length + 4 + (uint64_t)D * W;
        if (o.length <= 1 || vo.length == 0 || (p.slot_cap && total > p.slot_cap)) {
            if (threadIdx.x == 0) {
                raise(a.status, SB_ERR_NYI, page, 547);
                EncOut z = o;
                z.length = 0;
                z.pad = 0;
                a.outs[page] = z;
            }
            return;
        }
        wg_copy(o.slot + o.length, vo.slot, vo.length);
        uint8_t* q = o.slot + o.length + vo.length;
        if (c.ptype == SB_TYPE_BINARY || c.ptype == SB_TYPE_LARGE_BINARY) {
            // entries `u64 len | bytes` in dictionary order (binary/dict.rs:55-93), positions = scan of (8 + len)
            const uint32_t ow = c.ptype == SB_TYPE_BINARY ? 4u : 8u;
            const uint8_t* offs = c.offsets + p.row0 * ow;
            auto beg = [=](uint64_t i) { return ow == 4 ? (uint64_t)ldu32(offs + i * 4) : ldu64(offs + i * 8); };
            uint64_t epos = 0;
            for (uint32_t kb = 0; kb < D; kb += TILE_ROWS) {
                const uint32_t n = min((uint32_t)TILE_ROWS, D - kb);
                for (uint32_t i = threadIdx.x; i < TILE_ROWS; i += WG) {
                    uint32_t len = 0;
                    if (i < n) {
                        const uint64_t r = firsts[kb + i];
                        len = (uint32_t)(beg(r + 1) - beg(r)) + 8;
                    }
                    s_ent[sidx((int)i)] = len;
                }
                __syncthreads();
                const uint32_t tot = tile_incl_scan(s_ent, s_w);
                for (uint32_t i = threadIdx.x; i < n; i += WG) {
                    const uint64_t r = firsts[kb + i];
                    const uint64_t b = beg(r), e = beg(r + 1);
                    uint8_t* d = q + 4 + epos + s_ent[sidx((int)i)] - (e - b) - 8;
                    stu64(d, e - b);
                    for (uint64_t k = 0; k < e - b; k++) d[8 + k] = c.values[b + k];
                }
                epos += tot;
                __syncthreads();
            }
            if (threadIdx.x == 0) {
                stu32(q, D);
                const uint64_t pos = c.nullable ? def_section_bytes(p.rows) : 0;
                const uint64_t btotal = o.length + vo.length + 4 + epos;
                put_hdr9(o.slot + pos + 9, SB_CODEC_FREQ, (uint32_t)(o.length + vo.length - pos - 18), (uint32_t)(p.rows * 4));
                put_hdr9(o.slot + pos, SB_CODEC_DICT, (uint32_t)(btotal - pos - 9), (uint32_t)c.values_len_total);  // binary/mod.rs:88
                EncOut z = o;
                z.length = btotal;
                z.pad = 0;
                a.outs[page] = z;
            }
            return;
        }
        const uint8_t* vals = c.values + p.row0 * W;
        const bool lead_null = c.validity && !bit_at(c.validity, c.validity_bit_offset + p.row0);
        for (uint32_t k = threadIdx.x; k < D; k += WG) {
            const uint32_t r = firsts[k];
            for (uint32_t b = 0; b < W; b++) q[4 + (uint64_t)k * W + b] = (r == 0 && lead_null) ? (uint8_t)0 : vals[(uint64_t)r * W + b];  // dict.rs:46-50
        }
        if (threadIdx.x == 0) {
            stu32(q, D);
            const uint64_t pos = c.nullable ? def_section_bytes(p.rows) : 0;
            put_hdr9(o.slot + pos + 9, SB_CODEC_FREQ, (uint32_t)(o.length + vo.length - pos - 18), (uint32_t)(p.rows * 4));
            put_hdr9(o.slot + pos, SB_CODEC_DICT, (uint32_t)(total - pos - 9), (uint32_t)(p.rows * W));
            EncOut z = o;
            z.length = total;
            z.pad = 0;
            a.outs[page] = z;
        }
        return;
    }
    if (pcodec != SB_CODEC_FREQ) return;
    const EncOut o = a.outs[page];
    if (o.codec != SB_CODEC_FREQ || o.length == 0 || o.pad == 2) return;  // prep raised / binary page already complete
    const EncOut vo = a.outs[a.n_pages + page];
    const EncCol c = get_col(a, p.col);
    if (vo.length == 0) {
        if (threadIdx.x == 0) {
            raise(a.status, SB_ERR_NYI, page, 543);  // the exceptions block could not be encoded on the device
            EncOut z = o;
            z.length = 0;
            a.outs[page] = z;
        }
        return;
    }
    if (o.length + vo.length > p.slot_cap) {  // only reachable when Freq is forced on a page that is mostly exceptions
        if (threadIdx.x == 0) {
            raise(a.status, SB_ERR_NYI, page, 544);
            EncOut z = o;
            z.length = 0;
            a.outs[page] = z;
        }
        return;
    }
    wg_copy(o.slot + o.length, vo.slot, vo.length);
    if (threadIdx.x == 0) {
        const uint64_t pos = c.nullable ? def_section_bytes(p.rows) : 0;
        const uint64_t total = o.length + vo.length;
        put_hdr9(o.slot + pos, SB_CODEC_FREQ, (uint32_t)(total - pos - 9), (uint32_t)(p.rows * c.width));
        EncOut z = o;
        z.length = total;
        a.outs[page] = z;
    }
}

// pages whose codec is LZ4 (CommonCompression::Lz4 as the default, or chosen by the selector):
// def levels + hdr9 + one LZ4 block (binary: offsets block + values block), one workgroup per page
// Two instances: ZSTD = false handles LZ4 / Snappy pages (and Zstd pages without encoder scratch: stored frames), ZSTD =
// true the pages that go through the Zstd encoder — its Huffman / FSE stages need ~160 VGPRs, which would cut the
// LZ4 instance from 6 to 3 workgroups per CU.
#ifndef SB_LZ4_HB
#define SB_LZ4_HB 12
#endif
// the block(s) a Basic page compresses: bitmap bytes / re-based offsets / values (first block), a binary page's value bytes
struct LzBlocks {
    const uint8_t *src_a, *src_b;
    uint32_t n_a, n_b;
    uint64_t first;     // binary: offsets[row0]
    bool stage_a;       // the first block is built in the page's staging area (p.aux_off)
};
__device__ __forceinline__ LzBlocks lz4_page_blocks(const EncodeArgs& a, const EncCol& c, const EncPage& p) {
    LzBlocks b{nullptr, nullptr, 0, 0, 0, false};
    const uint64_t N = p.rows;
    uint8_t* stage = a.scratch + p.aux_off;
    if (c.ptype == SB_TYPE_BOOLEAN) {
        const uint64_t boff = c.values_bit_offset + p.row0;
        b.stage_a = (boff & 7) != 0;
        b.src_a = b.stage_a ? stage : c.values + (boff >> 3);
        b.n_a = (uint32_t)((N + 7) / 8);
    } else if (c.ptype == SB_TYPE_BINARY || c.ptype == SB_TYPE_LARGE_BINARY) {
        const uint32_t ow = c.width;
        const uint8_t* offs = c.offsets + p.row0 * ow;
        b.first = ow == 4 ? (uint64_t)ldu32(offs) : ldu64(offs);
        const uint64_t last = ow == 4 ? (uint64_t)ldu32(offs + N * 4) : ldu64(offs + N * 8);
        b.stage_a = true;
        b.src_a = stage;
        b.n_a = (uint32_t)((N + 1) * ow);
        b.src_b = c.values + b.first;
        b.n_b = (uint32_t)(last - b.first);
    } else {
        b.src_a = c.values + p.row0 * c.width;
        b.n_a = (uint32_t)(N * c.width);
    }
    return b;
}
// chunk bytes of a block of n bytes: the call's piece size (Zstd), or by the block's length (LZ4, Snappy)
__device__ __forceinline__ uint32_t lzc_chunk_of(const EncodeArgs& a, uint32_t n) { return a.lzc_codec == SB_CODEC_ZSTD ? a.lzc_chunk : lz4_chunk_bytes(n); }
// chunk by chunk (k_enc_lz4_plan / _chunks / _stitch) or as one block by one wave (k_enc_emit_lz4)?  A pure function of
// the page, so that every kernel decides alike.
__device__ __forceinline__ bool lz4_page_chunked(const EncodeArgs& a, int32_t bc, uint32_t page, const LzBlocks& b, uint64_t zst_off = ~0ull) {
    if (!a.lzc_plan || page >= a.n_pages || max(b.n_a, b.n_b) <= a.lzc_chunk || bc != a.lzc_codec) return false;
    if (bc == SB_CODEC_LZ4) return !(a.flags & SB_WRITE_LZ4_EXACT);
    if (bc == SB_CODEC_SNAPPY) return true;
    return bc == SB_CODEC_ZSTD && a.zpar_scratch && zst_off != ~0ull;
}

template <bool ZSTD>
__global__ void __launch_bounds__(WG) k_enc_emit_lz4(EncodeArgs a) {
    // one LDS area: the 16 KiB the stored-Zstd / Snappy / exact-LZ4 paths use, or the matcher's table + ring (+ Zstd entropy tables)
    __shared__ union {
        uint32_t tab[4096];
        Lz4EncLds<SB_LZ4_HB, 13> lz;
        typename std::conditional<ZSTD, ZEncLds, uint32_t>::type ze;
    } sh;
    uint32_t* const tab = sh.tab;
    __shared__ uint32_t s_sz;
    if (a.use_counts && a.codec_counts[SB_CODEC_LZ4] + a.codec_counts[SB_CODEC_ZSTD] + a.codec_counts[SB_CODEC_SNAPPY] == 0) return;
    const uint32_t page = spread_block(blockIdx.x, gridDim.x) + a.page_base;
    const EncPage p = get_page(a, page);
    const int32_t bc = codec_of(a, p, page);
    if (bc != SB_CODEC_LZ4 && bc != SB_CODEC_ZSTD && bc != SB_CODEC_SNAPPY) return;
    if (ZSTD != (bc == SB_CODEC_ZSTD && p.zst_off != ~0ull)) return;   // the other instance's page
    const EncCol c = get_col(a, p.col);
    if (c.ptype == SB_TYPE_NULL) return;
    if (a.lzc_plan && lz4_page_chunked(a, bc, page, lz4_page_blocks(a, c, p), p.zst_off)) return;   // compressed chunk by chunk
    uint8_t* slot = page_slot(a, c, p);
    const uint64_t N = p.rows;
    uint64_t pos = 0;
    if (c.nullable) {
        uint8_t* bits = def_header(slot, N);
        def_bits_page(bits, ValidView{c.validity, c.validity_bit_offset}, p.row0, N, c.rows);
        pos = def_section_bytes(N);
    }
    uint8_t* blk = slot + pos;
    uint8_t* stage = a.scratch + p.aux_off;
    const bool is_bin = c.ptype == SB_TYPE_BINARY || c.ptype == SB_TYPE_LARGE_BINARY;
    auto compress = [&](const uint8_t* src, uint32_t n, uint8_t* dst) -> uint32_t {
        __syncthreads();
        if constexpr (ZSTD) {
            uint32_t zs = 0;
            if (threadIdx.x < 64) zs = zstd_compress_wave(src, n, dst, sh.ze, a.scratch + p.zst_off);
            if (threadIdx.x == 0) s_sz = zs;
            __syncthreads();
            return s_sz;
        } else {
            if (bc == SB_CODEC_ZSTD) return zstd_store_frame_wg(src, n, dst, tab);
        }
        uint32_t sz = 0;
        if (threadIdx.x < 64)
            sz = bc == SB_CODEC_SNAPPY ? snappy_compress_wave<SB_LZ4_HB, 13>(src, n, dst, sh.lz) : (a.flags & SB_WRITE_LZ4_EXACT) ? lz4_compress_wave(src, n, dst, tab) : lz4_compress_wave_fast<SB_LZ4_HB, 13>(src, n, dst, sh.lz);
        if (threadIdx.x == 0) s_sz = sz;
        __syncthreads();
        return s_sz;
    };
    uint64_t length;
    if (c.ptype == SB_TYPE_BOOLEAN) {  // boolean/mod.rs:44-54
        const uint64_t boff = c.values_bit_offset + p.row0, nbytes = (N + 7) / 8;
        const uint8_t* src = c.values + (boff >> 3);
        if (boff & 7) {
            const uint64_t total_bits = c.values_bit_offset + c.rows;
            for (uint64_t b = threadIdx.x; b < nbytes; b += WG) {
                uint32_t w = bits32(c.values, boff + b * 8, total_bits);
                const uint64_t nb = min((uint64_t)8, N - b * 8);
                if (nb < 8) w &= (1u << nb) - 1;
                stage[b] = (uint8_t)w;
            }
            src = stage;
        }
        const uint32_t sz = compress(src, (uint32_t)nbytes, blk + 9);
        if (threadIdx.x == 0) put_hdr9(blk, (uint32_t)bc, sz, (uint32_t)N);
        length = pos + 9 + sz;
    } else if (is_bin) {  // binary/mod.rs:42-81
        const uint32_t ow = c.width;
        const uint8_t* offs = c.offsets + p.row0 * ow;
        const uint64_t first = ow == 4 ? (uint64_t)ldu32(offs) : ldu64(offs);
        const uint64_t last = ow == 4 ? (uint64_t)ldu32(offs + N * 4) : ldu64(offs + N * 8);
        const uint64_t obytes = (N + 1) * ow, vbytes = last - first;
        for (uint64_t i = threadIdx.x; i <= N; i += WG) {
            if (ow == 4)
                stu32(stage + i * 4, (uint32_t)(ldu32(offs + i * 4) - first));
            else
                stu64(stage + i * 8, ldu64(offs + i * 8) - first);
        }
        const uint32_t s1 = compress(stage, (uint32_t)obytes, blk + 9);
        if (threadIdx.x == 0) put_hdr9(blk, (uint32_t)bc, s1, (uint32_t)obytes);
        uint8_t* b2 = blk + 9 + s1;
        const uint32_t s2 = compress(c.values + first, (uint32_t)vbytes, b2 + 9);
        if (threadIdx.x == 0) put_hdr9(b2, (uint32_t)bc, s2, (uint32_t)vbytes);
        length = pos + 9 + s1 + 9 + s2;
    } else {
        const uint32_t w = c.width;
        const uint32_t sz = compress(c.values + p.row0 * w, (uint32_t)(N * w), blk + 9);
        if (threadIdx.x == 0) put_hdr9(blk, (uint32_t)bc, sz, (uint32_t)(N * w));
        length = pos + 9 + sz;
    }
    if (threadIdx.x == 0) {
        EncOut o{length, 0, slot, (uint32_t)bc, 0};
        a.outs[page] = o;
    }
}

// ---- LZ4 blocks of more than one chunk (64 KiB): a page is one LZ4 block per buffer in the format, and one wave per block
// leaves a 1 MiB page to a single latency-bound wave while the rest of the chip idles.  The block is cut into chunks
// compressed independently (no match crosses a chunk border, so the output is a valid block whatever the order), one
// wave per chunk, into a pool slot each; the stitch pass joins them: a chunk's trailing literals — which no sequence
// holds — become part of the next chunk's first sequence, whose token and length bytes are rewritten.
// plan: def levels + staged first block + the chunk list (one workgroup per page).
__global__ void __launch_bounds__(WG) k_enc_lz4_plan(EncodeArgs a) {
    __shared__ uint32_t s_base;
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) a.lzc_count[1] = 0;   // the ticket of k_enc_zstd_chunks' waves
    if (a.use_counts && a.codec_counts[a.lzc_codec & 31] == 0) return;
    const uint32_t page = spread_block(blockIdx.x, gridDim.x) + a.page_base;
    const EncPage p = get_page(a, page);
    const int32_t bc = codec_of(a, p, page);
    if (bc != a.lzc_codec) return;
    const EncCol c = get_col(a, p.col);
    if (c.ptype == SB_TYPE_NULL) return;
    const LzBlocks b = lz4_page_blocks(a, c, p);
    if (!lz4_page_chunked(a, bc, page, b, p.zst_off)) return;
    const uint64_t N = p.rows;
    // gridDim.y workgroups share the staging loops of a long page; the def levels and the chunk list are part 0's
    const uint64_t t0 = (uint64_t)blockIdx.y * WG + threadIdx.x, tstep = (uint64_t)gridDim.y * WG;
    if (c.nullable && blockIdx.y == 0) {
        uint8_t* bits = def_header(page_slot(a, c, p), N);
        def_bits_page(bits, ValidView{c.validity, c.validity_bit_offset}, p.row0, N, c.rows);
    }
    uint8_t* stage = a.scratch + p.aux_off;
    if (c.ptype == SB_TYPE_BOOLEAN) {
        if (b.stage_a) {   // bitmap re-packed from bit 0 (boolean/mod.rs:44-54)
            const uint64_t boff = c.values_bit_offset + p.row0, total_bits = c.values_bit_offset + c.rows;
            for (uint64_t i = t0; i < b.n_a; i += tstep) {
                uint32_t w = bits32(c.values, boff + i * 8, total_bits);
                const uint64_t nb = min((uint64_t)8, N - i * 8);
                if (nb < 8) w &= (1u << nb) - 1;
                stage[i] = (uint8_t)w;
            }
        }
    } else if (b.src_b) {  // offsets re-based to 0 (binary/mod.rs:45-55)
        const uint32_t ow = c.width;
        const uint8_t* offs = c.offsets + p.row0 * ow;
        for (uint64_t i = t0; i <= N; i += tstep) {
            if (ow == 4)
                stu32(stage + i * 4, (uint32_t)(ldu32(offs + i * 4) - b.first));
            else
                stu64(stage + i * 8, ldu64(offs + i * 8) - b.first);
        }
    }
    if (blockIdx.y) return;
    const uint32_t ch_a = lzc_chunk_of(a, b.n_a), ch_b = lzc_chunk_of(a, b.n_b);
    const uint32_t na = (b.n_a + ch_a - 1) / ch_a, nb = (b.n_b + ch_b - 1) / ch_b;
    if (threadIdx.x == 0) {
        uint32_t base = atomicAdd(a.lzc_count, na + nb);
        if (base + na + nb > a.lzc_cap) {   // (offsets that run past values_len)
            raise(a.status, SB_ERR_INVALID, page, 560);
            base = ~0u;
        } else {
            a.lzc_plan[page] = LzChunkPlan{base, na, nb, 0};
        }
        s_base = base;
    }
    __syncthreads();
    if (s_base == ~0u) return;
    for (uint32_t i = threadIdx.x; i < na + nb; i += WG) a.lzc_list[s_base + i] = LzChunkDesc{page, i < na ? i : (0x80000000u | (i - na))};
}

// chunks: one wave per chunk, sequences into the chunk's pool slot.  The kernel's throughput is the number of waves a CU
// holds (each is latency-bound), i.e. LDS per wave: measured on C3' with a 2^12 / 2^11 / 2^10-entry table 70 / 83 / 93
// GB/s for 499.8 / 500.9 / 502.5 MB of pages — the ring holds ~5 KiB of history, but 1024 entries cost 29 % on small-integer columns (1000 distinct 4-byte patterns), so 2048 it is.
#ifndef SB_LZC_HB
#define SB_LZC_HB 11
#endif
__global__ void __launch_bounds__(64) k_enc_lz4_chunks(EncodeArgs a) {
    __shared__ Lz4EncLds<SB_LZC_HB, 13> L;
    if (a.lzc_codec == SB_CODEC_ZSTD) return;   // (a call's chunks are all of one codec; Zstd blocks: k_enc_zstd_chunks)
    const uint32_t total = min(*a.lzc_count, a.lzc_cap);
    for (uint32_t i = blockIdx.x; i < total; i += gridDim.x) {
        const LzChunkDesc d = a.lzc_list[i];
        const EncPage p = get_page(a, d.page);
        const EncCol c = get_col(a, p.col);
        const LzBlocks b = lz4_page_blocks(a, c, p);
        const bool second = d.idx >> 31;
        const uint8_t* src = second ? b.src_b : b.src_a;
        const uint32_t n = second ? b.n_b : b.n_a;
        const uint32_t ch = lz4_chunk_bytes(n);
        const uint32_t c0 = (d.idx & 0x7FFFFFFFu) * ch, c1 = min(n, c0 + ch);
        uint8_t* slot = a.lzc_pool + (uint64_t)i * LZC_SLOT;
        uint32_t anchor = c0;
        wave_sync();
        const uint32_t len = a.lzc_codec == SB_CODEC_SNAPPY ? snappy_compress_range<SB_LZC_HB, 13, true>(src, n, c0, c1, slot + 16, L)
                                                            : lz4_compress_range<SB_LZC_HB, 13, true>(src, n, c0, c1, slot + 16, L, &anchor);
        if (threadIdx.x == 0) {
            stu32(slot, len);
            stu32(slot + 4, anchor);
        }
    }
}

// Zstd: the same plan with 16 KiB pieces, each written as a FRAME of its own — concatenated frames are one valid
// compressed buffer for ZSTD_decompress (what the reference calls), and they share nothing, so both directions run one
// wave per piece (the serial part of a piece, its FSE state chain, is what bounds a wave; 32 waves per 512 KiB page
// instead of one).  The price is the cold start of every piece: no history, a Huffman table per piece.
__global__ void __launch_bounds__(64) k_enc_zstd_chunks(EncodeArgs a) {
    __shared__ ZEncLds Z;
    if (a.lzc_codec != SB_CODEC_ZSTD) return;
    const uint32_t total = min(*a.lzc_count, a.lzc_cap);
    uint8_t* scratch = a.zpar_scratch + (uint64_t)blockIdx.x * zstd_scratch_bytes(ZPAR_CH);
    // pieces are handed out by a ticket: a piece with sequences takes a wave ~2 ms, a literals-only one a tenth of that, and a
    // fixed stride gave some waves several of the long ones
    for (;;) {
        uint32_t i = 0;
        if ((threadIdx.x & 63) == 0) i = atomicAdd(a.lzc_count + 1, 1u);
        i = (uint32_t)__builtin_amdgcn_readfirstlane((int)i);
        if (i >= total) break;
        const LzChunkDesc d = a.lzc_list[i];
        const EncPage p = get_page(a, d.page);
        const EncCol c = get_col(a, p.col);
        const LzBlocks b = lz4_page_blocks(a, c, p);
        const bool second = d.idx >> 31;
        const uint8_t* src = second ? b.src_b : b.src_a;
        const uint32_t n = second ? b.n_b : b.n_a;
        const uint32_t c0 = (d.idx & 0x7FFFFFFFu) * a.lzc_chunk, c1 = min(n, c0 + a.lzc_chunk);
        uint8_t* slot = a.lzc_pool + (uint64_t)i * LZC_SLOT;
        wave_sync();
        const uint32_t len = zstd_compress_block_alone(src, n, c0, c1, slot + 16, Z, scratch, a.lzc_chunk);
        if (threadIdx.x == 0) stu32(slot, len);
        wave_stores_visible();
    }
}

// the frames [chunk0, chunk0 + nch) of the pieces of src[0, n) back to back; returns their size.  sh: 2 * WG + 8 words.
// (SNAPPY: the stream's uvarint length followed by the chunks' elements — the same concatenation)
// part / parts: the workgroups (blockIdx.y) that share one block — every one of them computes all offsets, each copies the
// chunks k with k % parts == part; the fixed bytes are written by part 0
template <bool SNAPPY>
__device__ uint32_t zstd_stitch_frame(const EncodeArgs& a, uint32_t n, uint32_t chunk0, uint32_t nch, uint8_t* dst, uint32_t* sh,
                                      uint32_t part = 0, uint32_t parts = 1) {
    const uint32_t t = threadIdx.x, lane = t & 63, w = t >> 6;
    uint32_t *s_off = sh, *s_len = sh + WG, *s_w = sh + 2 * WG;
    uint32_t run = 0;
    if (SNAPPY) {
        if (t == 0 && part == 0) snappy_put_preamble(dst, n);
        run = snappy_preamble_bytes(n);
        if (nch == 0) return run;
    } else if (nch == 0) {   // an empty buffer: one frame with one empty raw block
        if (t == 0 && part == 0) {
            const uint32_t h = ze_frame_header(dst, 0);
            dst[h] = 1; dst[h + 1] = 0; dst[h + 2] = 0;
        }
        return ze_frame_header_bytes(0) + 3;
    }   // (Zstd: the chunks are whole frames, nothing in front of them)
    for (uint32_t b0 = 0; b0 < nch; b0 += WG) {
        const uint32_t k = b0 + t;
        const uint32_t len = k < nch ? ldu32(a.lzc_pool + (uint64_t)(chunk0 + k) * LZC_SLOT) : 0u;
        const uint32_t isum = wave_scan_dpp(len);
        if (lane == 63) s_w[w] = isum;
        __syncthreads();
        uint32_t at = run + isum - len;
        for (uint32_t q = 0; q < WG / 64; q++)
            if (q < w) at += s_w[q];
        s_off[t] = at;
        s_len[t] = len;
        uint32_t tot = 0;
        for (uint32_t q = 0; q < WG / 64; q++) tot += s_w[q];
        __syncthreads();
        run += tot;
        const uint32_t cnt = min((uint32_t)WG, nch - b0);
        for (uint32_t j = (part + parts - b0 % parts) % parts; j < cnt; j += parts)
            wg_copy(dst + s_off[j], a.lzc_pool + (uint64_t)(chunk0 + b0 + j) * LZC_SLOT + 16, s_len[j]);
        __syncthreads();
    }
    return run;
}

// lz4_put_head by a whole workgroup (every thread calls it): a literal run of 48 MB has a 188 KB length extension
// (part / parts: the workgroups that share it)
__device__ __forceinline__ void lz4_put_head_wg(uint8_t* o, uint32_t lit, uint32_t mcode, uint32_t part = 0, uint32_t parts = 1) {
    if (lit < 15 + 255 * 64) {   // short: one lane
        if (threadIdx.x == 0 && part == 0) lz4_put_head(o, lit, mcode);
        return;
    }
    const uint32_t r = lit - 15, n255 = r / 255;
    if (threadIdx.x == 0 && part == 0) {
        o[0] = (uint8_t)(0xF0u | min(mcode, 15u));
        o[1 + n255] = (uint8_t)(r - n255 * 255);
    }
    const uint32_t per = ((n255 + parts - 1) / parts + 15) & ~15u;
    const uint32_t b0 = min(n255, per * part), b1 = min(n255, b0 + per);
    const u32x4 ff = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    const uint32_t n16 = (b1 - b0) / 16;
    for (uint32_t i = threadIdx.x; i < n16; i += WG) stu128(o + 1 + b0 + 16 * i, ff);
    for (uint32_t i = b0 + 16 * n16 + threadIdx.x; i < b1; i += WG) o[1 + i] = 255;
}

// joins the nch chunks (slots of slot_stride bytes at pool: u32 size | u32 tail anchor | 8 pad | sequences) of the block
// src[0, n) at dst; returns the block size.  sh: 5 * WG + 8 words.
__device__ uint32_t lz4_stitch_block(const uint8_t* pool, uint32_t slot_stride, uint32_t chunk_bytes, const uint8_t* src, uint32_t n,
                                     uint32_t nch, uint8_t* dst, uint32_t* sh, uint32_t part, uint32_t parts) {
    const uint32_t t = threadIdx.x, lane = t & 63, w = t >> 6;
    uint32_t *s_off = sh, *s_cs = sh + WG, *s_ll = sh + 2 * WG, *s_e1 = sh + 3 * WG, *s_len = sh + 4 * WG, *s_w = sh + 5 * WG;
    uint32_t run_base = 0;    // bytes of the block written by the chunks before this batch
    uint32_t run_carry = 0;   // where the literals begin that no sequence holds yet
    for (uint32_t b0 = 0; b0 < nch; b0 += WG) {
        const uint32_t k = b0 + t;
        const uint8_t* slot = pool + (uint64_t)(k < nch ? k : 0) * slot_stride;
        uint32_t len = 0, anchor = 0, ll = 0, e1 = 0;
        if (k < nch) {
            len = ldu32(slot);
            anchor = ldu32(slot + 4);
        }
        if (len) {  // literal length of the chunk's first sequence
            ll = slot[16] >> 4;
            if (ll == 15)
                for (;;) {
                    const uint32_t x = slot[17 + e1];
                    e1++;
                    ll += x;
                    if (x != 255) break;
                }
        }
        // literals pending when chunk k begins start at the tail anchor of the last chunk before it that holds a sequence
        uint32_t inc = len ? anchor : 0;
        for (uint32_t d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(inc, d, 64);
            if (lane >= d) inc = max(inc, o);
        }
        uint32_t exc = __shfl_up(inc, 1, 64);
        if (lane == 0) exc = 0;
        if (lane == 63) s_w[4 + w] = inc;
        __syncthreads();
        uint32_t cs = max(run_carry, exc);
        for (uint32_t q = 0; q < WG / 64; q++)
            if (q < w) cs = max(cs, s_w[4 + q]);
        const uint32_t carry = len ? k * chunk_bytes - cs : 0;
        const uint32_t size = len ? (carry ? lz4_head_bytes(carry + ll) - 1 + carry + len - e1 : len) : 0;
        const uint32_t isum = wave_scan_dpp(size);
        if (lane == 63) s_w[w] = isum;
        __syncthreads();
        uint32_t at = run_base + isum - size;
        for (uint32_t q = 0; q < WG / 64; q++)
            if (q < w) at += s_w[q];
        s_off[t] = at;
        s_cs[t] = cs;
        s_ll[t] = ll;
        s_e1[t] = e1;
        s_len[t] = len;
        uint32_t tot = 0, bmax = 0;
        for (uint32_t q = 0; q < WG / 64; q++) {
            tot += s_w[q];
            bmax = max(bmax, s_w[4 + q]);
        }
        __syncthreads();
        run_base += tot;
        run_carry = max(run_carry, bmax);
        const uint32_t cnt = min((uint32_t)WG, nch - b0);
        for (uint32_t j = (part + parts - b0 % parts) % parts; j < cnt; j += parts) {
            const uint32_t len_j = s_len[j];
            if (!len_j) continue;
            const uint8_t* sl = pool + (uint64_t)(b0 + j) * slot_stride + 16;
            uint8_t* d = dst + s_off[j];
            const uint32_t carry_j = (b0 + j) * chunk_bytes - s_cs[j];
            if (!carry_j) {
                wg_copy(d, sl, len_j);
                continue;
            }
            const uint32_t ll_j = s_ll[j], e1_j = s_e1[j], nl = carry_j + ll_j, hb = lz4_head_bytes(nl);
            lz4_put_head_wg(d, nl, sl[0] & 15u);
            wg_copy(d + hb, src + s_cs[j], nl);
            wg_copy(d + hb + nl, sl + 1 + e1_j + ll_j, len_j - 1 - e1_j - ll_j);
        }
        __syncthreads();
    }
    const uint32_t lit = n - run_carry, hb = lz4_head_bytes(lit);   // the literals-only last sequence (the whole block when
    lz4_put_head_wg(dst + run_base, lit, 0, part, parts);            // nothing matched): shared like the chunks
    const uint32_t per = ((lit + parts - 1) / parts + 63) & ~63u;
    const uint32_t l0 = min(lit, per * part), l1 = min(lit, l0 + per);
    if (l1 > l0) wg_copy(dst + run_base + hb + l0, src + run_carry + l0, l1 - l0);
    return run_base + hb + lit;
}

__global__ void __launch_bounds__(WG) k_enc_lz4_stitch(EncodeArgs a) {
    __shared__ uint32_t sh[5 * WG + 8];
    if (a.use_counts && a.codec_counts[a.lzc_codec & 31] == 0) return;
    const uint32_t page = spread_block(blockIdx.x, gridDim.x) + a.page_base;
    const LzChunkPlan pl = a.lzc_plan[page];
    if (pl.n_a + pl.n_b == 0) return;
    const EncPage p = get_page(a, page);
    const EncCol c = get_col(a, p.col);
    const LzBlocks b = lz4_page_blocks(a, c, p);
    const uint32_t codec = (uint32_t)a.lzc_codec;
    const bool zstd = codec == SB_CODEC_ZSTD, snap = codec == SB_CODEC_SNAPPY;
    uint8_t* slot = page_slot(a, c, p);
    const uint64_t N = p.rows;
    const uint64_t pos = c.nullable ? def_section_bytes(N) : 0;
    uint8_t* blk = slot + pos;
    // gridDim.y workgroups share the page's chunks (few pages of many chunks: a one-page column)
    const uint32_t part = blockIdx.y, parts = gridDim.y;
    const uint32_t s1 = zstd ? zstd_stitch_frame<false>(a, b.n_a, pl.base, pl.n_a, blk + 9, sh, part, parts)
                        : snap ? zstd_stitch_frame<true>(a, b.n_a, pl.base, pl.n_a, blk + 9, sh, part, parts)
                             : lz4_stitch_block(a.lzc_pool + (uint64_t)pl.base * LZC_SLOT, LZC_SLOT, lz4_chunk_bytes(b.n_a), b.src_a, b.n_a, pl.n_a, blk + 9, sh, part, parts);
    if (threadIdx.x == 0 && part == 0) put_hdr9(blk, codec, s1, c.ptype == SB_TYPE_BOOLEAN ? (uint32_t)N : b.n_a);
    uint64_t length = pos + 9 + s1;
    if (b.src_b) {
        uint8_t* b2 = blk + 9 + s1;
        const uint32_t s2 = zstd ? zstd_stitch_frame<false>(a, b.n_b, pl.base + pl.n_a, pl.n_b, b2 + 9, sh, part, parts)
                            : snap ? zstd_stitch_frame<true>(a, b.n_b, pl.base + pl.n_a, pl.n_b, b2 + 9, sh, part, parts)
                                 : lz4_stitch_block(a.lzc_pool + (uint64_t)(pl.base + pl.n_a) * LZC_SLOT, LZC_SLOT, lz4_chunk_bytes(b.n_b), b.src_b, b.n_b, pl.n_b, b2 + 9, sh, part, parts);
        if (threadIdx.x == 0 && part == 0) put_hdr9(b2, codec, s2, b.n_b);
        length += 9 + s2;
    }
    if (threadIdx.x == 0 && part == 0) {
        EncOut o{length, 0, slot, codec, 0};
        a.outs[page] = o;
    }
}

// pages with codec None: (page, tile) parallel plain copies
__device__ void emit_tile(const EncodeArgs& a, uint32_t page, uint32_t tile) {
    const EncPage p = get_page(a, page);
    if (p.codec < CODEC_ON_DEVICE) return;  // table entry not in use
    const EncCol c = get_col(a, p.col);
    if (c.ptype == SB_TYPE_NULL) {
        if (tile == 0 && threadIdx.x == 0) {
            EncOut o{0, 0, c.out, 0, 0};
            a.outs[page] = o;
        }
        return;
    }
    if (codec_of(a, p, page) != SB_CODEC_NONE) return;
    const uint64_t N = p.rows;
    const uint64_t r0 = (uint64_t)tile * TILE_ROWS;
    if (r0 >= N && !(tile == 0)) return;
    const uint32_t rows = r0 < N ? (uint32_t)min((uint64_t)TILE_ROWS, N - r0) : 0;
    uint8_t* slot = page_slot(a, c, p);
    uint64_t pos = 0;
    if (c.nullable) {
        uint8_t* bits = tile == 0 ? def_header(slot, N) : slot + 4 + uleb_len((((N + 7) / 8) << 1) | 1);
        if (rows) def_bits_tile(bits, ValidView{c.validity, c.validity_bit_offset}, p.row0, N, c.rows, r0, rows);
        pos = def_section_bytes(N);
    }
    uint8_t* blk = slot + pos;
    const bool is_bin = c.ptype == SB_TYPE_BINARY || c.ptype == SB_TYPE_LARGE_BINARY;
    if (c.ptype == SB_TYPE_BOOLEAN) {
        // bitmap bytes: re-packed from bit 0 when the slice is not byte aligned, else the raw bytes
        // of the shared buffer, trailing bits of the last byte included (boolean/mod.rs:44-54)
        const uint64_t boff = c.values_bit_offset + p.row0;
        const uint64_t nbytes = (N + 7) / 8;
        if (tile == 0 && threadIdx.x == 0) put_hdr9(blk, SB_CODEC_NONE, (uint32_t)nbytes, (uint32_t)N);
        const uint32_t tb = (rows + 7) / 8;
        uint8_t* d = blk + 9 + (r0 >> 3);
        const uint64_t total_bits = c.values_bit_offset + c.rows;
        for (uint32_t b = threadIdx.x; b < tb; b += WG) {
            uint32_t w = bits32(c.values, boff + r0 + (uint64_t)b * 8, (boff & 7) ? total_bits : ((total_bits + 7) & ~7ull));
            if ((boff & 7) != 0) {
                const uint32_t nb = min(8u, rows - b * 8);
                if (nb < 8) w &= (1u << nb) - 1;
            }
            d[b] = (uint8_t)w;
        }
        if (tile == 0 && threadIdx.x == 0) {
            EncOut o{pos + 9 + nbytes, 0, slot, SB_CODEC_NONE, 0};
            a.outs[page] = o;
        }
    } else if (is_bin) {
        const uint32_t ow = c.width;
        const uint8_t* offs = c.offsets + p.row0 * ow;
        const uint64_t first = ow == 4 ? (uint64_t)ldu32(offs) : ldu64(offs);
        const uint64_t last = ow == 4 ? (uint64_t)ldu32(offs + N * 4) : ldu64(offs + N * 8);
        const uint64_t obytes = (N + 1) * ow, vbytes = last - first;
        uint8_t* ob = blk + 9;
        // offsets re-based to 0 (binary/mod.rs:45-55); entry N is written by the last tile
        const uint32_t cnt = rows + ((r0 + rows == N) ? 1 : 0);
        for (uint32_t i = threadIdx.x; i < cnt; i += WG) {
            if (ow == 4)
                stu32(ob + (r0 + i) * 4, (uint32_t)(ldu32(offs + (r0 + i) * 4) - first));
            else
                stu64(ob + (r0 + i) * 8, ldu64(offs + (r0 + i) * 8) - first);
        }
        uint8_t* vb = ob + obytes;
        const uint32_t ntiles = (uint32_t)max((uint64_t)1, (N + TILE_ROWS - 1) / TILE_ROWS);
        const uint64_t per = ((vbytes + ntiles - 1) / ntiles + 15) & ~(uint64_t)15;
        const uint64_t b0 = min(vbytes, per * tile), b1 = min(vbytes, per * (tile + 1));
        if (b1 > b0) wg_copy(vb + 9 + b0, c.values + first + b0, b1 - b0);
        if (tile == 0 && threadIdx.x == 0) {
            put_hdr9(blk, SB_CODEC_NONE, (uint32_t)obytes, (uint32_t)obytes);
            put_hdr9(vb, SB_CODEC_NONE, (uint32_t)vbytes, (uint32_t)vbytes);
            EncOut o{pos + 9 + obytes + 9 + vbytes, 0, slot, SB_CODEC_NONE, 0};
            a.outs[page] = o;
        }
    } else {
        const uint32_t w = c.width;
        if (tile == 0 && threadIdx.x == 0) put_hdr9(blk, SB_CODEC_NONE, (uint32_t)(N * w), (uint32_t)(N * w));
        if (rows) wg_copy(blk + 9 + r0 * w, c.values + (p.row0 + r0) * w, (uint64_t)rows * w);
        if (tile == 0 && threadIdx.x == 0) {
            EncOut o{pos + 9 + N * w, 0, slot, SB_CODEC_NONE, 0};
            a.outs[page] = o;
        }
    }
}

// grid = (pages, T): a workgroup takes the tiles blockIdx.y, blockIdx.y + T, ... of its page.  With the codec
// known on the host T = tiles per page (every tile its own workgroup); in adaptive mode, where few pages stay
// plain, T = 1: one workgroup per page finds out that there is nothing to do, not one per tile.
__global__ void __launch_bounds__(WG) k_enc_emit_tiles(EncodeArgs a, uint32_t max_tiles) {
    if (a.use_counts && a.codec_counts[SB_CODEC_NONE] == 0 && !a.null_cols) return;
    const uint32_t page = blockIdx.x + a.page_base;
    if (gridDim.y < max_tiles) {  // (adaptive) look at the page before walking its tiles
        const EncPage p = get_page(a, page);
        if (p.codec < CODEC_ON_DEVICE) return;
        if (get_col(a, p.col).ptype != SB_TYPE_NULL && codec_of(a, p, page) != SB_CODEC_NONE) return;
    }
    for (uint32_t tile = blockIdx.y; tile < max_tiles; tile += gridDim.y) {
        emit_tile(a, page, tile);
        __syncthreads();
    }
}

// page lengths -> offsets in the column's output, PageMeta for the host, capacity check.  One WAVE per column: a lane per
// page, 64 pages per step, the offsets an exclusive wave scan of head + length plus the carry of the steps before (one
// thread per column, eight pages fetched together, took 40 us for the 153 pages of a C4 column).
__global__ void __launch_bounds__(64) k_enc_layout(EncodeArgs a, const uint64_t* res_off) {
    const uint32_t ci = blockIdx.x, lane = threadIdx.x;
    if (ci >= a.n_cols) return;
    const EncCol c = a.cols[ci];
    uint64_t* res = a.results + res_off[ci];
    uint64_t off = 0;   // (wave-uniform)
    bool bad = false;
    for (uint32_t k0 = 0; k0 < c.n_pages; k0 += 64) {
        const uint32_t k = k0 + lane;
        const bool in = k < c.n_pages;
        const uint32_t pg = c.first_page + min(k, c.n_pages - 1);
        const uint64_t length = a.outs[pg].length;
        const uint64_t head = a.pages[pg].head_bytes, rows = a.pages[pg].rows, doff = a.pages[pg].direct_off;
        const uint32_t dir = a.pages[pg].direct;
        const uint64_t sz = in ? head + length : 0;
        uint64_t incl = sz;
        for (int o = 1; o < 64; o <<= 1) {
            const uint64_t u = __shfl_up(incl, o, 64);
            if ((int)lane >= o) incl += u;
        }
        const uint64_t my_off = off + incl - sz;
        if (in) {
            if ((length == 0 || a.outs[pg].pad == 3) && c.ptype != SB_TYPE_NULL) bad = true;   // (pad 3: a Dict page still waiting for its Freq-coded indices)
            a.outs[pg].out_off = my_off + head;  // the block goes behind the page's head
            res[k] = head + length;
            res[c.n_pages + k] = rows;
            if (dir && doff != my_off) bad = true;
        }
        off += __shfl(incl, 63, 64);
    }
    const bool any_bad = __ballot(bad) != 0;
    if (lane == 0 && blockIdx.x == 0) a.codec_counts[28] = *a.freq_count;   // (pages that went through the Freq kernels, for the next call's hint)
    if (lane == 0) {
        res[2 * c.n_pages] = off;
        if (any_bad && a.skips) {   // a page whose kernel was skipped on a hint: the interval is issued again in full
            atomicOr(&a.status->kinds, KIND_REPLAY);
        } else {
            if (off > c.out_cap) raise(a.status, SB_ERR_INVALID, c.first_page, 600);
            if (any_bad) raise(a.status, SB_ERR_EXTERNAL, c.first_page, 601);
        }
    }
}

__global__ void __launch_bounds__(WG) k_enc_compact(EncodeArgs a) {
    const uint32_t page = blockIdx.x;
    const EncPage p = get_page(a, page);
    const EncOut o = a.outs[page];
    const EncCol c = get_col(a, p.col);
    if (o.out_off + o.length > c.out_cap) return;
    if (blockIdx.y == 0 && p.head_bytes && c.heads)  // the page's head (nested level section)
        wg_copy(c.out + o.out_off - p.head_bytes, c.heads + p.head_off, p.head_bytes);
    if (p.direct) return;
    // gridDim.y workgroups share the page's 64 KiB chunks (the grid is sized for the worst-case slot; most pages
    // are far smaller, so the host keeps gridDim.y small when there are many pages)
    for (uint64_t b0 = (uint64_t)blockIdx.y * COMPACT_CHUNK; b0 < o.length; b0 += (uint64_t)gridDim.y * COMPACT_CHUNK) {
        const uint64_t n = min((uint64_t)COMPACT_CHUNK, o.length - b0);
        wg_copy(c.out + o.out_off + b0, o.slot + b0, n);
    }
}

}  // namespace sb

using namespace sb;

// ------------------------------------------------------------------------------ host side
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static uint32_t enc_type_width(int32_t t) {
    switch (t) {
        case SB_TYPE_INT8:
        case SB_TYPE_UINT8:
            return 1;
        case SB_TYPE_INT16:
        case SB_TYPE_UINT16:
            return 2;
        case SB_TYPE_INT32:
        case SB_TYPE_UINT32:
        case SB_TYPE_FLOAT32:
        case SB_TYPE_BINARY:
            return 4;
        case SB_TYPE_INT64:
        case SB_TYPE_UINT64:
        case SB_TYPE_FLOAT64:
        case SB_TYPE_LARGE_BINARY:
            return 8;
        case SB_TYPE_INT128:
            return 16;
        case SB_TYPE_INT256:
            return 32;
    }
    return 0;
}
static bool enc_is_binary(int32_t t) { return t == SB_TYPE_BINARY || t == SB_TYPE_LARGE_BINARY; }

// page arithmetic of encode_chunk (src/write/common.rs:54-58,79-86)
static uint64_t page_size_of(uint64_t rows, const sb_write_options* o) {
    uint64_t ps = o && o->max_page_size ? o->max_page_size : rows;
    return ps < rows ? ps : rows;
}

// worst-case bytes of one page's fixed part (everything but binary value bytes)
static uint64_t slot_fixed_bytes(int32_t ptype, int32_t nullable, uint64_t N) {
    const uint64_t w = enc_type_width(ptype);
    uint64_t b = 64 + (nullable ? def_section_bytes(N) : 0);
    if (ptype == SB_TYPE_BOOLEAN) return b + 9 + 5 * N + 16;  // RLE worst case: 5 bytes per row
    if (enc_is_binary(ptype)) return b + 9 + (N + 1) * w + 9 + 9 + N * 8 + 4 + 8 * N + 64;  // + value share
    // max(None, RLE = N*(4+w), Dict = 9 + 8N + 4 + N*w)
    return b + 9 + 9 + N * (w + 8) + 4 + 64;
}

extern "C" {

uint64_t sb_write_bound(int32_t physical_type, int32_t is_nullable, uint64_t rows, uint64_t values_len,
                        const sb_write_options* opts, uint64_t* n_pages) {
    if (rows == 0) {
        if (n_pages) *n_pages = 0;
        return 0;
    }
    const uint64_t ps = page_size_of(rows, opts);
    const uint64_t np = (rows + ps - 1) / ps;
    if (n_pages) *n_pages = np;
    uint64_t total = 0;
    for (uint64_t r = 0; r < rows; r += ps) total += slot_fixed_bytes(physical_type, is_nullable, r + ps > rows ? rows - r : ps);
    if (enc_is_binary(physical_type)) total += values_len + values_len / 64 + 64 * np;
    return total;
}

static int32_t write_columns_impl(sb_ctx* ctx, sb_column_write* cols, uint64_t n, const sb_write_options* opts, int32_t mem);
// SB_MEM_HOST calls of many columns in groups (see sb_read_columns): the pages of group g travel back while group g + 1's
// Arrow buffers travel in.  How many bytes a column's pages are is known when its results are back, so the host waits for
// group g's results (its inputs and kernels run behind group g + 1's copies, which are already queued) and sends exactly
// out_len bytes per column on the copy stream; the last group's pages go at the synchronize.
static void send_group_pages(sb_ctx* ctx, hipStream_t cs, sb::StageSlot* slot, size_t pd0, size_t pd1, size_t cb0, size_t cb1) {
    if (!slot || !slot->done || hipEventSynchronize(slot->done) != hipSuccess) return;
    for (size_t k = cb0; k < cb1; k++) {
        auto& cb = ctx->copybacks[k];
        if (cb.issued || !cb.used) continue;
        for (size_t q = pd0; q < pd1; q++) {
            const Pending& p = ctx->pending[q];
            if (p.kind != Pending::WRITE_COL || &((sb_column_write*)p.user)->out_len != cb.used) continue;
            const uint64_t* lens = (const uint64_t*)p.host;   // [n_pages lengths][n_pages num_values][total]
            const size_t nb = (size_t)std::min<uint64_t>(cb.n, lens[2 * p.n]);
            if (nb && hipMemcpyAsync(cb.host, cb.dev, nb, hipMemcpyDeviceToHost, cs) == hipSuccess) cb.issued = true;
            break;
        }
    }
}
int32_t sb_write_columns(sb_ctx* ctx, sb_column_write* cols, uint64_t n, const sb_write_options* opts, int32_t mem) {
    int32_t rc = SB_OK;
    uint64_t groups = 1;
    if (ctx && cols && opts && mem == SB_MEM_HOST && n >= 4) {
        uint64_t bytes = 0;
        for (uint64_t i = 0; i < n; i++) bytes += cols[i].rows * 8 + cols[i].values_len;   // (what travels, roughly)
        groups = bytes >= (32ull << 20) ? std::min<uint64_t>(ctx->host_groups_max, n / 2) : 1;
    }
    hipStream_t cs = groups > 1 ? ctx->copy_stream_get() : nullptr;
    if (!cs) {
        rc = write_columns_impl(ctx, cols, n, opts, mem);
    } else {
        const uint64_t per = (n + groups - 1) / groups;
        sb::StageSlot* prev_slot = nullptr;
        size_t ppd0 = 0, ppd1 = 0, pcb0 = 0, pcb1 = 0;
        for (uint64_t g0 = 0; g0 < n; g0 += per) {
            const size_t pd0 = ctx->pending.size(), cb0 = ctx->copybacks.size();
            rc = write_columns_impl(ctx, cols + g0, std::min<uint64_t>(per, n - g0), opts, mem);
            if (rc != SB_OK) break;
            if (prev_slot) send_group_pages(ctx, cs, prev_slot, ppd0, ppd1, pcb0, pcb1);   // (this group's copies in are queued behind it)
            prev_slot = ctx->last_slot;
            ppd0 = pd0; ppd1 = ctx->pending.size(); pcb0 = cb0; pcb1 = ctx->copybacks.size();
        }
    }
    if (rc == SB_OK && ctx && n && !ctx->in_replay) ctx->calls.push_back(sb_ctx::Call{1, cols, n, *opts, mem});   // (for a replay: sb_host.h)
    return rc;
}
static int32_t write_columns_impl(sb_ctx* ctx, sb_column_write* cols, uint64_t n, const sb_write_options* opts, int32_t mem) {
    if (!ctx || (!cols && n) || !opts) return SB_ERR_INVALID;
    if (n == 0) return SB_OK;
    (void)hipSetDevice(ctx->device);
    hipStream_t s = ctx->stream;
    // codec known on the host? (forced, or default_compress_ratio == None => Basic(default))
    int32_t host_codec = -1;
    if (opts->force_codec >= 0 && !((opts->forbidden_compressions >> opts->force_codec) & 1))
        host_codec = opts->force_codec;
    else if (!opts->has_default_compress_ratio)
        host_codec = opts->default_compression;
    const bool adaptive = host_codec < 0;  // codec chosen per page on the device (k_enc_select)
    const uint32_t forb = opts->forbidden_compressions;
    if (host_codec == SB_CODEC_FREQ)
        for (uint64_t i = 0; i < n; i++) {
            const int32_t t = cols[i].physical_type;
            if (t == SB_TYPE_BOOLEAN || t == SB_TYPE_NULL)
                return ctx->fail(SB_ERR_OUT_OF_SPEC, "Unknown compression codec Freq for boolean");
        }
    // Freq pages (chosen or forced) send their exceptions through a second wave of the same kernels
    bool freq_possible = false;
    const bool dict_freq = host_codec == SB_CODEC_DICT &&  // a forced Dict page whose u32 indices may become a Freq block
                           (opts->force_index_codec == SB_CODEC_FREQ ||
                            (opts->force_index_codec < 0 && opts->has_default_compress_ratio && !((forb >> SB_CODEC_FREQ) & 1)));
    if (host_codec == SB_CODEC_FREQ || dict_freq || (adaptive && !((forb >> SB_CODEC_FREQ) & 1)))
        for (uint64_t i = 0; i < n; i++) {
            const int32_t t = cols[i].physical_type;
            freq_possible |= t != SB_TYPE_BOOLEAN && t != SB_TYPE_NULL;
        }

    // ---- plan cache: everything below that depends only on the shape of the call (sb_ctx::EncPlan)
    uint64_t plan_key = 0xcbf29ce484222325ull;
    std::vector<uint64_t>& key_words = ctx->enc_plan_probe;   // every word that goes into the hash: a hit compares them all
    key_words.clear();
    {
        auto mixk = [&](uint64_t v) {
            plan_key = (plan_key ^ v) * 0x100000001b3ull;
            plan_key ^= plan_key >> 29;
            key_words.push_back(v);
        };
        mixk(n); mixk((uint64_t)mem);
        mixk((uint64_t)opts->default_compression); mixk((uint64_t)opts->has_default_compress_ratio); mixk(opts->max_page_size);
        mixk(opts->forbidden_compressions); mixk((uint64_t)(int64_t)opts->force_codec); mixk((uint64_t)(int64_t)opts->force_index_codec);
        mixk(opts->flags); mixk(opts->rng_seed);
        for (uint64_t i = 0; i < n; i++) {
            const sb_column_write& c = cols[i];
            mixk((uint64_t)(int64_t)c.physical_type); mixk((uint64_t)c.is_nullable); mixk(c.rows); mixk(c.values_len); mixk(c.out_capacity);
            mixk(c.first_page_index); mixk(c.n_pages_capacity); mixk(c.page_rows ? c.n_pages_in + 1 : 0);
            if (c.page_rows)
                for (uint64_t q = 0; q < c.n_pages_in; q++) {
                    mixk(c.page_rows[q]);
                    mixk(c.page_head_bytes ? c.page_head_bytes[q] : 0);
                }
        }
    }
    sb_ctx::EncPlan& plan = ctx->enc_plan;
    // (a 64-bit hash is not proof of an identical shape: a collision would reuse page offsets and row counts of another
    // call and write outside the scratch and output areas)
    bool hit = plan.valid && plan.key == plan_key && plan.n == n && plan.key_words == key_words;
    if (hit) {   // (a second, independent look at the shape: the page count per column)
        for (uint64_t i = 0; i < n && hit; i++) {
            const uint64_t ps = page_size_of(cols[i].rows, opts);
            const uint64_t np = cols[i].page_rows ? cols[i].n_pages_in : (cols[i].rows ? (cols[i].rows + ps - 1) / ps : 0);
            hit = np == plan.col_pages[i];
        }
    }
    uint64_t P = 0, max_tiles = 1, max_chunks = 1;
    bool big_possible = false;
    // LZ4 blocks of more than LZC_CH bytes are compressed chunk by chunk (flat pages, the matcher that is free to choose)
    const bool zs_possible = host_codec == SB_CODEC_ZSTD || (adaptive && opts->default_compression == SB_CODEC_ZSTD);
    const bool sn_possible = host_codec == SB_CODEC_SNAPPY || (adaptive && opts->default_compression == SB_CODEC_SNAPPY);
    const bool lz_possible = zs_possible || sn_possible || (!(opts->flags & SB_WRITE_LZ4_EXACT) &&
                             (host_codec == SB_CODEC_LZ4 || (adaptive && opts->default_compression == SB_CODEC_LZ4)));
    uint64_t lz_chunk = zs_possible ? ZPAR_CH : LZC_CH;
    uint64_t lz_cap = 0, lz_cap_small = 0;   // (_small: with Zstd pieces of ZPAR_CH_SMALL bytes, see below)
    bool lz_any = false, lz_any_small = false;
    for (uint64_t i = 0; i < n; i++) {
        sb_column_write& c = cols[i];
        if (c.physical_type < 0 || c.physical_type > SB_TYPE_NULL) return ctx->fail(SB_ERR_INVALID, "bad physical_type");
        // a flat column of zero rows: encode_chunk panics upstream (common.rs:54-58 divides by the page size).  A
        // nested leaf with explicit paging may hold zero leaf slots (every list empty or null): write_nested
        // compresses an empty leaf block per page.
        if (c.rows == 0 && !(c.page_rows && c.n_pages_in))
            return ctx->fail(SB_ERR_OUT_OF_SPEC, "encode_chunk on an empty chunk panics upstream");
        if (c.rows && c.physical_type != SB_TYPE_NULL && !c.values && !(enc_is_binary(c.physical_type) && c.values_len == 0))
            return ctx->fail(SB_ERR_INVALID, "values is null");  // (a binary column of empty strings has no value bytes)
        if (enc_is_binary(c.physical_type) && !c.offsets) return ctx->fail(SB_ERR_INVALID, "offsets is null");
        const uint64_t ps = page_size_of(c.rows, opts);
        const uint64_t np = c.page_rows ? c.n_pages_in : (c.rows + ps - 1) / ps;
        if (np > c.n_pages_capacity || !c.out_metas) return ctx->fail(SB_ERR_INVALID, "out_metas too small");
        // (long pages that may become Dict pages run their index arrays as virtual pages: sb_dict_big.h)
        big_possible |= adaptive && c.rows >= std::min<uint64_t>(SEL_BIG_ROWS, BIN_BIG_ROWS);
        if (!c.out_pages && c.physical_type != SB_TYPE_NULL) return ctx->fail(SB_ERR_INVALID, "out_pages is null");
        P += np;
        if (hit) continue;   // (the per-page arithmetic of this shape is in the plan)
        if (c.page_rows) {
            uint64_t sum = 0;
            for (uint64_t q = 0; q < np; q++) sum += c.page_rows[q];
            if (sum != c.rows) return ctx->fail(SB_ERR_INVALID, "page_rows do not add up to rows");
        }
        {
            uint64_t mx = ps;
            if (c.page_rows)
                for (uint64_t q = 0; q < np; q++) mx = std::max<uint64_t>(mx, c.page_rows[q]);
            max_tiles = std::max<uint64_t>(max_tiles, (mx + TILE_ROWS - 1) / TILE_ROWS);
        }
        if (lz_possible && c.physical_type != SB_TYPE_NULL) {   // upper bound of the LZ4 chunks this column can ask for
            const bool bin = enc_is_binary(c.physical_type);
            const uint64_t w = enc_type_width(c.physical_type);
            auto first_block = [&](uint64_t N) {
                return c.physical_type == SB_TYPE_BOOLEAN ? (N + 7) / 8 : bin ? (N + 1) * w : N * w;
            };
            for (uint64_t q = 0, r = 0; q < np; q++) {
                const uint64_t N = c.page_rows ? c.page_rows[q] : std::min<uint64_t>(ps, c.rows - r);
                r += N;
                const uint64_t fb = first_block(N);
                const uint64_t ch = zs_possible ? lz_chunk : lz4_chunk_bytes(fb);
                lz_cap += (fb + ch - 1) / ch;
                lz_any |= fb > lz_chunk;
                lz_cap_small += (fb + ZPAR_CH_SMALL - 1) / ZPAR_CH_SMALL;
                lz_any_small |= fb > ZPAR_CH_SMALL;
            }
            if (bin) {   // (a page's value bytes are not known here: LZC_LONG of them in one page need that many in the column)
                lz_cap += c.values_len / (zs_possible ? lz_chunk : lz4_chunk_bytes(c.values_len)) + np + 1;
                lz_any |= c.values_len > lz_chunk;
                lz_cap_small += c.values_len / ZPAR_CH_SMALL + np + 1;
                lz_any_small |= c.values_len > ZPAR_CH_SMALL;
            }
        }
    }
    // a call with fewer Zstd pieces than chunk waves (one nested array, a short column): smaller pieces, more waves — the
    // call's time is the latency of one piece there, not the chip's throughput
    if (zs_possible && !hit && lz_cap < ZPAR_WAVES) {
        lz_chunk = ZPAR_CH_SMALL;
        lz_cap = lz_cap_small;
        lz_any = lz_any_small;
    }
    if (!lz_any) lz_cap = 0;
    if (hit) {
        max_tiles = plan.max_tiles;
        lz_cap = plan.lz_cap;
        lz_chunk = plan.lz_chunk;
    }
    if (lz_cap >= 0x7FFFFFFFull) return ctx->fail(SB_ERR_INVALID, "too many LZ4 chunks in one call");
    if (P >= 0x7FFFFFFFull) return ctx->fail(SB_ERR_INVALID, "too many pages in one call");

    // SB_MEM_HOST: the caller holds host Arrow buffers (the reference's shape); stage them over PCIe
    std::vector<const uint8_t*> dv(n, nullptr), dval(n, nullptr), doff(n, nullptr), dheads(n, nullptr);
    std::vector<uint8_t*> dout(n, nullptr);
    if (mem == SB_MEM_HOST) {
        auto stage_in = [&](const void* host, size_t bytes, const uint8_t** out) -> bool {
            *out = nullptr;
            if (!host || !bytes) return true;
            uint8_t* d = ctx->stage_alloc(bytes);
            if (!d) return false;
            *out = d;
            return hipMemcpyAsync(d, host, bytes, hipMemcpyHostToDevice, s) == hipSuccess;
        };
        for (uint64_t i = 0; i < n; i++) {
            const sb_column_write& c = cols[i];
            const uint32_t w = enc_type_width(c.physical_type);
            size_t vbytes = c.physical_type == SB_TYPE_BOOLEAN ? (size_t)((c.values_bit_offset + c.rows + 7) / 8)
                            : enc_is_binary(c.physical_type) ? (size_t)c.values_len : (size_t)(c.rows * w);
            if (!stage_in(c.values, vbytes, &dv[i]) ||
                !stage_in(c.validity, (size_t)((c.validity_bit_offset + c.rows + 7) / 8), &dval[i]) ||
                !stage_in(enc_is_binary(c.physical_type) ? c.offsets : nullptr, (size_t)((c.rows + 1) * w), &doff[i]))
                return ctx->fail(SB_ERR_EXTERNAL, "staging of host buffers failed");
            if (c.page_heads && c.page_head_bytes) {  // nested level sections travel like every other DEVICE buffer
                size_t hb = 0;
                for (uint64_t q = 0; q < c.n_pages_in; q++) hb += (size_t)c.page_head_bytes[q];
                if (!stage_in(c.page_heads, hb, &dheads[i])) return ctx->fail(SB_ERR_EXTERNAL, "staging of host buffers failed");
            }
            if (c.out_capacity) {
                if (!(dout[i] = ctx->stage_alloc(c.out_capacity)))
                    return ctx->fail(SB_ERR_EXTERNAL, "hipMalloc(out_pages) failed");
                ctx->copybacks.push_back({c.out_pages, dout[i], (size_t)c.out_capacity, &cols[i].out_len});
            }
        }
    }


    size_t off = 0;
    const size_t o_cols = off;
    off = align_up(off + n * sizeof(EncCol), 64);
    const size_t o_resoff = off;
    off = align_up(off + n * sizeof(uint64_t), 64);
    const size_t upload_bytes = off;
    const size_t o_outs = off;
    off = align_up(off + 2 * P * sizeof(EncOut), 64);
    const size_t o_results = off;
    const size_t results_words = 2 * P + n;
    off = align_up(off + results_words * sizeof(uint64_t), 64);
    const size_t o_codecs = off;
    off = align_up(off + 2 * P * sizeof(int32_t), 64);
    const size_t o_freqcnt = off;
    off = align_up(off + 64 + 32 * sizeof(uint32_t), 64);  // freq_count | codec_counts[32]
    const size_t o_lzplan = off;                             // (inside the region zeroed per call: count | per-page plans)
    off = align_up(off + (lz_cap ? 64 + P * sizeof(LzChunkPlan) : 0), 64);
    const size_t o_vcols = off;
    off = align_up(off + (freq_possible || big_possible ? P : 0) * sizeof(EncCol), 64);
    const size_t o_vpages = off;
    off = align_up(off + (freq_possible || big_possible ? P : 0) * sizeof(EncPage), 64);
    const size_t o_lzlist = off;
    off = align_up(off + lz_cap * sizeof(LzChunkDesc), 64);
    if (!ensure(ctx, ctx->tables, off)) return ctx->fail(SB_ERR_EXTERNAL, "hipMalloc(tables) failed");

    // staging: [cols | result offsets | readback of the results | (plan miss) the page table]
    const size_t o_hres = upload_bytes;
    const size_t o_hcounts = align_up(o_hres + results_words * sizeof(uint64_t), 64);   // codec counts of the call (ENC_HINT)
    const size_t o_hpages = o_hcounts + 128;
    StageSlot* slot = acquire_slot(ctx, o_hpages + (hit ? 0 : P * sizeof(EncPage)) + 64);
    if (!slot) return ctx->fail(SB_ERR_EXTERNAL, "hipHostMalloc(staging) failed");
    EncCol* hc = (EncCol*)(slot->host + o_cols);
    EncPage* hp = hit ? nullptr : (EncPage*)(slot->host + o_hpages);
    uint64_t* hro = (uint64_t*)(slot->host + o_resoff);
    if (!hit) {
        plan.valid = false;
        plan.counts_valid = false;
        memset(plan.prev_counts, 0, sizeof plan.prev_counts);
        plan.bin_pages = plan.bin_unfused = false;
        if (!ensure(ctx, plan.pages, P * sizeof(EncPage) + 64)) return ctx->fail(SB_ERR_EXTERNAL, "hipMalloc(page table) failed");
        plan.col_first.assign(n, 0);
        plan.col_pages.assign(n, 0);
        plan.hro.assign(n, 0);
        for (int k = 0; k < 5; k++) {
            plan.bigw[k].clear();
            plan.big_secs[k] = 0;
        }
    }

    size_t scratch_off = 0;
    uint64_t pi = 0, res_off = 0;
    bool any_tiles = false, any_pages = false, any_compact = false, any_lz4 = false;
    for (uint64_t i = 0; i < n; i++) {
        const sb_column_write& c = cols[i];
        EncCol& d = hc[i];
        memset(&d, 0, sizeof d);
        d.values = mem == SB_MEM_HOST ? dv[i] : (const uint8_t*)c.values;
        d.validity = mem == SB_MEM_HOST ? dval[i] : c.validity;
        d.offsets = mem == SB_MEM_HOST ? doff[i] : (const uint8_t*)c.offsets;
        d.out = mem == SB_MEM_HOST ? dout[i] : c.out_pages;
        d.heads = mem == SB_MEM_HOST ? dheads[i] : c.page_heads;
        d.values_bit_offset = c.values_bit_offset;
        d.values_len = c.values_len;
        d.values_len_total = c.column_values_len ? c.column_values_len : c.values_len;
        d.validity_bit_offset = c.validity_bit_offset;
        d.out_cap = c.out_capacity;
        d.rows = c.rows;
        d.ptype = c.physical_type;
        d.nullable = c.is_nullable;
        d.width = enc_type_width(c.physical_type);
        d.first_page = (uint32_t)pi;
        d.fkind = c.physical_type == SB_TYPE_FLOAT32 ? 1 : c.physical_type == SB_TYPE_FLOAT64 ? 2 : 0;
        d.nk = c.physical_type == SB_TYPE_FLOAT32 ? NK_F32 : c.physical_type == SB_TYPE_FLOAT64 ? NK_F64
               : (c.physical_type >= SB_TYPE_UINT8 && c.physical_type <= SB_TYPE_UINT64) ? NK_UNSIGNED : NK_SIGNED;
        const uint64_t ps = page_size_of(c.rows, opts);
        const bool bin = enc_is_binary(c.physical_type);
        int32_t codec = host_codec;
        // sizes known up front => write straight to the final position
        const bool direct = !adaptive && !bin && (codec == SB_CODEC_NONE || codec == SB_CODEC_ONEVALUE) &&
                            c.physical_type != SB_TYPE_NULL;
        uint64_t direct_off = 0, k = 0;
        if (hit) {
            d.first_page = plan.col_first[i];
            d.n_pages = plan.col_pages[i];
            hro[i] = plan.hro[i];
            continue;
        }
        if (bin) scratch_off = align_up(scratch_off, 16);
        const size_t col_slot_base = scratch_off;
        uint64_t head_off = 0;
        for (uint64_t r = 0; r < c.rows || (c.page_rows && k < c.n_pages_in); k++, pi++) {
            EncPage& p = hp[pi];
            memset(&p, 0, sizeof p);
            const uint64_t N = c.page_rows ? c.page_rows[k] : (r + ps > c.rows ? c.rows - r : ps);
            p.head_bytes = c.page_head_bytes ? c.page_head_bytes[k] : 0;
            p.head_off = head_off;
            head_off += p.head_bytes;
            p.row0 = r;
            p.rows = N;
            p.col = (uint32_t)i;
            p.codec = codec;
            p.icodec = opts->force_index_codec;
            p.seed = page_seed_of(opts->rng_seed, c.first_page_index + k);
            p.direct = direct ? 1 : 0;
            if (adaptive && !bin && N >= SEL_BIG_ROWS && d.width <= 8 && c.physical_type != SB_TYPE_BOOLEAN &&
                c.physical_type != SB_TYPE_NULL) {
                const uint32_t secs = (uint32_t)((N + big_sec_rows(N) - 1) / big_sec_rows(N));
                const int k = d.width == 1 ? 0 : d.width == 2 ? 1 : d.width == 4 ? 2 : 3;
                plan.bigw[k].push_back((uint32_t)pi);
                plan.big_secs[k] = std::max(plan.big_secs[k], secs);
                if (!((forb >> SB_CODEC_DICT) & 1) || freq_possible) p.bigx_off = 1;   // (placed with the aux areas below)
            } else if (adaptive && bin && N >= BIN_BIG_ROWS && !((forb >> SB_CODEC_DICT) & 1)) {   // (row hashes exist: Dict is a candidate)
                const uint32_t secs = (uint32_t)((N + big_sec_rows(N) - 1) / big_sec_rows(N));
                plan.bigw[4].push_back((uint32_t)pi);
                plan.big_secs[4] = std::max(plan.big_secs[4], secs);
                p.bigx_off = 1;
            }
            if (direct) {
                p.direct_off = direct_off;
                const uint64_t body = c.physical_type == SB_TYPE_BOOLEAN
                                          ? (codec == SB_CODEC_NONE ? (N + 7) / 8 : 1)
                                          : (codec == SB_CODEC_NONE ? N * d.width : d.width);
                direct_off += p.head_bytes + (c.is_nullable ? def_section_bytes(N) : 0) + 9 + body;
            } else if (c.physical_type != SB_TYPE_NULL) {
                p.slot_off = scratch_off;
                scratch_off += align_up(slot_fixed_bytes(c.physical_type, c.is_nullable, N), 16);
                any_compact = true;
                if (freq_possible && !bin && c.physical_type != SB_TYPE_BOOLEAN) {
                    // a Freq page also holds the Roaring bitmap (<= 8 KiB + 8 B per 64 Ki rows, + header)
                    scratch_off += align_up(N / 8 + 16 * (N / 65536 + 1) + 8192 + 64, 16);
                    p.slot_cap = scratch_off - p.slot_off;
                    p.ex_off = scratch_off;  // exception values
                    scratch_off += align_up(N * d.width + 64, 16);
                    p.vslot_off = scratch_off;  // slot of the exceptions block (a non-nullable page of <= N rows)
                    scratch_off += align_up(slot_fixed_bytes(c.physical_type, 0, N), 16);
                } else if (freq_possible && bin && !((forb >> SB_CODEC_DICT) & 1) && (adaptive || codec == SB_CODEC_DICT)) {
                    // a binary Dict page with Freq-coded indices: < N/2 + 1 exception indices and their block
                    // (placed after all slots, below: a binary column's slots share one region with its value bytes)
                    p.ex_off = ~0ull;
                    p.vaux_bytes = 64;  // the idx / firsts / D record of the Dict page
                }
            }
            if (codec == SB_CODEC_DICT || codec == SB_CODEC_FREQ ||  // (forced Freq: exact counts when no value has a majority)
                (adaptive && !((forb >> SB_CODEC_DICT) & 1) && c.physical_type != SB_TYPE_BOOLEAN &&
                                           c.physical_type != SB_TYPE_NULL)) {
                uint64_t M = 64;
                while (M < 2 * N) M <<= 1;
                p.aux_bytes = (M + 3 * N) * 4;
            }
            // (a long page counts its distinct keys exactly in a table of 8-byte keys at the start of the aux area: sb_select_big.h)
            if (p.aux_bytes && p.bigx_off && N >= std::min<uint64_t>(SEL_BIG_ROWS, BIN_BIG_ROWS))
                p.aux_bytes = std::max<uint64_t>(p.aux_bytes, big_tab_slots(N) * 8);
            p.h64_off = ~0ull;
            if (bin && p.aux_bytes && N) p.h64_off = 0;   // (placed with the aux areas below)
            if (bin && adaptive) {
                plan.bin_pages = true;
                if (!(p.h64_off == 0 && opts->has_default_compress_ratio && bp_fits(N, p.aux_bytes) && !(N >= BP_BIG_ROWS && p.bigx_off) &&
                      !((forb >> SB_CODEC_DICT) & 1)))
                    plan.bin_unfused = true;
            }
            p.zst_off = ~0ull;
            if (c.physical_type != SB_TYPE_NULL &&
                (codec == SB_CODEC_ZSTD || (adaptive && opts->default_compression == SB_CODEC_ZSTD)))
                p.zst_off = 0;
            if (p.vslot_off && !((forb >> SB_CODEC_DICT) & 1)) {  // the exceptions block may be a Dict block
                uint64_t M = 64;
                while (M < 2 * N) M <<= 1;
                p.vaux_bytes = (M + 3 * N) * 4;
            }
            if (codec == SB_CODEC_LZ4 || codec == SB_CODEC_ZSTD || codec == SB_CODEC_SNAPPY ||
                (adaptive && (opts->default_compression == SB_CODEC_LZ4 || opts->default_compression == SB_CODEC_ZSTD ||
                              opts->default_compression == SB_CODEC_SNAPPY))) {
                any_lz4 = true;  // staging for re-based offsets / re-packed bitmaps
                const uint64_t st = bin ? (N + 1) * d.width + 16 : (c.physical_type == SB_TYPE_BOOLEAN ? (N + 7) / 8 + 16 : 0);
                if (st > p.aux_bytes) p.aux_bytes = st;
            }
            if (codec == SB_CODEC_NONE || (adaptive && opts->default_compression == SB_CODEC_NONE))
                any_tiles = true;
            if (codec != SB_CODEC_NONE && codec != SB_CODEC_LZ4 && codec != SB_CODEC_ZSTD && codec != SB_CODEC_SNAPPY) any_pages = true;
            if (p.head_bytes) any_compact = true;  // k_enc_compact also places the heads
            r += N;
            if (N == 0 && !c.page_rows) break;
        }
        if (bin) scratch_off += align_up(c.values_len + c.values_len / 64 + 64 * k + 64, 16);
        (void)col_slot_base;
        d.n_pages = (uint32_t)k;
        hro[i] = res_off;
        plan.col_first[i] = d.first_page;
        plan.col_pages[i] = d.n_pages;
        plan.hro[i] = res_off;
        res_off += 2 * k + 1;
        if (direct && direct_off > c.out_capacity) return ctx->fail(SB_ERR_INVALID, "out_capacity too small");
    }
    // Dict aux areas after the slots
    for (uint64_t q = 0; q < P && !hit; q++) {
        if (hp[q].aux_bytes) {
            scratch_off = align_up(scratch_off, 16);
            hp[q].aux_off = scratch_off;
            scratch_off += hp[q].aux_bytes;
        }
        if (hp[q].h64_off == 0) {
            scratch_off = align_up(scratch_off, 16);
            hp[q].h64_off = scratch_off;
            scratch_off += hp[q].rows * 8;
        }
        if (hp[q].bigx_off == 1) {
            scratch_off = align_up(scratch_off, 64);
            hp[q].bigx_off = scratch_off;
            scratch_off += ((forb >> SB_CODEC_DICT) & 1) ? BIGX_HEAD : dbig_layout(hp[q].rows, enc_is_binary(hc[hp[q].col].ptype)).total;
        }
        if (hp[q].zst_off == 0) {   // (one block is at most 128 KiB whatever the page holds)
            scratch_off = align_up(scratch_off, 16);
            hp[q].zst_off = scratch_off;
            scratch_off += zstd_scratch_bytes(ZE_BLOCK);
        }
        if (hp[q].vaux_bytes) {
            scratch_off = align_up(scratch_off, 16);
            hp[q].vaux_off = scratch_off;
            scratch_off += hp[q].vaux_bytes;
        }
        if (hp[q].ex_off == ~0ull) {  // binary Dict page with Freq-coded indices: exception indices and their block
            const uint64_t nh = hp[q].rows / 2 + 1;
            scratch_off = align_up(scratch_off, 16);
            hp[q].ex_off = scratch_off;
            scratch_off += align_up(nh * 4 + 64, 16);
            hp[q].vslot_off = scratch_off;
            scratch_off += align_up(slot_fixed_bytes(SB_TYPE_UINT32, 0, nh), 16);
        }
        if (!hp[q].direct) {
            const EncCol& d = hc[hp[q].col];
            uint64_t cap = std::max<uint64_t>(slot_fixed_bytes(d.ptype, d.nullable, hp[q].rows), hp[q].slot_cap);
            if (enc_is_binary(d.ptype)) cap += d.values_len;
            max_chunks = std::max<uint64_t>(max_chunks, (cap + COMPACT_CHUNK - 1) / COMPACT_CHUNK);
        }
    }
    scratch_off = align_up(scratch_off, 16);
    size_t lz_pool_off = scratch_off;
    scratch_off += (size_t)lz_cap * LZC_SLOT;
    size_t zpar_off = scratch_off;
    const uint32_t zpar_waves = (uint32_t)std::min<uint64_t>(lz_cap, ZPAR_WAVES);
    if (lz_cap && zs_possible) scratch_off += (size_t)zpar_waves * zstd_scratch_bytes(ZPAR_CH);
    if (hit) {
        scratch_off = plan.scratch_total;
        lz_pool_off = plan.lz_pool_off;
        zpar_off = plan.zpar_off;
        max_chunks = plan.max_chunks;
        any_tiles = plan.any_tiles;
        any_pages = plan.any_pages;
        any_compact = plan.any_compact;
        any_lz4 = plan.any_lz4;
    }
    if (!ensure(ctx, ctx->scratch, scratch_off + 64)) return ctx->fail(SB_ERR_EXTERNAL, "hipMalloc(scratch) failed");

    uint8_t* tb = ctx->tables.p;
    hipError_t e = hipMemcpyAsync(tb, slot->host, upload_bytes, hipMemcpyHostToDevice, s);
    if (e != hipSuccess) return check_hip(ctx, e, "table upload");
    if (!hit) {   // the page table goes to the plan's own device buffer and stays there
        e = hipMemcpyAsync(plan.pages.p, hp, P * sizeof(EncPage), hipMemcpyHostToDevice, s);
        if (e != hipSuccess) return check_hip(ctx, e, "page table upload");
        if (const size_t nb = plan.bigw[0].size() + plan.bigw[1].size() + plan.bigw[2].size() + plan.bigw[3].size() + plan.bigw[4].size()) {   // (pageable source: the copy is staged before the call returns)
            if (!ensure(ctx, plan.big, nb * sizeof(uint32_t) + 64)) return ctx->fail(SB_ERR_EXTERNAL, "hipMalloc(long-page list) failed");
            std::vector<uint32_t> both;
            for (int k = 0; k < 5; k++) both.insert(both.end(), plan.bigw[k].begin(), plan.bigw[k].end());
            e = hipMemcpyAsync(plan.big.p, both.data(), nb * sizeof(uint32_t), hipMemcpyHostToDevice, s);
            if (e == hipSuccess) e = hipStreamSynchronize(s);   // (`both` goes out of scope; plan misses are rare)
            if (e != hipSuccess) return check_hip(ctx, e, "long-page list upload");
        }
        plan.key = plan_key;
        plan.key_words = key_words;
        plan.n = n;
        plan.P = P;
        plan.max_tiles = max_tiles;
        plan.max_chunks = max_chunks;
        plan.lz_cap = lz_cap;
        plan.lz_chunk = lz_chunk;
        plan.any_tiles = any_tiles;
        plan.any_pages = any_pages;
        plan.any_compact = any_compact;
        plan.any_lz4 = any_lz4;
        plan.scratch_total = scratch_off;
        plan.lz_pool_off = lz_pool_off;
        plan.zpar_off = zpar_off;
        plan.valid = true;
    }

    EncodeArgs a;
    a.cols = (const EncCol*)(tb + o_cols);
    a.pages = (const EncPage*)plan.pages.p;
    a.outs = (EncOut*)(tb + o_outs);
    a.scratch = ctx->scratch.p;
    a.status = ctx->d_status;
    a.results = (uint64_t*)(tb + o_results);
    a.codecs = (int32_t*)(tb + o_codecs);
    a.ratio = opts->default_compress_ratio;
    a.has_ratio = opts->has_default_compress_ratio ? 1u : 0u;
    a.forbidden = forb;
    a.n_pages = (uint32_t)P;
    a.n_cols = (uint32_t)n;
    a.default_compression = (uint32_t)opts->default_compression;
    a.vcols = (const EncCol*)(tb + o_vcols);
    a.vpages = (const EncPage*)(tb + o_vpages);
    a.page_base = 0;
    a.nested_force = opts->force_index_codec;
    a.flags = opts->flags;
    a.lzc_plan = lz_cap ? (LzChunkPlan*)(tb + o_lzplan + 64) : nullptr;
    a.lzc_count = (uint32_t*)(tb + o_lzplan);
    a.lzc_list = (LzChunkDesc*)(tb + o_lzlist);
    a.lzc_pool = ctx->scratch.p + lz_pool_off;
    a.lzc_cap = (uint32_t)lz_cap;
    a.lzc_chunk = (uint32_t)lz_chunk;
    a.lzc_codec = zs_possible ? SB_CODEC_ZSTD : sn_possible ? SB_CODEC_SNAPPY : SB_CODEC_LZ4;
    a.zpar_scratch = lz_cap && zs_possible ? ctx->scratch.p + zpar_off : nullptr;
    a.freq_count = (uint32_t*)(tb + o_freqcnt);
    a.codec_counts = (uint32_t*)(tb + o_freqcnt + 64);
    a.use_counts = 0;
    a.pre_hashed = 0;
    a.redo = 0;
    a.null_cols = 0;
    for (uint64_t i = 0; i < n; i++) a.null_cols |= cols[i].physical_type == SB_TYPE_NULL ? 1u : 0u;
    // one memset: page outputs (length 0 = not emitted), results, device-chosen codecs, the Freq page counter
    (void)hipMemsetAsync(tb + o_outs, 0, o_vcols - o_outs, s);
    if (host_codec == SB_CODEC_FREQ) (void)hipMemsetAsync(a.freq_count, 1, 4, s);  // forced: every page is a Freq page (non-zero)
    auto kind_of = [](const EncCol& d) {
        return d.ptype == SB_TYPE_BOOLEAN ? 0 : d.ptype == SB_TYPE_BINARY ? -4 : d.ptype == SB_TYPE_LARGE_BINARY ? -8 : (int)d.width;
    };
    std::vector<int> kinds;
    for (uint64_t i = 0; i < n; i++) {
        if (hc[i].ptype == SB_TYPE_NULL) continue;
        const int kd = kind_of(hc[i]);
        bool seen = false;
        for (int q : kinds) seen |= q == kd;
        if (!seen) kinds.push_back(kd);
    }
    // One wave of select + emit kernels over the table entries [aa.page_base, aa.page_base + P).
    // wave_adaptive: codecs are chosen on the device; wave_codec: the one codec otherwise (-1: several possible).
    // (what the last call with this plan chose: see EncPlan.last_counts)
    const bool big_hint = hit && plan.counts_valid && adaptive && !ctx->no_hints;
    {
        size_t nbig_prim = plan.bigw[0].size() + plan.bigw[1].size() + plan.bigw[2].size() + plan.bigw[3].size(), nbig = nbig_prim + plan.bigw[4].size();
        a.skips = 0;
        if (big_hint && nbig && !((forb >> SB_CODEC_DICT) & 1) && !plan.last_counts[SB_CODEC_DICT]) a.skips |= SKIP_DICT_BIG;
        if (big_hint && nbig_prim && freq_possible && !plan.last_counts[SB_CODEC_FREQ]) a.skips |= SKIP_FREQ_BIG;
    }
    // launches that the last call with this plan would not have needed (its pages per codec, read back with its results):
    // skipped; a page that needs one after all stays unwritten and the interval is replayed with everything (k_enc_layout)
    uint32_t emit_skips = 0;
    auto unused_peek = [&](int slot) { return big_hint && !plan.last_counts[slot]; };
    auto unused = [&](int slot) {
        if (!big_hint || plan.last_counts[slot]) return false;
        emit_skips |= SKIP_EMIT;
        return true;
    };
    auto run_wave = [&](const EncodeArgs& aa_in, bool wave_adaptive, int32_t wave_codec, bool nested) -> int32_t {
        EncodeArgs aa = aa_in;
        aa.use_counts = wave_adaptive && !nested ? 1u : 0u;
        aa.pre_hashed = 0;
        aa.redo = 0;
        // binary pages of BP_MIN_ROWS .. 65 536 rows: codec and dictionary in one pass (sb_bin_page.h); the chain below only for the rest
        // (the flag also switches on k_enc_prim_dict for integer Dict pages: binary pages or not)
        aa.bin_fused = wave_adaptive && !nested && ctx->bin_fused && !(opts->flags & SB_WRITE_DEBUG_VERIFY_FAIL_BIT) ? 1u : 0u;
        const bool old_chain = !aa.bin_fused || plan.bin_unfused;
        int n_bin_kinds = 0;
        for (int kd : kinds) n_bin_kinds += kd < 0 ? 1 : 0;
        const bool any_bin = n_bin_kinds > 0;
        const uint64_t max_rows = max_tiles * TILE_ROWS;
        const dim3 tile_grid((uint32_t)P, (uint32_t)((max_rows + BH_ROWS - 1) / BH_ROWS));
        // Column kinds work on disjoint pages: in an adaptive wave over several kinds (a mixed schema: C4) every kind's chain
        // selector -> [string check, re-selection] -> page kernels runs on a stream of its own between a fork and a join, so
        // that a few hundred pages per kind share the chip instead of taking turns.  Not while profiling.
        const bool multi = wave_adaptive && !nested && kinds.size() >= 2 && n_bin_kinds < (int)kinds.size() && !ctx->profile && side_streams(ctx);
        // binary kinds (the longest chain: hash, selector, string check, dictionary pages) on the high-priority side stream,
        // the other kinds one after the other on the call's stream; without binary columns the kinds alternate between the
        // call's stream and a side stream.  (More streams than that only made every kernel slower: the page kernels are
        // latency chains, and four of them sharing the CUs doubled the binary selector's time.)
        auto stream_of = [&](size_t ki) -> hipStream_t {
            if (!multi) return s;
            if (n_bin_kinds) return kinds[ki] < 0 ? ctx->side[0] : s;
            return (ki & 1) ? ctx->side[1] : s;
        };
        uint32_t side_mask = 0;
        if (multi) side_mask = n_bin_kinds ? 1u : 2u;
        if (multi && n_bin_kinds == (int)kinds.size()) side_mask = 0;   // (only binary kinds: nothing to overlap with)
        auto launch_hash = [&](hipStream_t st) {   // row hashes of the binary pages, tile-parallel, before their selector
            if (aa.bin_fused && plan.bin_pages) {
                KScope k(ctx, "k_enc_bin_page");
                for (int kd : kinds) {
                    if (kd == -4) k_enc_bin_page<int32_t><<<(uint32_t)P, BP_WG, 0, st>>>(aa);
                    if (kd == -8) k_enc_bin_page<int64_t><<<(uint32_t)P, BP_WG, 0, st>>>(aa);
                }
            }
            if (!old_chain) return;
            KScope k(ctx, "k_enc_bin_hash");
            k_enc_bin_hash<<<tile_grid, WG, 0, st>>>(aa);
        };
        // long pages (>= 2^18 rows) of 1- / 2- / 4- / 8-byte values: section-parallel statistics, the same decision, and the
        // RLE pages among them written section-parallel too (sb_select_big.h); they were left CODEC_PENDING by the page selectors
        auto launch_big = [&](int kd, hipStream_t st) {
            const int k = kd == 1 ? 0 : kd == 2 ? 1 : kd == 4 ? 2 : kd == 8 ? 3 : kd < 0 ? 4 : -1;   // (binary pages: a list of their own)
            if (k < 0 || aa.page_base != 0) return;
            const uint32_t nbig = (uint32_t)plan.bigw[k].size();
            if (!nbig) return;
            size_t skip = 0;
            for (int q = 0; q < k; q++) skip += plan.bigw[q].size();
            const uint32_t* list = (const uint32_t*)plan.big.p + skip;
            const dim3 sg(plan.big_secs[k], nbig), pg(1, nbig);
#define SB_BIG_W(KERNEL, GRID, THREADS)                                    \
    do {                                                                   \
        if (kd == 1) KERNEL<1><<<GRID, THREADS, 0, st>>>(aa, list, 0u);        \
        else if (kd == 2) KERNEL<2><<<GRID, THREADS, 0, st>>>(aa, list, 0u);   \
        else if (kd == 4) KERNEL<4><<<GRID, THREADS, 0, st>>>(aa, list, 0u);   \
        else KERNEL<8><<<GRID, THREADS, 0, st>>>(aa, list, 0u);   /* 8-byte values, and binary pages as their u64 row hashes */ \
    } while (0)
            {
                KScope kk(ctx, "k_sel_big_sec");
                k_sel_big_init<<<dim3(4, nbig), WG, 0, st>>>(aa, list, 0u);
                const dim3 sgp(sg.x * BIG_SEC_SPLIT, nbig);
                SB_BIG_W(k_sel_big_sec, sgp, WG);
            }
            {
                KScope kk(ctx, "k_sel_big_merge");
                SB_BIG_W(k_sel_big_merge, pg, WG);
            }
            {
                KScope kk(ctx, "k_sel_big_clear");
                k_sel_big_clear<<<sg, WG, 0, st>>>(aa, list, 0u);
            }
            {
                KScope kk(ctx, "k_sel_big_count");
                const dim3 cg(sg.x * BIG_COUNT_SPLIT, nbig);
                SB_BIG_W(k_sel_big_count, cg, WG);
            }
            {
                KScope kk(ctx, "k_sel_big_decide");
                SB_BIG_W(k_sel_big_decide, pg, WG);
            }
            if (!((forb >> SB_CODEC_RLE) & 1) && kd > 0) {
                KScope kk(ctx, "k_rle_big");
                SB_BIG_W(k_rle_big_count, sg, WG);
                SB_BIG_W(k_rle_big_plan, pg, WG);
                SB_BIG_W(k_rle_big_emit, sg, WG);
                SB_BIG_W(k_rle_big_done, pg, 64);
            }
#undef SB_BIG_W
            if (!((forb >> SB_CODEC_DICT) & 1) && (!big_hint || plan.last_counts[SB_CODEC_DICT])) {   // long Dict pages: sb_dict_big.h
                const dim3 gg(256, nbig), sg4(sg.x * DBIG_SPLIT, nbig);
                const uint32_t vo = (uint32_t)P;
#define SB_DBIG_W(KERNEL, GRID, THREADS)                                   \
    do {                                                                   \
        if (kd == 1) KERNEL<1><<<GRID, THREADS, 0, st>>>(aa, list);        \
        else if (kd == 2) KERNEL<2><<<GRID, THREADS, 0, st>>>(aa, list);   \
        else if (kd == 4) KERNEL<4><<<GRID, THREADS, 0, st>>>(aa, list);   \
        else if (kd == 8) KERNEL<8><<<GRID, THREADS, 0, st>>>(aa, list);   \
        else if (kd == -4) KERNEL<-4><<<GRID, THREADS, 0, st>>>(aa, list); \
        else KERNEL<-8><<<GRID, THREADS, 0, st>>>(aa, list);               \
    } while (0)
                {
                    KScope kk(ctx, "k_dict_big_insert");
                    SB_DBIG_W(k_dict_big_clear, gg, WG);
                    SB_DBIG_W(k_dict_big_insert, sg4, WG);
                }
                {
                    KScope kk(ctx, "k_dict_big_ids");
                    SB_DBIG_W(k_dict_big_mark, gg, WG);
                    SB_DBIG_W(k_dict_big_rank, sg, WG);
                    SB_DBIG_W(k_dict_big_ids, gg, WG);
                }
                {
                    KScope kk(ctx, "k_dict_big_idx");
                    SB_DBIG_W(k_dict_big_idx, sg4, WG);
                    if (kd == -4) k_dict_big_verify<-4><<<sg4, WG, 0, st>>>(aa, list);
                    else if (kd == -8) k_dict_big_verify<-8><<<sg4, WG, 0, st>>>(aa, list);
                }
                {   // the index arrays as virtual pages of u32: selected ...
                    KScope kk(ctx, "k_sel_big(indices)");
                    k_sel_big_init<<<dim3(4, nbig), WG, 0, st>>>(aa, list, vo);
                    k_sel_big_sec<4><<<dim3(sg.x * BIG_SEC_SPLIT, nbig), WG, 0, st>>>(aa, list, vo);
                    k_sel_big_merge<4><<<pg, WG, 0, st>>>(aa, list, vo);
                    k_sel_big_count<4><<<dim3(sg.x * BIG_COUNT_SPLIT, nbig), WG, 0, st>>>(aa, list, vo);   // (BIG_COUNT_SPLIT workgroups per section)
                    k_sel_big_decide<4><<<pg, WG, 0, st>>>(aa, list, vo);
                }
                {   // ... and written
                    KScope kk(ctx, "k_nested_big");
                    if (!((forb >> SB_CODEC_RLE) & 1)) {
                        k_rle_big_count<4><<<sg, WG, 0, st>>>(aa, list, vo);
                        k_rle_big_plan<4><<<pg, WG, 0, st>>>(aa, list, vo);
                        k_rle_big_emit<4><<<sg, WG, 0, st>>>(aa, list, vo);
                        k_rle_big_done<4><<<pg, 64, 0, st>>>(aa, list, vo);
                    }
                    const dim3 tg((uint32_t)std::min<uint64_t>(max_tiles, 4096), nbig);
                    k_bp_big<0><<<tg, WG, 0, st>>>(aa, list);
                    k_bp_big<1><<<pg, WG, 0, st>>>(aa, list);
                    k_bp_big<2><<<tg, WG, 0, st>>>(aa, list);
                    k_plain_big<<<dim3(1024, nbig), WG, 0, st>>>(aa, list);
                }
                {
                    KScope kk(ctx, "k_dict_big_finish");
                    SB_DBIG_W(k_dict_big_finish, pg, WG);
                    SB_DBIG_W(k_dict_big_values, gg, WG);
                }
#undef SB_DBIG_W
            }
        };
        auto launch_selectors = [&](int kd, hipStream_t st) {
            if (nested && (kd <= 0 || kd > 8)) return;
            if (!nested && (kd == 4 || kd == 8)) {  // statistics + speculative RLE in one pass
                bool any_f = false, any_i = false;
                for (uint64_t i = 0; i < n; i++)
                    if ((int)hc[i].width == kd && hc[i].ptype != SB_TYPE_BOOLEAN && !enc_is_binary(hc[i].ptype)) {
                        any_f |= hc[i].fkind != 0;
                        any_i |= hc[i].fkind == 0;
                    }
                // lane = raw run first; pages with runs too short for that go through the lane = row kernel
                if (any_f) {
                    {
                        KScope k(ctx, kd == 4 ? "k_enc_select_runs<4, 1>" : "k_enc_select_runs<8, 2>");
                        if (kd == 4)
                            k_enc_select_runs<4, 1><<<(uint32_t)P, WG, 0, st>>>(aa);
                        else
                            k_enc_select_runs<8, 2><<<(uint32_t)P, WG, 0, st>>>(aa);
                    }
                    if (!unused(31)) {   // (pages the run-level kernel left to the row-level one)
                        KScope k(ctx, kd == 4 ? "k_enc_select_rle<4, 1>" : "k_enc_select_rle<8, 2>");
                        if (kd == 4)
                            k_enc_select_rle<4, 1><<<(uint32_t)P, WG, 0, st>>>(aa);
                        else
                            k_enc_select_rle<8, 2><<<(uint32_t)P, WG, 0, st>>>(aa);
                    }
                }
                if (any_i) {
                    {
                        KScope k(ctx, kd == 4 ? "k_enc_select_runs<4, 0>" : "k_enc_select_runs<8, 0>");
                        if (kd == 4)
                            k_enc_select_runs<4, 0><<<(uint32_t)P, WG, 0, st>>>(aa);
                        else
                            k_enc_select_runs<8, 0><<<(uint32_t)P, WG, 0, st>>>(aa);
                    }
                    if (!unused(31)) {
                        KScope k(ctx, kd == 4 ? "k_enc_select_rle<4, 0>" : "k_enc_select_rle<8, 0>");
                        if (kd == 4)
                            k_enc_select_rle<4, 0><<<(uint32_t)P, WG, 0, st>>>(aa);
                        else
                            k_enc_select_rle<8, 0><<<(uint32_t)P, WG, 0, st>>>(aa);
                    }
                }
                launch_big(kd, st);
                return;
            }
            char nm[48];
            snprintf(nm, sizeof nm, "k_enc_select<%d>", kd);
            if (kd > 0 || kd == 0 || old_chain) {
                KScope k(ctx, nm);
                enc_select_kernel(kd)<<<(uint32_t)P, WG, 0, st>>>(aa);
            }
            if (!nested && (kd == 1 || kd == 2 || kd < 0)) launch_big(kd, st);
        };
        // the dictionaries the binary selectors handed over: strings checked tile-parallel, pages that failed selected again
        // exactly (workgroups of all other pages return at once); kd_only: the one binary kind of this stream, or 0 = both
        auto launch_verify = [&](int kd_only, hipStream_t st) {
            if (!old_chain) return;
            {
                KScope k(ctx, "k_enc_bin_verify");
                const uint32_t tpp = (uint32_t)((max_rows + BV_ROWS - 1) / BV_ROWS);
            k_enc_bin_verify<<<(uint32_t)(((P + 7) / 8) * 8 * tpp), WG, 0, st>>>(aa, tpp);
            }
            aa.redo = 1;
            for (int kd : kinds)
                if (kd < 0 && (!kd_only || kd == kd_only)) enc_select_kernel(kd)<<<(uint32_t)P, WG, 0, st>>>(aa);
            aa.redo = 0;
        };
        auto launch_emit = [&](int kd, hipStream_t st) -> int32_t {
            // one kernel instance per (kind, codec) that can occur in the batch
            static const int32_t CAND[6] = {SB_CODEC_ONEVALUE, SB_CODEC_DICT, SB_CODEC_RLE, SB_CODEC_BITPACKING,
                                            SB_CODEC_DELTA_BITPACKING, SB_CODEC_PATAS};
            if (nested && (kd <= 0 || kd > 8)) return SB_OK;
            for (int32_t cd : CAND) {
                if (!wave_adaptive && cd != wave_codec) continue;
                if (wave_adaptive) {
                    if ((forb >> cd) & 1) continue;
                    if ((cd == SB_CODEC_BITPACKING || cd == SB_CODEC_DELTA_BITPACKING) && kd != 4) continue;
                    if (kd == 0 && cd == SB_CODEC_DICT) continue;
                    if (kd < 0 && cd == SB_CODEC_RLE) continue;
                    if (cd == SB_CODEC_PATAS) {  // candidates of float columns only (double/mod.rs:271-277)
                        bool any_float = false;
                        for (uint64_t i = 0; i < n; i++) any_float |= hc[i].fkind != 0 && (int)hc[i].width == kd;
                        if (!any_float) continue;
                    }
                }
                EncPageKernel kf = enc_page_kernel(kd, cd);
                if (!kf) continue;
                if (cd == SB_CODEC_DICT && aa.bin_fused && (kd == 1 || kd == 2 || kd == 4) && !unused_peek(SB_CODEC_DICT)) {
                    KScope k(ctx, "k_enc_prim_dict");   // integer Dict pages of up to 65 536 rows: sb_bin_page.h
                    if (kd == 4) k_enc_prim_dict<4><<<(uint32_t)P, BP_WG, 0, st>>>(aa);
                    else if (kd == 2) k_enc_prim_dict<2><<<(uint32_t)P, BP_WG, 0, st>>>(aa);
                    else k_enc_prim_dict<1><<<(uint32_t)P, BP_WG, 0, st>>>(aa);
                }
                // (4- / 8-byte RLE pages come out of the fused selectors: slot 29 counts the ones the page kernel had to write)
                // (binary Dict pages come out of k_enc_bin_page whole when their index block is bit-packed: slot 30 likewise)
                if (wave_adaptive && !nested &&
                    unused(cd == SB_CODEC_RLE && (kd == 4 || kd == 8) ? 29 : cd == SB_CODEC_DICT && kd < 0 && aa.bin_fused && !old_chain ? 30 : (int)cd))
                    continue;
                char nm[48];
                snprintf(nm, sizeof nm, "k_enc_emit_pages<%d, %d>", kd, (int)cd);
                KScope k(ctx, nm);
                kf<<<(uint32_t)P, WG, 0, st>>>(aa);
            }
            if (!nested && !wave_adaptive && wave_codec != SB_CODEC_NONE && wave_codec != SB_CODEC_FREQ && wave_codec > 3 &&
                !enc_page_kernel(kd, wave_codec))
                return ctx->fail(SB_ERR_NYI, "no device encoder for this codec and column type");
            return SB_OK;
        };
        const bool emit_pages_wanted = nested || any_pages;

        if (multi) {
            if (any_bin) aa.pre_hashed = 1;
            side_fork(ctx, side_mask);
            int32_t rc = SB_OK;
            if (any_bin) {   // the binary chain, in order, on the high-priority side stream
                hipStream_t st = ctx->side[0];
                launch_hash(st);
                for (int kd : kinds)
                    if (kd < 0) launch_selectors(kd, st);
                launch_verify(0, st);
                for (int kd : kinds)
                    if (kd < 0 && emit_pages_wanted && rc == SB_OK) rc = launch_emit(kd, st);
            }
            for (size_t ki = 0; ki < kinds.size() && rc == SB_OK; ki++) {
                const int kd = kinds[ki];
                if (kd < 0) continue;
                hipStream_t st = stream_of(ki);
                launch_selectors(kd, st);
                if (emit_pages_wanted) rc = launch_emit(kd, st);
            }
            side_join(ctx, side_mask);
            if (rc != SB_OK) return rc;
        } else {
            if (wave_adaptive && !nested && any_bin) {
                launch_hash(s);
                aa.pre_hashed = 1;
            }
            if (wave_adaptive)
                for (int kd : kinds) launch_selectors(kd, s);
            if (aa.pre_hashed) launch_verify(0, s);
        }
        const int32_t dc = opts->default_compression;
        const bool basic_comp = dc == SB_CODEC_LZ4 || dc == SB_CODEC_ZSTD || dc == SB_CODEC_SNAPPY;
        if (!nested && wave_adaptive && any_tiles && !aa.null_cols && unused(SB_CODEC_NONE)) {
            // (no plain page last time; Null columns' empty pages are recorded by this kernel: never skipped then)
        } else if (nested ? (wave_adaptive ? dc == SB_CODEC_NONE : wave_codec == SB_CODEC_NONE) : any_tiles) {
            KScope k(ctx, K_ENC_TILES);
            // (adaptive: few pages stay plain, so one workgroup per page looks — unless the pages are few and long)
            const uint32_t ty = wave_adaptive ? (uint32_t)std::min<uint64_t>(max_tiles, std::max<uint64_t>(1, 2048 / P)) : (uint32_t)max_tiles;
            k_enc_emit_tiles<<<dim3((uint32_t)P, ty), WG, 0, s>>>(aa, (uint32_t)max_tiles);
        }
        const bool basic_unused = !nested && wave_adaptive && any_lz4 && big_hint && !plan.last_counts[SB_CODEC_LZ4] && !plan.last_counts[SB_CODEC_ZSTD] &&
                                  !plan.last_counts[SB_CODEC_SNAPPY];
        if (basic_unused) {
            emit_skips |= SKIP_EMIT;
        } else if (nested ? (wave_adaptive ? basic_comp : (wave_codec >= 1 && wave_codec <= 3)) : any_lz4) {
            if (!nested && aa.lzc_plan) {   // blocks of more than one chunk: one wave per chunk, then joined
                {
                    KScope k(ctx, "k_enc_lz4_plan");
                    k_enc_lz4_plan<<<dim3((uint32_t)P, (uint32_t)std::min<uint64_t>(std::max<uint64_t>(1, max_tiles / 8), std::max<uint64_t>(1, 1024 / P))), WG, 0, s>>>(aa);
                }
                if (aa.zpar_scratch) {
                    KScope k(ctx, "k_enc_zstd_chunks");
                    k_enc_zstd_chunks<<<zpar_waves, 64, 0, s>>>(aa);
                } else {
                    KScope k(ctx, "k_enc_lz4_chunks");
                    k_enc_lz4_chunks<<<(uint32_t)std::min<uint64_t>(aa.lzc_cap, 1u << 20), 64, 0, s>>>(aa);
                }
                KScope k(ctx, "k_enc_lz4_stitch");
                k_enc_lz4_stitch<<<dim3((uint32_t)P, (uint32_t)std::min<uint64_t>(std::max<uint64_t>(1, max_chunks / 4), std::max<uint64_t>(1, 2048 / P))), WG, 0, s>>>(aa);
            }
            KScope k(ctx, K_ENC_LZ4);
            k_enc_emit_lz4<false><<<(uint32_t)P, WG, 0, s>>>(aa);
            if (dc == SB_CODEC_ZSTD || wave_codec == SB_CODEC_ZSTD) {   // (virtual pages inherit their page's encoder scratch)
                KScope kz(ctx, "k_enc_emit_lz4<true>");
                k_enc_emit_lz4<true><<<(uint32_t)P, WG, 0, s>>>(aa);
            }
        }
        if (emit_pages_wanted && !multi) {
            for (int kd : kinds) {
                const int32_t rc = launch_emit(kd, s);
                if (rc != SB_OK) return rc;
            }
        }
        return SB_OK;
    };
    {
        const int32_t rc = run_wave(a, adaptive, host_codec, false);
        if (rc != SB_OK) return rc;
    }
    if (freq_possible && adaptive && (!big_hint || plan.last_counts[SB_CODEC_FREQ])) {
        // long pages that chose Freq: prepared container-parallel, their exceptions block (a virtual page) selected and
        // written section-parallel when it has VBIG_ROWS rows or more (sb_freq_big.h, sb_dict_big.h)
        const uint32_t vo = (uint32_t)P;
        EncCol* vcols_rw = (EncCol*)(tb + o_vcols);
        EncPage* vpages_rw = (EncPage*)(tb + o_vpages);
        for (int k = 0; k < 4; k++) {
            const uint32_t nbig = (uint32_t)plan.bigw[k].size();
            if (!nbig) continue;
            size_t skip = 0;
            for (int q = 0; q < k; q++) skip += plan.bigw[q].size();
            const uint32_t* list = (const uint32_t*)plan.big.p + skip;
            const dim3 sg(plan.big_secs[k], nbig), pg(1, nbig);
            const dim3 cg((uint32_t)((max_tiles * TILE_ROWS + 65535) / 65536), nbig);
#define SB_FBIG_W(KERNEL, GRID, THREADS, ...)                                             \
    do {                                                                                  \
        if (k == 0) KERNEL<1><<<GRID, THREADS, 0, s>>>(a, list, ##__VA_ARGS__);           \
        else if (k == 1) KERNEL<2><<<GRID, THREADS, 0, s>>>(a, list, ##__VA_ARGS__);      \
        else if (k == 2) KERNEL<4><<<GRID, THREADS, 0, s>>>(a, list, ##__VA_ARGS__);      \
        else KERNEL<8><<<GRID, THREADS, 0, s>>>(a, list, ##__VA_ARGS__);                  \
    } while (0)
            {
                KScope kk(ctx, "k_freq_big");
                SB_FBIG_W(k_freq_big_count, dim3(cg.x * 4, nbig), WG);
                SB_FBIG_W(k_freq_big_plan, pg, WG, vcols_rw, vpages_rw);
                SB_FBIG_W(k_freq_big_emit, cg, WG);
                SB_FBIG_W(k_freq_big_done, pg, 64);
            }
            {
                KScope kk(ctx, "k_sel_big(exceptions)");
                k_sel_big_init<<<dim3(4, nbig), WG, 0, s>>>(a, list, vo);
                SB_FBIG_W(k_sel_big_sec, dim3(sg.x * BIG_SEC_SPLIT, nbig), WG, vo);
                SB_FBIG_W(k_sel_big_merge, pg, WG, vo);
                k_sel_big_clear<<<sg, WG, 0, s>>>(a, list, vo);
                SB_FBIG_W(k_sel_big_count, dim3(sg.x * BIG_COUNT_SPLIT, nbig), WG, vo);
                SB_FBIG_W(k_sel_big_decide, pg, WG, vo);
            }
            {
                KScope kk(ctx, "k_nested_big");
                if (!((forb >> SB_CODEC_RLE) & 1)) {
                    SB_FBIG_W(k_rle_big_count, sg, WG, vo);
                    SB_FBIG_W(k_rle_big_plan, pg, WG, vo);
                    SB_FBIG_W(k_rle_big_emit, sg, WG, vo);
                    SB_FBIG_W(k_rle_big_done, pg, 64, vo);
                }
                if (k == 2) {
                    const dim3 tg((uint32_t)std::min<uint64_t>(max_tiles, 4096), nbig);
                    k_bp_big<0><<<tg, WG, 0, s>>>(a, list);
                    k_bp_big<1><<<pg, WG, 0, s>>>(a, list);
                    k_bp_big<2><<<tg, WG, 0, s>>>(a, list);
                }
                k_plain_big<<<dim3(1024, nbig), WG, 0, s>>>(a, list);
            }
#undef SB_FBIG_W
        }
    }
    if (freq_possible && adaptive && unused(28) ) {
        // (no page went through the Freq kernels last time: not launched)
    } else if (freq_possible) {  // Freq pages: bitmap + exceptions, then the exceptions block like any other block
        {
            KScope k(ctx, K_ENC_FREQ);
            k_enc_freq_prep<<<(uint32_t)std::min<uint64_t>(P, 1024), WG, 0, s>>>(a, (EncCol*)(tb + o_vcols), (EncPage*)(tb + o_vpages));
        }
        bool has4 = false, wide = false;
        for (int kd : kinds) has4 |= kd == 4;
        for (uint64_t i = 0; i < n; i++)  // only integers get there: a mostly-one-value float column takes Freq itself
            wide |= (hc[i].fkind == 0 || dict_freq) && hc[i].ptype != SB_TYPE_BOOLEAN && hc[i].ptype != SB_TYPE_NULL && !enc_is_binary(hc[i].ptype) &&
                    (hc[i].width <= 2 || hc[i].width >= 8);
        for (uint64_t i = 0; i < n; i++) wide |= enc_is_binary(hc[i].ptype);
        if (!has4 && wide && (adaptive || dict_freq) && !((forb >> SB_CODEC_DICT) & 1)) {  // Freq-coded u32 indices of Dict pages
            KScope k(ctx, K_ENC_FREQ);
            k_enc_nested<4><<<(uint32_t)std::min<uint64_t>(P, 512), WG, 0, s>>>(a);
        }
        for (int kd : kinds) {
            if (kd != 1 && kd != 2 && kd != 4 && kd != 8 && kd != 16 && kd != 32) continue;
            KScope k(ctx, K_ENC_FREQ);
            if (kd == 16)
                k_enc_nested<16><<<(uint32_t)std::min<uint64_t>(P, 512), WG, 0, s>>>(a);
            else if (kd == 32)
                k_enc_nested<32><<<(uint32_t)std::min<uint64_t>(P, 512), WG, 0, s>>>(a);
            else if (kd == 1)
                k_enc_nested<1><<<(uint32_t)std::min<uint64_t>(P, 512), WG, 0, s>>>(a);
            else if (kd == 2)
                k_enc_nested<2><<<(uint32_t)std::min<uint64_t>(P, 512), WG, 0, s>>>(a);
            else if (kd == 4)
                k_enc_nested<4><<<(uint32_t)std::min<uint64_t>(P, 512), WG, 0, s>>>(a);
            else
                k_enc_nested<8><<<(uint32_t)std::min<uint64_t>(P, 512), WG, 0, s>>>(a);
        }
        KScope k(ctx, K_ENC_FREQ);
        k_enc_freq_finish<<<(uint32_t)P, WG, 0, s>>>(a);
    }
    a.skips |= emit_skips;
    {
        KScope k(ctx, K_ENC_LAYOUT);
        k_enc_layout<<<(uint32_t)n, 64, 0, s>>>(a, (const uint64_t*)(tb + o_resoff));
    }
    if (any_compact) {
        KScope k(ctx, K_ENC_COMPACT);
        k_enc_compact<<<dim3((uint32_t)P, (uint32_t)std::min<uint64_t>(max_chunks, std::max<uint64_t>(1, 8192 / P))), WG, 0, s>>>(a);
    }
    e = hipGetLastError();
    if (e != hipSuccess) return check_hip(ctx, e, "encode launch");

    uint8_t* hres = slot->host + o_hres;
    e = hipMemcpyAsync(hres, a.results, results_words * sizeof(uint64_t), hipMemcpyDeviceToHost, s);
    if (e != hipSuccess) return check_hip(ctx, e, "metas readback");
    if (adaptive) {   // the pages per codec, for the next call with this plan
        e = hipMemcpyAsync(slot->host + o_hcounts, a.codec_counts, 128, hipMemcpyDeviceToHost, s);
        if (e != hipSuccess) return check_hip(ctx, e, "codec counts readback");
        Pending pd;
        pd.kind = Pending::ENC_HINT;
        pd.user = nullptr;
        pd.host = slot->host + o_hcounts;
        pd.n = plan_key;
        ctx->pending.push_back(pd);
    }
    (void)hipEventRecord(slot->done, s);
    slot->in_flight = true;
    ctx->last_slot = slot;
    for (uint64_t i = 0; i < n; i++) {
        Pending pd;
        pd.kind = Pending::WRITE_COL;
        pd.user = &cols[i];
        pd.host = hres + hro[i] * sizeof(uint64_t);
        pd.n = hc[i].n_pages;
        ctx->pending.push_back(pd);
    }
    return SB_OK;
}

}  // extern "C"
