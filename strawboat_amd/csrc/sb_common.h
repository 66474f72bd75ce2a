// strawboat-hip: structures shared by host code and gfx950 kernels.
//
// Data layout in HBM: a call works on a batch of leaf columns.  The host
// uploads one ColDesc per column and one PageTask per page; the parse kernel turns each
// page's headers into a PageDesc and fills the tile table; every later kernel is indexed
// either by page (plan kernels, one workgroup per page) or by tile (expand kernels, one
// workgroup per TILE_ROWS rows of one page).
//
// MEMORY-ORDER CONTRACT for scratch areas in HBM (slot tables, row hashes, tag tables, run records, tile entries).
// Fences inside the page kernels are WORKGROUP scope (an agent-scope release / acquire writes back and invalidates the
// XCD's L2: ~100 us per fence once the chip is busy).  Workgroup scope orders the accesses of the waves of ONE workgroup —
// they share a CU and its vector L1 — and says nothing to another CU.  So every scratch area obeys one of two rules:
//   (1) SINGLE OWNER inside a kernel: between two kernel boundaries, the bytes of a scratch area that one workgroup
//       writes are read by that workgroup only.  The owner is fixed by the launch geometry: blockIdx.x = page for the page
//       kernels, (blockIdx.x, blockIdx.y) = (page, tile / section) for the tile- and section-parallel ones, whose
//       areas are indexed by that pair.  A workgroup never reads another (page, tile)'s scratch inside the kernel
//       that wrote it.
//   (2) KERNEL BOUNDARY between owners: a consumer with another geometry (k_enc_bin_hash's (page, tile) grid -> the
//       page selector -> k_enc_bin_verify's grid -> the page emitters; k_sel_big_count -> _merge -> _sec; k_rle_big_count
//       -> _plan -> _emit; k_parse -> k_plan -> k_expand*) is a LATER KERNEL, ordered after the producer by stream order
//       or by a fork / join event edge when the chain runs on a side stream (side_fork / side_join, sb_api.hip): the
//       end-of-kernel release and start-of-kernel acquire are agent scope and do what the in-kernel fences do not.
// What workgroups of ONE launch do share is touched with AGENT-scope operations only, and says so where it stands: job
// queues and counters (atomicAdd / __hip_atomic_fetch_add, push_job_if), the key tables in HBM that all sections of a long
// page insert into (SlotTable / k_sel_big_count: agent-scope load, CAS and fetch_min — a table is complete at the kernel
// boundary; the one in-kernel read of shared progress is the relaxed agent-scope load of the distinct counter by which
// sections stop early once Dict's limit is passed), and last_workgroup_done (sb_decode.hip: __threadfence() = agent-scope
// release before the counter, acquire after it in the workgroup that finds itself last).
// tests/test_gpu_configs.py::test_c4_all_columns_in_one_call runs the binary chain on the side stream next to the
// primitive kinds (sb_ctx_side_forks shows it did) and compares every page with the oracle's.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/strawboat_hip.h"

namespace sb {

constexpr int TILE_ROWS = 4096;  // rows of one page handled by one workgroup
constexpr int WG = 256;          // threads per workgroup = 4 wave64
constexpr int ROWS_PER_THREAD = TILE_ROWS / WG;

struct ColDesc {
    const uint8_t* pages;  // concatenated pages of the column
    uint64_t pages_len;
    uint8_t* values;
    uint64_t values_cap;
    uint8_t* validity;
    uint8_t* offsets;
    uint64_t offsets_cap;
    uint64_t rows;
    int32_t ptype;
    int32_t nullable;
    uint32_t width;       // bytes per value (primitives), 4/8 for binary offsets, 0 bool
    uint32_t first_page;  // index into the page tables
    uint32_t n_pages;
    uint32_t bits_aligned;  // every page starts at a multiple of 32 rows: bitmap words are never shared
};

struct PageTask {
    uint64_t in_off;      // byte offset of the page in ColDesc.pages
    uint64_t length;      // PageMeta.length
    uint64_t num_values;  // PageMeta.num_values
    uint64_t out_row;     // first row of the page in the column's output
    uint64_t aux_off;     // byte offset of the page's u32 aux area in the scratch buffer
    uint64_t infl_off;    // byte offset of the page's inflate area in the scratch buffer
    uint32_t col;
    uint32_t first_tile;
};

struct TileTask {
    uint32_t page;
    uint32_t tile;
    uint32_t col;  // = tasks[page].col: lets k_expand fetch its three descriptors in one round trip
    uint32_t k0;   // RLE pages (k_plan): first run of the tile ...
    uint32_t kend; // ... and one past its last run; 0 / 0 otherwise
    uint32_t pad;
};

// result of parsing one page's headers (hdr9 = u8 codec | u32 compressed | u32 uncompressed,
// reference src/read/read_basic.rs:181-189)
struct PageDesc {
    const uint8_t* def_bits;  // validity bits of the def-level section (NULL: not nullable)
    const uint8_t* body;      // body of the page's (first) block
    const uint8_t* ibody;     // Dict: body of the nested u32 index block
    const uint8_t* dict;      // Dict: first entry; binary OneValue: value bytes
    const uint8_t* vbody;     // binary Basic: body of the values block
    const uint8_t* src;       // where the expand step reads "plain" data from (body or inflated copy)
    const uint8_t* isrc;      // same for nested indices
    uint32_t csize, usize;    // of the first block
    uint32_t icsize;          // nested block compressed size
    uint32_t dict_n;          // Dict: entries; binary OneValue: value length
    uint32_t vcsize, vusize;  // binary Basic values block
    uint8_t codec;            // page codec
    uint8_t icodec;           // nested codec (Dict) or 255
    uint8_t ok;               // 1 = parsed, expand may run
    uint8_t pad;
    uint32_t n_runs;          // RLE: number of runs (filled by plan)
    uint32_t tile_base;       // first entry of the page in the compact tile list (k_parse)
    uint32_t pad2;
    uint64_t val_bytes;       // binary: value bytes this page produces
    uint64_t val_base;        // binary: first value byte of the page in the column output (colscan)
    uint64_t off_last;        // binary: last offset of the page as decoded (page relative)
    uint64_t off_base;        // binary: offset base of the page (colscan)
};

struct Status {
    int32_t code;   // first error (SB_ERR_*), 0 = ok
    uint32_t page;  // page index within the call
    uint32_t where; // kernel-specific tag
    uint32_t kinds; // KIND_* bits of what the calls have met so far (never cleared: the host sizes later calls by it)
};
constexpr uint32_t KIND_ZSTD = 1u;   // a Zstd buffer was queued
constexpr uint32_t KIND_LZ4_GIANT = 4u;   // an LZ4 block of >= LZG_MIN compressed bytes was queued (sb_lz4_giant.h is launched for contexts that meet them)
constexpr uint32_t KIND_REPLAY = 8u;      // a page was left undone because a kernel it needed was not launched on a hint: sb_ctx_synchronize re-issues the interval's calls with everything launched
constexpr uint32_t KIND_REPLAY_LZG = 16u; // ... and it was an LZ4 block of megabytes that asked (the context turns the block-parallel chain on again)
constexpr uint32_t KIND_QUEUE_A = 32u;    // a read call queued inflate jobs for queue A (Basic blocks of primitive pages, index / offset blocks)
constexpr uint32_t KIND_TILES = 64u;      // a read call had tile tasks (k_expand / k_expand_binary)
constexpr uint32_t KIND_ZSEQ_LONG = 2u;   // a Zstd block of >= 8192 sequences was met (zb_hdr): the sequence chains are the long pole

// one general-purpose block (LZ4 / Zstd / Snappy) to inflate: src -> dst
struct InflateJob {
    const uint8_t* src;
    uint8_t* dst;
    uint32_t csize;
    uint32_t out_len;
    uint32_t codec;     // codec id | JOB_REL
    uint32_t page;
};
constexpr uint32_t JOB_REL = 0x100;   // dst is a byte offset from the page's value base (cols[col].values + descs[page].val_base)

// one Freq page (integer/freq.rs:90-127), logged by k_parse: its exceptions block is an ordinary
// BLOCK<T> that a second decode pass expands, then k_freq_scatter puts the values in place
struct FreqEntry {
    const uint8_t* roaring;  // serialized RoaringBitmap of the exception rows
    const uint8_t* nested;   // BLOCK<T exceptions>
    uint8_t* out;            // the page's values in the column output
    uint64_t nested_len;     // bytes from `nested` to the end of the page
    uint64_t rows;
    uint32_t roaring_len;
    uint32_t n_exceptions;
    uint32_t ptype, width;
    uint32_t page, pad;
};

// ---- the block-parallel Zstd pipeline (sb_zstd_blocks.h): descriptors and pools of one call
struct ZbBlock {
    const uint8_t* src;     // block content (after the 3-byte block header)
    uint32_t bsize;         // content bytes in the stream (RLE block: 1)
    uint32_t frame;
    uint32_t btype;         // 0 raw, 1 RLE, 2 compressed
    uint32_t out_size;      // raw / RLE: from the header; compressed: literals + match bytes (zb_seq; nseq == 0: regen)
    uint32_t ltype, lstreams, regen, lcsize, lpay;   // literals section: type, streams, regenerated / compressed size, payload offset
    uint32_t huf_def;       // block whose literal payload starts with the Huffman tree in use
    uint32_t nseq, modes, seq_off;                   // sequences: count, modes byte, offset of the byte after it
    uint32_t def[3];        // LL, OF, ML: block whose description defines the table in use; ZB_NONE: predefined
    uint32_t desc[3];       // zb_hdr: offsets (in this block) of its own descriptions (modes 1 and 2)
    uint32_t bits_off;      // zb_hdr: offset of the sequence bit stream
    uint64_t lit_pos;       // the block's literals in the literal pool (types 1, 2, 3)
    uint64_t rec_pos;       // the block's records in the record pool
};
struct ZbFrame {
    uint8_t* dst;
    uint32_t out_len;
    uint32_t first, nblocks;
    uint32_t job;           // queue entry
    uint32_t page;
    uint32_t punt;          // != 0: the one-wave decoder takes the frame
    uint32_t avail;         // bytes of the queue entry's buffer (loads never reach beyond it)
    uint32_t queue;         // 0: queue A, 1: queue Z (executed after k_colscan)
    uint32_t rel;           // dst is relative to the page's value base (JOB_REL)
    uint32_t wg;            // executed by zb_exec_wg (many short sequences) instead of zb_exec
    const uint8_t* base;    // the queue entry's buffer
};
struct ZbPools {
    ZbBlock* blocks;
    ZbFrame* frames;
    uint8_t* lit;
    uint64_t* rec;          // 12 bytes per sequence: literal length, match length, offset value (rec_pos counts sequences)
    uint32_t* counters;     // [0] blocks, [1] frames, [4..5] literal bytes, [6..7] records, [8..11] blocks with sequences per size class
    uint32_t* lists;        // zb_hdr: the blocks with sequences, by size class (4 x block_cap indices)
    uint32_t block_cap, frame_cap;
    uint64_t lit_cap, rec_cap;
    uint32_t min_csize;     // frames shorter than this stay with the one-wave / lane-per-frame paths
    uint32_t wg_exec;       // 0: every frame through the wave executor (SB_ZSTD_BLOCKS_WG=0)
    unsigned long long* stats;   // totals of the context: [0] frames decoded, [1] frames handed back, [2] blocks, [3] sequences
    uint32_t* kinds;             // Status.kinds of the call
};


// the pool of inflate waves (k_inflate) and its per-wave areas
constexpr uint32_t INFLATE_POOL = 2048;          // waves
constexpr uint32_t ZREC_PER_WAVE = 128 * 1024;   // pre-decoded Zstd sequence records (8 bytes each) per pool wave

// one LZ4 block of megabytes decoded by the whole chip (sb_lz4_giant.h)
constexpr uint32_t LZG_MIN = 2u << 20;     // compressed bytes
constexpr uint32_t LZG_JOBS = 16;
constexpr uint32_t LZG_CH = 4096;          // positions per chunk
constexpr uint32_t LZG_GROUP = 64;         // chunks per group
constexpr uint32_t LZG_WIN = 8192;         // entries per window of k_lzg_jump
constexpr uint32_t LZG_LITS = 4096, LZG_LIT_MIN = 32768;
struct LzgJob {
    const uint8_t* src;
    uint8_t* dst;
    uint32_t n, out_len, page, nchunks, ngroups, nwin;
    uint32_t err, left;     // first error (0: none); windows with unresolved entries
    uint32_t queue, slot;   // where the job came from
    uint32_t* eo;           // [n][2]: exit of the chunk from the position, output bytes up to there
    uint32_t* gtab;         // [n][2]: exit of the GROUP from the position, output bytes up to there (k_lzg_groups)
    uint32_t* gent;         // [ngroups][2]: the chain's entry position / output position (LZG_NONE: the chain skips the group)
    uint32_t* cent;         // [nchunks][2]
    uint32_t* ent;          // [out_len] entries
    uint32_t* wdone;        // [nwin]
    uint32_t* lits;         // [LZG_LITS][4]: literal runs of >= LZG_LIT_MIN bytes (source position, output position, length), copied by all workgroups
    uint32_t nlits, pad0;
};
struct LzgArgs {
    LzgJob* jobs;           // LZG_JOBS
    uint32_t* njobs;
    uint8_t* pool;
    uint64_t pool_bytes;
    Status* st;
};


struct DecodeArgs {
    const ColDesc* cols;
    const PageTask* tasks;
    PageDesc* descs;
    TileTask* tiles;
    uint8_t* scratch;
    Status* status;
    InflateJob* jobs_a;  // capacity job_cap_a
    InflateJob* jobs_b;  // capacity job_cap_b
    // queue Z (calls with binary columns): the Zstd payloads nothing in the planning steps waits for — Basic pages of
    // primitives (absolute dst) and the VALUE blocks of Basic binary pages, whose place in the column's values buffer is only
    // known after k_colscan (JOB_REL: dst is an offset from the page's value base).  k_parse fills it, so the block-parallel
    // Zstd pipeline runs its entropy stages for queue A and queue Z in ONE pass, before k_plan; the frames of queue Z are
    // executed after k_colscan.  null: no such queue in this call.
    InflateJob* jobs_z;
    uint32_t* job_counts;  // [0] = queue A, [1] = queue B, [2] tiles, [3] planned pages, [4] page-level RLE, [5] / [6] workgroups of k_parse / k_colscan that are done, [8] / [9] / [11] lengths of A / B / Z when complete, [10] = queue Z, [12] = binary Dict pages whose tile totals k_plan left to k_bin_tile_sums
    uint8_t* zlit;         // Zstd literal buffers, one per inflate wave
    uint64_t* zrec;        // Zstd sequence records, one arena per inflate wave (k_inflate's lane-per-frame pre-decode)
    uint32_t n_pages;
    uint32_t n_cols;
    uint32_t n_tiles;
    FreqEntry* freq_log;   // Freq pages found by k_parse (shared by the calls of one synchronize interval)
    uint32_t* freq_count;
    uint32_t freq_cap;
    uint32_t no_freq;      // second pass: an exceptions block never holds a Freq block (freq.rs:78-79)
    uint32_t sizes_only;   // sb_read_columns_sizes: only what values_len depends on is inflated (nested index blocks)
    uint32_t defer_payloads;  // the call has binary columns (queue B runs): Basic payloads nothing waits for go there too
    uint32_t job_cap_a, job_cap_b;  // entries of the two job queues (2 * n_pages + room for the frames of split Zstd buffers)
    uint32_t lz4_big_min;  // LZ4 blocks of at least this many compressed bytes go to k_inflate_lz4_big (0xFFFFFFFF: the call has no page that long)
    ZbPools zb;            // block-parallel Zstd pipeline (zb.blocks == nullptr: not launched for this call)
    // frame split of LONG multi-frame Zstd buffers (k_zsplit_scan / k_zsplit_chain; null: every buffer is walked by one lane):
    // zs_hdr[0] = entries listed, [1] = segment records handed out, [16 + 2 i], [17 + 2 i] = (queue entry, first record) of
    // listed entry i; zs_segs: records of 8 words (count | 7 positions of the frame magic inside a 16 KiB segment)
    uint32_t* zs_hdr;
    uint32_t* zs_segs;
    uint32_t zs_seg_cap;
    // long RLE pages (a call with few pages of >= 2^18 rows): `rle_parts` workgroups per page, rle_sums[page * parts + part]
    uint32_t rle_parts;
    uint64_t* rle_sums;
    // long bit-packed pages (the same calls): bp_guess[page] = blocks from the first on that share its width (k_bp_guess)
    uint32_t* bp_guess;
    // LZ4 blocks of LZG_MIN compressed bytes and more, block-parallel (sb_lz4_giant.h); lzg.jobs == nullptr: not in this call
    LzgArgs lzg;
    uint32_t lzg_chunks, lzg_wins, lzg_rounds, lzg_jobs;   // grid sizes: the longest page / the largest output of the call / pages long enough
    uint32_t zb_skipped;    // the block-parallel Zstd pipeline was left out on a hint (the context's last intervals read no Zstd buffer): a Zstd buffer of a megabyte or more asks for the replay instead of going frame by frame through the one-wave decoder (20 ms for a 96 MB page against 1.3)
    uint32_t read_skips;    // RSKIP_* bits: kernels left out because the context's last read interval had no work for them (k_plan asks for the replay when this call has)
    uint32_t lzg_skipped;   // the call has pages long enough but the context's last intervals met no such block: the chain is not launched, a block that shows up after all asks for a replay (KIND_REPLAY)
};
constexpr uint32_t LZ4_BIG_MIN = 64u << 10;
constexpr uint32_t RSKIP_QUEUE_A = 1u, RSKIP_TILES = 2u;   // k_zstd_split + k_inflate + k_inflate_lz4 of queue A / k_expand

}  // namespace sb

// ------------------------------------------------------------------ device helpers
#if defined(__HIPCC__)
namespace sb {

// unaligned little-endian loads: page bytes sit at arbitrary byte offsets (9-byte headers,
// def-level sections); gfx950 global loads handle unaligned addresses in hardware.
// All page bytes and Arrow buffers live in HBM: casting to the global address space makes the
// compiler emit global_load/global_store instead of flat_* (pointers read from descriptor tables
// are otherwise "generic", and flat accesses tie up the LDS counter as well).
typedef __attribute__((address_space(1))) const uint8_t* gcptr;
typedef __attribute__((address_space(1))) uint8_t* gptr;
__device__ __forceinline__ uint8_t ldu8(const uint8_t* p) { return *(gcptr)p; }
__device__ __forceinline__ uint16_t ldu16(const uint8_t* p) {
    uint16_t v;
    __builtin_memcpy(&v, (gcptr)p, 2);
    return v;
}
__device__ __forceinline__ uint32_t ldu32(const uint8_t* p) {
    uint32_t v;
    __builtin_memcpy(&v, (gcptr)p, 4);
    return v;
}
__device__ __forceinline__ uint64_t ldu64(const uint8_t* p) {
    uint64_t v;
    __builtin_memcpy(&v, (gcptr)p, 8);
    return v;
}
// aligned u32 / u64 accesses to HBM through generic pointers (workspace arrays handed around as plain pointers): the
// cast makes them global_load / global_store — a flat access also counts on lgkmcnt, so every wait for an LDS result
// would wait for the outstanding HBM accesses too and nothing overlaps
typedef __attribute__((address_space(1))) const uint32_t* gc32p;
typedef __attribute__((address_space(1))) uint32_t* g32p;
typedef __attribute__((address_space(1))) const uint64_t* gc64p;
typedef __attribute__((address_space(1))) uint64_t* g64p;
typedef __attribute__((address_space(3))) uint32_t* l32p;
__device__ __forceinline__ uint32_t gld32(const uint32_t* p) { return *(gc32p)p; }
__device__ __forceinline__ void gst32(uint32_t* p, uint32_t v) { *(g32p)p = v; }
__device__ __forceinline__ uint64_t gld64(const uint64_t* p) { return *(gc64p)p; }
__device__ __forceinline__ void gst64(uint64_t* p, uint64_t v) { *(g64p)p = v; }
// A table of u32 slots that lives in LDS or in HBM (chosen at run time): every access names its address space.
struct SlotTable {
    uint32_t* p;
    bool in_lds;
    __device__ __forceinline__ uint32_t ld(uint32_t h) const {
        return in_lds ? *((l32p)p + h) : __hip_atomic_load((g32p)p + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __device__ __forceinline__ void st(uint32_t h, uint32_t v) const {
        if (in_lds) *((l32p)p + h) = v;
        else *((g32p)p + h) = v;
    }
    // returns the value found (== cmp: the slot now holds val)
    __device__ __forceinline__ uint32_t cas(uint32_t h, uint32_t cmp, uint32_t val) const {
        uint32_t e = cmp;
        if (in_lds)
            __hip_atomic_compare_exchange_strong((l32p)p + h, &e, val, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else
            __hip_atomic_compare_exchange_strong((g32p)p + h, &e, val, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return e;
    }
    __device__ __forceinline__ void amin(uint32_t h, uint32_t v) const {
        if (in_lds) __hip_atomic_fetch_min((l32p)p + h, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else __hip_atomic_fetch_min((g32p)p + h, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
};
__device__ __forceinline__ void stu32(uint8_t* p, uint32_t v) { __builtin_memcpy((gptr)p, &v, 4); }
__device__ __forceinline__ void stu64(uint8_t* p, uint64_t v) { __builtin_memcpy((gptr)p, &v, 8); }

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
struct __attribute__((packed, aligned(1))) U16B {
    u32x4 v;
};
__device__ __forceinline__ u32x4 ldu128(const uint8_t* p) {
    u32x4 v;
    __builtin_memcpy(&v, (gcptr)p, 16);
    return v;
}
__device__ __forceinline__ void stu128(uint8_t* p, u32x4 v) { __builtin_memcpy((gptr)p, &v, 16); }

// value of W bytes as a register bundle
template <int W>
struct Val;
template <>
struct Val<1> {
    uint8_t x;
};
template <>
struct Val<2> {
    uint16_t x;
};
template <>
struct Val<4> {
    uint32_t x;
};
template <>
struct Val<8> {
    uint64_t x;
};
template <>
struct Val<16> {
    u32x4 x;
};
template <>
struct Val<32> {
    u32x4 x, y;
};
template <int W>
__device__ __forceinline__ Val<W> ld_val(const uint8_t* p) {  // unaligned
    Val<W> v;
    __builtin_memcpy(&v, (gcptr)p, W);
    return v;
}
template <int W>
__device__ __forceinline__ void st_val(uint8_t* p, Val<W> v) {  // p aligned to min(W,16)
    __builtin_memcpy((gptr)p, &v, W);
}

__device__ __forceinline__ void raise(Status* st, int32_t code, uint32_t page, uint32_t where) {
    if (atomicCAS(&st->code, 0, code) == 0) {
        st->page = page;
        st->where = where;
    }
}

// padded index into a TILE_ROWS-entry LDS array: one pad word per 16 entries so that a
// thread owning 16 consecutive entries (the scan layout) hits distinct banks
__device__ __forceinline__ int sidx(int i) { return i + (i >> 4); }
constexpr int SIDX_WORDS = TILE_ROWS + TILE_ROWS / 16;

// wave64 inclusive scan (u32 wrapping add)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}
__device__ __forceinline__ uint64_t wave_incl_scan64(uint64_t v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint64_t t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// Inclusive scan of the TILE_ROWS u32 entries of `a` (sidx layout), in place.  Thread t owns
// entries [16t, 16t+16).  `wsum` is a 4-entry LDS scratch.  Returns the tile total.
__device__ __forceinline__ uint32_t tile_incl_scan(uint32_t* a, uint32_t* wsum) {
    const int t = threadIdx.x;
    uint32_t loc[ROWS_PER_THREAD];
    uint32_t run = 0;
#pragma unroll
    for (int j = 0; j < ROWS_PER_THREAD; j++) {
        run += a[sidx(t * ROWS_PER_THREAD + j)];
        loc[j] = run;
    }
    uint32_t incl = wave_incl_scan(run);
    if ((t & 63) == 63) wsum[t >> 6] = incl;
    __syncthreads();
    uint32_t base = incl - run;
    const int w = t >> 6;
    uint32_t w0 = wsum[0], w1 = wsum[1], w2 = wsum[2], w3 = wsum[3];
    if (w > 0) base += w0;
    if (w > 1) base += w1;
    if (w > 2) base += w2;
#pragma unroll
    for (int j = 0; j < ROWS_PER_THREAD; j++) a[sidx(t * ROWS_PER_THREAD + j)] = base + loc[j];
    __syncthreads();
    return w0 + w1 + w2 + w3;
}

// inclusive max-scan over the TILE_ROWS entries of `a` (sidx layout), in place
__device__ __forceinline__ void tile_incl_scan_max(uint32_t* a, uint32_t* wsum) {
    const int t = threadIdx.x;
    uint32_t loc[ROWS_PER_THREAD];
    uint32_t run = 0;
#pragma unroll
    for (int j = 0; j < ROWS_PER_THREAD; j++) {
        run = max(run, a[sidx(t * ROWS_PER_THREAD + j)]);
        loc[j] = run;
    }
    uint32_t incl = run;
    const int lane = t & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t o = __shfl_up(incl, d, 64);
        if (lane >= d) incl = max(incl, o);
    }
    uint32_t excl = __shfl_up(incl, 1, 64);
    if (lane == 0) excl = 0;
    __syncthreads();
    if (lane == 63) wsum[t >> 6] = incl;
    __syncthreads();
    const int w = t >> 6;
    uint32_t base = excl;
    if (w > 0) base = max(base, wsum[0]);
    if (w > 1) base = max(base, wsum[1]);
    if (w > 2) base = max(base, wsum[2]);
#pragma unroll
    for (int j = 0; j < ROWS_PER_THREAD; j++) a[sidx(t * ROWS_PER_THREAD + j)] = max(base, loc[j]);
    __syncthreads();
}

// workgroup sum of one u32 / u64 per thread
__device__ __forceinline__ uint64_t wg_sum64(uint64_t v, uint64_t* wsum4) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) wsum4[threadIdx.x >> 6] = v;
    __syncthreads();
    return wsum4[0] + wsum4[1] + wsum4[2] + wsum4[3];
}

// BitPacker4x geometry: value j of a 128-block lives in lane (j&3) at slot (j>>2);
// slot i of width nb starts at bit i*nb of the lane's stream; word k of lane l is the
// u32 at index 4k+l of the block payload.
__device__ __forceinline__ uint32_t bp4x_extract(const uint8_t* payload, uint32_t nb, int j) {
    if (nb == 0) return 0;
    const int l = j & 3, i = j >> 2;
    const uint32_t bitpos = (uint32_t)i * nb, word = bitpos >> 5, sh = bitpos & 31;
    uint32_t v = ldu32(payload + 4 * (4 * word + l)) >> sh;
    if (sh + nb > 32) v |= ldu32(payload + 4 * (4 * (word + 1) + l)) << (32 - sh);
    if (nb < 32) v &= (1u << nb) - 1;
    return v;
}

}  // namespace sb
#endif
