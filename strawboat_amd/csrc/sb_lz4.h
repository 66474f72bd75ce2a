// strawboat-hip: LZ4 block codec for gfx950, one wave64 per block (reference: src/compression/basic.rs:87-91
// decompress_lz4 -> LZ4_decompress_safe, :108-120 compress_lz4 -> LZ4_compress_default; raw LZ4 *block* format,
// one block per page sub-buffer, no frame, no size prefix).
//
// The format is sequential by nature (a token's position depends on every literal length before it, a match
// may read bytes the previous sequence produced), so both directions keep one wave per block and make the
// per-sequence chain short instead:
//
//  decode   * the compressed bytes are staged in LDS (2 KiB, refilled with 16-byte loads), never read from
//             HBM byte by byte;
//           * token parsing is speculative: lane l parses "a sequence starts at ip + l" for 64 consecutive
//             byte positions at once (token, length extensions, offset), then the real chain of sequence
//             starts is walked with v_readlane / v_writelane (a dozen scalar instructions per sequence, no
//             memory access);
//           * output is assembled in an LDS staging window and flushed with aligned 16-byte stores; literals go
//             LDS -> LDS, matches whose source was flushed earlier ("far") are fetched lane-per-match with ONE
//             store->load wait per batch of up to 128 sequences instead of one per sequence, matches inside the
//             window ("near") are LDS -> LDS in sequence order;
//           * long literal runs / long matches (incompressible pages, runs) go HBM -> HBM with 16-byte copies.
//  encode   * format-valid, NOT liblz4's bytes (the byte-exact greedy parse stays available as
//             lz4_compress_wave behind SB_WRITE_LZ4_EXACT): 64 positions per step, one per lane — 4-byte hash
//             into a 4096-entry u16 table in LDS, candidate check (plus the 8/4/2/1-byte periods columnar data
//             repeats with), 16 bytes of match extension per memory round trip, then a greedy left-to-right
//             selection of non-overlapping matches over the whole window (a scalar walk over the match mask),
//             so one step emits every sequence of the window, not one;
//           * after steps without any match the positions of a step spread out (stride 1, 2, 3, ...), the same
//             idea as LZ4's skip acceleration: incompressible pages cost ~45 steps per 64 KiB, not 1024;
//           * input window and output staging live in LDS; candidates older than the window come from HBM
//             (read-only input: no coherence wait);
//           * the end-of-block rules of the format are kept (last 5 bytes literals, last match starts at least
//             12 bytes before the end) so every LZ4 decoder accepts the block.
#pragma once
#include "sb_common.h"

// phase timing for scripts/micro/lz4_probe.hip (compiled out of the library)
#ifdef SB_LZ4_PROFILE
#define LZP_BEGIN unsigned long long lzp_acc[24] = {0}; unsigned long long lzp_t = __builtin_readcyclecounter();
#define LZP(i) do { const unsigned long long n_ = __builtin_readcyclecounter(); lzp_acc[i] += n_ - lzp_t; lzp_t = n_; } while (0)
#define LZP_CNT(i, v) do { lzp_acc[i] += (v); } while (0)
#define LZP_END do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) for (int i_ = 0; i_ < 24; i_++) g_prof[i_] += lzp_acc[i_]; } while (0)
#define LZP_RESET do { lzp_t = __builtin_readcyclecounter(); } while (0)
#define MTP_START do { pt = __builtin_readcyclecounter(); } while (0)
#define MTP(i) do { const unsigned long long n_ = __builtin_readcyclecounter(); if (prof) prof[i] += n_ - pt; pt = n_; } while (0)
#define MTP_CNT(i, v) do { if (prof) prof[i] += (v); } while (0)
#else
#define LZP_BEGIN
#define LZP(i)
#define LZP_CNT(i, v)
#define LZP_END
#define LZP_RESET
#define MTP_START
#define MTP(i)
#define MTP_CNT(i, v)
#endif

namespace sb {

// compiler + LDS ordering point inside one wave (ds operations of a wave execute in order)
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// this wave's earlier global stores are visible to its later global loads
__device__ __forceinline__ void wave_stores_visible() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ uint32_t rdlane(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }
// lane l of a vector takes the wave-uniform `val` (v_cmp + v_cndmask; this clang has no v_writelane builtin)
__device__ __forceinline__ uint32_t wrlane(uint32_t val, uint32_t l, uint32_t old) {
    return (threadIdx.x & 63) == l ? val : old;
}
// wave64 inclusive add-scan with DPP (row_shr 1/2/4/8 inside each row of 16 lanes, then row_bcast:15 / row_bcast:31
// across rows — the gfx9 recipe): six VALU instructions, no LDS crossbar
__device__ __forceinline__ uint32_t wave_scan_dpp(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return v;
}
__device__ __forceinline__ uint32_t lane_rank(uint64_t m) {  // set bits of m below my lane
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
}
__device__ __forceinline__ uint32_t div255(uint32_t x) { return (uint32_t)(((uint64_t)x * 0x80808081ull) >> 39); }

// n bytes HBM -> HBM by one wave (no overlap): 16-byte stores to aligned addresses, unaligned 16-byte loads,
// four in flight per lane
__device__ __forceinline__ void wave_copy_g2g(uint8_t* dst, const uint8_t* src, uint32_t n) {
    const uint32_t lane = threadIdx.x & 63;
    uint32_t head = (uint32_t)((16 - ((uintptr_t)dst & 15)) & 15);
    if (head > n) head = n;
    if (lane < head) dst[lane] = ldu8(src + lane);
    const uint32_t nvec = (n - head) >> 4;
    uint8_t* d = dst + head;
    const uint8_t* s = src + head;
    uint32_t k = lane;
    for (; k + 192 < nvec; k += 256) {
        const u32x4 a = ldu128(s + 16 * (uint64_t)k), b = ldu128(s + 16 * (uint64_t)(k + 64));
        const u32x4 c = ldu128(s + 16 * (uint64_t)(k + 128)), e = ldu128(s + 16 * (uint64_t)(k + 192));
        stu128(d + 16 * (uint64_t)k, a);
        stu128(d + 16 * (uint64_t)(k + 64), b);
        stu128(d + 16 * (uint64_t)(k + 128), c);
        stu128(d + 16 * (uint64_t)(k + 192), e);
    }
    for (; k < nvec; k += 64) stu128(d + 16 * (uint64_t)k, ldu128(s + 16 * (uint64_t)k));
    const uint32_t done = head + (nvec << 4);
    if (lane < n - done) dst[done + lane] = ldu8(src + done + lane);
}

// 8 bytes at byte index x of a dword array in LDS whose length is a power of two (wm = dwords - 1): three aligned
// dword reads issued together + two v_alignbyte, one LDS round trip instead of eight
__device__ __forceinline__ uint64_t lds_rd8_ring(const uint32_t* a, uint32_t x, uint32_t wm) {
    const uint32_t w = x >> 2;
    const uint32_t d0 = a[w & wm], d1 = a[(w + 1) & wm], d2 = a[(w + 2) & wm];
    return (uint64_t)__builtin_amdgcn_alignbyte(d1, d0, x & 3) | ((uint64_t)__builtin_amdgcn_alignbyte(d2, d1, x & 3) << 32);
}
__device__ __forceinline__ uint64_t lds_rd8(const uint32_t* a, uint32_t x) {  // no wrap; a[] has 8 bytes of slack
    const uint32_t w = x >> 2;
    const uint32_t d0 = a[w], d1 = a[w + 1], d2 = a[w + 2];
    return (uint64_t)__builtin_amdgcn_alignbyte(d1, d0, x & 3) | ((uint64_t)__builtin_amdgcn_alignbyte(d2, d1, x & 3) << 32);
}
// the low nb (<= 8) bytes of v to ring bytes [x, x + nb)
__device__ __forceinline__ void lds_wr_bytes(uint8_t* ring, uint32_t rm, uint32_t x, uint64_t v, uint32_t nb) {
#pragma unroll
    for (uint32_t b = 0; b < 8; b++)
        if (b < nb) ring[(x + b) & rm] = (uint8_t)(v >> (8 * b));
}

// ------------------------------------------------------------------------------------------------ decode
constexpr uint32_t LZD_IB = 2048;        // compressed bytes staged in LDS
constexpr uint32_t LZD_SHORT_LIT = 300;  // literals up to here are copied from the staged input
constexpr uint32_t LZD_MARGIN = 64 + 3 + LZD_SHORT_LIT + 2 + 3 + 16;  // window + extensions + literal + offset + extensions
// (a lone wave is latency bound — ~8 cycles per dependent instruction — so throughput comes from waves per SIMD: the
// LDS footprint is kept near 12 KiB per wave, 12 waves per CU)
constexpr uint32_t LZD_RING = 8192;      // output ring (bytes): the current batch and at least 4 KiB of history
constexpr uint32_t LZD_BATCH = 4096;     // output bytes per batch
constexpr uint32_t LZD_MB = 320;         // matches per batch
struct Lz4DecLds {
    __attribute__((aligned(16))) uint8_t ib[LZD_IB + 32];
    __attribute__((aligned(16))) uint8_t ring[LZD_RING];
    uint32_t m_dst[LZD_MB];  // match destination (output position)
    uint32_t m_len[LZD_MB];
    uint16_t m_off[LZD_MB];
};

// sum of a length extension (bytes 255 ... 255 x) starting at src[x], 64 bytes per step; returns the position behind it
__device__ __forceinline__ uint32_t lz4_ext_sum(const uint8_t* src, uint32_t x, uint32_t n, uint32_t* len) {
    const uint32_t lane = threadIdx.x & 63;
    for (;;) {
        if (x >= n) return n + 1;  // ran off the block
        const uint32_t b = x + lane < n ? (uint32_t)ldu8(src + x + lane) : 0u;
        const uint64_t nz = __ballot(b != 255);
        const uint32_t take = nz ? (uint32_t)__builtin_ctzll(nz) : 64u;
        *len += 255 * take;
        if (nz) {
            *len += rdlane(b, take);
            return x + take + 1;
        }
        x += 64;
    }
}

// Decode one LZ4 block (executed by ONE wave64).  Returns 0, or a non-zero tag on a malformed stream
// (LZ4_decompress_safe < 0 upstream => Error::External).
//
// Positions: `op` counts output bytes; g = op + a0 with a0 = dst & 15, so that g % 16 is the byte's place in its
// 16-byte line in HBM.  Output byte g lives at ring[g % LZD_RING] until it is overwritten LZD_RING bytes later.
// fl = g-position up to which the ring has been written to HBM (whole 16-byte groups).
__device__ uint32_t lz4_inflate_block(const uint8_t* src, uint32_t n, uint8_t* dst, uint32_t out_len, Lz4DecLds& L) {
    const uint32_t lane = threadIdx.x & 63;
    if (n == 0) return out_len != 0 ? 100u : 0u;
    const uint32_t a0 = (uint32_t)((uintptr_t)dst & 15);
    uint8_t* gbase = dst - a0;
    constexpr uint32_t RM = LZD_RING - 1;
    uint32_t op = 0;          // output position (bytes produced so far, incl. the ring-only part)
    uint32_t fl = 0;          // g-positions below fl are in HBM
    uint32_t ring_lo = 0;     // g-positions below ring_lo are NOT in the ring (after a straight HBM -> HBM sequence)
    uint32_t bstart = 0;      // op at the start of the current batch
    uint32_t ip = 0;          // input position
    uint32_t ibase = 0;       // ib holds src[ibase, ibase + ibn), ibase a multiple of 16
    uint32_t ibn = 0;
    uint32_t nm = 0;          // matches recorded in the current batch
    uint32_t err = 0;
    bool done = false;
    LZP_BEGIN

    auto refill = [&](uint32_t from) {
        ibase = from & ~15u;
        ibn = min(LZD_IB, (n - ibase + 15) & ~15u);
        wave_sync();
        for (uint32_t k = lane * 16; k < ibn; k += 64 * 16) {
            if (ibase + k + 16 <= n) {
                *(u32x4*)(L.ib + k) = ldu128(src + ibase + k);
            } else {  // the last group may reach past the block: byte by byte
                for (uint32_t b = 0; b < 16; b++) L.ib[k + b] = ibase + k + b < n ? ldu8(src + ibase + k + b) : (uint8_t)0;
            }
        }
        wave_sync();
    };
    // ring -> HBM: g-positions [fl, g_end).  Whole 16-byte groups with aligned stores; `all` also writes the partial
    // last group (byte by byte; a later flush rewrites it from the ring).  Bytes below a0 are never written.
    auto flush = [&](uint32_t g_end, bool all) {
        wave_sync();
        const uint32_t full_end = g_end & ~15u;
        uint32_t g = fl;
        if (g < full_end && g < a0) {  // (fl == 0) the first group starts in front of dst: its bytes one by one
            if (lane >= a0 && lane < 16) gbase[lane] = L.ring[lane];
            g = 16;
        }
        const uint32_t ng = full_end > g ? (full_end - g) >> 4 : 0;
        for (uint32_t k = lane; k < ng; k += 64) {
            const uint32_t gg = g + 16 * k;
            stu128(gbase + gg, *(const u32x4*)(L.ring + (gg & RM)));
        }
        if (all) {  // the partial last group
            const uint32_t x = full_end + lane;
            if (x < g_end && x >= a0) gbase[x] = L.ring[x & RM];
        }
        if (full_end > fl) fl = full_end;
        wave_sync();
    };
    // the matches recorded in L.m_* (in sequence order), then flush
    auto run_batch = [&](bool all) {
        const uint32_t g_op = op + a0;
        LZP(4);
        LZP_CNT(18, 1);
        if (nm) {
            // g-positions >= ring_min are in the ring now and stay there until the batch is complete
            const uint32_t ring_min = max(ring_lo, g_op > LZD_RING ? g_op - LZD_RING : 0u);
            bool waited = false;
            for (uint32_t k0 = 0; k0 < nm; k0 += 64) {
                const uint32_t k = k0 + lane;
                const bool have = k < nm;
                uint32_t d = 0, ml = 0, off = 1;
                if (have) { d = L.m_dst[k] + a0; ml = L.m_len[k]; off = L.m_off[k]; }
                const uint32_t s = d - off;                    // g-position of the source
                const uint32_t send = s + min(ml, off);        // every source byte lies in [s, send)
                const bool in_ring = have && s >= ring_min;
                const bool far = have && !in_ring && send <= fl;   // all source bytes are in HBM
                const bool mixed = have && !in_ring && !far;       // straddles: serial path below
                // ---- far matches: HBM -> ring, lane per match (one store->load wait per batch)
                if (__ballot(far || mixed) && !waited) {
                    wave_stores_visible();
                    waited = true;
                }
                if (far) {
                    if (off >= 16 && s + ((ml + 15) & ~15u) <= fl) {
                        for (uint32_t i = 0; i < ml; i += 16) {
                            const u32x4 v = ldu128(gbase + s + i);
                            uint64_t lo = (uint64_t)v.x | ((uint64_t)v.y << 32), hi = (uint64_t)v.z | ((uint64_t)v.w << 32);
                            const uint32_t nb = min(16u, ml - i);
                            for (uint32_t b = 0; b < nb; b++) {
                                L.ring[(d + i + b) & RM] = (uint8_t)lo;
                                lo = (lo >> 8) | (hi << 56);
                                hi >>= 8;
                            }
                        }
                    } else {
                        for (uint32_t i = 0, j = 0; i < ml; i++) {
                            L.ring[(d + i) & RM] = ldu8(gbase + s + j);
                            if (++j == off) j = 0;
                        }
                    }
                }
                wave_sync();
                LZP(5);
                LZP_CNT(20, __popcll(__ballot(far)));
                LZP_CNT(21, __popcll(__ballot(in_ring)));
                // ---- ring matches of this chunk in sequence order, the wave copies one match at a time: lane i moves byte i
                // (a lone wave pays ~8 cycles per dependent instruction and ~20 per taken branch, so the loop body is
                // kept to three v_readlane, one LDS read and one LDS write; matches of up to 64 bytes take one pass)
                uint64_t rem = __ballot(in_ring || mixed);
                const uint64_t mixed_m = __ballot(mixed);
                while (rem) {
                    const uint32_t first = (uint32_t)__builtin_ctzll(rem);
                    rem &= rem - 1;
                    const uint32_t fd = rdlane(d, first), fml = rdlane(ml, first), foff = rdlane(off, first);
                    const uint32_t fs = fd - foff;
                    if (__builtin_expect(fml <= 64 && foff >= fml && !((mixed_m >> first) & 1), 1)) {   // the common shape: one pass
                        if (lane < fml) L.ring[(fd + lane) & RM] = L.ring[(fs + lane) & RM];
                        LZP_CNT(19, 1);
                        continue;
                    }
                    if (__builtin_expect((mixed_m >> first) & 1, 0)) {  // byte source chosen per byte
                        const bool periodic = foff < 64 && foff < fml;
                        for (uint32_t i0 = 0; i0 < fml; i0 += 64) {
                            const uint32_t i = i0 + lane;
                            if (i < fml) {
                                const uint32_t sp = fs + (periodic ? i % foff : i);
                                const uint8_t v = sp < fl ? ldu8(gbase + sp) : L.ring[sp & RM];
                                L.ring[(fd + i) & RM] = v;
                            }
                            wave_sync();
                        }
                        continue;
                    }
                    if (foff >= 64 || foff >= fml) {          // no lane reads a byte of this pass's writes
                        for (uint32_t i0 = 0; i0 < fml; i0 += 64) {
                            const uint32_t i = i0 + lane;
                            if (i < fml) L.ring[(fd + i) & RM] = L.ring[(fs + i) & RM];
                        }
                    } else {                                  // short period: every byte comes from [fs, fs + foff)
                        const uint32_t j0 = lane % foff, step = 64 % foff;
                        uint32_t j = j0;
                        for (uint32_t i0 = 0; i0 < fml; i0 += 64) {
                            const uint32_t i = i0 + lane;
                            if (i < fml) L.ring[(fd + i) & RM] = L.ring[(fs + j) & RM];
                            j += step;
                            if (j >= foff) j -= foff;
                        }
                    }
                    LZP_CNT(19, 1);
                }
                wave_sync();
                LZP(6);
            }
            nm = 0;
        }
        flush(g_op, all);
        bstart = op;
        LZP(7);
    };

    refill(0);
    while (!done && !err) {
        LZP(4);
        if (ip + LZD_MARGIN > ibase + ibn && ibase + ibn < n) refill(ip);
        LZP(0);
        LZP_CNT(16, 1);
        // ---- speculative parse: lane l assumes a sequence starts at ip + l
        const uint32_t p = ip + lane;
        uint32_t lit = 0, ml = 0, off = 0, nxt = 0, flags = 0;  // flags: 1 = big (serial path), 2 = last sequence, 4 = invalid
        uint32_t q = p + 1;
        if (p < n) {
            const uint32_t tok = L.ib[p - ibase];
            lit = tok >> 4;
            if (lit == 15) {
                uint32_t b = 255, cnt = 0;
                while (b == 255 && cnt < 2 && q < n) { b = L.ib[q - ibase]; q++; lit += b; cnt++; }
                if (b == 255) flags |= 1;
            }
            if (lit > LZD_SHORT_LIT) flags |= 1;
            if (!(flags & 1)) {
                const uint32_t moff = q + lit;
                if (moff >= n) {
                    flags |= (moff == n) ? 2u : 4u;   // the last sequence ends exactly at the block end
                    nxt = n;
                } else if (moff + 2 > n) {
                    flags |= 4;
                } else {
                    off = (uint32_t)L.ib[moff - ibase] | ((uint32_t)L.ib[moff + 1 - ibase] << 8);
                    uint32_t r = moff + 2;
                    ml = tok & 15;
                    if (ml == 15) {
                        uint32_t b = 255, cnt = 0;
                        while (b == 255 && cnt < 2 && r < n) { b = L.ib[r - ibase]; r++; ml += b; cnt++; }
                        if (b == 255) flags |= 1;
                    }
                    ml += 4;
                    nxt = r;
                }
            }
        } else {
            flags = 4;
        }
        // ---- walk the chain of real sequence starts (wave-uniform): accepted lanes get their output position and,
        // for matches, their slot in the batch
        LZP(1);
        const uint32_t tot = lit + ml;                   // output bytes of my sequence
        // (the loop is the serial part of the parse: one v_readlane, the mask update and ONE taken branch per sequence;
        // everything that can be computed per lane afterwards is)
        const uint32_t comb = (flags & 5) ? 0x1000u | flags : ((flags & 2) ? 0x2000u : nxt - ip);   // (nxt - ip < 0x1000)
        uint64_t M = 0;
        uint32_t cur = 0, code = 0;
        do {
            const uint32_t v = rdlane(comb, cur);
            if (v >= 0x1000) { code = v; break; }
            M |= 1ull << cur;
            cur = v;
        } while (cur < 64);
        uint32_t stop = 0;                               // 1 = batch full, 2 = big, 3 = last, 4 = bad
        if (code & 0x2000) { stop = 3; M |= 1ull << cur; }
        else if (code & 4) stop = 4;
        else if (code & 1) stop = 2;
        // output positions and batch slots of the accepted lanes; cut the window where the batch is full
        const uint32_t room = LZD_BATCH - (op - bstart);   // output bytes the batch still takes
        bool mine = (M >> lane) & 1;
        uint32_t incl = wave_scan_dpp(mine ? tot : 0u);
        uint32_t rank = lane_rank(M);
        const uint64_t over = __ballot(mine && (incl > room || nm + rank >= LZD_MB));
        if (over) {
            const uint32_t c = (uint32_t)__builtin_ctzll(over);
            M &= (1ull << c) - 1;
            mine = (M >> lane) & 1;
            stop = 1;
            cur = c;
        }
        const uint32_t my_op = op + incl - tot, my_k = nm + rank;
        const uint32_t cnt_all = (uint32_t)__popcll(M);
        const uint32_t cnt_m = cnt_all - ((stop == 3) ? 1u : 0u);
        const uint32_t acc = cnt_all ? rdlane(incl, 63u - (uint32_t)__builtin_clzll(M)) : 0u;
        LZP(2);
        LZP_CNT(17, cnt_m);
        bool bad = false;
        if (mine) {
            if (my_op + tot > out_len) bad = true;
            if (!(flags & 2) && (off == 0 || off > my_op + lit)) bad = true;
        }
        if (__ballot(bad)) { err = 103; break; }
        {   // literals: staged input -> ring; per lane up to 32 bytes, longer ones wave-wide
            const uint32_t wpos = my_op + a0, rpos = q - ibase;
            const uint32_t nl = mine ? min(lit, 32u) : 0u;
            for (uint32_t i = 0; i < nl; i += 8)
                lds_wr_bytes(L.ring, RM, wpos + i, lds_rd8((const uint32_t*)L.ib, rpos + i), nl - i);
            uint64_t lm = __ballot(mine && lit > 32);
            while (lm) {
                const uint32_t l = (uint32_t)__builtin_ctzll(lm);
                lm &= lm - 1;
                const uint32_t w2 = rdlane(wpos, l), r2 = rdlane(rpos, l), n2 = rdlane(lit, l);
                for (uint32_t i = 32 + lane; i < n2; i += 64) L.ring[(w2 + i) & RM] = L.ib[r2 + i];
            }
        }
        LZP(3);
        if (mine && !(flags & 2)) {
            L.m_dst[my_k] = my_op + lit;
            L.m_len[my_k] = ml;
            L.m_off[my_k] = (uint16_t)off;
        }
        nm += cnt_m;
        op += acc;
        wave_sync();
        if (stop == 0) {                      // left the window through an ordinary sequence
            ip += cur;
            continue;
        }
        if (stop == 4) { err = 101; break; }
        if (stop == 3) {                      // the last sequence was consumed
            ip = n;
            done = true;
            break;
        }
        ip += cur;                            // position of the sequence that did not fit / is big
        run_batch(stop == 2);
        if (stop != 2) continue;
        // ---- one big sequence, wave-serial: long literal run and / or long match, straight HBM -> HBM
        // (run_batch(true) has written every byte produced so far)
        uint32_t x = ip;
        const uint32_t t = ldu8(src + x);
        x++;
        uint32_t bl = t >> 4;
        if (bl == 15) x = lz4_ext_sum(src, x, n, &bl);
        if (x > n || bl > n - x || bl > out_len - op) { err = 103; break; }
        wave_copy_g2g(dst + op, src + x, bl);
        x += bl;
        op += bl;
        if (x == n) {
            done = true;                       // it was the last sequence
        } else {
            if (n - x < 2) { err = 104; break; }
            const uint32_t boff = (uint32_t)ldu8(src + x) | ((uint32_t)ldu8(src + x + 1) << 8);
            x += 2;
            uint32_t bm = t & 15;
            if (bm == 15) x = lz4_ext_sum(src, x, n, &bm);
            bm += 4;
            if (x > n || boff == 0 || boff > op || bm > out_len - op) { err = 106; break; }
            wave_stores_visible();
            if (boff >= bm) {
                wave_copy_g2g(dst + op, dst + op - boff, bm);
            } else if (boff >= 1024) {         // overlapping, long period: chunks of `boff` bytes, each complete before the next
                for (uint32_t c0 = 0; c0 < bm; c0 += boff) {
                    wave_copy_g2g(dst + op + c0, dst + op - boff + c0, min(boff, bm - c0));
                    wave_stores_visible();
                }
            } else {                           // short period: every byte comes from [op - boff, op)
                const uint8_t* hist = dst + op - boff;
                for (uint32_t i = lane; i < bm; i += 64) dst[op + i] = ldu8(hist + i % boff);
            }
            op += bm;
        }
        // restart the ring at the new output position: nothing below is in the ring any more; the partial 16-byte
        // group below op is read back so that the next aligned flush rewrites the same bytes
        wave_stores_visible();
        fl = (op + a0) & ~15u;
        ring_lo = fl;
        bstart = op;
        {
            const uint32_t keep = (op + a0) - fl;
            if (lane < keep && fl + lane >= a0) L.ring[(fl + lane) & RM] = ldu8(gbase + fl + lane);
        }
        wave_sync();
        ip = x;
        LZP(8);
    }
    if (!err && done) run_batch(true);
    LZP_END;
    if (err) return err;
    if (!done) return 101;
    if (op != out_len) return 107;
    return 0;
}

// ------------------------------------------------------------------------------------------------ sequence executor
// Executes decoded LZ sequences (literal length, match length, distance) whose literals sit in one contiguous HBM buffer
// — the Zstd decoder's sequence stage (sb_zstd.h) — through the same LDS output ring as lz4_inflate_block: a batch of up
// to 64 sequences costs one HBM round trip for its literals and at most one store->load wait for matches that reach
// behind the ring, instead of two waits per sequence.
constexpr uint32_t LZX_RING = 8192, LZX_BATCH = 4096;
#ifdef ZB_TL
__device__ unsigned long long g_lzx_t[8];
#define LZXT_BEGIN unsigned long long lzxt = __builtin_readcyclecounter();
#define LZXT(i) do { const unsigned long long n_ = __builtin_readcyclecounter(); if (blockIdx.x < 2 && (threadIdx.x & 63) == 0) atomicAdd(&g_lzx_t[i], n_ - lzxt); lzxt = n_; } while (0)
#define LZXT_CNT(i, v) do { if (blockIdx.x < 2 && (threadIdx.x & 63) == 0) atomicAdd(&g_lzx_t[i], (unsigned long long)(v)); } while (0)
#else
#define LZXT_BEGIN
#define LZXT(i)
#define LZXT_CNT(i, v)
#endif
struct LzSeqLds {
    __attribute__((aligned(16))) uint8_t ring[LZX_RING];
    uint32_t ls[65], lo[64];   // HBM-mode batches: where a sequence's literal run starts in the batch's literal stream / in the output
};
struct LzSeqExec {
    LzSeqLds& L;
    uint8_t* dst;
    uint32_t a0;
    uint8_t* gbase;
    uint32_t fl = 0, ring_lo = 0;
    static constexpr uint32_t RM = LZX_RING - 1;
    __device__ LzSeqExec(LzSeqLds& l, uint8_t* d) : L(l), dst(d), a0((uint32_t)((uintptr_t)d & 15)), gbase(d - ((uintptr_t)d & 15)) {}
    __device__ void flush(uint32_t g_end, bool all) {
        const uint32_t lane = threadIdx.x & 63;
        wave_sync();
        const uint32_t full_end = g_end & ~15u;
        uint32_t g = fl;
        if (g < full_end && g < a0) {
            if (lane >= a0 && lane < 16) gbase[lane] = L.ring[lane];
            g = 16;
        }
        const uint32_t ng = full_end > g ? (full_end - g) >> 4 : 0;
        for (uint32_t k = lane; k < ng; k += 64) {
            const uint32_t gg = g + 16 * k;
            stu128(gbase + gg, *(const u32x4*)(L.ring + (gg & RM)));
        }
        if (all) {
            const uint32_t x = full_end + lane;
            if (x < g_end && x >= a0) gbase[x] = L.ring[x & RM];
        }
        if (full_end > fl) fl = full_end;
        wave_sync();
    }
    // after bytes were written straight to HBM up to output position op: nothing below is in the ring any more
    __device__ void restart(uint32_t op) {
        const uint32_t lane = threadIdx.x & 63;
        wave_stores_visible();
        fl = (op + a0) & ~15u;
        ring_lo = fl;
        const uint32_t keep = (op + a0) - fl;
        if (lane < keep && fl + lane >= a0) L.ring[(fl + lane) & RM] = ldu8(gbase + fl + lane);
        wave_sync();
    }
    // Which matches of the batch (bit j = sequence j) write bytes that lane's match reads: the destinations [mdst, mdst + ml)
    // are sorted and disjoint, so they are the sequences from the first one that ends behind my source's start to the last
    // one that starts in front of my source's end — two binary searches over the published starts / ends, once per batch.
    // A match may go as soon as none of these is pending; everything else it reads (literals, earlier batches) is final.
    __device__ uint64_t dep_mask(uint32_t mdst, uint32_t ml, uint32_t msrc, uint32_t msend) {
        const uint32_t lane = threadIdx.x & 63;
        wave_sync();
        L.lo[lane] = mdst;
        L.ls[lane] = mdst + ml;
        wave_sync();
        uint32_t lo_ = 0, hi_ = 64;     // ja = the first j whose end is > msrc: count of j with end <= msrc
        uint32_t lo2 = 0, hi2 = 64;     // jb + 1 = count of j whose start is < msend
#pragma unroll
        for (int st = 0; st < 7; st++) {
            if (lo_ < hi_) {
                const uint32_t mid = (lo_ + hi_) >> 1;
                if (L.ls[mid] <= msrc) lo_ = mid + 1;
                else hi_ = mid;
            }
            if (lo2 < hi2) {
                const uint32_t mid = (lo2 + hi2) >> 1;
                if (L.lo[mid] < msend) lo2 = mid + 1;
                else hi2 = mid;
            }
        }
        const uint32_t ja = lo_, je = lo2;   // sequences [ja, je)
        if (je <= ja) return 0ull;
        const uint64_t upto = je >= 64 ? ~0ull : ((1ull << je) - 1);
        return upto & ~((1ull << ja) - 1);
    }
    // One batch: lane k < nb holds sequence k (ll, ml, off; off validated by the caller: 0 < off <= its match position).
    // lit = the batch's literals, contiguous.  op = output position of the batch's first byte.  Returns the bytes produced.
    // has_pre: pre8 holds the first min(ll, 8) literal bytes of the lane's sequence (requested by the caller a batch ahead)
    __device__ uint32_t run(uint32_t nb, uint32_t ll, uint32_t ml, uint32_t off, const uint8_t* lit, uint32_t op, bool has_pre = false,
                            uint64_t pre8 = 0) {
        const uint32_t lane = threadIdx.x & 63;
        const bool have = lane < nb;
        const uint32_t tot = have ? ll + ml : 0u;
        const uint32_t incl = wave_scan_dpp(tot);
        const uint32_t total = rdlane(incl, 63);
        const uint32_t lincl = wave_scan_dpp(have ? ll : 0u);
        const uint32_t my_op = op + incl - tot;             // my literals start here, my match at my_op + ll
        const uint32_t my_lit = lincl - (have ? ll : 0u);  // offset of my literals in `lit`
        const uint64_t bigm = __ballot(have && (ll > 128 || ml > 512));
        LZXT_BEGIN
        if (total > LZX_BATCH || bigm) {
            // ---- a batch with long literal runs / long matches goes through HBM, in two phases.  Literals depend on
            // nothing: every run of the batch is copied to its place first, wave-wide, no waits in between.  Then the
            // matches: short ones that do not overlap themselves lane per match, every one whose source ends at or in front
            // of the first pending match's destination at once (the bytes there are final), one store -> load wait per
            // round; a long or self-overlapping match wave-wide when it is the first pending one.
            flush(op + a0, true);
            LZXT(1);
            // The batch's literal runs are ONE contiguous stretch of `lit`.  Lane per run: its first and its last 16 bytes (runs
            // of less than 16 bytes: 8 + 4 + 2 + 1).  Then the 16-byte chunks of the stretch that lie INSIDE one run: chunk c
            // goes to lane c % 64, eight chunks per lane in flight (1 KiB per wave and load, coalesced); a chunk finds its run
            // by a binary search over the runs' start offsets (LDS).  (Lane per run alone waited for HBM once per 64 bytes of
            // the LONGEST run of the batch: 40 us per batch on C5's Int64 leaves.)
            {
                const uint32_t T = rdlane(lincl, 63);
                L.ls[lane] = have ? my_lit : T;
                L.lo[lane] = my_op;
                if (lane == 0) L.ls[64] = T;
                wave_sync();
                if (have && ll) {
                    const uint8_t* sp = lit + my_lit;
                    uint8_t* dp = dst + my_op;
                    if (ll >= 16) {
                        const u32x4 h = ldu128(sp), t = ldu128(sp + ll - 16);
                        stu128(dp, h);
                        stu128(dp + ll - 16, t);
                    } else {
                        uint32_t o = 0;
                        uint64_t v8 = 0;
                        uint32_t v4 = 0, v2 = 0, v1 = 0;
                        if (ll & 8) v8 = ldu64(sp);
                        if (ll & 4) v4 = ldu32(sp + (ll & 8));
                        if (ll & 2) v2 = ldu16(sp + (ll & 12));
                        if (ll & 1) v1 = ldu8(sp + (ll & 14));
                        if (ll & 8) { stu64(dp, v8); o = 8; }
                        if (ll & 4) { stu32(dp + o, v4); o += 4; }
                        if (ll & 2) { const uint16_t w = (uint16_t)v2; __builtin_memcpy((gptr)(dp + o), &w, 2); o += 2; }
                        if (ll & 1) *(gptr)(dp + o) = (uint8_t)v1;
                    }
                }
                const uint32_t nchunks = T >> 4;   // (full chunks only: the last bytes of the stretch are some run's last bytes)
                for (uint32_t c0 = 0; c0 < nchunks; c0 += 64 * 8) {
                    uint32_t kk[8];
                    u32x4 v[8];
#pragma unroll
                    for (uint32_t j = 0; j < 8; j++) {
                        const uint32_t x = 16 * (c0 + 64 * j + lane);
                        uint32_t lo_ = 0, hi_ = 64;   // the last run that starts at or in front of x (runs of length 0 never win)
#pragma unroll
                        for (int st = 0; st < 6; st++) {
                            const uint32_t mid = (lo_ + hi_) >> 1;
                            if (L.ls[mid] <= x) lo_ = mid;
                            else hi_ = mid;
                        }
                        kk[j] = lo_;
                    }
                    bool in_[8];
#pragma unroll
                    for (uint32_t j = 0; j < 8; j++) {
                        const uint32_t x = 16 * (c0 + 64 * j + lane);
                        in_[j] = c0 + 64 * j + lane < nchunks && x + 16 <= L.ls[kk[j] + 1];   // (chunks over a run border: the runs' own moves)
                        if (in_[j]) v[j] = ldu128(lit + x);
                    }
#pragma unroll
                    for (uint32_t j = 0; j < 8; j++) {
                        const uint32_t x = 16 * (c0 + 64 * j + lane);
                        if (in_[j]) stu128(dst + L.lo[kk[j]] + (x - L.ls[kk[j]]), v[j]);
                    }
                }
            }
            LZXT(2);
            wave_stores_visible();
            LZXT(3);
            const uint32_t md = my_op + ll, ms = md - off;          // (off <= md: checked by the caller)
            const bool ism_ = have && ml > 0;
            const bool smp = ism_ && ml <= 32 && off >= ml;
            const uint32_t msend = ms + min(ml, off);
            uint64_t rem_ = __ballot(ism_);
            const uint64_t smp_m = __ballot(smp);
            const uint64_t deps_ = dep_mask(have ? md : 0xFFFFFFFFu, have ? ml : 0u, ms, msend);
            while (rem_) {
                const uint32_t first = (uint32_t)__builtin_ctzll(rem_);
                if ((smp_m >> first) & 1) {
                    const bool go = ((rem_ >> lane) & 1) && smp && !(deps_ & rem_);
                    if (go) {
                        uint64_t v[4] = {0, 0, 0, 0};
#pragma unroll
                        for (uint32_t j = 0; j < 4; j++) {
                            if (8 * j + 8 <= ml) {
                                v[j] = ldu64(dst + ms + 8 * j);
                            } else if (8 * j < ml) {
                                for (uint32_t q = 0; 8 * j + q < ml; q++) v[j] |= (uint64_t)ldu8(dst + ms + 8 * j + q) << (8 * q);
                            }
                        }
#pragma unroll
                        for (uint32_t j = 0; j < 4; j++) {
                            if (8 * j + 8 <= ml) {
                                stu64(dst + md + 8 * j, v[j]);
                            } else if (8 * j < ml) {
                                for (uint32_t q = 0; 8 * j + q < ml; q++) *(gptr)(dst + md + 8 * j + q) = (uint8_t)(v[j] >> (8 * q));
                            }
                        }
                    }
                    rem_ &= ~__ballot(go);
                    wave_stores_visible();
                    LZXT_CNT(5, 1);
                    continue;
                }
                rem_ &= rem_ - 1;
                const uint32_t km = rdlane(ml, first), ko = rdlane(off, first), o = rdlane(md, first);
                if (ko >= km) {
                    wave_copy_g2g(dst + o, dst + o - ko, km);
                } else if (ko >= 1024) {
                    for (uint32_t c0 = 0; c0 < km; c0 += ko) {
                        wave_copy_g2g(dst + o + c0, dst + o - ko + c0, min(ko, km - c0));
                        wave_stores_visible();
                    }
                } else {
                    const uint8_t* hist = dst + o - ko;
                    for (uint32_t i = lane; i < km; i += 64) dst[o + i] = ldu8(hist + i % ko);
                }
                wave_stores_visible();
            }
            LZXT(4);
            restart(op + total);
            LZXT(0);
            LZXT_CNT(6, 1);
            return total;
        }
        const uint32_t g_end = op + total + a0;
        // ---- literals: HBM -> ring, lane per sequence, 8 bytes per round trip
        if (have) {   // (four loads in flight per wait)
            for (uint32_t i0 = 0; i0 < ll; i0 += 32) {
                uint64_t v[4] = {0, 0, 0, 0};
#pragma unroll
                for (uint32_t j = 0; j < 4; j++) {
                    const uint32_t i = i0 + 8 * j;
                    if (i >= ll) continue;
                    const uint32_t nbytes = min(8u, ll - i);
                    if (i == 0 && has_pre) {
                        v[j] = pre8;
                    } else if (nbytes == 8) {
                        v[j] = ldu64(lit + my_lit + i);
                    } else {
                        for (uint32_t b = 0; b < nbytes; b++) v[j] |= (uint64_t)ldu8(lit + my_lit + i + b) << (8 * b);
                    }
                }
#pragma unroll
                for (uint32_t j = 0; j < 4; j++) {
                    const uint32_t i = i0 + 8 * j;
                    if (i < ll) lds_wr_bytes(L.ring, RM, my_op + a0 + i, v[j], min(8u, ll - i));
                }
            }
        }
        wave_sync();
        LZXT(1);
        // ---- matches
        const uint32_t ring_min = max(ring_lo, g_end > LZX_RING ? g_end - LZX_RING : 0u);
        const uint32_t d = my_op + ll + a0, s = d - off;
        const uint32_t send = s + min(ml, off);
        const bool ism = have && ml > 0;
        const bool in_ring = ism && s >= ring_min;
        const bool far = ism && !in_ring && send <= fl;
        const bool mixed = ism && !in_ring && !far;
        if (__ballot(far || mixed)) wave_stores_visible();
        if (far) {
            if (ml <= 16 && off >= ml) {   // (8 bytes per load where that stays inside what has been flushed)
                uint64_t v0 = 0, v1 = 0;
                if (s + 8 <= fl) {
                    v0 = ldu64(gbase + s);
                } else {
                    for (uint32_t q = 0; q < ml && q < 8; q++) v0 |= (uint64_t)ldu8(gbase + s + q) << (8 * q);
                }
                if (ml > 8) {
                    if (s + 16 <= fl) {
                        v1 = ldu64(gbase + s + 8);
                    } else {
                        for (uint32_t q = 8; q < ml; q++) v1 |= (uint64_t)ldu8(gbase + s + q) << (8 * (q - 8));
                    }
                }
                lds_wr_bytes(L.ring, RM, d, v0, min(ml, 8u));
                if (ml > 8) lds_wr_bytes(L.ring, RM, d + 8, v1, ml - 8);
            } else {
                for (uint32_t i = 0, j = 0; i < ml; i++) {
                    L.ring[(d + i) & RM] = ldu8(gbase + s + j);
                    if (++j == off) j = 0;
                }
            }
        }
        wave_sync();
        uint64_t rem = __ballot(in_ring || mixed);
        const uint64_t mixed_m = __ballot(mixed);
        LZXT(2);
        // short matches that do not overlap themselves are copied LANE PER MATCH, many at a time: everything in front of
        // the first pending match's destination is final (literals and far matches are in place, earlier matches done),
        // so every pending short match whose source ends there or earlier may go at once — 8 + 8 source bytes read, then
        // written.  Text whose matches reach back further than a batch's ~300 bytes of output: one or two rounds per batch
        // instead of one LDS round trip per match
        const bool simple = in_ring && ml <= 16 && off >= ml;
        const uint64_t simple_m = __ballot(simple);
        const uint64_t deps = simple_m ? dep_mask(have ? d : 0xFFFFFFFFu, have ? ml : 0u, s, send) : 0ull;
        while (rem) {
            const uint32_t first = (uint32_t)__builtin_ctzll(rem);
            if ((simple_m >> first) & 1) {
                const bool go = ((rem >> lane) & 1) && simple && !(deps & rem);
                const bool long_ = __ballot(go && ml > 8) != 0;   // (uniform: most rounds move matches of <= 8 bytes only)
                if (go) {
                    const uint32_t* r32 = (const uint32_t*)L.ring;
                    const uint64_t v0 = lds_rd8_ring(r32, s & RM, RM >> 2);
                    uint64_t v1 = 0;
                    if (long_) v1 = lds_rd8_ring(r32, (s + 8) & RM, RM >> 2);
                    lds_wr_bytes(L.ring, RM, d, v0, min(ml, 8u));
                    if (long_) lds_wr_bytes(L.ring, RM, d + 8, v1, ml > 8 ? ml - 8 : 0u);
                }
                rem &= ~__ballot(go);
                wave_sync();
                LZXT_CNT(7, 1);
                continue;
            }
            rem &= rem - 1;
            LZXT_CNT(5, 1);
            const uint32_t fd = rdlane(d, first), fml = rdlane(ml, first), foff = rdlane(off, first);
            const uint32_t fs = fd - foff;
            if (__builtin_expect(fml <= 64 && foff >= fml && !((mixed_m >> first) & 1), 1)) {
                if (lane < fml) L.ring[(fd + lane) & RM] = L.ring[(fs + lane) & RM];
                continue;
            }
            const bool mx = (mixed_m >> first) & 1;
            const bool periodic = foff < 64 && foff < fml;
            for (uint32_t i0 = 0; i0 < fml; i0 += 64) {
                const uint32_t i = i0 + lane;
                if (i < fml) {
                    const uint32_t sp = fs + (periodic ? i % foff : i);
                    const uint8_t v = (mx && sp < fl) ? ldu8(gbase + sp) : L.ring[sp & RM];
                    L.ring[(fd + i) & RM] = v;
                }
                wave_sync();
            }
        }
        LZXT(3);
        flush(g_end, false);
        LZXT(4);
        return total;
    }
    __device__ void finish(uint32_t op) { flush(op + a0, true); }
};

// ------------------------------------------------------------------------------------------------ encode
// The matcher looks back only as far as its LDS ring reaches (HB hash bits, 2^RB ring bytes): on zipf text 8 KiB of
// history costs ~7 % of liblz4's ratio, 14 KiB ~5 % — and every probe, candidate check and match extension is an LDS
// access, no HBM round trip inside a step.  Only matches longer than LZE_CAP are extended against HBM, wave-wide.
constexpr uint32_t LZE_OUT = 2048;       // output staging
constexpr uint32_t LZE_CAP = 32;         // per-lane match extension per step; longer matches extend wave-wide
constexpr uint32_t LZE_LIT_LANE = 48;    // sequences with more literals than this are emitted wave-wide, one by one
constexpr uint32_t LZE_AHEAD = 2048;     // input loaded this far beyond the step's first position
template <int HB, int RB>
struct Lz4EncLds {
    uint16_t tab[1u << HB];                                     // position & 0xFFFF of the last occurrence of a hash
    __attribute__((aligned(16))) uint32_t ring[(1u << RB) / 4]; // input byte x lives at ring byte x % 2^RB
    __attribute__((aligned(16))) uint8_t out[LZE_OUT + 32];
};

__device__ __forceinline__ uint32_t lz4_bound(uint32_t n) { return n + n / 255 + 16; }

// The matcher, shared by the LZ4 and the Zstd encoder: hash table + input ring in LDS, one step = 64 probe positions.
// next() runs steps until one finds matches and leaves, per lane, the position p, the match (cand, mlen) and — for the
// lanes chosen by the greedy / lazy selection (mask C) — nothing else to decide: the caller emits the sequences
// [lit_start .. p) + match and calls advance().  A chunk [c0, c1) bounds the sequences (matches end at or before
// c1 - tail_lits, start at or before c1 - min_tail); the history in the ring and the table carries over between chunks.
template <int HB, int RB>
struct LzMatcher {
    static constexpr uint32_t R = 1u << RB, RWM = R / 4 - 1;
    Lz4EncLds<HB, RB>& L;
    const uint8_t* src;
    uint32_t n;                            // bytes of the whole input
    uint32_t hi = 0;                       // input loaded into the ring up to here (multiple of 1024, or >= n)
    uint32_t lo0 = 0;                      // lowest input position the ring ever held (chunks compressed on their own start late)
    uint32_t anchor = 0, base = 0, stride = 1;
    uint32_t mflimit = 0, matchlimit = 0;  // a match may start at positions <= mflimit and must end at or before matchlimit
    // results of next()
    uint64_t C = 0;
    uint32_t p = 0, mlen = 0, cand = 0, covered = 0;
    unsigned long long* prof = nullptr;   // (SB_LZ4_PROFILE builds: the caller's phase accumulators)
    unsigned long long pt = 0;

    __device__ LzMatcher(Lz4EncLds<HB, RB>& l, const uint8_t* s, uint32_t nn) : L(l), src(s), n(nn) {}
    __device__ void init() {
        const uint32_t lane = threadIdx.x & 63;
        for (uint32_t i = lane; i < (1u << HB); i += 64) L.tab[i] = 0xFFFF;
        wave_sync();
    }
    __device__ void begin_chunk(uint32_t c0, uint32_t mfl, uint32_t mtl) {
        anchor = base = c0;
        stride = 1;
        mflimit = mfl;
        matchlimit = mtl;
    }
    // A chunk compressed by a wave of its own (c0 % 1024 == 0): the 5 KiB of input before it — what the ring of a wave
    // that had compressed them would still hold — are loaded and entered into the table, so matches reach back over the
    // chunk border as they do inside a chunk (the input is all there; only the OUTPUT of the chunks is independent).
    // history = false: no match may reach below c0 (the chunk becomes a Zstd frame of its own: a frame's window starts with it)
    __device__ void begin_alone(uint32_t c0, uint32_t mfl, uint32_t mtl, bool history = true) {
        const uint32_t lane = threadIdx.x & 63;
        hi = lo0 = (history && c0 >= 5120) ? c0 - 5120 : c0;
        begin_chunk(c0, mfl, mtl);
        if (c0 == 0 || !history) return;
        fill(c0 + 1024);
        for (uint32_t x = lo0 + lane; x < c0; x += 64) L.tab[(rd4(x) * 2654435761u) >> (32 - HB)] = (uint16_t)(x & 0xFFFF);
        wave_sync();
    }
    __device__ void advance() {
        anchor = covered;
        base = max(covered, base + 1);
        stride = 1;
    }
    __device__ __forceinline__ uint32_t rd4(uint32_t x) const {
        const uint32_t w = (x >> 2) & RWM;
        return __builtin_amdgcn_alignbyte(L.ring[(w + 1) & RWM], L.ring[w], x & 3);
    }
    __device__ __forceinline__ uint64_t rd8(uint32_t x) const {
        const uint32_t w = (x >> 2) & RWM;
        const uint32_t a = L.ring[w], b = L.ring[(w + 1) & RWM], c = L.ring[(w + 2) & RWM];
        return (uint64_t)__builtin_amdgcn_alignbyte(b, a, x & 3) | ((uint64_t)__builtin_amdgcn_alignbyte(c, b, x & 3) << 32);
    }
    __device__ void fill(uint32_t upto) {      // 1 KiB per instruction, 16-byte loads
        const uint32_t lane = threadIdx.x & 63;
        wave_sync();
        while (hi < n && hi < upto) {
            const uint32_t x = hi + 16 * lane;
            u32x4 v = {0, 0, 0, 0};
            if (x + 16 <= n) {
                v = ldu128(src + x);
            } else if (x < n) {
                uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;
                for (uint32_t b = 0; x + b < n; b++) {
                    const uint32_t bv = (uint32_t)ldu8(src + x + b) << ((b & 3) * 8);
                    if (b < 4) w0 |= bv; else if (b < 8) w1 |= bv; else if (b < 12) w2 |= bv; else w3 |= bv;
                }
                v = u32x4{w0, w1, w2, w3};
            }
            *(u32x4*)(L.ring + ((x & (R - 1)) >> 2)) = v;
            hi += 1024;
        }
        wave_sync();
    }
    // true: a step with chosen matches is ready; false: the chunk is exhausted (anchor = start of its trailing literals)
    __device__ bool next() {
        const uint32_t lane = threadIdx.x & 63;
        MTP_START;
        while (base <= mflimit) {
            fill(base + LZE_AHEAD);
            MTP(0);
            MTP_CNT(16, 1);
            const uint32_t lo_valid = max(hi > R ? hi - R : 0, lo0);   // positions below are not in the ring
            p = base + lane * stride;
            // a probe needs its LZE_CAP + 16 bytes of look-ahead in the ring (a step spread wider probes only its front part)
            const bool act = p <= mflimit && (p + LZE_CAP + 16 <= hi || hi >= n);
            cand = 0;
            mlen = 0;
            bool open = false;    // the extension stopped at LZE_CAP, not at a mismatch
            if (act) {
                // round trip 1: the eight dwords that hold bytes [p - 8, p + 20); t[k] = the 4 bytes at p - 8 + 4k
                const uint32_t w0 = (p - 8) >> 2, q = p & 3;
                uint32_t D[8], t[7];
    #pragma unroll
                for (int k = 0; k < 8; k++) D[k] = L.ring[(w0 + k) & RWM];
    #pragma unroll
                for (int k = 0; k < 7; k++) t[k] = __builtin_amdgcn_alignbyte(D[k + 1], D[k], q);
                const uint32_t v4 = t[2];
                // round trip 2: the hash table
                const uint32_t h = (v4 * 2654435761u) >> (32 - HB);
                const uint32_t e = L.tab[h];
                L.tab[h] = (uint16_t)(p & 0xFFFF);
                uint32_t c = 0;
                bool ok = false;
                if (stride == 1) {   // the periods columnar data repeats with (positions of this very step are not in the table
                                     // yet, and the nearest candidate gives the longest runs): 8, 4, 2, 1 bytes back
                    if (p >= lo0 + 8 && t[0] == v4) { c = p - 8; ok = true; }
                    else if (p >= lo0 + 4 && t[1] == v4) { c = p - 4; ok = true; }
                    else if (p >= lo0 + 2 && ((t[1] >> 16) | (t[2] << 16)) == v4) { c = p - 2; ok = true; }
                    else if (p >= lo0 + 1 && ((t[1] >> 24) | (t[2] << 8)) == v4) { c = p - 1; ok = true; }
                }
                if (!ok && e != 0xFFFF) {   // the last position with this hash (mod 64 Ki), if it is still in the ring
                    c = (p & ~0xFFFFu) | e;
                    if (c >= p) c -= 0x10000u;                   // (wraps above p when there is no such position)
                    ok = c < p && c >= lo_valid;
                }
                if (ok) {
                    // round trip 3: 16 candidate bytes against the 16 bytes at p
                    const uint32_t cw = c >> 2, cq = c & 3;
                    uint32_t E[5];
    #pragma unroll
                    for (int k = 0; k < 5; k++) E[k] = L.ring[(cw + k) & RWM];
                    uint32_t d;
                    if ((d = __builtin_amdgcn_alignbyte(E[1], E[0], cq) ^ v4) != 0) ok = false;
                    else if ((d = __builtin_amdgcn_alignbyte(E[2], E[1], cq) ^ t[3]) != 0) mlen = 4 + (__builtin_ctz(d) >> 3);
                    else if ((d = __builtin_amdgcn_alignbyte(E[3], E[2], cq) ^ t[4]) != 0) mlen = 8 + (__builtin_ctz(d) >> 3);
                    else if ((d = __builtin_amdgcn_alignbyte(E[4], E[3], cq) ^ t[5]) != 0) mlen = 12 + (__builtin_ctz(d) >> 3);
                    else { mlen = 16; open = true; }
                }
                if (ok) {
                    cand = c;
                    const uint32_t room = matchlimit - p;        // longest match allowed here (>= 7)
                    while (open && mlen < LZE_CAP) {             // further round trips: 8 bytes each
                        const uint64_t d = rd8(c + mlen) ^ rd8(p + mlen);
                        if (d) { mlen += (uint32_t)__builtin_ctzll(d) >> 3; open = false; break; }
                        mlen += 8;
                    }
                    if (mlen >= room) { mlen = room; open = false; }
                } else {
                    mlen = 0;
                    open = false;
                }
            }
            const uint64_t mm = __ballot(mlen >= 4);
            MTP(1);
            if (!mm) {
                base += 64 * stride;
                if (stride < 24) stride++;
                continue;
            }
            // ---- greedy left-to-right selection of non-overlapping matches of this step.  Lazy matching first, per lane: a
            // match steps aside when one of the next three positions starts a match that is longer by more than the
            // distance.  Then the serial part — a wave-uniform walk over the match mask — is one v_readlane and one taken
            // branch per chosen sequence; literal starts, sizes and staging offsets are computed per lane afterwards.
            uint64_t msel = mm;
            if (stride == 1) {
                const uint32_t m1 = __shfl_down(mlen, 1, 64), m2 = __shfl_down(mlen, 2, 64), m3 = __shfl_down(mlen, 3, 64);
                const bool dom = mlen >= 4 && ((lane < 63 && m1 > mlen + 1) || (lane < 62 && m2 > mlen + 2) || (lane < 61 && m3 > mlen + 3));
                msel &= ~__ballot(dom);
            }
            const uint32_t mcomb = mlen | (open ? 0x80000000u : 0u);
            C = 0;
            covered = anchor;            // everything below is emitted or pending as literals of the next sequence
            // wave-wide extension of the match at lane l (position pl) beyond its per-lane part, 64 x 4 bytes per step from HBM
            auto extend = [&](uint32_t l, uint32_t pl, uint32_t ml_l) -> uint32_t {
                const uint32_t c = rdlane(cand, l);
                for (;;) {
                    const uint32_t x = ml_l + 4 * lane;
                    bool eq = pl + x + 4 <= matchlimit;
                    if (eq) eq = ldu32(src + c + x) == ldu32(src + pl + x);
                    const uint64_t ne = __ballot(!eq);
                    const uint32_t take = ne ? (uint32_t)__builtin_ctzll(ne) : 64u;
                    ml_l += 4 * take;
                    if (ne) break;
                }
                while (pl + ml_l < matchlimit && ldu8(src + c + ml_l) == ldu8(src + pl + ml_l)) ml_l++;  // < 4 steps
                mlen = wrlane(ml_l, l, mlen);
                return ml_l;
            };
            if (stride == 1) {
                uint32_t rel = 0;                 // (anchor <= base: the first lane is free)
                do {
                    const uint64_t m = (msel >> rel) << rel;
                    if (!m) break;
                    const uint32_t l = (uint32_t)__builtin_ctzll(m);
                    const uint32_t v = rdlane(mcomb, l);
                    uint32_t ml_l = v & 0x7FFFFFFFu;
                    if (__builtin_expect(v >> 31, 0)) ml_l = extend(l, base + l, ml_l);
                    C |= 1ull << l;
                    rel = l + ml_l;
                } while (rel < 64);
                if (C) covered = base + rel;
            } else {
                for (;;) {
                    // lanes whose position is >= covered form a suffix: first lane = ceil((covered - base) / stride)
                    const uint32_t first_lane = covered <= base ? 0u : (covered - base + stride - 1) / stride;
                    if (first_lane >= 64) break;
                    const uint64_t m = (msel >> first_lane) << first_lane;
                    if (!m) break;
                    const uint32_t l = (uint32_t)__builtin_ctzll(m);
                    const uint32_t v = rdlane(mcomb, l);
                    uint32_t ml_l = v & 0x7FFFFFFFu;
                    const uint32_t pl = base + l * stride;
                    if (v >> 31) ml_l = extend(l, pl, ml_l);
                    C |= 1ull << l;
                    covered = pl + ml_l;
                }
            }
            MTP(2);
            return true;
        }
        return false;
    }
};

// The sequences of src[c0, c1) — a whole block (c0 = 0, c1 = n) or one chunk of it compressed on its own (ALONE: no
// match reaches below c0 or beyond c1) — written to dst; executed by ONE wave64.  Returns their size; *tail_anchor =
// where the literals begin that no sequence holds (they open the next chunk's first sequence, or close the block).
// LZ4's end-of-block rules hold for every chunk: the last match starts >= 12 bytes before n, the last 5 bytes are literals.
template <int HB, int RB, bool ALONE>
__device__ uint32_t lz4_compress_range(const uint8_t* src, uint32_t n, uint32_t c0, uint32_t c1, uint8_t* dst, Lz4EncLds<HB, RB>& L,
                                       uint32_t* tail_anchor) {
    constexpr uint32_t R = 1u << RB, RWM = R / 4 - 1;
    const uint32_t lane = threadIdx.x & 63;
    uint32_t outp = 0;        // bytes already written to dst
    uint32_t on = 0;          // bytes staged in L.out
    LZP_BEGIN
    auto flush_out = [&]() {
        LZP(3);
        wave_sync();
        for (uint32_t k = lane; k < on; k += 64) dst[outp + k] = L.out[k];
        outp += on;
        on = 0;
        wave_sync();
        LZP(4);
    };
    // token + literal-length extension for one sequence written by lane 0 straight to dst; returns its size
    auto put_head = [&](uint32_t lit, uint32_t mcode) -> uint32_t {
        if (lane == 0) {
            uint32_t o = outp;
            dst[o++] = (uint8_t)((min(lit, 15u) << 4) | min(mcode, 15u));
            if (lit >= 15) {
                uint32_t r = lit - 15;
                while (r >= 255) { dst[o++] = 255; r -= 255; }
                dst[o++] = (uint8_t)r;
            }
        }
        return 1 + (lit >= 15 ? 1 + div255(lit - 15) : 0);
    };
    *tail_anchor = c0;
    if (n < 13 || c0 + 12 > n) return 0;   // LZ4_minLength = mflimit + 1: no match possible
    LzMatcher<HB, RB> mt(L, src, n);
#ifdef SB_LZ4_PROFILE
    mt.prof = lzp_acc;
#endif
    mt.init();
    if (ALONE)
        mt.begin_alone(c0, min(c1 - 4, n - 12), min(c1, n - 5));
    else
        mt.begin_chunk(c0, n - 12, n - 5);
    while (mt.next()) {
        LZP_RESET;
        const bool chosen = (mt.C >> lane) & 1;
        uint32_t lit_start_v = mt.anchor;
        {
            const uint64_t below = mt.C & ((1ull << lane) - 1);
            const uint32_t prevl = below ? 63u - (uint32_t)__builtin_clzll(below) : 0u;
            const uint32_t pe = __shfl(mt.p + mt.mlen, prevl, 64);     // end of the previous chosen match
            if (below) lit_start_v = pe;
        }
        uint32_t sz = 0;
        if (chosen) {
            const uint32_t lit = mt.p - lit_start_v, mcode = mt.mlen - 4;
            sz = 1 + (lit >= 15 ? 1 + div255(lit - 15) : 0) + lit + 2 + (mcode >= 15 ? 1 + div255(mcode - 15) : 0);
        }
        const uint64_t big = __ballot(chosen && mt.p - lit_start_v > LZE_LIT_LANE);
        const uint32_t incl_sz = wave_scan_dpp(sz);
        const uint32_t out_off_v = incl_sz - sz;
        const uint32_t run = rdlane(incl_sz, 63);
        LZP(6);
        LZP_CNT(17, __popcll(mt.C));
        if (big || run > LZE_OUT) LZP_CNT(18, 1);
        if (!big && run <= LZE_OUT) {
            // ---- every chosen lane writes its own sequence into the staging buffer
            if (on + run > LZE_OUT) flush_out();
            if (chosen) {
                const uint32_t lit = mt.p - lit_start_v, mcode = mt.mlen - 4, off = mt.p - mt.cand;
                uint8_t* o = L.out + on + out_off_v;
                uint32_t k = 0;
                o[k++] = (uint8_t)((min(lit, 15u) << 4) | min(mcode, 15u));
                if (lit >= 15) {
                    uint32_t r = lit - 15;
                    while (r >= 255) { o[k++] = 255; r -= 255; }
                    o[k++] = (uint8_t)r;
                }
                for (uint32_t i = 0; i < lit; i += 8) {   // (<= 48 bytes back: in the ring)
                    const uint64_t v = lds_rd8_ring(L.ring, (lit_start_v + i) & (R - 1), RWM);
                    const uint32_t nb = min(8u, lit - i);
#pragma unroll
                    for (uint32_t b = 0; b < 8; b++)
                        if (b < nb) o[k + b] = (uint8_t)(v >> (8 * b));
                    k += nb;
                }
                o[k++] = (uint8_t)off;
                o[k++] = (uint8_t)(off >> 8);
                if (mcode >= 15) {
                    uint32_t r = mcode - 15;
                    while (r >= 255) { o[k++] = 255; r -= 255; }
                    o[k++] = (uint8_t)r;
                }
            }
            on += run;
            wave_sync();
            LZP(3);
        } else {
            // ---- one sequence at a time, straight to dst (long literal runs)
            flush_out();
            uint64_t m = mt.C;
            while (m) {
                const uint32_t l = (uint32_t)__builtin_ctzll(m);
                m &= m - 1;
                const uint32_t pl = mt.base + l * mt.stride;
                const uint32_t s_ls = rdlane(lit_start_v, l), s_mc = rdlane(mt.mlen, l) - 4, s_off = pl - rdlane(mt.cand, l);
                const uint32_t s_lit = pl - s_ls;
                outp += put_head(s_lit, s_mc);
                wave_copy_g2g(dst + outp, src + s_ls, s_lit);
                outp += s_lit;
                if (lane == 0) {
                    uint32_t o = outp;
                    dst[o++] = (uint8_t)s_off;
                    dst[o++] = (uint8_t)(s_off >> 8);
                    if (s_mc >= 15) {
                        uint32_t r = s_mc - 15;
                        while (r >= 255) { dst[o++] = 255; r -= 255; }
                        dst[o++] = (uint8_t)r;
                    }
                }
                outp += 2 + (s_mc >= 15 ? 1 + div255(s_mc - 15) : 0);
            }
        }
        mt.advance();
    }
    LZP(3);
    flush_out();
    *tail_anchor = mt.anchor;
    LZP(5);
    LZP_END;
    return outp;
}

// token + literal-length extension of a sequence with `lit` literals and match code `mcode`, by one lane; returns the size
__device__ __forceinline__ uint32_t lz4_put_head(uint8_t* o, uint32_t lit, uint32_t mcode) {
    uint32_t k = 0;
    o[k++] = (uint8_t)((min(lit, 15u) << 4) | min(mcode, 15u));
    if (lit >= 15) {
        uint32_t r = lit - 15;
        while (r >= 255) { o[k++] = 255; r -= 255; }
        o[k++] = (uint8_t)r;
    }
    return k;
}
__device__ __forceinline__ uint32_t lz4_head_bytes(uint32_t lit) { return 1 + (lit >= 15 ? 1 + div255(lit - 15) : 0); }

// Compress src[0, n) into dst (capacity >= lz4_bound(n)); executed by ONE wave64; returns the block size.
template <int HB, int RB>
__device__ uint32_t lz4_compress_wave_fast(const uint8_t* src, uint32_t n, uint8_t* dst, Lz4EncLds<HB, RB>& L) {
    const uint32_t lane = threadIdx.x & 63;
    uint32_t anchor = 0;
    uint32_t outp = lz4_compress_range<HB, RB, false>(src, n, 0, n, dst, L, &anchor);
    const uint32_t lit = n - anchor;     // the literals-only last sequence
    if (lane == 0) lz4_put_head(dst + outp, lit, 0);
    outp += lz4_head_bytes(lit);
    wave_copy_g2g(dst + outp, src + anchor, lit);
    return outp + lit;
}


// ------------------------------------------------------------------------------------------------ Snappy encode
// Raw Snappy stream (what snap::raw::Encoder emits and snap::raw::Decoder / any Snappy library reads; reference call
// sites src/compression/basic.rs:137-152 / :99-106): uvarint(uncompressed length), then literal and copy elements.
// The parse is the LZ4 matcher's (64 probe positions per step over the LDS ring); only the element syntax differs:
//   literal: tag (len-1) << 2 for len <= 60, else tag (59 + nb) << 2 followed by nb little-endian bytes of len-1
//   copy1  : len 4..11, offset < 2048: ((off >> 8) << 5) | ((len - 4) << 2) | 1, off & 0xFF
//   copy2  : len 1..64: ((len - 1) << 2) | 2, off lo, off hi      (longer matches: several copies)
__device__ __forceinline__ uint32_t snappy_lit_head(uint32_t lit) {   // bytes of a literal element's tag (lit >= 1)
    const uint32_t m = lit - 1;
    return m < 60 ? 1u : m < (1u << 8) ? 2u : m < (1u << 16) ? 3u : m < (1u << 24) ? 4u : 5u;
}
template <class P>
__device__ __forceinline__ uint32_t snappy_put_lit_head(P o, uint32_t lit) {
    const uint32_t m = lit - 1;
    if (m < 60) {
        o[0] = (uint8_t)(m << 2);
        return 1;
    }
    const uint32_t nb = m < (1u << 8) ? 1u : m < (1u << 16) ? 2u : m < (1u << 24) ? 3u : 4u;
    o[0] = (uint8_t)((59 + nb) << 2);
    for (uint32_t k = 0; k < nb; k++) o[1 + k] = (uint8_t)(m >> (8 * k));
    return 1 + nb;
}
// the copy elements of one match: full 64-byte copies first (60 when 65..67 bytes remain, so that the rest is >= 4)
__device__ __forceinline__ uint32_t snappy_copy_bytes(uint32_t ml, uint32_t off) {
    uint32_t sz = 0;
    while (ml >= 68) { sz += 3; ml -= 64; }
    if (ml > 64) { sz += 3; ml -= 60; }
    return sz + ((ml < 12 && off < 2048) ? 2u : 3u);
}
template <class P>
__device__ __forceinline__ uint32_t snappy_put_copies(P o, uint32_t ml, uint32_t off) {
    uint32_t k = 0;
    while (ml >= 68) {
        o[k++] = (uint8_t)((63u << 2) | 2u); o[k++] = (uint8_t)off; o[k++] = (uint8_t)(off >> 8);
        ml -= 64;
    }
    if (ml > 64) {
        o[k++] = (uint8_t)((59u << 2) | 2u); o[k++] = (uint8_t)off; o[k++] = (uint8_t)(off >> 8);
        ml -= 60;
    }
    if (ml < 12 && off < 2048) {
        o[k++] = (uint8_t)(((off >> 8) << 5) | ((ml - 4) << 2) | 1u); o[k++] = (uint8_t)off;
    } else {
        o[k++] = (uint8_t)(((ml - 1) << 2) | 2u); o[k++] = (uint8_t)off; o[k++] = (uint8_t)(off >> 8);
    }
    return k;
}

// The elements of src[c0, c1) — a whole buffer or one chunk of it compressed by a wave of its own (ALONE: matches reach
// back over the chunk border through the pre-loaded history, copies never cross it) — incl. the literal element that
// closes the range: Snappy elements carry no state, so the chunks' outputs concatenate to one stream.  ONE wave64.
template <int HB, int RB, bool ALONE>
__device__ uint32_t snappy_compress_range(const uint8_t* src, uint32_t n, uint32_t c0, uint32_t c1, uint8_t* dst, Lz4EncLds<HB, RB>& L) {
    constexpr uint32_t R = 1u << RB, RWM = R / 4 - 1;
    const uint32_t lane = threadIdx.x & 63;
    uint32_t outp = 0, on = 0;
    auto flush_out = [&]() {
        wave_sync();
        for (uint32_t k = lane; k < on; k += 64) dst[outp + k] = L.out[k];
        outp += on;
        on = 0;
        wave_sync();
    };
    auto put_literal = [&](uint32_t from, uint32_t lit) {   // wave-wide literal element straight to dst
        if (!lit) return;
        if (lane == 0) snappy_put_lit_head(dst + outp, lit);
        outp += snappy_lit_head(lit);
        wave_copy_g2g(dst + outp, src + from, lit);
        outp += lit;
    };
    uint32_t anchor = c0;
    if (c1 - c0 >= 16) {
        LzMatcher<HB, RB> mt(L, src, n);
        mt.init();
        if (ALONE)
            mt.begin_alone(c0, c1 - 4, c1);
        else
            mt.begin_chunk(c0, c1 - 4, c1);
        while (mt.next()) {
            const bool chosen = (mt.C >> lane) & 1;
            uint32_t lit_start_v = mt.anchor;
            {
                const uint64_t below = mt.C & ((1ull << lane) - 1);
                const uint32_t prevl = below ? 63u - (uint32_t)__builtin_clzll(below) : 0u;
                const uint32_t pe = __shfl(mt.p + mt.mlen, prevl, 64);
                if (below) lit_start_v = pe;
            }
            const uint32_t lit = mt.p - lit_start_v, off = mt.p - mt.cand;
            uint32_t sz = 0;
            if (chosen) sz = (lit ? snappy_lit_head(lit) + lit : 0u) + snappy_copy_bytes(mt.mlen, off);
            const uint64_t big = __ballot(chosen && (lit > LZE_LIT_LANE || mt.mlen > 256));
            const uint32_t incl_sz = wave_scan_dpp(sz);
            const uint32_t out_off_v = incl_sz - sz;
            const uint32_t run = rdlane(incl_sz, 63);
            if (!big && run <= LZE_OUT) {
                if (on + run > LZE_OUT) flush_out();
                if (chosen) {
                    uint8_t* o = L.out + on + out_off_v;
                    uint32_t k = 0;
                    if (lit) {
                        k = snappy_put_lit_head(o, lit);
                        for (uint32_t i = 0; i < lit; i += 8) {   // (<= 48 bytes back: in the ring)
                            const uint64_t v = lds_rd8_ring(L.ring, (lit_start_v + i) & (R - 1), RWM);
                            const uint32_t nb = min(8u, lit - i);
#pragma unroll
                            for (uint32_t b = 0; b < 8; b++)
                                if (b < nb) o[k + b] = (uint8_t)(v >> (8 * b));
                            k += nb;
                        }
                    }
                    snappy_put_copies(o + k, mt.mlen, off);
                }
                on += run;
                wave_sync();
            } else {
                flush_out();
                uint64_t m = mt.C;
                while (m) {
                    const uint32_t l = (uint32_t)__builtin_ctzll(m);
                    m &= m - 1;
                    const uint32_t pl = mt.base + l * mt.stride;
                    const uint32_t s_ls = rdlane(lit_start_v, l), s_ml = rdlane(mt.mlen, l), s_off = pl - rdlane(mt.cand, l);
                    put_literal(s_ls, pl - s_ls);
                    // copies: the 64-byte ones lane-parallel, the tail by lane 0
                    const uint32_t full = (s_ml - 4) / 64;   // 64-byte copies that leave a rest of 4..67 bytes
                    const uint32_t rest = s_ml - 64 * full;
                    for (uint32_t k = lane; k < full; k += 64) {
                        uint8_t* o = dst + outp + 3 * k;
                        o[0] = (uint8_t)((63u << 2) | 2u); o[1] = (uint8_t)s_off; o[2] = (uint8_t)(s_off >> 8);
                    }
                    outp += 3 * full;
                    if (lane == 0) snappy_put_copies(dst + outp, rest, s_off);
                    outp += snappy_copy_bytes(rest, s_off);
                }
            }
            mt.advance();
        }
        anchor = mt.anchor;
    }
    flush_out();
    put_literal(anchor, c1 - anchor);
    wave_stores_visible();
    return outp;
}
__device__ __forceinline__ uint32_t snappy_put_preamble(uint8_t* dst, uint32_t n) {   // uvarint(n) by one lane; returns its size
    uint32_t o = 0, v = n;
    while (v >= 0x80) {
        dst[o++] = (uint8_t)(v | 0x80);
        v >>= 7;
    }
    dst[o++] = (uint8_t)v;
    return o;
}
__device__ __forceinline__ uint32_t snappy_preamble_bytes(uint32_t n) { return n < (1u << 7) ? 1u : n < (1u << 14) ? 2u : n < (1u << 21) ? 3u : n < (1u << 28) ? 4u : 5u; }

// Compress src[0, n) into dst (capacity >= 32 + n + n / 6); executed by ONE wave64; returns the stream size.
template <int HB, int RB>
__device__ uint32_t snappy_compress_wave(const uint8_t* src, uint32_t n, uint8_t* dst, Lz4EncLds<HB, RB>& L) {
    if ((threadIdx.x & 63) == 0) snappy_put_preamble(dst, n);
    const uint32_t h = snappy_preamble_bytes(n);
    return h + snappy_compress_range<HB, RB, false>(src, n, 0, n, dst + h, L);
}

}  // namespace sb
