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

// ------------------------------------------------------------------------------------------------ decode
constexpr uint32_t LZD_IB = 2048;        // compressed bytes staged in LDS
constexpr uint32_t LZD_SHORT_LIT = 300;  // literals up to here are copied from the staged input
constexpr uint32_t LZD_MARGIN = 64 + 3 + LZD_SHORT_LIT + 2 + 3 + 16;  // window + extensions + literal + offset + extensions
constexpr uint32_t LZD_ST = 8192;        // output staging window (bytes)
constexpr uint32_t LZD_MB = 128;         // matches per batch
struct Lz4DecLds {
    __attribute__((aligned(16))) uint8_t ib[LZD_IB + 16];
    __attribute__((aligned(16))) uint8_t st[LZD_ST + 16];
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
__device__ uint32_t lz4_inflate_block(const uint8_t* src, uint32_t n, uint8_t* dst, uint32_t out_len, Lz4DecLds& L) {
    const uint32_t lane = threadIdx.x & 63;
    if (n == 0) return out_len != 0 ? 100u : 0u;
    // g-space: G = output position + a0, so that G % 16 == address % 16 (aligned 16-byte flushes)
    const uint32_t a0 = (uint32_t)((uintptr_t)dst & 15);
    uint8_t* gbase = dst - a0;
    uint32_t S0 = 0;          // g-position of st[0] (multiple of 16)
    uint32_t op = 0;          // output position (bytes produced so far, incl. staged)
    uint32_t ip = 0;          // input position
    uint32_t ibase = 0;       // ib holds src[ibase, ibase + ibn), ibase a multiple of 16
    uint32_t ibn = 0;
    uint32_t nm = 0;          // matches recorded in the current batch
    uint32_t err = 0;
    bool done = false;

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
    // staging -> HBM: g-positions [S0, g_end).  final: everything; otherwise whole 16-byte groups, the partial last
    // group stays at the front of the window.  Bytes below a0 (in front of dst) are never written.
    auto flush = [&](uint32_t g_end, bool final) {
        wave_sync();
        const uint32_t full_end = final ? g_end : (g_end & ~15u);
        uint32_t g = S0;
        if (g < a0) {  // (S0 == 0) the first group starts in front of dst
            const uint32_t e = min(16u, full_end);
            if (lane >= a0 && lane < e) gbase[lane] = L.st[lane];
            g = min(16u, max(full_end, S0));
        }
        const uint32_t ng = full_end > g ? (full_end - g) >> 4 : 0;
        for (uint32_t k = lane; k < ng; k += 64) stu128(gbase + g + 16 * k, *(const u32x4*)(L.st + (g - S0) + 16 * k));
        g += ng << 4;
        if (final && g < g_end && lane < g_end - g) gbase[g + lane] = L.st[g - S0 + lane];
        if (!final && full_end > S0) {
            const uint32_t keep = g_end - full_end;
            const uint32_t v = lane < keep ? (uint32_t)L.st[full_end - S0 + lane] : 0u;
            wave_sync();
            if (lane < keep) L.st[lane] = (uint8_t)v;
            S0 = full_end;
        }
        wave_sync();
    };
    // the matches recorded in L.m_* (in sequence order), then flush
    auto run_batch = [&](bool final) {
        if (nm) {
            wave_stores_visible();  // earlier flushes must have landed before far sources are read back
            const uint32_t o_flushed = S0 > a0 ? S0 - a0 : 0;  // output positions below this are in HBM
            uint64_t near_m[2] = {0, 0};
            for (uint32_t k0 = 0; k0 < nm; k0 += 64) {
                const uint32_t k = k0 + lane;
                bool near = false;
                if (k < nm) {
                    const uint32_t d = L.m_dst[k], ml = L.m_len[k], off = L.m_off[k];
                    const uint32_t s = d - off;
                    if (s + min(ml, off) <= o_flushed) {  // far: the source bytes are in HBM
                        uint8_t* w = L.st + (d + a0 - S0);
                        if (off >= 16 && s + ((ml + 15) & ~15u) <= o_flushed) {
                            for (uint32_t i = 0; i < ml; i += 16) {
                                const u32x4 v = ldu128(dst + s + i);
                                uint64_t lo = (uint64_t)v.x | ((uint64_t)v.y << 32), hi = (uint64_t)v.z | ((uint64_t)v.w << 32);
                                const uint32_t nb = min(16u, ml - i);
                                for (uint32_t b = 0; b < nb; b++) {
                                    w[i + b] = (uint8_t)lo;
                                    lo = (lo >> 8) | (hi << 56);
                                    hi >>= 8;
                                }
                            }
                        } else {
                            for (uint32_t i = 0; i < ml; i++) w[i] = ldu8(dst + s + (off >= ml ? i : i % off));
                        }
                    } else {
                        near = true;
                    }
                }
                near_m[k0 >> 6] = __ballot(near);
            }
            wave_sync();
            // near matches in sequence order, the wave copies one match at a time (LDS -> LDS)
            for (uint32_t half = 0; half < 2; half++) {
                uint64_t m = near_m[half];
                while (m) {
                    const uint32_t k = (uint32_t)__builtin_ctzll(m) + half * 64;
                    m &= m - 1;
                    const uint32_t d = L.m_dst[k], ml = L.m_len[k], off = L.m_off[k];
                    const uint32_t s = d - off;
                    const bool mixed = s < o_flushed;  // the source straddles the flushed / staged boundary
                    const bool periodic = off < 64 && off < ml;  // then every byte comes from [s, s + off), all below d
                    for (uint32_t i0 = 0; i0 < ml; i0 += 64) {
                        const uint32_t i = i0 + lane;
                        if (i < ml) {
                            const uint32_t sp = s + (periodic ? i % off : i);
                            uint8_t v;
                            if (mixed && sp < o_flushed) v = ldu8(dst + sp); else v = L.st[sp + a0 - S0];
                            L.st[d + i + a0 - S0] = v;
                        }
                        wave_sync();
                    }
                }
            }
            nm = 0;
        }
        flush(op + a0, final);
    };

    refill(0);
    while (!done && !err) {
        if (ip + LZD_MARGIN > ibase + ibn && ibase + ibn < n) refill(ip);
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
        const uint32_t tot = lit + ml;                   // output bytes of my sequence
        uint64_t M = 0;
        uint32_t cur = 0, acc = 0, cnt_m = 0, stop = 0;  // stop: 1 = batch full, 2 = big, 3 = last, 4 = bad
        uint32_t my_op = 0, my_k = 0;
        const uint32_t room = LZD_ST - (op + a0 - S0);   // staging bytes left
        while (cur < 64) {
            const uint32_t f = rdlane(flags, cur);
            if (f & 4) { stop = 4; break; }
            if (f & 1) { stop = 2; break; }
            const uint32_t t = rdlane(tot, cur);
            if (acc + t > room || nm + cnt_m >= LZD_MB) { stop = 1; break; }
            M |= 1ull << cur;
            my_op = wrlane(op + acc, cur, my_op);
            my_k = wrlane(nm + cnt_m, cur, my_k);
            acc += t;
            if (f & 2) { stop = 3; break; }
            cnt_m++;
            cur = rdlane(nxt, cur) - ip;
        }
        const bool mine = (M >> lane) & 1;
        bool bad = false;
        if (mine) {
            if (my_op + tot > out_len) bad = true;
            if (!(flags & 2) && (off == 0 || off > my_op + lit)) bad = true;
        }
        if (__ballot(bad)) { err = 103; break; }
        {   // literals: staged input -> staging window; per lane up to 32 bytes, longer ones wave-wide
            const uint32_t wpos = my_op + a0 - S0, rpos = q - ibase;
            const uint32_t nl = mine ? min(lit, 32u) : 0u;
            for (uint32_t i = 0; i < nl; i++) L.st[wpos + i] = L.ib[rpos + i];
            uint64_t lm = __ballot(mine && lit > 32);
            while (lm) {
                const uint32_t l = (uint32_t)__builtin_ctzll(lm);
                lm &= lm - 1;
                const uint32_t w2 = rdlane(wpos, l), r2 = rdlane(rpos, l), n2 = rdlane(lit, l);
                for (uint32_t i = 32 + lane; i < n2; i += 64) L.st[w2 + i] = L.ib[r2 + i];
            }
        }
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
        run_batch(false);
        if (stop != 2) continue;
        // ---- one big sequence, wave-serial: long literal run and / or long match, straight HBM -> HBM.
        // The window holds only the < 16 carried bytes: write them out for good, restart the window afterwards.
        flush(op + a0, true);
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
        // restart the staging window at the new output position: the partial 16-byte group below it is read back
        // so that the next aligned flush rewrites the same bytes
        wave_stores_visible();
        S0 = (op + a0) & ~15u;
        {
            const uint32_t keep = (op + a0) - S0;
            if (lane < keep && S0 + lane >= a0) L.st[lane] = ldu8(gbase + S0 + lane);
        }
        wave_sync();
        ip = x;
    }
    if (err) return err;
    if (!done) return 101;
    run_batch(true);
    if (op != out_len) return 107;
    return 0;
}

// ------------------------------------------------------------------------------------------------ encode
// Workspace: 16 KiB of LDS (the callers' uint32_t[4096]).
constexpr uint32_t LZE_HASH_BITS = 12;
constexpr uint32_t LZE_WIN = 4096;       // input window kept in LDS: src[wb, wb + wn), wn <= LZE_WIN + 80
constexpr uint32_t LZE_OUT = 3072;       // output staging
constexpr uint32_t LZE_CAP = 64;         // per-lane match extension per step; longer matches extend wave-wide
constexpr uint32_t LZE_LIT_LANE = 48;    // sequences with more literals than this are emitted wave-wide, one by one
struct Lz4EncLds {
    uint16_t tab[1u << LZE_HASH_BITS];                          // 8 KiB: position & 0xFFFF of the last occurrence
    __attribute__((aligned(16))) uint32_t win[(LZE_WIN + 96) / 4];  // ~4 KiB
    __attribute__((aligned(16))) uint8_t out[LZE_OUT + 32];     // ~3 KiB
};
static_assert(sizeof(Lz4EncLds) <= 16384, "the callers pass a 16 KiB workspace");

__device__ __forceinline__ uint32_t lz4_bound(uint32_t n) { return n + n / 255 + 16; }

// Compress src[0, n) into dst (capacity >= lz4_bound(n)); executed by ONE wave64; returns the block size.
__device__ uint32_t lz4_compress_wave_fast(const uint8_t* src, uint32_t n, uint8_t* dst, uint32_t* ws) {
    Lz4EncLds& L = *reinterpret_cast<Lz4EncLds*>(ws);
    const uint32_t lane = threadIdx.x & 63;
    uint32_t outp = 0;        // bytes already written to dst
    uint32_t on = 0;          // bytes staged in L.out
    auto flush_out = [&]() {
        wave_sync();
        for (uint32_t k = lane; k < on; k += 64) dst[outp + k] = L.out[k];
        outp += on;
        on = 0;
        wave_sync();
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
    auto emit_last = [&](uint32_t anchor) {  // the literals-only last sequence
        const uint32_t lit = n - anchor;
        flush_out();
        outp += put_head(lit, 0);
        wave_copy_g2g(dst + outp, src + anchor, lit);
        outp += lit;
    };
    if (n < 13) {  // LZ4_minLength = mflimit + 1: no match possible
        emit_last(0);
        return outp;
    }
    for (uint32_t i = lane; i < (1u << LZE_HASH_BITS); i += 64) L.tab[i] = 0xFFFF;
    const uint32_t mflimit = n - 12;      // a match may start at positions <= mflimit
    const uint32_t matchlimit = n - 5;    // and must end at or before this position
    uint32_t anchor = 0, base = 0, stride = 1;
    uint32_t wb = 0, wn = 0;              // window: L.win holds src[wb, wb + wn)
    auto load_window = [&](uint32_t from) {
        wb = from & ~15u;
        wn = min(LZE_WIN + 80u, n - wb);
        wave_sync();
        uint8_t* w8 = (uint8_t*)L.win;
        for (uint32_t k = lane * 16; k < wn; k += 64 * 16) {
            if (wb + k + 16 <= n) *(u32x4*)(w8 + k) = ldu128(src + wb + k);
            else for (uint32_t b = 0; b < 16; b++) w8[k + b] = wb + k + b < n ? ldu8(src + wb + k + b) : (uint8_t)0;
        }
        wave_sync();
    };
    load_window(0);
    // 4 bytes at position x: from the window (two aligned dwords + v_alignbyte) when inside, else from HBM
    auto win4 = [&](uint32_t x) -> uint32_t {
        const uint32_t y = x - wb;
        return __builtin_amdgcn_alignbyte(L.win[(y >> 2) + 1], L.win[y >> 2], y & 3);
    };
    auto in_win = [&](uint32_t x, uint32_t len) -> bool { return x >= wb && x + len + 4 <= wb + wn; };  // (+4: the second dword)
    auto ld4 = [&](uint32_t x) -> uint32_t { return in_win(x, 4) ? win4(x) : ldu32(src + x); };

    while (base <= mflimit) {
        {   // the step's positions and LZE_CAP + 16 bytes of look-ahead must be in the window
            const uint32_t need = base + 63 * stride + LZE_CAP + 24;
            if (need > wb + wn && wb + wn < n) load_window(base >= 1024 ? base - 1024 : 0);  // keep 1 KiB of history
        }
        const uint32_t p = base + lane * stride;
        const bool act = p <= mflimit && in_win(p, 4);   // (a step wider than the window probes only its front part)
        uint32_t cand = 0, mlen = 0;
        if (act) {
            const uint32_t v4 = win4(p);
            const uint32_t h = (v4 * 2654435761u) >> (32 - LZE_HASH_BITS);
            const uint32_t e = L.tab[h];
            L.tab[h] = (uint16_t)(p & 0xFFFF);
            // candidate from the table: the last position with this hash (mod 64 Ki), if inside the 64 KiB reach
            uint32_t c = (p & ~0xFFFFu) | e;
            if (c >= p) c -= 0x10000u;                       // (wraps above p when there is no such position)
            bool ok = e != 0xFFFF && c < p && p - c <= 65535u;
            uint32_t have = 0;                               // matching bytes known so far
            if (ok) {
                if (in_win(c, 16) && in_win(p, 16)) {
                    ok = win4(c) == v4;
                    if (ok) have = 4;
                } else if (c + 16 <= n && in_win(p, 16)) {  // one round trip: 16 candidate bytes
                    const u32x4 cv = ldu128(src + c);
                    ok = cv.x == v4;
                    if (ok) {
                        have = 4;
                        uint32_t d;
                        if ((d = cv.y ^ win4(p + 4)) != 0) have = 4 + (__builtin_ctz(d) >> 3);
                        else if ((d = cv.z ^ win4(p + 8)) != 0) have = 8 + (__builtin_ctz(d) >> 3);
                        else if ((d = cv.w ^ win4(p + 12)) != 0) have = 12 + (__builtin_ctz(d) >> 3);
                        else have = 16;
                        if (have < 16) have |= 0x80000000u;  // mismatch found: final
                    }
                } else {
                    ok = ldu32(src + c) == v4;
                    if (ok) have = 4;
                }
            }
            if (!ok && stride == 1) {   // the periods columnar data repeats with (positions of this very step are not in
                                        // the table yet): 8, 4, 2, 1 bytes back
                if (p >= wb + 8 && win4(p - 8) == v4) { c = p - 8; ok = true; }
                else if (p >= wb + 4 && win4(p - 4) == v4) { c = p - 4; ok = true; }
                else if (p >= wb + 2 && win4(p - 2) == v4) { c = p - 2; ok = true; }
                else if (p >= wb + 1 && win4(p - 1) == v4) { c = p - 1; ok = true; }
                if (ok) have = 4;
            }
            if (ok) {
                cand = c;
                const uint32_t room = matchlimit - p;        // longest match allowed here (>= 7)
                bool final = (have & 0x80000000u) != 0;
                mlen = have & 0x7FFFFFFFu;
                while (!final && mlen < LZE_CAP && mlen + 4 <= room) {
                    const uint32_t d = ld4(c + mlen) ^ ld4(p + mlen);
                    if (d) { mlen += __builtin_ctz(d) >> 3; final = true; } else mlen += 4;
                }
                if (!final && mlen < LZE_CAP) {              // fewer than 4 bytes of room left: byte by byte
                    while (mlen < room && ldu8(src + c + mlen) == ldu8(src + p + mlen)) mlen++;
                }
                if (mlen > room) mlen = room;
            }
        }
        const uint64_t mm = __ballot(mlen >= 4);
        if (!mm) {
            base += 64 * stride;
            stride++;
            continue;
        }
        // ---- greedy left-to-right selection of non-overlapping matches of this step (wave-uniform walk).  Chosen lanes
        // get the start of their literals, the offset of their sequence in the staging buffer and its size.
        uint64_t C = 0, big = 0;
        uint32_t covered = anchor;            // everything below is emitted or pending as literals of the next sequence
        uint32_t lit_start_v = 0, out_off_v = 0, run = 0;
        uint32_t first_lane = 0;
        for (;;) {
            // lanes whose position is >= covered form a suffix: first lane = ceil((covered - base) / stride)
            first_lane = covered > base ? (covered - base + stride - 1) / stride : 0;
            if (first_lane >= 64) break;
            const uint64_t m = (mm >> first_lane) << first_lane;
            if (!m) break;
            const uint32_t l = (uint32_t)__builtin_ctzll(m);
            const uint32_t pl = base + l * stride;
            uint32_t ml_l = rdlane(mlen, l);
            if (ml_l >= LZE_CAP) {   // extend wave-wide, 64 x 4 bytes per step
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
            }
            const uint32_t lit = pl - covered, mcode = ml_l - 4;
            const uint32_t sz = 1 + (lit >= 15 ? 1 + div255(lit - 15) : 0) + lit + 2 + (mcode >= 15 ? 1 + div255(mcode - 15) : 0);
            C |= 1ull << l;
            if (lit > LZE_LIT_LANE) big |= 1ull << l;
            lit_start_v = wrlane(covered, l, lit_start_v);
            out_off_v = wrlane(run, l, out_off_v);
            run += sz;
            covered = pl + ml_l;
        }
        const bool chosen = (C >> lane) & 1;
        if (!big && run <= LZE_OUT) {
            // ---- every chosen lane writes its own sequence into the staging buffer
            if (on + run > LZE_OUT) flush_out();
            if (chosen) {
                const uint32_t lit = p - lit_start_v, mcode = mlen - 4, off = p - cand;
                uint8_t* o = L.out + on + out_off_v;
                uint32_t k = 0;
                o[k++] = (uint8_t)((min(lit, 15u) << 4) | min(mcode, 15u));
                if (lit >= 15) {
                    uint32_t r = lit - 15;
                    while (r >= 255) { o[k++] = 255; r -= 255; }
                    o[k++] = (uint8_t)r;
                }
                const uint8_t* w8 = (const uint8_t*)L.win;
                for (uint32_t i = 0; i < lit; i++) {
                    const uint32_t x = lit_start_v + i;
                    o[k++] = (x >= wb && x < wb + wn) ? w8[x - wb] : ldu8(src + x);
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
        } else {
            // ---- one sequence at a time, straight to dst (long literal runs)
            flush_out();
            uint64_t m = C;
            while (m) {
                const uint32_t l = (uint32_t)__builtin_ctzll(m);
                m &= m - 1;
                const uint32_t pl = base + l * stride;
                const uint32_t s_ls = rdlane(lit_start_v, l), s_mc = rdlane(mlen, l) - 4, s_off = pl - rdlane(cand, l);
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
        anchor = covered;
        base = max(covered, base + 1);
        stride = 1;
    }
    emit_last(anchor);
    return outp;
}

}  // namespace sb
