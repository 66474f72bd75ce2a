// sb_lz4_big.h — LZ4 blocks of 64 KiB of compressed bytes and more: one WORKGROUP per block, no serial walk.
//
// A page is ONE LZ4 block per buffer in the format (src/compression/basic.rs:108-121: lz4::block::decompress_to_buffer
// over the whole buffer), so a 1 MiB page of text is ~100 000 sequences that depend on each other twice: where a
// sequence starts depends on all sequences before it, and what a match copies may have been produced by the match
// right before it.  sb_lz4.h's one-wave decoder walks both chains in order (≈ 45 cycles per output byte alone on a
// SIMD).  Here both chains are resolved by pointer jumping in a few data-parallel rounds:
//
//   parse   2 KiB of compressed bytes at a time: every byte position computes where the sequence behind the one that
//           WOULD start there begins (nxt[i]), then where the chain from i leaves i's 64-byte segment (a few hops
//           over nxt) and i's 512-byte segment (a few hops over the 64-byte exits) — all positions at once, loads
//           only.  The positions really on the chain follow top down from the chunk's entry point: <= 8 hops over
//           the 512-byte exits, then 8 threads hop over the 64-byte exits, then 64 threads walk their segment.
//           Marked positions become records (literal source, literal length, match length, distance) with output
//           positions from a workgroup scan.
//   copy    8 KiB of output at a time: one u16 entry per output byte, either FINAL (0x8000 | byte: a literal, or a
//           match byte whose source lies in front of the window and is read back from HBM) or a POINTER to the
//           window byte it copies.  A round is ent[p] = ent[ent[p]] for every pointer entry — a final source hands over
//           its byte, a pointer source its own pointer — so chains of matches (and overlapping matches, distance <
//           length) are done after log2(depth) rounds.  The window is then packed and stored with aligned 16 B stores.
//
// Sequences whose header does not fit the staged input (literal runs of more than ~280 bytes) end a chunk and are
// parsed from HBM by one thread as the next chunk's only record; their bytes flow through the same windows.
// Every read of the input is bounded by the block, every output position by out_len; a block that does not end in a
// literals-only sequence exactly at its last byte, or does not produce exactly out_len bytes, is refused.
#pragma once
#include "sb_lz4.h"

#if defined(SB_LZ4_BIG_PROFILE)
// phase clocks of thread 0 of workgroup 0 (scripts/micro/lz4_big_probe.hip), kept in registers and written at the end
#define LBP_BEGIN unsigned long long lbp_acc[24] = {0}; unsigned long long lbp_t0 = __builtin_readcyclecounter();
#define LBP(i) do { const unsigned long long n_ = __builtin_readcyclecounter(); lbp_acc[i] += n_ - lbp_t0; lbp_t0 = n_; } while (0)
#define LBP_CNT(i, v) do { lbp_acc[i] += (v); } while (0)
#define LBP_END do { if (blockIdx.x == 0 && threadIdx.x == 0) for (int i_ = 0; i_ < 24; i_++) g_prof_big[i_] += lbp_acc[i_]; } while (0)
#else
#define LBP_BEGIN
#define LBP(i)
#define LBP_CNT(i, v)
#define LBP_END
#endif

namespace sb {

constexpr uint32_t LB_T = 256;                 // threads of the workgroup
constexpr uint32_t LB_CH = 2048;               // compressed bytes parsed per round
constexpr uint32_t LB_S = LB_CH + 288;         // staged input: a sequence starting in the chunk with <= ~280 header + literal bytes is complete
constexpr uint32_t LB_REC = 512;               // sequence records per round
constexpr uint32_t LB_WIN = 8192;              // output window (bytes)
constexpr uint32_t LB_PPT = LB_CH / LB_T;     // chunk positions per thread (8: the rank step reads them as two words)
constexpr uint32_t LB_RPT = LB_REC / LB_T;    // records per thread
constexpr uint32_t LB_BPT = LB_WIN / LB_T;     // window bytes per thread (<= 32: one pending bit each)
constexpr uint32_t LB_SEG_A = 64, LB_SEG_B = 512;   // the two levels of segments of the chunk
static_assert(LB_PPT == 8 && LB_BPT <= 32 && LB_CH % LB_SEG_B == 0, "sb_lz4_big.h: geometry");

struct Lz4BigLds {
    union {
        struct {
            __attribute__((aligned(16))) uint8_t mark[LB_CH + 16];
            uint16_t nxt[LB_CH + 2];    // start of the sequence behind the one that would start at i (LB_CH: outside the chunk)
            uint16_t ex_a[LB_CH + 2];   // first position of the chain from i outside i's 64-byte segment
            uint16_t ex_b[LB_CH + 2];   // ... outside i's 512-byte segment
            uint16_t ent_a[LB_CH / LB_SEG_A], ent_b[LB_CH / LB_SEG_B];   // where the chain from the entry point enters each segment (0xFFFF: it does not)
        } p;
        __attribute__((aligned(16))) uint16_t ent[LB_WIN + 16];
    };
    __attribute__((aligned(16))) uint8_t in[LB_S + 16];   // the chunk's compressed bytes (literals of the windows come from here)
    uint32_t r_out[LB_REC + 4];   // output position of the record's first byte; [nrec] = end of the round's output
    uint32_t r_lit[LB_REC];       // input position of its literals
    uint32_t r_off[LB_REC];       // match distance (directly behind r_lit: the entries phase picks one of the two with one load)
    uint32_t r_ll[LB_REC];        // literal length (the rest of the record's bytes are the match)
    uint32_t wsum[8];
    uint64_t wsum64[4];
    uint32_t nrec, next, final_, err, total;
};

struct LbSeq {
    uint32_t ll, lit, ml, off, end;
    uint32_t kind;   // 0 sequence with a match, 1 last sequence (literals up to the end of the block), 2 needs bytes beyond `lim`, 3 malformed
};

// the sequence whose token is byte i of a reader that holds bytes [0, lim) of the `room` bytes up to the block's end
template <class RD>
__device__ __forceinline__ LbSeq lb_seq(RD rd, uint32_t i, uint32_t lim, uint32_t room) {
    LbSeq s;
    s.ll = s.lit = s.ml = s.off = s.end = 0;
    s.kind = 0;
    const uint32_t t = rd(i);
    uint32_t j = i + 1, ll = t >> 4;
    if (ll == 15) {
        uint32_t b;
        do {
            if (j >= lim) {
                s.kind = j >= room ? 3u : 2u;
                return s;
            }
            b = rd(j++);
            ll += b;
            if (ll > room) {
                s.kind = 3;
                return s;
            }
        } while (b == 255);
    }
    s.ll = ll;
    s.lit = j;
    if (ll > room - j) {   // (j <= lim <= room)
        s.kind = 3;
        return s;
    }
    j += ll;
    if (j == room) {
        s.kind = j > lim ? 2u : 1u;   // (the literals of a complete sequence lie inside the reader's bytes)
        s.end = j;
        return s;
    }
    if (j + 2 > room) {
        s.kind = 3;
        return s;
    }
    if (j + 2 > lim) {
        s.kind = 2;
        return s;
    }
    s.off = rd(j) | (rd(j + 1) << 8);
    j += 2;
    uint32_t ml = t & 15;
    if (ml == 15) {
        uint32_t b;
        do {
            if (j >= lim) {
                s.kind = j >= room ? 3u : 2u;
                return s;
            }
            b = rd(j++);
            ml += b;
            if (ml > 0xFFFFFF00u - 4) {
                s.kind = 3;
                return s;
            }
        } while (b == 255);
    }
    s.ml = ml + 4;
    if (s.off == 0) s.kind = 3;
    s.end = j;
    return s;
}

// The same by a whole WAVE (all 64 lanes call it with the same arguments), for headers with thousands of length bytes: a
// literal run of megabytes has one 255 per 255 bytes, and one lane reading them from HBM byte by byte is a chain of
// dependent loads (4 ms per megabyte of literals).  Every lane looks at 16 bytes, a ballot finds the first byte that is
// not 255.  `src` points at the token, `room` = bytes up to the block's end.
__device__ inline LbSeq lb_seq_wave(const uint8_t* src, uint32_t room) {
    const uint32_t lane = threadIdx.x & 63;
    LbSeq s;
    s.ll = s.lit = s.ml = s.off = s.end = 0;
    s.kind = 3;
    // length bytes from position j on: returns the sum, moves j behind the last one; false: they run into the block's end
    auto ext = [&](uint32_t& j, uint32_t& acc) -> bool {
        for (;;) {
            const uint32_t p = j + 16 * lane;
            uint32_t lead = 0;   // leading 255s among my 16 bytes (bytes beyond the block count as "not 255")
            if (p + 16 <= room) {
                const u32x4 v = ldu128(src + p);
                const uint64_t a = ~((uint64_t)v.x | ((uint64_t)v.y << 32)), b = ~((uint64_t)v.z | ((uint64_t)v.w << 32));
                lead = a ? (uint32_t)__builtin_ctzll(a) / 8 : (b ? 8 + (uint32_t)__builtin_ctzll(b) / 8 : 16u);
            } else {
                while (lead < 16 && p + lead < room && ldu8(src + p + lead) == 255) lead++;
            }
            const uint64_t full = __ballot(lead == 16);
            if (full == ~0ull) {
                if (acc > 0xF0000000u) return false;
                acc += 255u * 1024u;
                j += 1024;
                continue;
            }
            const uint32_t f = (uint32_t)__builtin_ctzll(~full);
            const uint32_t lf = rdlane(lead, f);
            const uint32_t n255 = 16 * f + lf;
            const uint32_t at = j + n255;   // the byte that ends the run
            if (at >= room) return false;
            if (acc > 0xF0000000u) return false;
            acc += 255u * n255 + (uint32_t)ldu8(src + at);
            j = at + 1;
            return true;
        }
    };
    if (room == 0) return s;
    const uint32_t t = ldu8(src);
    uint32_t j = 1, ll = t >> 4;
    if (ll == 15 && !ext(j, ll)) return s;
    s.ll = ll;
    s.lit = j;
    if (ll > room - j) return s;
    j += ll;
    if (j == room) {
        s.kind = 1;
        s.end = j;
        return s;
    }
    if (j + 2 > room) return s;
    s.off = (uint32_t)ldu8(src + j) | ((uint32_t)ldu8(src + j + 1) << 8);
    j += 2;
    uint32_t ml = t & 15;
    if (ml == 15 && !ext(j, ml)) return s;
    if (ml > 0xFFFFFF00u - 4) return s;
    s.ml = ml + 4;
    s.end = j;
    s.kind = s.off ? 0u : 3u;
    return s;
}

// The common shape of a sequence header, without loops: at most one extension byte per length, everything up to the
// byte behind the match-length extension inside the staged input.  tok = in[i], b1 = in[i + 1]; `e` = position of the first
// byte behind the offset.  Returns false when the general parser has to look (255 extension bytes, the last sequence,
// a header near the end of the staged bytes).
__device__ __forceinline__ uint32_t lb_hdr_e(uint32_t i, uint32_t tok, uint32_t b1) {
    const uint32_t l4 = tok >> 4;
    return l4 != 15 ? i + 3 + l4 : i + 4 + 15 + b1;
}
__device__ __forceinline__ bool lb_hdr_simple(uint32_t tok, uint32_t b1, uint32_t ext, uint32_t e, uint32_t sl) {
    return ((tok >> 4) != 15 || b1 != 255) && ((tok & 15) != 15 || ext != 255) && e < sl;
}

// wave64 inclusive max-scan with DPP (same recipe as wave_scan_dpp; identity 0)
__device__ __forceinline__ uint32_t wave_scan_max_dpp(uint32_t v) {
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false));
    return v;
}

// One block by the workgroup (LB_T threads).  Returns 0 or an error code (uniform).
// GIANT (sb_lz4_giant.h): the workgroup takes the sequences that START in [c_begin, c_end) of a block whose chain of
// sequence starts has been found block-wide — c_begin is on the chain, op_begin the output position of its sequence — and
// writes one u32 ENTRY per output byte instead of bytes: 0x80000000 | byte, or the absolute output position the byte
// copies when that lies in front of the window (the windows of other workgroups are not written yet: the entries are
// resolved by pointer jumping over the whole block afterwards).  In the window's LDS entries such a byte is a ROOT like
// a literal: 0x4000 | its window position.
constexpr uint32_t LB_ROOT = 0x4000u;
template <bool GIANT = false>
__device__ uint32_t lz4_inflate_block_wg(const uint8_t* src, uint32_t n, uint8_t* dst, uint32_t out_len, Lz4BigLds& L, uint32_t c_begin = 0,
                                         uint32_t c_end = 0, uint32_t op_begin = 0, uint32_t* ent32 = nullptr, uint32_t* lits = nullptr,
                                         uint32_t* nlits = nullptr) {
    const uint32_t t = threadIdx.x, lane = t & 63, wv = t >> 6;
    if (n == 0) return out_len != 0 ? 100u : 0u;
    uint32_t c0 = c_begin;   // input position of the next sequence
    uint32_t op = op_begin;  // output bytes produced
    if (t == 0) L.err = 0;
    LBP_BEGIN
    for (;;) {
        LBP_CNT(16, 1);
        // ------------------------------------------------------------------ parse: the chunk that starts at c0
        const uint32_t room = n - c0;
        const uint32_t sl = min(LB_S, room);
        const uint32_t npos = GIANT ? min(min(LB_CH, room), c_end - c0) : min(LB_CH, room);   // (GIANT: sequences from c_end on are the next workgroup's)
        __syncthreads();   // (the window entries of the round before are dead)
        for (uint32_t k = t * 16; k < sl; k += LB_T * 16) {
            if (c0 + k + 16 <= n) {
                *(u32x4*)(L.in + k) = ldu128(src + c0 + k);
            } else {
                for (uint32_t b = 0; b < 16; b++) L.in[k + b] = c0 + k + b < n ? ldu8(src + c0 + k + b) : (uint8_t)0;
            }
        }
        if (t == 0) {
            L.nrec = 0;
            L.final_ = 0;
            L.next = c0;
        }
        __syncthreads();
        LBP(0);
        auto rd_lds = [&](uint32_t i) -> uint32_t { return L.in[i]; };
        auto rd_mix = [&](uint32_t x) -> uint32_t { return x < sl ? (uint32_t)L.in[x] : (uint32_t)ldu8(src + c0 + x); };
        // An entry sequence with 270 literals and more is a round of its own, parsed by one thread (the positions inside its
        // length bytes and literals are not looked at: in a run of 255s every one of them would walk the whole run).
        const bool lone = sl > 1 && (L.in[0] >> 4) == 15 && L.in[1] == 255;   // (uniform)
        if (lone) {
            if (wv == 0) {   // (its length bytes may be thousands: the wave reads them 1 KiB at a time)
                const LbSeq sq = lb_seq_wave(src + c0, room);
                if (t != 0) {
                } else if (sq.kind == 3) {
                    L.err = 101;
                } else {
                    L.r_lit[0] = c0 + sq.lit;
                    L.r_ll[0] = sq.ll;
                    L.r_off[0] = sq.off;
                    L.r_out[0] = sq.ll + sq.ml;
                    if (sq.ll > out_len || sq.ml > out_len) L.err = 102;
                    L.nrec = 1;
                    L.next = c0 + sq.end;
                    if (sq.kind == 1) L.final_ = 1;
                }
            }
        } else {
        uint32_t nx[LB_PPT];
        {   // where would the next sequence start?  (offset and literal bytes are checked when the position turns out to
            // be on the chain)
            uint32_t tok[LB_PPT], b1[LB_PPT], ext[LB_PPT], slow = 0;
#pragma unroll
            for (uint32_t k = 0; k < LB_PPT; k++) {
                tok[k] = L.in[t + LB_T * k];
                b1[k] = L.in[t + LB_T * k + 1];
            }
#pragma unroll
            for (uint32_t k = 0; k < LB_PPT; k++) ext[k] = L.in[min(lb_hdr_e(t + LB_T * k, tok[k], b1[k]), LB_S - 1)];
#pragma unroll
            for (uint32_t k = 0; k < LB_PPT; k++) {
                const uint32_t i = t + LB_T * k;
                nx[k] = LB_CH;
                if (i < npos) {
                    const uint32_t e = lb_hdr_e(i, tok[k], b1[k]);
                    if (lb_hdr_simple(tok[k], b1[k], ext[k], e, sl)) {   // (e < sl <= room: the offset lies inside the block)
                        const uint32_t end = e + ((tok[k] & 15) == 15 ? 1u : 0u);
                        if (end < npos) nx[k] = end;
                    } else {
                        slow |= 1u << k;
                    }
                }
            }
            while (slow) {   // (rare: 255 extension bytes, headers near the end of the staged bytes)
                const uint32_t k = (uint32_t)__builtin_ctz(slow);
                slow &= slow - 1;
                const LbSeq q = lb_seq(rd_lds, t + LB_T * k, sl, room);
                const uint32_t v = q.kind == 0 && q.end < npos ? q.end : LB_CH;
#pragma unroll
                for (uint32_t kk = 0; kk < LB_PPT; kk++)
                    if (kk == k) nx[kk] = v;
            }
#pragma unroll
            for (uint32_t k = 0; k < LB_PPT; k++) L.p.nxt[t + LB_T * k] = (uint16_t)nx[k];
            *(uint64_t*)(L.p.mark + 8 * t) = 0;
        }
        if (t == 0) L.p.nxt[LB_CH] = (uint16_t)LB_CH;
        if (t < LB_CH / LB_SEG_A) L.p.ent_a[t] = 0xFFFF;
        if (t < LB_CH / LB_SEG_B) L.p.ent_b[t] = 0xFFFF;
        __syncthreads();
        LBP(1);
        // exits: the first position at or behind the end of its 64-byte segment that the chain from i reaches, then the
        // same over 512-byte segments by hopping over the 64-byte exits (16 independent walks per thread, loads only)
        {
            uint32_t j[LB_PPT];
#pragma unroll
            for (uint32_t k = 0; k < LB_PPT; k++) j[k] = nx[k];
            for (bool go = true; go;) {
                go = false;
                uint32_t v[LB_PPT];
#pragma unroll
                for (uint32_t k = 0; k < LB_PPT; k++) v[k] = L.p.nxt[j[k]];
#pragma unroll
                for (uint32_t k = 0; k < LB_PPT; k++) {
                    if (j[k] < ((t + LB_T * k) | (LB_SEG_A - 1)) + 1) {
                        j[k] = v[k];
                        go = true;
                    }
                }
            }
#pragma unroll
            for (uint32_t k = 0; k < LB_PPT; k++) L.p.ex_a[t + LB_T * k] = (uint16_t)j[k];
            if (t == 0) L.p.ex_a[LB_CH] = (uint16_t)LB_CH;
            __syncthreads();
            for (bool go = true; go;) {
                go = false;
                uint32_t v[LB_PPT];
#pragma unroll
                for (uint32_t k = 0; k < LB_PPT; k++) v[k] = L.p.ex_a[j[k]];
#pragma unroll
                for (uint32_t k = 0; k < LB_PPT; k++) {
                    if (j[k] < ((t + LB_T * k) | (LB_SEG_B - 1)) + 1) {
                        j[k] = v[k];
                        go = true;
                    }
                }
            }
#pragma unroll
            for (uint32_t k = 0; k < LB_PPT; k++) L.p.ex_b[t + LB_T * k] = (uint16_t)j[k];
        }
        __syncthreads();
        // the chain from the entry point, top down: entry of every 512-byte segment, of every 64-byte segment, every position
        if (t == 0) {
            for (uint32_t p = 0; p < LB_CH; p = L.p.ex_b[p]) L.p.ent_b[p / LB_SEG_B] = (uint16_t)p;
        }
        __syncthreads();
        if (t < LB_CH / LB_SEG_B) {
            uint32_t p = L.p.ent_b[t];
            if (p != 0xFFFF)
                for (; p < LB_SEG_B * (t + 1); p = L.p.ex_a[p]) L.p.ent_a[p / LB_SEG_A] = (uint16_t)p;
        }
        __syncthreads();
        if (t < LB_CH / LB_SEG_A) {
            uint32_t p = L.p.ent_a[t];
            if (p != 0xFFFF)
                for (; p < LB_SEG_A * (t + 1); p = L.p.nxt[p]) L.p.mark[p] = 1;
        }
        __syncthreads();
        LBP(2);
        // ranks of the marked positions (thread = 8 consecutive positions)
        const uint64_t mk = *(const uint64_t*)(L.p.mark + 8 * t);
        const uint32_t cnt = (uint32_t)__popcll(mk);
        const uint32_t incl = wave_scan_dpp(cnt);
        if (lane == 63) L.wsum[wv] = incl;
        __syncthreads();
        uint32_t rank = incl - cnt;
        for (uint32_t q = 0; q < wv; q++) rank += L.wsum[q];
        const uint32_t total = L.wsum[0] + L.wsum[1] + L.wsum[2] + L.wsum[3];
        const uint32_t last_rank = min(total, LB_REC) - 1;   // (total >= 1: the entry point is marked)
        if (cnt) {   // the list of marked positions (ex_b is dead)
            for (uint32_t b = 0; b < 8; b++) {
                if (!((mk >> (8 * b)) & 1)) continue;
                const uint32_t r = rank++;
                if (r <= LB_REC) L.p.ex_b[r] = (uint16_t)(8 * t + b);
            }
        }
        __syncthreads();
        {   // records: thread = ranks t, t + 256, ... (the loads of its four headers side by side)
            uint32_t pi[LB_RPT], tok[LB_RPT], b1[LB_RPT], ext[LB_RPT], o0[LB_RPT], o1[LB_RPT];
#pragma unroll
            for (uint32_t q = 0; q < LB_RPT; q++) pi[q] = t + LB_T * q <= last_rank ? (uint32_t)L.p.ex_b[t + LB_T * q] : 0u;
#pragma unroll
            for (uint32_t q = 0; q < LB_RPT; q++) {
                tok[q] = L.in[pi[q]];
                b1[q] = L.in[pi[q] + 1];
            }
#pragma unroll
            for (uint32_t q = 0; q < LB_RPT; q++) {
                const uint32_t e = min(lb_hdr_e(pi[q], tok[q], b1[q]), LB_S - 1);
                ext[q] = L.in[e];
                o0[q] = L.in[e - 2];
                o1[q] = L.in[e - 1];
            }
#pragma unroll
            for (uint32_t q = 0; q < LB_RPT; q++) {
                const uint32_t r = t + LB_T * q, i = pi[q];
                if (r > last_rank) continue;
                LbSeq sq;
                const uint32_t e = lb_hdr_e(i, tok[q], b1[q]);
                if (lb_hdr_simple(tok[q], b1[q], ext[q], e, sl)) {
                    const uint32_t l4 = tok[q] >> 4, m4 = tok[q] & 15;
                    sq.ll = l4 != 15 ? l4 : 15 + b1[q];
                    sq.lit = e - 2 - sq.ll;
                    sq.off = o0[q] | (o1[q] << 8);
                    sq.ml = 4 + m4 + (m4 == 15 ? ext[q] : 0u);
                    sq.end = e + (m4 == 15 ? 1u : 0u);
                    sq.kind = sq.off ? 0u : 3u;
                } else {
                    sq = lb_seq(rd_lds, i, sl, room);
                    if (sq.kind == 2) {
                        if (r != 0) {   // (the last mark) leave it to the next round, where it is the entry point
                            L.nrec = r;
                            L.next = c0 + i;
                            continue;
                        }
                        sq = lb_seq(rd_mix, i, room, room);
                    }
                }
                if (sq.kind == 3) {
                    L.err = 101;
                    continue;
                }
                L.r_lit[r] = c0 + sq.lit;
                L.r_ll[r] = sq.ll;
                L.r_off[r] = sq.off;
                L.r_out[r] = sq.ll + sq.ml;
                if (sq.ll > out_len || sq.ml > out_len) L.err = 102;
                if (r == last_rank) {
                    L.nrec = r + 1;
                    L.next = total > LB_REC ? c0 + (uint32_t)L.p.ex_b[LB_REC] : c0 + sq.end;   // more sequences than records: the next round starts at the next mark
                    if (sq.kind == 1) L.final_ = 1;
                }
            }
        }
        }
        __syncthreads();
        if (L.err) return L.err;
        if (lone) {   // its literals go straight from the input to the output; what is left of the record is its match
            const uint32_t ll = L.r_ll[0];
            if (ll > out_len - op) return 103;
            const uint8_t* g = src + L.r_lit[0];
            if constexpr (GIANT) {
                // a run of tens of kilobytes and more is listed for k_lzg_lits (every workgroup of the chip copies a share:
                // one workgroup moving a 5 MB run entry by entry took 17 ms); shorter ones 16 bytes per thread and step
                __syncthreads();
                if (t == 0) {
                    L.total = 0;
                    if (ll >= LZG_LIT_MIN && lits) {
                        const uint32_t at = atomicAdd(nlits, 1u);
                        if (at < LZG_LITS) {
                            lits[4 * at] = L.r_lit[0];
                            lits[4 * at + 1] = op;
                            lits[4 * at + 2] = ll;
                            L.total = 1;
                        }
                    }
                }
                __syncthreads();
                if (!L.total) {
                    for (uint32_t k = t * 16; k < ll; k += LB_T * 16) {
                        if (k + 16 <= ll) {
                            const u32x4 v = ldu128(g + k);
                            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                            for (int q = 0; q < 4; q++) {
                                u32x4 o;
                                o.x = 0x80000000u | (w[q] & 0xFFu);
                                o.y = 0x80000000u | ((w[q] >> 8) & 0xFFu);
                                o.z = 0x80000000u | ((w[q] >> 16) & 0xFFu);
                                o.w = 0x80000000u | (w[q] >> 24);
                                __builtin_memcpy(ent32 + (uint64_t)op + k + 4 * q, &o, 16);
                            }
                        } else {
                            for (uint32_t b = k; b < ll; b++) ent32[(uint64_t)op + b] = 0x80000000u | (uint32_t)ldu8(g + b);
                        }
                    }
                }
            } else {
            uint8_t* d = dst + op;
            uint32_t head = (uint32_t)((16 - ((uintptr_t)d & 15)) & 15);
            if (head > ll) head = ll;
            if (t < head) d[t] = ldu8(g + t);
            const uint32_t nvec = (ll - head) >> 4;
            for (uint32_t k = t; k < nvec; k += LB_T) stu128(d + head + 16 * (uint64_t)k, ldu128(g + head + 16 * (uint64_t)k));
            const uint32_t done = head + 16 * nvec;
            if (t < ll - done) d[done + t] = ldu8(g + done + t);
            }
            wave_stores_visible();
            op += ll;
            __syncthreads();
            if (t == 0) {
                L.r_ll[0] = 0;
                L.r_out[0] -= ll;
            }
            __syncthreads();
        }
        LBP(3);
        const uint32_t nrec = L.nrec;
        LBP_CNT(19, nrec);
        // output positions: exclusive scan of the record lengths (thread = LB_RPT consecutive records)
        uint32_t len4[LB_RPT];
        uint64_t sum = 0;
#pragma unroll
        for (uint32_t q = 0; q < LB_RPT; q++) {
            const uint32_t r = LB_RPT * t + q;
            len4[q] = r < nrec ? L.r_out[r] : 0;
            sum += len4[q];
        }
        const uint64_t incl64 = wave_incl_scan64(sum);
        if (lane == 63) L.wsum64[wv] = incl64;
        __syncthreads();
        uint64_t base = incl64 - sum;
        for (uint32_t q = 0; q < wv; q++) base += L.wsum64[q];
        const uint64_t round_out = L.wsum64[0] + L.wsum64[1] + L.wsum64[2] + L.wsum64[3];
        if (round_out > (uint64_t)(out_len - op)) return 103;   // (uniform) more output than the caller expects
        {
            uint32_t o = op + (uint32_t)base;
#pragma unroll
            for (uint32_t q = 0; q < LB_RPT; q++) {
                const uint32_t r = LB_RPT * t + q;
                if (r < nrec) {
                    L.r_out[r] = o;
                    // a match may not reach in front of the block
                    if (len4[q] > L.r_ll[r] && (uint32_t)L.r_off[r] > o + L.r_ll[r]) L.err = 104;
                }
                o += len4[q];
            }
        }
        const uint32_t o_end = op + (uint32_t)round_out;
        if (t == 0) L.r_out[nrec] = o_end;
        __syncthreads();
        if (L.err) return L.err;
        const uint32_t next = L.next, fin = L.final_;
        LBP(4);

        // ------------------------------------------------------------------ copy: windows of the round's output
        const uint32_t nwin = (o_end - op + LB_WIN - 1) / LB_WIN;   // windows of equal size
        const uint32_t wstep = nwin ? min(LB_WIN, ((o_end - op + nwin - 1) / nwin + 255) & ~255u) : LB_WIN;
        for (uint32_t w0 = op; w0 < o_end; w0 += wstep) {
            const uint32_t wl = min(wstep, o_end - w0);
            LBP_CNT(18, 1);
            __syncthreads();
            for (uint32_t k = t * 8; k < wl; k += LB_T * 8) *(u32x4*)(L.ent + k) = u32x4{0, 0, 0, 0};
            __syncthreads();
            for (uint32_t r = t; r < nrec; r += LB_T) {
                const uint32_t o = L.r_out[r];
                if (o >= w0 && o - w0 < wl && L.r_out[r + 1] > o) L.ent[o - w0] = (uint16_t)(r + 1);
            }
            __syncthreads();
            LBP(5);
            // every byte finds its record: max-scan of the markers (wave = a quarter of the window), record + 1 left in ent
            {
                const uint32_t wq = ((wl + 255) / 256) * 64;   // bytes per wave, a multiple of 64
                const uint32_t b0 = wv * wq;
                uint32_t carry = 0;   // record + 1 that covers the byte in front of the row
                if (b0 < wl) {
                    uint32_t lo = 0, hi = nrec;   // largest r with r_out[r] <= w0 + b0 (r_out[0] <= w0)
                    const uint32_t x = w0 + b0;
                    while (hi - lo > 1) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if (L.r_out[mid] <= x)
                            lo = mid;
                        else
                            hi = mid;
                    }
                    carry = lo + 1;
                }
                const uint32_t b1 = min(b0 + wq, wl);
                for (uint32_t rb = b0; rb < b1; rb += 512) {
                    uint32_t m[8];
#pragma unroll
                    for (uint32_t u = 0; u < 8; u++) {
                        const uint32_t p = rb + 64 * u + lane;
                        m[u] = p < b1 ? (uint32_t)L.ent[p] : 0u;
                    }
#pragma unroll
                    for (uint32_t u = 0; u < 8; u++) {
                        const uint32_t p = rb + 64 * u + lane;
                        const uint32_t sc = max(wave_scan_max_dpp(m[u]), carry);
                        carry = rdlane(sc, 63);
                        if (p < b1) L.ent[p] = (uint16_t)sc;
                    }
                }
            }
            __syncthreads();
            LBP(6);
            // first entries (thread owns the bytes t + 256 k, eight at a time, the loads of a level side by side): a
            // literal, a byte from in front of the window, or a pointer
            uint32_t pend = 0;
            uint32_t gmask = 0;      // match bytes of the thread whose source lies in front of the window (HBM): second pass
#pragma unroll 1
            for (uint32_t k0 = 0; k0 * LB_T < wl; k0 += 8) {
                uint32_t id[8], kk[8], x[8];
#pragma unroll
                for (uint32_t u = 0; u < 8; u++) {
                    const uint32_t p = t + LB_T * (k0 + u);
                    id[u] = p < wl ? (uint32_t)L.ent[p] - 1 : 0u;
                }
#pragma unroll
                for (uint32_t u = 0; u < 8; u++) {
                    const uint32_t p = t + LB_T * (k0 + u);
                    kk[u] = w0 + p - L.r_out[id[u]];   // byte of the record
                    x[u] = L.r_ll[id[u]];
                }
#pragma unroll
                for (uint32_t u = 0; u < 8; u++) {
                    const bool lit = kk[u] < x[u];
                    kk[u] = lit ? kk[u] : 0xFFFFFFFFu;   // (literal: its index; match: marker)
                    x[u] = (&L.r_lit[0])[(lit ? 0u : LB_REC) + id[u]];   // literal source or match distance
                }
#pragma unroll
                for (uint32_t u = 0; u < 8; u++) {
                    const uint32_t p = t + LB_T * (k0 + u);
                    if (p < wl) {
                        if (kk[u] != 0xFFFFFFFFu) {   // (the literals of a record of the window machinery lie in the staged input)
                            L.ent[p] = (uint16_t)(0x8000u | L.in[x[u] + kk[u] - c0]);
                        } else if (x[u] <= p) {
                            L.ent[p] = (uint16_t)(p - x[u]);
                            pend |= 1u << (k0 + u);
                        } else if (GIANT) {
                            L.ent[p] = (uint16_t)(LB_ROOT | p);   // a root: its source is another workgroup's window
                        } else {
                            gmask |= 1u << (k0 + u);   // (ent[p] keeps the record until the second pass)
                        }
                    }
                }
            }
            LBP_CNT(21, __popc(gmask));
            while (gmask) {   // sixteen loads in flight
                uint32_t kq[16], gv[16];
#pragma unroll
                for (int u = 0; u < 16; u++) {
                    kq[u] = gmask ? (uint32_t)__builtin_ctz(gmask) : 32u;
                    gmask &= gmask - 1;   // (0 stays 0)
                    gv[u] = kq[u] < 32 ? (uint32_t)L.ent[t + LB_T * kq[u]] - 1 : 0u;
                }
#pragma unroll
                for (int u = 0; u < 16; u++) gv[u] = L.r_off[gv[u]];
#pragma unroll
                for (int u = 0; u < 16; u++) gv[u] = kq[u] < 32 ? (uint32_t)ldu8(dst + (w0 + t + LB_T * kq[u] - gv[u])) : 0u;
#pragma unroll
                for (int u = 0; u < 16; u++)
                    if (kq[u] < 32) L.ent[t + LB_T * kq[u]] = (uint16_t)(0x8000u | gv[u]);
            }
            __syncthreads();
            LBP(7);
            // pointer jumping: every pointer entry takes over the entry it points to (a byte, or a pointer further back);
            // four pending bytes of the thread in flight
            for (;;) {
                LBP_CNT(20, 1);
                int any = 0;
                uint32_t m = pend;
                while (m) {
                    uint32_t kq[4], pp[4], ss[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        kq[u] = m ? (uint32_t)__builtin_ctz(m) : 32u;
                        m &= m - 1;   // (0 stays 0)
                        pp[u] = kq[u] < 32 ? (uint32_t)L.ent[t + LB_T * kq[u]] : 0u;
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++) ss[u] = L.ent[pp[u]];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        if (kq[u] < 32) {
                            L.ent[t + LB_T * kq[u]] = (uint16_t)ss[u];
                            if (ss[u] & (0x8000u | (GIANT ? LB_ROOT : 0u)))
                                pend &= ~(1u << kq[u]);
                            else
                                any = 1;
                        }
                    }
                }
                if (!__syncthreads_or(any)) break;
            }
            LBP(8);
            // window -> HBM: 16-byte groups of the line frame of dst + w0
            if constexpr (GIANT) {
                // entries: a byte, or where the byte's root copies from (the root's record by a search over the round's records)
                for (uint32_t pp = t; pp < wl; pp += LB_T) {
                    const uint32_t e = L.ent[pp];
                    uint32_t v;
                    if (e & 0x8000u) {
                        v = 0x80000000u | (e & 0xFFu);
                    } else {
                        const uint32_t sp = e & (LB_ROOT - 1), x = w0 + sp;
                        uint32_t lo = 0, hi = nrec;   // largest r with r_out[r] <= x
                        while (hi - lo > 1) {
                            const uint32_t mid = (lo + hi) >> 1;
                            if (L.r_out[mid] <= x) lo = mid;
                            else hi = mid;
                        }
                        v = x - L.r_off[lo];   // (< x: checked above, error 104)
                    }
                    ent32[(uint64_t)w0 + pp] = v;
                }
            } else {
                const uint32_t a0 = (uint32_t)((uintptr_t)(dst + w0) & 15);
                uint8_t* gb = dst + w0 - a0;
                const uint32_t ng = (a0 + wl + 15) >> 4;
                for (uint32_t g = t; g < ng; g += LB_T) {
                    const int32_t f0 = (int32_t)(16 * g) - (int32_t)a0;   // window byte of the group's first byte
                    if (f0 >= 0 && (uint32_t)f0 + 16 <= wl) {
                        uint32_t w4[4];
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            w4[q] = (uint32_t)(L.ent[f0 + 4 * q] & 0xFF) | ((uint32_t)(L.ent[f0 + 4 * q + 1] & 0xFF) << 8) |
                                    ((uint32_t)(L.ent[f0 + 4 * q + 2] & 0xFF) << 16) | ((uint32_t)(L.ent[f0 + 4 * q + 3] & 0xFF) << 24);
                        }
                        stu128(gb + 16 * g, u32x4{w4[0], w4[1], w4[2], w4[3]});
                    } else {
                        for (int b = 0; b < 16; b++) {
                            const int32_t f = f0 + b;
                            if (f >= 0 && (uint32_t)f < wl) gb[16 * g + b] = (uint8_t)L.ent[f];
                        }
                    }
                }
            }
            wave_stores_visible();   // later windows read these bytes back (matches that reach in front of their window)
            LBP(9);
        }
        op = o_end;
        c0 = next;
        if (fin) break;
        if (GIANT && c0 >= c_end) return 0;   // the sequences behind belong to the next workgroup
        if (c0 >= n) return 105;   // the input ended without a literals-only last sequence
    }
    LBP_END;
    if (c0 != n) return 106;
    if (op != out_len) return 107;
    return 0;
}

}  // namespace sb
