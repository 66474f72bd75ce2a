// sb_lz4_giant.h — ONE LZ4 block of megabytes, decoded by the whole chip.
//
// The reference writes a column with default paging as ONE page and a Basic(LZ4) page is ONE block per buffer
// (src/compression/basic.rs:87-91,108-120): a 12 M-row Int64 column is a 68 MB block.  k_inflate_lz4_big gives a block to one
// workgroup, which walks it round by round: 0.1 GB/s.  Both serial chains of the format are resolved block-wide here:
//
//   where sequences start   k_lzg_exits   (chunk of 4096 positions x job)  every position as a possible sequence start: the
//                                         start behind it and its output bytes; then, by hops inside the chunk (64 / 512 /
//                                         4096-byte levels, loads only), where the chain from the position LEAVES the chunk and
//                                         how many bytes it produces on the way — two words per position of the block
//                           k_lzg_groups  (group of 64 chunks x job)       the same over a group, for the 4096 positions of
//                                         its first chunk, hopping over the chunks' exits
//                           k_lzg_chain   (1 thread / job)                 the real chain from position 0, a group per step:
//                                         entry position and output position of every group
//                           k_lzg_cents   (1 thread / group)               ... of every chunk
//   what bytes are          k_lzg_windows (chunk x job)                    lz4_inflate_block_wg<GIANT>: the chunk's sequences
//                                         from its entry, windows resolved in LDS; one u32 entry per output byte — a byte, or
//                                         the output position it copies when that lies in front of the window
//                           k_lzg_jump    (window of 8192 entries x job, a launch per two rounds)   entry = entry of its
//                                         source: chains through earlier windows halve per round (offsets are <= 65 535, so
//                                         a chain of matches may run through every window of the block: log2 rounds)
//                           k_lzg_pack    entries -> bytes
//
// A sequence whose header does not end within 64 bytes of its token (a literal run of > 16 000 bytes) stops the tables: the
// chain walkers parse it themselves (one thread, once), and lz4_inflate_block_wg copies its literals as a round of its own.
// Jobs are picked from the inflate queues (k_lzg_pick: LZ4, >= LZG_MIN compressed bytes, at most LZG_JOBS per call, pool
// space permitting); everything else stays with k_inflate_lz4_big.  Refusals as there: a malformed header, a match that
// reaches in front of the block, input or output that does not end where the block says.
#pragma once
#include "sb_lz4_big.h"

namespace sb {

constexpr uint32_t LZG_STAGE = LZG_CH + 320;
constexpr uint32_t CODEC_LZG = 0xFD;       // a queue entry taken by this path (the other inflate kernels skip it)
constexpr uint32_t LZG_STOP_LONG = 0x80000000u;   // | position: the chain stops AT a sequence the tables do not cover
constexpr uint32_t LZG_STOP_BAD = 0xFFFFFFFFu;
constexpr uint32_t LZG_NONE = 0xFFFFFFFFu;

__device__ __forceinline__ void lzg_fail(LzgJob* j, uint32_t code) { atomicCAS(&j->err, 0u, code); }

// ---------------------------------------------------------------------------------------------------- pick
__global__ void __launch_bounds__(256) k_lzg_pick(LzgArgs g, InflateJob* qa, const uint32_t* na, InflateJob* qb, const uint32_t* nb, uint32_t cap,
                                                  uint32_t max_jobs /* rows of the worker grids: a job beyond them would never be decoded */) {
    __shared__ uint32_t s_n;
    __shared__ uint32_t s_list[LZG_JOBS][2];
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    for (uint32_t q = 0; q < 2; q++) {
        InflateJob* jobs = q ? qb : qa;
        if (!jobs) continue;
        const uint32_t n = min(*(q ? nb : na), cap);
        for (uint32_t j = threadIdx.x; j < n; j += blockDim.x) {
            const InflateJob jb = jobs[j];
            if (jb.codec != SB_CODEC_LZ4 || jb.csize < LZG_MIN || jb.csize >= 0x7FFFFFF0u || jb.out_len >= 0x7FFFFFF0u || jb.out_len == 0) continue;
            const uint32_t at = atomicAdd(&s_n, 1u);
            if (at < LZG_JOBS) {
                s_list[at][0] = q;
                s_list[at][1] = j;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x) return;
    const uint32_t cand = min(min(s_n, LZG_JOBS), max_jobs);   // (the rest stay with the one-workgroup decoder)
    uint64_t cur = 0;
    uint32_t taken = 0;
    for (uint32_t k = 0; k < cand; k++) {
        InflateJob* jobs = s_list[k][0] ? qb : qa;
        const InflateJob jb = jobs[s_list[k][1]];
        LzgJob o;
        __builtin_memset(&o, 0, sizeof o);
        o.src = jb.src;
        o.dst = jb.dst;
        o.n = jb.csize;
        o.out_len = jb.out_len;
        o.page = jb.page;
        o.nchunks = (jb.csize + LZG_CH - 1) / LZG_CH;
        o.ngroups = (o.nchunks + LZG_GROUP - 1) / LZG_GROUP;
        o.nwin = (jb.out_len + LZG_WIN - 1) / LZG_WIN;
        o.left = o.nwin;
        o.queue = s_list[k][0];
        o.slot = s_list[k][1];
        auto take = [&](uint64_t bytes) -> uint8_t* {
            uint8_t* p = g.pool + cur;
            cur += (bytes + 255) & ~255ull;
            return p;
        };
        const uint64_t before = cur;
        o.eo = (uint32_t*)take((uint64_t)o.n * 8);
        o.gtab = (uint32_t*)take((uint64_t)o.n * 8);
        o.gent = (uint32_t*)take((uint64_t)o.ngroups * 8);
        o.cent = (uint32_t*)take((uint64_t)o.nchunks * 8);
        o.ent = (uint32_t*)take((uint64_t)o.out_len * 4);
        o.wdone = (uint32_t*)take((uint64_t)o.nwin * 4);
        o.lits = (uint32_t*)take((uint64_t)LZG_LITS * 16);
        if (cur > g.pool_bytes) {   // no room: the block stays with the workgroup decoder
            cur = before;
            continue;
        }
        g.jobs[taken++] = o;
        jobs[s_list[k][1]].codec = CODEC_LZG;
    }
    *g.njobs = taken;
}

// the tables that say "not on the chain" / "not done" cleared
__global__ void __launch_bounds__(256) k_lzg_clear(LzgArgs g) {
    if (blockIdx.y >= *g.njobs) return;
    const LzgJob j = g.jobs[blockIdx.y];
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = tid; i < (uint64_t)j.ngroups * 2; i += nth) j.gent[i] = LZG_NONE;
    for (uint64_t i = tid; i < (uint64_t)j.nchunks * 2; i += nth) j.cent[i] = LZG_NONE;
    for (uint64_t i = tid; i < j.nwin; i += nth) j.wdone[i] = 0;
}

// The sequence whose token is byte i of a block of `room` bytes, with at most LZG_EXT length bytes per length: kind 2 = a
// longer header (a literal run / match of > 16 000 bytes: the chain walkers parse it, once), 3 = malformed.  In a run of
// 255s every position is such a header: unbounded, each of them would walk the run.
constexpr uint32_t LZG_EXT = 64;
constexpr int LZG_PASSES = 12;   // passes of k_lzg_jump per launch
template <class RD>
__device__ __forceinline__ LbSeq lzg_seq(RD rd, uint32_t i, uint32_t room) {
    LbSeq s;
    s.ll = s.lit = s.ml = s.off = s.end = 0;
    s.kind = 3;
    const uint32_t t = rd(i);
    uint32_t j = i + 1, ll = t >> 4;
    if (ll == 15) {
        uint32_t b, cnt = 0;
        do {
            if (j >= room) return s;
            if (++cnt > LZG_EXT) {
                s.kind = 2;
                return s;
            }
            b = rd(j++);
            ll += b;
        } while (b == 255);
    }
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
    s.off = rd(j) | (rd(j + 1) << 8);
    j += 2;
    uint32_t ml = t & 15;
    if (ml == 15) {
        uint32_t b, cnt = 0;
        do {
            if (j >= room) return s;
            if (++cnt > LZG_EXT) {
                s.kind = 2;
                return s;
            }
            b = rd(j++);
            ml += b;
        } while (b == 255);
    }
    s.ml = ml + 4;
    s.end = j;
    s.kind = s.off ? 0u : 3u;
    return s;
}

// ---------------------------------------------------------------------------------------------------- exits of a chunk
struct LzgExitLds {
    __attribute__((aligned(16))) uint8_t in[LZG_STAGE + 16];
    uint16_t pos[LZG_CH];     // next position on the chain inside the chunk; 0x8000 | j: the chain leaves the chunk through j
    uint32_t sum[LZG_CH];     // output bytes up to there
    uint32_t nabs[LZG_CH];    // where the sequence at j ends (absolute), or a stop code
};
__global__ void __launch_bounds__(256, 2) k_lzg_exits(LzgArgs g) {
    __shared__ LzgExitLds L;
    if (blockIdx.y >= *g.njobs) return;
    LzgJob* jp = g.jobs + blockIdx.y;
    const LzgJob j = *jp;
    if (blockIdx.x >= j.nchunks) return;
    const uint32_t t = threadIdx.x;
    const uint32_t c0 = blockIdx.x * LZG_CH, n = j.n;
    const uint32_t room = n - c0, npos = min(LZG_CH, room), sl = min(LZG_STAGE, room);
    const uint8_t* src = j.src;
    for (uint32_t k = t * 16; k < sl; k += 256 * 16) {
        if (c0 + k + 16 <= n) {
            *(u32x4*)(L.in + k) = ldu128(src + c0 + k);
        } else {
            for (uint32_t b = 0; b < 16; b++) L.in[k + b] = c0 + k + b < n ? ldu8(src + c0 + k + b) : (uint8_t)0;
        }
    }
    __syncthreads();
    auto rd_mix = [&](uint32_t x) -> uint32_t { return x < sl ? (uint32_t)L.in[x] : (uint32_t)ldu8(src + c0 + x); };
    constexpr uint32_t PPT = LZG_CH / 256;
#pragma unroll 1
    for (uint32_t k = 0; k < PPT; k++) {
        const uint32_t i = t + 256 * k;
        if (i >= npos) {
            L.pos[i] = (uint16_t)(0x8000u | i);
            L.sum[i] = 0;
            L.nabs[i] = LZG_STOP_BAD;
            continue;
        }
        LbSeq sq;
        const uint32_t tok = L.in[i], b1 = L.in[i + 1];
        const uint32_t e = lb_hdr_e(i, tok, b1);
        const uint32_t ext = L.in[min(e, LZG_STAGE - 1)];
        if (lb_hdr_simple(tok, b1, ext, e, sl)) {
            const uint32_t l4 = tok >> 4, m4 = tok & 15;
            sq.ll = l4 != 15 ? l4 : 15 + b1;
            sq.off = (uint32_t)L.in[e - 2] | ((uint32_t)L.in[e - 1] << 8);
            sq.ml = 4 + m4 + (m4 == 15 ? ext : 0u);
            sq.end = e + (m4 == 15 ? 1u : 0u);
            sq.kind = sq.off ? 0u : 3u;
        } else {
            sq = lzg_seq(rd_mix, i, room);
        }
        if (sq.kind == 0 && sq.end >= room) sq.kind = 3;   // (the block ends in a literals-only sequence)
        uint32_t nx, ol, na;
        if (sq.kind == 0) {
            ol = sq.ll + sq.ml;
            na = c0 + sq.end;
            if (sq.ml > 0x7FFFFFFFu - sq.ll) {   // (lengths beyond any output: malformed for this path)
                ol = 0;
                na = LZG_STOP_BAD;
            }
            nx = sq.end < npos && na != LZG_STOP_BAD ? sq.end : (0x8000u | i);
        } else if (sq.kind == 1) {   // the block's last sequence: literals up to its end
            ol = sq.ll;
            na = n;
            nx = 0x8000u | i;
        } else if (sq.kind == 2) {
            ol = 0;
            na = LZG_STOP_LONG | (c0 + i);
            nx = 0x8000u | i;
        } else {
            ol = 0;
            na = LZG_STOP_BAD;
            nx = 0x8000u | i;
        }
        L.pos[i] = (uint16_t)nx;
        L.sum[i] = ol;
        L.nabs[i] = na;
    }
    __syncthreads();
    // levels: where the chain from i leaves i's 64- / 512-byte segment / the chunk, and the bytes on the way
    uint32_t jj[PPT], ss[PPT];
    const uint32_t LV[3] = {64, 512, LZG_CH};
    for (int lv = 0; lv < 3; lv++) {
        const uint32_t seg = LV[lv];
#pragma unroll
        for (uint32_t k = 0; k < PPT; k++) {
            const uint32_t i = t + 256 * k;
            const uint32_t send = (i | (seg - 1)) + 1;
            uint32_t p = i, s = 0;
            for (;;) {
                if (p & 0x8000u) break;
                if (p >= send) break;
                const uint32_t a = L.sum[p];
                s = s > 0x7FFFFFFFu - a ? 0x7FFFFFFFu : s + a;
                p = L.pos[p];
            }
            jj[k] = p;
            ss[k] = s;
        }
        __syncthreads();
#pragma unroll
        for (uint32_t k = 0; k < PPT; k++) {
            const uint32_t i = t + 256 * k;
            // (a chain that left through j keeps naming j: nabs[j] says where to; its bytes are all in the sum)
            L.pos[i] = (uint16_t)jj[k];
            L.sum[i] = ss[k];
        }
        __syncthreads();
    }
#pragma unroll
    for (uint32_t k = 0; k < PPT; k++) {
        const uint32_t i = t + 256 * k;
        if (i >= npos) continue;
        const uint32_t last = jj[k] & 0x7FFFu;
        uint32_t* o = j.eo + 2 * (uint64_t)(c0 + i);
        o[0] = L.nabs[last];
        o[1] = ss[k];
    }
}

// one hop over the chunk tables by a WAVE (all lanes, the same arguments); a LONG stop is parsed on the spot, its length
// bytes 1 KiB at a time (lb_seq_wave).  Returns false on error.
__device__ inline bool lzg_hop(const LzgJob& j, LzgJob* jp, uint32_t& pos, uint64_t& out) {
    const uint32_t e = j.eo[2 * (uint64_t)pos], o = j.eo[2 * (uint64_t)pos + 1];
    out += o;
    if (e == LZG_STOP_BAD) {
        lzg_fail(jp, 101);
        return false;
    }
    if (e & LZG_STOP_LONG) {
        const uint32_t p = e & 0x7FFFFFFFu;
        const uint32_t room = j.n - p;
        const LbSeq sq = lb_seq_wave(j.src + p, room);
        if (sq.kind == 3 || (sq.kind == 0 && sq.end >= room)) {
            lzg_fail(jp, 101);
            return false;
        }
        out += (uint64_t)sq.ll + sq.ml;
        pos = sq.kind == 1 ? j.n : p + sq.end;
        return true;
    }
    pos = e;
    return true;
}

// Exits of a GROUP of LZG_GROUP chunks for EVERY position of the group, by suffix composition: the last chunk's exits are
// the group's; a position of chunk c whose chunk exit lies inside the group takes over what that exit position has (one
// gather per position and pass, 64 dependent passes per group, all groups side by side).  A chain that enters a group
// anywhere — long literal runs jump over chunks — leaves it with ONE lookup.
constexpr uint32_t LZG_GT = 1024;   // threads of k_lzg_groups: a pass is a latency chain, four positions per thread
__global__ void __launch_bounds__(LZG_GT) k_lzg_groups(LzgArgs g) {
    if (blockIdx.y >= *g.njobs) return;
    const LzgJob j = g.jobs[blockIdx.y];
    const uint32_t grp = blockIdx.x;
    if (grp >= j.ngroups) return;
    const uint32_t cfirst = grp * LZG_GROUP, clast = min(j.nchunks, cfirst + LZG_GROUP);
    const uint64_t g1 = min((uint64_t)j.n, (uint64_t)clast * LZG_CH);
    for (uint32_t c = clast; c-- > cfirst;) {
        const uint64_t p0 = (uint64_t)c * LZG_CH;
        uint32_t e[LZG_CH / LZG_GT], b[LZG_CH / LZG_GT], e2[LZG_CH / LZG_GT], b2[LZG_CH / LZG_GT];
#pragma unroll
        for (uint32_t k = 0; k < LZG_CH / LZG_GT; k++) {
            const uint64_t p = p0 + threadIdx.x + LZG_GT * k;
            e[k] = p < j.n ? j.eo[2 * p] : LZG_STOP_BAD;
            b[k] = p < j.n ? j.eo[2 * p + 1] : 0u;
        }
#pragma unroll
        for (uint32_t k = 0; k < LZG_CH / LZG_GT; k++) {
            const bool in = e[k] != LZG_STOP_BAD && !(e[k] & LZG_STOP_LONG) && e[k] < g1;   // (stops end the table: the walker takes single hops from there)
            e2[k] = in ? j.gtab[2 * (uint64_t)e[k]] : 0u;
            b2[k] = in ? j.gtab[2 * (uint64_t)e[k] + 1] : 0u;
            if (in) {
                // (a stop further on: keep the hop to the position in front of it — the table may only name exits it has summed up to)
                if (e2[k] == LZG_STOP_BAD || (e2[k] & LZG_STOP_LONG)) {
                    e2[k] = e[k];
                    b2[k] = b[k];
                } else {
                    b2[k] = b2[k] > 0x7FFFFFFFu - b[k] ? 0x7FFFFFFFu : b2[k] + b[k];
                }
            } else {
                e2[k] = e[k];
                b2[k] = b[k];
            }
        }
#pragma unroll
        for (uint32_t k = 0; k < LZG_CH / LZG_GT; k++) {
            const uint64_t p = p0 + threadIdx.x + LZG_GT * k;
            if (p < j.n) {
                j.gtab[2 * p] = e2[k];
                j.gtab[2 * p + 1] = b2[k];
            }
        }
        __threadfence_block();
        __syncthreads();
    }
}

__global__ void __launch_bounds__(64) k_lzg_chain(LzgArgs g) {
    if (blockIdx.x >= *g.njobs) return;   // (one wave, every lane with the same state: the long headers are read by all of them)
    LzgJob* jp = g.jobs + blockIdx.x;
    const LzgJob j = *jp;
    uint32_t pos = 0;
    uint64_t out = 0;
    const uint64_t GB = (uint64_t)LZG_GROUP * LZG_CH;
    while (pos < j.n) {
        const uint32_t grp = (uint32_t)(pos / GB);
        const uint64_t g0 = (uint64_t)grp * GB, g1 = min((uint64_t)j.n, g0 + GB);
        if (threadIdx.x == 0) {
            j.gent[2 * grp] = pos;
            j.gent[2 * grp + 1] = (uint32_t)out;
        }
        if (out > j.out_len) {
            lzg_fail(jp, 103);
            return;
        }
        (void)g0;
        {
            const uint32_t ge = j.gtab[2 * (uint64_t)pos], gb = j.gtab[2 * (uint64_t)pos + 1];
            if (ge != LZG_STOP_BAD && !(ge & LZG_STOP_LONG)) {   // (a table entry that ends in a stop is the chunk table's own: single hops)
                out += gb;
                pos = ge;
                if (pos >= g1) continue;
            }
        }
        while (pos < g1)
            if (!lzg_hop(j, jp, pos, out)) return;
    }
    if (pos != j.n) lzg_fail(jp, 106);
    else if (out != j.out_len) lzg_fail(jp, out > j.out_len ? 103 : 107);
}

__global__ void __launch_bounds__(64) k_lzg_cents(LzgArgs g) {   // a wave per group
    if (blockIdx.y >= *g.njobs) return;
    LzgJob* jp = g.jobs + blockIdx.y;
    const LzgJob j = *jp;
    const uint32_t grp = blockIdx.x;
    if (grp >= j.ngroups || j.err) return;
    uint32_t pos = j.gent[2 * grp];
    if (pos == LZG_NONE) return;
    uint64_t out = j.gent[2 * grp + 1];
    const uint64_t g1 = min((uint64_t)j.n, ((uint64_t)grp + 1) * LZG_GROUP * LZG_CH);
    while (pos < g1) {
        const uint32_t c = pos / LZG_CH;
        if (threadIdx.x == 0) {
            j.cent[2 * c] = pos;
            j.cent[2 * c + 1] = (uint32_t)out;
        }
        const uint32_t cend = min(j.n, (c + 1) * LZG_CH);
        while (pos < cend)   // (one hop leaves the chunk, unless it was a LONG stop short of the chunk's end: then the chain goes on inside)
            if (!lzg_hop(j, jp, pos, out)) return;
    }
}

// ---------------------------------------------------------------------------------------------------- bytes
__global__ void __launch_bounds__(LB_T, 4) k_lzg_windows(LzgArgs g) {
    __shared__ Lz4BigLds lds;
    if (blockIdx.y >= *g.njobs) return;
    LzgJob* jp = g.jobs + blockIdx.y;
    const LzgJob j = *jp;
    if (blockIdx.x >= j.nchunks || j.err) return;
    const uint32_t pos = j.cent[2 * blockIdx.x];
    if (pos == LZG_NONE) return;
    const uint32_t cend = min(j.n, (blockIdx.x + 1) * LZG_CH);
    const uint32_t e = lz4_inflate_block_wg<true>(j.src, j.n, nullptr, j.out_len, lds, pos, cend, j.cent[2 * blockIdx.x + 1], j.ent, j.lits, &jp->nlits);
    if (e && threadIdx.x == 0) lzg_fail(jp, e);
}

// the long literal runs k_lzg_windows listed: bytes -> entries, every workgroup a share of every run
__global__ void __launch_bounds__(256) k_lzg_lits(LzgArgs g) {
    if (blockIdx.y >= *g.njobs) return;
    const LzgJob j = g.jobs[blockIdx.y];
    const uint32_t nl = min(j.nlits, LZG_LITS);
    if (j.err || !nl) return;
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (uint64_t)gridDim.x * blockDim.x;
    for (uint32_t r = 0; r < nl; r++) {
        const uint8_t* src = j.src + j.lits[4 * r];
        uint32_t* dst = j.ent + j.lits[4 * r + 1];
        const uint32_t len = j.lits[4 * r + 2];
        for (uint64_t k = tid * 16; k < len; k += nth * 16) {
            if (k + 16 <= len) {
                const u32x4 v = ldu128(src + k);
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    u32x4 o;
                    o.x = 0x80000000u | (w[q] & 0xFFu);
                    o.y = 0x80000000u | ((w[q] >> 8) & 0xFFu);
                    o.z = 0x80000000u | ((w[q] >> 16) & 0xFFu);
                    o.w = 0x80000000u | (w[q] >> 24);
                    __builtin_memcpy(dst + k + 4 * q, &o, 16);
                }
            } else {
                for (uint64_t b = k; b < len; b++) dst[b] = 0x80000000u | (uint32_t)ldu8(src + b);
            }
        }
    }
}

// entries that copy an entry in front of their window take over that entry: a byte, or a position further back
__global__ void __launch_bounds__(256) k_lzg_jump(LzgArgs g) {
    __shared__ uint32_t s_left;
    if (blockIdx.y >= *g.njobs) return;
    LzgJob* jp = g.jobs + blockIdx.y;
    const LzgJob j = *jp;
    if (blockIdx.x >= j.nwin || j.err) return;
    if (__hip_atomic_load(&jp->left, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;
    if (j.wdone[blockIdx.x]) return;
    const uint64_t w0 = (uint64_t)blockIdx.x * LZG_WIN;
    const uint32_t wl = (uint32_t)min((uint64_t)LZG_WIN, (uint64_t)j.out_len - w0);
    uint32_t* ent = j.ent + w0;
    constexpr uint32_t EPT = LZG_WIN / 256;
    uint32_t v[EPT];
#pragma unroll
    for (uint32_t k = 0; k < EPT; k++) {
        const uint32_t p = threadIdx.x + 256 * k;
        v[k] = p < wl ? ent[p] : 0x80000000u;
    }
    // (workgroups start in window order and a window's sources lie in earlier windows: by the time a window runs, most of what
    // it points at is final — a few passes inside one launch do what a launch per pass needed fourteen of)
    for (int it = 0; it < LZG_PASSES; it++) {
        uint32_t mine = 0;
        uint32_t s[EPT];
#pragma unroll
        for (uint32_t k = 0; k < EPT; k++) {
            // (a position is always in front of its entry: checked when the windows were written — the clamp only keeps a
            // damaged table from reading outside the entries)
            s[k] = (v[k] & 0x80000000u) ? v[k] : __hip_atomic_load(j.ent + min(v[k], j.out_len - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (uint32_t k = 0; k < EPT; k++) {
            if (!(v[k] & 0x80000000u)) {
                v[k] = s[k];
                const uint32_t p = threadIdx.x + 256 * k;
                __hip_atomic_store(ent + p, v[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (!(v[k] & 0x80000000u)) mine++;
            }
        }
        if (threadIdx.x == 0) s_left = 0;
        __syncthreads();
        if (mine) atomicAdd(&s_left, mine);
        __syncthreads();
        if (s_left == 0) {
            if (threadIdx.x == 0) {
                j.wdone[blockIdx.x] = 1;
                __hip_atomic_fetch_sub(&jp->left, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            return;
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256) k_lzg_pack(LzgArgs g) {
    if (blockIdx.y >= *g.njobs) return;
    LzgJob* jp = g.jobs + blockIdx.y;
    const LzgJob j = *jp;
    if (j.err) {
        if (blockIdx.x == 0 && threadIdx.x == 0) raise(g.st, SB_ERR_EXTERNAL, j.page, j.err);
        return;
    }
    // 16 output bytes per thread and step, stored as one aligned vector where dst allows
    const uint64_t ngrp = ((uint64_t)j.out_len + 15) / 16;
    bool bad = false;
    for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q < ngrp; q += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t b0 = q * 16;
        const uint32_t nb = (uint32_t)min((uint64_t)16, (uint64_t)j.out_len - b0);
        uint32_t w4[4] = {0, 0, 0, 0};
        if (nb == 16) {
            const u32x4* e4 = (const u32x4*)(j.ent + b0);
#pragma unroll
            for (int q4 = 0; q4 < 4; q4++) {
                const u32x4 e = e4[q4];
                bad |= !((e.x & e.y & e.z & e.w) & 0x80000000u);
                w4[q4] = (e.x & 0xFFu) | ((e.y & 0xFFu) << 8) | ((e.z & 0xFFu) << 16) | ((e.w & 0xFFu) << 24);
            }
            if (((uintptr_t)(j.dst + b0) & 15) == 0) {
                stu128(j.dst + b0, u32x4{w4[0], w4[1], w4[2], w4[3]});
            } else {
                for (int b = 0; b < 16; b++) j.dst[b0 + b] = (uint8_t)(w4[b >> 2] >> (8 * (b & 3)));
            }
        } else {
            for (uint32_t b = 0; b < nb; b++) {
                const uint32_t e = j.ent[b0 + b];
                bad |= !(e & 0x80000000u);
                j.dst[b0 + b] = (uint8_t)e;
            }
        }
    }
    if (bad) raise(g.st, SB_ERR_EXTERNAL, j.page, 108);   // an entry that never met a byte (cannot happen: sources lie in front)
}

}  // namespace sb
