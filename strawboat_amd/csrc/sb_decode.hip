// strawboat-hip: page decode kernels for gfx950 (MI355X).
//
// Replaces, on the device, the per-page work of the reference's batch read path:
//   read_validity            src/read/read_basic.rs:36-63
//   decompress_integer       src/compression/integer/mod.rs:72-117   (+ rle.rs, dict.rs, bp.rs,
//   decompress_double        src/compression/double/mod.rs:69-114      delta_bp.rs, one_value.rs)
//   decompress_boolean       src/compression/boolean/mod.rs:63-102   (+ rle.rs, one_value.rs)
//   decompress_binary        src/compression/binary/mod.rs:95-183    (+ dict.rs, one_value.rs)
// and the page concatenation of read_integer / read_boolean / read_binary
// (src/read/array/integer.rs:210-238, boolean.rs:191-219, binary.rs:223-265).
//
// Kernel sequence of one call (all on one stream, no host round trip):
//   k_parse    1 thread / page      headers -> PageDesc, tile table, inflate job queues
//   k_inflate  1 wave / job         LZ4 blocks -> output or scratch (queue A, later queue B)
//   k_plan     1 workgroup / page   page-wide scans: RLE run starts, bit-pack block offsets,
//                                   delta bases, binary-dict entry offsets and byte totals
//   k_colscan  1 workgroup / column binary columns: cross-page value/offset bases
//   k_expand   1 workgroup / tile   TILE_ROWS rows of one page -> Arrow buffers
// Bandwidth-bound integer work: no MFMA; coalesced 16-byte stores, unaligned 16-byte loads,
// LDS for the per-tile scans, wave64 shuffles for the scan carries.
#include "sb_host.h"
#include "sb_zstd.h"
#include "sb_zstd_blocks.h"
#include "sb_lz4.h"
#include "sb_lz4_big.h"
#include "sb_lz4_giant.h"

namespace sb {

__device__ __forceinline__ bool is_basic(uint32_t c) { return c <= 3; }
__device__ __forceinline__ bool is_binary(int32_t t) { return t == SB_TYPE_BINARY || t == SB_TYPE_LARGE_BINARY; }

__device__ __forceinline__ void push_job(InflateJob* q, uint32_t* cnt, const uint8_t* src, uint32_t csize,
                                         uint8_t* dst, uint32_t out_len, uint32_t codec, uint32_t page);
// a Basic payload that no planning step reads: queue A when the call has no binary column (there is no later phase), else
// queue Z (Zstd: its entropy stages run with queue A's) or queue B (the other codecs, inflated next to the binary value blocks)
// push_job for the lanes with `pred`, called by ALL lanes of a branch with the SAME queue: one atomic per wave — the lane that
// leads takes popcount slots, every lane its rank among them.  (The compiler aggregates a plain atomicAdd by itself only
// while the counter's address is provably wave-uniform; a queue chosen per page is not, and 65 536 one-thread atomics on one
// word made k_parse ten times slower.)
__device__ __forceinline__ void push_job_if(bool pred, InflateJob* q, uint32_t* cnt, const uint8_t* src, uint32_t csize, uint8_t* dst,
                                            uint32_t out_len, uint32_t codec, uint32_t page) {
    const uint64_t m = __ballot(pred);
    if (!m) return;   // (uniform)
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
    uint32_t base = 0;
    if (pred && rank == 0) base = __hip_atomic_fetch_add(cnt, (uint32_t)__popcll(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    base = (uint32_t)__builtin_amdgcn_readlane((int)base, __builtin_ctzll(m));
    if (pred) {
        InflateJob j;
        j.src = src;
        j.dst = dst;
        j.csize = csize;
        j.out_len = out_len;
        j.codec = codec;
        j.page = page;
        q[base + rank] = j;
    }
}
__device__ __forceinline__ void push_payload(const DecodeArgs& a, const uint8_t* src, uint32_t csize, uint8_t* dst, uint32_t out_len,
                                             uint32_t codec, uint32_t page) {
    // every lane that gets here offers its payload to each queue in turn: the queues are the call's (wave-uniform)
    const bool z = a.defer_payloads && a.jobs_z && codec == SB_CODEC_ZSTD;
    const bool b = a.defer_payloads && !z;
    push_job_if(!a.defer_payloads, a.jobs_a, a.job_counts, src, csize, dst, out_len, codec, page);
    push_job_if(b, a.jobs_b, a.job_counts + 1, src, csize, dst, out_len, codec, page);
    if (a.jobs_z) push_job_if(z, a.jobs_z, a.job_counts + 10, src, csize, dst, out_len, codec, page);
}
__device__ __forceinline__ void push_job(InflateJob* q, uint32_t* cnt, const uint8_t* src, uint32_t csize,
                                         uint8_t* dst, uint32_t out_len, uint32_t codec, uint32_t page) {
    uint32_t i = atomicAdd(cnt, 1u);
    InflateJob j;
    j.src = src;
    j.dst = dst;
    j.csize = csize;
    j.out_len = out_len;
    j.codec = codec;
    j.page = page;
    q[i] = j;
}

// ---- Zstd buffers made of several frames (what this library's encoder writes for buffers of more than 32 KiB, and what
// ZSTD_decompress accepts from anyone): one job per frame, so that a page's frames decode on waves of their own.
constexpr uint32_t CODEC_SPLIT = 0xFF;   // a queue entry whose frames were queued one by one (k_inflate skips it)
// One frame at src[pos..n): its size and its content size.  false: not a plain frame with a known content size.
__device__ inline bool zstd_frame_extent(const uint8_t* src, uint32_t n, uint32_t pos, uint32_t* fsize, uint32_t* fcs_out) {
    if (n - pos < 6 || ldu32(src + pos) != 0xFD2FB528u) return false;
    uint32_t ip = pos + 4;
    const uint8_t fhd = ldu8(src + ip++);
    const int fcs_flag = fhd >> 6, single = (fhd >> 5) & 1, checksum = (fhd >> 2) & 1, did = fhd & 3;
    if ((fhd & 0x08) || did) return false;
    if (!single) ip += 1;
    const int fcs_bytes = fcs_flag == 0 ? (single ? 1 : 0) : (1 << fcs_flag);
    if (fcs_bytes == 0 || fcs_bytes == 8 || n - ip < (uint32_t)fcs_bytes) return false;
    uint32_t fcs = 0;
    for (int i = 0; i < fcs_bytes; i++) fcs |= (uint32_t)ldu8(src + ip + i) << (8 * i);
    if (fcs_bytes == 2) fcs += 256;
    ip += fcs_bytes;
    for (;;) {
        if (n - ip < 3) return false;
        const uint32_t bh = (uint32_t)ldu8(src + ip) | ((uint32_t)ldu8(src + ip + 1) << 8) | ((uint32_t)ldu8(src + ip + 2) << 16);
        ip += 3;
        const uint32_t btype = (bh >> 1) & 3, bsize = bh >> 3;
        if (btype == 3) return false;
        const uint32_t body = btype == 1 ? 1u : bsize;
        if (n - ip < body) return false;
        ip += body;
        if (bh & 1) break;
    }
    if (checksum) {
        if (n - ip < 4) return false;
        ip += 4;
    }
    *fsize = ip - pos;
    *fcs_out = fcs;
    return true;
}
// The same for the common shape — a frame of ONE block whose header sits in the first 16 bytes — from a single 16-byte
// load: the walk over a buffer of many frames is a chain of dependent HBM reads, one per frame this way instead of ~8
__device__ inline bool zstd_frame_extent_fast(const uint8_t* src, uint32_t n, uint32_t pos, uint32_t* fsize, uint32_t* fcs_out) {
    if (n - pos >= 16) {
        const u32x4 q = ldu128(src + pos);
        const uint64_t lo = (uint64_t)q.x | ((uint64_t)q.y << 32), hi = (uint64_t)q.z | ((uint64_t)q.w << 32);
        auto byte = [&](uint32_t i) { return (uint32_t)((i < 8 ? lo >> (8 * i) : hi >> (8 * (i - 8))) & 0xFF); };
        const uint32_t fhd = byte(4);
        const int fcs_flag = fhd >> 6, single = (fhd >> 5) & 1, checksum = (fhd >> 2) & 1, did = fhd & 3;
        const int fcs_bytes = fcs_flag == 0 ? (single ? 1 : 0) : (1 << fcs_flag);
        if (q.x == 0xFD2FB528u && !(fhd & 0x08) && !did && !checksum && fcs_bytes != 0 && fcs_bytes != 8) {
            uint32_t ip = 5 + (single ? 0 : 1);
            uint32_t fcs = 0;
            for (int i = 0; i < fcs_bytes; i++) fcs |= byte(ip + i) << (8 * i);
            if (fcs_bytes == 2) fcs += 256;
            ip += fcs_bytes;   // <= 10
            const uint32_t bh = byte(ip) | (byte(ip + 1) << 8) | (byte(ip + 2) << 16);
            const uint32_t btype = (bh >> 1) & 3, bsize = bh >> 3;
            if ((bh & 1) && btype != 3) {
                const uint32_t body = btype == 1 ? 1u : bsize;
                if (n - pos - (ip + 3) < body) return false;
                *fsize = ip + 3 + body;
                *fcs_out = fcs;
                return true;
            }
        }
    }
    return zstd_frame_extent(src, n, pos, fsize, fcs_out);
}
// k_zstd_split, one thread per queue entry, launched when the queue is complete (*n0_p = its length then, recorded by the
// last workgroup of the kernel that filled it): every Zstd entry that is >= 2 well-formed frames whose content sizes add
// up to the entry's output is replaced by one entry per frame (the walk reads a few bytes per frame).  Anything else stays
// as it is and is decoded by one wave, frame after frame.  Threads only look at entries below *n0_p, so the entries other
// threads append meanwhile are never walked.
// the frames of queue entry j, found by one lane walking the buffer
__device__ void zstd_split_walk(InflateJob* q, uint32_t* cnt, uint32_t cap, uint32_t j, const InflateJob& job) {
    // ONE walk: the frames go to queue slots reserved SPLIT_BATCH at a time as they are found; when the buffer turns
    // out not to be a plain chain of frames the slots written so far are marked "skip" and the entry keeps its frames
    // (the queue's consumers are launched after this kernel)
    constexpr uint32_t SPLIT_BATCH = 64, MAX_BATCHES = 16;   // batch k holds 64 (k + 1) slots: 16 reservations reach 8 704 frames
    uint32_t first_pos, first_fs = 0, first_fc = 0;
    if (!zstd_frame_extent_fast(job.src, job.csize, 0, &first_fs, &first_fc) || first_fs >= job.csize || first_fc > job.out_len) return;
    first_pos = first_fs;   // (a single frame: nothing to split)
    uint32_t bases[MAX_BATCHES], nb = 0, room = 0, at = 0, nf = 0, pos = 0, total = 0;
    bool ok = true;
    uint32_t fs = first_fs, fc = first_fc;
    (void)first_pos;
    for (;;) {
        if (room == 0) {
            if (nb == MAX_BATCHES) {
                ok = false;
                break;
            }
            const uint32_t want = SPLIT_BATCH * (nb + 1);
            const uint32_t b = atomicAdd(cnt, want);
            if (b + want > cap) {   // no room: the part of the reservation inside the queue is marked "skip" (slots at
                ok = false;         // or beyond cap are never read: the consumers clamp the count to cap)
                for (uint32_t k = b; k < cap && k < b + want; k++) q[k].codec = CODEC_SPLIT;
                break;
            }
#pragma unroll
            for (uint32_t k = 0; k < MAX_BATCHES; k++)
                if (k == nb) bases[k] = b;
            at = b;
            room = want;
            nb++;
        }
        InflateJob f = job;
        f.src = job.src + pos;
        f.dst = job.dst + total;
        f.csize = fs;
        f.out_len = fc;
        q[at++] = f;
        room--;
        nf++;
        pos += fs;
        total += fc;
        if (pos >= job.csize) break;
        if (!zstd_frame_extent_fast(job.src, job.csize, pos, &fs, &fc) || fc > job.out_len - total) {
            ok = false;
            break;
        }
    }
    ok = ok && nf >= 2 && total == job.out_len && pos == job.csize;
#pragma unroll
    for (uint32_t k = 0; k < MAX_BATCHES; k++) {   // the unused tail of the last batch — or, on failure, every slot of mine
        if (k >= nb) break;
        const uint32_t want = SPLIT_BATCH * (k + 1);
        const uint32_t lo = ok ? (k + 1 == nb ? at : bases[k] + want) : bases[k];
        for (uint32_t i = lo; i < bases[k] + want; i++) q[i].codec = CODEC_SPLIT;
    }
    if (ok) q[j].codec = CODEC_SPLIT;
}

// ---- LONG buffers (>= ZS_BIG bytes: a one-page column): the walk above is a chain of dependent HBM reads, one per frame —
// 3 000 frames, 4.6 ms for a 96 MB page.  Every frame begins with the magic number, so: all workgroups scan the buffer for
// it (a candidate per hit, filed under its 16 KiB segment); one workgroup per buffer then lists the candidates in position
// order in LDS, computes every candidate's extent in parallel (one 16-byte load each), links candidate -> the candidate
// at its end by binary search, and ONE lane follows the links from position 0 through LDS (30 ns per frame).  A magic
// inside compressed data is just a candidate off the chain.  Anything unexpected (a segment with > 7 hits, > 4096
// candidates, a broken chain) falls back to zstd_split_walk for that buffer.
constexpr uint32_t ZS_BIG = 1u << 20, ZS_SEG = 16384, ZS_SLOTS = 7, ZS_LIST = 16, ZS_MAXC = 4096;
__global__ void __launch_bounds__(WG) k_zsplit_scan(const InflateJob* q, DecodeArgs a) {
    const uint32_t e = blockIdx.y;
    if (e >= min(a.zs_hdr[0], ZS_LIST)) return;
    const uint32_t j = a.zs_hdr[16 + 2 * e], seg0 = a.zs_hdr[17 + 2 * e];
    if (j == 0xFFFFFFFFu) return;
    const InflateJob job = q[j];
    const uint32_t nseg = (job.csize + ZS_SEG - 1) / ZS_SEG;
    for (uint32_t seg = blockIdx.x; seg < nseg; seg += gridDim.x) {
        uint32_t* rec = a.zs_segs + (uint64_t)(seg0 + seg) * 8;
        const uint32_t b = seg * ZS_SEG + threadIdx.x;   // lane = byte position: a wave's load is one or two cache lines
#pragma unroll 8
        for (uint32_t k = 0; k < ZS_SEG / WG; k++) {
            const uint32_t pos = b + k * WG;
            if (pos + 4 > job.csize) break;
            if (ldu32(job.src + pos) == 0xFD2FB528u) {
                const uint32_t r = atomicAdd(&rec[0], 1u);
                if (r < ZS_SLOTS) rec[1 + r] = pos;
            }
        }
    }
}
__global__ void __launch_bounds__(WG) k_zsplit_chain(InflateJob* q, uint32_t* cnt, uint32_t cap, DecodeArgs a) {
    __shared__ uint32_t s_pos[ZS_MAXC], s_fc[ZS_MAXC];
    __shared__ uint16_t s_nxt[ZS_MAXC], s_path[ZS_MAXC], s_jb[ZS_MAXC], s_anc[ZS_MAXC / 64];
    __shared__ uint32_t s_w[8], s_state[4];   // s_state: [0] fail, [1] frames on the chain, [2] queue base
    __shared__ unsigned long long s_w64[4];
    const uint32_t e = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
    if (e >= min(a.zs_hdr[0], ZS_LIST)) return;
    const uint32_t j = a.zs_hdr[16 + 2 * e], seg0 = a.zs_hdr[17 + 2 * e];
    if (j == 0xFFFFFFFFu) return;
    const InflateJob job = q[j];
    const uint32_t nseg = (job.csize + ZS_SEG - 1) / ZS_SEG;
    constexpr uint32_t DEAD = 0xFFFF, END = 0xFFFE;
    if (t == 0) s_state[0] = 0;
    __syncthreads();
    // ---- candidates in position order
    uint32_t C = 0;
    for (uint32_t g0 = 0; g0 < nseg; g0 += WG) {
        const uint32_t seg = g0 + t;
        uint32_t p[ZS_SLOTS], n = 0;
        if (seg < nseg) {
            const uint32_t* rec = a.zs_segs + (uint64_t)(seg0 + seg) * 8;
            n = rec[0];
            if (n > ZS_SLOTS) {
                s_state[0] = 1;
                n = 0;
            }
#pragma unroll
            for (uint32_t k = 0; k < ZS_SLOTS; k++) p[k] = k < n ? rec[1 + k] : 0xFFFFFFFFu;
#pragma unroll
            for (uint32_t x = 0; x < ZS_SLOTS; x++)   // (7 values: a few compare-exchanges)
#pragma unroll
                for (uint32_t y = 0; y + 1 < ZS_SLOTS - x; y++)
                    if (p[y] > p[y + 1]) {
                        const uint32_t tmp = p[y];
                        p[y] = p[y + 1];
                        p[y + 1] = tmp;
                    }
        }
        const uint32_t incl = wave_incl_scan(n);
        if (lane == 63) s_w[w] = incl;
        __syncthreads();
        uint32_t base = C + incl - n;
        for (uint32_t x = 0; x < w; x++) base += s_w[x];
        const uint32_t tot = s_w[0] + s_w[1] + s_w[2] + s_w[3];
        if (base + n <= ZS_MAXC) {
#pragma unroll
            for (uint32_t k = 0; k < ZS_SLOTS; k++)
                if (k < n) s_pos[base + k] = p[k];
        }
        C += tot;
        __syncthreads();
        if (C > ZS_MAXC) break;
    }
    if (t == 0 && (C > ZS_MAXC || C < 2 || s_pos[0] != 0)) s_state[0] = 1;
    __syncthreads();
    if (!s_state[0]) {
        // ---- extents, links
        for (uint32_t c = t; c < C; c += WG) {
            uint32_t fs = 0, fc = 0;
            uint32_t nx = DEAD;
            if (zstd_frame_extent_fast(job.src, job.csize, s_pos[c], &fs, &fc)) {
                const uint32_t target = s_pos[c] + fs;
                if (target == job.csize) {
                    nx = END;
                } else {
                    uint32_t lo = c, hi = C;   // positions ascend: the candidate at `target`, if any, is behind c
                    while (hi - lo > 1) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if (s_pos[mid] <= target) lo = mid;
                        else hi = mid;
                    }
                    if (s_pos[lo] == target && lo != c) nx = lo;
                }
            }
            s_nxt[c] = (uint16_t)nx;
            s_fc[c] = fc;
        }
        __syncthreads();
        // ---- the chain from position 0.  One lane hopping frame by frame through LDS took 0.5 ms for the 2 930 frames of a 96 MB
        // column (a dependent LDS round trip per hop).  Links only go forward, so: 64-hop jumps for every candidate by pointer
        // doubling (six rounds, all threads), the anchors — every 64th frame of the chain — by one lane over those jumps, the
        // 64 frames behind each anchor by a thread each.
        {
            uint16_t* ja = s_path;   // (the path is written once the jumps are no longer read)
            for (uint32_t c = t; c < C; c += WG) ja[c] = s_nxt[c];
            __syncthreads();
            for (int r = 0; r < 6; r++) {
                const uint16_t* src = (r & 1) ? s_jb : ja;
                uint16_t* dst = (r & 1) ? ja : s_jb;
                for (uint32_t c = t; c < C; c += WG) {
                    const uint32_t n = src[c];
                    dst[c] = n >= END ? (uint16_t)n : src[n];
                }
                __syncthreads();
            }   // (six rounds: the 64-hop jumps are back in ja)
            if (t == 0) {
                uint32_t c = 0, na = 0;
                bool ok = true;
                for (;;) {
                    s_anc[na++] = (uint16_t)c;
                    const uint32_t n = ja[c];
                    if (n == END) break;
                    if (n == DEAD || na >= ZS_MAXC / 64) {
                        ok = false;
                        break;
                    }
                    c = n;
                }
                s_state[0] = ok ? 0u : 1u;
                s_state[1] = na;
            }
            __syncthreads();
        }
        if (!s_state[0]) {
            const uint32_t na = s_state[1];
            __syncthreads();   // (the last anchor's thread replaces the count)
            if (t < na) {
                uint32_t c = s_anc[t];
                for (uint32_t h = 0; h < 64; h++) {
                    s_path[64 * t + h] = (uint16_t)c;
                    const uint32_t n = s_nxt[c];
                    if (n >= END) {   // (only behind the last anchor: the others are 64 hops away from the next one)
                        if (t == na - 1) s_state[1] = 64 * t + h + 1;
                        break;
                    }
                    c = n;
                }
            }
            __syncthreads();
            const uint32_t k = s_state[1];
            unsigned long long mine = 0;
            for (uint32_t i = t; i < k; i += WG) mine += s_fc[s_path[i]];
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) mine += __shfl_down(mine, d, 64);
            if (lane == 0) s_w64[w] = mine;
            __syncthreads();
            if (t == 0) {
                // (content sizes that add up to the buffer's size: no partial sum can be larger)
                const bool ok = k >= 2 && s_w64[0] + s_w64[1] + s_w64[2] + s_w64[3] == (unsigned long long)job.out_len;
                if (ok) {
                    const uint32_t b = atomicAdd(cnt, k);
                    s_state[3] = b + k > cap ? 1u : 0u;   // no room: the slots inside the queue are marked "skip" below, the buffer keeps its frames
                    s_state[2] = b;
                }
                s_state[0] = ok ? 0u : 1u;
            }
            __syncthreads();
        }
    }
    if (s_state[0]) {   // not a plain chain of frames (or too many of them): the one-lane walk decides
        if (t == 0) zstd_split_walk(q, cnt, cap, j, job);
        return;
    }
    const uint32_t nf = s_state[1], qb = s_state[2];
    if (s_state[3]) {
        for (uint32_t k = qb + t; k < cap && k < qb + nf; k += WG) q[k].codec = CODEC_SPLIT;
        return;
    }
    // ---- one queue entry per frame: destinations = prefix sums of the content sizes along the chain
    uint32_t run = 0;
    for (uint32_t k0 = 0; k0 < nf; k0 += WG) {
        const uint32_t k = k0 + t;
        const uint32_t c = k < nf ? s_path[k] : 0u;
        const uint32_t fc = k < nf ? s_fc[c] : 0u;
        const uint32_t incl = wave_incl_scan(fc);
        __syncthreads();
        if (lane == 63) s_w[w] = incl;
        __syncthreads();
        uint32_t at = run + incl - fc;
        for (uint32_t x = 0; x < w; x++) at += s_w[x];
        if (k < nf) {
            const uint32_t pos = s_pos[c];
            const uint32_t end = k + 1 < nf ? s_pos[s_path[k + 1]] : job.csize;
            InflateJob f = job;
            f.src = job.src + pos;
            f.dst = job.dst + at;
            f.csize = end - pos;
            f.out_len = fc;
            q[qb + k] = f;
        }
        run += s_w[0] + s_w[1] + s_w[2] + s_w[3];
    }
    if (t == 0) q[j].codec = CODEC_SPLIT;
}

__global__ void __launch_bounds__(WG) k_zstd_split(InflateJob* q, uint32_t* cnt, const uint32_t* n0_p, uint32_t cap, Status* st, DecodeArgs a) {
    const uint32_t n0 = *n0_p;
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n0 || j >= cap) return;
    {
        const InflateJob job = q[j];
        if (job.codec == SB_CODEC_LZ4 && job.csize >= (2u << 20) && !(st->kinds & KIND_LZ4_GIANT)) atomicOr(&st->kinds, KIND_LZ4_GIANT);
        if ((job.codec & ~JOB_REL) != SB_CODEC_ZSTD) return;
        if (!(st->kinds & KIND_ZSTD)) atomicOr(&st->kinds, KIND_ZSTD);   // (the host sizes the block pipeline's pools for later calls)
        if (a.zb_skipped && job.csize >= (1u << 20)) {   // a long buffer and no pipeline: not decoded now, the interval is replayed with it
            atomicOr(&st->kinds, KIND_REPLAY);
            q[j].codec = CODEC_SPLIT;
            return;
        }
        if (a.zs_segs && job.csize >= ZS_BIG) {   // a long buffer: listed for the scan kernels when there is room
            const uint32_t slot = atomicAdd(&a.zs_hdr[0], 1u);
            if (slot < ZS_LIST) {
                const uint32_t nseg = (job.csize + ZS_SEG - 1) / ZS_SEG;
                const uint32_t s0 = atomicAdd(&a.zs_hdr[1], nseg);
                if (s0 + nseg <= a.zs_seg_cap) {
                    a.zs_hdr[16 + 2 * slot] = j;
                    a.zs_hdr[17 + 2 * slot] = s0;
                    return;
                }
                a.zs_hdr[16 + 2 * slot] = 0xFFFFFFFFu;
            }
        }
        zstd_split_walk(q, cnt, cap, j, job);
    }
}
// true in every thread of the LAST workgroup of the grid to get here (all threads of every workgroup must call it)
__device__ bool last_workgroup_done(uint32_t* counter) {
    __shared__ uint32_t s_last_wg;
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) s_last_wg = atomicAdd(counter, 1u) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (s_last_wg) __threadfence();
    return s_last_wg != 0;
}

// -------------------------------------------------------------------------------- parse
#ifdef SB_TIMELINE  // scripts/micro/expand_timeline.hip: s_memtime stamps of one tile's phases
__device__ unsigned long long* g_dtl;
#define DTL(p)                                                                                           \
    do {                                                                                                 \
        if (g_dtl && blockIdx.x == 5000 && threadIdx.x == 0) g_dtl[(p)] = __builtin_readcyclecounter(); \
    } while (0)
// k_plan: workgroup 100, thread 0: g_dtl[16 + p]
#define PTL(p)                                                                                                \
    do {                                                                                                      \
        if (g_dtl && blockIdx.x == 100 && threadIdx.x == 0) g_dtl[16 + (p)] = __builtin_readcyclecounter(); \
    } while (0)
// k_inflate: wave 0, lane 0: g_dtl[32 + p] += cycles since the last stamp (scripts/micro/inflate_timeline.hip)
#define ITL_BEGIN unsigned long long itl_t = __builtin_readcyclecounter();
#define ITL(p)                                                                          \
    do {                                                                                \
        const unsigned long long n_ = __builtin_readcyclecounter();                     \
        if (g_dtl && blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) g_dtl[32 + (p)] += n_ - itl_t; \
        itl_t = n_;                                                                     \
    } while (0)
#else
#define DTL(p)
#define PTL(p)
#define ITL_BEGIN
#define ITL(p)
#endif

// primitive RLE pages of <= 8-byte values are expanded by one workgroup per page (k_expand_rle)
__device__ __forceinline__ bool rle_by_page(const ColDesc& c, const PageDesc& d) {
    return d.ok && d.codec == SB_CODEC_RLE && c.ptype != SB_TYPE_BOOLEAN && c.ptype != SB_TYPE_NULL && !is_binary(c.ptype) &&
           c.width <= 8;
}

// (what a page asks of the wave afterwards: its tile entries)
struct TileReq {
    uint32_t need, base, ntiles, col;
};
__device__ __forceinline__ TileReq parse_page(const DecodeArgs& a, const uint32_t p) {
    const PageTask t = a.tasks[p];
    const ColDesc c = a.cols[t.col];
    PageDesc d;
    __builtin_memset(&d, 0, sizeof(d));
    d.icodec = 255;
    const uint64_t N = t.num_values;
    const uint32_t ntiles = (uint32_t)((N + TILE_ROWS - 1) / TILE_ROWS);
#define FAIL(code, tag)                   \
    do {                                  \
        raise(a.status, (code), p, (tag)); \
        a.descs[p] = d;                   \
        return TileReq{0, 0, 0, 0};       \
    } while (0)
    if (c.ptype == SB_TYPE_NULL) {  // empty pages (src/read/array/null.rs:48-52)
        a.descs[p] = d;
        return TileReq{0, 0, 0, 0};
    }
    if (t.in_off + t.length > c.pages_len) FAIL(SB_ERR_IO, 1);
    const uint8_t* cur = c.pages + t.in_off;
    const uint8_t* end = cur + t.length;
    // ---- def-level section: u32 def_len | ULEB128((ceil(N/8)<<1)|1) | bits  (read_basic.rs:36-63)
    if (c.nullable) {
        if (end - cur < 4) FAIL(SB_ERR_IO, 2);
        const uint32_t def_len = ldu32(cur);
        cur += 4;
        if ((uint64_t)(end - cur) < def_len) FAIL(SB_ERR_IO, 3);
        if (def_len == 0) {
            if (N != 0) FAIL(SB_ERR_OUT_OF_SPEC, 4);  // reference: validity length mismatch
        } else {
            uint64_t ind = 0;
            uint32_t sh = 0, k = 0;
            for (;;) {
                if (k >= def_len || k >= 10) FAIL(SB_ERR_OUT_OF_SPEC, 5);
                uint8_t b = cur[k++];
                ind |= (uint64_t)(b & 0x7F) << sh;
                sh += 7;
                if (!(b & 0x80)) break;
            }
            if (!(ind & 1)) FAIL(SB_ERR_OUT_OF_SPEC, 6);  // RLE run: unreachable!() upstream
            uint64_t nbytes = ind >> 1;
            if (nbytes > def_len - k) nbytes = def_len - k;
            if (nbytes * 8 < N) FAIL(SB_ERR_OUT_OF_SPEC, 7);
            d.def_bits = cur + k;
        }
        cur += def_len;
    }
    // ---- first block header
    if (end - cur < 9) FAIL(SB_ERR_IO, 8);
    d.codec = cur[0];
    d.csize = ldu32(cur + 1);
    d.usize = ldu32(cur + 5);
    cur += 9;
    d.body = cur;
    d.src = cur;
    if ((uint64_t)(end - cur) < d.csize) FAIL(SB_ERR_IO, 9);
    const uint32_t codec = d.codec;
    if (!(codec <= 3 || (codec >= 10 && codec <= 16))) FAIL(SB_ERR_OUT_OF_SPEC, 10);
    uint8_t* infl = a.scratch + t.infl_off;

    if (c.ptype == SB_TYPE_BOOLEAN) {
        const uint64_t nbytes = (N + 7) / 8;
        if (codec == SB_CODEC_NONE) {
            if (d.csize != nbytes) FAIL(SB_ERR_OUT_OF_SPEC, 11);
        } else if (is_basic(codec)) {
            // (payloads that no planning step reads are inflated in queue B when it runs, next to the value blocks of the
            // binary columns, instead of in front of them: the phases of a call are serial, their jobs are not)
            if (!a.sizes_only) push_payload(a, d.body, d.csize, infl, (uint32_t)nbytes, codec, p);
            d.src = infl;
        } else if (codec == SB_CODEC_ONEVALUE) {
            if (d.csize < 1) FAIL(SB_ERR_OUT_OF_SPEC, 12);
        } else if (codec != SB_CODEC_RLE) {
            FAIL(SB_ERR_OUT_OF_SPEC, 13);
        }
    } else if (is_binary(c.ptype)) {
        const uint32_t ow = c.width;
        if (is_basic(codec)) {
            // BLOCK(offsets) | hdr9 | BLOCK(values)   (binary/mod.rs:119-173)
            const uint64_t obytes = (N + 1) * ow;
            if (codec == SB_CODEC_NONE) {
                if (d.csize != obytes) FAIL(SB_ERR_OUT_OF_SPEC, 14);
            } else {   // (the sizing pass reads the page's value bytes from the second header below: no need for the offsets)
                if (!a.sizes_only) push_job(a.jobs_a, a.job_counts, d.body, d.csize, infl, (uint32_t)obytes, codec, p);
                d.src = infl;
            }
            const uint8_t* h2 = d.body + d.csize;
            if (end - h2 < 9) FAIL(SB_ERR_IO, 15);
            d.vcsize = ldu32(h2 + 1);
            d.vusize = ldu32(h2 + 5);
            d.vbody = h2 + 9;
            if ((uint64_t)(end - d.vbody) < d.vcsize) FAIL(SB_ERR_IO, 16);
            if (codec == SB_CODEC_NONE && d.vcsize != d.vusize) FAIL(SB_ERR_OUT_OF_SPEC, 17);
            d.val_bytes = d.vusize;
            // a Zstd value block is queued HERE (queue Z, dst relative to the page's value base: k_colscan places the page)
            if (codec == SB_CODEC_ZSTD && a.jobs_z && !a.sizes_only)
                push_job(a.jobs_z, a.job_counts + 10, d.vbody, d.vcsize, nullptr, d.vusize, SB_CODEC_ZSTD | JOB_REL, p);
        } else if (codec == SB_CODEC_ONEVALUE) {  // u32 len | bytes  (binary/one_value.rs:70-97)
            if (d.csize < 4) FAIL(SB_ERR_IO, 18);
            d.dict_n = ldu32(d.body);
            d.dict = d.body + 4;
            if ((uint64_t)(end - d.dict) < d.dict_n) FAIL(SB_ERR_OUT_OF_SPEC, 19);
            d.val_bytes = (uint64_t)d.dict_n * N;
        } else if (codec == SB_CODEC_DICT) {
            // handled below (shared with primitives)
        } else if (codec == SB_CODEC_FREQ) {
            // u64 top_len | top | u32 rb_size | Roaring | per exception `u64 len | bytes` (binary/freq.rs:103-145).
            // k_plan turns the page into a virtual Dict page: entry 0 = the top value, entry 1 + k = the k-th
            // exception record, index of a row = 0 or 1 + its rank in the bitmap.
            if ((uint64_t)(end - d.body) < 8) FAIL(SB_ERR_IO, 40);
            const uint64_t top_len = ldu64(d.body);
            if ((uint64_t)(end - d.body) - 8 < top_len) FAIL(SB_ERR_OUT_OF_SPEC, 41);  // "data size is less than"
            const uint8_t* rbp = d.body + 8 + top_len;
            if (end - rbp < 4) FAIL(SB_ERR_IO, 42);
            const uint32_t rb_size = ldu32(rbp);
            if ((uint64_t)(end - rbp) - 4 < rb_size) FAIL(SB_ERR_IO, 43);
            if (rb_size < 8) FAIL(SB_ERR_EXTERNAL, 44);
            d.dict = d.body;
            d.vbody = rbp + 4;
            d.vcsize = rb_size;
            d.vusize = (uint32_t)top_len;
            d.icodec = SB_CODEC_NONE;
            d.isrc = infl;  // u32 index per row, written by k_plan
        } else {
            FAIL(SB_ERR_OUT_OF_SPEC, 20);  // "Unknown compression codec ... for binary"
        }
    } else {
        const uint32_t w = c.width;
        if (codec == SB_CODEC_NONE) {
            if (d.csize != N * w) FAIL(SB_ERR_OUT_OF_SPEC, 21);
        } else if (is_basic(codec)) {
            // inflate straight into the column's values buffer (integer/mod.rs:97-107)
            if (!a.sizes_only) push_payload(a, d.body, d.csize, c.values + t.out_row * w, (uint32_t)(N * w), codec, p);
            d.src = nullptr;  // nothing left for expand
        } else if (codec == SB_CODEC_ONEVALUE) {
            if (d.csize < w) FAIL(SB_ERR_IO, 22);
        } else if (codec == SB_CODEC_BITPACKING || codec == SB_CODEC_DELTA_BITPACKING) {
            if (w != 4 || c.ptype == SB_TYPE_FLOAT32) FAIL(SB_ERR_OUT_OF_SPEC, 23);
            if (N % 128 != 0) FAIL(SB_ERR_OUT_OF_SPEC, 24);  // whole blocks only (bp.rs:72-84)
        } else if (codec == SB_CODEC_RLE || codec == SB_CODEC_DICT) {
        } else if (codec == SB_CODEC_PATAS) {
            if (c.ptype == SB_TYPE_FLOAT32) FAIL(SB_ERR_NYI, 26);   // f32 Patas decode is broken upstream
            if (c.ptype != SB_TYPE_FLOAT64) FAIL(SB_ERR_OUT_OF_SPEC, 27);  // "Unknown compression codec Patas for integer"
            if (!a.sizes_only) push_payload(a, d.body, d.csize, c.values + t.out_row * w, (uint32_t)(N * w), codec, p);
            d.src = nullptr;
        } else if (codec == SB_CODEC_FREQ) {  // top[w] | u32 rb_size | roaring | BLOCK<T exceptions>  (freq.rs:71-83)
            if (a.no_freq) FAIL(SB_ERR_OUT_OF_SPEC, 28);
            if ((uint64_t)(end - d.body) < (uint64_t)w + 4) FAIL(SB_ERR_IO, 29);
            const uint32_t rb_size = ldu32(d.body + w);
            const uint8_t* rb = d.body + w + 4;
            if ((uint64_t)(end - rb) < rb_size) FAIL(SB_ERR_IO, 33);
            // cardinality from the portable header: cookie, container count, (key, cardinality - 1) pairs
            if (rb_size < 8) FAIL(SB_ERR_EXTERNAL, 34);
            const uint32_t cookie = ldu32(rb);
            uint32_t nc, hp;
            if ((cookie & 0xFFFF) == 12347) {
                nc = (cookie >> 16) + 1;
                hp = 4 + (nc + 7) / 8;
            } else if (cookie == 12346) {
                nc = ldu32(rb + 4);
                hp = 8;
            } else {
                FAIL(SB_ERR_EXTERNAL, 35);
            }
            if (nc > 65536 || (uint64_t)hp + 4ull * nc > rb_size) FAIL(SB_ERR_EXTERNAL, 36);
            uint64_t card = 0;
            for (uint32_t k = 0; k < nc; k++) card += (uint64_t)ldu16(rb + hp + 4 * k + 2) + 1;
            if (card > N) FAIL(SB_ERR_OUT_OF_SPEC, 37);
            const uint8_t* nested = rb + rb_size;
            if (end - nested < 9) FAIL(SB_ERR_IO, 38);
            if (!a.freq_log) {  // header-only pass (sb_read_columns_sizes): nothing to log
                d.ok = 1;
                a.descs[p] = d;
                return TileReq{0, 0, 0, 0};
            }
            const uint32_t slot = atomicAdd(a.freq_count, 1u);
            if (slot >= a.freq_cap) FAIL(SB_ERR_INVALID, 39);
            FreqEntry fe;
            fe.roaring = rb;
            fe.nested = nested;
            fe.out = c.values + t.out_row * w;
            fe.nested_len = (uint64_t)(end - nested);
            fe.rows = N;
            fe.roaring_len = rb_size;
            fe.n_exceptions = (uint32_t)card;
            fe.ptype = c.ptype;
            fe.width = w;
            fe.page = p;
            fe.pad = 0;
            a.freq_log[slot] = fe;
        } else {
            FAIL(SB_ERR_OUT_OF_SPEC, 25);
        }
    }
    if (codec == SB_CODEC_DICT && c.ptype != SB_TYPE_BOOLEAN) {
        // BLOCK<u32 indices> | u32 n | entries   (integer/dict.rs:75-103, binary/dict.rs:95-140)
        if (d.csize < 9) FAIL(SB_ERR_IO, 26);
        d.icodec = d.body[0];
        d.icsize = ldu32(d.body + 1);
        d.ibody = d.body + 9;
        d.isrc = d.ibody;
        if ((uint64_t)(end - d.ibody) < (uint64_t)d.icsize + 4) FAIL(SB_ERR_IO, 27);
        const uint32_t ic = d.icodec;
        if (ic == SB_CODEC_NONE) {
            if (d.icsize != N * 4) FAIL(SB_ERR_OUT_OF_SPEC, 28);
        } else if (is_basic(ic)) {
            push_job(a.jobs_a, a.job_counts, d.ibody, d.icsize, infl, (uint32_t)(N * 4), ic, p);
            d.isrc = infl;
        } else if (ic == SB_CODEC_BITPACKING || ic == SB_CODEC_DELTA_BITPACKING) {
            if (N % 128 != 0) FAIL(SB_ERR_OUT_OF_SPEC, 29);
        } else if (ic == SB_CODEC_ONEVALUE) {
            if (d.icsize < 4) FAIL(SB_ERR_IO, 30);
        } else if (ic == SB_CODEC_FREQ) {
            // The u32 indices as a Freq block: top[4] | u32 rb_size | Roaring | BLOCK<u32 exceptions> (integer/freq.rs:71-83).
            // The reference writes this for a column that is mostly one value but may not use Freq itself (integers whose
            // maximum is below 256, freq.rs:146).  k_plan materialises the N indices in the inflate area; the exceptions
            // block cannot be Dict or Freq again (both forbidden by then) and is taken as plain / inflated / one-value.
            if (d.icsize < 8) FAIL(SB_ERR_IO, 45);
            const uint8_t* fend = d.ibody + d.icsize;
            const uint32_t rb_size = ldu32(d.ibody + 4);
            const uint8_t* rb = d.ibody + 8;
            if ((uint64_t)(fend - rb) < (uint64_t)rb_size + 9) FAIL(SB_ERR_IO, 46);
            if (rb_size < 8) FAIL(SB_ERR_EXTERNAL, 47);
            const uint32_t cookie = ldu32(rb);
            uint32_t nc, hp;
            if ((cookie & 0xFFFF) == 12347) {
                nc = (cookie >> 16) + 1;
                hp = 4 + (nc + 7) / 8;
            } else if (cookie == 12346) {
                nc = ldu32(rb + 4);
                hp = 8;
            } else {
                FAIL(SB_ERR_EXTERNAL, 48);
            }
            if (nc > 65536 || (uint64_t)hp + 4ull * nc > rb_size) FAIL(SB_ERR_EXTERNAL, 49);
            uint64_t card = 0;
            for (uint32_t k = 0; k < nc; k++) card += (uint64_t)ldu16(rb + hp + 4 * k + 2) + 1;
            if (card > N) FAIL(SB_ERR_OUT_OF_SPEC, 50);
            const uint8_t* eh = rb + rb_size;
            const uint32_t ec = eh[0], ecsize = ldu32(eh + 1);
            const uint8_t* ebody = eh + 9;
            if ((uint64_t)(fend - ebody) < ecsize) FAIL(SB_ERR_IO, 51);
            d.vbody = rb;
            d.vcsize = rb_size;
            d.vusize = (uint32_t)card;
            d.pad = (uint8_t)ec;
            if (ec == SB_CODEC_NONE) {
                if (ecsize != card * 4) FAIL(SB_ERR_OUT_OF_SPEC, 52);
                d.src = ebody;
            } else if (is_basic(ec)) {
                uint8_t* exd = infl + ((N * 4 + 15) & ~(uint64_t)15);
                push_job(a.jobs_a, a.job_counts, ebody, ecsize, exd, (uint32_t)(card * 4), ec, p);
                d.src = exd;
            } else if (ec == SB_CODEC_ONEVALUE) {
                if (ecsize < 4) FAIL(SB_ERR_IO, 53);
                d.src = ebody;
            } else if (ec == SB_CODEC_RLE || ec == SB_CODEC_BITPACKING || ec == SB_CODEC_DELTA_BITPACKING) {
                if (ec != SB_CODEC_RLE && card % 128 != 0) FAIL(SB_ERR_OUT_OF_SPEC, 55);  // whole blocks only (bp.rs:72-84)
                d.src = ebody;  // k_plan expands the block into the inflate area first
            } else {
                FAIL(SB_ERR_OUT_OF_SPEC, 54);
            }
            d.isrc = infl;
        } else if (ic != SB_CODEC_RLE) {
            FAIL(SB_ERR_OUT_OF_SPEC, 31);  // Dict inside Dict is never written (integer/dict.rs:60-62)
        }
        const uint8_t* q = d.ibody + d.icsize;
        d.dict_n = ldu32(q);
        d.dict = q + 4;
        if (!is_binary(c.ptype)) {
            if ((uint64_t)(end - d.dict) < (uint64_t)d.dict_n * c.width) FAIL(SB_ERR_OUT_OF_SPEC, 32);
        }
    }
    d.ok = 1;
    // Tile tasks for k_expand / k_expand_binary, appended to a compact list (job_counts[2] entries):
    // pages that a page-level kernel expands (k_expand_rle) contribute none, so a batch of RLE pages
    // does not launch tens of thousands of workgroups that only find out they have nothing to do.
    // pages that k_plan / k_expand_rle have work for: those kernels return at once when there are none
    // (one atomic per wave: thousands of pages adding to one word serialise)
    {
        const bool by_page = rle_by_page(c, d);
        const bool plan = !by_page && (is_binary(c.ptype) || codec == SB_CODEC_DICT || codec == SB_CODEC_RLE ||
                                       codec == SB_CODEC_BITPACKING || codec == SB_CODEC_DELTA_BITPACKING);
        const uint64_t mb = __ballot(by_page), mp = __ballot(plan), act = __ballot(true);
        if ((threadIdx.x & 63) == (uint32_t)(__ffsll((long long)act) - 1)) {
            if (mb) atomicAdd(&a.job_counts[4], (uint32_t)__popcll(mb));
            if (mp) atomicAdd(&a.job_counts[3], (uint32_t)__popcll(mp));
        }
    }
    const bool need_tiles = !rle_by_page(c, d) && ntiles;
    uint32_t tbase = 0;
    if (need_tiles) {
        tbase = atomicAdd(&a.job_counts[2], ntiles);
        d.tile_base = tbase;
    }
    a.descs[p] = d;
    return TileReq{need_tiles ? 1u : 0u, tbase, ntiles, t.col};
#undef FAIL
}
__global__ void __launch_bounds__(WG) k_parse(DecodeArgs a) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    TileReq tr{0, 0, 0, 0};
    if (p < a.n_pages) tr = parse_page(a, p);
    // the tile entries of the wave's pages, written by ALL lanes of the wave together — also the lanes without a page (a 1 M-row
    // page has 245 tiles: one thread writing them one by one was 44 us of a 0.46 ms C1 decode; a call with ONE 12 M-row
    // page has one lane with a page: 2 930 entries, 0.14 ms of a 0.19 ms decode)
    {
        uint64_t m = __ballot(tr.need != 0);
        const uint32_t lane = threadIdx.x & 63;
        while (m) {
            const int l = __ffsll((long long)m) - 1;
            m &= m - 1;
            const uint32_t b = __shfl(tr.base, l, 64), nt = __shfl(tr.ntiles, l, 64), pg = __shfl(p, l, 64), col = __shfl(tr.col, l, 64);
            for (uint32_t i = lane; i < nt; i += 64) {
                TileTask tt;
                tt.page = pg;
                tt.tile = i;
                tt.col = col;
                tt.k0 = tt.kend = tt.pad = 0;
                a.tiles[b + i] = tt;
            }
        }
    }
    // queue A is complete when the last workgroup is done: its length goes to job_counts[8] for k_zstd_split
    if (last_workgroup_done(&a.job_counts[5]) && threadIdx.x == 0) {
        a.job_counts[8] = __hip_atomic_load(&a.job_counts[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a.job_counts[11] = __hip_atomic_load(&a.job_counts[10], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (queue Z is complete too)
    }
}

// -------------------------------------------------------------------------------- inflate (LZ4)
// One wave per block.  Sequences are parsed wave-uniformly (every lane walks the same token
// stream); literal and match copies are spread over the 64 lanes.  A match may overlap its own
// output (offset < length): lane i reads history byte (i mod offset), which was written by an
// earlier sequence, so the copy is still fully parallel.
__device__ void lz4_inflate_wave(const InflateJob& j, Status* st) {
    const int lane = threadIdx.x & 63;
    const uint8_t* src = j.src;
    uint8_t* dst = j.dst;
    const uint32_t n = j.csize, out_len = j.out_len;
    uint32_t ip = 0, op = 0;
    if (n == 0) {
        if (out_len != 0 && lane == 0) raise(st, SB_ERR_EXTERNAL, j.page, 100);
        return;
    }
    for (;;) {
        if (ip >= n) {
            if (lane == 0) raise(st, SB_ERR_EXTERNAL, j.page, 101);
            return;
        }
        const uint32_t token = src[ip++];
        uint32_t lit = token >> 4;
        if (lit == 15) {
            uint32_t s;
            do {
                if (ip >= n) {
                    if (lane == 0) raise(st, SB_ERR_EXTERNAL, j.page, 102);
                    return;
                }
                s = src[ip++];
                lit += s;
            } while (s == 255);
        }
        if (lit > n - ip || lit > out_len - op) {
            if (lane == 0) raise(st, SB_ERR_EXTERNAL, j.page, 103);
            return;
        }
        for (uint32_t i = lane; i < lit; i += 64) dst[op + i] = src[ip + i];
        ip += lit;
        op += lit;
        if (ip == n) break;  // last sequence carries literals only
        if (n - ip < 2) {
            if (lane == 0) raise(st, SB_ERR_EXTERNAL, j.page, 104);
            return;
        }
        const uint32_t off = (uint32_t)src[ip] | ((uint32_t)src[ip + 1] << 8);
        ip += 2;
        uint32_t ml = token & 15;
        if (ml == 15) {
            uint32_t s;
            do {
                if (ip >= n) {
                    if (lane == 0) raise(st, SB_ERR_EXTERNAL, j.page, 105);
                    return;
                }
                s = src[ip++];
                ml += s;
            } while (s == 255);
        }
        ml += 4;
        if (off == 0 || off > op || ml > out_len - op) {
            if (lane == 0) raise(st, SB_ERR_EXTERNAL, j.page, 106);
            return;
        }
        // make this wave's earlier stores visible to its loads
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const uint8_t* hist = dst + op - off;
        for (uint32_t i = lane; i < ml; i += 64) dst[op + i] = hist[off >= ml ? i : i % off];
        op += ml;
    }
    if (op != out_len && lane == 0) raise(st, SB_ERR_EXTERNAL, j.page, 107);
}

// one wave per workgroup; the grid is a fixed pool of waves that loops over the job queue, so the
// per-wave Zstd literal buffers (zlit) are a fixed pool too
constexpr uint32_t ZLIT_STRIDE = 128 * 1024 + 64;
// Snappy raw format (basic.rs:99-106 -> snap::raw::Decoder [3P]): uvarint length, then literal /
// copy elements.  Same shape as the LZ4 decoder: elements are parsed wave-uniformly, the byte
// copies are spread over the 64 lanes.
__device__ void snappy_inflate_wave(const InflateJob& j, Status* st) {
    const int lane = threadIdx.x & 63;
    const uint8_t* src = j.src;
    uint8_t* dst = j.dst;
    const uint32_t n = j.csize, out_len = j.out_len;
    uint32_t ip = 0, op = 0;
    auto bad = [&](uint32_t tag) {
        if (lane == 0) raise(st, SB_ERR_EXTERNAL, j.page, tag);
    };
    uint64_t ulen = 0;
    for (uint32_t sh = 0;; sh += 7) {
        if (ip >= n || sh > 28) return bad(130);
        const uint32_t b = src[ip++];
        ulen |= (uint64_t)(b & 0x7F) << sh;
        if (!(b & 0x80)) break;
    }
    if (ulen != out_len) return bad(131);
    while (ip < n) {
        const uint32_t tag = src[ip++];
        uint32_t len, off = 0;
        if ((tag & 3) == 0) {  // literal
            len = tag >> 2;
            if (len >= 60) {
                const uint32_t nb = len - 59;
                if (n - ip < nb) return bad(132);
                len = 0;
                for (uint32_t k = 0; k < nb; k++) len |= (uint32_t)src[ip + k] << (8 * k);
                ip += nb;
            }
            len += 1;
            if (len > n - ip || len > out_len - op) return bad(133);
            for (uint32_t i = lane; i < len; i += 64) dst[op + i] = src[ip + i];
            ip += len;
            op += len;
            continue;
        }
        if ((tag & 3) == 1) {
            if (n - ip < 1) return bad(134);
            len = ((tag >> 2) & 7) + 4;
            off = ((tag >> 5) << 8) | src[ip];
            ip += 1;
        } else if ((tag & 3) == 2) {
            if (n - ip < 2) return bad(135);
            len = (tag >> 2) + 1;
            off = (uint32_t)src[ip] | ((uint32_t)src[ip + 1] << 8);
            ip += 2;
        } else {
            if (n - ip < 4) return bad(136);
            len = (tag >> 2) + 1;
            off = (uint32_t)src[ip] | ((uint32_t)src[ip + 1] << 8) | ((uint32_t)src[ip + 2] << 16) | ((uint32_t)src[ip + 3] << 24);
            ip += 4;
        }
        if (off == 0 || off > op || len > out_len - op) return bad(137);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // this wave's earlier stores -> its loads
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const uint8_t* hist = dst + op - off;
        for (uint32_t i = lane; i < len; i += 64) dst[op + i] = hist[off >= len ? i : i % off];
        op += len;
    }
    if (op != out_len) bad(138);
}

// Patas pages of f64 (double/patas.rs:106-133): first value, then per value `u16 packed | sig bytes`,
// packed = ref_diff << 9 | (sig_bytes & 7) << 6 | trailing_zeros; value = (bits << tz) ^ out[i - diff].
// Record positions are a pointer chase and references may chain, so a wave takes 64 values at a time:
// the window of input bytes is staged in LDS, lane 0 walks the record boundaries there, every lane
// then decodes its own record, and references are resolved in rounds (a lane is done once the lane
// or earlier output it refers to is).  Runs as an inflate job straight into the column's values.
__device__ void patas_inflate_wave(const InflateJob& j, Status* st, uint8_t* s_win /* 64*10+8 */, uint16_t* s_pos /* 65 */) {
    const int lane = threadIdx.x & 63;
    const uint8_t* src = j.src;
    unsigned long long* out = (unsigned long long*)j.dst;
    const uint32_t n = j.csize;
    const uint64_t N = j.out_len / 8;
    auto bad = [&](uint32_t tag) {
        if (lane == 0) raise(st, SB_ERR_OUT_OF_SPEC, j.page, tag);
    };
    if (N == 0) return bad(140);  // upstream indexes the first value unconditionally
    if (n < 8) return bad(141);
    if (lane == 0) out[0] = ldu64(src);
    uint32_t ip = 8;
    for (uint64_t base = 1; base < N; base += 64) {
        const uint32_t nb = (uint32_t)min((uint64_t)64, N - base);
        const uint32_t wl = min(nb * 10, n - ip);  // bytes of input that can belong to this batch
        __builtin_amdgcn_wave_barrier();
        for (uint32_t i = lane; i < wl; i += 64) s_win[i] = src[ip + i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (lane == 0) {  // record boundaries
            uint32_t pos = 0;
            bool ok = true;
            for (uint32_t k = 0; k < nb; k++) {
                s_pos[k] = (uint16_t)pos;
                if (pos + 2 > wl) {
                    ok = false;
                    break;
                }
                const uint32_t pk = (uint32_t)s_win[pos] | ((uint32_t)s_win[pos + 1] << 8);
                uint32_t sb = (pk >> 6) & 7;
                if ((pk & 0x3F) < 63 && sb == 0) sb = 8;  // unpack (patas.rs:152-163)
                pos += 2 + sb;
                if (pos > wl) {
                    ok = false;
                    break;
                }
            }
            s_pos[64] = ok ? (uint16_t)pos : (uint16_t)0xFFFF;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const uint32_t used = s_pos[64];
        if (used == 0xFFFF) return bad(142);  // truncated page (read_exact EOF upstream)
        // my record
        const bool act = (uint32_t)lane < nb;
        uint64_t sval = 0;
        uint32_t diff = 1;
        if (act) {
            const uint32_t pos = s_pos[lane];
            const uint32_t pk = (uint32_t)s_win[pos] | ((uint32_t)s_win[pos + 1] << 8);
            diff = (pk >> 9) & 0x7F;
            uint32_t sb = (pk >> 6) & 7;
            const uint32_t tz = pk & 0x3F;
            if (tz < 63 && sb == 0) sb = 8;
            uint64_t v = 0;
            for (uint32_t b = 0; b < sb; b++) v |= (uint64_t)s_win[pos + 2 + b] << (8 * b);
            sval = v << tz;
        }
        const uint64_t i = base + lane;
        const bool bad_ref = act && (diff == 0 || diff > i);
        if (__ballot(bad_ref)) return bad(143);
        // references: earlier batches come from memory, in-batch ones from the lane that holds them
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const uint64_t ref = act ? i - diff : 0;
        const bool outside = ref < base;
        uint64_t x = 0;
        bool done = !act;
        if (act && outside) {
            x = sval ^ __hip_atomic_load(&out[ref], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            done = true;
        }
        const int srcl = act && !outside ? (int)(ref - base) : 0;
        for (;;) {
            const uint64_t dm = __ballot(done);
            if (dm == ~0ull) break;
            const uint64_t px = __shfl(x, srcl, 64);
            if (!done && ((dm >> srcl) & 1)) {
                x = sval ^ px;
                done = true;
            }
        }
        if (act) out[i] = x;
        ip += used;
    }
}

// One job of queue entries that k_inflate owns (all lanes, uniform arguments).  recs: the frame's pre-decoded sequences or null.
__device__ __forceinline__ void inflate_one(const InflateJob& j0, Status* st, ZWork& wk, uint8_t* zlit, uint8_t* s_win, uint16_t* s_pos,
                                            const uint64_t* recs, const RelCtx& rc) {
    InflateJob j = j0;
    if (j.codec & JOB_REL) {   // (queue Z: the page's place in the values buffer is known since k_colscan)
        j.dst = job_dst(rc, j.dst, j.page, true);
        j.codec &= ~JOB_REL;
        if (!j.dst) return;    // the page's values do not fit the caller's buffer: not expanded
    }
    if (j.codec == SB_CODEC_LZ4 || j.codec == CODEC_SPLIT || j.codec == CODEC_ZB) {
        // k_inflate_lz4 owns the LZ4 blocks; a split Zstd buffer is decoded through its frames' entries; the block pipeline
        // (sb_zstd_blocks.h) has decoded the frames it took
    } else if (j.codec == SB_CODEC_ZSTD) {
        zstd_inflate_wave(j.src, j.csize, j.dst, j.out_len, &wk, zlit, recs);
        if (threadIdx.x == 0 && wk.err) raise(st, SB_ERR_EXTERNAL, j.page, 120 + (uint32_t)wk.err);
        __syncthreads();
    } else if (j.codec == SB_CODEC_SNAPPY) {
        snappy_inflate_wave(j, st);
    } else if (j.codec == SB_CODEC_PATAS) {
        patas_inflate_wave(j, st, s_win, s_pos);
    } else if (threadIdx.x == 0) {
        raise(st, SB_ERR_OUT_OF_SPEC, j.page, 110);
    }
}

// A pool of waves over the job queue.  With few jobs every wave takes one job at a time.  With many (>= 4 per wave) a wave
// takes up to 64 consecutive jobs: the Zstd frames among them that qualify (sb_zstd.h, z_lane_frame) have their FSE
// sequence streams decoded LANE PER FRAME into the wave's record arena first — 64 serial state chains side by side
// instead of one after the other — and the wave then executes the jobs one by one from the records.
union InflateLds {   // the one-wave decoder's workspace and the lane-per-stream Huffman phase never live at the same time
    ZWork wk;
    ZHufLanes hl;
};
__global__ void __launch_bounds__(64) k_inflate(const InflateJob* jobs, const uint32_t* count, Status* st,
                                                uint8_t* zlit, uint64_t* zrec, RelCtx rc, uint32_t cap) {
    __shared__ InflateLds u;
    ZWork& wk = u.wk;
    __shared__ ZLaneTabs zt;
    __shared__ uint8_t s_win[64 * 10 + 8];
    __shared__ uint16_t s_pos[65];
    const uint32_t njobs = min(*count, cap);   // (k_zstd_split may have asked for slots beyond the queue's end)
    const uint32_t lane = threadIdx.x;
    if (threadIdx.x == 0) wk.pre_built = 0;
    uint8_t* my_lit = zlit + (uint64_t)blockIdx.x * ZLIT_STRIDE;
    // batch size: a multiple of the 16 frames the literals-only phase decodes together (a group of 2 frames costs a wave as
    // much time as a group of 16), at most the 64 lanes of the lane-per-frame phases
    // (from 2 jobs per pool wave on; the lane-per-frame sequence phases also need the record arena)
    const uint32_t B = njobs >= 2 * gridDim.x ? min(64u, max(16u, (njobs / gridDim.x) & ~15u)) : 1u;
    if (B > 1) {
        for (uint32_t i = lane; i < 64; i += 64) {
            zt.ll[i] = g_zpre.ll[i];
            zt.ml[i] = g_zpre.ml[i];
            zt.xll[i] = g_zpre.xll[i];
            zt.xml[i] = g_zpre.xml[i];
            if (i < 32) zt.of[i] = g_zpre.of[i];
        }
    }
    __syncthreads();
    if (B == 1) {
        for (uint32_t job = blockIdx.x; job < njobs; job += gridDim.x) inflate_one(jobs[job], st, wk, my_lit, s_win, s_pos, nullptr, rc);
        return;
    }
    uint64_t* arena = zrec ? zrec + (uint64_t)blockIdx.x * ZREC_PER_WAVE : nullptr;
    ITL_BEGIN
    for (uint32_t base = blockIdx.x * B; base < njobs; base += gridDim.x * B) {
        const uint32_t nb = min(B, njobs - base);
        InflateJob mine;
        mine.codec = 0xFFFFFFFFu;
        mine.src = nullptr;
        mine.dst = nullptr;
        mine.csize = mine.out_len = 0;
        if (lane < nb) {
            mine = jobs[base + lane];
            if (mine.codec & JOB_REL) {
                mine.dst = job_dst(rc, mine.dst, mine.page, true);
                mine.codec = mine.dst ? (mine.codec & ~JOB_REL) : CODEC_SPLIT;
            }
        }
        const bool zs = lane < nb && mine.codec == SB_CODEC_ZSTD;
        // ---- phase H: literals-only frames, 16 at a time, lane per stream (sb_zstd.h)
        ZHufFrame hf;
        hf.ls = nullptr;
        hf.dst = mine.dst;
        hf.lleft = hf.regen = 0;
        const bool lit_only = zs && z_lane_litonly(mine.src, mine.csize, mine.out_len, &hf.ls, &hf.lleft, &hf.regen);
        ITL(0);
        uint64_t todo = __ballot(lit_only);
        uint64_t done_m = 0;
        if (todo) {
            ZHufLanes& H = u.hl;
            while (todo) {
                // the lowest ZH_GROUP frames of `todo`: frame of lane L gets slot g = its rank
                const uint32_t my_g = (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(todo >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)todo, 0));
                const bool in_group = ((todo >> lane) & 1) && my_g < ZH_GROUP;
                const uint64_t gm = __ballot(in_group);
                if (in_group) H.fr[my_g] = hf;
                const uint32_t ng = (uint32_t)__popcll(gm);
                __syncthreads();
                ITL(1);
                if (lane < ng) z_lane_huf_table(H, lane);   // (the weights, lane per frame)
                __syncthreads();
                z_wave_huf_fill(H, ng);
                __syncthreads();
                ITL(2);
                bool ok = true, act = false;
                const uint32_t g = lane >> 2, j = lane & 3;
                const uint8_t* sp = nullptr;
                uint8_t* sd = nullptr;
                uint32_t sn = 0, outn = 0, mbits = 0;
                if (g < ng && H.bits[g]) {
                    act = true;
                    const ZHufFrame f = H.fr[g];
                    const uint8_t* q = f.ls + H.str0[g];
                    const uint32_t left = f.lleft - H.str0[g];
                    const uint32_t s1 = ldu16(q), s2 = ldu16(q + 2), s3 = ldu16(q + 4);
                    const uint32_t per = (f.regen + 3) / 4;
                    if (6 + s1 + s2 + s3 > left || per * 3 > f.regen) {
                        ok = false;
                    } else {
                        const uint32_t so = j == 0 ? 0u : j == 1 ? s1 : j == 2 ? s1 + s2 : s1 + s2 + s3;
                        sn = j == 0 ? s1 : j == 1 ? s2 : j == 2 ? s3 : left - 6 - s1 - s2 - s3;
                        outn = j < 3 ? per : f.regen - 3 * per;
                        sp = q + 6 + so;
                        sd = f.dst + j * per;
                        mbits = H.bits[g];
                    }
                }
                __syncthreads();   // (the tables' scratch becomes the staging area)
                {
                    const bool sok = z_wave_huf_streams(H, g, mbits, sp, sn, sd, outn, act && ok);
                    ok = ok && sok;
                }
                ITL(3);
                // a frame is done when its four streams decoded; anything else is left to the one-wave decoder (and its errors)
                const uint64_t okm = __ballot(act && ok);
                uint64_t grp_done = 0;
                for (uint32_t gg = 0; gg < ng; gg++)
                    if (((okm >> (4 * gg)) & 15) == 15) grp_done |= 1ull << gg;
                if (in_group && ((grp_done >> my_g) & 1)) done_m |= 1ull << lane;   // (per lane; combined below)
                todo &= ~gm;
                __syncthreads();
            }
            done_m = __ballot(done_m != 0);
            if (threadIdx.x == 0) wk.pre_built = 0;   // (the phase used ZWork's LDS)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_s_waitcnt(0);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            __syncthreads();
        }
        ITL(4);
        // phase 0 (lane per frame): how many sequences, and does the frame qualify
        uint32_t cnt = (arena && zs && !((done_m >> lane) & 1)) ? z_lane_frame(mine.src, mine.csize, mine.out_len, zt, nullptr, 0) : ZPRE_NONE;
        uint32_t start = 0;
        while (start < nb) {
            const uint32_t need = (lane >= start && cnt != ZPRE_NONE) ? cnt : 0u;
            const uint32_t incl = wave_scan_dpp(need);
            uint64_t over = __ballot(lane >= start && lane < nb && incl > ZREC_PER_WAVE);
            uint32_t end = over ? (uint32_t)__builtin_ctzll(over) : nb;
            if (end == start) {            // one frame larger than the arena: the one-wave path
                if (lane == start) cnt = ZPRE_NONE;
                end = start + 1;
            }
            // phase 1 (lane per frame): the records of frames [start, end)
            const uint32_t my_off = incl - need;
            bool ok = false;
            if (lane >= start && lane < end && need)
                ok = z_lane_frame(mine.src, mine.csize, mine.out_len, zt, arena + my_off, need) == need;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_s_waitcnt(0);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            ITL(5);
            // phase 2 (the wave per job, in queue order)
            for (uint32_t k = start; k < end; k++) {
                if ((done_m >> k) & 1) continue;   // decoded in phase H
                const uint32_t k_ok = (uint32_t)__builtin_amdgcn_readlane((int)(ok ? 1u : 0u), (int)k);
                const uint32_t k_off = (uint32_t)__builtin_amdgcn_readlane((int)my_off, (int)k);
                inflate_one(jobs[base + k], st, wk, my_lit, s_win, s_pos, k_ok ? arena + k_off : nullptr, rc);
            }
            ITL(6);
            start = end;
        }
    }
}

// LZ4 blocks: one wave per block from a pool of waves that loops over the job queue (sb_lz4.h: compressed
// bytes and an 8 KiB output window in LDS, speculative 64-position token parse, matches batched)
constexpr uint32_t LZ4_POOL = 2816;   // = the waves that are resident (13.4 KB of LDS each: 11 per CU); 4096 left a thin second round (continuity i64 read 1840 -> 1900 GB/s, utf8 695 -> 730)
__global__ void __launch_bounds__(64) k_inflate_lz4(const InflateJob* jobs, const uint32_t* count, Status* st, uint32_t big_min, uint32_t cap) {
    __shared__ Lz4DecLds lds;
    const uint32_t njobs = min(*count, cap);
    for (uint32_t job = blockIdx.x; job < njobs; job += gridDim.x) {
        const InflateJob j = jobs[job];
        if (j.codec != SB_CODEC_LZ4 || j.csize >= big_min) continue;
        const uint32_t e = lz4_inflate_block(j.src, j.csize, j.dst, j.out_len, lds);
        if (e && threadIdx.x == 0) raise(st, SB_ERR_EXTERNAL, j.page, e);
    }
}
// LZ4 blocks of LZ4_BIG_MIN compressed bytes and more: one workgroup per block (sb_lz4_big.h: sequence starts and match
// chains by pointer doubling).  Launched only when a page of the call is that long (DecodeArgs.lz4_big_min).
constexpr uint32_t LZ4_BIG_POOL = 4096;   // (1024 are resident; blocks differ in size by 20 x, so a workgroup per block — handed out by the hardware as slots free — beats a strided pool)
__global__ void __launch_bounds__(LB_T, 4) k_inflate_lz4_big(const InflateJob* jobs, const uint32_t* count, Status* st, uint32_t big_min, uint32_t cap,
                                                             uint32_t lzg_skipped) {
    __shared__ Lz4BigLds lds;
    const uint32_t njobs = min(*count, cap);
    for (uint32_t job = blockIdx.x; job < njobs; job += gridDim.x) {
        const InflateJob j = jobs[job];
        if (j.codec != SB_CODEC_LZ4 || j.csize < big_min) continue;
        if (lzg_skipped && j.csize >= LZG_MIN) {   // a block for the block-parallel chain, which this call did not launch: replay
            if (threadIdx.x == 0) atomicOr(&st->kinds, KIND_REPLAY | KIND_REPLAY_LZG);
            continue;
        }
        const uint32_t e = lz4_inflate_block_wg(j.src, j.csize, j.dst, j.out_len, lds);
        if (e && threadIdx.x == 0) raise(st, SB_ERR_EXTERNAL, j.page, e);
        __syncthreads();
    }
}

// -------------------------------------------------------------------------------- plan
// aux layouts (u32 words at scratch + PageTask.aux_off):
//   RLE (page or nested):   [0..R]    run_start (exclusive prefix of counts, clamped to N), R+1 words
//                           [R+1.. ]  tile_k0[ntiles]: run that contains row tile*TILE_ROWS
//   bit-packing:            [0..nblk] byte offset of each 128-block header inside the body
//                           [nblk+1..] tile_base[ntiles]: delta prefix at the start of each tile
//   binary Dict:            after the index aux (if any): ent_off[dict_n+1] (u32 byte offsets
//                           of the entries' bytes relative to PageDesc.dict), tile_bytes[ntiles+1]
struct AuxIdx {          // where things live for the *index producing* codec of a page
    uint32_t* base;      // start of its aux words
};

// number of aux words used by the index codec of a page
__device__ __forceinline__ uint32_t idx_aux_words(uint32_t codec, uint32_t n_runs, uint64_t N) {
    const uint32_t ntiles = (uint32_t)((N + TILE_ROWS - 1) / TILE_ROWS);
    if (codec == SB_CODEC_RLE) return n_runs + 1 + ntiles;
    if (codec == SB_CODEC_BITPACKING || codec == SB_CODEC_DELTA_BITPACKING) return (uint32_t)(N / 128) + 1 + ntiles;
    return 0;
}

// binary Dict pages of at least this many tiles leave their per-tile byte totals to k_bin_tile_sums / k_bin_tile_scan
// (every tile by a workgroup of its own) instead of walking the tiles one after the other inside the page's k_plan
constexpr uint32_t BIN_DEFER_TILES = 4;
constexpr uint64_t VAL_BYTES_DEFERRED = ~0ull;   // PageDesc.val_bytes of such a page between k_plan and k_bin_tile_scan
// their exact 64-bit totals per tile, behind tile_bytes[ntiles + 1]
__device__ __forceinline__ uint64_t* bin_tile_sums(const uint32_t* tile_bytes, uint32_t ntiles) {
    return (uint64_t*)(((uintptr_t)(tile_bytes + ntiles + 1) + 7) & ~(uintptr_t)7);
}

// RLE plan: scan the run counts of `body` (records of 4+W bytes) until they cover N rows.
// Returns the number of runs (uniform).  LDS: s_a (SIDX_WORDS u32).
__device__ uint32_t plan_rle(const uint8_t* body, uint32_t csize, uint32_t rec, uint64_t N, uint32_t* aux,
                             uint32_t aux_cap_words, uint32_t* s_a, uint64_t* s_w64, Status* st, uint32_t page,
                             TileTask* page_tiles = nullptr) {
    const int t = threadIdx.x;
    const uint32_t max_runs = csize / rec;
    const uint32_t ntiles = (uint32_t)((N + TILE_ROWS - 1) / TILE_ROWS);
    __shared__ uint32_t s_nruns;
    __shared__ uint64_t s_carry;
    if (t == 0) {
        s_nruns = 0xFFFFFFFFu;
        s_carry = 0;
    }
    __syncthreads();
    // pass 1: run starts
    for (uint32_t base = 0; base < max_runs; base += TILE_ROWS) {
        uint64_t loc[ROWS_PER_THREAD];
        uint64_t run = 0;
#pragma unroll
        for (int jx = 0; jx < ROWS_PER_THREAD; jx++) {
            uint32_t k = base + t * ROWS_PER_THREAD + jx;
            uint64_t cnt = k < max_runs ? (uint64_t)ldu32(body + (uint64_t)k * rec) : 0;
            run += cnt;
            loc[jx] = run;
        }
        uint64_t incl = wave_incl_scan64(run);
        __syncthreads();
        if ((t & 63) == 63) s_w64[t >> 6] = incl;
        __syncthreads();
        uint64_t pre = s_carry + incl - run;
        const int w = t >> 6;
        if (w > 0) pre += s_w64[0];
        if (w > 1) pre += s_w64[1];
        if (w > 2) pre += s_w64[2];
        const uint64_t chunk_total = s_w64[0] + s_w64[1] + s_w64[2] + s_w64[3];
#pragma unroll
        for (int jx = 0; jx < ROWS_PER_THREAD; jx++) {
            uint32_t k = base + t * ROWS_PER_THREAD + jx;
            uint64_t start = pre + (jx ? loc[jx - 1] : 0);  // exclusive
            uint64_t endr = pre + loc[jx];
            if (k < max_runs && start < N) {
                if (k + 1 < aux_cap_words) aux[k] = (uint32_t)start;
                if (endr >= N) s_nruns = k + 1;  // exactly one run crosses N
            }
        }
        __syncthreads();
        if (t == 0) s_carry += chunk_total;
        __syncthreads();
        if (s_nruns != 0xFFFFFFFFu) break;
    }
    uint32_t R = s_nruns;
    if (R == 0xFFFFFFFFu) {
        if (N == 0) {
            R = 0;
        } else {
            if (t == 0) raise(st, SB_ERR_IO, page, 200);  // runs end before N rows (read_u32 EOF upstream)
            return 0xFFFFFFFFu;
        }
    }
    if (R + 1 + ntiles > aux_cap_words) {
        if (t == 0) raise(st, SB_ERR_INVALID, page, 201);
        return 0xFFFFFFFFu;
    }
    if (t == 0) aux[R] = (uint32_t)N;
    __syncthreads();
    // pass 2: tile_k0
    uint32_t* tile_k0 = aux + R + 1;
    for (uint32_t k = t; k < R; k += WG) {
        uint64_t s = aux[k], e = aux[k + 1];
        if (e > s) {
            uint64_t tl = (s + TILE_ROWS - 1) / TILE_ROWS;
            for (; tl * TILE_ROWS < e && tl < ntiles; tl++) tile_k0[tl] = k;
        }
    }
    (void)s_a;
    __syncthreads();
    if (page_tiles)  // the tile's run range travels with its task: no dependent aux loads in k_expand
        for (uint32_t tl = t; tl < ntiles; tl += WG) {
            page_tiles[tl].k0 = tile_k0[tl];
            page_tiles[tl].kend = tl + 1 < ntiles ? tile_k0[tl + 1] + 1 : R;
        }
    return R;
}

// bit-packing plan: walk the block headers (u8 num_bits | 16*num_bits bytes) of `body`, record
// the header offsets and, for delta pages, the running sum at each tile start.
// The walk is a serial pointer chase (block k+1's position depends on block k's header); the
// body is staged through LDS in windows so each step costs an LDS read, not an HBM miss.
constexpr int BP_WINDOW = 16 * 1024;   // (k_plan: 17 KB of tile words + this window + <= 128 VGPRs = 4 workgroups per CU)
// The head of a long bit-packed body: its first BPG_HEAD bytes walked block by block by one lane (the ids of a Dict page
// grow with the rows: 7, 8, 9 ... bits in the first blocks, one width from there on).  Every thread calls; returns the
// first block not walked and its position (s_hw: 2 words of LDS); aux (optional) receives the walked blocks' positions.
constexpr uint32_t BPG_HEAD = 4096;
__device__ __forceinline__ void bp_head_walk(const uint8_t* body, uint32_t csize, uint32_t nblk, uint8_t* s_win, uint32_t* s_hw, uint32_t* aux,
                                             uint32_t* blk_s, uint32_t* pos_s) {
    const uint32_t wlen = min(csize, BPG_HEAD);
    __syncthreads();
    for (uint32_t i = threadIdx.x * 16; i < wlen; i += WG * 16) {
        if (i + 16 <= wlen) {
            *(u32x4*)(s_win + i) = ldu128(body + i);
        } else {
            for (uint32_t b = i; b < wlen; b++) s_win[b] = body[b];
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t pos = 0, blk = 0;
        while (blk < nblk && pos < wlen) {
            const uint32_t nb = s_win[pos];
            if (nb > 32 || pos + 1 + 16 * nb > csize) break;   // (left to the caller's walk, which reports it)
            if (aux) aux[blk] = pos;
            pos += 1 + 16 * nb;
            blk++;
        }
        s_hw[0] = blk;
        s_hw[1] = pos;
    }
    __syncthreads();
    *blk_s = s_hw[0];
    *pos_s = s_hw[1];
}

// guess: k_bp_guess's answer for this body (the blocks behind its head that share the first one's width), or null
__device__ bool plan_bp(const uint8_t* body, uint32_t csize, uint64_t N, bool delta, uint32_t* aux,
                        uint8_t* s_win, uint32_t* s_a, uint32_t* s_w, Status* st, uint32_t page, const uint32_t* guess = nullptr) {
    const int t = threadIdx.x;
    const uint32_t nblk = (uint32_t)(N / 128);
    const uint32_t ntiles = (uint32_t)((N + TILE_ROWS - 1) / TILE_ROWS);
#ifdef SB_PLAN_PRINTF
    const unsigned long long pp0 = __builtin_readcyclecounter();
#endif
    __shared__ uint32_t s_pos, s_blk, s_err;
    if (t == 0) {
        s_pos = 0;
        s_blk = 0;
        s_err = 0;
    }
    __syncthreads();
    // Long pages (a one-page column: 93 750 blocks for 12 M rows, 16 ms of one-lane walking): the blocks of a stretch
    // usually share one width, so GUESS that the blocks from s_blk on all have the width of the first of them, let every
    // thread check its share of the predicted header positions, accept the stretch up to the first header that disagrees
    // and guess again from there; after a few stretches (widths that keep changing) the one-lane walk takes the rest.
    if (nblk >= 4096) {
        __shared__ uint32_t s_bad;
        if (guess) {   // the head block by block, then the stretch k_bp_guess checked (94 000 header bytes 145 bytes apart touch
                       // every line of the body: 13.6 MB through ONE CU were 0.13 ms of this page's plan)
            uint32_t blk0, pos0;
            bp_head_walk(body, csize, nblk, s_win, s_w, aux, &blk0, &pos0);
            if (blk0 < nblk && pos0 < csize && body[pos0] <= 32) {
                const uint32_t stride = 1 + 16 * (uint32_t)body[pos0];
                const uint32_t fit = (uint32_t)min((uint64_t)(nblk - blk0), ((uint64_t)csize - pos0) / stride);
                const uint32_t good = min(fit, *guess);
                for (uint32_t k = t; k < good; k += WG) aux[blk0 + k] = pos0 + k * stride;
                blk0 += good;
                pos0 += good * stride;
            }
            __syncthreads();
            if (t == 0) {
                s_blk = blk0;
                s_pos = pos0;
            }
            __syncthreads();
        }
        for (int round = 0; round < 16 && s_blk < nblk && !s_err; round++) {
            const uint32_t blk0 = s_blk, pos0 = s_pos;
            if (pos0 >= csize) break;   // (the walk below reports it)
            const uint32_t nb = body[pos0];
            if (nb > 32) break;
            const uint32_t stride = 1 + 16 * nb;
            const uint32_t fit = (uint32_t)min((uint64_t)(nblk - blk0), ((uint64_t)csize - pos0) / stride);   // blocks that lie inside the body
            __syncthreads();
            if (t == 0) s_bad = fit;
            __syncthreads();
            uint32_t bad = fit;
            constexpr uint32_t HF = 8;   // header bytes of a thread in flight
            for (uint32_t k0 = t; k0 < fit && bad == fit; k0 += HF * WG) {
                uint32_t hb[HF];
#pragma unroll
                for (uint32_t u = 0; u < HF; u++) {
                    const uint32_t k = k0 + u * WG;
                    hb[u] = k < fit ? (uint32_t)body[pos0 + (uint64_t)k * stride] : nb;
                }
#pragma unroll
                for (uint32_t u = 0; u < HF; u++)
                    if (hb[u] != nb) bad = min(bad, k0 + u * WG);
            }
            if (bad < fit) atomicMin(&s_bad, bad);
            __syncthreads();
            const uint32_t good = s_bad;   // blocks blk0 .. blk0 + good - 1 have this width
            for (uint32_t k = t; k < good; k += WG) aux[blk0 + k] = pos0 + k * stride;
            __syncthreads();
            if (t == 0) {
                s_blk = blk0 + good;
                s_pos = pos0 + good * stride;
            }
            __syncthreads();
            if (good == 0) break;
        }
    }
    while (s_blk < nblk && !s_err) {
        const uint32_t win0 = s_pos;
        const uint32_t wlen = min((uint32_t)BP_WINDOW, csize - min(csize, win0));
        __syncthreads();
        for (uint32_t i = t * 16; i < wlen; i += WG * 16) {
            if (i + 16 <= wlen) {
                *(u32x4*)(s_win + i) = ldu128(body + win0 + i);
            } else {
                for (uint32_t b = i; b < wlen; b++) s_win[b] = body[win0 + b];
            }
        }
        __syncthreads();
        if (t == 0) {
            uint32_t pos = win0, blk = s_blk;
            while (blk < nblk && pos - win0 < wlen) {
                uint32_t nb = s_win[pos - win0];
                if (nb > 32 || pos + 1 + 16 * nb > csize) {
                    s_err = 1;
                    break;
                }
                aux[blk] = pos;
                pos += 1 + 16 * nb;
                blk++;
            }
            if (blk < nblk && pos >= csize) s_err = 1;
            s_pos = pos;
            s_blk = blk;
        }
        __syncthreads();
    }
    if (s_err) {
        if (t == 0) raise(st, SB_ERR_IO, page, 210);
        return false;
    }
    if (t == 0) aux[nblk] = s_pos;
    __syncthreads();
#ifdef SB_PLAN_PRINTF
    if (t == 0 && nblk >= 4096) printf("plan_bp page %u: hdr0 %u hdr@145 %u hdr@aux1 %u aux1 %u aux2 %u nblk %u csize %u guess %d/%u cycles %llu delta %d\n", page, body[0], body[145], body[aux[1]], aux[1], aux[2], nblk, csize, guess ? 1 : 0, guess ? *guess : 0u, __builtin_readcyclecounter() - pp0, (int)delta);
#endif
    uint32_t* tile_base = aux + nblk + 1;
    if (!delta) return true;
    // delta pages: value[j] = sum of all deltas up to j (initial 0, delta_bp.rs:73,88); the
    // expand step needs the running sum at each tile start
    uint32_t carry = 0;
    for (uint32_t tl = 0; tl < ntiles; tl++) {
        if (t == 0) tile_base[tl] = carry;
        const uint32_t rows = (uint32_t)min((uint64_t)TILE_ROWS, N - (uint64_t)tl * TILE_ROWS);
        uint32_t acc = 0;
        for (uint32_t j = t; j < rows; j += WG) {
            uint32_t blk = tl * (TILE_ROWS / 128) + (j >> 7);
            const uint8_t* hp = body + aux[blk];
            acc += bp4x_extract(hp + 1, hp[0], j & 127);
        }
#pragma unroll
        for (int dd = 32; dd > 0; dd >>= 1) acc += __shfl_down(acc, dd, 64);
        __syncthreads();
        if ((t & 63) == 0) s_w[t >> 6] = acc;
        __syncthreads();
        carry += s_w[0] + s_w[1] + s_w[2] + s_w[3];
    }
    (void)s_a;
    return true;
}

// ---- index tiles -----------------------------------------------------------------------
// Produce the u32 values of rows [r0, r0+rows) of a u32 stream coded with `codec` into the LDS
// array s_a (sidx layout).  Used for Dict indices and for top-level 4-byte integer pages.
struct U32Stream {
    const uint8_t* src;   // body (or inflated copy)
    const uint32_t* aux;  // plan output
    uint32_t codec;
    uint32_t n_runs;
    uint64_t N;
};

// The first 4*WG run starts of a tile, fetched ahead of time (RLE pages of ~32-row runs have ~130)
struct RleStarts {
    uint32_t st[4];
};
__device__ __forceinline__ RleStarts rle_fetch_starts(const uint32_t* aux, uint32_t k0, uint32_t kend) {
    RleStarts r;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t k = k0 + 1 + threadIdx.x + j * WG;
        r.st[j] = k < kend ? aux[k] : 0xFFFFFFFFu;
    }
    return r;
}
// s_a[row] <- index of the run covering row (relative to k0), via scatter of run starts + scan
__device__ void rle_tile_runidx_k(const uint32_t* aux, uint32_t k0, uint32_t kend, uint32_t tile, uint32_t rows,
                                  uint32_t* s_a, uint32_t* s_w, const RleStarts& pre) {
    const int t = threadIdx.x;
    const uint32_t r0 = tile * TILE_ROWS;
    for (int i = t; i < SIDX_WORDS; i += WG) s_a[i] = 0;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; j++)
        if (pre.st[j] >= r0 && pre.st[j] < r0 + rows) atomicAdd(&s_a[sidx((int)(pre.st[j] - r0))], 1u);
    for (uint32_t k = k0 + 1 + t + 4 * WG; k < kend; k += WG) {
        uint32_t s = aux[k];
        if (s >= r0 && s < r0 + rows) atomicAdd(&s_a[sidx((int)(s - r0))], 1u);
    }
    __syncthreads();
    tile_incl_scan(s_a, s_w);
}
__device__ void rle_tile_runidx(const uint32_t* aux, uint32_t R, uint64_t N, uint32_t tile, uint32_t rows,
                                uint32_t* s_a, uint32_t* s_w) {
    const int t = threadIdx.x;
    const uint32_t ntiles = (uint32_t)((N + TILE_ROWS - 1) / TILE_ROWS);
    const uint32_t* tile_k0 = aux + R + 1;
    const uint32_t k0 = tile_k0[tile];
    const uint32_t kend = tile + 1 < ntiles ? tile_k0[tile + 1] + 1 : R;  // runs starting before the tile end
    const uint32_t r0 = tile * TILE_ROWS;
    for (int i = t; i < SIDX_WORDS; i += WG) s_a[i] = 0;
    __syncthreads();
    for (uint32_t k = k0 + 1 + t; k < kend; k += WG) {
        uint32_t s = aux[k];
        if (s >= r0 && s < r0 + rows) atomicAdd(&s_a[sidx((int)(s - r0))], 1u);
    }
    __syncthreads();
    tile_incl_scan(s_a, s_w);
}

__device__ void u32_tile_to_lds(const U32Stream& s, uint32_t tile, uint32_t rows, uint32_t* s_a, uint32_t* s_w) {
    const int t = threadIdx.x;
    const uint64_t r0 = (uint64_t)tile * TILE_ROWS;
    switch (s.codec) {
        case SB_CODEC_NONE:
        case SB_CODEC_LZ4:
        case SB_CODEC_ZSTD:
        case SB_CODEC_SNAPPY: {  // plain u32 (inflated beforehand if compressed)
            const uint8_t* p = s.src + r0 * 4;
            for (uint32_t i = t; i < rows; i += WG) s_a[sidx((int)i)] = ldu32(p + (uint64_t)i * 4);
            __syncthreads();
            break;
        }
        case SB_CODEC_ONEVALUE: {
            const uint32_t v = ldu32(s.src);
            for (uint32_t i = t; i < rows; i += WG) s_a[sidx((int)i)] = v;
            __syncthreads();
            break;
        }
        case SB_CODEC_RLE: {
            rle_tile_runidx(s.aux, s.n_runs, s.N, tile, rows, s_a, s_w);
            const uint32_t k0 = (s.aux + s.n_runs + 1)[tile];
            for (uint32_t i = t; i < rows; i += WG) {
                uint32_t k = k0 + s_a[sidx((int)i)];
                s_a[sidx((int)i)] = ldu32(s.src + (uint64_t)k * 8 + 4);
            }
            __syncthreads();
            break;
        }
        case SB_CODEC_BITPACKING:
        case SB_CODEC_DELTA_BITPACKING: {
            const uint32_t nblk = (uint32_t)(s.N / 128);
            for (uint32_t j = t; j < rows; j += WG) {
                uint32_t blk = tile * (TILE_ROWS / 128) + (j >> 7);
                const uint8_t* hp = s.src + s.aux[blk];
                s_a[sidx((int)j)] = bp4x_extract(hp + 1, hp[0], j & 127);
            }
            for (uint32_t j = rows + t; j < TILE_ROWS; j += WG) s_a[sidx((int)j)] = 0;
            __syncthreads();
            if (s.codec == SB_CODEC_DELTA_BITPACKING) {
                tile_incl_scan(s_a, s_w);
                const uint32_t base = (s.aux + nblk + 1)[tile];
                for (uint32_t j = t; j < rows; j += WG) s_a[sidx((int)j)] += base;
                __syncthreads();
            }
            break;
        }
        default:
            break;
    }
}

// Walks the containers of a serialized RoaringBitmap (portable format: array / bitmap / run) with one
// workgroup: put(value, k) for the k-th set bit.  Returns the number of bits visited (~0 on a truncated
// or malformed stream).
template <class Put>
__device__ uint64_t roaring_walk(const uint8_t* rb, uint32_t rb_len, uint32_t* s_a, uint32_t* s_w, Put put) {
    const int t = threadIdx.x;
    if (rb_len < 8) return ~0ull;
    const uint32_t cookie = ldu32(rb);
    const bool has_run = (cookie & 0xFFFF) == 12347;
    if (!has_run && cookie != 12346) return ~0ull;
    const uint32_t nc = has_run ? (cookie >> 16) + 1 : ldu32(rb + 4);
    const uint8_t* run_bits = rb + 4;
    const uint32_t hp = has_run ? 4 + (nc + 7) / 8 : 8;
    if (nc > 65536 || (uint64_t)hp + 4ull * nc > rb_len) return ~0ull;
    uint32_t pos = hp + 4 * nc;
    if (!has_run || nc >= 4) pos += 4 * nc;  // offset header
    uint64_t cum = 0;
    for (uint32_t ci = 0; ci < nc; ci++) {
        const uint32_t hi = (uint32_t)ldu16(rb + hp + 4 * ci) << 16;
        const uint32_t card = (uint32_t)ldu16(rb + hp + 4 * ci + 2) + 1;
        const bool run = has_run && ((run_bits[ci >> 3] >> (ci & 7)) & 1);
        if (run) {
            if (pos + 2 > rb_len) return ~0ull;
            const uint32_t nr = ldu16(rb + pos);
            pos += 2;
            if (pos + 4ull * nr > rb_len) return ~0ull;
            if (t == 0) {  // runs are rare in what roaring's serialize_into writes (never without run_optimize)
                uint64_t k = cum;
                for (uint32_t r = 0; r < nr; r++) {
                    const uint32_t s0 = ldu16(rb + pos + 4 * r), len = ldu16(rb + pos + 4 * r + 2);
                    for (uint32_t v = s0; v <= s0 + len; v++) put(hi | v, k++);
                }
            }
            pos += 4 * nr;
        } else if (card > 4096) {  // bitmap container: 1024 u64 words
            if (pos + 8192ull > rb_len) return ~0ull;
            const uint8_t* bm = rb + pos;
            __syncthreads();
            // thread t owns 64-bit words [4t, 4t + 4): popcounts -> exclusive prefix over the workgroup
            uint64_t wd[4];
            uint32_t mine = 0;
            for (int q = 0; q < 4; q++) {
                wd[q] = ldu64(bm + (uint64_t)(4 * t + q) * 8);
                mine += (uint32_t)__popcll(wd[q]);
            }
            const uint32_t incl = wave_incl_scan(mine);
            if ((t & 63) == 63) s_w[t >> 6] = incl;
            __syncthreads();
            uint32_t k = incl - mine;
            for (int pw = 0; pw < 3; pw++)
                if (pw < (t >> 6)) k += s_w[pw];
            for (int q = 0; q < 4; q++) {
                uint64_t m = wd[q];
                while (m) {
                    const int b = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    put(hi | (uint32_t)((4 * t + q) * 64 + b), cum + k++);
                }
            }
            __syncthreads();
            pos += 8192;
        } else {  // array container: sorted u16 values
            if (pos + 2ull * card > rb_len) return ~0ull;
            for (uint32_t k = t; k < card; k += WG) put(hi | ldu16(rb + pos + 2 * k), cum + k);
            pos += 2 * card;
        }
        cum += card;
    }
    (void)s_a;
    return cum;
}

// binary Dict plan: entry offsets (serial walk over `u64 len | bytes` records, staged through
// LDS) and per-tile byte totals of the page.
__device__ bool plan_bin_dict(PageDesc& d, const U32Stream& is, uint64_t N, const uint8_t* page_end, uint32_t* aux,
                              uint32_t aux_cap_words, uint8_t* s_win, uint32_t* s_a, uint32_t* s_w, uint64_t* s_w64,
                              Status* st, uint32_t page, uint32_t* defer_count /* NULL: tile totals here */,
                              uint32_t gap = 0 /* bytes between entry 0 and entry 1 (Freq) */) {
    const int t = threadIdx.x;
    const uint32_t ntiles = (uint32_t)((N + TILE_ROWS - 1) / TILE_ROWS);
    const uint32_t D = d.dict_n;
    const bool defer = defer_count && ntiles >= BIN_DEFER_TILES;
    if ((uint64_t)D + 1 + ntiles + 1 + (defer ? 2ull * ntiles + 2 : 0) > aux_cap_words) {
        if (t == 0) raise(st, SB_ERR_INVALID, page, 220);
        return false;
    }
    uint32_t* ent_off = aux;              // D+1 entries: offset of entry k's record (its u64 len) from d.dict
    uint32_t* tile_bytes = aux + D + 1;   // ntiles+1: exclusive prefix of value bytes per tile
    const uint64_t avail = (uint64_t)(page_end - d.dict);
    __shared__ uint32_t s_pos, s_ent, s_err;
    if (t == 0) {
        s_pos = 0;
        s_ent = 0;
        s_err = 0;
    }
    __syncthreads();
    while (s_ent < D && !s_err) {
        const uint32_t win0 = s_pos;
        const uint32_t wlen = (uint32_t)min((uint64_t)BP_WINDOW, avail - min(avail, (uint64_t)win0));
        __syncthreads();
        for (uint32_t i = t * 16; i < wlen; i += WG * 16) {
            if (i + 16 <= wlen) {
                *(u32x4*)(s_win + i) = ldu128(d.dict + win0 + i);
            } else {
                for (uint32_t b = i; b < wlen; b++) s_win[b] = d.dict[win0 + b];
            }
        }
        __syncthreads();
        // ---- the entries of the window, found in PARALLEL.  An entry starts with a u64 length whose upper half is zero, and
        // in text no other four bytes are: byte position p is a candidate when bytes p+4 .. p+7 are zero, candidates come in
        // runs of up to four (p = c-3 .. c in front of a length field at c), and the LAST of a run is the entry.  The picks
        // are compacted in order and VERIFIED — the first is the window's start, each one's successor sits 8 + length behind
        // it — so the result is exact: anything else (binary data with zero bytes, empty strings) takes the serial walk
        // below, which costs a lone lane ~100 ns per entry (0.6 ms for the 6 000 entries of a C3 page).
        bool par_done = false;
        if (!(gap && s_ent == 0) && wlen >= 64) {
            uint16_t* list = (uint16_t*)s_a;                   // <= wlen / 8 picks (s_a: SIDX_WORDS words)
            const uint32_t lane = t & 63, wv = t >> 6;
            const uint32_t p0 = (uint32_t)t * 64;              // my 64 byte positions
            uint64_t zlo = 0;                                   // bit i: byte p0 + 4 + i is zero (i < 64)
            uint32_t zhi = 0;                                   // bits for bytes p0 + 68 .. p0 + 71
            if (p0 < wlen) {
                const uint32_t* w32 = (const uint32_t*)(s_win + p0 + 4);
#pragma unroll
                for (int k = 0; k < 17; k++) {
                    const uint32_t b = p0 + 4 + 4 * (uint32_t)k;
                    uint32_t w = 0xFFFFFFFFu;
                    if (b + 4 <= wlen) w = w32[k];
                    const uint32_t zm = ((w & 0xFFu) == 0 ? 1u : 0u) | ((w & 0xFF00u) == 0 ? 2u : 0u) | ((w & 0xFF0000u) == 0 ? 4u : 0u) |
                                        ((w & 0xFF000000u) == 0 ? 8u : 0u);
                    if (k < 16) zlo |= (uint64_t)zm << (4 * k);
                    else zhi = zm;
                }
            }
            // cand bit i (i <= 64): positions p0 + i; needs zero bits i .. i+3
            const uint64_t z1 = (zlo >> 1) | ((uint64_t)(zhi & 1) << 63), z2 = (zlo >> 2) | ((uint64_t)(zhi & 3) << 62),
                           z3 = (zlo >> 3) | ((uint64_t)(zhi & 7) << 61);
            uint64_t cand = zlo & z1 & z2 & z3;
            const uint32_t cand64 = (zhi & 15) == 15 ? 1u : 0u;      // position p0 + 64 (the next thread's first)
            // only positions whose 8-byte length field lies inside the window
            if (p0 + 64 + 8 > wlen) {
                const uint32_t okn = wlen >= p0 + 8 ? wlen - 8 - p0 + 1 : 0;   // positions p0 .. p0 + okn - 1
                cand &= okn >= 64 ? ~0ull : ((1ull << okn) - 1);
            }
            const bool c64 = cand64 && p0 + 64 + 8 <= wlen;
            const uint64_t pick = cand & ~((cand >> 1) | ((uint64_t)(c64 ? 1 : 0) << 63));
            const uint32_t cnt = (uint32_t)__popcll(pick);
            const uint32_t incl = wave_incl_scan(cnt);
            if (lane == 63) s_w[wv] = incl;
            __syncthreads();
            uint32_t base_k = incl - cnt;
            for (uint32_t pw = 0; pw < wv; pw++) base_k += s_w[pw];
            const uint32_t m = s_w[0] + s_w[1] + s_w[2] + s_w[3];
            {
                uint64_t pk = pick;
                uint32_t k = base_k;
                while (pk) {
                    const uint32_t i = (uint32_t)__builtin_ctzll(pk);
                    pk &= pk - 1;
                    list[k++] = (uint16_t)(p0 + i);
                }
            }
            __syncthreads();
            // verify the chain: entry j is good when its successor is the next pick; the verified PREFIX is taken (a length
            // field cut by the window's end leaves a false last pick)
            __syncthreads();
            if (t == 0) s_w[0] = (m >= 2 && list[0] == 0) ? m - 1 : 0u;
            __syncthreads();
            if (m >= 2) {
                for (uint32_t j = t; j + 1 < m; j += WG) {
                    const uint32_t pj = list[j], pn = list[j + 1];
                    const uint32_t x = pj & 3;
                    const uint32_t* q = (const uint32_t*)(s_win + (pj & ~3u));
                    const uint32_t len = __builtin_amdgcn_alignbyte(q[1], q[0], x);
                    if (pn != pj + 8 + len) atomicMin(&s_w[0], j);
                }
            }
            __syncthreads();
            const uint32_t e0 = s_ent;
            const uint32_t K = min(s_w[0], D - e0);
            if (K) {
                for (uint32_t j = t; j < K; j += WG) ent_off[e0 + j] = win0 + list[j];
                __syncthreads();
                if (t == 0) {
                    s_pos = win0 + list[K];
                    s_ent = e0 + K;
                    if (e0 + K < D && (uint64_t)s_pos + 8 > avail) s_err = 1;
                }
                par_done = true;
            }
            __syncthreads();
        }
        if (!par_done && t == 0) {
            uint32_t pos = win0, e = s_ent;
            while (e < D && pos - win0 + 8 <= wlen) {
                uint64_t len;
                __builtin_memcpy(&len, s_win + (pos - win0), 8);
                if (len > avail - pos - 8) {
                    s_err = 1;
                    break;
                }
                ent_off[e] = pos;
                pos += 8 + (uint32_t)len;
                e++;
                if (e == 1 && gap) {  // Freq: the bitmap sits between the top value and the exceptions
                    if ((uint64_t)pos + gap > avail) {
                        s_err = 1;
                        break;
                    }
                    pos += gap;
                    break;  // the window moves
                }
            }
            if (e < D && (uint64_t)pos + 8 > avail) s_err = 1;
            s_pos = pos;
            s_ent = e;
        }
        __syncthreads();
    }
    if (s_err) {
        if (t == 0) raise(st, SB_ERR_OUT_OF_SPEC, page, 221);
        return false;
    }
    if (t == 0) ent_off[D] = s_pos;
    __syncthreads();
    PTL(2);
    if (defer) {   // the tiles' totals by (page, tile) workgroups, their prefix by k_bin_tile_scan
        if (t == 0) {
            d.val_bytes = VAL_BYTES_DEFERRED;
            atomicAdd(defer_count, 1u);
        }
        return true;
    }
    uint64_t carry = 0;
    for (uint32_t tl = 0; tl < ntiles; tl++) {
        const uint32_t rows = (uint32_t)min((uint64_t)TILE_ROWS, N - (uint64_t)tl * TILE_ROWS);
        u32_tile_to_lds(is, tl, rows, s_a, s_w);
        uint64_t acc = 0;
        bool bad = false;
        for (uint32_t i = t; i < rows; i += WG) {
            uint32_t k = s_a[sidx((int)i)];
            if (k >= D) {
                bad = true;
                break;
            }
            const uint64_t pr = ldu64((const uint8_t*)(ent_off + k));   // (one 8-byte gather for the offset pair)
            acc += (uint32_t)(pr >> 32) - (uint32_t)pr - 8 - (k == 0 ? gap : 0);
        }
        if (bad) raise(st, SB_ERR_OUT_OF_SPEC, page, 222);
        if (t == 0) tile_bytes[tl] = (uint32_t)carry;
        carry += wg_sum64(acc, s_w64);
        __syncthreads();
    }
    if (t == 0) {
        tile_bytes[ntiles] = (uint32_t)carry;
        d.val_bytes = carry;
    }
    __syncthreads();
    PTL(3);
    return true;
}

// The bit-packed body k_plan will walk for a page of 4096 blocks and more (a top-level page, or the indices of a Dict page),
// or null.  plan_bp guesses that the blocks behind the body's head (bp_head_walk) all have one width; the guess is checked
// here by BPG_PARTS workgroups per page: bp_guess[page] = the first of them that disagrees (~0 from the host's memset: none).
constexpr uint32_t BPG_PARTS = 64;
__device__ __forceinline__ const uint8_t* bp_long_body(const DecodeArgs& a, const PageDesc& d, const PageTask& t, const ColDesc& c, uint32_t* csize) {
    if (!d.ok || c.ptype == SB_TYPE_BOOLEAN || t.num_values / 128 < 4096 || rle_by_page(c, d)) return nullptr;
    if (d.codec == SB_CODEC_BITPACKING || d.codec == SB_CODEC_DELTA_BITPACKING) {
        *csize = (uint32_t)(c.pages + t.in_off + t.length - d.body);
        return d.body;
    }
    if (d.codec == SB_CODEC_DICT && (d.icodec == SB_CODEC_BITPACKING || d.icodec == SB_CODEC_DELTA_BITPACKING)) {
        *csize = d.icsize;
        return d.ibody;
    }
    return nullptr;
}
__global__ void __launch_bounds__(WG) k_bp_guess(DecodeArgs a) {
    if (a.job_counts[3] == 0) return;
    const uint32_t p = blockIdx.y;
    const PageDesc d = a.descs[p];
    const PageTask t = a.tasks[p];
    const ColDesc c = a.cols[t.col];
    uint32_t csize = 0;
    const uint8_t* body = bp_long_body(a, d, t, c, &csize);
    if (!body || csize == 0) return;
    const uint32_t nblk = (uint32_t)(t.num_values / 128);
    __shared__ __attribute__((aligned(16))) uint8_t s_win[BPG_HEAD];
    __shared__ uint32_t s_hw[2];
    uint32_t blk0, pos0;
    bp_head_walk(body, csize, nblk, s_win, s_hw, nullptr, &blk0, &pos0);   // (every workgroup of the page: 4 KB, a few dozen blocks)
    if (blk0 >= nblk || pos0 >= csize) return;
    const uint32_t nb = body[pos0];
    if (nb > 32) return;
    const uint32_t stride = 1 + 16 * nb;
    const uint32_t fit = (uint32_t)min((uint64_t)(nblk - blk0), ((uint64_t)csize - pos0) / stride);
    uint32_t bad = 0xFFFFFFFFu;
    for (uint32_t k0 = blockIdx.x * WG + threadIdx.x; k0 < fit && bad == 0xFFFFFFFFu; k0 += 4 * WG * gridDim.x) {
        uint32_t hb[4];
#pragma unroll
        for (uint32_t u = 0; u < 4; u++) {
            const uint32_t k = k0 + u * WG * gridDim.x;
            hb[u] = k < fit ? (uint32_t)body[pos0 + (uint64_t)k * stride] : nb;
        }
#pragma unroll
        for (uint32_t u = 0; u < 4; u++)
            if (hb[u] != nb) bad = min(bad, k0 + u * WG * gridDim.x);
    }
    if (bad != 0xFFFFFFFFu) atomicMin(a.bp_guess + p, bad);
}

// 4 workgroups per CU (LDS and registers): a 64-column x 16-page batch is 1024 pages = ONE round of the chip; at 3 per CU
// it took two, and a page's plan is a latency chain of ~1 ms whatever else runs
__global__ void __launch_bounds__(WG, 4) k_plan(DecodeArgs a) {
    // what this call queued, for the launches of the next interval's calls (sb_read_columns leaves out the inflate kernels of
    // queue A / the tile kernel of primitives when the last read interval had nothing for them); a call that has work for a
    // kernel it left out says so: the interval is issued again with everything (sb_ctx_synchronize)
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const uint32_t na = a.job_counts[0], nt = a.job_counts[2];
        uint32_t k = (na ? KIND_QUEUE_A : 0u) | (nt ? KIND_TILES : 0u);
        if (((a.read_skips & RSKIP_QUEUE_A) && na) || ((a.read_skips & RSKIP_TILES) && nt)) k |= KIND_REPLAY;
        if (k) atomicOr(&a.status->kinds, k);
    }
    if (a.read_skips & RSKIP_QUEUE_A)
        if (a.job_counts[0]) return;   // (nothing was inflated: no plan on what is not there)
    if (a.job_counts[3] == 0) return;  // no page of this call needs a plan (k_parse counts them)
#ifdef SB_PLAN_PRINTF
    const unsigned long long kp0 = __builtin_readcyclecounter();
#endif
    const uint32_t p = blockIdx.x;
    __shared__ uint32_t s_a[SIDX_WORDS];
    __shared__ __attribute__((aligned(16))) uint8_t s_win[BP_WINDOW];
    __shared__ uint32_t s_w[4];
    __shared__ uint64_t s_w64[4];
    PageDesc d = a.descs[p];
    if (!d.ok) return;
    const PageTask t = a.tasks[p];
    const ColDesc c = a.cols[t.col];
    const uint64_t N = t.num_values;
    uint32_t* aux = (uint32_t*)(a.scratch + t.aux_off);
    const uint32_t aux_cap = (uint32_t)((t.infl_off - t.aux_off) / 4);
    const uint8_t* page_end = c.pages + t.in_off + t.length;
    uint32_t* defer_count = a.sizes_only ? nullptr : a.job_counts + 12;   // (sb_read_columns_sizes launches no tile kernels)
    bool changed = false;
    if (c.ptype == SB_TYPE_BOOLEAN) {
        if (d.codec == SB_CODEC_RLE) {  // runs of u32 count | u8 value (boolean/rle.rs:41-55)
            uint32_t R = plan_rle(d.body, (uint32_t)(page_end - d.body), 5, N, aux, aux_cap, s_a, s_w64, a.status, p);
            if (R == 0xFFFFFFFFu) d.ok = 0;
            d.n_runs = R;
            changed = true;
        }
    } else if (rle_by_page(c, d)) {
        return;  // k_expand_rle walks the runs itself
    } else if (d.codec == SB_CODEC_RLE) {  // runs see the rest of the buffer (integer/mod.rs:108-110)
        uint32_t R = plan_rle(d.body, (uint32_t)(page_end - d.body), 4 + c.width, N, aux, aux_cap, s_a, s_w64,
                              a.status, p, a.tiles + d.tile_base);
        if (R == 0xFFFFFFFFu) d.ok = 0;
        d.n_runs = R;
        changed = true;
    } else if (d.codec == SB_CODEC_BITPACKING || d.codec == SB_CODEC_DELTA_BITPACKING) {
        uint32_t gcs = 0;
        const uint32_t* guess = a.bp_guess && bp_long_body(a, d, t, c, &gcs) == d.body ? a.bp_guess + p : nullptr;
        if (!plan_bp(d.body, (uint32_t)(page_end - d.body), N, d.codec == SB_CODEC_DELTA_BITPACKING, aux, s_win, s_a,
                     s_w, a.status, p, guess)) {
            d.ok = 0;
            changed = true;
        }
    } else if (d.codec == SB_CODEC_DICT) {
        uint32_t ic = d.icodec;
        if (ic == SB_CODEC_FREQ) {  // indices = top everywhere, exceptions scattered by the Roaring bitmap (see k_parse)
            uint32_t* idx = (uint32_t*)(a.scratch + t.infl_off);
            const uint32_t top = ldu32(d.ibody);
            const uint32_t E = d.vusize;
            const uint8_t* ex = d.src;
            const bool one = d.pad == SB_CODEC_ONEVALUE;
            const uint32_t ec = d.pad;
            if ((ec == SB_CODEC_RLE || ec == SB_CODEC_BITPACKING || ec == SB_CODEC_DELTA_BITPACKING) && E) {
                // exceptions coded with RLE / bit-packing: expand them to a plain u32 array behind the index array
                const uint8_t* eh = d.vbody + d.vcsize;
                const uint32_t ecsize = ldu32(eh + 1);
                uint32_t* plain = (uint32_t*)(a.scratch + t.infl_off + ((N * 4 + 15) & ~(uint64_t)15));
                uint32_t runs = 0;
                bool okx = true;
                if (ec == SB_CODEC_RLE) {
                    runs = plan_rle(ex, ecsize, 8, E, aux, aux_cap, s_a, s_w64, a.status, p);
                    okx = runs != 0xFFFFFFFFu;
                } else {
                    okx = plan_bp(ex, ecsize, E, ec == SB_CODEC_DELTA_BITPACKING, aux, s_win, s_a, s_w, a.status, p);
                }
                if (okx) {
                    const U32Stream es{ex, aux, ec, runs, E};
                    for (uint32_t tl = 0; tl * TILE_ROWS < E; tl++) {
                        const uint32_t rows = min((uint32_t)TILE_ROWS, E - tl * TILE_ROWS);
                        __syncthreads();
                        u32_tile_to_lds(es, tl, rows, s_a, s_w);
                        for (uint32_t i = threadIdx.x; i < rows; i += WG) plain[tl * TILE_ROWS + i] = s_a[sidx((int)i)];
                    }
                    __syncthreads();
                    ex = (const uint8_t*)plain;
                } else {
                    d.ok = 0;
                }
            }
            for (uint64_t i = threadIdx.x; i < N; i += WG) idx[i] = top;
            __syncthreads();
            const uint64_t cum = roaring_walk(d.vbody, d.vcsize, s_a, s_w, [&](uint64_t row, uint64_t k) {
                if (row >= N || k >= E) {
                    raise(a.status, SB_ERR_OUT_OF_SPEC, p, 226);  // exception index out of bounds
                    return;
                }
                idx[row] = one ? ldu32(ex) : ldu32(ex + 4 * k);
            });
            __syncthreads();
            if (cum != E && d.ok) {
                if (threadIdx.x == 0) raise(a.status, SB_ERR_EXTERNAL, p, 225);  // malformed RoaringBitmap
                d.ok = 0;
            }
            d.icodec = SB_CODEC_NONE;
            d.isrc = (const uint8_t*)idx;
            ic = SB_CODEC_NONE;
            changed = true;
        }
        if (ic == SB_CODEC_RLE) {
            uint32_t R = plan_rle(d.ibody, (uint32_t)(page_end - d.ibody), 8, N, aux, aux_cap, s_a, s_w64, a.status, p);
            if (R == 0xFFFFFFFFu) d.ok = 0;
            d.n_runs = R;
            changed = true;
        } else if (ic == SB_CODEC_BITPACKING || ic == SB_CODEC_DELTA_BITPACKING) {
            PTL(0);
#ifdef SB_PLAN_PRINTF
            if (threadIdx.x == 0) printf("  before plan_bp: %llu  bp_guess %p ic %u body==%d\n", __builtin_readcyclecounter() - kp0, (void*)a.bp_guess, ic, (int)(d.ibody != nullptr));
#endif
            uint32_t gcs = 0;
            const uint32_t* guess = a.bp_guess && ic == d.icodec && bp_long_body(a, d, t, c, &gcs) == d.ibody ? a.bp_guess + p : nullptr;
            if (!plan_bp(d.ibody, d.icsize, N, ic == SB_CODEC_DELTA_BITPACKING, aux, s_win, s_a, s_w, a.status, p, guess)) {
                d.ok = 0;
                changed = true;
            }
            PTL(1);
#ifdef SB_PLAN_PRINTF
            if (threadIdx.x == 0) printf("  after plan_bp: %llu\n", __builtin_readcyclecounter() - kp0);
#endif
        }
        if (d.ok && is_binary(c.ptype)) {
            const uint32_t used = idx_aux_words(ic, d.n_runs, N);
            U32Stream is{d.isrc, aux, ic, d.n_runs, N};
            __syncthreads();
            if (!plan_bin_dict(d, is, N, page_end, aux + used, aux_cap - used, s_win, s_a, s_w, s_w64, a.status, p, defer_count))
                d.ok = 0;
            changed = true;
        }
    }
    else if (d.codec == SB_CODEC_FREQ && is_binary(c.ptype)) {  // virtual Dict page (see k_parse)
        uint32_t* idx = (uint32_t*)(a.scratch + t.infl_off);
        __shared__ uint32_t s_inrange;
        if (threadIdx.x == 0) s_inrange = 0;
        for (uint64_t i = threadIdx.x; i < N; i += WG) idx[i] = 0;
        __syncthreads();
        const uint64_t cum = roaring_walk(d.vbody, d.vcsize, s_a, s_w, [&](uint64_t row, uint64_t k) {
            if (row < N) {  // bits past the page are never consulted (`contains(i)` for i < length)
                idx[row] = (uint32_t)k + 1;
                atomicAdd(&s_inrange, 1u);
            }
        });
        __syncthreads();
        if (cum == ~0ull) {
            if (threadIdx.x == 0) raise(a.status, SB_ERR_EXTERNAL, p, 224);  // RoaringBitmap::deserialize_from failed
            d.ok = 0;
        } else {
            d.dict_n = s_inrange + 1;
            U32Stream is{(const uint8_t*)idx, aux, SB_CODEC_NONE, 0, N};
            if (!plan_bin_dict(d, is, N, page_end, aux, aux_cap, s_win, s_a, s_w, s_w64, a.status, p, defer_count, 4 + d.vcsize)) d.ok = 0;
        }
        changed = true;
    }
    // binary Basic: last decoded offset of the page (needed for the cross-page offset base)
    if (d.ok && is_binary(c.ptype)) {
        if (is_basic(d.codec)) {
            d.off_last = c.width == 4 ? (uint64_t)ldu32(d.src + N * 4) : ldu64(d.src + N * 8);
        } else {
            d.off_last = d.val_bytes;
        }
        changed = true;
    }
    if (changed && threadIdx.x == 0) a.descs[p] = d;
#ifdef SB_PLAN_PRINTF
    if (threadIdx.x == 0 && N >= (1u << 19)) printf("k_plan page %u: codec %u icodec %u N %llu cycles %llu\n", p, d.codec, d.icodec, (unsigned long long)N, __builtin_readcyclecounter() - kp0);
#endif
}

// -------------------------------------------------------------------------------- binary Dict: tile totals of long pages
// The value bytes a tile of a binary Dict page produces (what plan_bin_dict's second half adds up tile after tile): one
// workgroup per entry of the compact tile list, for the pages k_plan left to it (val_bytes == VAL_BYTES_DEFERRED).  A
// 3 M-row page is 732 tiles: 3.3 ms inside its k_plan workgroup, one round of the chip here.
__global__ void __launch_bounds__(WG) k_bin_tile_sums(DecodeArgs a) {
    __shared__ uint32_t s_a[SIDX_WORDS];
    __shared__ uint32_t s_w[4];
    __shared__ uint64_t s_w64[4];
    if (a.job_counts[12] == 0) return;
    const uint32_t count = a.job_counts[2];
    for (uint32_t ti = blockIdx.x; ti < count; ti += gridDim.x) {
        const TileTask tt = a.tiles[ti];
        const PageDesc d = a.descs[tt.page];
        if (!d.ok || d.val_bytes != VAL_BYTES_DEFERRED) continue;
        const PageTask t = a.tasks[tt.page];
        const uint64_t N = t.num_values;
        const uint32_t ntiles = (uint32_t)((N + TILE_ROWS - 1) / TILE_ROWS);
        const uint32_t* aux = (const uint32_t*)(a.scratch + t.aux_off);
        const uint32_t gap = d.codec == SB_CODEC_FREQ ? 4 + d.vcsize : 0;
        const U32Stream is{d.isrc, aux, d.icodec, d.n_runs, N};
        const uint32_t used = d.codec == SB_CODEC_FREQ ? 0 : idx_aux_words(d.icodec, d.n_runs, N);
        const uint32_t* ent_off = aux + used;
        const uint32_t D = d.dict_n;
        uint64_t* sums = bin_tile_sums(ent_off + D + 1, ntiles);
        const uint32_t rows = (uint32_t)min((uint64_t)TILE_ROWS, N - (uint64_t)tt.tile * TILE_ROWS);
        __syncthreads();
        u32_tile_to_lds(is, tt.tile, rows, s_a, s_w);
        uint64_t acc = 0;
        bool bad = false;
        for (uint32_t i = threadIdx.x; i < rows; i += WG) {
            const uint32_t k = s_a[sidx((int)i)];
            if (k >= D) {
                bad = true;
                break;
            }
            const uint64_t pr = ldu64((const uint8_t*)(ent_off + k));
            acc += (uint32_t)(pr >> 32) - (uint32_t)pr - 8 - (k == 0 ? gap : 0);
        }
        if (bad) raise(a.status, SB_ERR_OUT_OF_SPEC, tt.page, 222);
        const uint64_t total = wg_sum64(acc, s_w64);
        if (threadIdx.x == 0) sums[tt.tile] = total;
    }
}
// exclusive prefix of a deferred page's tile totals -> tile_bytes, the page's val_bytes / off_last; one wave per page
__global__ void __launch_bounds__(WG) k_bin_tile_scan(DecodeArgs a) {
    if (a.job_counts[12] == 0) return;
    const uint32_t lane = threadIdx.x & 63;
    for (uint32_t p = blockIdx.x * (WG / 64) + (threadIdx.x >> 6); p < a.n_pages; p += gridDim.x * (WG / 64)) {
        const PageDesc d = a.descs[p];
        if (!d.ok || d.val_bytes != VAL_BYTES_DEFERRED) continue;
        const PageTask t = a.tasks[p];
        const uint64_t N = t.num_values;
        const uint32_t ntiles = (uint32_t)((N + TILE_ROWS - 1) / TILE_ROWS);
        uint32_t* aux = (uint32_t*)(a.scratch + t.aux_off);
        const uint32_t used = d.codec == SB_CODEC_FREQ ? 0 : idx_aux_words(d.icodec, d.n_runs, N);
        uint32_t* tile_bytes = aux + used + d.dict_n + 1;
        const uint64_t* sums = bin_tile_sums(tile_bytes, ntiles);
        uint64_t carry = 0;
        for (uint32_t t0 = 0; t0 < ntiles; t0 += 64) {
            const uint64_t v = t0 + lane < ntiles ? sums[t0 + lane] : 0;
            uint64_t incl = v;
            for (int o = 1; o < 64; o <<= 1) {
                const uint64_t u = __shfl_up(incl, o, 64);
                if ((int)lane >= o) incl += u;
            }
            if (t0 + lane < ntiles) tile_bytes[t0 + lane] = (uint32_t)(carry + incl - v);
            carry += __shfl(incl, 63, 64);
        }
        if (lane == 0) {
            tile_bytes[ntiles] = (uint32_t)carry;
            a.descs[p].val_bytes = carry;
            a.descs[p].off_last = carry;
        }
    }
}

// -------------------------------------------------------------------------------- colscan
// binary columns: value-byte base and offset base of each page (the running `last` of
// decompress_binary, binary/mod.rs:121,136-144, and values.len()), queue-B inflate jobs for
// compressed values blocks, and the column's total value bytes.
// One WAVE per column, a lane per page, 64 pages per step: the bases are exclusive wave scans of the pages' value bytes and
// last offsets plus the carry of the steps before.  (One thread per column walked its pages one dependent load after the
// other: 76 us for the 153 pages of a C4 column, and a column of many short pages by the millisecond.)
__device__ __forceinline__ void colscan_column(const DecodeArgs& a, uint64_t* col_values_len, const uint32_t ci) {
    const uint32_t lane = threadIdx.x & 63;
    const ColDesc c = a.cols[ci];
    if (!is_binary(c.ptype)) {
        if (lane == 0) col_values_len[ci] = c.ptype == SB_TYPE_BOOLEAN ? (c.rows + 7) / 8 : c.rows * c.width;
        return;
    }
    uint64_t vbase = 0, obase = 0;   // (wave-uniform carries)
    for (uint32_t k0 = 0; k0 < c.n_pages; k0 += 64) {
        const bool in = k0 + lane < c.n_pages;
        const uint32_t p = c.first_page + min(k0 + lane, c.n_pages - 1);
        const PageDesc d = a.descs[p];
        const bool was_ok = in && d.ok;
        // a page that was not ok adds nothing; one that does not fit still counts (the column's values_len is reported)
        const uint64_t vb = was_ok ? d.val_bytes : 0, ob = was_ok ? d.off_last : 0;
        uint64_t vi = vb, oi = ob;
        for (int o = 1; o < 64; o <<= 1) {
            const uint64_t uv = __shfl_up(vi, o, 64), uo = __shfl_up(oi, o, 64);
            if ((int)lane >= o) {
                vi += uv;
                oi += uo;
            }
        }
        const uint64_t my_vbase = vbase + vi - vb;
        // page 0's offsets are taken verbatim (incl. offsets[0]); later pages add the running last offset
        const uint64_t my_obase = obase + oi - ob;
        // a page whose value bytes do not fit the caller's buffer is not expanded at all (the tile kernels
        // skip !ok pages): nothing is written past values_cap, the column's values_len is still reported
        const bool fits = my_vbase + d.val_bytes <= c.values_cap;
        if (in) {
            a.descs[p].val_base = my_vbase;
            a.descs[p].off_base = my_obase;
            if (!fits && d.ok) a.descs[p].ok = 0;
        }
        const bool job = was_ok && fits && is_basic(d.codec) && d.codec != SB_CODEC_NONE && !(d.codec == SB_CODEC_ZSTD && a.jobs_z);
        push_job_if(job, a.jobs_b, a.job_counts + 1, d.vbody, d.vcsize, c.values + my_vbase, d.vusize, d.codec, p);
        vbase += __shfl(vi, 63, 64);
        obase += __shfl(oi, 63, 64);
    }
    if (lane == 0) {
        col_values_len[ci] = vbase;
        if (vbase > c.values_cap) raise(a.status, SB_ERR_INVALID, c.first_page, 300);
    }
}
__global__ void __launch_bounds__(64) k_colscan(DecodeArgs a, uint64_t* col_values_len) {
    const uint32_t ci = blockIdx.x;   // (one wave per column)
    if (ci < a.n_cols) colscan_column(a, col_values_len, ci);
    // queue B is complete now (k_parse's deferred payloads + the value blocks queued above): split its multi-frame entries
    if (!a.sizes_only && last_workgroup_done(&a.job_counts[6]) && threadIdx.x == 0)   // queue B is complete: its length for k_zstd_split
        a.job_counts[9] = __hip_atomic_load(&a.job_counts[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// -------------------------------------------------------------------------------- expand
// copy `n` bytes from an arbitrarily aligned source to dst; the bulk moves as 16-byte stores
__device__ __forceinline__ void tile_copy_bytes(uint8_t* dst, const uint8_t* src, uint32_t n) {
    const int t = threadIdx.x;
    uint32_t head = (uint32_t)((16 - ((uintptr_t)dst & 15)) & 15);
    if (head > n) head = n;
    if ((uint32_t)t < head) dst[t] = src[t];
    const uint32_t nvec = (n - head) >> 4;
    uint8_t* d16 = dst + head;
    const uint8_t* s16 = src + head;
    uint32_t i = t;
    for (; i + 3 * WG < nvec; i += 4 * WG) {  // 4 loads in flight per lane
        u32x4 v0 = ldu128(s16 + (uint64_t)i * 16);
        u32x4 v1 = ldu128(s16 + (uint64_t)(i + WG) * 16);
        u32x4 v2 = ldu128(s16 + (uint64_t)(i + 2 * WG) * 16);
        u32x4 v3 = ldu128(s16 + (uint64_t)(i + 3 * WG) * 16);
        *(u32x4*)(d16 + (uint64_t)i * 16) = v0;
        *(u32x4*)(d16 + (uint64_t)(i + WG) * 16) = v1;
        *(u32x4*)(d16 + (uint64_t)(i + 2 * WG) * 16) = v2;
        *(u32x4*)(d16 + (uint64_t)(i + 3 * WG) * 16) = v3;
    }
    for (; i < nvec; i += WG) *(u32x4*)(d16 + (uint64_t)i * 16) = ldu128(s16 + (uint64_t)i * 16);
    const uint32_t tail0 = head + (nvec << 4);
    if (tail0 + t < n) dst[tail0 + t] = src[tail0 + t];
}

// write `rows` values of W bytes, row i produced by get(i); 16-byte stores when the
// destination allows it
template <int W, class F>
__device__ __forceinline__ void emit_rows(uint8_t* dst, uint32_t rows, F get) {
    const int t = threadIdx.x;
    if constexpr (W < 16) {
        constexpr int RPL = 16 / W;
        if (((uintptr_t)dst & 15) == 0) {
            if (rows == TILE_ROWS) {  // full tile: every gather of the thread is issued before its first store
                constexpr int G = TILE_ROWS / RPL / WG;
                Val<W> v[G][RPL];
#pragma unroll
                for (int j = 0; j < G; j++)
#pragma unroll
                    for (int r = 0; r < RPL; r++) v[j][r] = get((uint32_t)(t + j * WG) * RPL + r);
#pragma unroll
                for (int j = 0; j < G; j++) {
                    u32x4 pk;
                    __builtin_memcpy(&pk, v[j], 16);
                    stu128(dst + (uint64_t)(t + j * WG) * 16, pk);
                }
                return;
            }
            const uint32_t full = rows / RPL;
            for (uint32_t g = t; g < full; g += WG) {
                Val<W> v[RPL];
#pragma unroll
                for (int r = 0; r < RPL; r++) v[r] = get(g * RPL + r);
                u32x4 pk;
                __builtin_memcpy(&pk, v, 16);
                stu128(dst + (uint64_t)g * 16, pk);
            }
            for (uint32_t i = full * RPL + t; i < rows; i += WG) st_val<W>(dst + (uint64_t)i * W, get(i));
            return;
        }
    }
    for (uint32_t i = t; i < rows; i += WG) st_val<W>(dst + (uint64_t)i * W, get(i));
}

// OR `nbits` (<= 32) bits `v` into the LSB-first bitmap at bit position `pos`
__device__ __forceinline__ void bitmap_put(uint8_t* bm, uint64_t pos, uint32_t v, uint32_t nbits, bool aligned) {
    if (nbits < 32) v &= (1u << nbits) - 1;
    uint32_t* w = (uint32_t*)bm + (pos >> 5);
    const uint32_t sh = (uint32_t)(pos & 31);
    if (aligned || (sh == 0 && nbits == 32)) {  // word owned by this page alone: plain store
        *w = v;
        return;
    }
    if (v << sh) atomicOr(w, v << sh);
    if (sh && (v >> (32 - sh))) atomicOr(w + 1, v >> (32 - sh));
}

// copy rows [r0, r0+rows) of an LSB-first source bitmap (starting at bit 0 of src) to the
// destination bitmap at bit (dst_bit0 + r0 ...)
// 32 bits of a page's def-level section starting at (byte-aligned) bit `sb`
__device__ __forceinline__ uint32_t tile_bits_load(const uint8_t* src, uint64_t sb, uint64_t src_total_bits) {
    const uint8_t* p = src + (sb >> 3);
    const uint64_t bytes_left = ((src_total_bits + 7) >> 3) - (sb >> 3);
    if (bytes_left >= 4) return ldu32(p);
    uint32_t v = 0;
    for (uint32_t b = 0; b < bytes_left; b++) v |= (uint32_t)p[b] << (8 * b);
    return v;
}
__device__ __forceinline__ void tile_copy_bits(uint8_t* dst_bm, uint64_t dst_bit0, const uint8_t* src, uint64_t r0,
                                               uint32_t rows, uint64_t src_total_bits, bool aligned) {
    const int t = threadIdx.x;
    const uint32_t ngroups = (rows + 31) / 32;
    for (uint32_t g = t; g < ngroups; g += WG) {
        const uint64_t sb = r0 + (uint64_t)g * 32;  // r0 is a multiple of TILE_ROWS => byte aligned
        const uint32_t nb = min(32u, rows - g * 32);
        const uint8_t* p = src + (sb >> 3);
        uint32_t v;
        const uint64_t bytes_left = ((src_total_bits + 7) >> 3) - (sb >> 3);
        if (bytes_left >= 4) {
            v = ldu32(p);
        } else {
            v = 0;
            for (uint32_t b = 0; b < bytes_left; b++) v |= (uint32_t)p[b] << (8 * b);
        }
        bitmap_put(dst_bm, dst_bit0 + sb, v, nb, aligned);
    }
}

template <int W>
__device__ void expand_prim(const ColDesc& c, const PageTask& t, const PageDesc& d, uint32_t tile, uint32_t rows,
                            uint8_t* scratch, uint32_t* s_a, uint32_t* s_w, Status* st, uint32_t page, uint32_t tk0,
                            uint32_t tkend, const RleStarts& rle_pre) {
    const uint64_t r0 = (uint64_t)tile * TILE_ROWS;
    uint8_t* dst = c.values + (t.out_row + r0) * W;
    const uint32_t* aux = (const uint32_t*)(scratch + t.aux_off);
    switch (d.codec) {
        case SB_CODEC_NONE:  // integer/mod.rs:97-107 + basic.rs:67-70
            tile_copy_bytes(dst, d.src + r0 * W, rows * W);
            break;
        case SB_CODEC_LZ4:
        case SB_CODEC_ZSTD:
        case SB_CODEC_SNAPPY:
            break;  // inflated straight into place by k_inflate
        case SB_CODEC_FREQ:        // top value everywhere; k_freq_scatter overwrites the exception rows
        case SB_CODEC_ONEVALUE: {  // integer/one_value.rs:77-94
            const Val<W> v = ld_val<W>(d.body);
            emit_rows<W>(dst, rows, [&](uint32_t) { return v; });
            break;
        }
        case SB_CODEC_RLE: {  // integer/rle.rs:106-134
            rle_tile_runidx_k(aux, tk0, tkend, tile, rows, s_a, s_w, rle_pre);
            DTL(3);
            const uint32_t k0 = tk0;
            const uint8_t* body = d.body;
            emit_rows<W>(dst, rows, [&](uint32_t i) {
                uint32_t k = k0 + s_a[sidx((int)i)];
                return ld_val<W>(body + (uint64_t)k * (4 + W) + 4);
            });
            DTL(4);
            break;
        }
        case SB_CODEC_DICT: {  // integer/dict.rs:75-103
            U32Stream is{d.isrc, aux, d.icodec, d.n_runs, t.num_values};
            u32_tile_to_lds(is, tile, rows, s_a, s_w);
            const uint8_t* dict = d.dict;
            const uint32_t D = d.dict_n;
            bool bad = false;
            emit_rows<W>(dst, rows, [&](uint32_t i) {
                uint32_t k = s_a[sidx((int)i)];
                if (k >= D) {
                    bad = true;
                    k = 0;
                }
                return ld_val<W>(dict + (uint64_t)k * W);
            });
            if (bad) raise(st, SB_ERR_OUT_OF_SPEC, page, 400);
            break;
        }
        case SB_CODEC_BITPACKING:
        case SB_CODEC_DELTA_BITPACKING: {  // integer/bp.rs:66-86, delta_bp.rs:70-92
            if constexpr (W == 4) {
                U32Stream vs{d.body, aux, d.codec, 0, t.num_values};
                u32_tile_to_lds(vs, tile, rows, s_a, s_w);
                emit_rows<4>(dst, rows, [&](uint32_t i) {
                    Val<4> v;
                    v.x = s_a[sidx((int)i)];
                    return v;
                });
            }
            break;
        }
        default:
            break;
    }
}

__device__ void expand_bool(const ColDesc& c, const PageTask& t, const PageDesc& d, uint32_t tile, uint32_t rows,
                            uint8_t* scratch, uint32_t* s_a, uint32_t* s_w) {
    const uint64_t r0 = (uint64_t)tile * TILE_ROWS;
    const int tid = threadIdx.x;
    if (is_basic(d.codec)) {  // boolean/mod.rs:87-92
        tile_copy_bits(c.values, t.out_row, d.src, r0, rows, t.num_values, c.bits_aligned);
    } else if (d.codec == SB_CODEC_ONEVALUE) {  // boolean/one_value.rs:54-61
        const uint32_t v = d.body[0] > 0 ? 0xFFFFFFFFu : 0u;
        const uint32_t ngroups = (rows + 31) / 32;
        for (uint32_t g = tid; g < ngroups; g += WG)
            bitmap_put(c.values, t.out_row + r0 + (uint64_t)g * 32, v, min(32u, rows - g * 32), c.bits_aligned);
    } else if (d.codec == SB_CODEC_RLE) {  // boolean/rle.rs:41-55
        const uint32_t* aux = (const uint32_t*)(scratch + t.aux_off);
        rle_tile_runidx(aux, d.n_runs, t.num_values, tile, rows, s_a, s_w);
        const uint32_t k0 = (aux + d.n_runs + 1)[tile];
        for (uint32_t i0 = 0; i0 < TILE_ROWS; i0 += WG) {
            const uint32_t i = i0 + tid;
            bool bit = false;
            if (i < rows) {
                uint32_t k = k0 + s_a[sidx((int)i)];
                bit = d.body[(uint64_t)k * 5 + 4] != 0;
            }
            const uint64_t m = __ballot(bit);
            if ((tid & 31) == 0 && i < rows) {
                const uint32_t half = (tid & 32) ? (uint32_t)(m >> 32) : (uint32_t)m;
                bitmap_put(c.values, t.out_row + r0 + i, half, min(32u, rows - i), c.bits_aligned);
            }
        }
    }
}

template <class O>
__device__ void expand_binary(const ColDesc& c, const PageTask& t, const PageDesc& d, uint32_t tile, uint32_t rows,
                              uint8_t* scratch, uint32_t* s_a, uint32_t* s_len, uint32_t* s_w, Status* st,
                              uint32_t page) {
    const uint64_t r0 = (uint64_t)tile * TILE_ROWS;
    const int tid = threadIdx.x;
    const uint64_t N = t.num_values;
    O* out_off = (O*)c.offsets;
    const bool first_page = t.out_row == 0;
    if (is_basic(d.codec)) {
        // offsets: page 0 verbatim (N+1 entries); later pages drop their leading entry and add
        // the running last offset (binary/mod.rs:136-144)
        const uint8_t* so = d.src;
        for (uint32_t i = tid; i < rows; i += WG) {
            O v;
            __builtin_memcpy(&v, so + (r0 + i + 1) * sizeof(O), sizeof(O));
            out_off[t.out_row + r0 + i + 1] = (O)(v + (O)d.off_base);
        }
        if (first_page && tile == 0 && tid == 0) {
            O v;
            __builtin_memcpy(&v, so, sizeof(O));
            out_off[0] = v;
        }
        // values: this tile copies its share of the page's value bytes
        if (d.codec == SB_CODEC_NONE) {
            const uint32_t ntiles = (uint32_t)((N + TILE_ROWS - 1) / TILE_ROWS);
            const uint64_t per = (((uint64_t)d.vusize + ntiles - 1) / ntiles + 15) & ~(uint64_t)15;
            const uint64_t b0 = min((uint64_t)d.vusize, per * tile), b1 = min((uint64_t)d.vusize, per * (tile + 1));
            if (b1 > b0) tile_copy_bytes(c.values + d.val_base + b0, d.vbody + b0, (uint32_t)(b1 - b0));
        }
        return;
    }
    if (tile == 0 && first_page && tid == 0) out_off[0] = 0;
    if (d.codec == SB_CODEC_ONEVALUE) {  // binary/one_value.rs:70-97
        const uint32_t len = d.dict_n;
        for (uint32_t i = tid; i < rows; i += WG)
            out_off[t.out_row + r0 + i + 1] = (O)(d.off_base + (uint64_t)len * (r0 + i + 1));
        uint8_t* vdst = c.values + d.val_base + (uint64_t)len * r0;
        const uint64_t total = (uint64_t)len * rows;
        if (len) {
            for (uint64_t b = tid; b < total; b += WG) vdst[b] = d.dict[b % len];
        }
        return;
    }
    if (d.codec == SB_CODEC_DICT || d.codec == SB_CODEC_FREQ) {  // binary/dict.rs:95-140; Freq as a virtual Dict page
        const uint32_t* aux = (const uint32_t*)(scratch + t.aux_off);
        const uint32_t gap = d.codec == SB_CODEC_FREQ ? 4 + d.vcsize : 0;
        U32Stream is{d.isrc, aux, d.icodec, d.n_runs, N};
        const uint32_t used = d.codec == SB_CODEC_FREQ ? 0 : idx_aux_words(d.icodec, d.n_runs, N);
        const uint32_t* ent_off = aux + used;
        const uint32_t D = d.dict_n;
        const uint32_t* tile_bytes = ent_off + D + 1;
        u32_tile_to_lds(is, tile, rows, s_a, s_w);
        // A divergent access costs the CU's address unit a cycle per lane, so the tile is sized in gathers per row: the
        // entry's offset pair with ONE 8-byte load (its offset then replaces the index in s_a), and the bytes with at most
        // two loads and two stores for strings of up to 32 bytes (the second move overlaps the first instead of a byte tail).
        for (uint32_t i = tid; i < TILE_ROWS; i += WG) {
            uint32_t len = 0, eo = 0;
            if (i < rows) {
                const uint32_t k = s_a[sidx((int)i)];
                if (k < D) {
                    const uint64_t pr = ldu64((const uint8_t*)(ent_off + k));
                    eo = (uint32_t)pr;
                    len = (uint32_t)(pr >> 32) - eo - 8 - (k == 0 ? gap : 0);
                    eo += 8;
                } else {
                    eo = 0xFFFFFFFFu;
                }
            }
            s_len[sidx((int)i)] = len;
            s_a[sidx((int)i)] = eo;
        }
        __syncthreads();
        tile_incl_scan(s_len, s_w);
        const uint64_t tb = tile_bytes[tile];
        uint8_t* vdst = c.values + d.val_base + tb;
        for (uint32_t i = tid; i < rows; i += WG) {
            const uint32_t endb = s_len[sidx((int)i)];
            out_off[t.out_row + r0 + i + 1] = (O)(d.off_base + tb + endb);
            const uint32_t eo = s_a[sidx((int)i)];
            if (eo != 0xFFFFFFFFu) {
                const uint32_t len = endb - (i ? s_len[sidx((int)i - 1)] : 0u);
                const uint8_t* sp = d.dict + eo;
                uint8_t* dp = vdst + (endb - len);
                if (len >= 16) {
                    uint32_t b = 0;
                    for (; b + 16 <= len; b += 16) stu128(dp + b, ldu128(sp + b));
                    if (b < len) stu128(dp + len - 16, ldu128(sp + len - 16));
                } else if (len >= 8) {
                    const uint64_t v0 = ldu64(sp), v1 = ldu64(sp + len - 8);
                    stu64(dp, v0);
                    stu64(dp + len - 8, v1);
                } else if (len >= 4) {
                    const uint32_t v0 = ldu32(sp), v1 = ldu32(sp + len - 4);
                    stu32(dp, v0);
                    stu32(dp + len - 4, v1);
                } else if (len >= 2) {
                    const uint32_t v0 = ldu16(sp), v1 = ldu16(sp + len - 2);
                    *(__attribute__((address_space(1))) uint16_t*)(dp) = (uint16_t)v0;
                    *(__attribute__((address_space(1))) uint16_t*)(dp + len - 2) = (uint16_t)v1;
                } else if (len == 1) {
                    *(gptr)dp = ldu8(sp);
                }
            }
        }
        (void)st;
        (void)page;
    }
}

// tile -> (page, tile) with an XCD-aware order: block b runs on XCD b%8, so give every XCD a
// contiguous range of the tile table (tiles of one page share dictionary / aux lines in one L2)
__device__ __forceinline__ uint32_t xcd_tile_index() {
    const uint32_t nb = gridDim.x, b = blockIdx.x;
    const uint32_t per = nb / 8, rem = nb % 8, x = b % 8, q = b / 8;
    return x * per + min(x, rem) + q;
}

// primitives + booleans (17 KB LDS: 8 workgroups per CU)
// Under a saturated memory system every dependent HBM round trip of a tile costs 2-3 us, and a
// tile's chain is what bounds this kernel (8 workgroups per CU in flight).  So the chain is kept
// short: the tile task carries the column index and, for RLE pages, the tile's run range (k_plan),
// which lets the three descriptors, the validity word and the run starts be fetched in two steps;
// only the gather of the run values follows.
constexpr uint32_t TILE_GRID = 4096;  // the tile kernels walk the compact tile list with a grid of at most this many
                                      // workgroups: the host only knows an upper bound of the list's length, and
                                      // entries that do not exist (pages expanded by k_expand_rle) should not each
                                      // cost a workgroup launch
__device__ void expand_tile(const DecodeArgs& a, uint32_t ti, uint32_t* s_a, uint32_t* s_w) {
    static_assert(TILE_ROWS / 32 <= WG, "one validity word per thread");
    DTL(0);
    const TileTask tt = a.tiles[ti];
    const PageDesc d = a.descs[tt.page];
    const PageTask t = a.tasks[tt.page];
    const ColDesc c = a.cols[tt.col];
    if (!d.ok) return;
    if (is_binary(c.ptype)) return;  // k_expand_binary
    const uint64_t r0 = (uint64_t)tt.tile * TILE_ROWS;
    const uint32_t rows = (uint32_t)min((uint64_t)TILE_ROWS, t.num_values - r0);
    DTL(1);
    // issue the validity word and (RLE) the run starts together, consume them afterwards
    const uint32_t g = threadIdx.x, ngroups = (rows + 31) / 32;
    const bool has_vb = d.def_bits && g < ngroups;
    uint32_t vb = 0;
    if (has_vb) vb = tile_bits_load(d.def_bits, r0 + (uint64_t)g * 32, t.num_values);
    RleStarts st;
    if (d.codec == SB_CODEC_RLE && c.ptype != SB_TYPE_BOOLEAN)
        st = rle_fetch_starts((const uint32_t*)(a.scratch + t.aux_off), tt.k0, tt.kend);
    if (has_vb) bitmap_put(c.validity, t.out_row + r0 + (uint64_t)g * 32, vb, min(32u, rows - g * 32), c.bits_aligned);
    DTL(2);
    if (c.ptype == SB_TYPE_BOOLEAN) {
        expand_bool(c, t, d, tt.tile, rows, a.scratch, s_a, s_w);
        return;
    }
    switch (c.width) {
        case 1:
            expand_prim<1>(c, t, d, tt.tile, rows, a.scratch, s_a, s_w, a.status, tt.page, tt.k0, tt.kend, st);
            break;
        case 2:
            expand_prim<2>(c, t, d, tt.tile, rows, a.scratch, s_a, s_w, a.status, tt.page, tt.k0, tt.kend, st);
            break;
        case 4:
            expand_prim<4>(c, t, d, tt.tile, rows, a.scratch, s_a, s_w, a.status, tt.page, tt.k0, tt.kend, st);
            break;
        case 8:
            expand_prim<8>(c, t, d, tt.tile, rows, a.scratch, s_a, s_w, a.status, tt.page, tt.k0, tt.kend, st);
            break;
        case 16:
            expand_prim<16>(c, t, d, tt.tile, rows, a.scratch, s_a, s_w, a.status, tt.page, tt.k0, tt.kend, st);
            break;
        case 32:
            expand_prim<32>(c, t, d, tt.tile, rows, a.scratch, s_a, s_w, a.status, tt.page, tt.k0, tt.kend, st);
            break;
    }
}

__global__ void __launch_bounds__(WG) k_expand(DecodeArgs a) {
    __shared__ uint32_t s_a[SIDX_WORDS];
    __shared__ uint32_t s_w[4];
    const uint32_t count = a.job_counts[2];
    for (uint32_t ti = blockIdx.x; ti < count; ti += gridDim.x) {
        expand_tile(a, ti, s_a, s_w);
        __syncthreads();
    }
}

// ---- RLE pages of <= 8-byte values, one workgroup per page (integer/rle.rs:106-134, double/rle.rs:105-135).
// The runs are walked in chunks of 4 per thread: counts are scanned into start rows (u64 carry, the
// reference adds u32 counts until the page's row count is reached), the chunk's values are staged in
// LDS, and the rows the chunk covers are written tile by tile: run starts scattered into a flag
// array, scanned into a run index per row, values gathered from LDS, 16-byte stores.  No plan pass,
// no run-start array in HBM, descriptors and validity handled once per page, and the next chunk of
// records is already in flight while the current one is expanded.
constexpr int RLE_RPT = 4;                      // runs per thread per chunk
constexpr uint32_t RLE_CHUNK = WG * RLE_RPT;    // 1024 runs

template <int W>
__device__ void expand_rle_page(const ColDesc& c, const PageTask& t, const PageDesc& d, uint32_t* s_flag, uint8_t* s_vals_raw,
                                uint32_t* s_w, uint64_t* s_w64, Status* st, uint32_t page, uint32_t part = 0, uint32_t parts = 1,
                                const uint64_t* sums = nullptr /* rows covered by each part's runs (k_rle_sums), parts > 1 */) {
    constexpr int REC = 4 + W;
    const int tid = threadIdx.x;
    const uint64_t N = t.num_values;
    uint8_t* dst = c.values + t.out_row * W;
    const uint8_t* body = d.body;
    const uint8_t* page_end = c.pages + t.in_off + t.length;
    const uint32_t max_runs = (uint32_t)((uint64_t)(page_end - body) / REC);
    Val<W>* s_vals = (Val<W>*)s_vals_raw;
    uint32_t ncnt[RLE_RPT];
    Val<W> nval[RLE_RPT];
    auto fetch = [&](uint32_t base) {
#pragma unroll
        for (int j = 0; j < RLE_RPT; j++) {
            const uint32_t k = base + (uint32_t)tid * RLE_RPT + j;
            const bool in = k < max_runs;
            const uint8_t* r = body + (uint64_t)(in ? k : 0) * REC;
            ncnt[j] = in ? ldu32(r) : 0;
            nval[j] = in ? ld_val<W>(r + 4) : Val<W>{};
        }
    };
    // a long page is shared by `parts` workgroups: each takes a range of run chunks, its first row comes from the sums of
    // the ranges before it
    const uint32_t nchunks = (max_runs + RLE_CHUNK - 1) / RLE_CHUNK, cpp = (nchunks + parts - 1) / parts;
    const uint32_t b0 = part * cpp * RLE_CHUNK;
    const bool owns_end = (uint64_t)(part + 1) * cpp >= nchunks;
    const uint64_t b1 = owns_end ? ~0ull : (uint64_t)(part + 1) * cpp * RLE_CHUNK;
    if (part && b0 >= max_runs) return;   // (the part that owns the end reports runs that stop short of N)
    uint64_t carry = 0;  // rows covered by the chunks before this one
    for (uint32_t q = 0; q < part; q++) carry += sums[q];
    if (max_runs) fetch(b0);
    for (uint64_t base64 = b0; carry < N && base64 < b1; base64 += RLE_CHUNK) {
        const uint32_t base = (uint32_t)base64;
        if (base64 >= max_runs) {
            if (tid == 0) raise(st, SB_ERR_IO, page, 200);  // runs end before N rows (read_u32 EOF upstream)
            return;
        }
        uint32_t cnt[RLE_RPT];
#pragma unroll
        for (int j = 0; j < RLE_RPT; j++) {
            cnt[j] = ncnt[j];
            s_vals[tid * RLE_RPT + j] = nval[j];
        }
        if (base + RLE_CHUNK < max_runs) fetch(base + RLE_CHUNK);
        // start rows of my runs
        uint64_t loc[RLE_RPT], run = 0;
#pragma unroll
        for (int j = 0; j < RLE_RPT; j++) {
            run += cnt[j];
            loc[j] = run;
        }
        const uint64_t incl = wave_incl_scan64(run);
        __syncthreads();  // previous chunk's readers of s_w64 / s_vals are done; my s_vals stores are ordered before the tile loop's barriers
        if ((tid & 63) == 63) s_w64[tid >> 6] = incl;
        __syncthreads();
        uint64_t pre = carry + incl - run;
        const int w = tid >> 6;
        if (w > 0) pre += s_w64[0];
        if (w > 1) pre += s_w64[1];
        if (w > 2) pre += s_w64[2];
        const uint64_t chunk_total = s_w64[0] + s_w64[1] + s_w64[2] + s_w64[3];
        uint64_t start[RLE_RPT];
#pragma unroll
        for (int j = 0; j < RLE_RPT; j++) start[j] = pre + (j ? loc[j - 1] : 0);
        const uint64_t S0 = carry, S1 = min(N, carry + chunk_total);
        // the run that reaches row N must end exactly there: upstream pushes whole runs and then
        // asserts the decoded length (read/array/integer.rs:81)
#pragma unroll
        for (int j = 0; j < RLE_RPT; j++)
            if (start[j] < N && start[j] + cnt[j] > N) raise(st, SB_ERR_OUT_OF_SPEC, page, 202);
        for (uint64_t tile_lo = S0 / TILE_ROWS * TILE_ROWS; tile_lo < S1; tile_lo += TILE_ROWS) {
            const uint64_t lo = max(S0, tile_lo), hi = min(S1, tile_lo + TILE_ROWS);
            static_assert(SIDX_WORDS % 4 == 0, "16-byte clears");
            for (int i = tid; i < SIDX_WORDS / 4; i += WG) ((u32x4*)s_flag)[i] = u32x4{0, 0, 0, 0};
            // A = (runs of the chunk that start at or before row lo) - 1: the run covering row lo
            uint32_t le = 0;
#pragma unroll
            for (int j = 0; j < RLE_RPT; j++) le += (base + (uint32_t)tid * RLE_RPT + j < max_runs && start[j] <= lo) ? 1u : 0u;
            __syncthreads();
#pragma unroll
            for (int j = 0; j < RLE_RPT; j++)
                if (base + (uint32_t)tid * RLE_RPT + j < max_runs && start[j] > lo && start[j] < hi)
                    atomicAdd(&s_flag[sidx((int)(start[j] - tile_lo))], 1u);
            // workgroup sum of `le` (s_w is free until tile_incl_scan uses it)
            uint32_t v = le;
#pragma unroll
            for (int dlt = 32; dlt > 0; dlt >>= 1) v += __shfl_down(v, dlt, 64);
            if ((tid & 63) == 0) s_w[tid >> 6] = v;
            __syncthreads();
            const uint32_t A = s_w[0] + s_w[1] + s_w[2] + s_w[3] - 1;
            __syncthreads();
            tile_incl_scan(s_flag, s_w);
            const uint32_t off = (uint32_t)(lo - tile_lo);
            emit_rows<W>(dst + lo * W, (uint32_t)(hi - lo), [&](uint32_t i) {
                const uint32_t k = A + s_flag[sidx((int)(off + i))];
                return s_vals[k];
            });
            __syncthreads();  // s_flag / s_w are reused by the next tile
        }
        carry += chunk_total;
        if (chunk_total == 0 && base + RLE_CHUNK >= max_runs && carry < N) {
            if (tid == 0) raise(st, SB_ERR_IO, page, 200);
            return;
        }
    }
}

// the page's def-level bits -> the column's validity bitmap (all tiles at once)
__device__ void page_copy_bits(uint8_t* dst_bm, uint64_t dst_bit0, const uint8_t* src, uint64_t N, bool aligned) {
    constexpr int U = 8;
    const uint32_t nwords = (uint32_t)((N + 31) / 32);
    for (uint32_t g0 = threadIdx.x; g0 < nwords; g0 += WG * U) {
        uint32_t v[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t g = g0 + u * WG;
            v[u] = g < nwords ? tile_bits_load(src, (uint64_t)g * 32, N) : 0;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t g = g0 + u * WG;
            if (g < nwords) bitmap_put(dst_bm, dst_bit0 + (uint64_t)g * 32, v[u], (uint32_t)min((uint64_t)32, N - (uint64_t)g * 32), aligned);
        }
    }
}

__global__ void __launch_bounds__(WG) k_expand_rle(DecodeArgs a) {
    __shared__ __attribute__((aligned(16))) uint32_t s_flag[SIDX_WORDS];
    __shared__ __attribute__((aligned(16))) uint8_t s_vals[RLE_CHUNK * 8];
    __shared__ uint32_t s_w[4];
    __shared__ uint64_t s_w64[4];
    if (a.job_counts[4] == 0) return;  // no RLE page of <= 8-byte values in this call
    const uint32_t p = blockIdx.x;
    const PageDesc d = a.descs[p];
    const PageTask t = a.tasks[p];
    const ColDesc c = a.cols[t.col];
    if (!rle_by_page(c, d)) return;
    const uint32_t part = blockIdx.y, parts = gridDim.y;
    const uint64_t* sums = parts > 1 ? a.rle_sums + (uint64_t)p * parts : nullptr;
    if (d.def_bits && part == 0) page_copy_bits(c.validity, t.out_row, d.def_bits, t.num_values, c.bits_aligned);
    switch (c.width) {
        case 1:
            expand_rle_page<1>(c, t, d, s_flag, s_vals, s_w, s_w64, a.status, p, part, parts, sums);
            break;
        case 2:
            expand_rle_page<2>(c, t, d, s_flag, s_vals, s_w, s_w64, a.status, p, part, parts, sums);
            break;
        case 4:
            expand_rle_page<4>(c, t, d, s_flag, s_vals, s_w, s_w64, a.status, p, part, parts, sums);
            break;
        default:
            expand_rle_page<8>(c, t, d, s_flag, s_vals, s_w, s_w64, a.status, p, part, parts, sums);
            break;
    }
}

// rows covered by the runs of every part of a long RLE page (grid = pages x parts, parts > 1: see expand_rle_page)
__global__ void __launch_bounds__(WG) k_rle_sums(DecodeArgs a) {
    __shared__ uint64_t s_w64[4];
    if (a.job_counts[4] == 0) return;
    const uint32_t p = blockIdx.x, part = blockIdx.y, parts = gridDim.y;
    const PageDesc d = a.descs[p];
    const PageTask t = a.tasks[p];
    const ColDesc c = a.cols[t.col];
    if (!rle_by_page(c, d)) return;
    const uint32_t REC = 4 + c.width;
    const uint8_t* page_end = c.pages + t.in_off + t.length;
    const uint32_t max_runs = (uint32_t)((uint64_t)(page_end - d.body) / REC);
    const uint32_t nchunks = (max_runs + RLE_CHUNK - 1) / RLE_CHUNK, cpp = (nchunks + parts - 1) / parts;
    const uint64_t r0 = (uint64_t)part * cpp * RLE_CHUNK, r1 = min((uint64_t)max_runs, r0 + (uint64_t)cpp * RLE_CHUNK);
    uint64_t sum = 0;
    for (uint64_t k = r0 + threadIdx.x; k < r1; k += WG) sum += ldu32(d.body + k * REC);
    sum = wave_incl_scan64(sum);
    if ((threadIdx.x & 63) == 63) s_w64[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) a.rle_sums[(uint64_t)p * parts + part] = s_w64[0] + s_w64[1] + s_w64[2] + s_w64[3];
}

__device__ void expand_binary_tile(const DecodeArgs& a, uint32_t ti, uint32_t* s_a, uint32_t* s_len, uint32_t* s_w) {
    const TileTask tt = a.tiles[ti];
    const PageDesc d = a.descs[tt.page];
    if (!d.ok) return;
    const PageTask t = a.tasks[tt.page];
    const ColDesc c = a.cols[t.col];
    if (!is_binary(c.ptype)) return;
    const uint64_t r0 = (uint64_t)tt.tile * TILE_ROWS;
    const uint32_t rows = (uint32_t)min((uint64_t)TILE_ROWS, t.num_values - r0);
    if (d.def_bits) tile_copy_bits(c.validity, t.out_row, d.def_bits, r0, rows, t.num_values, c.bits_aligned);
    if (c.ptype == SB_TYPE_BINARY)
        expand_binary<int32_t>(c, t, d, tt.tile, rows, a.scratch, s_a, s_len, s_w, a.status, tt.page);
    else
        expand_binary<int64_t>(c, t, d, tt.tile, rows, a.scratch, s_a, s_len, s_w, a.status, tt.page);
}

__global__ void __launch_bounds__(WG) k_expand_binary(DecodeArgs a) {
    __shared__ uint32_t s_a[SIDX_WORDS];
    __shared__ uint32_t s_len[SIDX_WORDS];
    __shared__ uint32_t s_w[4];
    const uint32_t count = a.job_counts[2];
    for (uint32_t ti = blockIdx.x; ti < count; ti += gridDim.x) {
        expand_binary_tile(a, ti, s_a, s_len, s_w);
        __syncthreads();
    }
}

// -------------------------------------------------------------------------------- launcher
// Second pass of a Freq page: out[index of the k-th set bit] = exceptions[k] (freq.rs:120-122).
// One workgroup per page walks the Roaring containers (array / bitmap / run, portable format).
__global__ void __launch_bounds__(WG) k_freq_scatter(const FreqEntry* entries, const uint64_t* ex_off, const uint8_t* ex_base,
                                                       Status* st) {
    __shared__ uint32_t s_a[4];
    __shared__ uint32_t s_w[4];
    const FreqEntry fe = entries[blockIdx.x];
    const uint8_t* ex = ex_base + ex_off[blockIdx.x];
    const uint32_t w = fe.width;
    const uint64_t cum = roaring_walk(fe.roaring, fe.roaring_len, s_a, s_w, [&](uint64_t row, uint64_t k) {
        if (row >= fe.rows) {
            raise(st, SB_ERR_OUT_OF_SPEC, fe.page, 410);  // exception index out of bounds
            return;
        }
        const uint8_t* s = ex + k * w;
        uint8_t* d = fe.out + row * w;
        for (uint32_t b = 0; b < w; b++) d[b] = s[b];
    });
    if (cum != fe.n_exceptions && threadIdx.x == 0) raise(st, SB_ERR_EXTERNAL, fe.page, 411);  // malformed Roaring bitmap
}

void launch_freq_scatter(sb_ctx* ctx, const FreqEntry* entries, uint32_t n, const uint64_t* ex_off, const uint8_t* ex_base) {
    if (n) k_freq_scatter<<<n, WG, 0, ctx->stream>>>(entries, ex_off, ex_base, ctx->d_status);
}

// The block-parallel Zstd pipeline (launched when the context has met Zstd pages: a.zb.blocks != null).  Entropy stages
// for the frames of queue A and — calls with binary columns — of queue Z in one pass; the frames of queue A are executed
// right away, those of queue Z after k_colscan (launch_zb_exec).  Frames the pipeline takes are marked CODEC_ZB; k_inflate,
// launched after it, decodes the rest (and the frames handed back).
// multi-frame Zstd buffers of a queue -> one entry per frame (long buffers: frames found by the scan kernels)
static void launch_zstd_split(const DecodeArgs& a, InflateJob* q, uint32_t* cnt, const uint32_t* n0_p, uint32_t cap, hipStream_t s) {
    k_zstd_split<<<(2 * a.n_pages + WG - 1) / WG, WG, 0, s>>>(q, cnt, n0_p, cap, a.status, a);
    if (a.zs_segs) {
        k_zsplit_scan<<<dim3(1024, ZS_LIST), WG, 0, s>>>(q, a);
        k_zsplit_chain<<<ZS_LIST, WG, 0, s>>>(q, cnt, cap, a);
        (void)hipMemsetAsync(a.zs_hdr, 0, 4, s);   // the list is per queue (the segment records are handed out once per call)
    }
}
static RelCtx rel_ctx(const DecodeArgs& a) {
    RelCtx rc;
    rc.cols = a.cols;
    rc.tasks = a.tasks;
    rc.descs = a.descs;
    return rc;
}
static void launch_zb_exec(sb_ctx* ctx, const DecodeArgs& a, uint32_t queue) {
    if (!a.zb.blocks) return;
    // the two executors work on disjoint frames: side by side
    const bool multi = a.zb.wg_exec && !ctx->profile && side_streams(ctx);
    if (multi) side_fork(ctx, 1u);
    {
        KScope k(ctx, queue == 0 ? "zb_exec" : "zb_exec(values)");
        zb_exec<<<std::min<uint32_t>(a.zb.frame_cap, 4096u), 64, 0, ctx->stream>>>(queue == 0 ? a.jobs_a : a.jobs_z, a.status, a.zb, queue, rel_ctx(a));
    }
    if (a.zb.wg_exec) {
        KScope k(ctx, queue == 0 ? "zb_exec_wg" : "zb_exec_wg(values)");
        zb_exec_wg<<<std::min<uint32_t>(a.zb.frame_cap, 1024u), ZX_T, 0, multi ? ctx->side[0] : ctx->stream>>>(queue == 0 ? a.jobs_a : a.jobs_z, a.status, a.zb, queue, rel_ctx(a));
    }
    if (multi) side_join(ctx, 1u);
}
static void launch_zb(sb_ctx* ctx, const DecodeArgs& a) {
    if (!a.zb.blocks) return;
    hipStream_t s = ctx->stream;
    (void)hipMemsetAsync(a.zb.counters, 0, 16 * sizeof(uint32_t), s);
    {
        KScope k(ctx, "zb_scan");
        zb_scan<<<std::min<uint32_t>((a.job_cap_a + WG - 1) / WG, 1024u), WG, 0, s>>>(a.jobs_a, a.job_counts, a.zb, 0u, a.job_cap_a);
        if (a.jobs_z) zb_scan<<<std::min<uint32_t>((a.job_cap_a + WG - 1) / WG, 1024u), WG, 0, s>>>(a.jobs_z, a.job_counts + 10, a.zb, 1u, a.job_cap_a);
        zb_hdr<<<std::min<uint32_t>((a.zb.block_cap + WG - 1) / WG, 1024u), WG, 0, s>>>(a.zb);
    }
    // literals and sequences of a block are independent of each other (different pools): side by side on two streams
    const bool multi = !ctx->profile && side_streams(ctx);
    // Both are pools of 768 one-wave workgroups with ~53 KB of LDS each: either fills every CU's LDS, so the one on the
    // call's stream runs first and the other (on a low-priority side stream) moves into the LDS its workgroups free as they
    // retire.  The longer one should go first.  Frames libzstd wrote: zb_seq — its time is its longest chain (a 128 KiB
    // block of 28 800 sequences: 4.6 ms) while most of its workgroups retire long before that (C5 leaves written by
    // libzstd: 79 -> 95 GB/s).  This library's frames (mostly literals): zb_lit (the other order costs C5 1 ms of 7).
    // Which case a context is in comes back with Status.kinds (zb_hdr met a block of >= 8192 sequences in the last calls).
    const bool seq_first = ctx->zb_seq_long;
    if (multi) side_fork(ctx, 2u);
    {
        KScope k(ctx, "zb_seq");
        zb_seq<<<std::min<uint32_t>((a.zb.block_cap + ZS_BLOCKS - 1) / ZS_BLOCKS + 4, 768u), 64, 0, (multi && !seq_first) ? ctx->side[1] : s>>>(a.zb);
    }
    {
        KScope k(ctx, "zb_lit");
        zb_lit<<<std::min<uint32_t>((a.zb.block_cap + ZL_BLOCKS - 1) / ZL_BLOCKS, 768u), 64, 0, (multi && seq_first) ? ctx->side[1] : s>>>(a.zb);
    }
    if (multi) side_join(ctx, 2u);
    launch_zb_exec(ctx, a, 0u);
}

#ifdef ZB_TL
void debug_lzx_timers(uint64_t* out8) { (void)hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_lzx_t), 8 * sizeof(uint64_t)); }
#endif

// LZ4 blocks of LZG_MIN compressed bytes and more of one queue, block-parallel (sb_lz4_giant.h)
static void launch_lzg(sb_ctx* ctx, const DecodeArgs& a, InflateJob* q, const uint32_t* nq, uint32_t cap) {
    if (!a.lzg.jobs || !a.lzg_chunks) return;
    // Twenty launches over grids sized for the longest page cost a call without such blocks ~0.1 ms (a plain 1 M-row page
    // is long enough to qualify), so a context launches them only once it has met an LZ4 block of megabytes: the first call
    // with long pages looks (one host round trip, once per context), later calls go by what the last interval met
    // (Status.kinds, read at every synchronize).
    if (ctx->lzg_state == 2 && !ctx->no_hints) return;
    hipStream_t s = ctx->stream;
    const LzgArgs g = a.lzg;
    const uint32_t NJ = std::max<uint32_t>(1u, a.lzg_jobs);
    const uint32_t ngroups = (a.lzg_chunks + LZG_GROUP - 1) / LZG_GROUP;
    {
        KScope k(ctx, "k_lzg_exits");
        k_lzg_pick<<<1, 256, 0, s>>>(g, q, nq, nullptr, nullptr, cap, NJ);
        if (ctx->lzg_state == 0) {
            uint32_t nj = 0;
            if (hipMemcpyAsync(&nj, g.njobs, 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) nj = 1;
            if (nj) ctx->lzg_state = 1;   // (none in this queue: the other queue looks for itself; sb_ctx_synchronize settles on 2)
            if (!nj) return;
        }
        k_lzg_clear<<<dim3(64, NJ), 256, 0, s>>>(g);
        k_lzg_exits<<<dim3(a.lzg_chunks, NJ), 256, 0, s>>>(g);
    }
    {
        KScope k(ctx, "k_lzg_chain");
        k_lzg_groups<<<dim3(ngroups, NJ), LZG_GT, 0, s>>>(g);
        k_lzg_chain<<<NJ, 64, 0, s>>>(g);
        k_lzg_cents<<<dim3(ngroups, NJ), 64, 0, s>>>(g);
    }
    {
        KScope k(ctx, "k_lzg_windows");
        k_lzg_windows<<<dim3(a.lzg_chunks, NJ), LB_T, 0, s>>>(g);
        k_lzg_lits<<<dim3(512, NJ), 256, 0, s>>>(g);
    }
    {
        KScope k(ctx, "k_lzg_jump");
        for (uint32_t r = 0; r < a.lzg_rounds; r++) k_lzg_jump<<<dim3(a.lzg_wins, NJ), 256, 0, s>>>(g);
    }
    KScope k(ctx, "k_lzg_pack");
    k_lzg_pack<<<dim3(std::min<uint32_t>(a.lzg_wins * 2 + 1, 4096u), NJ), 256, 0, s>>>(g);
}

void launch_decode(sb_ctx* ctx, const DecodeArgs& a, bool any_binary, bool any_prim, uint64_t* col_values_len) {
    hipStream_t s = ctx->stream;
    (void)hipMemsetAsync(a.job_counts, 0, 16 * sizeof(uint32_t), s);
    const bool qa = !(a.read_skips & RSKIP_QUEUE_A);
    {
        KScope k(ctx, K_PARSE);
        k_parse<<<(a.n_pages + WG - 1) / WG, WG, 0, s>>>(a);
        if (qa) launch_zstd_split(a, a.jobs_a, a.job_counts, a.job_counts + 8, a.job_cap_a, s);
        if (a.jobs_z) launch_zstd_split(a, a.jobs_z, a.job_counts + 10, a.job_counts + 11, a.job_cap_a, s);
    }
    launch_zb(ctx, a);
    if (qa) {
        KScope k(ctx, K_INFLATE_A);
        k_inflate<<<min(a.job_cap_a, INFLATE_POOL), 64, 0, s>>>(a.jobs_a, a.job_counts, a.status, a.zlit, a.zrec, rel_ctx(a), a.job_cap_a);
    }
    if (qa) {
        KScope k(ctx, "k_inflate_lz4");
        k_inflate_lz4<<<min(a.zs_segs ? a.job_cap_a : 2 * a.n_pages, LZ4_POOL), 64, 0, s>>>(a.jobs_a, a.job_counts, a.status, a.lz4_big_min, a.job_cap_a);
    }
    launch_lzg(ctx, a, a.jobs_a, a.job_counts, a.job_cap_a);
    if (a.lz4_big_min != 0xFFFFFFFFu) {
        KScope k(ctx, "k_inflate_lz4_big");
        k_inflate_lz4_big<<<min(a.zs_segs ? a.job_cap_a : 2 * a.n_pages, LZ4_BIG_POOL), LB_T, 0, s>>>(a.jobs_a, a.job_counts, a.status, a.lz4_big_min, a.job_cap_a, a.lzg_skipped);
    }
    {
        KScope k(ctx, K_PLAN);
        if (a.bp_guess) {   // (calls with few, long pages)
            (void)hipMemsetAsync(a.bp_guess, 0xFF, (size_t)a.n_pages * sizeof(uint32_t), s);
            k_bp_guess<<<dim3(BPG_PARTS, a.n_pages), WG, 0, s>>>(a);
        }
        k_plan<<<a.n_pages, WG, 0, s>>>(a);
    }
    if (any_binary && a.n_tiles >= BIN_DEFER_TILES) {   // long binary Dict pages: the tiles' value bytes by a workgroup per tile
        KScope k(ctx, "k_bin_tile_sums");
        k_bin_tile_sums<<<min(a.n_tiles, TILE_GRID), WG, 0, s>>>(a);
        k_bin_tile_scan<<<min((a.n_pages + WG / 64 - 1) / (WG / 64), 1024u), WG, 0, s>>>(a);
    }
    if (any_binary) {  // (without binary columns the host knows every values_len itself)
        KScope k(ctx, K_COLSCAN);
        k_colscan<<<a.n_cols, 64, 0, s>>>(a, col_values_len);
        launch_zstd_split(a, a.jobs_b, a.job_counts + 1, a.job_counts + 9, a.job_cap_b, s);
    }
    if (a.jobs_z) {   // queue Z: the frames the pipeline took are executed now that every page has its place; the rest by k_inflate
        launch_zb_exec(ctx, a, 1u);
        KScope k(ctx, "k_inflate(zstd values)");
        k_inflate<<<min(a.job_cap_a, INFLATE_POOL), 64, 0, s>>>(a.jobs_z, a.job_counts + 10, a.status, a.zlit, a.zrec, rel_ctx(a), a.job_cap_a);
    }
    if (any_binary) {
        KScope k(ctx, K_INFLATE_B);
        k_inflate<<<min(a.job_cap_b, INFLATE_POOL), 64, 0, s>>>(a.jobs_b, a.job_counts + 1, a.status, a.zlit, a.zrec, rel_ctx(a), a.job_cap_a);
        KScope k2(ctx, "k_inflate_lz4(values)");
        k_inflate_lz4<<<min(a.zs_segs ? a.job_cap_a : 2 * a.n_pages, LZ4_POOL), 64, 0, s>>>(a.jobs_b, a.job_counts + 1, a.status, a.lz4_big_min, a.job_cap_a);
    }
    if (any_binary) launch_lzg(ctx, a, a.jobs_b, a.job_counts + 1, a.job_cap_a);
    if (any_binary && a.lz4_big_min != 0xFFFFFFFFu) {
        KScope k(ctx, "k_inflate_lz4_big(values)");
        k_inflate_lz4_big<<<min(a.zs_segs ? a.job_cap_a : 2 * a.n_pages, LZ4_BIG_POOL), LB_T, 0, s>>>(a.jobs_b, a.job_counts + 1, a.status, a.lz4_big_min, a.job_cap_a, a.lzg_skipped);
    }
    // the three expand kernels work on disjoint pages (page-level RLE, tiles of primitives, tiles of binary columns): side
    // by side on streams of their own when the call has both kinds of columns (a mixed schema), joined before the call ends
    const bool multi = any_prim && any_binary && a.n_tiles && !ctx->profile && side_streams(ctx);
    hipStream_t s1 = multi ? ctx->side[0] : s, s2 = multi ? ctx->side[1] : s;
    if (multi) side_fork(ctx, 3u);
    if (any_prim) {
        KScope k(ctx, K_EXPAND_RLE);
        if (a.rle_parts > 1) k_rle_sums<<<dim3(a.n_pages, a.rle_parts), WG, 0, s>>>(a);
        k_expand_rle<<<dim3(a.n_pages, std::max<uint32_t>(1u, a.rle_parts)), WG, 0, s>>>(a);
    }
    if (a.n_tiles && any_prim && !(a.read_skips & RSKIP_TILES)) {
        KScope k(ctx, K_EXPAND);
        k_expand<<<min(a.n_tiles, TILE_GRID), WG, 0, s1>>>(a);
    }
    if (a.n_tiles && any_binary) {
        KScope k(ctx, K_EXPAND_BIN);
        k_expand_binary<<<min(a.n_tiles, TILE_GRID), WG, 0, s2>>>(a);
    }
    if (multi) side_join(ctx, 3u);
}

void launch_parse_sizes(sb_ctx* ctx, const DecodeArgs& a, uint64_t* col_values_len) {
    hipStream_t s = ctx->stream;
    (void)hipMemsetAsync(a.job_counts, 0, 16 * sizeof(uint32_t), s);
    k_parse<<<(a.n_pages + WG - 1) / WG, WG, 0, s>>>(a);
    launch_zstd_split(a, a.jobs_a, a.job_counts, a.job_counts + 8, a.job_cap_a, s);
    launch_zb(ctx, a);
    k_inflate<<<min(a.job_cap_a, INFLATE_POOL), 64, 0, s>>>(a.jobs_a, a.job_counts, a.status, a.zlit, a.zrec, rel_ctx(a), a.job_cap_a);
    k_inflate_lz4<<<min(a.zs_segs ? a.job_cap_a : 2 * a.n_pages, LZ4_POOL), 64, 0, s>>>(a.jobs_a, a.job_counts, a.status, 0xFFFFFFFFu, a.job_cap_a);
    k_plan<<<a.n_pages, WG, 0, s>>>(a);
    k_colscan<<<a.n_cols, 64, 0, s>>>(a, col_values_len);
}

}  // namespace sb
