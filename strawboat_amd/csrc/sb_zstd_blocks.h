// strawboat-hip: Zstandard frames decoded BLOCK-PARALLEL (RFC 8878), codec id 2.
//
// Replaces zstd::bulk::decompress_to_buffer (libzstd) at the reference call site src/compression/basic.rs:93-97 for the
// frames libzstd itself writes: one frame per page buffer, a sequence of <= 128 KiB blocks (libzstd >= 1.5 also splits
// blocks), Huffman literals in four streams (often "treeless": the previous block's tree), FSE sequences with transmitted
// or repeated tables.  The one-wave decoder (sb_zstd.h) walks such a frame front to back: four lanes on the literal
// streams, one serial FSE chain, 60 ms for a 512 KiB page.  The entropy stages of different blocks are independent of each
// other — what a block inherits (a Huffman tree, FSE tables, the three repeat offsets) is either re-built from the block
// that transmitted it or carried SYMBOLICALLY — so this pipeline runs them for all blocks of all frames of a call at once:
//
//   zb_scan   thread / frame   frame header, block walk -> ZbBlock descriptors (literals header, sequence count, which
//                              earlier block defines the tree / the three tables in use), literal + record pool areas
//   zb_hdr    thread / block   the block's own FSE table descriptions located (their lengths come from parsing them)
//   zb_lit    wave / 16 blocks Huffman literals, LANE PER STREAM (64 streams): tables in LDS, every lane keeps a private
//                              128-byte window of its stream in LDS (chunks requested one round ahead), 32 symbols per
//                              round with a branch-free refill, 32 output bytes per lane and round -> literal pool
//   zb_seq    wave / 4 blocks  FSE sequence streams, LANE PER BLOCK: packed tables in LDS (one read per state and
//                              sequence), the bit stream through an LDS window, repeat offsets that reach back before
//                              the block stay symbolic (which of the three entry values, minus how much) -> 8-byte
//                              records (literal length, match length, offset) in the record pool
//   zb_exec   wave / frame     blocks in order: repeat offsets resolved, records executed through the LDS output ring
//                              (sb_lz4.h LzSeqExec), raw / RLE blocks copied
//
// Anything the pipeline does not take (several frames in one buffer, pools exhausted, a stream whose headers or entropy
// stages are malformed) stays with — or is handed back to — the one-wave decoder, which owns the error codes of those
// cases: a frame is "punted" by restoring its queue entry before k_inflate runs.  The one exception: faults that only
// show while the records are EXECUTED (an offset beyond the output so far, an overrun, a length mismatch) are raised by
// zb_exec / zb_exec_wg themselves as SB_ERR_EXTERNAL with detail 120 + n — part of dst may be written by then, as with the
// one-wave decoder, but the detail code of the same damaged frame can differ between SB_ZSTD_BLOCKS=0 and 1 (both refuse
// it; tests/test_gpu_zstd_blocks.py compares accept / refuse and the decoded bytes with the oracle, not the detail).
#pragma once
#include "sb_zstd.h"
#include "sb_lz4_big.h"   // wave_scan_max_dpp; zb_exec_wg copies windows the way lz4_inflate_block_wg does

namespace sb {

constexpr uint32_t CODEC_ZB = 0xFE;   // a Zstd queue entry taken by the block pipeline (k_inflate skips it)
constexpr uint32_t ZB_NONE = 0xFFFFFFFFu;

// symbolic repeat offsets: bit 27 | which of the block's three entry values << 20 | how much to subtract (20 bits)
constexpr uint32_t ZB_SYM = 1u << 27;
__device__ __forceinline__ uint32_t zb_sym(uint32_t id) { return ZB_SYM | (id << 20); }
__device__ __forceinline__ uint32_t zb_resolve(uint32_t v, uint32_t e0, uint32_t e1, uint32_t e2) {
    if (!(v & ZB_SYM)) return v;
    const uint32_t id = (v >> 20) & 3, dl = v & 0xFFFFFu;
    const uint32_t e = id == 0 ? e0 : id == 1 ? e1 : e2;
    return e > dl ? e - dl : 0u;   // 0: invalid (the caller reports it)
}
// v in terms of the state (a0, a1, a2), itself in terms of an earlier state: constants stay, "entry i minus d" becomes a_i minus d
__device__ __forceinline__ uint32_t zb_subst(uint32_t v, uint32_t a0, uint32_t a1, uint32_t a2) {
    if (!(v & ZB_SYM)) return v;
    const uint32_t id = (v >> 20) & 3, dl = v & 0xFFFFFu;
    const uint32_t a = id == 0 ? a0 : id == 1 ? a1 : a2;
    if (a & ZB_SYM) return a + dl;          // (deltas add up to at most the number of sequences of a block: < 2^20)
    return a > dl ? a - dl : 0u;            // 0: invalid, and stays so (0 minus anything is 0 here)
}

// where a queue entry's output goes: absolute, or (JOB_REL) relative to the page's value base, known after k_colscan;
// null: the page does not take part (its values do not fit the caller's buffer)
struct RelCtx {
    const ColDesc* cols;
    const PageTask* tasks;
    const PageDesc* descs;
};
__device__ __forceinline__ uint8_t* job_dst(const RelCtx& rc, uint8_t* dst, uint32_t page, bool rel) {
    if (!rel) return dst;
    const PageDesc& d = rc.descs[page];
    if (!d.ok) return nullptr;
    return rc.cols[rc.tasks[page].col].values + d.val_base + (uintptr_t)dst;
}
// literals section header at bs[0, n): false when it does not fit
__device__ inline bool zb_lit_header(const uint8_t* bs, uint32_t n, uint32_t* ltype, uint32_t* streams, uint32_t* regen, uint32_t* csize,
                                     uint32_t* hdr) {
    if (n < 1) return false;
    const uint8_t b0 = ldu8(bs);
    const uint32_t lt = b0 & 3, sf = (b0 >> 2) & 3;
    uint32_t bp = 1, rg = 0, cs = 0, st = 1;
    if (lt == 0 || lt == 1) {
        if (sf == 0 || sf == 2) {
            rg = b0 >> 3;
        } else if (sf == 1) {
            if (n < 2) return false;
            rg = (b0 >> 4) | ((uint32_t)ldu8(bs + 1) << 4);
            bp = 2;
        } else {
            if (n < 3) return false;
            rg = (b0 >> 4) | ((uint32_t)ldu8(bs + 1) << 4) | ((uint32_t)ldu8(bs + 2) << 12);
            bp = 3;
        }
        cs = lt == 0 ? rg : 1u;
    } else {
        if (sf == 0 || sf == 1) {
            if (n < 3) return false;
            const uint32_t v = (b0 >> 4) | ((uint32_t)ldu8(bs + 1) << 4) | ((uint32_t)ldu8(bs + 2) << 12);
            bp = 3;
            rg = v & 0x3FF;
            cs = v >> 10;
            st = sf == 0 ? 1 : 4;
        } else if (sf == 2) {
            if (n < 4) return false;
            const uint32_t v = (b0 >> 4) | ((uint32_t)ldu8(bs + 1) << 4) | ((uint32_t)ldu8(bs + 2) << 12) | ((uint32_t)ldu8(bs + 3) << 20);
            bp = 4;
            rg = v & 0x3FFF;
            cs = v >> 14;
            st = 4;
        } else {
            if (n < 5) return false;
            const uint64_t v = (b0 >> 4) | ((uint64_t)ldu8(bs + 1) << 4) | ((uint64_t)ldu8(bs + 2) << 12) | ((uint64_t)ldu8(bs + 3) << 20) |
                               ((uint64_t)ldu8(bs + 4) << 28);
            bp = 5;
            rg = (uint32_t)(v & 0x3FFFF);
            cs = (uint32_t)(v >> 18);
            st = 4;
        }
    }
    *ltype = lt;
    *streams = st;
    *regen = rg;
    *csize = cs;
    *hdr = bp;
    return true;
}

// ---------------------------------------------------------------------------------------------------- zb_scan
// One thread per queue entry.  An entry is taken when it is exactly one frame (no dictionary, content size — if the header
// has one — equal to the entry's output) whose blocks are well-formed as far as their headers go, and the pools have room.
__global__ void __launch_bounds__(WG) zb_scan(InflateJob* q, const uint32_t* count, ZbPools zp, uint32_t queue, uint32_t cap) {
    const uint32_t njobs = min(*count, cap);
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < njobs; j += gridDim.x * blockDim.x) {
        const InflateJob job = q[j];
        if ((job.codec & ~JOB_REL) != SB_CODEC_ZSTD || job.csize < zp.min_csize) continue;
        const uint8_t* src = job.src;
        const uint32_t n = job.csize;
        if (n < 9 || ldu32(src) != 0xFD2FB528u) continue;
        uint32_t ip = 4;
        const uint8_t fhd = ldu8(src + ip++);
        const uint32_t fcs_flag = fhd >> 6, single = (fhd >> 5) & 1, checksum = (fhd >> 2) & 1, did = fhd & 3;
        if ((fhd & 0x08) || did) continue;
        if (!single) ip += 1;
        const uint32_t fcs_bytes = fcs_flag == 0 ? single : (1u << fcs_flag);
        if (n - ip < fcs_bytes + 3) continue;
        if (fcs_bytes) {
            uint64_t fcs = 0;
            for (uint32_t i = 0; i < fcs_bytes; i++) fcs |= (uint64_t)ldu8(src + ip + i) << (8 * i);
            if (fcs_bytes == 2) fcs += 256;
            if (fcs != job.out_len) continue;
            ip += fcs_bytes;
        }
        const uint32_t ip0 = ip;
        // ---- first walk: count, validate the headers
        uint32_t nb = 0;
        uint64_t lit_need = 0, rec_need = 0, known_out = 0;
        bool ok = true;
        for (;;) {
            if (n - ip < 3) { ok = false; break; }
            const uint32_t bh = (uint32_t)ldu8(src + ip) | ((uint32_t)ldu8(src + ip + 1) << 8) | ((uint32_t)ldu8(src + ip + 2) << 16);
            ip += 3;
            const uint32_t btype = (bh >> 1) & 3, bsize = bh >> 3;
            if (btype == 3) { ok = false; break; }
            if (btype == 0 || btype == 1) {
                const uint32_t body = btype == 1 ? 1u : bsize;
                if (n - ip < body || bsize > 128 * 1024) { ok = false; break; }
                ip += body;
                known_out += bsize;
            } else {
                if (bsize > 128 * 1024 || n - ip < bsize || bsize < 2) { ok = false; break; }
                uint32_t lt, st, rg, cs, hd;
                if (!zb_lit_header(src + ip, bsize, &lt, &st, &rg, &cs, &hd)) { ok = false; break; }
                if (rg > 128 * 1024 || bsize - hd < cs) { ok = false; break; }
                uint32_t bp = hd + cs;
                if (bsize - bp < 1) { ok = false; break; }
                const uint8_t s0 = ldu8(src + ip + bp++);
                uint32_t nseq;
                if (s0 < 128) {
                    nseq = s0;
                } else if (s0 < 255) {
                    if (bsize - bp < 1) { ok = false; break; }
                    nseq = ((uint32_t)(s0 - 128) << 8) + ldu8(src + ip + bp);
                    bp += 1;
                } else {
                    if (bsize - bp < 2) { ok = false; break; }
                    nseq = (uint32_t)ldu8(src + ip + bp) + ((uint32_t)ldu8(src + ip + bp + 1) << 8) + 0x7F00;
                    bp += 2;
                }
                if (nseq) {
                    if (bsize - bp < 2) { ok = false; break; }   // modes byte + at least one byte of bit stream
                    if (ldu8(src + ip + bp) & 3) { ok = false; break; }
                } else if (bp != bsize) {
                    ok = false;
                    break;
                }
                if (lt != 0) lit_need += (rg + 15) & ~15u;
                rec_need += nseq;
                known_out += rg;
                ip += bsize;
            }
            nb++;
            if (bh & 1) break;
            if (nb >= 0x100000) { ok = false; break; }
        }
        if (!ok) continue;
        if (checksum) {
            if (n - ip < 4) continue;
            ip += 4;
        }
        if (ip != n || known_out > job.out_len) continue;   // several frames / trailing bytes: the one-wave path walks them
        // ---- pool areas
        // The reservation is all-or-nothing for the consumers: zb_hdr / zb_lit / zb_seq walk EVERY slot below counters[0], and
        // the block pool is neither cleared between calls nor by ensure() — a slot this thread reserved and then gave up
        // (another pool full) would still hold the previous call's descriptor, naming a live frame index of THIS call.  Every
        // path that gives up after the block atomicAdd therefore stores inert descriptors (frame = ZB_NONE: no consumer
        // looks further) into the part of its range that lies inside the pool.
        const uint32_t b0 = atomicAdd(&zp.counters[0], nb);
        auto give_up = [&]() {
            ZbBlock v;
            __builtin_memset(&v, 0, sizeof v);
            v.frame = ZB_NONE;
            v.btype = 3;
            v.huf_def = ZB_NONE;
            v.def[0] = v.def[1] = v.def[2] = ZB_NONE;
            for (uint64_t k = b0; k < (uint64_t)b0 + nb && k < zp.block_cap; k++) zp.blocks[k] = v;
            if (zp.stats) atomicAdd(&zp.stats[1], 1ull);   // counted with the frames handed back (sb_ctx_zstd_block_stats)
        };
        if ((uint64_t)b0 + nb > zp.block_cap) { give_up(); continue; }
        const uint64_t l0 = atomicAdd((unsigned long long*)&zp.counters[4], (unsigned long long)lit_need);
        if (l0 + lit_need > zp.lit_cap) { give_up(); continue; }
        const uint64_t r0 = atomicAdd((unsigned long long*)&zp.counters[6], (unsigned long long)rec_need);
        if (r0 + rec_need > zp.rec_cap) { give_up(); continue; }
        const uint32_t f = atomicAdd(&zp.counters[1], 1u);
        if (f >= zp.frame_cap) { give_up(); continue; }
        // ---- second walk: the descriptors
        ip = ip0;
        uint32_t huf_def = ZB_NONE, d_ll = ZB_NONE, d_of = ZB_NONE, d_ml = ZB_NONE;
        bool have_ll = false, have_of = false, have_ml = false;
        uint64_t lpos = l0, rpos = r0;
        bool bad = false;
        for (uint32_t k = 0; k < nb; k++) {
            const uint32_t bh = (uint32_t)ldu8(src + ip) | ((uint32_t)ldu8(src + ip + 1) << 8) | ((uint32_t)ldu8(src + ip + 2) << 16);
            ip += 3;
            const uint32_t btype = (bh >> 1) & 3, bsize = bh >> 3;
            ZbBlock b;
            __builtin_memset(&b, 0, sizeof b);
            b.src = src + ip;
            b.frame = f;
            b.btype = btype;
            b.huf_def = ZB_NONE;
            b.def[0] = b.def[1] = b.def[2] = ZB_NONE;
            if (btype != 2) {
                b.bsize = btype == 1 ? 1u : bsize;
                b.out_size = bsize;
                ip += b.bsize;
            } else {
                b.bsize = bsize;
                uint32_t hd;
                zb_lit_header(src + ip, bsize, &b.ltype, &b.lstreams, &b.regen, &b.lcsize, &hd);
                b.lpay = hd;
                if (b.ltype == 2) huf_def = b0 + k;
                if (b.ltype == 3 && huf_def == ZB_NONE) bad = true;
                if (b.ltype >= 2) b.huf_def = huf_def;
                if (b.ltype != 0) {
                    b.lit_pos = lpos;
                    lpos += (b.regen + 15) & ~15u;
                }
                uint32_t bp = hd + b.lcsize;
                const uint8_t s0 = ldu8(src + ip + bp++);
                if (s0 < 128) {
                    b.nseq = s0;
                } else if (s0 < 255) {
                    b.nseq = ((uint32_t)(s0 - 128) << 8) + ldu8(src + ip + bp);
                    bp += 1;
                } else {
                    b.nseq = (uint32_t)ldu8(src + ip + bp) + ((uint32_t)ldu8(src + ip + bp + 1) << 8) + 0x7F00;
                    bp += 2;
                }
                b.out_size = b.regen;
                b.rec_pos = rpos;
                rpos += b.nseq;
                if (b.nseq) {
                    b.modes = ldu8(src + ip + bp++);
                    b.seq_off = bp;
                    const uint32_t m_ll = (b.modes >> 6) & 3, m_of = (b.modes >> 4) & 3, m_ml = (b.modes >> 2) & 3;
                    // (a table stays in use until a block with sequences replaces it: mode 0 selects the predefined one)
                    if (m_ll != 3) { d_ll = m_ll ? b0 + k : ZB_NONE; have_ll = true; }
                    if (m_of != 3) { d_of = m_of ? b0 + k : ZB_NONE; have_of = true; }
                    if (m_ml != 3) { d_ml = m_ml ? b0 + k : ZB_NONE; have_ml = true; }
                    if (!have_ll || !have_of || !have_ml) bad = true;   // "repeat" with nothing to repeat
                    b.def[0] = d_ll;
                    b.def[1] = d_of;
                    b.def[2] = d_ml;
                }
                ip += bsize;
            }
            zp.blocks[b0 + k] = b;
        }
        ZbFrame fr;
        fr.dst = job.dst;
        fr.out_len = job.out_len;
        fr.first = b0;
        fr.nblocks = nb;
        fr.job = j;
        fr.page = job.page;
        fr.punt = bad ? 1u : 0u;
        fr.avail = n;
        fr.base = src;
        fr.queue = queue;
        fr.rel = (job.codec & JOB_REL) ? 1u : 0u;
        // many short sequences (less than 24 output bytes each): the workgroup executor, which pays per byte, not per sequence
        fr.wg = (zp.wg_exec && rec_need >= 2048 && (uint64_t)job.out_len < 24 * rec_need) ? 1u : 0u;
        zp.frames[f] = fr;
        q[j].codec = CODEC_ZB | (job.codec & JOB_REL);   // (zb_exec restores it for a punted frame)
    }
}

// length of an FSE table description (forward bit stream) at src[0, n), 0 when malformed
__device__ inline uint32_t zb_fse_desc_len(const uint8_t* src, uint32_t n, int max_sym, int max_log) {
    uint64_t bitpos = 0;
    auto peek = [&](int nb) -> uint32_t {
        uint64_t v = 0;
        const uint64_t byte = bitpos >> 3;
        for (int i = 0; i < 4 && byte + i < n; i++) v |= (uint64_t)ldu8(src + byte + i) << (8 * i);
        v >>= (bitpos & 7);
        return (uint32_t)(v & ((1ull << nb) - 1));
    };
    const int log = (int)peek(4) + 5;
    bitpos += 4;
    if (log > max_log) return 0;
    int remaining = (1 << log) + 1, threshold = 1 << log, nbits = log + 1, sym = 0;
    bool prev0 = false;
    while (remaining > 1 && sym <= max_sym) {
        if ((bitpos >> 3) >= n) return 0;
        if (prev0) {
            for (;;) {
                const uint32_t r = peek(2);
                bitpos += 2;
                sym += (int)r;
                if (r != 3) break;
                if ((bitpos >> 3) >= n) return 0;
            }
            prev0 = false;
            if (sym > max_sym) return 0;
            continue;
        }
        const int mx = (2 * threshold - 1) - remaining;
        int cnt;
        const uint32_t v = peek(nbits);
        if ((int)(v & (uint32_t)(threshold - 1)) < mx) {
            cnt = (int)(v & (uint32_t)(threshold - 1));
            bitpos += nbits - 1;
        } else {
            cnt = (int)(v & (uint32_t)(2 * threshold - 1));
            if (cnt >= threshold) cnt -= mx;
            bitpos += nbits;
        }
        cnt--;
        remaining -= cnt < 0 ? -cnt : cnt;
        sym++;
        prev0 = cnt == 0;
        while (remaining < threshold) {
            nbits--;
            threshold >>= 1;
        }
    }
    if (remaining != 1) return 0;
    const uint32_t used = (uint32_t)((bitpos + 7) >> 3);
    return used > n ? 0u : used;
}

// ---------------------------------------------------------------------------------------------------- zb_hdr
// One thread per block with sequences: where its own table descriptions and its bit stream start.
__global__ void __launch_bounds__(WG) zb_hdr(ZbPools zp) {
    const uint32_t nblocks = min(zp.counters[0], zp.block_cap);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nblocks; i += gridDim.x * blockDim.x) {
        ZbBlock& b = zp.blocks[i];
        if (b.btype != 2 || !b.nseq) continue;
        if (b.frame >= zp.frame_cap || zp.frames[b.frame].punt) continue;
        uint32_t pos = b.seq_off;
        bool ok = true;
        const int max_sym[3] = {35, 31, 52}, max_log[3] = {9, 8, 9};
        for (int t = 0; t < 3 && ok; t++) {
            const uint32_t mode = (b.modes >> (6 - 2 * t)) & 3;
            if (mode == 1) {
                if (pos >= b.bsize || ldu8(b.src + pos) > (uint32_t)max_sym[t]) ok = false;
                b.desc[t] = pos;
                pos += 1;
            } else if (mode == 2) {
                if (pos >= b.bsize) { ok = false; break; }
                const uint32_t len = zb_fse_desc_len(b.src + pos, b.bsize - pos, max_sym[t], max_log[t]);
                if (!len) ok = false;
                b.desc[t] = pos;
                pos += len;
            }
        }
        if (ok && pos >= b.bsize) ok = false;
        b.bits_off = pos;
        if (!ok) {
            zp.frames[b.frame].punt = 1;
        } else {   // size classes: zb_seq runs four blocks per wave in lockstep, so they should be about as long as each other
            const uint32_t cls = b.nseq >= 8192 ? 0u : b.nseq >= 2048 ? 1u : b.nseq >= 512 ? 2u : 3u;
            const uint32_t at = atomicAdd(&zp.counters[8 + cls], 1u);
            if (at < zp.block_cap) zp.lists[(size_t)cls * zp.block_cap + at] = i;
            if (cls == 0 && at == 0 && zp.kinds) atomicOr(zp.kinds, KIND_ZSEQ_LONG);
        }
    }
}

// ---------------------------------------------------------------------------------------------------- zb_lit
constexpr uint32_t ZL_BLOCKS = 16;               // blocks per wave and group: 64 streams
constexpr uint32_t ZL_TAB = 16 * 1024;           // table entries (u16) of a pass: eight 11-bit tables, sixteen 10-bit ones
constexpr uint32_t ZL_DESC = 160;                // bytes of a tree description staged in LDS (<= 1 + 127 used)
struct ZlParse {   // tree descriptions -> weights (done before the first stream is read: shares its bytes with the rings)
    __attribute__((aligned(16))) uint8_t desc[ZL_BLOCKS][ZL_DESC];
    ZFse fse[ZL_BLOCKS][64];
    int16_t norm[ZL_BLOCKS][16];
    uint16_t next[ZL_BLOCKS][16];
};
struct ZlBuild {   // weights -> sorted symbol lists (per table slot; kept for the passes)
    uint8_t wts[ZL_BLOCKS][256];
    uint8_t sorted[ZL_BLOCKS][256];
};
struct ZlLds {
    uint16_t tab[ZL_TAB];
    ZlBuild b;
    union {
        ZlParse p;
        uint32_t ring[32][64];       // stream word w of lane l: ring[w & 31][l] (a lane only ever touches its own bank)
    };
    uint32_t nw[ZL_BLOCKS];          // weights read per slot (0: no table)
    uint32_t bits[ZL_BLOCKS];        // code bits per slot (0: no table)
    uint32_t used[ZL_BLOCKS];        // bytes of the tree description
    uint32_t toff[ZL_BLOCKS];        // first entry of the slot's table in `tab`
    uint32_t cw[ZL_BLOCKS][12], lastw[ZL_BLOCKS];
};
static_assert(sizeof(ZlParse) <= 32 * 64 * 4 && sizeof(ZlLds) <= 53 * 1024, "zb_lit: three waves per CU");

// weights of a tree description d[0, n) (in LDS) -> wts[0, nw); *used = its bytes.  Returns nw (0: malformed)
__device__ inline uint32_t zb_huf_weights(const uint8_t* d, uint32_t n, uint8_t* wts, ZFse* t, int16_t* norm, uint16_t* next, uint32_t* used) {
    if (n < 1) return 0;
    const uint8_t hb = d[0];
    uint32_t nw;
    if (hb >= 128) {
        nw = hb - 127u;
        const uint32_t bytes = (nw + 1) / 2;
        if (n < 1 + bytes) return 0;
        for (uint32_t i = 0; i < nw; i++) {
            const uint8_t b = d[1 + i / 2];
            wts[i] = (i & 1) ? (b & 15) : (b >> 4);
        }
        *used = 1 + bytes;
        return nw;
    }
    const uint32_t clen = hb;
    if (n < 1 + clen || clen < 2) return 0;
    int nsym, log;
    const uint32_t hsz = z_fse_header(d + 1, clen, 12, 6, norm, &nsym, &log);
    if (!hsz || hsz >= clen) return 0;
    if (!z_fse_build2(t, next, norm, nsym, log)) return 0;
    const uint8_t* bs = d + 1 + hsz;
    const uint32_t bn = clen - hsz;
    if (bs[bn - 1] == 0) return 0;
    int64_t bitpos = (int64_t)(bn - 1) * 8 + (31 - __clz((int)bs[bn - 1]));
    uint32_t s1 = z_peek(bs, bitpos, log);
    bitpos -= log;
    uint32_t s2 = z_peek(bs, bitpos, log);
    bitpos -= log;
    nw = 0;
    for (;;) {
        if (nw >= 254) return 0;
        wts[nw++] = t[s1].symbol;
        if (bitpos < t[s1].nbits) {
            wts[nw++] = t[s2].symbol;
            break;
        }
        {
            const uint32_t nb = t[s1].nbits;
            s1 = t[s1].base + z_peek(bs, bitpos, (int)nb);
            bitpos -= nb;
        }
        wts[nw++] = t[s2].symbol;
        if (bitpos < t[s2].nbits) {
            wts[nw++] = t[s1].symbol;
            break;
        }
        {
            const uint32_t nb = t[s2].nbits;
            s2 = t[s2].base + z_peek(bs, bitpos, (int)nb);
            bitpos -= nb;
        }
    }
    *used = 1 + clen;
    return nw;
}

// Slot s: class sizes, code bits, the symbols in (weight, symbol) order — by the whole wave (cf. z_wave_huf_fill)
__device__ inline void zb_huf_classes(ZlLds& L, uint32_t s) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t nw = L.nw[s];   // (uniform)
    if (!nw) {
        if (lane == 0) L.bits[s] = 0;
        return;
    }
    uint32_t w[4];
    bool bad = false;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t sy = lane + 64 * j;
        w[j] = sy < nw ? (uint32_t)L.b.wts[s][sy] : 0u;
        bad = bad || w[j] > 11;
    }
    uint32_t cw[12], total = 0;
#pragma unroll
    for (uint32_t wv = 1; wv < 12; wv++) {
        uint32_t c = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) c += (uint32_t)__popcll(__ballot(w[j] == wv));
        cw[wv] = c;
        total += c << (wv - 1);
    }
    uint32_t max_bits = 0, lw = 0;
    bool okf = !__ballot(bad) && total != 0;
    if (okf) {
        max_bits = 32u - (uint32_t)__clz((int)total);
        const uint32_t left = (1u << max_bits) - total;
        okf = left != 0 && !(left & (left - 1)) && max_bits <= 11;
        lw = okf ? 32u - (uint32_t)__clz((int)left) : 0u;   // weight of the last symbol (implied)
    }
    if (!okf) {
        if (lane == 0) L.bits[s] = 0;
        return;
    }
#pragma unroll
    for (int j = 0; j < 4; j++)
        if (lane + 64 * j == nw) w[j] = lw;
    uint32_t nb = 0;
#pragma unroll
    for (uint32_t wv = 1; wv < 12; wv++) {
        const uint32_t c = cw[wv] + (wv == lw ? 1u : 0u);
        if (lane == 0) L.cw[s][wv] = c;
        uint32_t base = nb;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint64_t m = __ballot(w[j] == wv);
            if (w[j] == wv) L.b.sorted[s][base + lane_rank(m)] = (uint8_t)(lane + 64 * j);
            base += (uint32_t)__popcll(m);
        }
        nb += c;
    }
    if (lane == 0) {
        L.bits[s] = max_bits;
        L.lastw[s] = lw;
    }
}
// Slot s: its table (2^bits entries: symbol | code length << 8) at L.tab + L.toff[s]
__device__ inline void zb_huf_fill(ZlLds& L, uint32_t s) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t max_bits = L.bits[s];
    if (!max_bits) return;
    uint32_t start[12], before[12], cw[12];
    {
        uint32_t code = 0, nb = 0;
#pragma unroll
        for (uint32_t wv = 1; wv < 12; wv++) {
            cw[wv] = L.cw[s][wv];
            start[wv] = code;
            before[wv] = nb;
            code += cw[wv] << (wv - 1);
            nb += cw[wv];
        }
    }
    uint16_t* tab = L.tab + L.toff[s];
    const uint8_t* sl = L.b.sorted[s];
    for (uint32_t c = lane; c < (1u << max_bits); c += 64) {
        uint32_t cls = 1, st = 0, bf = 0;
#pragma unroll
        for (uint32_t wv = 1; wv < 12; wv++) {
            if (cw[wv] && c >= start[wv]) {
                cls = wv;
                st = start[wv];
                bf = before[wv];
            }
        }
        tab[c] = (uint16_t)((uint32_t)sl[bf + ((c - st) >> (cls - 1))] | ((max_bits + 1 - cls) << 8));
    }
}

// One Huffman stream per lane: sb_[0, sn) -> outn bytes at dst.  tab: the lane's table (2^mb entries) in LDS.  avail: bytes
// that may be read from sb_ on (>= sn).  The stream is read from its end downwards; the lane keeps stream words
// [lo / 4, lo / 4 + 32) in its column of `ring`, requests the 16-byte chunks the NEXT round may need at the start of a
// round and files them after it.  A round is 32 symbols: 16 x (refill-if-low, symbol, symbol) without a branch.
// Returns true when the stream decoded exactly (all bits consumed, nothing beyond).  All 64 lanes call it.
__device__ inline bool zb_huf_stream(ZlLds& L, const uint16_t* tab, uint32_t mb, const uint8_t* sb_, uint32_t sn, uint32_t avail, uint8_t* dst,
                                     uint32_t outn, bool act) {
    const uint32_t lane = threadIdx.x & 63;
    int32_t left = 0;
    if (act) {
        const uint8_t lastb = sn ? ldu8(sb_ + sn - 1) : (uint8_t)0;
        if (lastb == 0) act = false;
        else left = (int32_t)sn * 8 - (int32_t)(8 - (31 - (uint32_t)__clz((int)lastb)));   // the end mark and the bits above it are gone
    }
    const bool had = act;
    auto chunk = [&](int32_t b) -> u32x4 {   // stream bytes [b, b + 16), b >= 0; bytes beyond `avail` read as 0
        if ((uint32_t)b + 16 <= avail) return ldu128(sb_ + b);
        uint32_t w4[4] = {0, 0, 0, 0};
        for (uint32_t k = 0; k < 16; k++)
            if ((uint32_t)b + k < avail) w4[k >> 2] |= (uint32_t)ldu8(sb_ + b + k) << (8 * (k & 3));
        return u32x4{w4[0], w4[1], w4[2], w4[3]};
    };
    auto file = [&](int32_t b, u32x4 v) {   // chunk at byte b into the ring
        const uint32_t w0 = (uint32_t)b >> 2;
        L.ring[w0 & 31][lane] = v.x;
        L.ring[(w0 + 1) & 31][lane] = v.y;
        L.ring[(w0 + 2) & 31][lane] = v.z;
        L.ring[(w0 + 3) & 31][lane] = v.w;
    };
    const int32_t hi0 = (left + 7) >> 3;            // bytes that hold unread bits
    int32_t lo_loaded = 0;
    if (act) {
        const int32_t ctop = (hi0 - 1) >> 4;
        const int32_t c0 = ctop >= 7 ? ctop - 7 : 0;
        u32x4 v[8];
#pragma unroll
        for (int32_t k = 0; k < 8; k++)
            if (c0 + k <= ctop) v[k] = chunk(16 * (c0 + k));
#pragma unroll
        for (int32_t k = 0; k < 8; k++)
            if (c0 + k <= ctop) file(16 * (c0 + k), v[k]);
        lo_loaded = 16 * c0;
    }
    auto word = [&](int32_t w) -> uint32_t {   // (read unconditionally: a branch around the read would expose its latency)
        const uint32_t v = L.ring[(uint32_t)w & 31][lane];
        return w >= 0 ? v : 0u;
    };
    uint64_t bb = 0;     // the next bits of the stream, top-aligned
    uint32_t cnt = 0;    // how many of them are valid
    int32_t wi = -1;     // index of the word in nx
    uint32_t nx = 0;
    if (act) {
        const uint32_t kb = (uint32_t)((hi0 - 1) & 3) + 1;   // bytes of the top word that belong to the stream
        const int32_t wt = (hi0 - 1) >> 2;
        const uint32_t u0 = (uint32_t)(8 * hi0 - left);      // unused bits of the top byte (0 .. 7)
        bb = ((uint64_t)word(wt) << (32 + 8 * (4 - kb))) << u0;
        cnt = 8 * kb - u0;
        wi = wt - 1;
        nx = word(wi);
    }
    const uint32_t sh = 32 - mb;
    uint32_t done = 0;
#define ZB_REFILL()                                                       \
    do {                                                                  \
        const bool need_ = cnt <= 32;                                     \
        const uint64_t add_ = (uint64_t)nx << ((32 - cnt) & 31);          \
        bb |= need_ ? add_ : 0ull;                                        \
        cnt += need_ ? 32u : 0u;                                          \
        wi -= need_ ? 1 : 0;                                              \
        nx = word(wi);                                                    \
    } while (0)
#define ZB_SYMBOL(acc, shl)                                               \
    do {                                                                  \
        const uint32_t e_ = tab[(uint32_t)(bb >> 32) >> sh];              \
        const uint32_t len_ = e_ >> 8;                                    \
        bb <<= len_;                                                      \
        cnt -= len_;                                                      \
        left -= (int32_t)len_;                                            \
        acc |= (e_ & 255u) << (shl);                                      \
    } while (0)
    while (__ballot(act && done < outn)) {
        const bool run = act && done < outn;
        const bool full = run && outn - done >= 32;
        // ---- the chunks the next round may read: everything from word wi - 28 up (this round reads down to wi - 12)
        int32_t lo_new = lo_loaded;
        if (run) {
            const int32_t need = (4 * (wi - 28)) & ~15;
            lo_new = need < 0 ? 0 : need;
            if (lo_new > lo_loaded) lo_new = lo_loaded;
            if (lo_new < lo_loaded - 48) lo_new = lo_loaded - 48;   // (cannot happen: a round reads at most 12 words)
        }
        const int32_t nch = (lo_loaded - lo_new) >> 4;   // 0 .. 3
        u32x4 pre[3];
#pragma unroll
        for (int32_t k = 0; k < 3; k++)
            if (k < nch) pre[k] = ldu128(sb_ + (lo_loaded - 16 * (k + 1)));
        if (full) {
            uint32_t o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int i = 0; i < 16; i++) {
                ZB_REFILL();
                ZB_SYMBOL(o[i >> 1], 16 * (i & 1));
                ZB_SYMBOL(o[i >> 1], 16 * (i & 1) + 8);
            }
            stu128(dst + done, u32x4{o[0], o[1], o[2], o[3]});
            stu128(dst + done + 16, u32x4{o[4], o[5], o[6], o[7]});
            done += 32;
        } else if (run) {
            const uint32_t n_this = outn - done;
            for (uint32_t i = 0; i < n_this; i++) {
                ZB_REFILL();
                uint32_t v = 0;
                ZB_SYMBOL(v, 0);
                *(gptr)(dst + done + i) = (uint8_t)v;
            }
            done = outn;
        }
#pragma unroll
        for (int32_t k = 0; k < 3; k++)
            if (k < nch) file(lo_loaded - 16 * (k + 1), pre[k]);
        lo_loaded = lo_new;
        if (left < 0) act = false;
    }
#undef ZB_REFILL
#undef ZB_SYMBOL
    return had && act && done == outn && left == 0;
}

// A pool of waves over groups of 16 consecutive blocks.
__global__ void __launch_bounds__(64) zb_lit(ZbPools zp) {
    __shared__ ZlLds L;
    __shared__ struct {
        const uint8_t* pay[ZL_BLOCKS];   // literal payload of the block (after the literals header)
        const uint8_t* dpay[ZL_BLOCKS];  // literal payload of the block that holds the tree
        uint8_t* out[ZL_BLOCKS];
        uint32_t lcsize[ZL_BLOCKS], dlcsize[ZL_BLOCKS], regen[ZL_BLOCKS], streams[ZL_BLOCKS], avail[ZL_BLOCKS], frame[ZL_BLOCKS], slot[ZL_BLOCKS];
        uint32_t own[ZL_BLOCKS];         // the block carries its own tree (type 2)
    } S;
    const uint32_t lane = threadIdx.x;
    const uint32_t nblocks = min(zp.counters[0], zp.block_cap);
    const uint32_t ngroups = (nblocks + ZL_BLOCKS - 1) / ZL_BLOCKS;
    for (uint32_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        // ---- lanes 0 .. 15: the group's blocks
        bool want = false;
        uint32_t def = ZB_NONE;
        if (lane < ZL_BLOCKS) {
            const uint32_t bi = grp * ZL_BLOCKS + lane;
            S.slot[lane] = ZB_NONE;
            S.streams[lane] = 0;
            if (bi < nblocks) {
                const ZbBlock b = zp.blocks[bi];
                if (b.btype == 2 && b.ltype >= 2 && b.frame < zp.frame_cap) {
                    const ZbFrame f = zp.frames[b.frame];
                    if (!f.punt && b.huf_def != ZB_NONE && b.regen) {
                        want = true;
                        def = b.huf_def;
                        const ZbBlock d = b.ltype == 2 ? b : zp.blocks[def];
                        S.pay[lane] = b.src + b.lpay;
                        S.dpay[lane] = d.src + d.lpay;
                        S.lcsize[lane] = b.lcsize;
                        S.dlcsize[lane] = d.lcsize;
                        S.regen[lane] = b.regen;
                        S.streams[lane] = b.lstreams;
                        S.out[lane] = zp.lit + b.lit_pos;
                        S.avail[lane] = (uint32_t)(f.base + f.avail - (b.src + b.lpay));
                        S.frame[lane] = b.frame;
                        S.own[lane] = b.ltype == 2 ? 1u : 0u;
                    }
                }
            }
        }
        // table slots: consecutive blocks that use the same tree share one
        const uint32_t def_prev = (uint32_t)__shfl_up((int)def, 1, 64);
        const bool leader = want && (lane == 0 || def != def_prev);
        const uint64_t lm = __ballot(leader);
        const uint32_t nslots = (uint32_t)__popcll(lm);
        const uint32_t my_slot = (uint32_t)__popcll(lm & ((2ull << lane) - 1)) - 1;   // (lanes that want: slot of the nearest leader at or below)
        if (want) S.slot[lane] = my_slot;
        if (lane < ZL_BLOCKS) L.nw[lane] = 0;
        wave_sync();
        if (!nslots) continue;
        // ---- the tree descriptions into LDS (10 lanes x 16 bytes per slot), weights lane per slot, classes by the wave
        {
            // leader lane of slot s = the s-th set bit of lm
            uint64_t m = lm;
            for (uint32_t s = 0; s < nslots; s++) {
                const uint32_t ll_ = (uint32_t)__builtin_ctzll(m);
                m &= m - 1;
                const uint8_t* p = S.dpay[ll_];
                const uint32_t n = min(S.dlcsize[ll_], ZL_DESC);
                if (lane < ZL_DESC / 16) {
                    u32x4 v = u32x4{0, 0, 0, 0};
                    const uint32_t b = 16 * lane;
                    if (b + 16 <= n) {
                        v = ldu128(p + b);
                    } else if (b < n) {
                        uint32_t w4[4] = {0, 0, 0, 0};
                        for (uint32_t k = 0; b + k < n; k++) w4[k >> 2] |= (uint32_t)ldu8(p + b + k) << (8 * (k & 3));
                        v = u32x4{w4[0], w4[1], w4[2], w4[3]};
                    }
                    *(u32x4*)&L.p.desc[s][b] = v;
                }
            }
        }
        wave_sync();
        if (leader) {
            uint32_t used = 0;
            const uint32_t nw = zb_huf_weights(L.p.desc[my_slot], min(S.dlcsize[lane], ZL_DESC), L.b.wts[my_slot], L.p.fse[my_slot], L.p.norm[my_slot],
                                               L.p.next[my_slot], &used);
            L.nw[my_slot] = nw;
            L.used[my_slot] = used;
        }
        wave_sync();
        for (uint32_t s = 0; s < nslots; s++) {
            zb_huf_classes(L, s);
            wave_sync();
        }
        // ---- passes: as many slots as `tab` holds
        uint32_t s0 = 0;
        while (s0 < nslots) {
            uint32_t s1 = s0, acc = 0;
            while (s1 < nslots) {
                const uint32_t sz = L.bits[s1] ? (1u << L.bits[s1]) : 0u;
                if (acc + sz > ZL_TAB) break;
                if (lane == 0) L.toff[s1] = acc;
                acc += sz;
                s1++;
            }
            wave_sync();
            for (uint32_t s = s0; s < s1; s++) zb_huf_fill(L, s);
            wave_sync();
            // ---- the streams of the blocks whose slot is in [s0, s1): lane 4 g + j = stream j of block g
            const uint32_t g = lane >> 2, j = lane & 3;
            bool act = false, ok = true;
            const uint8_t* sp = nullptr;
            uint8_t* sd = nullptr;
            uint32_t sn = 0, outn = 0, mbits = 0, av = 0, toff = 0;
            const uint32_t slot = S.slot[g];
            const bool mine = slot != ZB_NONE && slot >= s0 && slot < s1;
            if (mine) {
                mbits = L.bits[slot];
                toff = L.toff[slot];
                const uint32_t used = S.own[g] ? L.used[slot] : 0u;
                const uint32_t lc = S.lcsize[g], regen = S.regen[g];
                if (!mbits || used > lc) {
                    ok = false;
                } else if (S.streams[g] == 1) {
                    if (j == 0) {
                        act = true;
                        sp = S.pay[g] + used;
                        sn = lc - used;
                        outn = regen;
                        sd = S.out[g];
                        av = S.avail[g] - used;
                    }
                } else {
                    const uint8_t* q = S.pay[g] + used;
                    const uint32_t left = lc - used;
                    if (left < 6) {
                        ok = false;
                    } else {
                        const uint32_t s1_ = ldu16(q), s2_ = ldu16(q + 2), s3_ = ldu16(q + 4);
                        const uint32_t per = (regen + 3) / 4;
                        if (6 + s1_ + s2_ + s3_ > left || per * 3 > regen) {
                            ok = false;
                        } else {
                            const uint32_t so = j == 0 ? 0u : j == 1 ? s1_ : j == 2 ? s1_ + s2_ : s1_ + s2_ + s3_;
                            sn = j == 0 ? s1_ : j == 1 ? s2_ : j == 2 ? s3_ : left - 6 - s1_ - s2_ - s3_;
                            outn = j < 3 ? per : regen - 3 * per;
                            sp = q + 6 + so;
                            sd = S.out[g] + j * per;
                            av = S.avail[g] - used - 6 - so;
                            act = true;
                        }
                    }
                }
            }
            const bool sok = zb_huf_stream(L, L.tab + toff, mbits, sp, sn, av, sd, outn, act);
            if (mine && (!ok || (act && !sok))) zp.frames[S.frame[g]].punt = 1;   // (the one-wave decoder names the error)
            wave_sync();
            if (s1 == s0) {   // a table larger than `tab` cannot happen (11 bits = 2048 entries)
                s1 = s0 + 1;
            }
            s0 = s1;
        }
        wave_sync();
    }
}

// ---------------------------------------------------------------------------------------------------- zb_seq
// Four blocks per wave, 16 lanes each (they run the same chain; lane 0 of the 16 stores), taken from zb_hdr's size classes,
// largest first: a wave's instructions serve four chains of about the same length.  (Four CONSECUTIVE blocks put every
// heavy block of the C5 batch — 2048 of 12 000 — into a wave of its own: three rounds of 768 resident waves; one block
// per wave, 3072 resident, made every chain slower: three waves per SIMD share its issue slots.)
constexpr uint32_t ZS_BLOCKS = 4;
constexpr uint32_t ZS_WIN = 1024;       // bytes of the bit stream staged per block
constexpr uint32_t ZS_DESC = 96;        // bytes of a table description staged in LDS
// A wave issues one instruction every ~4 cycles whatever it is, so the chain is written for INSTRUCTION COUNT: one 8-byte
// LDS read per table and sequence brings everything the step needs —
//   low word:  extra bits of the symbol | state bits << 8 | state base << 16;   high word: the value's baseline
//   (literal / match length baseline of the code; 1 << code for offsets) —
// the 64 stream bits below the position come from three dwords of the window and two v_alignbit, all six bit fields are
// cut out of them with one 64-bit shift + v_bfe each, the repeat-offset rules are selects, and errors accumulate in
// registers that are looked at after the loop.
struct ZsLds {
    uint64_t ll[ZS_BLOCKS][512], ml[ZS_BLOCKS][512], of[ZS_BLOCKS][256];
    // stream bytes [wlo - 8, wlo + ZS_WIN + 24) of the block's bit stream (bytes in front of the stream read as zeros)
    __attribute__((aligned(16))) uint8_t win[ZS_BLOCKS][ZS_WIN + 48];
    __attribute__((aligned(16))) uint8_t desc[ZS_BLOCKS][3][ZS_DESC];
    int16_t norm[ZS_BLOCKS][3][64];
    uint16_t next[ZS_BLOCKS][3][64];
    uint32_t log[ZS_BLOCKS][3];      // table log per block and table; 0xFF: could not be built
};
static_assert(sizeof(ZsLds) <= 53 * 1024, "zb_seq: three waves per CU");

// table t (0 LL, 1 OF, 2 ML) of one block from its description d[0, n) (LDS; mode 1: one symbol, mode 2: FSE) or the
// predefined distribution (mode 0) -> packed entries; returns the table log or 0xFF
__device__ inline uint32_t zs_build(uint64_t* tab, int t, uint32_t mode, const uint8_t* d, uint32_t n, int16_t* norm, uint16_t* next) {
    const int max_sym = t == 0 ? 35 : t == 1 ? 31 : 52, max_log = t == 1 ? 8 : 9;
    auto pack = [&](uint32_t s, uint32_t nb, uint32_t base) -> uint64_t {
        const uint32_t xb = t == 0 ? (uint32_t)Z_LL_BITS[s] : t == 1 ? s : (uint32_t)Z_ML_BITS[s];
        const uint32_t vb = t == 0 ? Z_LL_BASE[s] : t == 1 ? (1u << s) : Z_ML_BASE[s];
        return (uint64_t)(xb | (nb << 8) | ((base * 8) << 16)) | ((uint64_t)vb << 32);   // (base * 8: the next state as a byte offset)
    };
    if (mode == 0) {
        const uint32_t* pre = t == 0 ? g_zpre.ll : t == 1 ? g_zpre.of : g_zpre.ml;
        const int log = t == 1 ? 5 : 6;
        for (int i = 0; i < (1 << log); i++) {
            const uint32_t e = pre[i];
            tab[i] = pack(e & 255, (e >> 8) & 255, e >> 16);
        }
        return (uint32_t)log;
    }
    if (mode == 1) {
        if (n < 1 || d[0] > max_sym || (t == 1 && d[0] > 26)) return 0xFF;
        tab[0] = pack(d[0], 0, 0);
        return 0;
    }
    int nsym, log;
    if (!z_fse_header(d, n, max_sym, max_log, norm, &nsym, &log)) return 0xFF;
    if (t == 1 && nsym > 27) return 0xFF;   // offsets of 2^27 and more (a 128 MiB window): the frame-serial decoder
    const int size = 1 << log;
    int high = size - 1;
    for (int s = 0; s < nsym; s++) {
        if (norm[s] == -1) {
            tab[high--] = (uint64_t)s;
            next[s] = 1;
        } else {
            next[s] = (uint16_t)norm[s];
        }
    }
    const int step = (size >> 1) + (size >> 3) + 3, mask = size - 1;
    int pos = 0;
    for (int s = 0; s < nsym; s++)
        for (int i = 0; i < norm[s]; i++) {
            tab[pos] = (uint64_t)s;
            do {
                pos = (pos + step) & mask;
            } while (pos > high);
        }
    if (pos != 0) return 0xFF;
    for (int i = 0; i < size; i++) {
        const uint32_t s = (uint32_t)tab[i];
        const uint32_t ns = next[s]++;
        const uint32_t nb = (uint32_t)log - (31u - (uint32_t)__clz((int)ns));
        tab[i] = pack(s, nb, (ns << nb) - (uint32_t)size);
    }
    return (uint32_t)log;
}

__global__ void __launch_bounds__(64) zb_seq(ZbPools zp) {
    __shared__ ZsLds L;
    const uint32_t lane = threadIdx.x, gi = lane >> 4, li = lane & 15;
    const uint32_t nblocks = min(zp.counters[0], zp.block_cap);
    uint32_t ccount[4], gbase[5];
    gbase[0] = 0;
#pragma unroll
    for (int c = 0; c < 4; c++) {
        ccount[c] = min(zp.counters[8 + c], zp.block_cap);
        gbase[c + 1] = gbase[c] + (ccount[c] + ZS_BLOCKS - 1) / ZS_BLOCKS;
    }
    const uint32_t ngroups = gbase[4];
    for (uint32_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const uint32_t cls = grp >= gbase[3] ? 3u : grp >= gbase[2] ? 2u : grp >= gbase[1] ? 1u : 0u;
        const uint32_t cb = cls == 3 ? gbase[3] : cls == 2 ? gbase[2] : cls == 1 ? gbase[1] : 0u;
        const uint32_t cn = cls == 3 ? ccount[3] : cls == 2 ? ccount[2] : cls == 1 ? ccount[1] : ccount[0];
        const uint32_t at = (grp - cb) * ZS_BLOCKS + gi;
        const uint32_t bi = at < cn ? zp.lists[(size_t)cls * zp.block_cap + at] : ZB_NONE;
        ZbBlock b;
        b.nseq = 0;
        bool act = false;
        if (bi < nblocks) {
            b = zp.blocks[bi];
            if (b.btype == 2 && b.nseq && b.frame < zp.frame_cap) act = !zp.frames[b.frame].punt;
        }
        if (!__ballot(act)) continue;
        // ---- table descriptions into LDS: lanes 0 .. 5 of the group move 16 bytes each, table after table
        uint32_t mode0 = 0, mode1 = 0, mode2 = 0, dn0 = 0, dn1 = 0, dn2 = 0;
        if (act) {
#pragma unroll
            for (int t = 0; t < 3; t++) {
                const uint32_t df = b.def[t];
                if (df != ZB_NONE) {
                    const ZbBlock d = df == bi ? b : zp.blocks[df];
                    const uint32_t md = (d.modes >> (6 - 2 * t)) & 3;
                    const uint8_t* p = d.src + d.desc[t];
                    const uint32_t n = min(d.bsize - d.desc[t], ZS_DESC);
                    if (t == 0) { mode0 = md; dn0 = n; }
                    if (t == 1) { mode1 = md; dn1 = n; }
                    if (t == 2) { mode2 = md; dn2 = n; }
                    if (li < ZS_DESC / 16) {
                        const uint32_t o = 16 * li;
                        u32x4 v = u32x4{0, 0, 0, 0};
                        if (o + 16 <= n) {
                            v = ldu128(p + o);
                        } else if (o < n) {
                            uint32_t w4[4] = {0, 0, 0, 0};
                            for (uint32_t k = 0; o + k < n; k++) w4[k >> 2] |= (uint32_t)ldu8(p + o + k) << (8 * (k & 3));
                            v = u32x4{w4[0], w4[1], w4[2], w4[3]};
                        }
                        *(u32x4*)&L.desc[gi][t][o] = v;
                    }
                }
            }
        }
        wave_sync();
        // ---- the three tables of a block by lanes 0, 1, 2 of its group
        if (act && li < 3) {
            const int t = (int)li;
            uint64_t* tab = t == 0 ? L.ll[gi] : t == 1 ? L.of[gi] : L.ml[gi];
            const uint32_t md = t == 0 ? mode0 : t == 1 ? mode1 : mode2;
            const uint32_t n = t == 0 ? dn0 : t == 1 ? dn1 : dn2;
            L.log[gi][t] = zs_build(tab, t, md, L.desc[gi][t], n, L.norm[gi][t], L.next[gi][t]);
        }
        wave_sync();
        const uint32_t lll = L.log[gi][0], ofl = L.log[gi][1], mll = L.log[gi][2];
        bool ok = act && lll != 0xFF && ofl != 0xFF && mll != 0xFF;
        // ---- the bit stream through a window in LDS
        const uint8_t* sb_ = act ? b.src + b.bits_off : nullptr;
        const int32_t sn_ = act ? (int32_t)(b.bsize - b.bits_off) : 0;
        int32_t bitpos0 = 0;
        if (ok) {
            const uint8_t lastb = sn_ > 0 ? ldu8(sb_ + sn_ - 1) : (uint8_t)0;
            if (!lastb) ok = false;
            else bitpos0 = (sn_ - 1) * 8 + (31 - __clz((int)lastb));
        }
        int32_t wlo = 0;   // (a multiple of 16)
        int32_t bp = bitpos0;   // stream bit position - 8 * wlo: window bit index of the 64 bits below the position
        auto refill = [&](bool me) {   // the 16 lanes of a group whose block needs it move its window
            if (me) {
                const int32_t bitpos = bp + 8 * wlo;
                const int32_t hi_byte = min(sn_, (bitpos >> 3) + 9);
                const int32_t lo_byte = hi_byte > (int32_t)ZS_WIN ? (hi_byte - (int32_t)ZS_WIN) & ~15 : 0;
                for (uint32_t k = li * 16; k < ZS_WIN + 48; k += 16 * 16) {
                    const int32_t o = lo_byte - 8 + (int32_t)k;
                    u32x4 v = u32x4{0, 0, 0, 0};
                    if (o >= 0 && o + 16 <= sn_) {
                        v = ldu128(sb_ + o);
                    } else if (o + 16 > 0 && o < sn_) {
                        uint32_t w4[4] = {0, 0, 0, 0};
                        for (int32_t q = 0; q < 16; q++)
                            if (o + q >= 0 && o + q < sn_) w4[q >> 2] |= (uint32_t)ldu8(sb_ + o + q) << (8 * (q & 3));
                        v = u32x4{w4[0], w4[1], w4[2], w4[3]};
                    }
                    *(u32x4*)&L.win[gi][k] = v;
                }
                wlo = lo_byte;
                bp = bitpos - 8 * wlo;
            }
            wave_sync();
        };
        refill(ok);
        const uint32_t* w32 = (const uint32_t*)L.win[gi];
        auto top64 = [&](int32_t at) -> uint64_t {   // the 64 stream bits below window bit index `at` + 64, top-aligned
            const uint32_t idx = (uint32_t)at >> 5, shv = (uint32_t)at & 31;
            const uint32_t d0 = w32[idx], d1 = w32[idx + 1], d2 = w32[idx + 2];
            return (uint64_t)__builtin_amdgcn_alignbit(d1, d0, shv) | ((uint64_t)__builtin_amdgcn_alignbit(d2, d1, shv) << 32);
        };
        auto field = [](uint64_t C, uint32_t q, uint32_t w) -> uint32_t {   // w bits of C from bit q (q = 64 with w = 0 is fine)
            return __builtin_amdgcn_ubfe((uint32_t)(C >> (q & 63)), 0u, w);
        };
        uint32_t sl = 0, so = 0, sm = 0;   // the three states, as byte offsets into their tables
        if (ok) {   // initial states: LL, OF, ML
            const uint64_t C = top64(bp);
            sl = field(C, 64 - lll, lll) << 3;
            so = field(C, 64 - lll - ofl, ofl) << 3;
            sm = field(C, 64 - lll - ofl - mll, mll) << 3;
            bp -= (int32_t)(lll + ofl + mll);
            if (bp < 0) ok = false;
        }
        // records: literal length, match length, offset VALUE (1 .. 3: a repeat-offset code, resolved by zb_exec) — 12 bytes
        uint32_t* rec = (uint32_t*)zp.rec + 3 * (act ? b.rec_pos : 0);
        const uint32_t nseq = ok ? b.nseq : 0u;
        uint32_t k = 0;
        uint32_t over64 = 0;   // (looked at after the loop)
        const uint8_t* LLt = (const uint8_t*)L.ll[gi];
        const uint8_t* MLt = (const uint8_t*)L.ml[gi];
        const uint8_t* OFt = (const uint8_t*)L.of[gi];
        // one sequence; `last`: the block's last sequence updates no states
        auto step = [&](bool last) {
            const uint64_t el = *(const uint64_t*)(LLt + sl), em = *(const uint64_t*)(MLt + sm), eo = *(const uint64_t*)(OFt + so);
            const uint64_t C = top64(bp);
            const uint32_t lo_l = (uint32_t)el, lo_m = (uint32_t)em, lo_o = (uint32_t)eo;
            const uint32_t x_o = lo_o & 255, x_m = lo_m & 255, x_l = lo_l & 255;
            const uint32_t q1 = 64 - x_o, q2 = q1 - x_m, q3 = q2 - x_l;
            const uint32_t ofv = (uint32_t)(eo >> 32) + field(C, q1, x_o);
            const uint32_t mlen = (uint32_t)(em >> 32) + field(C, q2, x_m);
            const uint32_t llen = (uint32_t)(el >> 32) + field(C, q3, x_l);
            if (!last) {
                const uint32_t n_l = __builtin_amdgcn_ubfe(lo_l, 8, 8), n_m = __builtin_amdgcn_ubfe(lo_m, 8, 8), n_o = __builtin_amdgcn_ubfe(lo_o, 8, 8);
                const uint32_t q4 = q3 - n_l, q5 = q4 - n_m, q6 = q5 - n_o;
                // (more than 64 bits in one sequence — an offset beyond 2^21 next to a long match and a long literal run —
                // turns q6 negative: the frame goes to the frame-serial decoder, see over64 below)
                over64 |= q6;
                sl = (lo_l >> 16) + (field(C, q4, n_l) << 3);
                sm = (lo_m >> 16) + (field(C, q5, n_m) << 3);
                so = (lo_o >> 16) + (field(C, q6, n_o) << 3);
                bp -= (int32_t)(64 - q6);
            } else {
                bp -= (int32_t)(64 - q3);
            }
            // (the 16 lanes of the group hold the same record: one address, one transaction)
            const uint32_t r3[3] = {llen, mlen, ofv};
            __builtin_memcpy((gptr)(uint8_t*)rec, r3, 12);
            rec += 3;
            k++;
        };
        for (;;) {   // four sequences between two looks at the window
            const bool run = ok && k + 4 < nseq;
            if (!__ballot(run)) break;
            const bool low = run && wlo > 0 && (bp >> 3) < 64;
            if (__ballot(low)) refill(low);
            if (run) {
                step(false);
                step(false);
                step(false);
                step(false);
                if (bp < 0) ok = false;
            }
        }
        for (;;) {
            const bool run = ok && k + 1 < nseq;
            if (!__ballot(run)) break;
            const bool low = run && wlo > 0 && (bp >> 3) < 24;
            if (__ballot(low)) refill(low);
            if (run) {
                step(false);
                if (bp < 0) ok = false;
            }
        }
        {
            const bool run = ok && k + 1 == nseq;
            const bool low = run && wlo > 0 && (bp >> 3) < 24;
            if (__ballot(low)) refill(low);
            if (run) step(true);
        }
        if (act && li == 0) {
            // all bits consumed exactly, never more than 64 in one sequence
            if (ok && (k != b.nseq || bp != 0 || wlo != 0 || (int32_t)over64 < 0)) ok = false;
            if (!ok) zp.frames[b.frame].punt = 1;
        }
        wave_sync();
    }
}

// ---------------------------------------------------------------------------------------------------- zb_exec_wg
// Frames of many short sequences (text: a few output bytes per sequence) by a WORKGROUP per frame, the way sb_lz4_big.h
// copies an LZ4 block: 512 records per round — repeat offsets by the composition scan over the workgroup, output and
// literal positions by prefix sums — then the round's output in windows of 8 KiB: one u16 entry per output byte, either
// FINAL (0x8000 | byte: a literal, or a match byte whose source lies in front of the window, read back from HBM) or a
// POINTER to the window byte it copies; a round of `ent[p] = ent[ent[p]]` per pointer entry resolves chains of matches
// in log2(depth) rounds.  The wave executor (zb_exec) pays ~70 cycles per sequence in batch bookkeeping and dependency
// rounds; this one ~9 cycles per output BYTE whatever the sequences look like: zb_scan picks per frame.
constexpr uint32_t ZX_T = 256, ZX_REC = 512, ZX_WIN = 8192;
struct ZxLds {
    __attribute__((aligned(16))) uint16_t ent[ZX_WIN + 16];
    uint32_t r_out[ZX_REC + 4];   // output position of the record's first byte; [nrec] = end of the round's output
    uint32_t r_lit[ZX_REC];       // position of its literals in the block's literal buffer
    uint32_t r_off[ZX_REC];       // match distance (directly behind r_lit: the entries phase picks one of the two with one load)
    uint32_t r_ll[ZX_REC];        // literal length (the rest of the record's bytes are the match)
    uint32_t wt[4][3];            // repeat-offset maps of the waves
    uint32_t wsum[8];
    uint32_t err;
};
__device__ __forceinline__ void zx_copy(uint8_t* dst, const uint8_t* src, uint32_t n) {   // by the workgroup, no overlap
    const uint32_t t = threadIdx.x;
    uint32_t head = (uint32_t)((16 - ((uintptr_t)dst & 15)) & 15);
    if (head > n) head = n;
    if (t < head) dst[t] = ldu8(src + t);
    const uint32_t nvec = (n - head) >> 4;
    for (uint32_t i = t; i < nvec; i += ZX_T) stu128(dst + head + 16 * (uint64_t)i, ldu128(src + head + 16 * (uint64_t)i));
    const uint32_t done = head + (nvec << 4);
    if (t < n - done) dst[done + t] = ldu8(src + done + t);
}
__global__ void __launch_bounds__(ZX_T, 4) zb_exec_wg(InflateJob* q, Status* st, ZbPools zp, uint32_t queue, RelCtx rc) {
    __shared__ ZxLds L;
    const uint32_t t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const uint32_t nframes = min(zp.counters[1], zp.frame_cap);
    for (uint32_t fi = blockIdx.x; fi < nframes; fi += gridDim.x) {
        const ZbFrame f = zp.frames[fi];
        if (f.queue != queue || !f.wg || f.punt) continue;   // (punted frames are handed back by zb_exec)
        uint8_t* const dst = job_dst(rc, f.dst, f.page, f.rel != 0);
        if (!dst) continue;
        const uint32_t out_len = f.out_len;
        uint32_t op = 0, err = 0;
        uint32_t e0 = 1, e1 = 4, e2 = 8;
        __syncthreads();
        if (t == 0) L.err = 0;
        __syncthreads();
        for (uint32_t k = 0; k < f.nblocks && !err; k++) {
            const ZbBlock b = zp.blocks[f.first + k];
            if (b.btype == 0 || b.btype == 1) {
                if (out_len - op < b.out_size) { err = 10; break; }
                if (b.btype == 0) {
                    zx_copy(dst + op, b.src, b.out_size);
                } else {
                    const uint8_t v = ldu8(b.src);
                    for (uint32_t i = t; i < b.out_size; i += ZX_T) dst[op + i] = v;
                }
                op += b.out_size;
                wave_stores_visible();
                __syncthreads();
                continue;
            }
            const uint8_t* litp;
            if (b.ltype == 0) {
                litp = b.src + b.lpay;
            } else {
                uint8_t* lp = zp.lit + b.lit_pos;
                if (b.ltype == 1) {
                    const uint8_t v = ldu8(b.src + b.lpay);
                    for (uint32_t i = t; i < b.regen; i += ZX_T) lp[i] = v;
                    wave_stores_visible();
                    __syncthreads();
                }
                litp = lp;
            }
            uint32_t lit_pos = 0;
            const uint32_t* recs = (const uint32_t*)zp.rec + 3 * b.rec_pos;
            for (uint32_t done = 0; done < b.nseq && !err; done += ZX_REC) {
                const uint32_t nrec = min(ZX_REC, b.nseq - done);
                // ---- thread t: records 2 t and 2 t + 1 of the round
                uint32_t ll[2], ml[2], ofv[2];
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const uint32_t r = 2 * t + u;
                    ll[u] = ml[u] = 0;
                    ofv[u] = 1;
                    if (r < nrec) {
                        uint32_t v3[3];
                        __builtin_memcpy(v3, (gcptr)(const uint8_t*)(recs + 3 * (done + r)), 12);
                        ll[u] = v3[0];
                        ml[u] = v3[1];
                        ofv[u] = v3[2];
                    }
                }
                bool bad = ll[0] >= (1u << 18) || ml[0] >= (1u << 18) || ll[1] >= (1u << 18) || ml[1] >= (1u << 18);
                // ---- repeat offsets: the records' maps, composed over the thread, the wave, the workgroup (cf. zb_exec)
                uint32_t m0[2], m1[2], m2[2];
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const bool valid = 2 * t + u < nrec;
                    const uint32_t idx = ofv[u] - 1 + (ll[u] == 0 ? 1u : 0u);
                    const bool real = ofv[u] > 3;
                    if (valid && real && ofv[u] - 3 >= ZB_SYM) bad = true;
                    m0[u] = !valid ? zb_sym(0) : real ? ofv[u] - 3 : idx == 0 ? zb_sym(0) : idx == 1 ? zb_sym(1) : idx == 2 ? zb_sym(2) : (zb_sym(0) | 1u);
                    m1[u] = (valid && (real || idx >= 1)) ? zb_sym(0) : zb_sym(1);
                    m2[u] = (valid && (real || idx >= 2)) ? zb_sym(1) : zb_sym(2);
                }
                // the thread's two records together, then an inclusive scan over the wave
                uint32_t c0 = zb_subst(m0[1], m0[0], m1[0], m2[0]), c1 = zb_subst(m1[1], m0[0], m1[0], m2[0]), c2 = zb_subst(m2[1], m0[0], m1[0], m2[0]);
#pragma unroll
                for (uint32_t dlt = 1; dlt < 64; dlt <<= 1) {
                    const uint32_t a0_ = (uint32_t)__shfl_up((int)c0, dlt, 64), a1_ = (uint32_t)__shfl_up((int)c1, dlt, 64), a2_ = (uint32_t)__shfl_up((int)c2, dlt, 64);
                    if (lane >= dlt) {
                        const uint32_t n0 = zb_subst(c0, a0_, a1_, a2_), n1 = zb_subst(c1, a0_, a1_, a2_), n2 = zb_subst(c2, a0_, a1_, a2_);
                        c0 = n0;
                        c1 = n1;
                        c2 = n2;
                    }
                }
                __syncthreads();
                if (lane == 63) {
                    L.wt[wv][0] = c0;
                    L.wt[wv][1] = c1;
                    L.wt[wv][2] = c2;
                }
                __syncthreads();
                // the state in front of the thread's first record: the lanes before it in the wave, after the waves before it
                uint32_t p0 = (uint32_t)__shfl_up((int)c0, 1, 64), p1 = (uint32_t)__shfl_up((int)c1, 1, 64), p2 = (uint32_t)__shfl_up((int)c2, 1, 64);
                if (lane == 0) {
                    p0 = zb_sym(0);
                    p1 = zb_sym(1);
                    p2 = zb_sym(2);
                }
                uint32_t w0_ = zb_sym(0), w1_ = zb_sym(1), w2_ = zb_sym(2);   // the waves before mine, composed
                uint32_t tot0 = zb_sym(0), tot1 = zb_sym(1), tot2 = zb_sym(2);
                for (uint32_t w = 0; w < 4; w++) {
                    const uint32_t x0 = L.wt[w][0], x1 = L.wt[w][1], x2 = L.wt[w][2];
                    const uint32_t n0 = zb_subst(x0, tot0, tot1, tot2), n1 = zb_subst(x1, tot0, tot1, tot2), n2 = zb_subst(x2, tot0, tot1, tot2);
                    tot0 = n0;
                    tot1 = n1;
                    tot2 = n2;
                    if (w + 1 == wv) {
                        w0_ = tot0;
                        w1_ = tot1;
                        w2_ = tot2;
                    }
                }
                {
                    const uint32_t n0 = zb_subst(p0, w0_, w1_, w2_), n1 = zb_subst(p1, w0_, w1_, w2_), n2 = zb_subst(p2, w0_, w1_, w2_);
                    p0 = n0;
                    p1 = n1;
                    p2 = n2;
                }
                uint32_t off[2];
                {
                    const uint32_t s0 = zb_subst(m0[0], p0, p1, p2), s1 = zb_subst(m1[0], p0, p1, p2), s2 = zb_subst(m2[0], p0, p1, p2);
                    off[0] = zb_resolve(s0, e0, e1, e2);
                    off[1] = zb_resolve(zb_subst(m0[1], s0, s1, s2), e0, e1, e2);
                }
                {   // the state after the round
                    const uint32_t n0 = zb_resolve(tot0, e0, e1, e2), n1 = zb_resolve(tot1, e0, e1, e2), n2 = zb_resolve(tot2, e0, e1, e2);
                    e0 = n0;
                    e1 = n1;
                    e2 = n2;
                }
                // ---- positions: output (ll + ml) and literal (ll) prefix sums over the round
                const uint32_t len2 = ll[0] + ml[0] + ll[1] + ml[1], lit2 = ll[0] + ll[1];
                const uint32_t li = wave_scan_dpp(len2), lli = wave_scan_dpp(lit2);
                if (lane == 63) {
                    L.wsum[wv] = li;
                    L.wsum[4 + wv] = lli;
                }
                __syncthreads();
                uint32_t obase = li - len2, lbase = lli - lit2;
                for (uint32_t w = 0; w < wv; w++) {
                    obase += L.wsum[w];
                    lbase += L.wsum[4 + w];
                }
                const uint64_t round_out = (uint64_t)L.wsum[0] + L.wsum[1] + L.wsum[2] + L.wsum[3];
                const uint64_t round_lit = (uint64_t)L.wsum[4] + L.wsum[5] + L.wsum[6] + L.wsum[7];
                if (round_out > (uint64_t)(out_len - op) || (uint64_t)lit_pos + round_lit > b.regen) bad = true;
                {
                    uint32_t o = op + obase, lp_ = lit_pos + lbase;
#pragma unroll
                    for (int u = 0; u < 2; u++) {
                        const uint32_t r = 2 * t + u;
                        if (r < nrec) {
                            L.r_out[r] = o;
                            L.r_lit[r] = lp_;
                            L.r_off[r] = off[u];
                            L.r_ll[r] = ll[u];
                            if (ml[u] && (off[u] == 0 || off[u] > o + ll[u])) bad = true;   // a match may not reach in front of the frame
                        }
                        o += ll[u] + ml[u];
                        lp_ += ll[u];
                    }
                }
                if (bad) L.err = 30;
                const uint32_t o_end = op + (uint32_t)round_out;
                if (t == 0) L.r_out[nrec] = o_end;
                __syncthreads();
                if (L.err) { err = L.err; break; }
                // ---- windows of the round's output (sb_lz4_big.h's copy stage; literals come from the block's literal buffer)
                const uint32_t nwin = (o_end - op + ZX_WIN - 1) / ZX_WIN;
                const uint32_t wstep = nwin ? min(ZX_WIN, ((o_end - op + nwin - 1) / nwin + 255) & ~255u) : ZX_WIN;
                for (uint32_t w0 = op; w0 < o_end; w0 += wstep) {
                    const uint32_t wl = min(wstep, o_end - w0);
                    __syncthreads();
                    for (uint32_t i = t * 8; i < wl; i += ZX_T * 8) *(u32x4*)(L.ent + i) = u32x4{0, 0, 0, 0};
                    __syncthreads();
                    for (uint32_t r = t; r < nrec; r += ZX_T) {
                        const uint32_t o = L.r_out[r];
                        if (o >= w0 && o - w0 < wl && L.r_out[r + 1] > o) L.ent[o - w0] = (uint16_t)(r + 1);
                    }
                    __syncthreads();
                    {   // every byte finds its record: max-scan of the markers (wave = a quarter of the window)
                        const uint32_t wq = ((wl + 255) / 256) * 64;
                        const uint32_t b0 = wv * wq;
                        uint32_t carry = 0;
                        if (b0 < wl) {
                            uint32_t lo = 0, hi = nrec;   // largest r with r_out[r] <= w0 + b0 (r_out[0] <= w0)
                            const uint32_t x = w0 + b0;
                            while (hi - lo > 1) {
                                const uint32_t mid = (lo + hi) >> 1;
                                if (L.r_out[mid] <= x) lo = mid;
                                else hi = mid;
                            }
                            carry = lo + 1;
                        }
                        const uint32_t b1 = min(b0 + wq, wl);
                        for (uint32_t rb = b0; rb < b1; rb += 512) {
                            uint32_t m[8];
#pragma unroll
                            for (uint32_t u = 0; u < 8; u++) {
                                const uint32_t p_ = rb + 64 * u + lane;
                                m[u] = p_ < b1 ? (uint32_t)L.ent[p_] : 0u;
                            }
#pragma unroll
                            for (uint32_t u = 0; u < 8; u++) {
                                const uint32_t p_ = rb + 64 * u + lane;
                                const uint32_t sc = max(wave_scan_max_dpp(m[u]), carry);
                                carry = rdlane(sc, 63);
                                if (p_ < b1) L.ent[p_] = (uint16_t)sc;
                            }
                        }
                    }
                    __syncthreads();
                    // first entries (thread owns the bytes t + 256 k): a literal or a byte from in front of the window (both from
                    // HBM, second pass), or a pointer
                    uint32_t pend = 0, gmask = 0;
#pragma unroll 1
                    for (uint32_t k0 = 0; k0 * ZX_T < wl; k0 += 8) {
                        uint32_t id[8], kk[8], x[8];
#pragma unroll
                        for (uint32_t u = 0; u < 8; u++) {
                            const uint32_t p_ = t + ZX_T * (k0 + u);
                            id[u] = p_ < wl ? (uint32_t)L.ent[p_] - 1 : 0u;
                        }
#pragma unroll
                        for (uint32_t u = 0; u < 8; u++) {
                            const uint32_t p_ = t + ZX_T * (k0 + u);
                            kk[u] = w0 + p_ - L.r_out[id[u]];   // byte of the record
                            x[u] = L.r_ll[id[u]];
                        }
#pragma unroll
                        for (uint32_t u = 0; u < 8; u++) {
                            const uint32_t p_ = t + ZX_T * (k0 + u);
                            if (p_ < wl) {
                                if (kk[u] < x[u]) {
                                    gmask |= 1u << (k0 + u);   // a literal (ent[p] keeps the record until the second pass)
                                } else {
                                    const uint32_t dist = L.r_off[id[u]];
                                    if (dist <= p_) {
                                        L.ent[p_] = (uint16_t)(p_ - dist);
                                        pend |= 1u << (k0 + u);
                                    } else {
                                        gmask |= 1u << (k0 + u);
                                    }
                                }
                            }
                        }
                    }
                    while (gmask) {   // sixteen loads in flight
                        uint32_t kq[16], gv[16];
#pragma unroll
                        for (int u = 0; u < 16; u++) {
                            kq[u] = gmask ? (uint32_t)__builtin_ctz(gmask) : 32u;
                            gmask &= gmask - 1;   // (0 stays 0)
                            gv[u] = kq[u] < 32 ? (uint32_t)L.ent[t + ZX_T * kq[u]] - 1 : 0u;
                        }
#pragma unroll
                        for (int u = 0; u < 16; u++) {
                            if (kq[u] < 32) {
                                const uint32_t r = gv[u], p_ = t + ZX_T * kq[u];
                                const uint32_t kb = w0 + p_ - L.r_out[r];
                                gv[u] = kb < L.r_ll[r] ? (uint32_t)ldu8(litp + L.r_lit[r] + kb) : (uint32_t)ldu8(dst + (w0 + p_ - L.r_off[r]));
                            }
                        }
#pragma unroll
                        for (int u = 0; u < 16; u++)
                            if (kq[u] < 32) L.ent[t + ZX_T * kq[u]] = (uint16_t)(0x8000u | gv[u]);
                    }
                    __syncthreads();
                    for (;;) {   // pointer jumping: every pointer entry takes over the entry it points to
                        int any = 0;
                        uint32_t m = pend;
                        while (m) {
                            uint32_t kq[4], pp[4], ss[4];
#pragma unroll
                            for (int u = 0; u < 4; u++) {
                                kq[u] = m ? (uint32_t)__builtin_ctz(m) : 32u;
                                m &= m - 1;
                                pp[u] = kq[u] < 32 ? (uint32_t)L.ent[t + ZX_T * kq[u]] : 0u;
                            }
#pragma unroll
                            for (int u = 0; u < 4; u++) ss[u] = L.ent[pp[u]];
#pragma unroll
                            for (int u = 0; u < 4; u++) {
                                if (kq[u] < 32) {
                                    L.ent[t + ZX_T * kq[u]] = (uint16_t)ss[u];
                                    if (ss[u] & 0x8000u) pend &= ~(1u << kq[u]);
                                    else any = 1;
                                }
                            }
                        }
                        if (!__syncthreads_or(any)) break;
                    }
                    {   // window -> HBM: 16-byte groups of the line frame of dst + w0
                        const uint32_t a0 = (uint32_t)((uintptr_t)(dst + w0) & 15);
                        uint8_t* gb = dst + w0 - a0;
                        const uint32_t ng = (a0 + wl + 15) >> 4;
                        for (uint32_t g = t; g < ng; g += ZX_T) {
                            const int32_t f0 = (int32_t)(16 * g) - (int32_t)a0;
                            if (f0 >= 0 && (uint32_t)f0 + 16 <= wl) {
                                uint32_t w4[4];
#pragma unroll
                                for (int qq = 0; qq < 4; qq++) {
                                    w4[qq] = (uint32_t)(L.ent[f0 + 4 * qq] & 0xFF) | ((uint32_t)(L.ent[f0 + 4 * qq + 1] & 0xFF) << 8) |
                                             ((uint32_t)(L.ent[f0 + 4 * qq + 2] & 0xFF) << 16) | ((uint32_t)(L.ent[f0 + 4 * qq + 3] & 0xFF) << 24);
                                }
                                stu128(gb + 16 * g, u32x4{w4[0], w4[1], w4[2], w4[3]});
                            } else {
                                for (int bq = 0; bq < 16; bq++) {
                                    const int32_t fq = f0 + bq;
                                    if (fq >= 0 && (uint32_t)fq < wl) gb[16 * g + bq] = (uint8_t)L.ent[fq];
                                }
                            }
                        }
                    }
                    wave_stores_visible();   // later windows read these bytes back
                }
                op = o_end;
                lit_pos += (uint32_t)round_lit;
                __syncthreads();
            }
            if (err) break;
            const uint32_t rest = b.regen - lit_pos;
            if (out_len - op < rest) { err = 31; break; }
            zx_copy(dst + op, litp + lit_pos, rest);
            op += rest;
            wave_stores_visible();
            __syncthreads();
        }
        if (!err && op != out_len) err = 34;
        if (err && t == 0) raise(st, SB_ERR_EXTERNAL, f.page, 120 + err);
        if (t == 0) {
            unsigned long long ns = 0;
            for (uint32_t k = 0; k < f.nblocks; k++) ns += zp.blocks[f.first + k].nseq;
            atomicAdd(&zp.stats[0], 1ull);
            atomicAdd(&zp.stats[2], (unsigned long long)f.nblocks);
            atomicAdd(&zp.stats[3], ns);
        }
        wave_stores_visible();
        __syncthreads();
    }
}

// phase clocks of lane 0 of one wave (development: -DZB_TL; read back by sb_debug_zb_timers)
#ifdef ZB_TL
#define ZBT_BEGIN unsigned long long zbt_acc[12] = {0}; unsigned long long zbt_t = __builtin_readcyclecounter();
#define ZBT(i) do { const unsigned long long n_ = __builtin_readcyclecounter(); zbt_acc[i] += n_ - zbt_t; zbt_t = n_; } while (0)
#define ZBT_CNT(i, v) do { zbt_acc[i] += (v); } while (0)
#define ZBT_END(cond) do { if ((cond) && (threadIdx.x & 63) == 0) for (int i_ = 0; i_ < 12; i_++) atomicAdd(&zp.stats[4 + i_], zbt_acc[i_]); } while (0)
#else
#define ZBT_BEGIN
#define ZBT(i)
#define ZBT_CNT(i, v)
#define ZBT_END(cond)
#endif
// ---------------------------------------------------------------------------------------------------- zb_exec
// A pool of waves over the frames: blocks in order.
__global__ void __launch_bounds__(64) zb_exec(InflateJob* q, Status* st, ZbPools zp, uint32_t queue, RelCtx rc) {
    __shared__ LzSeqLds ring;
    const uint32_t lane = threadIdx.x;
    const uint32_t nframes = min(zp.counters[1], zp.frame_cap);
    for (uint32_t fi = blockIdx.x; fi < nframes; fi += gridDim.x) {
        const ZbFrame f = zp.frames[fi];
        if (f.queue != queue) continue;
        if (f.punt) {
            if (lane == 0) {
                q[f.job].codec = SB_CODEC_ZSTD | (f.rel ? JOB_REL : 0u);
                atomicAdd(&zp.stats[1], 1ull);
            }
            continue;
        }
        if (f.wg) continue;   // zb_exec_wg's
        uint8_t* const fdst = job_dst(rc, f.dst, f.page, f.rel != 0);
        if (!fdst) continue;
        if (lane == 0) {
            unsigned long long ns = 0;
            for (uint32_t k = 0; k < f.nblocks; k++) ns += zp.blocks[f.first + k].nseq;
            atomicAdd(&zp.stats[0], 1ull);
            atomicAdd(&zp.stats[2], (unsigned long long)f.nblocks);
            atomicAdd(&zp.stats[3], ns);
        }
        uint8_t* dst = fdst;
        const uint32_t out_len = f.out_len;
        ZBT_BEGIN
        uint32_t op = 0, err = 0;
        uint32_t e0 = 1, e1 = 4, e2 = 8;
        LzSeqExec ex(ring, dst);
        ex.restart(0);
        for (uint32_t k = 0; k < f.nblocks && !err; k++) {
            const ZbBlock b = zp.blocks[f.first + k];
            if (b.btype == 0) {
                if (out_len - op < b.out_size) { err = 10; break; }
                ex.finish(op);
                wave_copy_g2g(dst + op, b.src, b.out_size);
                op += b.out_size;
                ex.restart(op);
                continue;
            }
            if (b.btype == 1) {
                if (out_len - op < b.out_size) { err = 11; break; }
                ex.finish(op);
                const uint8_t v = ldu8(b.src);
                for (uint32_t i = lane; i < b.out_size; i += 64) dst[op + i] = v;
                op += b.out_size;
                ex.restart(op);
                continue;
            }
            const uint8_t* litp;
            if (b.ltype == 0) {
                litp = b.src + b.lpay;
            } else {
                uint8_t* lp = zp.lit + b.lit_pos;
                if (b.ltype == 1) {
                    const uint8_t v = ldu8(b.src + b.lpay);
                    for (uint32_t i = lane; i < b.regen; i += 64) lp[i] = v;
                    wave_stores_visible();
                }
                litp = lp;
            }
            uint32_t lit_pos = 0;
            // records are requested two batches ahead and the first 8 literal bytes of every sequence one batch ahead: a
            // batch of short sequences would otherwise wait for HBM twice
            struct Rec { uint32_t ll, ml, ofv; };
            const uint32_t* recs = (const uint32_t*)zp.rec + 3 * b.rec_pos;
            auto load_rec = [&](uint32_t base) -> Rec {
                Rec r;
                r.ll = r.ml = 0;
                r.ofv = 4;
                if (base + lane < b.nseq) __builtin_memcpy(&r, (gcptr)(const uint8_t*)(recs + 3 * (base + lane)), 12);
                return r;
            };
            auto load_lit8 = [&](uint32_t at, uint32_t ll) -> uint64_t {   // min(ll, 8) literal bytes at litp + at (never beyond regen)
                uint64_t v = 0;
                if (ll == 0 || at >= b.regen) return v;
                if (at + 8 <= b.regen) return ldu64(litp + at);
                for (uint32_t q = 0; at + q < b.regen && q < 8; q++) v |= (uint64_t)ldu8(litp + at + q) << (8 * q);
                return v;
            };
            Rec r_cur = load_rec(0), r_nxt = load_rec(64);
            uint64_t l8_cur = load_lit8(wave_scan_dpp(r_cur.ll) - r_cur.ll, r_cur.ll);
            for (uint32_t done = 0; done < b.nseq; done += 64) {
                const uint32_t nb = min(64u, b.nseq - done);
                const bool have = lane < nb;
                const Rec r_n2 = load_rec(done + 128);
                const uint32_t llen = r_cur.ll, mlen = r_cur.ml;
                if (__ballot(have && (llen >= (1u << 18) || mlen >= (1u << 18)))) { err = 29; break; }
                ZBT(0);
                // ---- repeat offsets (RFC 8878 3.1.1.5): what a sequence does to the three of them is a small map — push a new
                // offset, keep, swap, rotate, push "first minus one" — whose outputs are constants or "entry value i minus d";
                // an inclusive scan of the batch's maps under composition gives every lane the state after its sequence in
                // terms of the state the batch started with, and the offset a sequence uses is the first value of that state
                uint32_t t0, t1, t2;
                {
                    const uint32_t ofv = r_cur.ofv;
                    const uint32_t idx = ofv - 1 + (llen == 0 ? 1u : 0u);
                    const bool real = ofv > 3;
                    t0 = real ? ofv - 3 : idx == 0 ? zb_sym(0) : idx == 1 ? zb_sym(1) : idx == 2 ? zb_sym(2) : (zb_sym(0) | 1u);
                    t1 = (real || idx >= 1) ? zb_sym(0) : zb_sym(1);
                    t2 = (real || idx >= 2) ? zb_sym(1) : zb_sym(2);
                    if (__ballot(have && real && ofv - 3 >= ZB_SYM)) { err = 28; break; }
#pragma unroll
                    for (uint32_t dlt = 1; dlt < 64; dlt <<= 1) {
                        const uint32_t a0_ = (uint32_t)__shfl_up((int)t0, dlt, 64), a1_ = (uint32_t)__shfl_up((int)t1, dlt, 64),
                                       a2_ = (uint32_t)__shfl_up((int)t2, dlt, 64);
                        if (lane >= dlt) {
                            const uint32_t n0 = zb_subst(t0, a0_, a1_, a2_), n1 = zb_subst(t1, a0_, a1_, a2_), n2 = zb_subst(t2, a0_, a1_, a2_);
                            t0 = n0;
                            t1 = n1;
                            t2 = n2;
                        }
                    }
                }
                ZBT(1);
                const uint32_t off = have ? zb_resolve(t0, e0, e1, e2) : 1u;
                {   // the state after the batch's last sequence
                    const uint32_t l0 = rdlane(t0, nb - 1), l1 = rdlane(t1, nb - 1), l2 = rdlane(t2, nb - 1);
                    const uint32_t n0 = zb_resolve(l0, e0, e1, e2), n1 = zb_resolve(l1, e0, e1, e2), n2 = zb_resolve(l2, e0, e1, e2);
                    e0 = n0;
                    e1 = n1;
                    e2 = n2;
                }
                const uint32_t lsum = wave_scan_dpp(have ? llen : 0u), osum = wave_scan_dpp(have ? llen + mlen : 0u);
                const bool bad = have && ((uint64_t)lit_pos + lsum > b.regen || (uint64_t)op + osum > out_len || off == 0 || off > op + osum - mlen);
                if (__ballot(bad)) { err = 30; break; }
                const uint32_t lit_total = rdlane(lsum, 63);
                const uint64_t l8_nxt = load_lit8(lit_pos + lit_total + wave_scan_dpp(r_nxt.ll) - r_nxt.ll, r_nxt.ll);
                ZBT(2);
                ZBT_CNT(11, 1);
                op += ex.run(nb, llen, mlen, off, litp + lit_pos, op, true, l8_cur);
                ZBT(3);
                lit_pos += lit_total;
                r_cur = r_nxt;
                r_nxt = r_n2;
                l8_cur = l8_nxt;
                wave_sync();
            }
            if (err) break;
            const uint32_t rest = b.regen - lit_pos;
            if (out_len - op < rest) { err = 31; break; }
            ex.finish(op);
            if (rest >= 256) {
                wave_copy_g2g(dst + op, litp + lit_pos, rest);
            } else {
                for (uint32_t i = lane; i < rest; i += 64) dst[op + i] = ldu8(litp + lit_pos + i);
            }
            op += rest;
            ex.restart(op);
        }
        if (!err && op != out_len) err = 34;
        ZBT(4);
        ZBT_END(fi == 1 || fi == 0);
        if (err && lane == 0) raise(st, SB_ERR_EXTERNAL, f.page, 120 + err);
        wave_stores_visible();
    }
}

}  // namespace sb
