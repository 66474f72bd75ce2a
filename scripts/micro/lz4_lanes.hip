// A lane-serial LZ4 parse on the device, measured (the CPU model is scripts/sim/lz4_lanes_sim.py).
//
// The shipped writer (k_enc_lz4_chunks, sb_lz4.h) takes 64 POSITIONS per wave step and spends ~1 000 instructions on them
// (~14 per input byte: issue-bound at 83 GB/s on C3').  Here a LANE owns a 1 KiB sub-block of a 64 KiB region and walks it the
// way liblz4 does — hash the four bytes at pos, look at the bucket, check the candidates, extend the best, jump over the
// match — with the wave's 64 lanes sharing two-way hash tables in LDS (this region's and the region's before: the model's
// 1.008 x liblz4's size).  The kernel parses only: sequences (position, length) go to HBM, nothing is emitted — a real writer
// adds the token / literal copies (~20 % in the shipped kernel) and the join of the lanes' outputs (the stitch pass that exists
// for chunks).  What the micro answers: how fast is the parse, and is its size what the model says.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/micro/lz4_lanes.hip -o scripts/micro/bin/lz4_lanes.bin
//   scripts/micro/bin/lz4_lanes.bin scripts/micro/bin/c3_page.bin [copies = 1024] [regions per wave = 4] [liblz4 bytes]
// (c3_page.bin: one C3 values page, scripts/micro/dump_c3.py writes it; liblz4's size of it as printed there)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

constexpr uint32_t SUB = 1024, LANES = 64, REGION = SUB * LANES;
constexpr uint32_t HB = 11, NB = 1u << HB;           // buckets of two u16 ways: 8 KB per table, two tables per wave
constexpr uint32_t WPB = 4;                          // waves per workgroup (64 KB of LDS: two workgroups per CU)
constexpr uint32_t MFLIMIT = 12, LASTLIT = 5;
constexpr uint32_t MAXSEQ = SUB / 4;                 // sequences a lane can produce

__device__ __forceinline__ uint32_t ld4(const uint8_t* p) {
    uint32_t v;
    __builtin_memcpy(&v, (const __attribute__((address_space(1))) uint8_t*)p, 4);
    return v;
}
__device__ __forceinline__ uint64_t ld8(const uint8_t* p) {
    uint64_t v;
    __builtin_memcpy(&v, (const __attribute__((address_space(1))) uint8_t*)p, 8);
    return v;
}

struct Seq {
    uint32_t pos, len;
};

// one wave = RW consecutive regions of one copy of the block
__global__ void __launch_bounds__(64 * WPB) k_parse(const uint8_t* all, uint32_t n, uint32_t stride, uint32_t rw, uint32_t waves_per_block,
                                                   Seq* seqs /* copy 0 only: [region][lane][MAXSEQ] */, uint32_t* nseq /* copy 0: [region][lane] */,
                                                   unsigned long long* totals /* [0] sequences, [1] matched bytes, [2] steps */) {
    __shared__ uint32_t tabs[WPB][2][NB];
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t gw = blockIdx.x * WPB + wv;
    const uint32_t copy = gw / waves_per_block, part = gw % waves_per_block;
    const uint8_t* data = all + (uint64_t)copy * stride;
    const uint32_t nreg = (n + REGION - 1) / REGION;
    uint32_t cur = 0;   // which of the two tables is this region's
    for (uint32_t i = lane; i < NB; i += 64) {
        tabs[wv][0][i] = 0xFFFFFFFFu;
        tabs[wv][1][i] = 0xFFFFFFFFu;
    }
    unsigned long long my_seq = 0, my_bytes = 0, my_steps = 0;
    for (uint32_t r = part * rw; r < min(nreg, (part + 1) * rw); r++, cur ^= 1) {
        uint32_t* tnew = tabs[wv][cur];
        uint32_t* told = tabs[wv][cur ^ 1];
        if (r != part * rw)
            for (uint32_t i = lane; i < NB; i += 64) tnew[i] = 0xFFFFFFFFu;   // (told = the region before; the first region has none)
        const uint32_t r0 = r * REGION;
        uint32_t pos = r0 + lane * SUB;
        const uint32_t end = min(n, pos + SUB);
        const uint32_t lim = min(end, n > LASTLIT ? n - LASTLIT : 0u);   // a match may not run into the block's last five bytes (nor out of the sub-block)
        uint32_t ns = 0;
        Seq* out = seqs ? seqs + ((uint64_t)r * LANES + lane) * MAXSEQ : nullptr;
        bool act = pos < end;
        while (__ballot(act)) {
            if (!act) continue;
            my_steps++;
            if (pos + MFLIMIT > n || pos + 4 > end) {
                act = false;
                continue;
            }
            const uint32_t v = ld4(data + pos);
            const uint32_t h = (v * 2654435761u) >> (32 - HB);
            const uint32_t bn = tnew[h], bo = told[h];
            const uint32_t rel = pos - r0;
            if (rel != 0xFFFFu) tnew[h] = (bn << 16) | rel;   // way 0 = newest (0xFFFF is "empty": that one position is not entered)
            // four candidates: this region's two, the two of the region before
            uint32_t cand[4];
            cand[0] = (bn & 0xFFFFu) != 0xFFFFu ? r0 + (bn & 0xFFFFu) : 0xFFFFFFFFu;
            cand[1] = (bn >> 16) != 0xFFFFu ? r0 + (bn >> 16) : 0xFFFFFFFFu;
            cand[2] = ((bo & 0xFFFFu) != 0xFFFFu && r0 >= REGION) ? r0 - REGION + (bo & 0xFFFFu) : 0xFFFFFFFFu;
            cand[3] = ((bo >> 16) != 0xFFFFu && r0 >= REGION) ? r0 - REGION + (bo >> 16) : 0xFFFFFFFFu;
            uint32_t cv[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {   // (a later lane's position is in the table too: only earlier bytes can be copied)
                const bool ok = cand[k] < pos && pos - cand[k] <= 65535u;
                cand[k] = ok ? cand[k] : 0xFFFFFFFFu;
                cv[k] = ok ? ld4(data + cand[k]) : ~v;
            }
            uint32_t best = 0xFFFFFFFFu, bm = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (cand[k] == 0xFFFFFFFFu || cv[k] != v) continue;
                uint32_t mm = 4;
                bool open = true;
                while (open && pos + mm + 8 <= lim) {   // eight bytes at a time
                    const uint64_t x = ld8(data + cand[k] + mm) ^ ld8(data + pos + mm);
                    if (x) {
                        mm += (uint32_t)__builtin_ctzll(x) >> 3;
                        open = false;
                    } else {
                        mm += 8;
                    }
                }
                while (open && pos + mm < lim && data[cand[k] + mm] == data[pos + mm]) mm++;
                if (mm > bm) {
                    bm = mm;
                    best = cand[k];
                }
            }
            if (best == 0xFFFFFFFFu || pos + bm > lim) {
                pos += 1;
                act = pos < end;
                continue;
            }
            if (out && ns < MAXSEQ) out[ns] = Seq{pos, bm};
            ns++;
            my_bytes += bm;
            pos += bm;
            act = pos < end;
        }
        if (nseq && copy == 0) nseq[r * LANES + lane] = ns;
        my_seq += ns;
    }
    for (int o = 32; o; o >>= 1) {
        my_seq += __shfl_down(my_seq, o, 64);
        my_bytes += __shfl_down(my_bytes, o, 64);
        my_steps += __shfl_down(my_steps, o, 64);
    }
    if (lane == 0) {
        atomicAdd(&totals[0], my_seq);
        atomicAdd(&totals[1], my_bytes);
        atomicAdd(&totals[2], my_steps);
    }
}

static uint64_t seq_bytes(uint64_t lit, uint64_t mlen) {
    uint64_t s = 1 + lit + 2;
    if (lit >= 15) s += 1 + (lit - 15) / 255;
    const uint64_t m = mlen - 4;
    if (m >= 15) s += 1 + (m - 15) / 255;
    return s;
}

int main(int argc, char** argv) {
    if (argc < 2) {
        fprintf(stderr, "usage: %s page.bin [copies] [regions per wave] [liblz4 bytes]\n", argv[0]);
        return 2;
    }
    FILE* f = fopen(argv[1], "rb");
    if (!f) {
        perror(argv[1]);
        return 2;
    }
    std::vector<uint8_t> page;
    {
        uint8_t buf[65536];
        size_t k;
        while ((k = fread(buf, 1, sizeof buf, f)) > 0) page.insert(page.end(), buf, buf + k);
        fclose(f);
    }
    const uint32_t n = (uint32_t)page.size();
    const uint32_t P = argc > 2 ? (uint32_t)atoi(argv[2]) : 1024, RW = argc > 3 ? (uint32_t)atoi(argv[3]) : 4;
    const uint64_t ref = argc > 4 ? strtoull(argv[4], nullptr, 10) : 0;
    const uint32_t stride = (n + 64 + 255) & ~255u, nreg = (n + REGION - 1) / REGION, wpb = (nreg + RW - 1) / RW;
    uint8_t* d;
    hipMalloc(&d, (uint64_t)P * stride + 64);
    for (uint32_t c = 0; c < P; c++) hipMemcpy(d + (uint64_t)c * stride, page.data(), n, hipMemcpyHostToDevice);
    Seq* seqs;
    uint32_t* nseq;
    unsigned long long* totals;
    hipMalloc(&seqs, (uint64_t)nreg * LANES * MAXSEQ * sizeof(Seq));
    hipMalloc(&nseq, nreg * LANES * 4);
    hipMalloc(&totals, 24);
    const uint32_t waves = P * wpb, grid = (waves + WPB - 1) / WPB;
    printf("page %u bytes x %u copies = %.1f MB; %u regions of 64 KiB, %u per wave: %u waves in %u workgroups of %u\n", n, P, (double)P * n / 1e6, nreg, RW,
           waves, grid, 64 * WPB);
    // ---- the parse of copy 0, sized exactly (sequences in stream order; literals carried over the lanes' borders)
    hipMemset(totals, 0, 24);
    k_parse<<<(wpb + WPB - 1) / WPB, 64 * WPB>>>(d, n, stride, RW, wpb, seqs, nseq, totals);
    hipDeviceSynchronize();
    {
        std::vector<Seq> hs((size_t)nreg * LANES * MAXSEQ);
        std::vector<uint32_t> hn(nreg * LANES);
        hipMemcpy(hs.data(), seqs, hs.size() * sizeof(Seq), hipMemcpyDeviceToHost);
        hipMemcpy(hn.data(), nseq, hn.size() * 4, hipMemcpyDeviceToHost);
        uint64_t total = 0, anchor = 0, count = 0, bad = 0;
        for (uint32_t rl = 0; rl < nreg * LANES; rl++)
            for (uint32_t k = 0; k < hn[rl]; k++) {
                const Seq s = hs[(size_t)rl * MAXSEQ + k];
                if (s.pos < anchor || s.len < 4 || s.pos + s.len > n - LASTLIT) bad++;
                total += seq_bytes(s.pos - anchor, s.len);
                anchor = s.pos + s.len;
                count++;
            }
        const uint64_t lit = n - anchor;
        total += 1 + lit + (lit >= 15 ? 1 + (lit - 15) / 255 : 0);
        printf("copy 0: %llu sequences, %llu bytes as an LZ4 block", (unsigned long long)count, (unsigned long long)total);
        if (ref) printf(" = %.4f x liblz4's %llu", (double)total / (double)ref, (unsigned long long)ref);
        printf("; %llu malformed\n", (unsigned long long)bad);
    }
    // ---- all copies, timed
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) {
        hipMemset(totals, 0, 24);
        hipEventRecord(e0);
        k_parse<<<grid, 64 * WPB>>>(d, n, stride, RW, wpb, nullptr, nullptr, totals);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        unsigned long long t[3];
        hipMemcpy(t, totals, 24, hipMemcpyDeviceToHost);
        printf("parse of %u copies: %.3f ms = %.1f GB/s of input; %.2f bytes per lane step, %.1f bytes per sequence, %.0f %% of the bytes in matches\n", P, ms,
               (double)P * n / (ms * 1e-3) / 1e9, (double)P * n / (double)t[2], (double)P * n / (double)t[0], 100.0 * (double)t[1] / ((double)P * n));
    }
    return 0;
}
