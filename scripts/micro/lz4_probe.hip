// Phase timing of the wave-per-block LZ4 codec (strawboat_amd/csrc/sb_lz4.h) on one input file:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSB_LZ4_PROFILE -I strawboat_amd/csrc scripts/micro/lz4_probe.hip -o scripts/micro/lz4_probe.bin
//   scripts/micro/lz4_probe.bin <raw input file> [blocks]
// Every block of the launch works on the same input (its own output area): [blocks] = 1 shows the latency of a lone
// wave, 1024 the behaviour with the chip as full as the bench has it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define SB_LZ4_PROFILE 1
__device__ unsigned long long g_prof[32];
#include "sb_lz4.h"
#include "sb_zstd_enc.h"
#include "sb_zstd.h"
using namespace sb;

__global__ void __launch_bounds__(64) k_enc(const uint8_t* src, uint32_t n, uint8_t* dst, uint32_t cap, uint32_t* sizes) {
    __shared__ Lz4EncLds<11, 13> lds;
    const uint32_t sz = lz4_compress_wave_fast<11, 13>(src, n, dst + (size_t)blockIdx.x * cap, lds);
    if (threadIdx.x == 0) sizes[blockIdx.x] = sz;
}
__global__ void __launch_bounds__(64) k_dec(const uint8_t* comp, uint32_t cap, const uint32_t* sizes, uint8_t* out, uint32_t n, uint32_t* errs) {
    __shared__ Lz4DecLds lds;
    const uint32_t e = lz4_inflate_block(comp + (size_t)blockIdx.x * cap, sizes[blockIdx.x], out + (size_t)blockIdx.x * ((n + 255) & ~255u), n, lds);
    if (threadIdx.x == 0) errs[blockIdx.x] = e;
}
__global__ void __launch_bounds__(64) k_zenc(const uint8_t* src, uint32_t n, uint8_t* dst, uint32_t cap, uint32_t* sizes, uint8_t* scratch) {
    __shared__ ZEncLds lds;
    // inputs of up to 16 KiB are compressed the way the chunk kernel does it (a frame of its own: zstd_compress_block_alone)
    const uint32_t sz = n <= 32768 ? zstd_compress_block_alone(src, n, 0, n, dst + (size_t)blockIdx.x * cap, lds, scratch + (size_t)blockIdx.x * zstd_scratch_bytes(ZE_BLOCK), 32768)
                                   : zstd_compress_wave(src, n, dst + (size_t)blockIdx.x * cap, lds, scratch + (size_t)blockIdx.x * zstd_scratch_bytes(ZE_BLOCK));
    if (threadIdx.x == 0) sizes[blockIdx.x] = sz;
}
__global__ void __launch_bounds__(64) k_zdec(const uint8_t* comp, uint32_t cap, const uint32_t* sizes, uint8_t* out, uint32_t n, uint32_t* errs, uint8_t* zlit) {
    __shared__ ZWork wk;
    if (threadIdx.x == 0) wk.pre_built = 0;
    __syncthreads();
    const uint32_t got = zstd_inflate_wave(comp + (size_t)blockIdx.x * cap, sizes[blockIdx.x], out + (size_t)blockIdx.x * ((n + 255) & ~255u), n, &wk,
                                           zlit + (size_t)blockIdx.x * (128 * 1024 + 64));
    if (threadIdx.x == 0) errs[blockIdx.x] = wk.err ? (uint32_t)wk.err : (got == n ? 0u : 999u);
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
static const char* ENC[] = {"fill", "probe+cand+ext", "select", "emit", "flush_out", "last", "sizes+scan", ""};
static const char* DEC[] = {"refill", "parse", "walk", "literals", "records", "far", "groups", "flush", "big"};
int main(int argc, char** argv) {
    if (argc < 2) return 1;
    FILE* f = fopen(argv[1], "rb");
    if (!f) { printf("cannot open %s\n", argv[1]); return 1; }
    std::vector<uint8_t> in;
    uint8_t buf[65536];
    size_t r;
    while ((r = fread(buf, 1, sizeof buf, f)) > 0) in.insert(in.end(), buf, buf + r);
    fclose(f);
    const uint32_t n = (uint32_t)in.size();
    const int blocks = argc > 2 ? atoi(argv[2]) : 1;
    const uint32_t cap = (n + n / 255 + 16 + 255) & ~255u, ostride = (n + 255) & ~255u;
    uint8_t *d_in, *d_comp, *d_out;
    uint32_t *d_sizes, *d_errs;
    CK(hipMalloc(&d_in, n + 64));
    CK(hipMalloc(&d_comp, (size_t)cap * blocks));
    CK(hipMalloc(&d_out, (size_t)ostride * blocks + 64));
    CK(hipMalloc(&d_sizes, 4 * blocks));
    CK(hipMalloc(&d_errs, 4 * blocks));
    CK(hipMemcpy(d_in, in.data(), n, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    unsigned long long zero[32] = {0}, prof[32];
    for (int rep = 0; rep < 2; rep++) {
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_prof), zero, sizeof zero));
        hipEventRecord(e0);
        k_enc<<<blocks, 64>>>(d_in, n, d_comp, cap, d_sizes);
        hipEventRecord(e1);
        CK(hipDeviceSynchronize());
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        uint32_t sz;
        CK(hipMemcpy(&sz, d_sizes, 4, hipMemcpyDeviceToHost));
        CK(hipMemcpyFromSymbol(prof, HIP_SYMBOL(g_prof), sizeof prof));
        if (rep) {
            printf("encode: %u -> %u bytes, %d blocks, %.3f ms  (%.1f MB/s per wave, %.1f GB/s aggregate)\n", n, sz, blocks, ms, n / ms / 1e3, (double)n * blocks / ms / 1e6);
            unsigned long long tot = 0;
            for (int i = 0; i < 8; i++) tot += prof[i];
            for (int i = 0; i < 7; i++) printf("   %-16s %12llu ticks %5.1f %%\n", ENC[i], prof[i], 100.0 * prof[i] / (tot ? tot : 1));
            printf("   steps %llu, sequences %llu, big-path steps %llu\n", prof[16], prof[17], prof[18]);
        }
    }
    {
        uint8_t* d_scr;
        uint32_t* d_zs;
        CK(hipMalloc(&d_scr, zstd_scratch_bytes(ZE_BLOCK) * (size_t)blocks));
        CK(hipMalloc(&d_zs, 4 * blocks));
        uint8_t* d_z;
        CK(hipMalloc(&d_z, (size_t)cap * blocks));
        static const char* ZN[] = {"", "", "", "", "", "", "", "", "begin", "match", "record", "tail+visible", "literals", "sequences", "finish"};
        for (int rep = 0; rep < 2; rep++) {
            CK(hipMemcpyToSymbol(HIP_SYMBOL(g_prof), zero, sizeof zero));
            hipEventRecord(e0);
            k_zenc<<<blocks, 64>>>(d_in, n, d_z, cap, d_zs, d_scr);
            hipEventRecord(e1);
            CK(hipDeviceSynchronize());
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            uint32_t sz;
            CK(hipMemcpy(&sz, d_zs, 4, hipMemcpyDeviceToHost));
            CK(hipMemcpyFromSymbol(prof, HIP_SYMBOL(g_prof), sizeof prof));
            if (rep) {
                printf("zstd encode: %u -> %u bytes, %d blocks, %.3f ms (%.1f MB/s per wave)\n", n, sz, blocks, ms, n / ms / 1e3);
                unsigned long long tot = 0;
                for (int i = 8; i < 15; i++) tot += prof[i];
                for (int i = 8; i < 15; i++) printf("   %-16s %12llu ticks %5.1f %%\n", ZN[i], prof[i], 100.0 * prof[i] / (tot ? tot : 1));
                printf("   (literals = streams; before them: histogram + tree %llu, tree description %llu ticks)\n", prof[15], prof[7]);
                printf("   sequences %llu, literals %llu\n", prof[16], prof[17]);
            }
        }
        uint8_t* d_zlit;
        CK(hipMalloc(&d_zlit, (size_t)(128 * 1024 + 64) * blocks));
        static const char* DN[] = {"rest", "literals", "seq tables", "seq decode", "seq execute"};
        for (int rep = 0; rep < 2; rep++) {
            CK(hipMemcpyToSymbol(HIP_SYMBOL(g_prof), zero, sizeof zero));
            CK(hipMemset(d_out, 0xEE, (size_t)ostride * blocks));
            hipEventRecord(e0);
            k_zdec<<<blocks, 64>>>(d_z, cap, d_zs, d_out, n, d_errs, d_zlit);
            hipEventRecord(e1);
            CK(hipDeviceSynchronize());
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            CK(hipMemcpyFromSymbol(prof, HIP_SYMBOL(g_prof), sizeof prof));
            if (rep) {
                std::vector<uint8_t> out(n);
                uint32_t err;
                CK(hipMemcpy(&err, d_errs, 4, hipMemcpyDeviceToHost));
                CK(hipMemcpy(out.data(), d_out, n, hipMemcpyDeviceToHost));
                printf("zstd decode: err %u, %s, %.3f ms (%.1f MB/s per wave)\n", err, memcmp(out.data(), in.data(), n) ? "MISMATCH" : "round trip ok", ms, n / ms / 1e3);
                unsigned long long tot = 0;
                for (int i = 19; i < 24; i++) tot += prof[i];
                for (int i = 19; i < 24; i++) printf("   %-16s %12llu ticks %5.1f %%\n", DN[i - 19], prof[i], 100.0 * prof[i] / (tot ? tot : 1));
            }
        }
    }
    for (int rep = 0; rep < 2; rep++) {
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_prof), zero, sizeof zero));
        CK(hipMemset(d_out, 0xEE, (size_t)ostride * blocks));
        hipEventRecord(e0);
        k_dec<<<blocks, 64>>>(d_comp, cap, d_sizes, d_out, n, d_errs);
        hipEventRecord(e1);
        CK(hipDeviceSynchronize());
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        CK(hipMemcpyFromSymbol(prof, HIP_SYMBOL(g_prof), sizeof prof));
        if (rep) {
            std::vector<uint8_t> out(n);
            uint32_t err;
            CK(hipMemcpy(&err, d_errs + (blocks - 1), 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(out.data(), d_out + (size_t)ostride * (blocks - 1), n, hipMemcpyDeviceToHost));
            printf("decode: err %u, %s, %.3f ms  (%.1f MB/s per wave, %.1f GB/s aggregate)\n", err, memcmp(out.data(), in.data(), n) ? "MISMATCH" : "round trip ok", ms,
                   n / ms / 1e3, (double)n * blocks / ms / 1e6);
            unsigned long long tot = 0;
            for (int i = 0; i < 9; i++) tot += prof[i];
            for (int i = 0; i < 9; i++) printf("   %-16s %12llu ticks %5.1f %%\n", DEC[i], prof[i], 100.0 * prof[i] / (tot ? tot : 1));
            printf("   windows %llu, sequences %llu, batches %llu, groups %llu, far matches %llu, ring matches %llu\n", prof[16], prof[17], prof[18], prof[19], prof[20], prof[21]);
        }
    }
    return 0;
}
