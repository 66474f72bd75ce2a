// Phase timing of the workgroup-per-block LZ4 decoder (strawboat_amd/csrc/sb_lz4_big.h) on one input file:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I strawboat_amd/csrc scripts/micro/lz4_big_probe.hip -o scripts/micro/lz4_big_probe.bin
//   scripts/micro/lz4_big_probe.bin <raw file> <lz4 block of it> [blocks]
// [blocks] workgroups decode the block, each into an output area of its own: 1 = the latency of a lone workgroup,
// 1024 = the chip as full as the bench has it.  (python: oracle.sbo.block_compress(sbo.LZ4, raw) writes the block.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#ifndef NO_PROFILE
#define SB_LZ4_BIG_PROFILE 1
#endif
__device__ unsigned long long g_prof_big[32];
#include "sb_lz4.h"
#include "sb_lz4_big.h"
using namespace sb;

static bool slurp(const char* path, std::vector<uint8_t>& v) {
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    uint8_t buf[65536];
    size_t r;
    while ((r = fread(buf, 1, sizeof buf, f)) > 0) v.insert(v.end(), buf, buf + r);
    fclose(f);
    return true;
}
__global__ void __launch_bounds__(LB_T, 4) k_dec(const uint8_t* comp, const uint32_t* sizes, uint8_t* out, uint32_t n, uint32_t* errs) {
    __shared__ Lz4BigLds lds;
    const uint32_t e = lz4_inflate_block_wg(comp, sizes[0], out + (size_t)blockIdx.x * ((n + 255) & ~255u), n, lds);
    if (threadIdx.x == 0) errs[blockIdx.x] = e;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
static const char* PH[] = {"stage", "next", "marks", "records", "positions", "markers", "rowscan", "entries", "doubling", "store"};
int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    if (argc < 3) return 1;
    std::vector<uint8_t> in, comp;
    if (!slurp(argv[1], in) || !slurp(argv[2], comp)) { printf("cannot open the inputs\n"); return 1; }
    const uint32_t n = (uint32_t)in.size();
    const int blocks = argc > 3 ? atoi(argv[3]) : 1;
    const uint32_t cap = (uint32_t)comp.size(), ostride = (n + 255) & ~255u;
    uint8_t *d_in, *d_comp, *d_out;
    uint32_t *d_sizes, *d_errs;
    CK(hipMalloc(&d_in, n + 64));
    CK(hipMalloc(&d_comp, cap + 64));
    CK(hipMalloc(&d_out, (size_t)ostride * blocks + 64));
    CK(hipMalloc(&d_sizes, 64));
    CK(hipMalloc(&d_errs, 4 * blocks));
    CK(hipMemcpy(d_in, in.data(), n, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_comp, comp.data(), cap, hipMemcpyHostToDevice));
    const uint32_t csize = cap;
    CK(hipMemcpy(d_sizes, &csize, 4, hipMemcpyHostToDevice));
    printf("%u -> %u bytes, %d workgroups, LDS %zu B\n", n, csize, blocks, sizeof(Lz4BigLds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; rep++) {
        unsigned long long zero[32] = {0};
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_prof_big), zero, sizeof zero));
        CK(hipEventRecord(e0));
        k_dec<<<blocks, LB_T>>>(d_comp, d_sizes, d_out, n, d_errs);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<uint32_t> errs(blocks);
        CK(hipMemcpy(errs.data(), d_errs, 4 * blocks, hipMemcpyDeviceToHost));
        std::vector<uint8_t> out(n);
        CK(hipMemcpy(out.data(), d_out + (size_t)ostride * (blocks - 1), n, hipMemcpyDeviceToHost));
        const bool ok = memcmp(out.data(), in.data(), n) == 0;
        unsigned long long pr[32];
        CK(hipMemcpyFromSymbol(pr, HIP_SYMBOL(g_prof_big), sizeof pr));
        printf("rep %d: %.3f ms  %.1f GB/s  err %u  %s\n", rep, ms, (double)n * blocks / ms / 1e6, errs[0], ok ? "ok" : "MISMATCH");
        if (rep == 2) {
            unsigned long long tot = 0;
            for (int i = 0; i < 10; i++) tot += pr[i];
            for (int i = 0; i < 10; i++) printf("  %-10s %10llu cycles  %5.1f %%\n", PH[i], pr[i], 100.0 * pr[i] / (tot ? tot : 1));
            printf("  chunks %llu  windows %llu  records %llu  jump rounds %llu  HBM bytes of thread 0 %llu\n", pr[16], pr[18], pr[19], pr[20], pr[21]);
        }
    }
    return 0;
}
