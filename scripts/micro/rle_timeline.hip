// Timeline of the RLE page encoder's phases (s_memtime), one workgroup per 64 Ki-row f64 page,
// 1024 pages like bench.py.  Includes the library sources so that the device code is the product's.
#define SB_RLE_TIMELINE 1
#include "../../strawboat_amd/csrc/sb_encode.hip"  // link with sb_api.hip sb_decode.hip sb_nested.hip
#include <cstdio>
#include <random>
using namespace sb;

__global__ void __launch_bounds__(WG, 4) k_rle(const uint8_t* vals, const uint8_t* valid, uint64_t N, uint8_t* out, uint64_t slot,
                                               unsigned long long* tl) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[RleRows<8>::WORDS];
    const uint8_t* v = vals + (uint64_t)blockIdx.x * N * 8;
    ValidView vv{valid, (uint64_t)blockIdx.x * N};
    auto getv = [=](uint64_t i) { return ld_val<8>(v + i * 8); };
    enc_rle_rows<8, 2>(getv, vv, N, out + (uint64_t)blockIdx.x * slot, lds);
}

int main() {
    const uint64_t P = 1024, N = 65536, slot = N * 12 + 64;
    std::vector<uint64_t> h(P * N);
    std::vector<uint8_t> hv(P * N / 8);
    std::mt19937_64 rng(42);
    uint64_t cur = 0; int left = 0;
    for (auto& x : h) { if (left == 0) { cur = rng() % 256; left = 1 + rng() % 64; } x = cur; left--; }
    for (auto& b : hv) { b = 0; for (int k = 0; k < 8; k++) b |= (rng() % 10 != 0) << k; }
    uint8_t *d, *dv, *out; unsigned long long* tl;
    hipMalloc(&d, P * N * 8); hipMalloc(&dv, P * N / 8 + 64); hipMalloc(&out, P * slot); hipMalloc(&tl, 8 * 4096);
    hipMemcpy(d, h.data(), P * N * 8, hipMemcpyHostToDevice);
    hipMemcpy(dv, hv.data(), P * N / 8, hipMemcpyHostToDevice);
    hipMemset(tl, 0, 8 * 4096);
    hipMemcpyToSymbol(HIP_SYMBOL(g_tl), &tl, sizeof(tl));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; i++) k_rle<<<P, WG>>>(d, dv, N, out, slot, tl);
    hipEventRecord(a);
    for (int i = 0; i < 10; i++) k_rle<<<P, WG>>>(d, dv, N, out, slot, tl);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("k_rle: %.3f ms per launch (%.2f TB/s)\n", ms / 10, P * N * 8 / (ms / 10) / 1e9);
    std::vector<unsigned long long> t(4096);
    hipMemcpy(t.data(), tl, 8 * 4096, hipMemcpyDeviceToHost);
    // layout: wave w, chunk c, point p -> t[(w*16 + c)*8 + p]
    const char* names[8] = {"wait vn+stage", "B1", "fetch issue", "phase1+B2", "phase2", "scan+B3", "phase3", "B4"};
    for (int w = 0; w < 4; w++) {
        double acc[8] = {0};
        for (int c = 1; c < 15; c++)
            for (int p = 0; p < 8; p++) {
                unsigned long long prev = p ? t[(w * 16 + c) * 8 + p - 1] : t[(w * 16 + c - 1) * 8 + 7];
                acc[p] += (double)(t[(w * 16 + c) * 8 + p] - prev);
            }
        printf("wave %d:", w);
        double tot = 0;
        for (int p = 0; p < 8; p++) { printf("  %s=%.0f", names[p], acc[p] / 14); tot += acc[p] / 14; }
        printf("  | total/chunk=%.0f ticks\n", tot);
    }
    return 0;
}
