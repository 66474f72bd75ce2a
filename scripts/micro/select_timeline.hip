// Timeline of the adaptive selector (choose_prim<8>) on the bench.py shape: 1024 pages x 64 Ki f64.
#define SB_RLE_TIMELINE 1
#include "../../strawboat_amd/csrc/sb_encode.hip"  // link with sb_api.hip sb_decode.hip sb_nested.hip sb_file.cpp
#include <cstdio>
#include <random>
using namespace sb;

__global__ void __launch_bounds__(WG) k_sel(const uint8_t* vals, const uint8_t* valid, uint64_t N, int32_t* codecs, uint32_t forb) {
    __shared__ uint32_t lds_tab[SEL_LDS_SLOTS];
    __shared__ uint32_t s_misc[2 * WG + 16];
    __shared__ __attribute__((aligned(16))) uint8_t sample_mem[SAMPLE_CAP * 9 + 16];
    const uint8_t* v = vals + (uint64_t)blockIdx.x * N * 8;
    ValidView vv{valid, (uint64_t)blockIdx.x * N};
    SelectOpts so{2.0, 1u, forb, 0u, -1, mix64(42 + blockIdx.x), 0u};
    SelScratch sc{lds_tab, s_misc, sample_mem, nullptr, 0};
    const uint32_t c = choose_prim<8>([=](uint64_t i) { return ld_val<8>(v + i * 8); }, vv, N, NK_F64, so, sc);
    if (threadIdx.x == 0) codecs[blockIdx.x] = (int32_t)c;
}

int main() {
    const uint64_t P = 1024, N = 65536;
    std::vector<uint64_t> h(P * N);
    std::vector<uint8_t> hv(P * N / 8);
    std::mt19937_64 rng(42);
    double cur = 0; int left = 0;
    for (auto& x : h) { if (left == 0) { cur = (double)(rng() % 256); left = 1 + rng() % 64; } memcpy(&x, &cur, 8); left--; }
    for (auto& b : hv) { b = 0; for (int k = 0; k < 8; k++) b |= (rng() % 10 != 0) << k; }
    uint8_t *d, *dv; int32_t* codecs; unsigned long long* tl;
    hipMalloc(&d, P * N * 8); hipMalloc(&dv, P * N / 8 + 64); hipMalloc(&codecs, P * 4); hipMalloc(&tl, 8 * 4096);
    hipMemcpy(d, h.data(), P * N * 8, hipMemcpyHostToDevice);
    hipMemcpy(dv, hv.data(), P * N / 8, hipMemcpyHostToDevice);
    hipMemset(tl, 0, 8 * 4096);
    hipMemcpyToSymbol(HIP_SYMBOL(g_tl), &tl, sizeof(tl));
    const uint32_t forb = 0;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; i++) k_sel<<<P, WG>>>(d, dv, N, codecs, forb);
    hipEventRecord(a);
    for (int i = 0; i < 10; i++) k_sel<<<P, WG>>>(d, dv, N, codecs, forb);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    int32_t c0; hipMemcpy(&c0, codecs + 100, 4, hipMemcpyDeviceToHost);
    printf("k_sel: %.3f ms per launch (%.2f TB/s), page 100 -> codec %d\n", ms / 10, P * N * 8 / (ms / 10) / 1e9, c0);
    std::vector<unsigned long long> t(4096);
    hipMemcpy(t.data(), tl, 8 * 4096, hipMemcpyDeviceToHost);
    const char* names[11] = {"start", "stream", "reduce", "OneValue", "Freq", "Dict", "Patas", "RLE", "-", "-", "end"};
    for (int p = 1; p <= 10; p++) {
        if (!t[512 + p]) continue;
        int q = p - 1; while (q > 0 && !t[512 + q]) q--;
        printf("  %-10s +%llu ticks\n", names[p], t[512 + p] - t[512 + q]);
    }
    return 0;
}
