// Phase timeline of k_enc_bin_page (sb_bin_page.h) on the C3 shape: s_memtime stamps of workgroup 100, thread 0.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I strawboat_amd/csrc scripts/micro/binpage_timeline.hip \
//         strawboat_amd/csrc/sb_api.hip strawboat_amd/csrc/sb_decode.hip strawboat_amd/csrc/sb_nested.hip \
//         strawboat_amd/csrc/sb_file.cpp strawboat_amd/csrc/sb_schema.cpp -o scripts/micro/bin/binpage_timeline.bin
#define SB_RLE_TIMELINE 1
#include "../../strawboat_amd/csrc/sb_encode.hip"
#include <cmath>
#include <cstdio>
#include <random>
using namespace sb;

int main(int argc, char** argv) {
    const uint64_t P = (uint64_t)(argc > 1 ? atoi(argv[1]) : 1024), N = 65536, R = P * N;
    std::mt19937_64 rng(42);
    std::vector<uint32_t> wl(10000);
    for (auto& l : wl) l = 4 + rng() % 21;
    std::vector<double> cdf(10000);
    double acc = 0;
    for (int k = 1; k <= 10000; k++) { acc += std::pow((double)k, -1.1); cdf[k - 1] = acc; }
    const double total = acc + 10.0 * std::pow(10000.0, -0.1);
    std::vector<int32_t> offs(R + 1);
    std::vector<uint32_t> rank(R);
    offs[0] = 0;
    std::uniform_real_distribution<double> U(0, total);
    for (uint64_t i = 0; i < R; i++) {
        const double u = U(rng);
        uint32_t k = u >= acc ? 9999u : (uint32_t)(std::lower_bound(cdf.begin(), cdf.end(), u) - cdf.begin());
        rank[i] = k;
        offs[i + 1] = offs[i] + (int32_t)std::max<uint32_t>(wl[k], 1 + (k >= 1000 ? 4 : k >= 100 ? 3 : k >= 10 ? 2 : 1));
    }
    std::vector<uint8_t> vals((size_t)offs[R], (uint8_t)'x');
    for (uint64_t i = 0; i < R; i++) {
        char w[16];
        const int n = snprintf(w, sizeof w, "w%u", rank[i]);
        memcpy(&vals[offs[i]], w, n);
    }
    const uint64_t VL = vals.size();
    printf("%llu pages x %llu rows, %.1f MB of values\n", (unsigned long long)P, (unsigned long long)N, VL / 1e6);
    uint8_t *dv, *doff, *scratch;
    int32_t* codecs;
    unsigned long long* tl;
    hipMalloc(&dv, VL + 64); hipMalloc(&doff, (R + 1) * 4); hipMalloc(&codecs, 2 * P * 4); hipMalloc(&tl, 8 * 4096);
    hipMemcpy(dv, vals.data(), VL, hipMemcpyHostToDevice);
    hipMemcpy(doff, offs.data(), (R + 1) * 4, hipMemcpyHostToDevice);
    hipMemset(tl, 0, 8 * 4096);
    hipMemcpyToSymbol(HIP_SYMBOL(g_tl), &tl, sizeof(tl));
    std::vector<EncCol> cols(1);
    std::vector<EncPage> pages(P);
    memset(&cols[0], 0, sizeof(EncCol));
    cols[0].values = dv; cols[0].offsets = doff; cols[0].rows = R; cols[0].ptype = SB_TYPE_BINARY; cols[0].nullable = 0;
    cols[0].width = 4; cols[0].n_pages = P; cols[0].values_len = VL; cols[0].values_len_total = VL; cols[0].nk = NK_SIGNED;
    uint64_t so = 0;
    const uint64_t fixed = (16 + 8200 + 64 + (N + 1) * 8 + 64 + 15) / 16 * 16 + 4096;
    uint64_t M = 64;
    while (M < 2 * N) M <<= 1;
    for (uint64_t i = 0; i < P; i++) {
        memset(&pages[i], 0, sizeof(EncPage));
        pages[i].row0 = i * N; pages[i].rows = N; pages[i].slot_off = so; pages[i].seed = 42 + i; pages[i].codec = CODEC_ON_DEVICE; pages[i].icodec = -1;
        pages[i].zst_off = ~0ull;
        so += fixed;
    }
    so += VL + VL / 64 + 64 * P + 4096;
    for (uint64_t i = 0; i < P; i++) {
        so = (so + 15) & ~15ull;
        pages[i].aux_off = so; pages[i].aux_bytes = (M + 3 * N) * 4; so += pages[i].aux_bytes;
        so = (so + 15) & ~15ull;
        pages[i].h64_off = so; so += N * 8;
    }
    hipMalloc(&scratch, so + 4096);
    EncCol* dc; EncPage* dp; EncOut* outs; Status* st; uint32_t* fc;
    hipMalloc(&dc, sizeof(EncCol)); hipMalloc(&dp, P * sizeof(EncPage)); hipMalloc(&outs, 2 * P * sizeof(EncOut)); hipMalloc(&st, sizeof(Status)); hipMalloc(&fc, 64);
    hipMemcpy(dc, cols.data(), sizeof(EncCol), hipMemcpyHostToDevice);
    hipMemcpy(dp, pages.data(), P * sizeof(EncPage), hipMemcpyHostToDevice);
    hipMemset(st, 0, sizeof(Status)); hipMemset(fc, 0, 64); hipMemset(outs, 0, 2 * P * sizeof(EncOut));
    EncodeArgs a;
    memset(&a, 0, sizeof a);
    a.cols = dc; a.pages = dp; a.outs = outs; a.scratch = scratch; a.status = st; a.codecs = codecs; a.ratio = 2.0; a.has_ratio = 1;
    a.forbidden = 0; a.n_pages = P; a.n_cols = 1; a.default_compression = SB_CODEC_LZ4; a.freq_count = fc; a.nested_force = -1;
    { uint32_t* cc; hipMalloc(&cc, 128); hipMemset(cc, 0, 128); a.codec_counts = cc; }
    a.use_counts = 1; a.bin_fused = 1;
    hipEvent_t e0, e1, e2; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
    for (int i = 0; i < 2; i++) { k_enc_bin_page<int32_t><<<P, BP_WG>>>(a); k_enc_emit_pages<-4, SB_CODEC_DICT><<<P, WG>>>(a); }
    hipEventRecord(e0);
    k_enc_bin_page<int32_t><<<P, BP_WG>>>(a);
    hipEventRecord(e1);
    k_enc_emit_pages<-4, SB_CODEC_DICT><<<P, WG>>>(a);
    hipEventRecord(e2); hipEventSynchronize(e2);
    float m1, m2; hipEventElapsedTime(&m1, e0, e1); hipEventElapsedTime(&m2, e1, e2);
    Status hs; hipMemcpy(&hs, st, sizeof hs, hipMemcpyDeviceToHost);
    int32_t c100; hipMemcpy(&c100, codecs + 100, 4, hipMemcpyDeviceToHost);
    printf("k_enc_bin_page %.3f ms, k_enc_emit_pages<-4, Dict> %.3f ms; codec of page 100: %d; status %d\n", m1, m2, c100, (int)hs.code);
    std::vector<unsigned long long> t(4096);
    hipMemcpy(t.data(), tl, 8 * 4096, hipMemcpyDeviceToHost);
    const char* nm[128] = {};
    nm[60] = "bin_page: start"; nm[61] = "table init"; nm[62] = "row loop (thread 0)"; nm[63] = "barrier"; nm[64] = "nulls"; nm[65] = "sums + vote + count";
    nm[66] = "decision"; nm[67] = "bitmap, prefixes, ids";
    nm[20] = "emit: start"; nm[30] = "index selector"; nm[31] = "index block"; nm[32] = "entries";
    unsigned long long prev = 0;
    nm[69] = "  index array"; nm[70] = "  index codec"; nm[71] = "  bit-packed body"; nm[72] = "  entries";
    nm[68] = "page finished";
    for (int p : {60, 61, 62, 63, 64, 65, 66, 67, 69, 70, 71, 72, 68, 20, 30, 31, 32}) {
        const unsigned long long v = t[512 + p];
        if (!v) continue;
        if (p == 60 || p == 20) prev = v;
        printf("  %-26s +%8.1f us\n", nm[p], (double)(long long)(v - prev) / 100.0);
        prev = v;
    }
    printf("  row loop steps (us):");
    for (int k = 0; k < 16; k++) printf(" %.1f", (double)(long long)(t[512 + 81 + k] ? t[512 + 81 + k] - t[512 + 80 + k] : 0) / 100.0);
    printf("\n");
    return 0;
}
