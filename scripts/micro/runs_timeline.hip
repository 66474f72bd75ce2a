// Timeline of the run-level fused selection + RLE kernel (select_runs_page<8,2>) on the bench.py shape.
#define SB_RLE_TIMELINE 1
#include "../../strawboat_amd/csrc/sb_encode.hip"  // link with sb_api.hip sb_decode.hip sb_nested.hip sb_file.cpp
#include <cstdio>
#include <random>
using namespace sb;

int main(int argc, char** argv) {
    const uint64_t P = (uint64_t)(argc > 1 ? atoi(argv[1]) : 4096), N = 65536;
    std::vector<uint64_t> h(P * N);
    std::vector<uint8_t> hv(P * N / 8);
    std::mt19937_64 rng(42);
    double cur = 0; int left = 0;
    for (auto& x : h) { if (left == 0) { cur = (double)(rng() % 256); left = 1 + rng() % 64; } memcpy(&x, &cur, 8); left--; }
    for (auto& b : hv) { b = 0; for (int k = 0; k < 8; k++) b |= (rng() % 10 != 0) << k; }
    uint8_t *d, *dv, *scratch; int32_t* codecs; unsigned long long* tl;
    const uint64_t SLOT = 16 + 8200 + N * 12 + 64;
    hipMalloc(&d, P * N * 8); hipMalloc(&dv, P * N / 8 + 64); hipMalloc(&codecs, P * 4); hipMalloc(&tl, 8 * 4096);
    hipMalloc(&scratch, P * SLOT);
    hipMemcpy(d, h.data(), P * N * 8, hipMemcpyHostToDevice);
    hipMemcpy(dv, hv.data(), P * N / 8, hipMemcpyHostToDevice);
    hipMemset(tl, 0, 8 * 4096);
    hipMemcpyToSymbol(HIP_SYMBOL(g_tl), &tl, sizeof(tl));
    std::vector<EncCol> cols(1);
    std::vector<EncPage> pages(P);
    memset(&cols[0], 0, sizeof(EncCol));
    cols[0].values = d; cols[0].validity = dv; cols[0].rows = P * N; cols[0].ptype = SB_TYPE_FLOAT64; cols[0].nullable = 1;
    cols[0].width = 8; cols[0].n_pages = P; cols[0].fkind = 2; cols[0].nk = NK_F64;
    for (uint64_t i = 0; i < P; i++) {
        memset(&pages[i], 0, sizeof(EncPage));
        pages[i].row0 = i * N; pages[i].rows = N; pages[i].slot_off = i * SLOT; pages[i].seed = 42 + i; pages[i].codec = CODEC_ON_DEVICE; pages[i].icodec = -1;
    }
    EncCol* dc; EncPage* dp; EncOut* outs; Status* st; uint32_t* fc;
    hipMalloc(&dc, sizeof(EncCol)); hipMalloc(&dp, P * sizeof(EncPage)); hipMalloc(&outs, 2 * P * sizeof(EncOut)); hipMalloc(&st, sizeof(Status)); hipMalloc(&fc, 64);
    hipMemcpy(dc, cols.data(), sizeof(EncCol), hipMemcpyHostToDevice);
    hipMemcpy(dp, pages.data(), P * sizeof(EncPage), hipMemcpyHostToDevice);
    hipMemset(st, 0, sizeof(Status)); hipMemset(fc, 0, 64); hipMemset(outs, 0, 2 * P * sizeof(EncOut));
    EncodeArgs a;
    memset(&a, 0, sizeof a);
    a.cols = dc; a.pages = dp; a.outs = outs; a.scratch = scratch; a.status = st; a.codecs = codecs; a.ratio = 2.0; a.has_ratio = 1;
    a.forbidden = 0; a.n_pages = P; a.n_cols = 1; a.default_compression = 0; a.freq_count = fc; a.nested_force = -1; { uint32_t* cc; hipMalloc(&cc, 128); hipMemset(cc, 0, 128); a.codec_counts = cc; }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) k_enc_select_runs<8, 2><<<P, WG>>>(a);
    hipEventRecord(e0);
    for (int i = 0; i < 10; i++) k_enc_select_runs<8, 2><<<P, WG>>>(a);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    int32_t c0; hipMemcpy(&c0, codecs + 100, 4, hipMemcpyDeviceToHost);
    printf("k_enc_select_runs: %.3f ms per launch (%.2f TB/s of values), page 100 -> codec %d\n", ms / 10, P * N * 8 / (ms / 10) / 1e9, c0);
    std::vector<unsigned long long> t(4096);
    hipMemcpy(t.data(), tl, 8 * 4096, hipMemcpyDeviceToHost);
    printf("  page loop %llu ticks, decide %llu ticks (100 MHz ticks: x10 ns)\n", t[512 + 1] - t[512 + 0], t[512 + 10] - t[512 + 1]);
    {
        const char* sn[11] = {"loop start", "loop end", "reduce", "OneValue", "Freq", "Dict", "Patas", "RLE", "-", "-", "end"};
        for (int p = 2; p <= 10; p++) {
            if (!t[512 + p]) continue;
            int q = p - 1;
            while (q > 1 && !t[512 + q]) q--;
            printf("    decide: %-10s +%llu\n", sn[p], t[512 + p] - t[512 + q]);
        }
    }
    const char* nm[11] = {"top", "loads+compare", "barrier A", "run list", "barrier B", "stats", "fv+ballots", "barrier 1", "carries", "barrier 2", "records"};
    for (int p = 1; p <= 10; p++) printf("  %-14s +%llu\n", nm[p], t[600 + p] - t[600 + p - 1]);
    return 0;
}
