// Phase timeline of the Int32 adaptive path on the C4 shape (values uniform in [0, 1000)): k_enc_select_runs / _rle<4, 0>
// and k_enc_emit_pages<4, Dict> (LDS dictionary build, index selector, index block, entries).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I strawboat_amd/csrc scripts/micro/primdict_timeline.hip \
//         strawboat_amd/csrc/sb_api.hip strawboat_amd/csrc/sb_decode.hip strawboat_amd/csrc/sb_nested.hip \
//         strawboat_amd/csrc/sb_file.cpp strawboat_amd/csrc/sb_schema.cpp -o scripts/micro/primdict_timeline.bin
#define SB_RLE_TIMELINE 1
#include "../../strawboat_amd/csrc/sb_encode.hip"
#include <cstdio>
#include <random>
using namespace sb;

int main(int argc, char** argv) {
    const uint64_t P = (uint64_t)(argc > 1 ? atoi(argv[1]) : 2048), N = 65536, R = P * N;
    std::mt19937_64 rng(42);
    std::vector<int32_t> vals(R);
    for (auto& v : vals) v = (int32_t)(rng() % 1000);
    uint8_t *dv, *scratch;
    int32_t* codecs;
    unsigned long long* tl;
    hipMalloc(&dv, R * 4 + 64); hipMalloc(&codecs, 2 * P * 4); hipMalloc(&tl, 8 * 4096);
    hipMemcpy(dv, vals.data(), R * 4, hipMemcpyHostToDevice);
    hipMemset(tl, 0, 8 * 4096);
    hipMemcpyToSymbol(HIP_SYMBOL(g_tl), &tl, sizeof(tl));
    std::vector<EncCol> cols(1);
    std::vector<EncPage> pages(P);
    memset(&cols[0], 0, sizeof(EncCol));
    cols[0].values = dv; cols[0].rows = R; cols[0].ptype = SB_TYPE_INT32; cols[0].nullable = 0;
    cols[0].width = 4; cols[0].n_pages = P; cols[0].nk = NK_SIGNED;
    uint64_t so = 0;
    uint64_t M = 64;
    while (M < 2 * N) M <<= 1;
    for (uint64_t i = 0; i < P; i++) {
        memset(&pages[i], 0, sizeof(EncPage));
        pages[i].row0 = i * N; pages[i].rows = N; pages[i].slot_off = so; pages[i].seed = 42 + i; pages[i].codec = CODEC_ON_DEVICE; pages[i].icodec = -1;
        pages[i].zst_off = ~0ull; pages[i].h64_off = ~0ull;
        so += (64 + 9 + 9 + N * 12 + 4 + 64 + 15) / 16 * 16;
    }
    for (uint64_t i = 0; i < P; i++) {
        so = (so + 15) & ~15ull;
        pages[i].aux_off = so; pages[i].aux_bytes = (M + 3 * N) * 4; so += pages[i].aux_bytes;
    }
    hipMalloc(&scratch, so + 4096);
    EncCol* dc; EncPage* dp; EncOut* outs; Status* st; uint32_t* fc; uint64_t* results;
    hipMalloc(&dc, sizeof(EncCol)); hipMalloc(&dp, P * sizeof(EncPage)); hipMalloc(&outs, 2 * P * sizeof(EncOut)); hipMalloc(&st, sizeof(Status)); hipMalloc(&fc, 64);
    hipMalloc(&results, (2 * P + 8) * 8);
    hipMemcpy(dc, cols.data(), sizeof(EncCol), hipMemcpyHostToDevice);
    hipMemcpy(dp, pages.data(), P * sizeof(EncPage), hipMemcpyHostToDevice);
    hipMemset(st, 0, sizeof(Status)); hipMemset(fc, 0, 64); hipMemset(outs, 0, 2 * P * sizeof(EncOut));
    EncodeArgs a;
    memset(&a, 0, sizeof a);
    a.cols = dc; a.pages = dp; a.outs = outs; a.scratch = scratch; a.status = st; a.codecs = codecs; a.ratio = 2.0; a.has_ratio = 1;
    a.results = results;
    a.forbidden = 0; a.n_pages = P; a.n_cols = 1; a.default_compression = SB_CODEC_LZ4; a.freq_count = fc; a.nested_force = -1;
    { uint32_t* cc; hipMalloc(&cc, 128); hipMemset(cc, 0, 128); a.codec_counts = cc; }
    hipEvent_t e0, e1, e2, e3; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2); hipEventCreate(&e3);
    auto sel = [&]() { hipMemsetAsync(codecs, 0, 2 * P * 4); hipMemsetAsync(outs, 0, 2 * P * sizeof(EncOut)); k_enc_select_runs<4, 0><<<P, WG>>>(a); };
    for (int i = 0; i < 2; i++) { sel(); k_enc_select_rle<4, 0><<<P, WG>>>(a); k_enc_emit_pages<4, SB_CODEC_DICT><<<P, WG>>>(a); }
    hipMemsetAsync(codecs, 0, 2 * P * 4); hipMemsetAsync(outs, 0, 2 * P * sizeof(EncOut));
    hipEventRecord(e0);
    k_enc_select_runs<4, 0><<<P, WG>>>(a);
    hipEventRecord(e1);
    k_enc_select_rle<4, 0><<<P, WG>>>(a);
    hipEventRecord(e2);
    k_enc_emit_pages<4, SB_CODEC_DICT><<<P, WG>>>(a);
    hipEventRecord(e3); hipEventSynchronize(e3);
    float m1, m2, m3; hipEventElapsedTime(&m1, e0, e1); hipEventElapsedTime(&m2, e1, e2); hipEventElapsedTime(&m3, e2, e3);
    Status hs; hipMemcpy(&hs, st, sizeof hs, hipMemcpyDeviceToHost);
    int32_t c100; hipMemcpy(&c100, codecs + 100, 4, hipMemcpyDeviceToHost);
    printf("%llu pages: select_runs %.3f ms, select_rle %.3f ms, emit_pages<4, Dict> %.3f ms; codec of page 100: %d; status %d\n",
           (unsigned long long)P, m1, m2, m3, c100, (int)hs.code);
    std::vector<unsigned long long> t(4096);
    hipMemcpy(t.data(), tl, 8 * 4096, hipMemcpyDeviceToHost);
    const char* nm[64] = {};
    nm[50] = "emit: start"; nm[51] = "dictionary build"; nm[52] = "index selector (choose_prim<4>)"; nm[53] = "index block"; nm[54] = "entries";
    nm[0] = "select: start"; nm[1] = "select: end";
    unsigned long long prev = 0;
    for (int p : {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 50, 51, 52, 53, 54}) {
        const unsigned long long v = t[512 + p];
        if (!v) continue;
        if (p == 0 || p == 50) prev = v;
        printf("  [%2d] %-34s +%8.1f us\n", p, nm[p] ? nm[p] : "", (double)(long long)(v - prev) / 100.0);
        prev = v;
    }
    return 0;
}
