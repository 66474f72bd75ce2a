// does an LDS-DMA prefetch (global_load_lds_dwordx4: next chunk's rows HBM -> LDS, no VGPRs held across
// the run step) lift k_enc_select_runs<8, 2> off its 0.45-0.47 of the HBM peak?  The kernel is built twice from the
// library's own source — as shipped, and with SB_RUNS_DMA=1 (sb_select_runs.h) at the occupancies the extra 32 KB stage
// allows — and timed on the bench.py shape (4096 pages of 65 536 nullable f64 rows = 8 x the C2 batch's pages per launch).
//   hipcc --offload-arch=gfx950 -O3 -DSB_RUNS_DMA=0                scripts/micro/runs_dma.hip -o /tmp/runs_base
//   hipcc --offload-arch=gfx950 -O3 -DSB_RUNS_DMA=1 -DSB_RUNS_OCC=2 scripts/micro/runs_dma.hip -o /tmp/runs_dma2
// (link line as for runs_timeline.hip: + sb_api.hip sb_decode.hip sb_nested.hip sb_file.cpp sb_schema.cpp)
#include "../../strawboat_amd/csrc/sb_encode.hip"
#include <cstdio>
#include <random>
#define SB_STR2(x) #x
#define SB_STR(x) SB_STR2(x)
using namespace sb;

int main(int argc, char** argv) {
    const uint64_t P = (uint64_t)(argc > 1 ? atoi(argv[1]) : 8192), N = 65536;
    std::vector<uint64_t> h(P * N);
    std::vector<uint8_t> hv(P * N / 8);
    std::mt19937_64 rng(42);
    double cur = 0; int left = 0;
    for (auto& x : h) { if (left == 0) { cur = (double)(rng() % 256); left = 1 + rng() % 64; } memcpy(&x, &cur, 8); left--; }
    for (auto& b : hv) { b = 0; for (int k = 0; k < 8; k++) b |= (rng() % 10 != 0) << k; }
    uint8_t *d, *dv, *scratch; int32_t* codecs;
    const uint64_t SLOT = 16 + 8200 + N * 12 + 64;
    hipMalloc(&d, P * N * 8); hipMalloc(&dv, P * N / 8 + 64); hipMalloc(&codecs, P * 4);
    hipMalloc(&scratch, P * SLOT);
    hipMemcpy(d, h.data(), P * N * 8, hipMemcpyHostToDevice);
    hipMemcpy(dv, hv.data(), P * N / 8, hipMemcpyHostToDevice);
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
    a.forbidden = 0; a.n_pages = P; a.n_cols = 1; a.default_compression = 0; a.freq_count = fc; a.nested_force = -1;
    { uint32_t* cc; hipMalloc(&cc, 128); hipMemset(cc, 0, 128); a.codec_counts = cc; }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) k_enc_select_runs<8, 2><<<P, WG>>>(a);
    hipEventRecord(e0);
    for (int i = 0; i < 10; i++) k_enc_select_runs<8, 2><<<P, WG>>>(a);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<int32_t> hc(P);
    hipMemcpy(hc.data(), codecs, P * 4, hipMemcpyDeviceToHost);
    uint64_t rle = 0, sum = 0;
    std::vector<EncOut> ho(P);
    hipMemcpy(ho.data(), outs, P * sizeof(EncOut), hipMemcpyDeviceToHost);
    for (uint64_t i = 0; i < P; i++) { rle += hc[i] == SB_CODEC_RLE; sum += ho[i].length; }
    const double A = (double)P * (N * 8 + N / 8) + (double)sum;   // algorithmic bytes: Arrow bytes read once + pages written
    printf("SB_RUNS_DMA=%d SB_RUNS_TOUCH=" SB_STR(SB_RUNS_TOUCH) " occupancy %d: %.3f ms per launch of %llu pages -> %.2f TB/s algorithmic = %.3f of 8 TB/s; %llu RLE pages, %llu page bytes\n",
           (int)SB_RUNS_DMA, (int)SB_RUNS_OCC, ms / 10, (unsigned long long)P, A / (ms / 10) / 1e9, A / (ms / 10) / 1e9 / 8.0,
           (unsigned long long)rle, (unsigned long long)sum);
    return 0;
}
