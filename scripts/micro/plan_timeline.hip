// Timeline of k_plan on binary Dict pages (C3 shape: zipf Utf8, 64 Ki-row pages, adaptive writer) through the C API:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I strawboat_amd/csrc scripts/micro/plan_timeline.hip strawboat_amd/csrc/sb_api.hip \
//         strawboat_amd/csrc/sb_encode.hip strawboat_amd/csrc/sb_nested.hip strawboat_amd/csrc/sb_file.cpp strawboat_amd/csrc/sb_schema.cpp \
//         -o scripts/micro/plan_timeline.bin
#define SB_TIMELINE 1
#include "../../strawboat_amd/csrc/sb_decode.hip"
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>

#define CK(x) do { int32_t rc_ = (x); if (rc_ != 0) { printf("%s failed: %d %s\n", #x, rc_, sb_ctx_last_error(ctx)); return 1; } } while (0)

int main(int argc, char** argv) {
    const uint64_t B = (uint64_t)(argc > 1 ? atoi(argv[1]) : 64), ROWS = 1 << 20, PAGE = 65536;
    sb_ctx* ctx = nullptr;
    if (sb_ctx_create(0, nullptr, &ctx) != 0) return 1;
    std::mt19937_64 rng(42);
    std::vector<uint32_t> wl(10000);
    for (auto& l : wl) l = 4 + rng() % 21;
    std::vector<double> cdf(10000);
    double acc = 0;
    for (int k = 1; k <= 10000; k++) { acc += std::pow((double)k, -1.1); cdf[k - 1] = acc; }
    const double total = acc + 10.0 * std::pow(10000.0, -0.1);
    std::uniform_real_distribution<double> U(0, total);
    sb_write_options o;
    std::memset(&o, 0, sizeof o);
    o.max_page_size = PAGE; o.force_codec = -1; o.force_index_codec = -1; o.default_compression = SB_CODEC_LZ4;
    o.has_default_compress_ratio = 1; o.default_compress_ratio = 2.0; o.rng_seed = 42;
    std::vector<sb_column_write> wc(B);
    uint64_t np = 0;
    std::vector<std::vector<sb_page_meta>> metas(B);
    std::vector<uint64_t> vlen(B);
    for (uint64_t b = 0; b < B; b++) {
        std::vector<int32_t> offs(ROWS + 1);
        std::vector<uint32_t> rank(ROWS);
        offs[0] = 0;
        for (uint64_t i = 0; i < ROWS; i++) {
            const double u = U(rng);
            uint32_t k = u >= acc ? 9999u : (uint32_t)(std::lower_bound(cdf.begin(), cdf.end(), u) - cdf.begin());
            rank[i] = k;
            offs[i + 1] = offs[i] + (int32_t)std::max<uint32_t>(wl[k], 1 + (k >= 1000 ? 4 : k >= 100 ? 3 : k >= 10 ? 2 : 1));
        }
        std::vector<uint8_t> vals((size_t)offs[ROWS], (uint8_t)'x');
        for (uint64_t i = 0; i < ROWS; i++) { char w[16]; const int n = snprintf(w, sizeof w, "w%u", rank[i]); memcpy(&vals[offs[i]], w, n); }
        vlen[b] = vals.size();
        const uint64_t bound = sb_write_bound(SB_TYPE_BINARY, 0, ROWS, vals.size(), &o, &np);
        metas[b].resize(np);
        uint8_t *dv, *doff, *dout;
        hipMalloc(&dv, vals.size() + 64); hipMalloc(&doff, (ROWS + 1) * 4); hipMalloc(&dout, bound);
        hipMemcpy(dv, vals.data(), vals.size(), hipMemcpyHostToDevice);
        hipMemcpy(doff, offs.data(), (ROWS + 1) * 4, hipMemcpyHostToDevice);
        sb_column_write& c = wc[b];
        std::memset(&c, 0, sizeof c);
        c.physical_type = SB_TYPE_BINARY; c.is_nullable = 0; c.rows = ROWS; c.values = dv; c.values_len = vals.size(); c.offsets = doff;
        c.out_pages = dout; c.out_capacity = bound; c.out_metas = metas[b].data(); c.n_pages_capacity = np;
    }
    CK(sb_write_columns(ctx, wc.data(), B, &o, SB_MEM_DEVICE));
    CK(sb_ctx_synchronize(ctx));
    std::vector<sb_column_read> rc(B);
    for (uint64_t b = 0; b < B; b++) {
        sb_column_read& c = rc[b];
        std::memset(&c, 0, sizeof c);
        c.physical_type = SB_TYPE_BINARY; c.is_nullable = 0; c.pages = wc[b].out_pages; c.pages_len = wc[b].out_len;
        c.metas = metas[b].data(); c.n_pages = wc[b].n_pages;
        hipMalloc(&c.values, vlen[b] + 64); c.values_capacity = vlen[b] + 64;
        hipMalloc((void**)&c.offsets, (ROWS + 1) * 4); c.offsets_capacity = (ROWS + 1) * 4;
    }
    unsigned long long* tl;
    hipMalloc(&tl, 8 * 64); hipMemset(tl, 0, 8 * 64);
    hipMemcpyToSymbol(HIP_SYMBOL(sb::g_dtl), &tl, sizeof(tl));
    for (int i = 0; i < 3; i++) CK(sb_read_columns(ctx, rc.data(), B, SB_MEM_DEVICE));
    CK(sb_ctx_synchronize(ctx));
    hipStream_t s = (hipStream_t)sb_ctx_stream(ctx);
    hipEvent_t a, e; hipEventCreate(&a); hipEventCreate(&e);
    hipEventRecord(a, s);
    for (int i = 0; i < 5; i++) CK(sb_read_columns(ctx, rc.data(), B, SB_MEM_DEVICE));
    hipEventRecord(e, s);
    CK(sb_ctx_synchronize(ctx));
    float ms; hipEventElapsedTime(&ms, a, e);
    printf("decode of %llu columns (%llu pages each): %.3f ms per call\n", (unsigned long long)B, (unsigned long long)np, ms / 5);
    unsigned long long t[32];
    hipMemcpy(t, tl, 8 * 32, hipMemcpyDeviceToHost);
    const char* names[4] = {"k_plan: index stream (bit-packing header walk) starts", "index stream planned", "dictionary entries walked", "per-tile byte totals"};
    for (int p = 1; p < 4; p++) printf("  %-52s +%8.1f us\n", names[p], (double)(long long)(t[16 + p] - t[16 + p - 1]) / 2400.0);
    return 0;
}
