// Timeline of k_expand on RLE pages (bench.py shape, 64 columns) through the library's own C API.
#define SB_TIMELINE 1
#include "../../strawboat_amd/csrc/sb_decode.hip"  // link with sb_api.hip sb_encode.hip sb_nested.hip sb_file.cpp
#include <cstdio>
#include <random>
#include <cstring>

#define CK(x) do { int32_t rc_ = (x); if (rc_ != 0) { printf("%s failed: %d %s\n", #x, rc_, sb_ctx_last_error(ctx)); return 1; } } while (0)

int main() {
    const uint64_t B = 64, ROWS = 1000000, PAGE = 65536;
    sb_ctx* ctx = nullptr;
    if (sb_ctx_create(0, nullptr, &ctx) != 0) return 1;
    std::mt19937_64 rng(42);
    std::vector<double> h(ROWS);
    std::vector<uint8_t> hv((ROWS + 7) / 8);
    sb_write_options o;
    std::memset(&o, 0, sizeof o);
    o.max_page_size = PAGE;
    o.force_codec = SB_CODEC_RLE;
    o.force_index_codec = -1;
    uint64_t np = 0;
    const uint64_t bound = sb_write_bound(SB_TYPE_FLOAT64, 1, ROWS, 0, &o, &np);
    std::vector<sb_column_write> wc(B);
    std::vector<std::vector<sb_page_meta>> metas(B, std::vector<sb_page_meta>(np));
    for (uint64_t b = 0; b < B; b++) {
        double cur = 0; int left = 0;
        for (auto& x : h) { if (left == 0) { cur = (double)(rng() % 256); left = 1 + rng() % 64; } x = cur; left--; }
        for (auto& q : hv) { q = 0; for (int k = 0; k < 8; k++) q |= (rng() % 10 != 0) << k; }
        uint8_t *dv, *dm, *dout;
        hipMalloc(&dv, ROWS * 8); hipMalloc(&dm, hv.size() + 64); hipMalloc(&dout, bound);
        hipMemcpy(dv, h.data(), ROWS * 8, hipMemcpyHostToDevice);
        hipMemcpy(dm, hv.data(), hv.size(), hipMemcpyHostToDevice);
        sb_column_write& c = wc[b];
        std::memset(&c, 0, sizeof c);
        c.physical_type = SB_TYPE_FLOAT64; c.is_nullable = 1; c.rows = ROWS; c.values = dv; c.validity = dm;
        c.out_pages = dout; c.out_capacity = bound; c.out_metas = metas[b].data(); c.n_pages_capacity = np;
    }
    CK(sb_write_columns(ctx, wc.data(), B, &o, SB_MEM_DEVICE));
    CK(sb_ctx_synchronize(ctx));
    std::vector<sb_column_read> rc(B);
    for (uint64_t b = 0; b < B; b++) {
        sb_column_read& c = rc[b];
        std::memset(&c, 0, sizeof c);
        c.physical_type = SB_TYPE_FLOAT64; c.is_nullable = 1; c.pages = wc[b].out_pages; c.pages_len = wc[b].out_len;
        c.metas = metas[b].data(); c.n_pages = wc[b].n_pages;
        hipMalloc(&c.values, ROWS * 8); c.values_capacity = ROWS * 8;
        hipMalloc((void**)&c.validity, (ROWS + 31) / 32 * 4); c.validity_capacity = (ROWS + 31) / 32 * 4;
    }
    unsigned long long* tl;
    hipMalloc(&tl, 8 * 64); hipMemset(tl, 0, 8 * 64);
    hipMemcpyToSymbol(HIP_SYMBOL(sb::g_dtl), &tl, sizeof(tl));
    for (int i = 0; i < 3; i++) CK(sb_read_columns(ctx, rc.data(), B, SB_MEM_DEVICE));
    CK(sb_ctx_synchronize(ctx));
    hipStream_t s = (hipStream_t)sb_ctx_stream(ctx);
    hipEvent_t a, e; hipEventCreate(&a); hipEventCreate(&e);
    hipEventRecord(a, s);
    for (int i = 0; i < 10; i++) CK(sb_read_columns(ctx, rc.data(), B, SB_MEM_DEVICE));
    hipEventRecord(e, s);
    CK(sb_ctx_synchronize(ctx));
    float ms; hipEventElapsedTime(&ms, a, e);
    printf("decode of %llu columns: %.3f ms per call\n", (unsigned long long)B, ms / 10);
    unsigned long long t[8];
    hipMemcpy(t, tl, 64, hipMemcpyDeviceToHost);
    const char* names[5] = {"start", "descriptors", "validity bits", "run index (zero/scatter/scan)", "gather + store"};
    for (int p = 1; p < 5; p++) printf("  %-32s +%llu ticks\n", names[p], t[p] - t[p - 1]);
    return 0;
}
