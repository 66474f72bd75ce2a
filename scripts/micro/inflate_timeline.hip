// Phase times of k_inflate's batch mode (wave 0) on Zstd pages of an increasing Int32 column (what string offsets look
// like: literals-only frames) through the C API:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I strawboat_amd/csrc scripts/micro/inflate_timeline.hip strawboat_amd/csrc/sb_api.hip \
//         strawboat_amd/csrc/sb_encode.hip strawboat_amd/csrc/sb_nested.hip strawboat_amd/csrc/sb_file.cpp strawboat_amd/csrc/sb_schema.cpp \
//         -o scripts/micro/inflate_timeline.bin
//   scripts/micro/inflate_timeline.bin [columns = 128] [kind: 0 = offsets-like Int32, 1 = random-walk Int64]
#define SB_TIMELINE 1
#include <hip/hip_runtime.h>
__device__ unsigned long long* g_ztl;
#define ZTL_BEGIN unsigned long long ztl_t = __builtin_readcyclecounter();
#define ZTL(p)                                                                                             \
    do {                                                                                                   \
        const unsigned long long n_ = __builtin_readcyclecounter();                                        \
        if (g_ztl && blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) g_ztl[48 + (p)] += n_ - ztl_t;       \
        ztl_t = n_;                                                                                        \
    } while (0)
#include "../../strawboat_amd/csrc/sb_decode.hip"
#include <cstdio>
#include <cstring>
#include <random>

#define CK(x) do { int32_t rc_ = (x); if (rc_ != 0) { printf("%s failed: %d %s\n", #x, rc_, sb_ctx_last_error(ctx)); return 1; } } while (0)

int main(int argc, char** argv) {
    const uint64_t B = (uint64_t)(argc > 1 ? atoi(argv[1]) : 128), ROWS = 1 << 20, PAGE = 65536;
    const int kind = argc > 2 ? atoi(argv[2]) : 0;
    const uint32_t W = kind ? 8 : 4;
    sb_ctx* ctx = nullptr;
    if (sb_ctx_create(0, nullptr, &ctx) != 0) return 1;
    std::mt19937_64 rng(42);
    sb_write_options o;
    std::memset(&o, 0, sizeof o);
    o.max_page_size = PAGE; o.force_codec = -1; o.force_index_codec = -1; o.default_compression = SB_CODEC_ZSTD; o.rng_seed = 42;
    std::vector<sb_column_write> wc(B);
    uint64_t np = 0;
    std::vector<std::vector<sb_page_meta>> metas(B);
    for (uint64_t b = 0; b < B; b++) {
        std::vector<uint8_t> vals(ROWS * W);
        int64_t acc = 0;
        for (uint64_t i = 0; i < ROWS; i++) {
            acc += kind ? (int64_t)(rng() % 2000001) - 1000000 : (int64_t)(2 + rng() % 3);
            std::memcpy(&vals[i * W], &acc, W);
        }
        const uint64_t bound = sb_write_bound(kind ? SB_TYPE_INT64 : SB_TYPE_INT32, 0, ROWS, 0, &o, &np);
        metas[b].resize(np);
        uint8_t *dv, *dout;
        hipMalloc(&dv, vals.size() + 64); hipMalloc(&dout, bound);
        hipMemcpy(dv, vals.data(), vals.size(), hipMemcpyHostToDevice);
        sb_column_write& c = wc[b];
        std::memset(&c, 0, sizeof c);
        c.physical_type = kind ? SB_TYPE_INT64 : SB_TYPE_INT32; c.is_nullable = 0; c.rows = ROWS; c.values = dv; c.values_len = vals.size();
        c.out_pages = dout; c.out_capacity = bound; c.out_metas = metas[b].data(); c.n_pages_capacity = np;
    }
    CK(sb_write_columns(ctx, wc.data(), B, &o, SB_MEM_DEVICE));
    CK(sb_ctx_synchronize(ctx));
    std::vector<sb_column_read> rc(B);
    uint64_t pbytes = 0;
    for (uint64_t b = 0; b < B; b++) {
        sb_column_read& c = rc[b];
        std::memset(&c, 0, sizeof c);
        c.physical_type = wc[b].physical_type; c.is_nullable = 0; c.pages = wc[b].out_pages; c.pages_len = wc[b].out_len;
        c.metas = metas[b].data(); c.n_pages = wc[b].n_pages;
        hipMalloc(&c.values, ROWS * W + 64); c.values_capacity = ROWS * W + 64;
        pbytes += wc[b].out_len;
    }
    unsigned long long* tl;
    hipMalloc(&tl, 8 * 64); hipMemset(tl, 0, 8 * 64);
    hipMemcpyToSymbol(HIP_SYMBOL(sb::g_dtl), &tl, sizeof(tl));
    hipMemcpyToSymbol(HIP_SYMBOL(g_ztl), &tl, sizeof(tl));
    for (int i = 0; i < 2; i++) CK(sb_read_columns(ctx, rc.data(), B, SB_MEM_DEVICE));
    CK(sb_ctx_synchronize(ctx));
    hipMemset(tl, 0, 8 * 64);
    hipStream_t s = (hipStream_t)sb_ctx_stream(ctx);
    hipEvent_t a, e; hipEventCreate(&a); hipEventCreate(&e);
    hipEventRecord(a, s);
    CK(sb_read_columns(ctx, rc.data(), B, SB_MEM_DEVICE));
    hipEventRecord(e, s);
    CK(sb_ctx_synchronize(ctx));
    float ms; hipEventElapsedTime(&ms, a, e);
    printf("%llu columns x %llu rows of %u bytes: %.1f MB of pages, decode %.3f ms\n", (unsigned long long)B, (unsigned long long)ROWS, W, pbytes / 1e6, ms);
    unsigned long long t[64];
    hipMemcpy(t, tl, 8 * 64, hipMemcpyDeviceToHost);
    const char* names[7] = {"headers (lane per frame)", "group setup", "Huffman tables (lane per frame)", "streams (lane per stream)", "end of phase H",
                            "sequence pre-decode (lane per frame)", "one-wave path"};
    for (int p = 0; p < 7; p++) printf("  %-40s %10.1f us\n", names[p], (double)t[32 + p] / 2400.0);
    const char* zn[4] = {"streams: round setup", "streams: issue the prefetch", "streams: decode 32 symbols", "streams: ring writes + store"};
    for (int p = 0; p < 4; p++) printf("  %-40s %10.1f us\n", zn[p], (double)t[48 + p] / 2400.0);
    return 0;
}
