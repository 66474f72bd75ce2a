// Microbenchmark: how fast can "one workgroup per 512 KB page" stream 520 MB out of HBM on MI355X,
// as a function of load width, loads in flight, workgroups per page and barriers per chunk?
// (Calibrates the page encoders: they cannot beat the best number here.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned long long u64;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int VEC /*bytes per lane per load*/, int K /*loads in flight per thread*/, int BAR, int PRE>
__global__ void __launch_bounds__(256) k_stream(const uint8_t* base, u64 page_bytes, int wgs_per_page, u64* out) {
    const int page = blockIdx.x / wgs_per_page, part = blockIdx.x % wgs_per_page;
    const u64 span = page_bytes / wgs_per_page;
    const uint8_t* p = base + (u64)page * page_bytes + (u64)part * span;
    const int t = threadIdx.x;
    constexpr u64 CH = 256ull * K * VEC;
    u64 acc = 0;
    u32x4 cur[K];
    auto fetch = [&](u64 cb, u32x4* dst) {
#pragma unroll
        for (int u = 0; u < K; u++) {
            const uint8_t* q = p + cb + ((u64)u * 256 + t) * VEC;
            if constexpr (VEC == 16) dst[u] = *(const u32x4*)q;
            else if constexpr (VEC == 8) { uint2 v = *(const uint2*)q; dst[u] = u32x4{v.x, v.y, 0, 0}; }
            else { dst[u] = u32x4{*(const uint32_t*)q, 0, 0, 0}; }
        }
    };
    if (PRE) fetch(0, cur);
    for (u64 cb = 0; cb < span; cb += CH) {
        u32x4 v[K];
        if (PRE) {
#pragma unroll
            for (int u = 0; u < K; u++) v[u] = cur[u];
            if (BAR) __builtin_amdgcn_s_barrier();
            if (cb + CH < span) fetch(cb + CH, cur);
        } else {
            fetch(cb, v);
        }
#pragma unroll
        for (int u = 0; u < K; u++) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
        for (int b = 1; b < BAR; b++) __builtin_amdgcn_s_barrier();
    }
    if (acc == 0x123456789ull) out[0] = acc;
}

// thread t reads 128 contiguous bytes (8 x 16 B loads) -- lanes are 128 B apart: no lane coalescing,
// every cache line is consumed by one thread over 8 instructions
__global__ void __launch_bounds__(256) k_stream_rows(const uint8_t* base, u64 page_bytes, u64* out) {
    const uint8_t* p = base + (u64)blockIdx.x * page_bytes;
    const int t = threadIdx.x;
    u64 acc = 0;
    for (u64 cb = 0; cb < page_bytes; cb += 256 * 128) {
        u32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = *(const u32x4*)(p + cb + (u64)t * 128 + u * 16);
#pragma unroll
        for (int u = 0; u < 8; u++) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x123456789ull) out[0] = acc;
}

template <int VEC, int K, int BAR, int PRE>
void run(const char* name, const uint8_t* d, u64 pages, u64 page_bytes, int wpp, u64* out) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; i++) k_stream<VEC, K, BAR, PRE><<<pages * wpp, 256>>>(d, page_bytes, wpp, out);
    hipEventRecord(a);
    for (int i = 0; i < 20; i++) k_stream<VEC, K, BAR, PRE><<<pages * wpp, 256>>>(d, page_bytes, wpp, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 20;
    printf("%-44s wgs/page %d  %.3f ms  %.2f TB/s\n", name, wpp, ms, pages * page_bytes / ms / 1e9);
}

int main() {
    const u64 pages = 1024, page_bytes = 512 * 1024;
    uint8_t* d; u64* out;
    hipMalloc(&d, pages * page_bytes + 4096);
    hipMalloc(&out, 64);
    hipMemset(d, 1, pages * page_bytes);
    {
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        for (int i = 0; i < 3; i++) k_stream_rows<<<pages, 256>>>(d, page_bytes, out);
        hipEventRecord(a);
        for (int i = 0; i < 20; i++) k_stream_rows<<<pages, 256>>>(d, page_bytes, out);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); ms /= 20;
        printf("%-44s wgs/page 1  %.3f ms  %.2f TB/s\n", "thread-contiguous 128 B (8 x 16 B), no LDS", ms, pages * page_bytes / ms / 1e9);
    }
    for (int wpp : {1}) {
        run<8, 16, 0, 1>("8B x16 prefetch, no barrier", d, pages, page_bytes, wpp, out);
        run<8, 16, 4, 1>("8B x16 prefetch, 4 barriers/chunk", d, pages, page_bytes, wpp, out);
        run<16, 8, 0, 1>("16B x8 prefetch, no barrier", d, pages, page_bytes, wpp, out);
        run<16, 8, 4, 1>("16B x8 prefetch, 4 barriers/chunk", d, pages, page_bytes, wpp, out);
        run<16, 4, 0, 1>("16B x4 prefetch, no barrier", d, pages, page_bytes, wpp, out);
        run<8, 8, 0, 0>("8B x8 no prefetch", d, pages, page_bytes, wpp, out);
        run<16, 8, 0, 0>("16B x8 no prefetch", d, pages, page_bytes, wpp, out);
        run<16, 2, 0, 0>("16B x2 no prefetch", d, pages, page_bytes, wpp, out);
    }
    return 0;
}
