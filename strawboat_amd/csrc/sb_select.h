// strawboat-hip: adaptive codec selection on the device.
//
// Replaces gen_stats + choose_compressor + compress_sample_ratio of the reference
//   integers  src/compression/integer/mod.rs:179-229, 231-308, 310-347
//   floats    src/compression/double/mod.rs:178-229, 231-307, 309-347
//   booleans  src/compression/boolean/mod.rs:151-192, 194-238, 240-278
//   binary    src/compression/binary/mod.rs:265-291, 293-348
// and the per-codec compress_ratio functions (one_value.rs:53-59, freq.rs:129-151,
// dict.rs:109-120, rle.rs:58-60, bp.rs:92-100, delta_bp.rs:97-109, patas.rs:139-141).
//
// The reference walks every value through a HashMap (its dominant encode cost) and draws the
// sample positions from thread_rng().  Here one workgroup per page computes only the statistics
// the still-eligible candidates need:
//   * streaming pass: null count, "all slots equal" (OneValue), typed min/max, sortedness;
//   * exact distinct count with a row-index hash set (LDS first, HBM scratch when it overflows),
//     abandoned as soon as it exceeds N/3 (Dict is then ineligible, dict.rs:111-113);
//   * Boyer-Moore majority candidate + exact count for Freq's ">= 90 % of the rows" test;
//   * the 10 x 64-row sample (seeded: sb::sample_rand, shared with the CPU oracle) staged in
//     LDS and trial-sized for RLE / Bitpacking / DeltaBitpacking / Patas.
// Ratios are computed in f64 with the reference's expressions, so the decision is reproducible
// bit for bit against oracle/ for the same seed.
#pragma once
#include "sb_common.h"

namespace sb {

constexpr uint32_t SAMPLE_COUNT = 10, SAMPLE_SIZE = 64;  // src/compression/mod.rs:30-33
constexpr uint32_t SAMPLE_ROWS = SAMPLE_COUNT * SAMPLE_SIZE;
constexpr uint32_t SAMPLE_CAP = 656;  // LDS capacity: 640 sampled rows, or a whole page of up to 649 rows

// splitmix64 finaliser and sample position, identical to oracle/sbo.h
__host__ __device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ uint64_t sample_rand(uint64_t page_seed, uint32_t depth, uint32_t codec, uint32_t sample_i,
                                                uint64_t n) {
    const uint64_t r = mix64(page_seed ^ mix64(((uint64_t)depth << 40) | ((uint64_t)codec << 32) | sample_i));
    return __umul64hi(r, n);
}
__host__ __device__ __forceinline__ uint64_t page_seed_of(uint64_t column_seed, uint64_t page_index) {
    return mix64(column_seed ^ (page_index * 0xD6E8FEB86659FD93ull));
}

struct SelectOpts {
    double ratio;             // default_compress_ratio (only used when has_ratio)
    uint32_t has_ratio;
    uint32_t forbidden;       // bit per codec id
    uint32_t default_codec;   // CommonCompression id
    int32_t force;            // forced codec or -1
    uint64_t seed;            // page seed
    uint32_t depth;           // 0 = page level, 1 = nested (Dict indices)
};

// type class of a primitive column for ordering / as_i64()
enum NumKind : uint32_t { NK_SIGNED = 0, NK_UNSIGNED = 1, NK_F32 = 2, NK_F64 = 3 };

// canonical key for "distinct" / equality in the statistics: OrderedFloat semantics for floats
// (all NaNs equal, -0 == +0: src/compression/double/traits.rs:51), raw bits otherwise
template <int W>
__device__ __forceinline__ Val<W> stat_key(Val<W> v, uint32_t nk) {
    if constexpr (W == 4) {
        if (nk == NK_F32) {
            const uint32_t a = v.x & 0x7FFFFFFFu;
            if (a > 0x7F800000u) v.x = 0x7FC00000u;
            if (a == 0) v.x = 0;
        }
    }
    if constexpr (W == 8) {
        if (nk == NK_F64) {
            const uint64_t a = v.x & 0x7FFFFFFFFFFFFFFFull;
            if (a > 0x7FF0000000000000ull) v.x = 0x7FF8000000000000ull;
            if (a == 0) v.x = 0;
        }
    }
    return v;
}
template <int W>
__device__ __forceinline__ bool bits_eq(const Val<W>& a, const Val<W>& b) {
    if constexpr (W <= 8) {
        return a.x == b.x;
    } else if constexpr (W == 16) {
        return a.x.x == b.x.x && a.x.y == b.x.y && a.x.z == b.x.z && a.x.w == b.x.w;
    } else {
        return a.x.x == b.x.x && a.x.y == b.x.y && a.x.z == b.x.z && a.x.w == b.x.w && a.y.x == b.y.x &&
               a.y.y == b.y.y && a.y.z == b.y.z && a.y.w == b.y.w;
    }
}
// typed "a < b" for integers (floats are never ordered by the candidates that matter)
template <int W>
__device__ __forceinline__ bool int_lt(const Val<W>& a, const Val<W>& b, uint32_t nk) {
    if constexpr (W == 1) {
        return nk == NK_UNSIGNED ? a.x < b.x : (int8_t)a.x < (int8_t)b.x;
    } else if constexpr (W == 2) {
        return nk == NK_UNSIGNED ? a.x < b.x : (int16_t)a.x < (int16_t)b.x;
    } else if constexpr (W == 4) {
        return nk == NK_UNSIGNED ? a.x < b.x : (int32_t)a.x < (int32_t)b.x;
    } else if constexpr (W == 8) {
        return nk == NK_UNSIGNED ? a.x < b.x : (int64_t)a.x < (int64_t)b.x;
    } else {  // i128 / i256: signed, little-endian limbs
        uint32_t x[W / 4], y[W / 4];
        __builtin_memcpy(x, &a, W);
        __builtin_memcpy(y, &b, W);
        if ((int32_t)x[W / 4 - 1] != (int32_t)y[W / 4 - 1]) return (int32_t)x[W / 4 - 1] < (int32_t)y[W / 4 - 1];
        for (int k = W / 4 - 2; k >= 0; k--)
            if (x[k] != y[k]) return x[k] < y[k];
        return false;
    }
}
// IntegerType::as_i64 (src/compression/integer/traits.rs:5-39): `as i64` casts (wrapping / truncating)
template <int W>
__device__ __forceinline__ int64_t as_i64(const Val<W>& a, uint32_t nk) {
    if constexpr (W == 1) {
        return nk == NK_UNSIGNED ? (int64_t)a.x : (int64_t)(int8_t)a.x;
    } else if constexpr (W == 2) {
        return nk == NK_UNSIGNED ? (int64_t)a.x : (int64_t)(int16_t)a.x;
    } else if constexpr (W == 4) {
        return nk == NK_UNSIGNED ? (int64_t)a.x : (int64_t)(int32_t)a.x;
    } else if constexpr (W == 8) {
        return (int64_t)a.x;
    } else {
        int64_t lo;
        __builtin_memcpy(&lo, &a, 8);
        return lo;
    }
}
template <int W>
__device__ __forceinline__ uint32_t stat_hash(const Val<W>& v) {
    uint64_t w[(W + 7) / 8];
    for (int i = 0; i < (W + 7) / 8; i++) w[i] = 0;
    __builtin_memcpy(w, &v, W);
    uint64_t h = 0x9E3779B97F4A7C15ull;
    for (int i = 0; i < (W + 7) / 8; i++) h = mix64(h ^ w[i]);
    return (uint32_t)(h >> 7);
}
__device__ __forceinline__ uint32_t bits_needed(uint64_t v) { return v ? 64 - __clzll((long long)v) : 0; }

// block-wide reductions over one u32 per thread (s4: 4-entry LDS scratch)
__device__ __forceinline__ uint32_t wg_sum32(uint32_t v, uint32_t* s4) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s4[threadIdx.x >> 6] = v;
    __syncthreads();
    return s4[0] + s4[1] + s4[2] + s4[3];
}
__device__ __forceinline__ uint32_t wg_or32(uint32_t v, uint32_t* s4) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v |= __shfl_down(v, d, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s4[threadIdx.x >> 6] = v;
    __syncthreads();
    return s4[0] | s4[1] | s4[2] | s4[3];
}

constexpr uint32_t SEL_EMPTY = 0xFFFFFFFFu;
constexpr uint32_t SEL_LDS_SLOTS = 8192;  // LDS hash set (row indices), 32 KB

// ---- the sample (or the whole page when N/10 <= 64) staged in LDS ------------------------------
template <int W>
struct Sample {
    Val<W>* val;      // [n] values (null slots zeroed when drawn by sampling)
    uint8_t* valid;   // [n]
    uint32_t n;       // 640, or N when the whole page is used
    bool whole;
};

// rows of the sample: returns false when the whole array is to be used (N/10 <= 64)
__device__ __forceinline__ bool sample_row(uint64_t N, uint64_t seed, uint32_t depth, uint32_t trial, uint32_t k,
                                           uint64_t& row) {
    if (N / SAMPLE_COUNT <= SAMPLE_SIZE) {
        row = k;
        return false;
    }
    const uint64_t sep = N / SAMPLE_COUNT, rem = N % SAMPLE_COUNT;
    const uint32_t s = k / SAMPLE_SIZE, j = k % SAMPLE_SIZE;
    const uint64_t range_end = (s == SAMPLE_COUNT - 1 ? sep + rem : sep) - SAMPLE_SIZE;
    row = s * sep + sample_rand(seed, depth, trial, s, range_end) + j;
    return true;
}

// fill the LDS sample for trial `trial`: the rows of a thread (up to 3) are computed first and their validity bits and
// values requested together, so the gather exposes one memory round trip, not one per row
template <int W, class GetVal, class Valid>
__device__ void load_sample(GetVal getv, Valid valid, uint64_t N, uint64_t seed, uint32_t depth, uint32_t trial,
                            Sample<W>& s) {
    const bool whole = N / SAMPLE_COUNT <= SAMPLE_SIZE;
    s.whole = whole;
    s.n = whole ? (uint32_t)N : SAMPLE_ROWS;
    constexpr int R = (SAMPLE_CAP + WG - 1) / WG;
    uint64_t row[R];
    Val<W> x[R];
    bool v[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        const uint32_t k = threadIdx.x + (uint32_t)r * WG;
        row[r] = 0;
        if (k < s.n) sample_row(N, seed, depth, trial, k, row[r]);
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
        x[r] = getv(row[r]);
        v[r] = valid(row[r]);
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
        const uint32_t k = threadIdx.x + (uint32_t)r * WG;
        if (k >= s.n) continue;
        if (!whole && !v[r]) __builtin_memset(&x[r], 0, sizeof(x[r]));  // MutablePrimitiveArray pushes T::default()
        s.val[k] = x[r];
        s.valid[k] = v[r] ? 1 : 0;
    }
    __syncthreads();
}

// RLE size of the sample: runs * (4 + W) (rle.rs:64-104 on the sample array)
template <int W, int FK>
__device__ uint32_t sample_rle_runs(const Sample<W>& s, uint32_t* s4) {
    // boundary at valid k whose previous valid value differs; runs = boundaries + 1 (n > 0)
    uint32_t cnt = 0;
    for (uint32_t k = threadIdx.x; k < s.n; k += WG) {
        if (!s.valid[k]) continue;
        int p = (int)k - 1;
        while (p >= 0 && !s.valid[p]) p--;
        if (p >= 0 && !rle_eq<W, FK>(s.val[p], s.val[k])) cnt++;
    }
    const uint32_t b = wg_sum32(cnt, s4);
    return s.n ? b + 1 : 0;
}

// The sample rows depend only on (N, seed, trial), not on the data: their (random, latency-bound)
// loads are issued before the streaming pass and committed to LDS when the trial runs, so the
// trials at the end of every page do not each expose an HBM round trip.
template <int W>
struct SamplePre {
    static constexpr int R = (SAMPLE_CAP + WG - 1) / WG;  // sample rows per thread
    Val<W> v[R];
    uint32_t okm;  // bit r: row valid
};
template <int W, class GetVal, class Valid>
__device__ __forceinline__ SamplePre<W> prefetch_sample(GetVal getv, Valid valid, uint64_t N, uint64_t seed, uint32_t depth,
                                                        uint32_t trial) {
    SamplePre<W> p;
    const bool whole = N / SAMPLE_COUNT <= SAMPLE_SIZE;
    const uint32_t n = whole ? (uint32_t)N : SAMPLE_ROWS;
    p.okm = 0;
#pragma unroll
    for (int r = 0; r < SamplePre<W>::R; r++) {
        const uint32_t k = threadIdx.x + (uint32_t)r * WG;
        uint64_t row = 0;
        if (k < n) sample_row(N, seed, depth, trial, k, row);
        p.v[r] = getv(k < n ? row : 0);
        if (k < n && valid(row)) p.okm |= 1u << r;
    }
    return p;
}
template <int W>
__device__ void commit_sample(const SamplePre<W>& p, uint64_t N, Sample<W>& s) {
    const bool whole = N / SAMPLE_COUNT <= SAMPLE_SIZE;
    s.whole = whole;
    s.n = whole ? (uint32_t)N : SAMPLE_ROWS;
#pragma unroll
    for (int r = 0; r < SamplePre<W>::R; r++) {
        const uint32_t k = threadIdx.x + (uint32_t)r * WG;
        if (k >= s.n) continue;
        const bool v = (p.okm >> r) & 1;
        Val<W> x = p.v[r];
        if (!whole && !v) __builtin_memset(&x, 0, sizeof(x));  // MutablePrimitiveArray pushes T::default()
        s.val[k] = x;
        s.valid[k] = v ? 1 : 0;
    }
    __syncthreads();
}

// Bitpacking size of the sample (bp.rs:36-64): sum over 128-blocks of 1 + 16 * bits(OR)
__device__ uint32_t sample_bp_size(const Sample<4>& s, uint32_t* s4, uint32_t* s_blk /* >= 8 words */) {
    const uint32_t nblk = s.n / 128;
    if (threadIdx.x < 8) s_blk[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < nblk * 128; k += WG) atomicOr(&s_blk[k >> 7], s.val[k].x);
    __syncthreads();
    uint32_t size = 0;
    for (uint32_t b = 0; b < nblk; b++) size += 1 + 16 * (s_blk[b] ? 32 - __clz(s_blk[b]) : 0);
    (void)s4;
    return size;
}

// Patas size of the sample (patas.rs:36-104): W + sum of (2 + significant bytes)
template <int W>
__device__ uint32_t sample_patas_size(const Sample<W>& s, uint32_t* s4) {
    static_assert(W == 4 || W == 8, "Patas is for f32 / f64");
    // reference index: the most recent identical bit pattern if it is < 128 back, else i-1; an unseen
    // value refers to index 0 while i < 128 (indices.get().unwrap_or(0), patas.rs:59-65).
    // Most rows repeat their predecessor (d = 1).  The others are taken one at a time by the whole
    // wave: 64 lanes compare the 126 remaining window slots at once (two ballots).
    using B = decltype(s.val[0].x);
    const int lane = threadIdx.x & 63;
    uint32_t bytes = 0;
    for (uint32_t i0 = 0; i0 < s.n; i0 += WG) {  // uniform trip count: wave-wide ops inside
        const uint32_t i = i0 + threadIdx.x;
        const bool act = i < s.n && i > 0;
        const B me = act ? s.val[i].x : (B)0;
        int ref = -1;
        constexpr int NEAR = 6;  // every lane looks at its NEAR predecessors itself (runs, values split by a null slot)
        if (act) {
            B c[NEAR];
#pragma unroll
            for (int q = 0; q < NEAR; q++) c[q] = s.val[i > (uint32_t)q ? i - 1 - q : 0].x;
#pragma unroll
            for (int q = NEAR - 1; q >= 0; q--)
                if ((uint32_t)q < i && c[q] == me) ref = (int)i - 1 - q;
        }
        uint64_t todo = __ballot(act && ref < 0 && i > (uint32_t)NEAR);
        while (todo) {
            const int l = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const uint32_t il = i0 + (threadIdx.x & ~63u) + (uint32_t)l;
            const B val = s.val[il].x;  // same address in every lane: LDS broadcast
            const uint32_t back = il < 127 ? il : 127;
            const uint32_t dA = NEAR + 1 + (uint32_t)lane, dB = NEAR + 65 + (uint32_t)lane;
            const bool mA = dA <= back && s.val[il - dA].x == val;
            const bool mB = dB <= back && s.val[il - dB].x == val;
            const uint64_t bA = __ballot(mA), bB = __ballot(mB);
            const uint32_t d = bA ? NEAR + 1 + (uint32_t)(__ffsll((long long)bA) - 1)
                                  : (bB ? NEAR + 65 + (uint32_t)(__ffsll((long long)bB) - 1) : 0);
            if (lane == l && d) ref = (int)il - (int)d;
        }
        if (!act) continue;
        if (ref < 0) ref = i < 128 ? 0 : (int)i - 1;
        const auto x = s.val[i].x ^ s.val[ref].x;
        uint32_t sig_bits = 0;
        if (x != 0) {
            if constexpr (W == 8)
                sig_bits = 64 - __ffsll((long long)x) + 1 - __clzll((long long)x);
            else
                sig_bits = 32 - __ffs((int)x) + 1 - __clz((int)x);
        }
        bytes += 2 + (sig_bits >> 3) + ((sig_bits & 7) != 0);
    }
    return s.n ? W + wg_sum32(bytes, s4) : 0;
}

}  // namespace sb
