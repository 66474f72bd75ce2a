// ORACLE — TEST INFRASTRUCTURE ONLY (see sbo.h).  C ABI for ctypes (oracle/sbo.py).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <mutex>
#include <string>
#include <thread>
#include <cstdio>
#include <cstring>

#include "sbo.h"

using namespace sbo;

extern "C" {

struct sbo_column_in {
    int32_t ptype;
    int32_t nullable;
    uint64_t rows;
    const uint8_t* values;
    uint64_t values_bit_offset;
    uint64_t values_len;
    const uint8_t* validity;
    uint64_t validity_bit_offset;
    const uint8_t* offsets;
};

struct sbo_options {
    uint8_t default_compression;
    uint8_t has_ratio;
    uint8_t pad_[6];
    double ratio;
    uint64_t max_page_size;
    uint32_t forbidden_mask;
    int32_t force_codec;
    int32_t force_index_codec;
    int32_t pad2_;
    uint64_t rng_seed;
    uint64_t page_index0;
};

struct sbo_written {
    std::vector<uint8_t> bytes;
    std::vector<PageMeta> metas;
};

static void set_err(char* err, size_t cap, const char* m) {
    if (err && cap) {
        snprintf(err, cap, "%s", m);
    }
}

static ColumnIn to_col(const sbo_column_in* c) {
    ColumnIn col;
    col.ptype = c->ptype;
    col.nullable = c->nullable != 0;
    col.rows = c->rows;
    col.values = c->values;
    col.values_bit_offset = c->values_bit_offset;
    col.values_len = c->values_len;
    col.validity = c->validity;
    col.validity_bit_offset = c->validity_bit_offset;
    col.offsets = c->offsets;
    return col;
}
static WriteOptions to_opts(const sbo_options* o) {
    WriteOptions w;
    w.default_compression = o->default_compression;
    w.has_ratio = o->has_ratio != 0;
    w.ratio = o->ratio;
    w.max_page_size = o->max_page_size;
    w.forbidden_mask = o->forbidden_mask;
    w.force_codec = o->force_codec;
    w.force_index_codec = o->force_index_codec;
    w.rng_seed = o->rng_seed;
    w.page_index0 = o->page_index0;
    return w;
}

void* sbo_write_column(const sbo_column_in* c, const sbo_options* o, char* err, size_t errcap) {
    sbo_written* w = new sbo_written();
    try {
        write_column(to_col(c), to_opts(o), w->bytes, w->metas);
        return w;
    } catch (const std::exception& e) {
        set_err(err, errcap, e.what());
        delete w;
        return nullptr;
    }
}
// One page through write::write_simple, rows may be 0: the leaf BLOCK of a nested page whose lists are all
// empty / null.  write_nested (write/serialize.rs:134-198) still runs compress_* on the empty leaf slice:
// every Extend codec's ratio is 0 or NaN on empty statistics, so the block is Basic(default_compression)
// over zero bytes (hdr9 + the codec's encoding of an empty input).  A forced codec (this build's knob,
// absent upstream) is not applied to an empty page.
void* sbo_write_page(const sbo_column_in* c, const sbo_options* o, char* err, size_t errcap) {
    sbo_written* w = new sbo_written();
    try {
        ColumnIn col = to_col(c);
        WriteOptions wo = to_opts(o);
        if (col.rows == 0) wo.force_codec = -1;
        write_page(col, wo, w->bytes);
        w->metas.push_back(PageMeta{(uint64_t)w->bytes.size(), col.rows});
        return w;
    } catch (const std::exception& e) {
        set_err(err, errcap, e.what());
        delete w;
        return nullptr;
    }
}
uint64_t sbo_written_len(void* h) { return ((sbo_written*)h)->bytes.size(); }
const uint8_t* sbo_written_data(void* h) { return ((sbo_written*)h)->bytes.data(); }
uint64_t sbo_written_npages(void* h) { return ((sbo_written*)h)->metas.size(); }
const PageMeta* sbo_written_metas(void* h) { return ((sbo_written*)h)->metas.data(); }
void sbo_written_free(void* h) { delete (sbo_written*)h; }

void* sbo_read_column(int32_t ptype, int32_t nullable, const uint8_t* pages, uint64_t pages_len,
                      const PageMeta* metas, uint64_t n_pages, char* err, size_t errcap) {
    ColumnOut* out = new ColumnOut();
    try {
        read_column(ptype, nullable != 0, pages, pages_len, metas, n_pages, *out);
        return out;
    } catch (const std::exception& e) {
        set_err(err, errcap, e.what());
        delete out;
        return nullptr;
    }
}
uint64_t sbo_read_rows(void* h) { return ((ColumnOut*)h)->rows; }
uint64_t sbo_read_values_len(void* h) { return ((ColumnOut*)h)->values.size(); }
const uint8_t* sbo_read_values(void* h) { return ((ColumnOut*)h)->values.data(); }
uint64_t sbo_read_validity_len(void* h) { return ((ColumnOut*)h)->validity.size(); }
const uint8_t* sbo_read_validity(void* h) { return ((ColumnOut*)h)->validity.data(); }
uint64_t sbo_read_offsets_len(void* h) { return ((ColumnOut*)h)->offsets.size(); }
const uint8_t* sbo_read_offsets(void* h) { return ((ColumnOut*)h)->offsets.data(); }
void sbo_read_free(void* h) { delete (ColumnOut*)h; }

int32_t sbo_stat_column(int32_t ptype, int32_t nullable, const uint8_t* pages, uint64_t pages_len,
                        const PageMeta* metas, uint64_t n_pages, uint8_t* codecs, uint8_t* inner, char* err,
                        size_t errcap) {
    try {
        std::vector<uint8_t> c, in;
        stat_column(ptype, nullable != 0, pages, pages_len, metas, n_pages, c, in);
        memcpy(codecs, c.data(), c.size());
        memcpy(inner, in.data(), in.size());
        return 0;
    } catch (const std::exception& e) {
        set_err(err, errcap, e.what());
        return -1;
    }
}

// raw block codecs (cross-checked against liblz4 / libzstd / pyarrow in tests)
int64_t sbo_block_compress(int32_t codec, const uint8_t* src, uint64_t n, uint8_t* dst, uint64_t cap) {
    try {
        switch (codec) {
            case C_LZ4:
                return (int64_t)lz4_compress(src, n, dst, cap);
            case C_ZSTD:
                return (int64_t)zstd_compress(src, n, dst, cap);
            case C_SNAPPY:
                return (int64_t)snappy_compress(src, n, dst, cap);
        }
    } catch (const std::exception&) {
    }
    return -1;
}
uint64_t sbo_block_bound(int32_t codec, uint64_t n) {
    switch (codec) {
        case C_LZ4:
            return lz4_compress_bound(n);
        case C_ZSTD:
            return zstd_compress_bound(n);
        case C_SNAPPY:
            return snappy_compress_bound(n);
    }
    return n;
}
int32_t sbo_block_decompress(int32_t codec, const uint8_t* src, uint64_t n, uint8_t* dst, uint64_t out_len,
                             char* err, size_t errcap) {
    try {
        switch (codec) {
            case C_LZ4:
                lz4_decompress(src, n, dst, out_len);
                return 0;
            case C_ZSTD:
                zstd_decompress(src, n, dst, out_len);
                return 0;
            case C_SNAPPY:
                snappy_decompress(src, n, dst, out_len);
                return 0;
        }
        set_err(err, errcap, "unknown block codec");
        return -1;
    } catch (const std::exception& e) {
        set_err(err, errcap, e.what());
        return -1;
    }
}

uint32_t sbo_patas_pack(uint32_t a, uint32_t b, uint32_t c) { return patas_pack(a, b, c); }
void sbo_patas_unpack(uint32_t pk, uint32_t* out3) {  // patas.rs:152-163
    out3[0] = (pk >> 9) & 0x7F;
    out3[1] = (pk >> 6) & 7;
    out3[2] = pk & 0x3F;
    if (out3[2] < 63 && out3[1] == 0) out3[1] = 8;
}
uint8_t sbo_bitpack_num_bits(const uint32_t* in128) { return bitpack4x_num_bits(in128); }
void sbo_bitpack_pack(const uint32_t* in128, uint8_t nb, uint8_t* out, int32_t delta, uint32_t initial) {
    bitpack4x_pack(in128, nb, out, delta != 0, initial);
}
void sbo_bitpack_unpack(const uint8_t* in, uint8_t nb, uint32_t* out128, int32_t delta, uint32_t initial) {
    bitpack4x_unpack(in, nb, out128, delta != 0, initial);
}
// bit 0: liblz4.so.1 in use, bit 1: libzstd.so.1 in use (bench.py's cpu_baseline leg); versions as the libraries report them
int32_t sbo_system_codecs(int32_t on) { return system_codecs_enable(on); }
int32_t sbo_system_codec_version(int32_t which) { return system_codec_version(which); }
uint64_t sbo_sample_rand(uint64_t seed, uint32_t depth, uint32_t codec, uint32_t i, uint64_t n) {
    return sample_rand(seed, depth, codec, i, n);
}
uint64_t sbo_mix64(uint64_t z) { return mix64(z); }

// CPU-baseline timing helpers: encode+decode of one column `iters` times, best wall time.
// Returns seconds for (write, read) in out2; the caller states cores = 1.
int32_t sbo_time_roundtrip(const sbo_column_in* c, const sbo_options* o, int32_t iters, double* out2, char* err,
                           size_t errcap) {
    try {
        ColumnIn col = to_col(c);
        WriteOptions w = to_opts(o);
        double best_w = 1e30, best_r = 1e30;
        for (int it = 0; it < iters; it++) {
            std::vector<uint8_t> bytes;
            std::vector<PageMeta> metas;
            auto t0 = std::chrono::steady_clock::now();
            write_column(col, w, bytes, metas);
            auto t1 = std::chrono::steady_clock::now();
            ColumnOut out;
            read_column(col.ptype, col.nullable, bytes.data(), bytes.size(), metas.data(), metas.size(), out);
            auto t2 = std::chrono::steady_clock::now();
            double dw = std::chrono::duration<double>(t1 - t0).count();
            double dr = std::chrono::duration<double>(t2 - t1).count();
            if (dw < best_w) best_w = dw;
            if (dr < best_r) best_r = dr;
        }
        out2[0] = best_w;
        out2[1] = best_r;
        return 0;
    } catch (const std::exception& e) {
        set_err(err, errcap, e.what());
        return -1;
    }
}

// Page-parallel CPU baseline (leg 2): the (column, page) work items of `n_cols` columns are
// pulled by `threads` std::threads from one atomic counter.  Phase 1 encodes every page (write_page on the
// page's slice, the body of encode_chunk's loop), phase 2 decodes every page back (read_column on the
// one-page column).  out2 = wall seconds of (encode phase, decode phase), best of `iters`;
// out_bytes = page bytes produced.  threads = 1 is the single-threaded leg over the same items.
int32_t sbo_time_pages_mt(const sbo_column_in* cols, uint64_t n_cols, const sbo_options* o, int32_t threads,
                          int32_t iters, double* out2, uint64_t* out_bytes, char* err, size_t errcap) {
    struct Item {
        ColumnIn page;
        WriteOptions opts;
        std::vector<uint8_t> bytes;
        PageMeta meta;
    };
    try {
        WriteOptions w = to_opts(o);
        std::vector<Item> items;
        for (uint64_t ci = 0; ci < n_cols; ci++) {
            ColumnIn col = to_col(&cols[ci]);
            if (col.rows == 0) continue;
            const uint64_t ps = w.max_page_size ? std::min<uint64_t>(w.max_page_size, col.rows) : col.rows;
            const size_t wd = type_width(col.ptype);
            uint64_t k = 0;
            for (uint64_t off = 0; off < col.rows; off += ps, k++) {
                Item it;
                it.page = col;
                it.page.rows = off + ps > col.rows ? col.rows - off : ps;
                if (col.validity) it.page.validity_bit_offset = col.validity_bit_offset + off;
                if (col.ptype == T_BOOL)
                    it.page.values_bit_offset = col.values_bit_offset + off;
                else if (col.ptype == T_BIN32)
                    it.page.offsets = col.offsets + off * 4;
                else if (col.ptype == T_BIN64)
                    it.page.offsets = col.offsets + off * 8;
                else
                    it.page.values = col.values + off * wd;
                it.opts = w;
                it.opts.rng_seed = mix64(w.rng_seed ^ ((w.page_index0 + k) * 0xD6E8FEB86659FD93ull));
                it.meta.num_values = it.page.rows;
                it.meta.length = 0;
                items.push_back(std::move(it));
            }
        }
        if (threads < 1) threads = 1;
        std::string first_error;
        std::mutex mu;
        auto run_phase = [&](bool encode) {
            std::atomic<size_t> next{0};
            auto worker = [&]() {
                try {
                    for (;;) {
                        const size_t i = next.fetch_add(1);
                        if (i >= items.size()) break;
                        Item& it = items[i];
                        if (encode) {
                            it.bytes.clear();
                            write_page(it.page, it.opts, it.bytes);
                            it.meta.length = it.bytes.size();
                        } else {
                            ColumnOut out;
                            read_column(it.page.ptype, it.page.nullable, it.bytes.data(), it.bytes.size(), &it.meta, 1, out);
                        }
                    }
                } catch (const std::exception& e) {
                    std::lock_guard<std::mutex> g(mu);
                    if (first_error.empty()) first_error = e.what();
                }
            };
            auto t0 = std::chrono::steady_clock::now();
            std::vector<std::thread> th;
            for (int t = 1; t < threads; t++) th.emplace_back(worker);
            worker();
            for (auto& t : th) t.join();
            return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        };
        double best_w = 1e30, best_r = 1e30;
        for (int it = 0; it < iters; it++) {
            best_w = std::min(best_w, run_phase(true));
            best_r = std::min(best_r, run_phase(false));
        }
        if (!first_error.empty()) {
            set_err(err, errcap, first_error.c_str());
            return -1;
        }
        uint64_t total = 0;
        for (auto& it : items) total += it.bytes.size();
        out2[0] = best_w;
        out2[1] = best_r;
        if (out_bytes) *out_bytes = total;
        return 0;
    } catch (const std::exception& e) {
        set_err(err, errcap, e.what());
        return -1;
    }
}

}  // extern "C"
