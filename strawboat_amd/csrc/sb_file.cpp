// strawboat-hip: file framing (host only, no device code).
//
//   "ARROW2" 00 00 | pages of leaf column 0 | pages of leaf column 1 | ... | schema bytes |
//   u64 n_columns { u64 offset, u64 n_pages { u64 length, u64 num_values }* }* |
//   u32 schema_size | u32 meta_size | FF FF FF FF 00 00 00 00
//
// Writer side replaces NativeWriter::{start, write, finish} (src/write/writer.rs:91-167; magic and
// continuation marker src/lib.rs:34-35, write_continuation src/write/common.rs:124-128); the page
// bytes themselves come from sb_write_columns.  Reader side replaces read_meta / deserialize_meta
// (src/read/reader.rs:148-178), the schema-bytes half of infer_schema (:227-241) and the seeks of
// NativeReader (:87-146).  The schema flatbuffer is opaque here: arrow2's schema_to_bytes [3P] on
// the reference side, any Arrow IPC Schema message on ours.
#include <cerrno>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/strawboat_hip.h"

namespace {

thread_local std::string g_err;
int32_t fail(int32_t code, const std::string& m) {
    g_err = m;
    return code;
}
void put_u64(std::vector<uint8_t>& b, uint64_t v) {
    for (int i = 0; i < 8; i++) b.push_back((uint8_t)(v >> (8 * i)));
}
void put_u32(std::vector<uint8_t>& b, uint32_t v) {
    for (int i = 0; i < 4; i++) b.push_back((uint8_t)(v >> (8 * i)));
}
uint64_t get_u64(const uint8_t* p) {
    uint64_t v = 0;
    for (int i = 0; i < 8; i++) v |= (uint64_t)p[i] << (8 * i);
    return v;
}
uint32_t get_u32(const uint8_t* p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }

const uint8_t MAGIC[8] = {'A', 'R', 'R', 'O', 'W', '2', 0, 0};  // lib.rs:34 + the 2 pad bytes of writer.rs:100
const uint8_t EOS[8] = {0xFF, 0xFF, 0xFF, 0xFF, 0, 0, 0, 0};    // write_continuation(w, 0)

struct Column {
    uint64_t offset;
    std::vector<sb_page_meta> pages;
};

}  // namespace

// writer states of writer.rs:40-45
enum { ST_NONE = 0, ST_STARTED = 1, ST_WRITTEN = 2, ST_FINISHED = 3 };

struct sb_file_writer {
    FILE* f = nullptr;
    uint64_t offset = 0;
    int state = ST_NONE;
    std::vector<Column> cols;
};

struct sb_file_reader {
    FILE* f = nullptr;
    uint64_t size = 0;
    std::vector<Column> cols;
    std::vector<uint8_t> schema;
};

extern "C" {

const char* sb_file_last_error(void) { return g_err.c_str(); }

int32_t sb_file_writer_open(const char* path, sb_file_writer** out) {
    if (!path || !out) return fail(SB_ERR_INVALID, "null argument");
    FILE* f = fopen(path, "wb");
    if (!f) return fail(SB_ERR_IO, std::string("cannot create ") + path + ": " + strerror(errno));
    sb_file_writer* w = new sb_file_writer();
    w->f = f;
    *out = w;
    return SB_OK;
}

int32_t sb_file_writer_start(sb_file_writer* w) {
    if (!w) return fail(SB_ERR_INVALID, "null writer");
    if (w->state != ST_NONE) return fail(SB_ERR_OUT_OF_SPEC, "The strawboat file can only be started once");
    if (fwrite(MAGIC, 1, 8, w->f) != 8) return fail(SB_ERR_IO, "write failed");
    w->offset = 8;
    w->state = ST_STARTED;
    return SB_OK;
}

// One call per leaf column, in leaf order; all calls together are the single `write(chunk)` the
// format allows (writer.rs:108-112: one row group per file).
int32_t sb_file_writer_write_column(sb_file_writer* w, const uint8_t* pages, uint64_t pages_len, const sb_page_meta* metas,
                                    uint64_t n_pages) {
    if (!w || (!pages && pages_len) || (!metas && n_pages)) return fail(SB_ERR_INVALID, "null argument");
    if (w->state == ST_FINISHED)
        return fail(SB_ERR_OUT_OF_SPEC, "The strawboat file can only accept one RowGroup in a single file");
    if (w->state != ST_STARTED && w->state != ST_WRITTEN)
        return fail(SB_ERR_OUT_OF_SPEC,
                    "The strawboat file must be started before it can be written to. Call `start` before `write`");
    uint64_t sum = 0;
    for (uint64_t p = 0; p < n_pages; p++) sum += metas[p].length;
    if (sum != pages_len) return fail(SB_ERR_INVALID, "sum of PageMeta.length differs from the byte count");
    Column c;
    c.offset = w->offset;  // ColumnMeta.offset = absolute position of the first page (common.rs:76,111-114)
    c.pages.assign(metas, metas + n_pages);
    if (pages_len && fwrite(pages, 1, pages_len, w->f) != pages_len) return fail(SB_ERR_IO, "write failed");
    w->offset += pages_len;
    w->cols.push_back(std::move(c));
    w->state = ST_WRITTEN;
    return SB_OK;
}

int32_t sb_file_writer_finish(sb_file_writer* w, const uint8_t* schema_bytes, uint64_t schema_len, uint64_t* total_size) {
    if (!w || (!schema_bytes && schema_len)) return fail(SB_ERR_INVALID, "null argument");
    if (w->state != ST_WRITTEN)
        return fail(SB_ERR_OUT_OF_SPEC,
                    "The strawboat file must be written before it can be finished. Call `start` before `finish`");
    std::vector<uint8_t> meta;
    put_u64(meta, w->cols.size());
    for (const Column& c : w->cols) {
        put_u64(meta, c.offset);
        put_u64(meta, c.pages.size());
        for (const sb_page_meta& p : c.pages) {
            put_u64(meta, p.length);
            put_u64(meta, p.num_values);
        }
    }
    if (schema_len > 0xFFFFFFFFull || meta.size() > 0xFFFFFFFFull) return fail(SB_ERR_INVALID, "footer section over 4 GiB");
    std::vector<uint8_t> tail;
    put_u32(tail, (uint32_t)schema_len);
    put_u32(tail, (uint32_t)meta.size());
    tail.insert(tail.end(), EOS, EOS + 8);
    bool ok = (!schema_len || fwrite(schema_bytes, 1, schema_len, w->f) == schema_len) &&
              fwrite(meta.data(), 1, meta.size(), w->f) == meta.size() && fwrite(tail.data(), 1, tail.size(), w->f) == tail.size() &&
              fflush(w->f) == 0;
    if (!ok) return fail(SB_ERR_IO, "write failed");
    w->offset += schema_len + meta.size() + tail.size();
    if (total_size) *total_size = w->offset;
    w->state = ST_FINISHED;
    return SB_OK;
}

void sb_file_writer_close(sb_file_writer* w) {
    if (!w) return;
    if (w->f) fclose(w->f);
    delete w;
}

// ---------------------------------------------------------------------------------- reader
int32_t sb_file_reader_open(const char* path, sb_file_reader** out) {
    if (!path || !out) return fail(SB_ERR_INVALID, "null argument");
    FILE* f = fopen(path, "rb");
    if (!f) return fail(SB_ERR_IO, std::string("cannot open ") + path + ": " + strerror(errno));
    sb_file_reader* r = new sb_file_reader();
    r->f = f;
    auto bad = [&](int32_t code, const char* m) {
        fclose(f);
        delete r;
        return fail(code, m);
    };
    if (fseek(f, 0, SEEK_END) != 0) return bad(SB_ERR_IO, "seek failed");
    const long end = ftell(f);
    if (end < 24) return bad(SB_ERR_IO, "file shorter than header + footer");
    r->size = (uint64_t)end;
    uint8_t tail[16];
    if (fseek(f, end - 16, SEEK_SET) != 0 || fread(tail, 1, 16, f) != 16) return bad(SB_ERR_IO, "footer read failed");
    const uint64_t schema_size = get_u32(tail), meta_size = get_u32(tail + 4);  // reader.rs:229-233
    if (16 + meta_size + schema_size + 8 > r->size) return bad(SB_ERR_IO, "footer sizes exceed the file");
    std::vector<uint8_t> meta(meta_size);
    r->schema.resize(schema_size);
    if (fseek(f, (long)(r->size - 16 - meta_size - schema_size), SEEK_SET) != 0 ||
        (schema_size && fread(r->schema.data(), 1, schema_size, f) != schema_size) ||
        (meta_size && fread(meta.data(), 1, meta_size, f) != meta_size))
        return bad(SB_ERR_IO, "footer read failed");
    // deserialize_meta (reader.rs:148-166)
    size_t pos = 0;
    auto need = [&](size_t n) { return pos + n <= meta.size(); };
    if (!need(8)) return bad(SB_ERR_IO, "meta truncated");
    const uint64_t ncol = get_u64(meta.data());
    pos = 8;
    for (uint64_t i = 0; i < ncol; i++) {
        if (!need(16)) return bad(SB_ERR_IO, "meta truncated");
        Column c;
        c.offset = get_u64(meta.data() + pos);
        const uint64_t np = get_u64(meta.data() + pos + 8);
        pos += 16;
        if (np > (meta.size() - pos) / 16) return bad(SB_ERR_IO, "meta truncated");
        c.pages.resize(np);
        for (uint64_t p = 0; p < np; p++) {
            c.pages[p].length = get_u64(meta.data() + pos);
            c.pages[p].num_values = get_u64(meta.data() + pos + 8);
            pos += 16;
        }
        r->cols.push_back(std::move(c));
    }
    *out = r;
    return SB_OK;
}

uint64_t sb_file_reader_n_columns(const sb_file_reader* r) { return r ? r->cols.size() : 0; }

int32_t sb_file_reader_column(const sb_file_reader* r, uint64_t col, uint64_t* offset, uint64_t* n_pages, const sb_page_meta** pages) {
    if (!r || col >= r->cols.size()) return fail(SB_ERR_INVALID, "no such column");
    if (offset) *offset = r->cols[col].offset;
    if (n_pages) *n_pages = r->cols[col].pages.size();
    if (pages) *pages = r->cols[col].pages.data();
    return SB_OK;
}

int32_t sb_file_reader_schema(const sb_file_reader* r, const uint8_t** bytes, uint64_t* len) {
    if (!r || !bytes || !len) return fail(SB_ERR_INVALID, "null argument");
    *bytes = r->schema.data();
    *len = r->schema.size();
    return SB_OK;
}

// pages [first_page, first_page + n_pages) of a column into dst: ColumnMeta::slice (lib.rs:47-61)
// + the read_exact of NativeReader::next (reader.rs:117-127)
int32_t sb_file_reader_read_pages(sb_file_reader* r, uint64_t col, uint64_t first_page, uint64_t n_pages, uint8_t* dst,
                                  uint64_t capacity, uint64_t* bytes_read) {
    if (!r || col >= r->cols.size()) return fail(SB_ERR_INVALID, "no such column");
    const Column& c = r->cols[col];
    if (first_page > c.pages.size() || n_pages > c.pages.size() - first_page) return fail(SB_ERR_INVALID, "page range out of bounds");
    uint64_t off = c.offset, len = 0;
    for (uint64_t p = 0; p < first_page; p++) off += c.pages[p].length;
    for (uint64_t p = first_page; p < first_page + n_pages; p++) len += c.pages[p].length;
    if (len > capacity) return fail(SB_ERR_INVALID, "destination too small");
    if (off + len > r->size) return fail(SB_ERR_IO, "page range exceeds the file");
    if (len && (fseek(r->f, (long)off, SEEK_SET) != 0 || fread(dst, 1, len, r->f) != len)) return fail(SB_ERR_IO, "page read failed");
    if (bytes_read) *bytes_read = len;
    return SB_OK;
}

void sb_file_reader_close(sb_file_reader* r) {
    if (!r) return;
    if (r->f) fclose(r->f);
    delete r;
}

}  // extern "C"

// ---------------------------------------------------------------------------------- page inspector
// stat::stat_simple and friends (src/stat.rs:61-152): walk the block headers of one page.
namespace {
uint32_t stat_width(int32_t t) {
    switch (t) {
        case SB_TYPE_INT8: case SB_TYPE_UINT8: return 1;
        case SB_TYPE_INT16: case SB_TYPE_UINT16: return 2;
        case SB_TYPE_INT32: case SB_TYPE_UINT32: case SB_TYPE_FLOAT32: return 4;
        case SB_TYPE_INT64: case SB_TYPE_UINT64: case SB_TYPE_FLOAT64: return 8;
        case SB_TYPE_INT128: return 16;
        case SB_TYPE_INT256: return 32;
        default: return 0;
    }
}
uint32_t rd32(const uint8_t* p) {
    uint32_t v;
    memcpy(&v, p, 4);
    return v;
}
// stat_body (src/stat.rs:82-113); returns 0 or an error code, appends to out
int32_t stat_block(const uint8_t*& cur, const uint8_t* end, int32_t ptype, sb_page_info* out, uint32_t cap, uint32_t& n) {
    if (n >= cap) return SB_ERR_INVALID;
    if (end - cur < 9) return SB_ERR_IO;
    const uint32_t me0 = n;
    sb_page_info& pi = out[n++];
    memset(&pi, 0, sizeof pi);
    const uint32_t codec = cur[0];
    pi.compressed_size = rd32(cur + 1);
    pi.uncompressed_size = rd32(cur + 5);
    if (!(codec <= 3 || (codec >= 10 && codec <= 16))) return SB_ERR_OUT_OF_SPEC;  // Compression::from_codec
    pi.codec = (int32_t)codec;
    cur += 9;
    const uint8_t* body = cur;
    const bool is_bin = ptype == SB_TYPE_BINARY || ptype == SB_TYPE_LARGE_BINARY;
    if (codec == SB_CODEC_DICT) {  // stat_dict_body (:145-152): nested indices block, then u32 unique_num
        const uint8_t* q = body;
        pi.has_nested = 1;
        const uint32_t me = n - 1;
        // (the indices are a compress_integer::<u32> block: a nested Freq block inside has a 4-byte top value.
        // Upstream passes the column's type down, src/stat.rs:146, and misreads such a page.)
        const int32_t rc = stat_block(q, end, SB_TYPE_UINT32, out, cap, n);
        if (rc) return rc;
        if (end - q < 4) return SB_ERR_IO;
        out[me].unique_num = rd32(q);
    } else if (codec == SB_CODEC_FREQ) {  // stat_freq_body (:115-143)
        const uint8_t* q = body;
        if (is_bin) {
            if (end - q < 8) return SB_ERR_IO;
            uint64_t len;
            memcpy(&len, q, 8);
            if ((uint64_t)(end - q) - 8 < len || (uint64_t)(end - q) - 8 - len < 4) return SB_ERR_IO;
            pi.exceptions_bitmap_size = rd32(q + 8 + len);
        } else {
            const uint32_t w = stat_width(ptype);
            if (!w) return SB_ERR_OUT_OF_SPEC;  // unreachable!("type not supported") upstream
            if ((uint64_t)(end - q) < (uint64_t)w + 4) return SB_ERR_IO;
            const uint32_t bm = rd32(q + w);
            if ((uint64_t)(end - q) - w - 4 < bm) return SB_ERR_IO;
            q += w + 4 + bm;
            const uint32_t me = n - 1;
            out[me].exceptions_bitmap_size = bm;
            out[me].has_nested = 1;
            const int32_t rc = stat_block(q, end, ptype, out, cap, n);
            if (rc) return rc;
        }
    }
    const uint32_t csize = out[me0].compressed_size;
    if ((uint64_t)(end - body) < csize) return SB_ERR_IO;  // `*buffer = &buffer[compressed_size..]` panics upstream
    cur = body + csize;
    return SB_OK;
}
}  // namespace

extern "C" int32_t sb_stat_page(const uint8_t* page, uint64_t length, int32_t physical_type, int32_t is_nullable,
                                sb_page_info* out, uint32_t capacity, uint32_t* n_out) {
    if (!page || !out || !n_out || capacity == 0) return SB_ERR_INVALID;
    const uint8_t* cur = page;
    const uint8_t* end = page + length;
    int32_t has_vs = 0;
    uint32_t vs = 0;
    if (is_nullable) {  // stat_simple (:72-77)
        if (length < 4) return SB_ERR_IO;
        const uint32_t def_len = rd32(cur);
        if (length - 4 < def_len || length - 4 - def_len < 4) return SB_ERR_IO;
        cur += 4 + def_len;
        vs = rd32(cur);
        has_vs = 1;
    }
    uint32_t n = 0;
    const int32_t rc = stat_block(cur, end, physical_type, out, capacity, n);
    if (rc) return rc;
    out[0].has_validity_size = has_vs;
    out[0].validity_size = vs;
    *n_out = n;
    return SB_OK;
}
