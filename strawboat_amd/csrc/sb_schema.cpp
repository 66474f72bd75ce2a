// strawboat-hip: the footer's schema bytes (host only).
//
// NativeWriter::finish stores `schema_to_bytes(&schema, &default_ipc_fields(..))` (src/write/writer.rs:137-139): the
// bare Arrow IPC `Message` flatbuffer whose header is a `Schema` table (arrow-format's Message.fbs / Schema.fbs:
// MetadataVersion V5, body_length 0, no custom metadata on the message); infer_schema hands the same bytes to
// arrow2's deserialize_schema (src/read/reader.rs:227-241).  This file writes and reads that flatbuffer itself, so
// that a strawboat file's footer does not depend on another Arrow implementation:
//
//   Message { version: V5, header_type: Schema, header: Schema { endianness: Little, fields: [Field],
//             custom_metadata: [KeyValue]? }, bodyLength: 0 }
//   Field   { name, nullable, type_type, type, children: [Field] (always present, as arrow2's serialize_field writes
//             it), custom_metadata? }
//
// What is pinned: the tables, fields and values (arrow2 0.17 io/ipc/write/schema.rs serialize_schema /
// serialize_field / serialize_type).  Where a builder PLACES tables, strings and vtables inside the buffer is its own
// business (planus, flatc and this builder differ); every Arrow reader navigates by the offsets, and the tests round
// trip through pyarrow in both directions.  Scalars equal to their schema default are omitted, as planus does.
#include <cstring>
#include <string>
#include <vector>

#include "../../include/strawboat_hip.h"

namespace {

// ------------------------------------------------------------------ a minimal flatbuffer builder (back to front)
class Builder {
  public:
    // the buffer grows towards lower addresses; `buf_` holds the finished part at its END
    std::vector<uint8_t> buf_;
    size_t used_ = 0;      // bytes in use (at the end of buf_)
    size_t minalign_ = 1;

    uint32_t size() const { return (uint32_t)used_; }
    uint8_t* grow(size_t n) {
        if (used_ + n > buf_.size()) {
            const size_t ncap = std::max<size_t>(256, (buf_.size() + n) * 2);
            std::vector<uint8_t> nb(ncap, 0);
            if (used_) memcpy(nb.data() + ncap - used_, buf_.data() + buf_.size() - used_, used_);
            buf_.swap(nb);
        }
        used_ += n;
        return buf_.data() + buf_.size() - used_;
    }
    void pad_to(size_t align, size_t upcoming) {  // so that after `upcoming` more bytes the position is `align`-aligned
        if (align > minalign_) minalign_ = align;
        const size_t padn = (~(used_ + upcoming) + 1) & (align - 1);
        uint8_t* p = grow(padn);
        memset(p, 0, padn);
    }
    template <class T>
    void push(T v) {
        pad_to(sizeof(T), sizeof(T));
        uint8_t* p = grow(sizeof(T));
        memcpy(p, &v, sizeof(T));
    }
    // uoffset to an object that ends `off` bytes before the end of the buffer, as stored at the current position
    void push_uoffset(uint32_t off) {
        pad_to(4, 4);
        const uint32_t rel = (uint32_t)(used_ + 4) - off;
        uint8_t* p = grow(4);
        memcpy(p, &rel, 4);
    }
    uint32_t string(const std::string& s) {
        pad_to(4, s.size() + 1 + 4);
        uint8_t* p = grow(s.size() + 1);
        memcpy(p, s.data(), s.size());
        p[s.size()] = 0;
        push<uint32_t>((uint32_t)s.size());
        return size();
    }
    uint32_t offset_vector(const std::vector<uint32_t>& offs) {
        pad_to(4, 4 * offs.size() + 4);
        for (size_t i = offs.size(); i-- > 0;) push_uoffset(offs[i]);
        push<uint32_t>((uint32_t)offs.size());
        return size();
    }

    // ---- tables
    struct Slot {
        uint16_t id;
        uint16_t bytes;
        uint32_t at;  // position (size() after the field was written)
    };
    std::vector<Slot> slots_;
    void start_table() { slots_.clear(); }
    template <class T>
    void add(uint16_t id, T v, T def) {
        if (v == def) return;
        push<T>(v);
        slots_.push_back({id, (uint16_t)sizeof(T), size()});
    }
    void add_offset(uint16_t id, uint32_t off) {
        if (!off) return;
        push_uoffset(off);
        slots_.push_back({id, 4, size()});
    }
    uint32_t end_table() {
        // the table starts with an soffset to its vtable; the fields written so far sit behind it (higher addresses):
        // a field at position `at` begins (table_at - at) bytes after the table start
        pad_to(4, 4);
        grow(4);
        const uint32_t table_at = size();
        uint16_t nf = 0;
        for (auto& s : slots_) nf = std::max<uint16_t>(nf, (uint16_t)(s.id + 1));
        const uint16_t vt_size = (uint16_t)(4 + 2 * nf);
        std::vector<uint16_t> vt(2 + nf, 0);
        uint16_t obj = 4;
        for (auto& s : slots_) {
            const uint16_t fo = (uint16_t)(table_at - s.at);
            vt[2 + s.id] = fo;
            obj = std::max<uint16_t>(obj, (uint16_t)(fo + s.bytes));
        }
        vt[0] = vt_size;
        vt[1] = obj;
        pad_to(2, vt_size);
        uint8_t* vp = grow(vt_size);
        memcpy(vp, vt.data(), vt_size);
        const int32_t so = (int32_t)(size() - table_at);  // vtable address = table address - soffset
        memcpy(buf_.data() + buf_.size() - table_at, &so, 4);
        return table_at;
    }
    std::vector<uint8_t> finish(uint32_t root) {
        pad_to(std::max<size_t>(minalign_, 8), 4);
        push_uoffset(root);
        return std::vector<uint8_t>(buf_.end() - used_, buf_.end());
    }
};

// ------------------------------------------------------------------ bounds-checked reader
struct Reader {
    const uint8_t* b;
    size_t n;
    bool ok = true;
    template <class T>
    T rd(size_t p) {
        T v{};
        if (p > n || n - p < sizeof(T)) {   // (no p + sizeof(T): a hostile offset may sit just below SIZE_MAX)
            ok = false;
            return v;
        }
        memcpy(&v, b + p, sizeof(T));
        return v;
    }
    size_t indirect(size_t p) { return p + rd<uint32_t>(p); }
    // position of field `id` of the table at `t`, 0 when absent
    size_t field(size_t t, uint16_t id) {
        const int32_t so = rd<int32_t>(t);
        const int64_t vts = (int64_t)t - so;
        if (!ok || vts < 0 || (uint64_t)vts >= n) {   // the vtable lies inside the buffer
            ok = false;
            return 0;
        }
        const size_t vt = (size_t)vts;
        const uint16_t vsz = rd<uint16_t>(vt);
        if (!ok || 4 + 2 * (size_t)id + 2 > vsz) return 0;
        const uint16_t fo = rd<uint16_t>(vt + 4 + 2 * id);
        return fo ? t + fo : 0;
    }
    template <class T>
    T scalar(size_t t, uint16_t id, T def) {
        const size_t p = field(t, id);
        return p ? rd<T>(p) : def;
    }
    size_t table(size_t t, uint16_t id) {
        const size_t p = field(t, id);
        return p ? indirect(p) : 0;
    }
    bool str(size_t t, uint16_t id, std::string* out) {
        const size_t p = field(t, id);
        if (!p) return false;
        const size_t s = indirect(p);
        const uint32_t len = rd<uint32_t>(s);
        if (!ok || s > n || n - s < 4 || n - s - 4 < (size_t)len) {
            ok = false;
            return false;
        }
        out->assign((const char*)b + s + 4, len);
        return true;
    }
    // vector of offsets: element count and the position of element i's target
    uint32_t vec_len(size_t v) { return v ? rd<uint32_t>(v) : 0; }
    size_t vec_table(size_t v, uint32_t i) { return indirect(v + 4 + 4 * (size_t)i); }
};

thread_local std::string g_schema_err;
int32_t sfail(int32_t code, const char* m) {
    g_schema_err = m;
    return code;
}

// field ids of Schema.fbs
enum { F_NAME = 0, F_NULLABLE = 1, F_TYPE_TYPE = 2, F_TYPE = 3, F_DICT = 4, F_CHILDREN = 5, F_META = 6 };

uint32_t write_type(Builder& fb, const sb_schema_field& f) {
    fb.start_table();
    switch (f.type_id) {
        case SB_ARROW_INT:
            fb.add<int32_t>(0, f.bit_width, 0);
            fb.add<uint8_t>(1, f.is_signed ? 1 : 0, 0);
            break;
        case SB_ARROW_FLOATING_POINT:
            fb.add<int16_t>(0, (int16_t)f.precision, 0);
            break;
        case SB_ARROW_DECIMAL:
            fb.add<int32_t>(0, f.precision, 0);
            fb.add<int32_t>(1, f.scale, 0);
            fb.add<int32_t>(2, f.bit_width, 128);
            break;
        case SB_ARROW_DATE:
            fb.add<int16_t>(0, (int16_t)f.unit, 1);
            break;
        case SB_ARROW_TIME:
            fb.add<int16_t>(0, (int16_t)f.unit, 1);
            fb.add<int32_t>(1, f.bit_width, 32);
            break;
        case SB_ARROW_TIMESTAMP: {
            const uint32_t tz = f.timezone && f.timezone[0] ? fb.string(f.timezone) : 0;
            fb.start_table();
            fb.add<int16_t>(0, (int16_t)f.unit, 0);
            fb.add_offset(1, tz);
            break;
        }
        case SB_ARROW_DURATION:
            fb.add<int16_t>(0, (int16_t)f.unit, 1);
            break;
        case SB_ARROW_INTERVAL:
            fb.add<int16_t>(0, (int16_t)f.unit, 0);
            break;
        case SB_ARROW_FIXED_SIZE_BINARY:
        case SB_ARROW_FIXED_SIZE_LIST:
            fb.add<int32_t>(0, f.bit_width, 0);
            break;
        case SB_ARROW_MAP:
            fb.add<uint8_t>(0, f.is_signed ? 1 : 0, 0);
            break;
        default:  // Null, Binary, Utf8, Bool, List, Struct_, LargeBinary, LargeUtf8, LargeList: empty tables
            break;
    }
    return fb.end_table();
}

uint32_t write_key_values(Builder& fb, const char* packed, uint64_t n);

// writes fields[*pos] and its children (pre-order); returns the Field table
uint32_t write_field(Builder& fb, const sb_schema_field* fields, uint64_t n, uint64_t* pos, bool* bad) {
    if (*pos >= n) {
        *bad = true;
        return 0;
    }
    const sb_schema_field& f = fields[(*pos)++];
    std::vector<uint32_t> kids;
    for (int32_t k = 0; k < f.n_children && !*bad; k++) kids.push_back(write_field(fb, fields, n, pos, bad));
    if (*bad) return 0;
    const uint32_t meta = (f.n_metadata && f.metadata) ? write_key_values(fb, f.metadata, f.n_metadata) : 0;
    const uint32_t children = fb.offset_vector(kids);
    const uint32_t type = write_type(fb, f);
    const uint32_t name = fb.string(f.name ? f.name : "");
    fb.start_table();
    fb.add_offset(F_META, meta);
    fb.add_offset(F_CHILDREN, children);
    fb.add_offset(F_TYPE, type);
    fb.add_offset(F_NAME, name);
    fb.add<uint8_t>(F_TYPE_TYPE, (uint8_t)f.type_id, 0);
    fb.add<uint8_t>(F_NULLABLE, f.nullable ? 1 : 0, 0);
    return fb.end_table();
}

struct Parsed {
    std::vector<sb_schema_field> fields;
    std::vector<std::string> names, zones, metas;  // kept beside the structs until the strings are copied out
    size_t max_fields = 0;                         // a field costs >= 8 bytes of footer: a crafted footer whose vector
                                                   // slots all point at the same child table cannot expand beyond that
};

// [KeyValue] at field `id` of table `t` -> "key\0value\0" per pair, appended to *packed; returns the pair count
uint64_t read_key_values(Reader& r, size_t t, uint16_t id, std::string* packed) {
    const size_t mp = r.field(t, id);
    const size_t mv = mp ? r.indirect(mp) : 0;
    const uint32_t n = r.vec_len(mv);
    if (!r.ok || (size_t)n > r.n / 4) {
        r.ok = false;
        return 0;
    }
    for (uint32_t k = 0; k < n && r.ok; k++) {
        const size_t kv = r.vec_table(mv, k);
        std::string key, val;
        r.str(kv, 0, &key);
        r.str(kv, 1, &val);
        packed->append(key.c_str());   // (C strings on this ABI: cut at an embedded NUL)
        packed->push_back('\0');
        packed->append(val.c_str());
        packed->push_back('\0');
        // slots may all point at one table with a long string: honest metadata is never larger than the buffer it came from
        if (packed->size() > r.n + 2 * (size_t)n) r.ok = false;
    }
    return n;
}

uint32_t write_key_values(Builder& fb, const char* packed, uint64_t n) {
    std::vector<uint32_t> kvs;
    const char* p = packed;
    for (uint64_t k = 0; k < n; k++) {
        const std::string key(p);
        p += key.size() + 1;
        const std::string val(p);
        p += val.size() + 1;
        const uint32_t v = fb.string(val);
        const uint32_t ko = fb.string(key);
        fb.start_table();
        fb.add_offset(1, v);
        fb.add_offset(0, ko);
        kvs.push_back(fb.end_table());
    }
    return fb.offset_vector(kvs);
}

bool read_field(Reader& r, size_t t, Parsed& out, int depth) {
    if (depth > 64 || out.fields.size() >= out.max_fields) return false;
    sb_schema_field f;
    memset(&f, 0, sizeof f);
    std::string name;
    r.str(t, F_NAME, &name);
    f.nullable = r.scalar<uint8_t>(t, F_NULLABLE, 0);
    f.type_id = r.scalar<uint8_t>(t, F_TYPE_TYPE, 0);
    const size_t ty = r.table(t, F_TYPE);
    std::string tz;
    if (ty) {
        switch (f.type_id) {
            case SB_ARROW_INT:
                f.bit_width = r.scalar<int32_t>(ty, 0, 0);
                f.is_signed = r.scalar<uint8_t>(ty, 1, 0);
                break;
            case SB_ARROW_FLOATING_POINT:
                f.precision = r.scalar<int16_t>(ty, 0, 0);
                break;
            case SB_ARROW_DECIMAL:
                f.precision = r.scalar<int32_t>(ty, 0, 0);
                f.scale = r.scalar<int32_t>(ty, 1, 0);
                f.bit_width = r.scalar<int32_t>(ty, 2, 128);
                break;
            case SB_ARROW_DATE:
                f.unit = r.scalar<int16_t>(ty, 0, 1);
                break;
            case SB_ARROW_TIME:
                f.unit = r.scalar<int16_t>(ty, 0, 1);
                f.bit_width = r.scalar<int32_t>(ty, 1, 32);
                break;
            case SB_ARROW_TIMESTAMP:
                f.unit = r.scalar<int16_t>(ty, 0, 0);
                r.str(ty, 1, &tz);
                break;
            case SB_ARROW_DURATION:
                f.unit = r.scalar<int16_t>(ty, 0, 1);
                break;
            case SB_ARROW_INTERVAL:
                f.unit = r.scalar<int16_t>(ty, 0, 0);
                break;
            case SB_ARROW_FIXED_SIZE_BINARY:
            case SB_ARROW_FIXED_SIZE_LIST:
                f.bit_width = r.scalar<int32_t>(ty, 0, 0);
                break;
            case SB_ARROW_MAP:
                f.is_signed = r.scalar<uint8_t>(ty, 0, 0);
                break;
            default:
                break;
        }
    }
    const size_t cp = r.field(t, F_CHILDREN);
    const size_t cv = cp ? r.indirect(cp) : 0;
    const uint32_t nc = r.vec_len(cv);
    if (!r.ok || nc > 4096) return false;
    f.n_children = (int32_t)nc;
    std::string meta;
    f.n_metadata = read_key_values(r, t, F_META, &meta);
    if (!r.ok) return false;
    out.fields.push_back(f);
    out.names.push_back(name);
    out.zones.push_back(tz);
    out.metas.push_back(meta);
    for (uint32_t k = 0; k < nc; k++)
        if (!read_field(r, r.vec_table(cv, k), out, depth + 1)) return false;
    return r.ok;
}

}  // namespace

extern "C" {

const char* sb_schema_last_error(void) { return g_schema_err.c_str(); }

int32_t sb_schema_to_bytes(const sb_schema_field* fields, uint64_t n_fields, uint64_t n_top, const char* const* metadata,
                           uint64_t n_metadata, uint8_t* out, uint64_t capacity, uint64_t* len) {
    if ((!fields && n_fields) || !len || (n_metadata && !metadata)) return sfail(SB_ERR_INVALID, "null argument");
    Builder fb;
    uint64_t pos = 0;
    bool bad = false;
    std::vector<uint32_t> tops;
    for (uint64_t k = 0; k < n_top && !bad; k++) tops.push_back(write_field(fb, fields, n_fields, &pos, &bad));
    if (bad || pos != n_fields) return sfail(SB_ERR_INVALID, "n_children do not add up to n_fields");
    uint32_t meta = 0;
    if (n_metadata) {  // Schema.custom_metadata: present only when the schema carries metadata (serialize_schema)
        std::vector<uint32_t> kvs;
        for (uint64_t k = 0; k < n_metadata; k++) {
            const uint32_t v = fb.string(metadata[2 * k + 1] ? metadata[2 * k + 1] : "");
            const uint32_t key = fb.string(metadata[2 * k] ? metadata[2 * k] : "");
            fb.start_table();
            fb.add_offset(1, v);
            fb.add_offset(0, key);
            kvs.push_back(fb.end_table());
        }
        meta = fb.offset_vector(kvs);
    }
    const uint32_t fvec = fb.offset_vector(tops);
    fb.start_table();                       // Schema
    fb.add_offset(2, meta);
    fb.add_offset(1, fvec);
    fb.add<int16_t>(0, 0, 0);               // endianness Little = default
    const uint32_t schema = fb.end_table();
    fb.start_table();                       // Message
    fb.add<int64_t>(3, 0, 0);               // bodyLength 0 = default
    fb.add_offset(2, schema);
    fb.add<uint8_t>(1, 1, 0);               // header_type = MessageHeader::Schema
    fb.add<int16_t>(0, 4, 0);               // version = MetadataVersion::V5
    const uint32_t msg = fb.end_table();
    const std::vector<uint8_t> bytes = fb.finish(msg);
    *len = bytes.size();
    if (bytes.size() > capacity || !out) return sfail(SB_ERR_INVALID, "schema buffer too small");
    memcpy(out, bytes.data(), bytes.size());
    return SB_OK;
}

int32_t sb_schema_from_bytes(const uint8_t* bytes, uint64_t len, sb_schema_field* out, uint64_t capacity, uint64_t* n_fields,
                             uint64_t* n_top, char* strings, uint64_t strings_capacity, uint64_t* strings_len) {
    if (!bytes || !n_fields || !n_top || !strings_len) return sfail(SB_ERR_INVALID, "null argument");
    Reader r{bytes, (size_t)len};
    const size_t msg = r.indirect(0);
    if (!r.ok) return sfail(SB_ERR_OUT_OF_SPEC, "schema bytes: not a flatbuffer");
    if (r.scalar<uint8_t>(msg, 1, 0) != 1) return sfail(SB_ERR_OUT_OF_SPEC, "schema bytes: the message header is not a Schema");
    const size_t schema = r.table(msg, 2);
    if (!schema || !r.ok) return sfail(SB_ERR_OUT_OF_SPEC, "schema bytes: no Schema table");
    if (r.scalar<int16_t>(schema, 0, 0) != 0) return sfail(SB_ERR_NYI, "big-endian schema");
    const size_t fp = r.field(schema, 1);
    const size_t fv = fp ? r.indirect(fp) : 0;
    const uint32_t nt = r.vec_len(fv);
    Parsed p;
    p.max_fields = (size_t)len / 8 + 1;
    for (uint32_t k = 0; k < nt; k++)
        if (!read_field(r, r.vec_table(fv, k), p, 0))
            return sfail(SB_ERR_OUT_OF_SPEC, "schema bytes: malformed Field (or more fields than the buffer can hold)");
    if (!r.ok) return sfail(SB_ERR_OUT_OF_SPEC, "schema bytes: offset out of bounds");
    uint64_t need = 0;
    for (size_t i = 0; i < p.fields.size(); i++) need += p.names[i].size() + 1 + p.zones[i].size() + 1 + p.metas[i].size();
    *n_fields = p.fields.size();
    *n_top = nt;
    *strings_len = need;
    if (p.fields.size() > capacity || need > strings_capacity || !out || !strings) return sfail(SB_ERR_INVALID, "output arrays too small");
    char* w = strings;
    for (size_t i = 0; i < p.fields.size(); i++) {
        out[i] = p.fields[i];
        memcpy(w, p.names[i].c_str(), p.names[i].size() + 1);
        out[i].name = w;
        w += p.names[i].size() + 1;
        memcpy(w, p.zones[i].c_str(), p.zones[i].size() + 1);
        out[i].timezone = p.zones[i].empty() ? nullptr : w;
        w += p.zones[i].size() + 1;
        if (out[i].n_metadata) {
            memcpy(w, p.metas[i].data(), p.metas[i].size());
            out[i].metadata = w;
            w += p.metas[i].size();
        } else {
            out[i].metadata = nullptr;
        }
    }
    return SB_OK;
}

int32_t sb_schema_metadata_from_bytes(const uint8_t* bytes, uint64_t len, char* strings, uint64_t strings_capacity,
                                      uint64_t* n_pairs, uint64_t* strings_len) {
    if (!bytes || !n_pairs || !strings_len) return sfail(SB_ERR_INVALID, "null argument");
    Reader r{bytes, (size_t)len};
    const size_t msg = r.indirect(0);
    if (!r.ok) return sfail(SB_ERR_OUT_OF_SPEC, "schema bytes: not a flatbuffer");
    if (r.scalar<uint8_t>(msg, 1, 0) != 1) return sfail(SB_ERR_OUT_OF_SPEC, "schema bytes: the message header is not a Schema");
    const size_t schema = r.table(msg, 2);
    if (!schema || !r.ok) return sfail(SB_ERR_OUT_OF_SPEC, "schema bytes: no Schema table");
    std::string packed;
    const uint64_t n = read_key_values(r, schema, 2, &packed);
    if (!r.ok) return sfail(SB_ERR_OUT_OF_SPEC, "schema bytes: malformed custom_metadata");
    *n_pairs = n;
    *strings_len = packed.size();
    if (packed.size() > strings_capacity || (!strings && packed.size())) return sfail(SB_ERR_INVALID, "output buffer too small");
    if (packed.size()) memcpy(strings, packed.data(), packed.size());
    return SB_OK;
}

}  // extern "C"
