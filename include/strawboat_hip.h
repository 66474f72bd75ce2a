/*
 * strawboat_hip.h — C ABI of the MI355X-native page encode/decode path of strawboat.
 *
 * This is the drop-in boundary a host (the Rust crate, or the C++/Python host layer in
 * this repo) binds: plain pointers and sizes, no C++/torch types, `extern "C"`, int32
 * status codes, errors as strings per context.  Every entry point names the reference
 * seam it replaces (file:line under the sundy-li/strawboat tree).
 *
 * Unit of work: the *pages of leaf columns*.  A call takes a batch of columns; every
 * (column, page) is an independent work item scheduled over the GPU.
 * All work is enqueued on the context's HIP stream; nothing is valid on the host until
 * sb_ctx_synchronize() returned SB_OK.
 *
 * Memory: buffers marked DEVICE are HBM pointers (hipMalloc / torch tensors on the
 * context's device) when `mem == SB_MEM_DEVICE` — the measured configuration — or host
 * pointers when `mem == SB_MEM_HOST` (the library stages them over PCIe itself; this is
 * the shape the reference's `&[u8]` / `Vec<T>` callers have).
 */
#ifndef STRAWBOAT_HIP_H
#define STRAWBOAT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes: arrow2::error::Error variants used on this path
 *      (src/errors.rs:19-31, src/compression/mod.rs:78-80, basic.rs:104,116,129) */
#define SB_OK 0
#define SB_ERR_OUT_OF_SPEC (-1)  /* Error::OutOfSpec: unknown codec, corrupt page, size mismatch */
#define SB_ERR_EXTERNAL (-2)     /* Error::External: LZ4/Zstd stream rejected, HIP runtime failure */
#define SB_ERR_IO (-3)           /* Error::Io(UnexpectedEof): page shorter than its headers say */
#define SB_ERR_NYI (-4)          /* Error::NotYetImplemented: codec/type not handled on the device */
#define SB_ERR_INVALID (-5)      /* bad argument at the boundary (null pointer, capacity too small) */

/* ---- on-disk codec ids: enum Compression (src/compression/mod.rs:37-51,92-108) */
#define SB_CODEC_NONE 0
#define SB_CODEC_LZ4 1
#define SB_CODEC_ZSTD 2
#define SB_CODEC_SNAPPY 3
#define SB_CODEC_RLE 10
#define SB_CODEC_DICT 11
#define SB_CODEC_ONEVALUE 12
#define SB_CODEC_FREQ 13
#define SB_CODEC_BITPACKING 14
#define SB_CODEC_DELTA_BITPACKING 15
#define SB_CODEC_PATAS 16

/* ---- physical kinds: the dispatch key of batch_read::read_simple
 *      (src/read/batch_read.rs:37-63) and write_simple (src/write/serialize.rs:62-129).
 *      Utf8/LargeUtf8 are written as Binary/LargeBinary (serialize.rs:92-121). */
#define SB_TYPE_BOOLEAN 0
#define SB_TYPE_INT8 1
#define SB_TYPE_INT16 2
#define SB_TYPE_INT32 3
#define SB_TYPE_INT64 4
#define SB_TYPE_UINT8 5
#define SB_TYPE_UINT16 6
#define SB_TYPE_UINT32 7
#define SB_TYPE_UINT64 8
#define SB_TYPE_INT128 9
#define SB_TYPE_INT256 10
#define SB_TYPE_FLOAT32 11
#define SB_TYPE_FLOAT64 12
#define SB_TYPE_BINARY 13       /* i32 offsets: Binary, Utf8 */
#define SB_TYPE_LARGE_BINARY 14 /* i64 offsets: LargeBinary, LargeUtf8 */
#define SB_TYPE_NULL 15

#define SB_MEM_DEVICE 0
#define SB_MEM_HOST 1

/* PageMeta (src/lib.rs:75-80): bytes of the page in the file, rows (flat columns) */
typedef struct sb_page_meta {
    uint64_t length;
    uint64_t num_values;
} sb_page_meta;

/* WriteOptions (src/write/common.rs:37-45) plus the two knobs a deterministic build needs
 * (SURVEY.md App. B#1): a forced codec (the reference only has debug-build env switches,
 * src/util/env.rs:20-24) and a seed for the sampling in compress_sample_ratio
 * (src/compression/integer/mod.rs:310-347 uses thread_rng()). */
typedef struct sb_write_options {
    int32_t default_compression;       /* CommonCompression: SB_CODEC_NONE/LZ4/ZSTD/SNAPPY */
    int32_t has_default_compress_ratio;/* Option<f64> discriminant */
    double default_compress_ratio;
    uint64_t max_page_size;            /* 0 = None */
    uint32_t forbidden_compressions;   /* bit (1u << codec id) set = forbidden */
    int32_t force_codec;               /* -1 = choose; else the page codec */
    int32_t force_index_codec;         /* -1 = choose; else codec of Dict indices / Freq exceptions */
    uint32_t flags;                    /* SB_WRITE_* bits, 0 = defaults */
    uint64_t rng_seed;                 /* column-level seed; page p uses mix64(seed ^ p*K) */
} sb_write_options;

/* sb_write_options.flags.
 * SB_WRITE_LZ4_EXACT: LZ4 blocks byte-identical to liblz4's LZ4_compress_default (what the reference's lz4 crate calls,
 * src/compression/basic.rs:108-120) — the serial greedy parse, ~10x slower on the device.  Default: a parallel
 * parse that emits a format-valid block every LZ4 decoder (hence the reference) reads back to the same bytes;
 * compressed-byte identity is library-version dependent upstream (no lockfile) and not part of the page layout. */
#define SB_WRITE_LZ4_EXACT 1u
/* SB_WRITE_DEBUG_VERIFY_FAIL (tests only): the string check behind the adaptive binary selector's hashed key count reports a
 * collision on every page, so that the exact re-selection and the exact dictionary build run; the bytes written are the same. */
#define SB_WRITE_DEBUG_VERIFY_FAIL (1u << 30)

typedef struct sb_ctx sb_ctx;

/* One context per host thread / stream, like one NativeWriter or column reader per thread
 * upstream (src/read/deserialize.rs:28: iterators are Send + Sync, no shared state).
 * `hip_stream` may be NULL (the context then creates its own non-blocking stream). */
int32_t sb_ctx_create(int32_t device, void* hip_stream, sb_ctx** out);
void sb_ctx_destroy(sb_ctx* ctx);
/* waits for all enqueued work; returns the first error raised by a kernel since the last
 * synchronize (status word in HBM) or by the runtime */
int32_t sb_ctx_synchronize(sb_ctx* ctx);
const char* sb_ctx_last_error(sb_ctx* ctx);
void* sb_ctx_stream(sb_ctx* ctx);

/* ------------------------------------------------------------------ decode
 * Replaces, per leaf column, batch_read::read_simple -> read_integer / read_double /
 * read_boolean / read_binary (src/read/batch_read.rs:27-64; src/read/array/integer.rs:210-238,
 * boolean.rs:191-219, binary.rs:223-265), i.e. per page read_validity
 * (src/read/read_basic.rs:36-63) + decompress_integer|double|boolean|binary
 * (src/compression/integer/mod.rs:72-117, double/mod.rs:69-114, boolean/mod.rs:63-102,
 * binary/mod.rs:95-183).  Output buffers receive the pages back to back, exactly what the
 * reference appends to its Vec<T> / MutableBitmap / offsets+values Vecs. */
typedef struct sb_column_read {
    int32_t physical_type;   /* SB_TYPE_* */
    int32_t is_nullable;     /* schema field nullable => pages carry a def-level section */
    const uint8_t* pages;    /* DEVICE: the column's pages, concatenated as in the file */
    uint64_t pages_len;
    const sb_page_meta* metas; /* HOST: ColumnMeta.pages */
    uint64_t n_pages;
    /* outputs (DEVICE).  Capacities in bytes; bitmaps are written in 32-bit words, so their buffers are
     * 4-byte aligned with a capacity of 4*ceil(rows/32) (SB_ERR_INVALID otherwise). */
    void* values;            /* primitives: rows*w; boolean: ceil(rows/8) bitmap bytes; binary: value bytes */
    uint64_t values_capacity;
    uint8_t* validity;       /* ceil(rows/8) bitmap bytes, LSB-first; required iff is_nullable */
    uint64_t validity_capacity;
    void* offsets;           /* binary: (rows+1) offsets of i32/i64 */
    uint64_t offsets_capacity;
    /* results (HOST, valid after sb_ctx_synchronize) */
    uint64_t rows;           /* sum of num_values */
    uint64_t values_len;     /* bytes produced in `values` */
    /* optional (HOST, n_pages entries): byte offset of every page's data inside `pages`.  NULL =
     * the pages are back to back.  Used for the leaf blocks of nested pages, which sit behind
     * their level sections (sb_nested_read_levels reports the offsets). */
    const uint64_t* page_offsets;
} sb_column_read;

/* Enqueue the decode of `n` columns.  Binary columns whose value bytes are not known in
 * advance: call sb_read_columns_sizes first, or pass a capacity that is large enough
 * (the reference guesses 4x the page bytes, src/read/array/binary.rs:241). */
int32_t sb_read_columns(sb_ctx* ctx, sb_column_read* cols, uint64_t n, int32_t mem);
/* Parses only the page headers on the device and fills rows / values_len (synchronous). */
int32_t sb_read_columns_sizes(sb_ctx* ctx, sb_column_read* cols, uint64_t n, int32_t mem);

/* ------------------------------------------------------------------ encode
 * Replaces, per leaf column, the page loop of NativeWriter::encode_chunk
 * (src/write/common.rs:54-109): page slicing, then per page write::write -> write_simple
 * (src/write/serialize.rs:36-132): write_validity (:200-215) + compress_integer|double|
 * boolean|binary (src/compression/integer/mod.rs:35-70, double/mod.rs:32-67,
 * boolean/mod.rs:23-61, binary/mod.rs:26-93), and the PageMeta bookkeeping (:102-107).
 * Output: the column's pages back to back (the bytes NativeWriter would write_all) and
 * one sb_page_meta per page. */
typedef struct sb_column_write {
    int32_t physical_type;
    int32_t is_nullable;
    uint64_t rows;
    const void* values;          /* DEVICE: values buffer / boolean bitmap / binary value bytes */
    uint64_t values_bit_offset;  /* boolean: bit offset of row 0 in `values` */
    uint64_t values_len;         /* binary: byte length of the whole values buffer (array.values().len()) */
    const uint8_t* validity;     /* DEVICE or NULL (no validity bitmap) */
    uint64_t validity_bit_offset;
    const void* offsets;         /* DEVICE: binary offsets, rows+1 entries */
    /* outputs */
    uint8_t* out_pages;          /* DEVICE, capacity >= sb_write_bound() */
    uint64_t out_capacity;
    sb_page_meta* out_metas;     /* HOST, capacity n_pages_capacity; valid after synchronize */
    uint64_t n_pages_capacity;
    /* results (HOST, valid after sb_ctx_synchronize) */
    uint64_t n_pages;
    uint64_t out_len;
    /* optional explicit paging (nested leaves): rows of every page instead of max_page_size, and
     * a head (the page's level section, produced by sb_nested_write_levels) that is placed in
     * front of every page's block.  page_rows / page_head_bytes: HOST arrays of n_pages_in
     * entries; page_heads: DEVICE, the heads back to back. */
    const uint64_t* page_rows;
    const uint64_t* page_head_bytes;
    const uint8_t* page_heads;
    uint64_t n_pages_in;
    /* index of this call's first page inside its column (0 for a whole column).  A rank that encodes pages
     * [first, first + n) of a column (SURVEY §8e work items) passes `first`, so that the per-page sampling seeds — and
     * hence the adaptive codec choice — are those of a single writer. */
    uint64_t first_page_index;
    /* binary columns: byte length of the COLUMN's whole values buffer when `values` holds only the bytes of this page
     * range (0 = values_len).  Upstream slices arrays page by page without slicing the values buffer, so
     * array.values().len() — which enters the selector's total_bytes (binary/mod.rs:270) and the hdr9 of Dict / Freq /
     * OneValue pages (binary/mod.rs:88) — is the length of the whole column's buffer. */
    uint64_t column_values_len;
} sb_column_write;

/* upper bound of the encoded size of a column, and its page count (A.4 page arithmetic) */
uint64_t sb_write_bound(int32_t physical_type, int32_t is_nullable, uint64_t rows, uint64_t values_len,
                        const sb_write_options* opts, uint64_t* n_pages);
int32_t sb_write_columns(sb_ctx* ctx, sb_column_write* cols, uint64_t n, const sb_write_options* opts,
                         int32_t mem);

/* ------------------------------------------------------------------ nested level sections
 * A nested page is `u32 page_rows | u32 rep_len | u32 def_len | rep | def | leaf BLOCK`
 * (src/write/serialize.rs:135-198).  The level sections are produced / consumed here; the leaf
 * BLOCK goes through sb_write_columns / sb_read_columns with explicit paging (page_rows +
 * page_heads on write, page_offsets on read), so every codec of the flat path applies.
 *
 * sb_nested_level describes one node on the path root -> leaf (arrow2 `Nested`, to_nested in
 * src/write/serialize.rs:135): kind 0 primitive leaf, 1 list (i32 offsets), 2 large list (i64
 * offsets), 3 struct.  All pointers are DEVICE memory. */
#define SB_NESTED_PRIMITIVE 0
#define SB_NESTED_LIST 1
#define SB_NESTED_LARGE_LIST 2
#define SB_NESTED_STRUCT 3
#define SB_NESTED_MAX_DEPTH 8
typedef struct sb_nested_level {
    const uint8_t* validity;      /* LSB-first bitmap or NULL (all valid) */
    const void* offsets;          /* lists: length + 1 entries */
    uint64_t validity_bit_offset;
    uint64_t length;              /* elements at this level (levels[0].length == rows) */
    int32_t kind;
    int32_t is_optional;
} sb_nested_level;
typedef struct sb_nested_page {
    uint64_t level_bytes;  /* 12 + rep_len + def_len */
    uint64_t num_values;   /* PageMeta.num_values of a nested page = level entries (write/common.rs:84-92) */
    uint64_t leaf_start;   /* slice of the leaf array this page carries (slice_parquet_array) */
    uint64_t leaf_count;
} sb_nested_page;
/* replaces write_nested_validity (src/write/serialize.rs:217-232) + the page cut of
 * write/common.rs:79-107 for one nested leaf column.  Writes every page's level section back to
 * back into out_levels (DEVICE) and fills pages[] (HOST).  Synchronises the context. */
uint64_t sb_nested_levels_bound(const sb_nested_level* levels, uint32_t n_levels, uint64_t rows,
                                uint64_t max_page_size);
int32_t sb_nested_write_levels(sb_ctx* ctx, const sb_nested_level* levels, uint32_t n_levels, uint64_t rows,
                               uint64_t max_page_size, uint8_t* out_levels, uint64_t out_capacity,
                               sb_nested_page* pages, uint64_t n_pages_capacity, uint64_t* n_pages_out);

/* The same for the leaf columns of a whole call (every leaf of every nested array of a chunk — encode_chunk's loop over
 * the leaves, src/write/common.rs:60-116): ONE set of launches over all pages of all leaves and ONE host round trip for
 * the page cut of the batch (32 bytes per page), instead of one per leaf column. */
typedef struct sb_nested_levels_write {
    const sb_nested_level* levels;  /* root -> leaf */
    uint32_t n_levels;
    uint32_t reserved;
    uint64_t rows;                  /* top-level rows (levels[0].length) */
    uint8_t* out_levels;            /* DEVICE: the level sections of the pages, back to back */
    uint64_t out_capacity;          /* >= sb_nested_levels_bound() */
    sb_nested_page* pages;          /* HOST, n_pages_capacity entries */
    uint64_t n_pages_capacity;
    uint64_t n_pages;               /* result */
} sb_nested_levels_write;
int32_t sb_nested_write_levels_batch(sb_ctx* ctx, sb_nested_levels_write* items, uint64_t n, uint64_t max_page_size);
/* The same without the round trip: enqueues the kernels and returns; items[].n_pages is set at once (host arithmetic),
 * items[].pages[] are filled by the next sb_ctx_synchronize.  `items` and what it points to must live until then.  A writer
 * that encodes chunk after chunk of one shape enqueues the level call and the leaf call (sb_write_columns with the page
 * cut of the chunk before) together and checks the cut afterwards (src/write/serialize.rs:217-232 runs them back to back). */
int32_t sb_nested_write_levels_enqueue(sb_ctx* ctx, sb_nested_levels_write* items, uint64_t n, uint64_t max_page_size);

/* replaces read_validity_nested (src/read/read_basic.rs:65-173) over all pages of one nested leaf
 * column.  Per level the caller passes DEVICE outputs: `offsets` (lists: column-level i64 offsets,
 * elements + 1 entries) and `validity` (nullable list/struct levels: LSB-first bitmap; 4-byte
 * aligned, capacity a multiple of 4 bytes); `length` returns the elements decoded at that level.
 * leaf_validity (DEVICE, same alignment rule) receives the leaf's validity when the leaf is
 * nullable.  page_leaf_counts / page_block_offsets (HOST, n_pages entries) return the number of
 * leaf slots of every page and the byte offset of its leaf BLOCK inside `pages`: feed them to
 * sb_read_columns as metas[].num_values / page_offsets.  Synchronises the context. */
typedef struct sb_nested_level_out {
    int64_t* offsets;
    uint8_t* validity;
    uint64_t offsets_capacity;   /* entries */
    uint64_t validity_capacity;  /* bytes */
    uint64_t length;             /* result */
    int32_t kind;
    int32_t is_nullable;
} sb_nested_level_out;
/* the leaf columns of a whole call: the arguments of sb_nested_read_levels per item, one set of launches, one round trip */
typedef struct sb_nested_levels_read {
    const uint8_t* pages;           /* DEVICE: the column's pages back to back */
    uint64_t pages_len;
    const sb_page_meta* metas;      /* HOST */
    uint64_t n_pages;
    sb_nested_level_out* levels;    /* HOST array of n_levels entries (DEVICE outputs inside); .length returns */
    uint32_t n_levels;
    uint32_t reserved;
    uint8_t* leaf_validity;         /* DEVICE or NULL */
    uint64_t leaf_validity_capacity;
    uint64_t* page_leaf_counts;     /* HOST, n_pages entries, or NULL */
    uint64_t* page_block_offsets;   /* HOST, n_pages entries, or NULL */
} sb_nested_levels_read;
int32_t sb_nested_read_levels_batch(sb_ctx* ctx, sb_nested_levels_read* items, uint64_t n);
/* The same without the round trip: levels[].length, page_leaf_counts and page_block_offsets are filled by the next
 * sb_ctx_synchronize; `items` and what it points to must live until then.  A reader that knows the page cut (a second
 * pass over the same pages) enqueues the level call and sb_read_columns together and compares afterwards. */
int32_t sb_nested_read_levels_enqueue(sb_ctx* ctx, sb_nested_levels_read* items, uint64_t n);
int32_t sb_nested_read_levels(sb_ctx* ctx, const uint8_t* pages, uint64_t pages_len, const sb_page_meta* metas,
                              uint64_t n_pages, sb_nested_level_out* levels, uint32_t n_levels,
                              uint8_t* leaf_validity, uint64_t leaf_validity_capacity,
                              uint64_t* page_leaf_counts, uint64_t* page_block_offsets);

/* ------------------------------------------------------------------ file framing (host only)
 * "ARROW2" 00 00 | column pages ... | schema bytes | metas | u32 schema_size | u32 meta_size | EOS
 * (SURVEY App. A.1).  Writer: NativeWriter::start / write / finish (src/write/writer.rs:91-167);
 * one sb_file_writer_write_column per leaf column in leaf order, with the pages and PageMeta that
 * sb_write_columns produced (copied to HOST memory).  The schema is an opaque Arrow IPC Schema
 * message flatbuffer (arrow2 schema_to_bytes upstream).  Errors carry the reference's messages;
 * sb_file_last_error() is thread-local. */
typedef struct sb_file_writer sb_file_writer;
typedef struct sb_file_reader sb_file_reader;
const char* sb_file_last_error(void);
int32_t sb_file_writer_open(const char* path, sb_file_writer** out);
int32_t sb_file_writer_start(sb_file_writer* w);
int32_t sb_file_writer_write_column(sb_file_writer* w, const uint8_t* pages, uint64_t pages_len,
                                    const sb_page_meta* metas, uint64_t n_pages);
int32_t sb_file_writer_finish(sb_file_writer* w, const uint8_t* schema_bytes, uint64_t schema_len,
                              uint64_t* total_size);
void sb_file_writer_close(sb_file_writer* w);
/* Reader: read_meta / deserialize_meta (src/read/reader.rs:148-178), the schema bytes of
 * infer_schema (:227-241), and page ranges of a column (ColumnMeta::slice src/lib.rs:47-61 +
 * NativeReader::next src/read/reader.rs:117-127) into HOST memory. */
int32_t sb_file_reader_open(const char* path, sb_file_reader** out);
uint64_t sb_file_reader_n_columns(const sb_file_reader* r);
int32_t sb_file_reader_column(const sb_file_reader* r, uint64_t col, uint64_t* offset, uint64_t* n_pages,
                              const sb_page_meta** pages);
int32_t sb_file_reader_schema(const sb_file_reader* r, const uint8_t** bytes, uint64_t* len);
int32_t sb_file_reader_read_pages(sb_file_reader* r, uint64_t col, uint64_t first_page, uint64_t n_pages,
                                  uint8_t* dst, uint64_t capacity, uint64_t* bytes_read);
void sb_file_reader_close(sb_file_reader* r);

/* ------------------------------------------------------------------ schema bytes of the footer (host only)
 * NativeWriter::finish stores arrow2's schema_to_bytes(&schema, &default_ipc_fields(..)) (src/write/writer.rs:137-139):
 * the bare Arrow IPC Message flatbuffer with a Schema header (version V5, little endian, body_length 0); infer_schema
 * feeds it to deserialize_schema (src/read/reader.rs:227-241).  These two calls write / read that flatbuffer without
 * another Arrow implementation.  A schema is passed as its fields flattened in PRE-ORDER: every entry is followed by
 * its n_children children (List / LargeList / FixedSizeList: the item field; Struct: its fields; Map: the non-null
 * "entries" struct with the key and value fields).  type_id is the `Type` union tag of Schema.fbs. */
#define SB_ARROW_NULL 1
#define SB_ARROW_INT 2              /* bit_width 8/16/32/64, is_signed */
#define SB_ARROW_FLOATING_POINT 3   /* precision: 0 half, 1 single, 2 double */
#define SB_ARROW_BINARY 4
#define SB_ARROW_UTF8 5
#define SB_ARROW_BOOL 6
#define SB_ARROW_DECIMAL 7          /* precision, scale, bit_width 128 / 256 */
#define SB_ARROW_DATE 8             /* unit: 0 day (Date32), 1 millisecond (Date64) */
#define SB_ARROW_TIME 9             /* unit (TimeUnit), bit_width 32 / 64 */
#define SB_ARROW_TIMESTAMP 10       /* unit: 0 s, 1 ms, 2 us, 3 ns; timezone or NULL */
#define SB_ARROW_INTERVAL 11        /* unit: 0 year_month, 1 day_time, 2 month_day_nano */
#define SB_ARROW_LIST 12
#define SB_ARROW_STRUCT 13
#define SB_ARROW_FIXED_SIZE_BINARY 15 /* bit_width = byte width */
#define SB_ARROW_FIXED_SIZE_LIST 16   /* bit_width = list size */
#define SB_ARROW_MAP 17             /* is_signed = keys_sorted */
#define SB_ARROW_DURATION 18        /* unit (TimeUnit) */
#define SB_ARROW_LARGE_BINARY 19
#define SB_ARROW_LARGE_UTF8 20
#define SB_ARROW_LARGE_LIST 21
typedef struct sb_schema_field {
    const char* name;     /* UTF-8, NUL terminated */
    const char* timezone; /* Timestamp only; NULL = none */
    int32_t type_id;      /* SB_ARROW_* */
    int32_t nullable;
    int32_t n_children;
    int32_t bit_width;
    int32_t is_signed;
    int32_t precision;
    int32_t scale;
    int32_t unit;
    const char* metadata; /* Field.custom_metadata: n_metadata pairs packed as "key\0value\0key\0value\0..."; NULL = none */
    uint64_t n_metadata;
} sb_schema_field;
const char* sb_schema_last_error(void);
/* metadata: 2 * n_metadata strings (key, value, key, value ...) of Schema.custom_metadata, or NULL.  *len returns the
 * size of the flatbuffer (also when `capacity` is too small: SB_ERR_INVALID, call again). */
int32_t sb_schema_to_bytes(const sb_schema_field* fields, uint64_t n_fields, uint64_t n_top, const char* const* metadata,
                           uint64_t n_metadata, uint8_t* out, uint64_t capacity, uint64_t* len);
/* Parses schema bytes written by this library, arrow2 or any other Arrow implementation.  Field names / timezones are
 * copied into `strings`; the out[].name pointers point there.  *n_fields / *strings_len return the sizes needed. */
int32_t sb_schema_from_bytes(const uint8_t* bytes, uint64_t len, sb_schema_field* out, uint64_t capacity, uint64_t* n_fields,
                             uint64_t* n_top, char* strings, uint64_t strings_capacity, uint64_t* strings_len);
/* Schema.custom_metadata of the same bytes (arrow2's deserialize_schema keeps it, src/read/reader.rs:227-241): *n_pairs
 * pairs packed into `strings` as "key\0value\0..." (*strings_len bytes; SB_ERR_INVALID + sizes when it does not fit). */
int32_t sb_schema_metadata_from_bytes(const uint8_t* bytes, uint64_t len, char* strings, uint64_t strings_capacity,
                                      uint64_t* n_pairs, uint64_t* strings_len);

/* ------------------------------------------------------------------ page inspector (host only)
 * Replaces stat::stat_simple / stat_body / stat_dict_body / stat_freq_body (src/stat.rs:61-152): the
 * block structure of one page without decoding it.  `out` receives a chain: out[0] is the page's block,
 * out[k + 1] the block nested in out[k] (the u32 indices of a Dict block, the exceptions of a primitive
 * Freq block), PageInfo / DictPageBody / FreqPageBody flattened. */
typedef struct sb_page_info {
    int32_t codec;                   /* SB_CODEC_*: PageBody variant */
    int32_t has_validity_size;       /* PageInfo.validity_size is Some (nullable field, out[0] only) */
    uint32_t validity_size;          /* as upstream reports it: the u32 FOLLOWING the def-level section, i.e. the
                                        first 4 bytes of the block header (src/stat.rs:72-76), not the section's size */
    uint32_t compressed_size;
    uint32_t uncompressed_size;
    uint32_t unique_num;             /* Dict: DictPageBody.unique_num */
    uint32_t exceptions_bitmap_size; /* Freq: FreqPageBody.exceptions_bitmap_size */
    int32_t has_nested;              /* the next entry of `out` describes this block's nested block */
} sb_page_info;
/* Returns SB_OK and the chain length in *n_out (<= capacity); SB_ERR_IO when the page is shorter than
 * its headers say (upstream: slice panic), SB_ERR_OUT_OF_SPEC for an unknown codec id. */
int32_t sb_stat_page(const uint8_t* page, uint64_t length, int32_t physical_type, int32_t is_nullable,
                     sb_page_info* out, uint32_t capacity, uint32_t* n_out);

/* ------------------------------------------------------------------ measurement
 * Optional per-kernel timing with HIP events recorded on the context's stream around every
 * kernel launch (used by bench.py for the roofline figure).  Totals are accumulated at
 * sb_ctx_synchronize(). */
typedef struct sb_kernel_stat {
    const char* name;   /* kernel name as it appears in rocprofv3 traces (without namespace) */
    uint64_t launches;
    double total_ms;
} sb_kernel_stat;
int32_t sb_ctx_profile(sb_ctx* ctx, int32_t enable);  /* enable/disable; resets the totals */
uint32_t sb_ctx_profile_read(sb_ctx* ctx, sb_kernel_stat* out, uint32_t cap);
/* Totals of the block-parallel Zstd decoder (basic.rs:93-97: libzstd frames are decoded block by block, all blocks of a
 * call at once) since the context was created: out[0] frames it decoded, out[1] frames handed back to the frame-serial
 * decoder (pools exhausted, or a malformed stream whose error that decoder names), out[2] blocks, out[3] sequences.
 * Synchronises the context. */
int32_t sb_ctx_zstd_block_stats(sb_ctx* ctx, uint64_t out[4]);

/* Calls since the context was created whose kernels were spread over side streams next to the call's stream (a mixed
 * schema in one adaptive sb_write_columns call: the binary chain beside the primitive kinds; sb_read_columns with
 * primitive and binary pages; the Zstd block pipeline).  Diagnostics only: the parity tests use it to show that the
 * multi-stream path — whose kernels exchange a page's scratch areas across streams through fork / join events — is
 * the one that ran.  Does not synchronise. */
uint64_t sb_ctx_side_forks(sb_ctx* ctx);

/* Synchronize intervals since the context was created that were issued a second time: a call skips kernels that the
 * previous calls of the same shape did not need (long-page Dict / Freq writers, the block-parallel LZ4 reader, emitters of
 * codecs no page chose); when a page needs one after all it is left undone, and sb_ctx_synchronize re-issues the
 * interval's sb_write_columns / sb_read_columns calls with everything launched before it returns.  Results are the same
 * either way; the counter lets tests show which path ran.  Does not synchronise. */
uint64_t sb_ctx_replays(sb_ctx* ctx);

/* version / build info string ("strawboat-hip <ver> gfx950") */
const char* sb_version(void);

#ifdef __cplusplus
}
#endif
#endif /* STRAWBOAT_HIP_H */
