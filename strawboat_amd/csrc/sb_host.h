// strawboat-hip: host-side context (workspace, staging ring, error plumbing).
#pragma once
#include <string>
#include <algorithm>
#include <deque>
#include <vector>

#include "sb_common.h"

namespace sb {

struct DevBuf {
    uint8_t* p = nullptr;
    size_t cap = 0;
};

// pinned host staging slot: uploads (tables) and readbacks (status, sizes, metas)
struct StageSlot {
    uint8_t* host = nullptr;
    size_t cap = 0;
    hipEvent_t done = nullptr;
    bool in_flight = false;
};

struct Pending {  // results to hand back to the caller's structs at synchronize
    enum Kind { READ_COL, WRITE_COL, ENC_HINT, NESTED_W, NESTED_R } kind;   // ENC_HINT: the codec counts of a write call (n = the plan's key), 32 words
                                                                           // NESTED_W / _R: the page records of an enqueued level call (user = its items, n = how many)
    void* user;            // sb_column_read* / sb_column_write*
    const uint8_t* host;   // where the readback lands (pinned)
    uint64_t n;            // WRITE_COL: number of pages
    uint64_t bytes = 0;    // NESTED_*: bytes of the readback at `host`
};

enum KernelId {
    K_PARSE, K_INFLATE_A, K_PLAN, K_COLSCAN, K_INFLATE_B, K_EXPAND, K_EXPAND_BIN,
    K_ENC_TILES, K_ENC_PAGES, K_ENC_LAYOUT, K_ENC_COMPACT, K_ENC_SELECT, K_ENC_LZ4,
    K_ENC_PAGES_DICT, K_ENC_PAGES_ONEVALUE, K_ENC_PAGES_BP, K_EXPAND_RLE, K_ENC_PAGES_PATAS, K_ENC_FREQ, K_ENC_SELECT_ROWS, K_COUNT
};
struct ProfSpan {
    int id;
    hipEvent_t a, b;
};

}  // namespace sb

struct sb_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string last_error;
    int32_t sticky = 0;  // first host-side error since the last synchronize

    int lzg_state = 0;        // sb_lz4_giant.h: 0 = not known yet, 1 = this context meets LZ4 blocks of megabytes, 2 = it does not
    bool lzg_long_pages = false;   // a call since the last synchronize had pages long enough
    uint32_t lzg_idle = 0;         // intervals in a row with long pages and no LZ4 block of megabytes
    sb::DevBuf lzg_pool;   // sb_lz4_giant.h: tables and entries of LZ4 blocks of megabytes
    sb::DevBuf tables;   // ColDesc / PageTask / PageDesc / TileTask / jobs / counters
    sb::DevBuf scratch;  // per-page aux + inflate areas, encode slots
    sb::DevBuf staging;  // device staging for SB_MEM_HOST callers
    sb::DevBuf zlit;     // Zstd literal buffers (fixed pool of inflate waves)
    sb::DevBuf zrec;     // Zstd sequence records (one arena per inflate wave; allocated by the first batch-sized read)
    sb::Status* d_status = nullptr;
    sb::Status* h_status = nullptr;  // pinned
    // the block-parallel Zstd pipeline (sb_zstd_blocks.h): pools sized per call once the context has met Zstd pages
    // (Status.kinds, read back at every synchronize); SB_ZSTD_BLOCKS = 0 / 1 in the environment forces it off / on
    sb::DevBuf zb_blocks, zb_lit, zb_rec;
    unsigned long long* zb_stats = nullptr;   // device: sb_ctx_zstd_block_stats
    uint32_t kinds_seen = 0;
    bool zstd_recent = false;   // the read calls of the last synchronize interval met a Zstd buffer
    uint32_t read_calls = 0;     // read calls since the last synchronize
    uint32_t qa_idle = 0, tiles_idle = 0;   // read intervals in a row that queued no inflate job for queue A / no tile task: from two
                                            // on the next read calls leave those kernels out (a reader that alternates column kinds keeps them)
    uint32_t zstd_idle = 0;      // read intervals in a row that met no Zstd buffer
    bool zb_seq_long = false;    // the last Zstd calls held blocks of >= 8192 sequences: zb_seq is submitted before zb_lit
    int zb_mode = 2;             // 0 off, 1 always, 2 once Zstd has been seen
    uint32_t zb_wg_exec = 1;     // SB_ZSTD_BLOCKS_WG=0: frames of many short sequences through the wave executor too
    uint32_t zb_pool_div = 1;    // SB_ZSTD_BLOCKS_POOL_DIV (tests): pool estimates divided by this
    uint32_t zb_min_csize = 0;   // SB_ZSTD_BLOCKS_MIN: frames shorter than this keep the one-wave / lane-per-frame paths

    static constexpr int NSLOTS = 8;
    sb::StageSlot slots[NSLOTS];
    int next_slot = 0;
    size_t slot_cap_max = 0;                // largest staging request so far: a slot that grows, grows to this
    std::vector<void*> stale_host;          // outgrown staging buffers, freed at the next synchronize (hipHostFree drains the device)

    std::vector<sb::Pending> pending;
    std::deque<std::vector<uint8_t>> rescued;  // readbacks moved out of a recycled staging slot (acquire_slot)
    // SB_MEM_HOST copies to hand back after the stream drains: (host dst, device src, bytes)
    struct Copyback {
        void* host;
        const void* dev;
        size_t n;
        const uint64_t* used = nullptr;  // when set: only the first min(n, *used) bytes are copied (pages: out_len)
        bool issued = false;             // already on its way (copy stream): a call of many columns is cut into groups whose
                                         // copies back overlap the next group's copies in — the link's two directions are independent
    };
    uint32_t host_groups_max = 8;        // SB_HOST_GROUPS (1: SB_MEM_HOST calls are never cut into groups)
    hipStream_t copy_stream = nullptr;   // D2H copies of SB_MEM_HOST calls
    std::vector<hipEvent_t> pipe_ev;     // "group's kernels done" events of the open interval
    size_t pipe_ev_used = 0;
    sb::StageSlot* last_slot = nullptr;  // the staging slot of the last enqueued write call (its `done` event = results are back)
    bool freq_pass_ran = false;          // this synchronize ran the Freq second pass (copies issued early are repeated)
    hipEvent_t next_pipe_event() {
        if (pipe_ev_used == pipe_ev.size()) {
            hipEvent_t e = nullptr;
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
            pipe_ev.push_back(e);
        }
        return pipe_ev[pipe_ev_used++];
    }
    hipStream_t copy_stream_get() {
        if (!copy_stream && hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking) != hipSuccess) copy_stream = nullptr;
        return copy_stream;
    }
    std::vector<Copyback> copybacks;
    std::vector<void*> temp_dev;  // device temporaries to free at synchronize
    // SB_MEM_HOST staging areas: carved out of chunks that stay allocated (hipMalloc / hipFree per buffer and call
    // cost more than the PCIe copies); every carve-out lives until the next synchronize, which rewinds the chunks
    struct StageChunk {
        uint8_t* p;
        size_t cap, used;
    };
    std::vector<StageChunk> stage_chunks;
    size_t stage_hint = 0;  // capacity of the set of chunks that was last merged
    uint8_t* stage_alloc(size_t bytes) {
        bytes = (bytes + 64 + 255) & ~(size_t)255;
        for (auto& c : stage_chunks)
            if (c.cap - c.used >= bytes) {
                uint8_t* r = c.p + c.used;
                c.used += bytes;
                return r;
            }
        size_t total = 0;
        for (auto& c : stage_chunks) total += c.cap;
        const size_t cap = std::max<size_t>(std::max<size_t>(bytes, stage_hint), std::max<size_t>(total, (size_t)256 << 20));  // doubling growth
        uint8_t* p = nullptr;
        if (hipMalloc((void**)&p, cap) != hipSuccess) return nullptr;
        stage_chunks.push_back({p, cap, bytes});
        return p;
    }
    void stage_rewind() {  // after the stream has drained: one chunk of the total size replaces a fragmented set
        if (stage_chunks.size() > 2) {
            stage_hint = 0;
            for (auto& c : stage_chunks) {
                stage_hint += c.cap;
                (void)hipFree(c.p);
            }
            stage_chunks.clear();
        }
        for (auto& c : stage_chunks) c.used = 0;
    }
    void stage_release() {
        for (auto& c : stage_chunks) (void)hipFree(c.p);
        stage_chunks.clear();
    }
    // Freq pages logged by the decode calls since the last synchronize: [u32 count | pad][FreqEntry...]
    struct FreqLog {
        uint8_t* dev = nullptr;
        uint32_t cap = 0, reserved = 0;
    };
    std::vector<FreqLog> freq_logs;
    bool in_freq_pass = false;
    std::vector<sb_column_read> freq_cols;  // the second pass's one-page columns (alive until synchronize returns)
    std::vector<sb_page_meta> freq_metas;

    // optional per-kernel HIP-event timing (sb_ctx_profile)
    bool profile = false;
    std::vector<sb::ProfSpan> spans;
    std::vector<hipEvent_t> free_events;
    // The page table of the last write call (sb_write_columns): page arithmetic, scratch layout and launch shape depend only
    // on the columns' types, row counts, paging and the options — not on their buffers — so a writer that sends chunk after
    // chunk of one schema (or a benchmark that repeats a call) re-uses the device-resident table and the plan instead of
    // building and uploading ~170 bytes per page again (65 536 pages of 8192 booleans: 2 ms of host work per call).
    struct EncPlan {
        bool valid = false;
        uint64_t key = 0, n = 0;
        uint64_t P = 0, max_tiles = 1, max_chunks = 1, lz_cap = 0, lz_chunk = 0;
        bool any_tiles = false, any_pages = false, any_compact = false, any_lz4 = false;
        size_t scratch_total = 0, lz_pool_off = 0, zpar_off = 0;
        std::vector<uint64_t> key_words;   // the words the key was hashed from (compared on a hit)
        std::vector<uint32_t> col_first, col_pages;
        std::vector<uint64_t> hro;
        sb::DevBuf pages;   // EncPage[P] on the device
        // long pages of 4- / 8-byte values (adaptive calls): selected section-parallel (sb_select_big.h); page indices on
        // the device as [1-byte | 2-byte | 4-byte | 8-byte pages], grid.x = the most sections any page of a width has
        std::vector<uint32_t> bigw[5];          // by log2(width): 1-, 2-, 4-, 8-byte values; [4]: binary pages
        uint32_t big_secs[5] = {0, 0, 0, 0, 0};
        sb::DevBuf big;
        // pages per codec the last TWO completed calls with this plan chose (element-wise maximum; read back with the results):
        // a call leaves out the launches neither of them needed — a page that needs one after all stays unwritten and the
        // interval is replayed (k_enc_layout, sb_ctx_synchronize).  Two calls, not one: a writer that alternates two kinds
        // of data under one plan keeps the kernels of both instead of replaying every other call.
        uint32_t last_counts[32] = {0}, prev_counts[32] = {0};
        bool counts_valid = false;
        // binary pages of an adaptive call: any of them, and any that k_enc_bin_page (sb_bin_page.h) does not take (too
        // short / long): only those still need the hash -> select -> verify chain
        bool bin_pages = false, bin_unfused = false;
    } enc_plan;
    // Calls of the open synchronize interval, kept so that they can be issued again: kernels whose work the last calls did
    // not need are skipped on that hint (the long-page Dict / Freq chains, the block-parallel LZ4 reader, emitters of
    // codecs nobody chose); a page that needed one after all is left undone, the device says so (KIND_REPLAY) and
    // sb_ctx_synchronize re-issues the interval's calls with every kernel launched — a wrong guess costs one extra pass,
    // never a one-workgroup walk over a million-row page.  (The callers' column arrays and buffers live until the
    // synchronize anyway: the results are written into them there.)
    struct Call {
        int kind;   // 0 read, 1 write
        void* cols;
        uint64_t n;
        sb_write_options opts;
        int32_t mem;
    };
    std::vector<Call> calls;
    bool no_hints = false, in_replay = false;   // no_hints: SB_NO_HINTS=1, and during a replay
    uint64_t replays = 0;                       // sb_ctx_replays
    bool bin_fused = true;   // SB_BIN_FUSED=0: binary pages through the round-3 chain (A/B measurements, tests)
    std::vector<uint64_t> enc_plan_probe;   // the key words of the call at hand
    // side streams: kernels of a call that work on disjoint pages (the selector / emit chains of different column kinds, the
    // three expand kernels of a read) run side by side between a fork and a join on `stream`; seen from outside the call is
    // still one span of work on `stream`.  Not used while profiling (the per-kernel events bracket launches on `stream`).
    static constexpr int NSIDE = 3;
    hipStream_t side[NSIDE] = {nullptr, nullptr, nullptr};
    hipEvent_t fork_ev = nullptr, join_ev[NSIDE] = {nullptr, nullptr, nullptr};
    uint64_t side_forks = 0;   // calls whose kernels ran on side streams next to the call's stream (sb_ctx_side_forks)
    bool side_ready = false;
    struct ProfEntry {
        std::string name;  // the kernel's name as rocprofv3 prints it, without "void sb::" and the argument list
        double ms = 0;
        uint64_t n = 0;
    };
    std::vector<ProfEntry> prof;  // [0, K_COUNT): the KernelId entries; template instances are added by name
    int prof_id(const char* name) {
        for (size_t i = 0; i < prof.size(); i++)
            if (prof[i].name == name) return (int)i;
        ProfEntry e;
        e.name = name;
        prof.push_back(e);
        return (int)prof.size() - 1;
    }

    int32_t fail(int32_t code, const std::string& msg) {
        last_error = msg;
        if (!sticky) sticky = code;
        return code;
    }
};

namespace sb {
// brackets one kernel launch with events when profiling is on
struct KScope {
    sb_ctx* ctx;
    ProfSpan sp;
    KScope(sb_ctx* c, const char* name) : KScope(c, c->profile ? c->prof_id(name) : 0) {}
    KScope(sb_ctx* c, int id) : ctx(c) {
        if (!ctx->profile) return;
        sp.id = id;
        auto get = [&]() {
            hipEvent_t e;
            if (!ctx->free_events.empty()) {
                e = ctx->free_events.back();
                ctx->free_events.pop_back();
            } else {
                (void)hipEventCreate(&e);
            }
            return e;
        };
        sp.a = get();
        sp.b = get();
        (void)hipEventRecord(sp.a, ctx->stream);
    }
    ~KScope() {
        if (!ctx->profile) return;
        (void)hipEventRecord(sp.b, ctx->stream);
        ctx->spans.push_back(sp);
    }
};
bool ensure(sb_ctx* ctx, DevBuf& b, size_t need);
// side streams (created on first use); fork: they wait for everything queued on ctx->stream so far; join: ctx->stream waits
// for the side streams listed in `used_mask`
bool side_streams(sb_ctx* ctx);
void side_fork(sb_ctx* ctx, uint32_t used_mask);
void side_join(sb_ctx* ctx, uint32_t used_mask);
StageSlot* acquire_slot(sb_ctx* ctx, size_t need);
int32_t check_hip(sb_ctx* ctx, hipError_t e, const char* what);
}  // namespace sb

// sb_nested.hip: the page records of an enqueued level call -> the caller's structs (Pending::NESTED_W / _R)
void nested_write_finish(sb_nested_levels_write* items, uint64_t n, const uint8_t* host);
void nested_read_finish(sb_nested_levels_read* items, uint64_t n, const uint8_t* host);
