// strawboat-hip: C ABI (include/strawboat_hip.h) — context management and the decode entry
// points.  Host logic only: page bookkeeping (the arithmetic of read_integer & co: running row
// offset per page, src/read/array/integer.rs:210-238), table upload, kernel launches.
#include <cstdio>
#include <cstring>

#include "sb_host.h"

namespace sb {

void launch_decode(sb_ctx* ctx, const DecodeArgs& a, bool any_binary, bool any_prim, uint64_t* col_values_len);
void launch_parse_sizes(sb_ctx* ctx, const DecodeArgs& a, uint64_t* col_values_len);
void launch_freq_scatter(sb_ctx* ctx, const FreqEntry* entries, uint32_t n, const uint64_t* ex_off, const uint8_t* ex_base);

// names as rocprofv3 prints them (template instances are registered by name at their launch sites)
static const char* const KERNEL_NAMES[K_COUNT] = {"k_parse", "k_inflate", "k_plan", "k_colscan", "k_inflate(values)",
                                                  "k_expand", "k_expand_binary", "k_enc_emit_tiles",
                                                  "k_enc_emit_pages<RLE>", "k_enc_layout", "k_enc_compact", "k_enc_select", "k_enc_emit_lz4",
                                                  "k_enc_emit_pages<Dict>", "k_enc_emit_pages<OneValue>",
                                                  "k_enc_emit_pages<Bitpacking>", "k_expand_rle", "k_enc_emit_pages<Patas>", "k_enc_freq_prep/finish", "k_enc_select_rle"};

int32_t check_hip(sb_ctx* ctx, hipError_t e, const char* what) {
    if (e == hipSuccess) return SB_OK;
    return ctx->fail(SB_ERR_EXTERNAL, std::string(what) + ": " + hipGetErrorString(e));
}

bool ensure(sb_ctx* ctx, DevBuf& b, size_t need) {
    if (need <= b.cap) return true;
    // earlier launches may still use the old buffer
    if (b.p) {
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) return false;
        (void)hipFree(b.p);
        b.p = nullptr;
        b.cap = 0;
    }
    size_t cap = need + need / 4 + 4096;
    if (hipMalloc((void**)&b.p, cap) != hipSuccess) {
        b.p = nullptr;
        return false;
    }
    b.cap = cap;
    return true;
}

bool side_streams(sb_ctx* ctx) {
    if (ctx->side_ready) return true;
    int lo_prio = 0, hi_prio = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo_prio, &hi_prio);
    for (int i = 0; i < sb_ctx::NSIDE; i++) {
        // side[0] carries the longest chain of a mixed call (binary columns): highest priority, so that its workgroups are
        // placed first when several kernels compete for the CUs
        if (hipStreamCreateWithPriority(&ctx->side[i], hipStreamNonBlocking, i == 0 ? hi_prio : lo_prio) != hipSuccess) return false;
        if (hipEventCreateWithFlags(&ctx->join_ev[i], hipEventDisableTiming) != hipSuccess) return false;
    }
    if (hipEventCreateWithFlags(&ctx->fork_ev, hipEventDisableTiming) != hipSuccess) return false;
    ctx->side_ready = true;
    return true;
}
void side_fork(sb_ctx* ctx, uint32_t used_mask) {
    if (used_mask) ctx->side_forks++;
    (void)hipEventRecord(ctx->fork_ev, ctx->stream);
    for (int i = 0; i < sb_ctx::NSIDE; i++)
        if ((used_mask >> i) & 1) (void)hipStreamWaitEvent(ctx->side[i], ctx->fork_ev, 0);
}
void side_join(sb_ctx* ctx, uint32_t used_mask) {
    for (int i = 0; i < sb_ctx::NSIDE; i++)
        if ((used_mask >> i) & 1) {
            (void)hipEventRecord(ctx->join_ev[i], ctx->side[i]);
            (void)hipStreamWaitEvent(ctx->stream, ctx->join_ev[i], 0);
        }
}

StageSlot* acquire_slot(sb_ctx* ctx, size_t need) {
    StageSlot& s = ctx->slots[ctx->next_slot];
    ctx->next_slot = (ctx->next_slot + 1) % sb_ctx::NSLOTS;
    if (s.in_flight) {
        (void)hipEventSynchronize(s.done);
        s.in_flight = false;
    }
    // Results of earlier calls that were read back into this slot and not yet handed to their callers
    // (more than NSLOTS calls since the last synchronize): move them to heap storage owned by the
    // Pending entry before the slot is rewritten or freed.
    if (s.host)
        for (auto& p : ctx->pending) {
            if (p.host < s.host || p.host >= s.host + s.cap) continue;
            const size_t nb = p.kind == Pending::READ_COL ? 8 : p.kind == Pending::ENC_HINT ? 128 : (p.kind == Pending::NESTED_W || p.kind == Pending::NESTED_R) ? (size_t)p.bytes
                                                                                                   : (size_t)(2 * p.n + 1) * 8;
            ctx->rescued.emplace_back(p.host, p.host + nb);
            p.host = ctx->rescued.back().data();
        }
    if (s.cap < need) {
        // hipHostFree waits for the device: inside an enqueue it would drain the pipeline once per slot whenever a context
        // moves on to larger calls (the eight slots outgrown one after the other: +0.25 ms on each of the next eight calls)
        if (s.host) ctx->stale_host.push_back(s.host);
        ctx->slot_cap_max = std::max(ctx->slot_cap_max, need + need / 4 + 4096);
        size_t cap = ctx->slot_cap_max;
        if (hipHostMalloc((void**)&s.host, cap, hipHostMallocDefault) != hipSuccess) {
            s.host = nullptr;
            s.cap = 0;
            return nullptr;
        }
        s.cap = cap;
    }
    if (!s.done) (void)hipEventCreateWithFlags(&s.done, hipEventDisableTiming);
    return &s;
}

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static uint32_t type_width(int32_t t) {
    switch (t) {
        case SB_TYPE_INT8:
        case SB_TYPE_UINT8:
            return 1;
        case SB_TYPE_INT16:
        case SB_TYPE_UINT16:
            return 2;
        case SB_TYPE_INT32:
        case SB_TYPE_UINT32:
        case SB_TYPE_FLOAT32:
        case SB_TYPE_BINARY:
            return 4;
        case SB_TYPE_INT64:
        case SB_TYPE_UINT64:
        case SB_TYPE_FLOAT64:
        case SB_TYPE_LARGE_BINARY:
            return 8;
        case SB_TYPE_INT128:
            return 16;
        case SB_TYPE_INT256:
            return 32;
    }
    return 0;
}
static bool is_binary_t(int32_t t) { return t == SB_TYPE_BINARY || t == SB_TYPE_LARGE_BINARY; }

static const char* status_text(const Status& s, char* buf, size_t n) {
    const char* kind = s.code == SB_ERR_OUT_OF_SPEC ? "OutOfSpec"
                       : s.code == SB_ERR_EXTERNAL  ? "External"
                       : s.code == SB_ERR_IO        ? "Io(UnexpectedEof)"
                       : s.code == SB_ERR_NYI       ? "NotYetImplemented"
                                                    : "InvalidArgument";
    snprintf(buf, n, "%s raised on the device: page %u, site %u", kind, s.page, s.where);
    return buf;
}

}  // namespace sb

using namespace sb;

extern "C" {

const char* sb_version(void) { return "strawboat-hip 0.1 gfx950"; }

int32_t sb_ctx_create(int32_t device, void* hip_stream, sb_ctx** out) {
    if (!out) return SB_ERR_INVALID;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return SB_ERR_EXTERNAL;
    if (hipSetDevice(device) != hipSuccess) return SB_ERR_EXTERNAL;
    sb_ctx* ctx = new sb_ctx();
    ctx->device = device;
    for (int i = 0; i < K_COUNT; i++) ctx->prof_id(KERNEL_NAMES[i]);
    if (hip_stream) {
        ctx->stream = (hipStream_t)hip_stream;
    } else {
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
            delete ctx;
            return SB_ERR_EXTERNAL;
        }
        ctx->own_stream = true;
    }
    if (hipMalloc((void**)&ctx->d_status, sizeof(Status)) != hipSuccess ||
        hipHostMalloc((void**)&ctx->h_status, sizeof(Status), hipHostMallocDefault) != hipSuccess) {
        sb_ctx_destroy(ctx);
        return SB_ERR_EXTERNAL;
    }
    (void)hipMemsetAsync(ctx->d_status, 0, sizeof(Status), ctx->stream);
    if (hipMalloc((void**)&ctx->zb_stats, 16 * sizeof(unsigned long long)) != hipSuccess) {
        sb_ctx_destroy(ctx);
        return SB_ERR_EXTERNAL;
    }
    (void)hipMemsetAsync(ctx->zb_stats, 0, 16 * sizeof(unsigned long long), ctx->stream);
    if (const char* e = getenv("SB_ZSTD_BLOCKS")) ctx->zb_mode = e[0] == '0' ? 0 : e[0] == '1' ? 1 : 2;
    if (const char* e = getenv("SB_ZSTD_BLOCKS_WG")) ctx->zb_wg_exec = e[0] != '0';
    if (const char* e = getenv("SB_BIN_FUSED")) ctx->bin_fused = e[0] != '0';
    if (const char* e = getenv("SB_NO_HINTS")) ctx->no_hints = e[0] != '0';
    if (const char* e = getenv("SB_HOST_GROUPS")) ctx->host_groups_max = std::max<uint32_t>(1, (uint32_t)strtoul(e, nullptr, 10));
    if (const char* e = getenv("SB_ZSTD_BLOCKS_MIN")) ctx->zb_min_csize = (uint32_t)strtoul(e, nullptr, 10);
    // tests: divide the block pipeline's pool estimates so that a call runs out of pool space part-way (frames that do not
    // fit go back to the frame-serial decoder)
    if (const char* e = getenv("SB_ZSTD_BLOCKS_POOL_DIV")) ctx->zb_pool_div = std::max<uint32_t>(1, (uint32_t)strtoul(e, nullptr, 10));
    *out = ctx;
    return SB_OK;
}

void sb_ctx_destroy(sb_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (auto& s : ctx->slots) {
        if (s.host) (void)hipHostFree(s.host);
        if (s.done) (void)hipEventDestroy(s.done);
    }
    for (void* p : ctx->stale_host) (void)hipHostFree(p);
    for (void* p : ctx->temp_dev) (void)hipFree(p);
    ctx->stage_release();
    for (auto& sp : ctx->spans) {
        (void)hipEventDestroy(sp.a);
        (void)hipEventDestroy(sp.b);
    }
    for (auto e : ctx->free_events) (void)hipEventDestroy(e);
    for (auto& l : ctx->freq_logs) (void)hipFree(l.dev);
    if (ctx->tables.p) (void)hipFree(ctx->tables.p);
    if (ctx->scratch.p) (void)hipFree(ctx->scratch.p);
    if (ctx->staging.p) (void)hipFree(ctx->staging.p);
    if (ctx->zlit.p) (void)hipFree(ctx->zlit.p);
    if (ctx->zrec.p) (void)hipFree(ctx->zrec.p);
    if (ctx->zb_stats) (void)hipFree(ctx->zb_stats);
    if (ctx->zb_blocks.p) (void)hipFree(ctx->zb_blocks.p);
    if (ctx->zb_lit.p) (void)hipFree(ctx->zb_lit.p);
    if (ctx->lzg_pool.p) (void)hipFree(ctx->lzg_pool.p);
    if (ctx->zb_rec.p) (void)hipFree(ctx->zb_rec.p);
    if (ctx->enc_plan.pages.p) (void)hipFree(ctx->enc_plan.pages.p);
    if (ctx->d_status) (void)hipFree(ctx->d_status);
    if (ctx->h_status) (void)hipHostFree(ctx->h_status);
    for (int i = 0; i < sb_ctx::NSIDE; i++) {
        if (ctx->side[i]) (void)hipStreamDestroy(ctx->side[i]);
        if (ctx->join_ev[i]) (void)hipEventDestroy(ctx->join_ev[i]);
    }
    if (ctx->fork_ev) (void)hipEventDestroy(ctx->fork_ev);
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    for (hipEvent_t e : ctx->pipe_ev) (void)hipEventDestroy(e);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char* sb_ctx_last_error(sb_ctx* ctx) { return ctx ? ctx->last_error.c_str() : "null context"; }
void* sb_ctx_stream(sb_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

static int32_t read_columns_impl(sb_ctx* ctx, sb_column_read* cols, uint64_t n, int32_t mem, bool sizes_only);

// Freq pages (integer/freq.rs:90-127) found by the decode calls of this synchronize interval: their
// exception blocks are ordinary BLOCK<T>s, so they go through the decoder once more as one-page
// columns that land in a temporary buffer; k_freq_scatter then writes them over the top value.
// Only runs when k_parse logged a Freq page; costs one 4-byte readback per log otherwise.
static int32_t freq_second_pass(sb_ctx* ctx) {
    struct Batch {
        const FreqEntry* d_entries;
        uint32_t n;
        uint8_t* ex_base;
        uint64_t* d_off;
    };
    std::vector<Batch> batches;
    ctx->freq_cols.clear();
    ctx->freq_metas.clear();
    for (auto& log : ctx->freq_logs) {
        if (!log.reserved) continue;
        log.reserved = 0;
        uint32_t cnt = 0;
        if (hipMemcpy(&cnt, log.dev, 4, hipMemcpyDeviceToHost) != hipSuccess) return ctx->fail(SB_ERR_EXTERNAL, "freq log readback failed");
        if (cnt == 0) continue;
        (void)hipMemsetAsync(log.dev, 0, 16, ctx->stream);
        cnt = std::min(cnt, log.cap);
        std::vector<FreqEntry> ents(cnt);
        if (hipMemcpy(ents.data(), log.dev + 16, (size_t)cnt * sizeof(FreqEntry), hipMemcpyDeviceToHost) != hipSuccess)
            return ctx->fail(SB_ERR_EXTERNAL, "freq log readback failed");
        std::vector<uint64_t> ex_off(cnt);
        uint64_t total = 0;
        for (uint32_t i = 0; i < cnt; i++) {
            ex_off[i] = total;
            total += ((uint64_t)ents[i].n_exceptions * ents[i].width + 15) / 16 * 16;
        }
        Batch b;
        b.d_entries = (const FreqEntry*)(log.dev + 16);
        b.n = cnt;
        if (hipMalloc((void**)&b.ex_base, total + 64) != hipSuccess) return ctx->fail(SB_ERR_EXTERNAL, "hipMalloc(freq exceptions) failed");
        ctx->temp_dev.push_back(b.ex_base);
        if (hipMalloc((void**)&b.d_off, (size_t)cnt * 8) != hipSuccess) return ctx->fail(SB_ERR_EXTERNAL, "hipMalloc(freq offsets) failed");
        ctx->temp_dev.push_back(b.d_off);
        if (hipMemcpy(b.d_off, ex_off.data(), (size_t)cnt * 8, hipMemcpyHostToDevice) != hipSuccess)
            return ctx->fail(SB_ERR_EXTERNAL, "freq offsets upload failed");
        for (uint32_t i = 0; i < cnt; i++) {
            sb_column_read c;
            memset(&c, 0, sizeof c);
            c.physical_type = (int32_t)ents[i].ptype;
            c.is_nullable = 0;
            c.pages = ents[i].nested;
            c.pages_len = ents[i].nested_len;
            c.n_pages = 1;
            c.values = b.ex_base + ex_off[i];
            c.values_capacity = (uint64_t)ents[i].n_exceptions * ents[i].width;
            ctx->freq_cols.push_back(c);
            sb_page_meta m;
            m.length = ents[i].nested_len;
            m.num_values = ents[i].n_exceptions;
            ctx->freq_metas.push_back(m);
        }
        batches.push_back(b);
    }
    if (batches.empty()) return SB_OK;
    ctx->freq_pass_ran = true;
    for (size_t i = 0; i < ctx->freq_cols.size(); i++) ctx->freq_cols[i].metas = &ctx->freq_metas[i];
    ctx->in_freq_pass = true;
    int32_t rc = read_columns_impl(ctx, ctx->freq_cols.data(), ctx->freq_cols.size(), SB_MEM_DEVICE, false);
    ctx->in_freq_pass = false;
    if (rc != SB_OK) return rc;
    for (const Batch& b : batches) launch_freq_scatter(ctx, b.d_entries, b.n, b.d_off, b.ex_base);
    hipError_t e = hipMemcpyAsync(ctx->h_status, ctx->d_status, sizeof(Status), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return check_hip(ctx, e, "freq second pass");
    // (what this pass met is not what the next interval's calls met: the word was cleared before the pass was queued)
    if (ctx->h_status->kinds) (void)hipMemsetAsync(&ctx->d_status->kinds, 0, sizeof ctx->d_status->kinds, ctx->stream);
    if (ctx->h_status->code != 0) {
        char buf[160];
        rc = ctx->h_status->code;
        ctx->last_error = status_text(*ctx->h_status, buf, sizeof buf);
        (void)hipMemsetAsync(ctx->d_status, 0, sizeof(Status), ctx->stream);
    }
    return rc;
}

int32_t sb_ctx_synchronize(sb_ctx* ctx) {
    if (!ctx) return SB_ERR_INVALID;
    (void)hipSetDevice(ctx->device);
    int32_t rc = ctx->sticky;
    // status word travels with the stream
    hipError_t e = hipMemcpyAsync(ctx->h_status, ctx->d_status, sizeof(Status), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) {
        ctx->kinds_seen |= ctx->h_status->kinds & KIND_ZSTD;
        // what the LAST interval's read calls met decides what the next ones launch: a context that read Zstd pages once and
        // LZ4 / plain pages ever since stops paying for the block pipeline's launches (7 kernels, ~40 us of a 0.6 ms call)
        // (two read intervals in a row without one: a nested reader alternates level calls — no Zstd — and leaf calls)
        if (ctx->h_status->kinds & KIND_ZSTD) {
            ctx->zstd_recent = true;
            ctx->zstd_idle = 0;
        } else if (ctx->read_calls && ++ctx->zstd_idle >= 2) {
            ctx->zstd_recent = false;
        }
        if (ctx->read_calls) {
            ctx->qa_idle = (ctx->h_status->kinds & KIND_QUEUE_A) ? 0 : ctx->qa_idle + 1;
            ctx->tiles_idle = (ctx->h_status->kinds & KIND_TILES) ? 0 : ctx->tiles_idle + 1;
        }
        ctx->read_calls = 0;
        // (three intervals with long pages and no such block before the chain is dropped: a reader that alternates giant-LZ4
        // columns with long plain ones keeps it; a wrong 2 costs a replay, not a one-workgroup walk)
        if (ctx->h_status->kinds & KIND_LZ4_GIANT) {
            ctx->lzg_state = 1;
            ctx->lzg_idle = 0;
        } else if (ctx->lzg_long_pages && ctx->lzg_state == 0) {
            ctx->lzg_state = 2;
        } else if (ctx->lzg_long_pages && ctx->lzg_state == 1 && ++ctx->lzg_idle >= 3) {
            ctx->lzg_state = 2;
        }
        ctx->lzg_long_pages = false;
        // (not sticky: what the calls since the last synchronize looked like decides the order of the next call's entropy kernels)
        if (ctx->h_status->kinds & KIND_ZSTD) ctx->zb_seq_long = (ctx->h_status->kinds & KIND_ZSEQ_LONG) != 0;
        // the device only ever sets bits: the word is cleared here so that it describes the calls of ONE interval (the host's
        // kinds_seen keeps what must stay)
        if (ctx->h_status->kinds) (void)hipMemsetAsync(&ctx->d_status->kinds, 0, sizeof ctx->d_status->kinds, ctx->stream);
    }
    // A page was left undone because a kernel it needed had been skipped on a hint (KIND_REPLAY): drop what the interval
    // produced and issue its calls again with every kernel launched.  One extra pass instead of a one-workgroup walk.
    // (an error code of such an interval is not looked at: kernels behind a skipped one may have met what it did not produce;
    // a real error shows again in the replay)
    if (e == hipSuccess && (ctx->h_status->kinds & KIND_REPLAY) && rc == SB_OK && !ctx->in_replay && !ctx->calls.empty()) {
        if (ctx->h_status->code != 0) (void)hipMemsetAsync(ctx->d_status, 0, sizeof(Status), ctx->stream);
        for (auto& s : ctx->slots) s.in_flight = false;
        for (auto& sp : ctx->spans) {
            ctx->free_events.push_back(sp.a);
            ctx->free_events.push_back(sp.b);
        }
        ctx->spans.clear();
        {   // (the page records of enqueued level calls stay: those calls are not issued again)
            std::vector<Pending> keep;
            for (auto& p : ctx->pending)
                if (p.kind == Pending::NESTED_W || p.kind == Pending::NESTED_R) {
                    ctx->rescued.emplace_back(p.host, p.host + p.bytes);
                    p.host = ctx->rescued.back().data();
                    keep.push_back(p);
                }
            ctx->pending.swap(keep);
        }
        for (void* p : ctx->stale_host) (void)hipHostFree(p);
        ctx->stale_host.clear();
        if (ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);   // (copies of groups that ran: overwritten by the replay's)
        ctx->copybacks.clear();
        ctx->pipe_ev_used = 0;
        for (void* p : ctx->temp_dev) (void)hipFree(p);
        ctx->temp_dev.clear();
        ctx->stage_rewind();
        for (auto& log : ctx->freq_logs) {
            log.reserved = 0;
            (void)hipMemsetAsync(log.dev, 0, 16, ctx->stream);
        }
        std::vector<sb_ctx::Call> calls;
        calls.swap(ctx->calls);
        const bool saved = ctx->no_hints;
        ctx->no_hints = true;
        ctx->in_replay = true;
        ctx->replays++;
        if (ctx->h_status->kinds & KIND_REPLAY_LZG) ctx->lzg_state = 1;
        for (auto& cl : calls) {
            rc = cl.kind ? sb_write_columns(ctx, (sb_column_write*)cl.cols, cl.n, &cl.opts, cl.mem) : sb_read_columns(ctx, (sb_column_read*)cl.cols, cl.n, cl.mem);
            if (rc != SB_OK) break;
        }
        ctx->no_hints = saved;
        if (rc != SB_OK) ctx->sticky = rc;
        rc = sb_ctx_synchronize(ctx);
        ctx->in_replay = false;
        return rc;
    }
    ctx->calls.clear();
    if (e != hipSuccess) {
        rc = check_hip(ctx, e, "sb_ctx_synchronize");
    } else if (ctx->h_status->code != 0) {
        char buf[160];
        if (!rc) {
            rc = ctx->h_status->code;
            ctx->last_error = status_text(*ctx->h_status, buf, sizeof buf);
        }
        (void)hipMemsetAsync(ctx->d_status, 0, sizeof(Status), ctx->stream);
    }
    if (rc == SB_OK) rc = freq_second_pass(ctx);
    if (rc != SB_OK) {
        // a failed interval: its Freq records point into buffers the caller may reuse — drop them all
        for (auto& log : ctx->freq_logs) {
            log.reserved = 0;
            (void)hipMemsetAsync(log.dev, 0, 16, ctx->stream);
        }
        (void)hipStreamSynchronize(ctx->stream);
    }
    for (auto& s : ctx->slots) s.in_flight = false;
    for (auto& sp : ctx->spans) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, sp.a, sp.b) == hipSuccess) {
            ctx->prof[sp.id].ms += ms;
            ctx->prof[sp.id].n += 1;
        }
        ctx->free_events.push_back(sp.a);
        ctx->free_events.push_back(sp.b);
    }
    ctx->spans.clear();
    for (auto& p : ctx->pending) {
        if (p.kind == Pending::READ_COL) {
            sb_column_read* c = (sb_column_read*)p.user;
            uint64_t v;
            memcpy(&v, p.host, 8);
            c->values_len = v;
        } else if (p.kind == Pending::ENC_HINT) {
            if (ctx->enc_plan.valid && ctx->enc_plan.key == p.n && rc == SB_OK) {
                uint32_t now[32];
                memcpy(now, p.host, 128);
                for (int i = 0; i < 32; i++) ctx->enc_plan.last_counts[i] = std::max(now[i], ctx->enc_plan.prev_counts[i]);
                memcpy(ctx->enc_plan.prev_counts, now, 128);
                ctx->enc_plan.counts_valid = true;
            }
        } else if (p.kind == Pending::NESTED_W) {
            if (rc == SB_OK) nested_write_finish((sb_nested_levels_write*)p.user, p.n, p.host);
        } else if (p.kind == Pending::NESTED_R) {
            if (rc == SB_OK) nested_read_finish((sb_nested_levels_read*)p.user, p.n, p.host);
        } else {
            sb_column_write* c = (sb_column_write*)p.user;
            const uint64_t* lens = (const uint64_t*)p.host;  // [n_pages lengths][n_pages num_values][total]
            for (uint64_t i = 0; i < p.n && i < c->n_pages_capacity; i++) {
                c->out_metas[i].length = lens[i];
                c->out_metas[i].num_values = lens[p.n + i];
            }
            c->n_pages = p.n;
            c->out_len = lens[2 * p.n];
        }
    }
    ctx->pending.clear();
    ctx->rescued.clear();
    for (void* p : ctx->stale_host) (void)hipHostFree(p);   // (the stream is drained: nothing reads them any more)
    ctx->stale_host.clear();
    {   // SB_MEM_HOST: what was not sent back while the interval ran (all copies on the copy stream, one wait)
        // (the stream exists only in contexts that serve host-memory calls: one more stream in the process changes how the
        // runtime maps streams to hardware queues — C4's side streams lost their overlap, 1.03 -> 1.55 ms per read)
        hipStream_t cs = ctx->copybacks.empty() ? ctx->copy_stream : ctx->copy_stream_get();
        bool any = false;
        for (auto& cb : ctx->copybacks) {
            if (cb.issued && !ctx->freq_pass_ran) {
                any = true;
                continue;
            }
            const size_t nb = cb.used ? (size_t)std::min<uint64_t>(cb.n, *cb.used) : cb.n;
            if (rc == SB_OK && nb) {
                hipError_t ce = cs ? hipMemcpyAsync(cb.host, cb.dev, nb, hipMemcpyDeviceToHost, cs) : hipMemcpy(cb.host, cb.dev, nb, hipMemcpyDeviceToHost);
                if (ce != hipSuccess) rc = check_hip(ctx, ce, "copy back");
                any = true;
            }
        }
        if (any && cs) {
            hipError_t ce = hipStreamSynchronize(cs);
            if (ce != hipSuccess && rc == SB_OK) rc = check_hip(ctx, ce, "copy back");
        }
    }
    ctx->copybacks.clear();
    ctx->freq_pass_ran = false;
    ctx->pipe_ev_used = 0;
    for (void* p : ctx->temp_dev) (void)hipFree(p);
    ctx->temp_dev.clear();
    ctx->stage_rewind();
    ctx->sticky = 0;
    return rc;
}

uint64_t sb_ctx_side_forks(sb_ctx* ctx) { return ctx ? ctx->side_forks : 0; }
uint64_t sb_ctx_replays(sb_ctx* ctx) { return ctx ? ctx->replays : 0; }

int32_t sb_ctx_zstd_block_stats(sb_ctx* ctx, uint64_t out[4]) {
    if (!ctx || !out) return SB_ERR_INVALID;
    const int32_t rc = sb_ctx_synchronize(ctx);
    if (hipMemcpy(out, ctx->zb_stats, 4 * sizeof(uint64_t), hipMemcpyDeviceToHost) != hipSuccess) return SB_ERR_EXTERNAL;
    return rc;
}

#ifdef ZB_TL
extern "C++" { namespace sb { void debug_lzx_timers(uint64_t* out8); } }
extern "C" int32_t sb_debug_zb_timers(sb_ctx* ctx, uint64_t out[20]) {   // development only (not in the header)
    (void)sb_ctx_synchronize(ctx);
    if (hipMemcpy(out, ctx->zb_stats + 4, 12 * sizeof(uint64_t), hipMemcpyDeviceToHost) != hipSuccess) return SB_ERR_EXTERNAL;
    sb::debug_lzx_timers(out + 12);
    return SB_OK;
}
#endif

int32_t sb_ctx_profile(sb_ctx* ctx, int32_t enable) {
    if (!ctx) return SB_ERR_INVALID;
    int32_t rc = sb_ctx_synchronize(ctx);
    ctx->profile = enable != 0;
    for (auto& e : ctx->prof) {
        e.ms = 0;
        e.n = 0;
    }
    return rc;
}

uint32_t sb_ctx_profile_read(sb_ctx* ctx, sb_kernel_stat* out, uint32_t cap) {
    if (!ctx) return 0;
    uint32_t n = 0;
    for (size_t i = 0; i < ctx->prof.size() && n < cap; i++) {
        if (!ctx->prof[i].n) continue;
        out[n].name = ctx->prof[i].name.c_str();
        out[n].launches = ctx->prof[i].n;
        out[n].total_ms = ctx->prof[i].ms;
        n++;
    }
    return n;
}

// ------------------------------------------------------------------------------------ decode
static int32_t read_columns_impl(sb_ctx* ctx, sb_column_read* cols, uint64_t n, int32_t mem, bool sizes_only) {
    if (!ctx || (!cols && n)) return SB_ERR_INVALID;
    if (n == 0) return SB_OK;
    (void)hipSetDevice(ctx->device);
    hipStream_t s = ctx->stream;
    uint64_t P = 0, T = 0, max_page_len = 0, max_page_rows = 0, lzg_pages = 0;
    bool any_binary = false, any_prim = false;
    for (uint64_t i = 0; i < n; i++) {
        sb_column_read& c = cols[i];
        if (c.physical_type < 0 || c.physical_type > SB_TYPE_NULL) return ctx->fail(SB_ERR_INVALID, "bad physical_type");
        if (c.n_pages && !c.metas) return ctx->fail(SB_ERR_INVALID, "metas is null");
        if (c.pages_len && !c.pages && c.physical_type != SB_TYPE_NULL) return ctx->fail(SB_ERR_INVALID, "pages is null");
        uint64_t rows = 0;
        for (uint64_t p = 0; p < c.n_pages; p++) {
            rows += c.metas[p].num_values;
            T += (c.metas[p].num_values + TILE_ROWS - 1) / TILE_ROWS;
            max_page_len = std::max<uint64_t>(max_page_len, c.metas[p].length);
            if (c.metas[p].length >= LZG_MIN) lzg_pages += is_binary_t(c.physical_type) ? 2 : 1;   // (blocks that may go block-parallel: sb_lz4_giant.h)
            max_page_rows = std::max<uint64_t>(max_page_rows, c.metas[p].num_values);
        }
        c.rows = rows;
        c.values_len = 0;
        P += c.n_pages;
        if (is_binary_t(c.physical_type))
            any_binary = true;
        else if (c.physical_type != SB_TYPE_NULL)
            any_prim = true;
        if (!sizes_only && c.physical_type != SB_TYPE_NULL && rows) {
            const uint32_t w = type_width(c.physical_type);
            if (!c.values) return ctx->fail(SB_ERR_INVALID, "values is null");
            if (c.is_nullable && (!c.validity || c.validity_capacity < (rows + 31) / 32 * 4))
                return ctx->fail(SB_ERR_INVALID, "validity buffer missing or smaller than 4*ceil(rows/32) bytes");
            if (is_binary_t(c.physical_type)) {
                if (!c.offsets || c.offsets_capacity < (rows + 1) * w)
                    return ctx->fail(SB_ERR_INVALID, "offsets buffer missing or too small");
            } else if (c.physical_type == SB_TYPE_BOOLEAN) {
                if (c.values_capacity < (rows + 31) / 32 * 4)
                    return ctx->fail(SB_ERR_INVALID, "boolean values buffer smaller than 4*ceil(rows/32) bytes");
            } else if (c.values_capacity < rows * w) {
                return ctx->fail(SB_ERR_INVALID, "values buffer too small");
            }
        }
    }
    if (P >= 0x7FFFFFFFull || T >= 0x7FFFFFFFull) return ctx->fail(SB_ERR_INVALID, "too many pages in one call");

    // ---- table layout
    uint64_t pages_bytes = 0;
    for (uint64_t i = 0; i < n; i++) pages_bytes += cols[i].pages_len;
    const size_t zs_extra = (size_t)std::min<uint64_t>(1u << 20, pages_bytes / 256 + 64);
    size_t off = 0;
    const size_t o_cols = off;
    off = align_up(off + n * sizeof(ColDesc), 64);
    const size_t o_tasks = off;
    off = align_up(off + P * sizeof(PageTask), 64);
    const size_t upload_bytes = off;
    const size_t o_descs = off;
    off = align_up(off + P * sizeof(PageDesc), 64);
    const size_t o_tiles = off;
    off = align_up(off + T * sizeof(TileTask), 64);
    // queue entries: 2 per page + room for the frames of Zstd buffers that are several frames (one entry per frame)
    const size_t job_cap = 2 * P + zs_extra;
    const size_t o_jobs_a = off;
    off = align_up(off + job_cap * sizeof(InflateJob), 64);
    const size_t o_jobs_b = off;
    off = align_up(off + job_cap * sizeof(InflateJob), 64);
    // queue Z (calls with binary columns that produce values): Zstd payloads known to k_parse, entropy stages with queue A's
    const bool want_z = any_binary && !sizes_only;
    const size_t o_jobs_z = off;
    if (want_z) off = align_up(off + job_cap * sizeof(InflateJob), 64);
    const size_t o_counts = off;
    off = align_up(off + 64, 64);
    const size_t o_vlen = off;
    off = align_up(off + n * sizeof(uint64_t), 64);
    // the block-parallel Zstd pipeline: frames + counters here, blocks / literals / records in pools of their own
    const bool zb_on = ctx->zb_mode == 1 || (ctx->zb_mode == 2 && (ctx->zstd_recent || ctx->no_hints));
    if (!ctx->in_freq_pass) ctx->read_calls++;   // (the Freq second pass runs inside a synchronize: not a call of the next interval)
    const size_t o_zb_counts = off;
    if (zb_on) off = align_up(off + 64, 64);
    const size_t o_zb_frames = off;
    if (zb_on) off = align_up(off + job_cap * sizeof(ZbFrame), 64);
    // long multi-frame Zstd buffers (a one-page column written by this library): frames found by a scan (sb_decode.hip)
    const bool zs_on = zb_on && !sizes_only && max_page_len >= (1u << 20);
    const uint64_t zs_seg_cap = zs_on ? pages_bytes / 16384 + 2 * P + 64 : 0;
    const size_t o_zs = off;
    if (zs_on) off = align_up(off + 256 + zs_seg_cap * 32, 64);
    // long RLE pages: few pages of many rows are shared by several workgroups each (sb_decode.hip: k_rle_sums)
    const uint32_t rle_parts = (!sizes_only && max_page_rows >= (1u << 18) && P > 0 && P <= 1024) ? (uint32_t)std::min<uint64_t>(256, 2048 / P) : 1u;
    const size_t o_rle = off;
    if (rle_parts > 1) off = align_up(off + P * rle_parts * sizeof(uint64_t), 64);
    const size_t o_bpg = off;
    if (rle_parts > 1) off = align_up(off + P * sizeof(uint32_t), 64);
    if (!ensure(ctx, ctx->tables, off)) return ctx->fail(SB_ERR_EXTERNAL, "hipMalloc(tables) failed");

    StageSlot* slot = acquire_slot(ctx, upload_bytes + n * sizeof(uint64_t));
    if (!slot) return ctx->fail(SB_ERR_EXTERNAL, "hipHostMalloc(staging) failed");
    ColDesc* hc = (ColDesc*)(slot->host + o_cols);
    PageTask* ht = (PageTask*)(slot->host + o_tasks);

    // SB_MEM_HOST: stage inputs/outputs in device temporaries
    std::vector<uint8_t*> dev_pages(n, nullptr), dev_values(n, nullptr), dev_validity(n, nullptr), dev_offsets(n, nullptr);
    if (mem == SB_MEM_HOST) {
        for (uint64_t i = 0; i < n; i++) {
            sb_column_read& c = cols[i];
            auto alloc = [&](size_t bytes, uint8_t** out) -> bool {
                *out = nullptr;
                if (!bytes) return true;
                return (*out = ctx->stage_alloc(bytes)) != nullptr;
            };
            if (!alloc(c.pages_len, &dev_pages[i])) return ctx->fail(SB_ERR_EXTERNAL, "hipMalloc(pages) failed");
            if (c.pages_len &&
                hipMemcpyAsync(dev_pages[i], c.pages, c.pages_len, hipMemcpyHostToDevice, s) != hipSuccess)
                return ctx->fail(SB_ERR_EXTERNAL, "H2D pages failed");
            if (!sizes_only) {
                if (!alloc(c.values_capacity, &dev_values[i]) || !alloc(c.is_nullable ? c.validity_capacity : 0, &dev_validity[i]) ||
                    !alloc(is_binary_t(c.physical_type) ? c.offsets_capacity : 0, &dev_offsets[i]))
                    return ctx->fail(SB_ERR_EXTERNAL, "hipMalloc(outputs) failed");
            }
        }
    }

    size_t scratch_off = 0;
    uint64_t page_i = 0, tile_i = 0;
    for (uint64_t i = 0; i < n; i++) {
        const sb_column_read& c = cols[i];
        ColDesc& d = hc[i];
        memset(&d, 0, sizeof d);
        d.pages = mem == SB_MEM_HOST ? dev_pages[i] : c.pages;
        d.pages_len = c.pages_len;
        d.values = mem == SB_MEM_HOST ? dev_values[i] : (uint8_t*)c.values;
        d.values_cap = sizes_only ? ~0ull : c.values_capacity;
        d.validity = mem == SB_MEM_HOST ? dev_validity[i] : c.validity;
        d.offsets = mem == SB_MEM_HOST ? dev_offsets[i] : (uint8_t*)c.offsets;
        d.offsets_cap = c.offsets_capacity;
        d.rows = c.rows;
        d.ptype = c.physical_type;
        d.nullable = c.is_nullable;
        d.width = type_width(c.physical_type);
        d.first_page = (uint32_t)page_i;
        d.n_pages = (uint32_t)c.n_pages;
        uint64_t in_off = 0, out_row = 0;
        d.bits_aligned = 1;
        for (uint64_t p = 0; p + 1 < c.n_pages; p++)
            if (c.metas[p].num_values % 32) d.bits_aligned = 0;
        for (uint64_t p = 0; p < c.n_pages; p++, page_i++) {
            PageTask& t = ht[page_i];
            const uint64_t N = c.metas[p].num_values, L = c.metas[p].length;
            const uint64_t ntiles = (N + TILE_ROWS - 1) / TILE_ROWS;
            t.in_off = c.page_offsets ? c.page_offsets[p] : in_off;
            t.length = L;
            if (c.page_offsets && t.in_off + L > c.pages_len) return ctx->fail(SB_ERR_IO, "page_offsets + length exceeds pages_len");
            t.num_values = N;
            t.out_row = out_row;
            t.col = (uint32_t)i;
            t.first_tile = (uint32_t)tile_i;
            t.aux_off = scratch_off;
            scratch_off += align_up((L / 4 + N / 128 + 4 * ntiles + 16) * 4, 16);   // (4 * ntiles: tile_k0 / tile_base + tile_bytes, and the u64 tile totals of a long binary Dict page)
            t.infl_off = scratch_off;
            scratch_off += align_up((N + 1) * 8 + 16, 16);
            in_off += L;
            out_row += N;
            tile_i += ntiles;
        }
        if (in_off > c.pages_len) return ctx->fail(SB_ERR_IO, "sum of PageMeta.length exceeds pages_len");
    }
    if (!ensure(ctx, ctx->scratch, scratch_off + 64)) return ctx->fail(SB_ERR_EXTERNAL, "hipMalloc(scratch) failed");
    // the inflate pool's per-wave areas: a literal buffer of one block, and (calls with at least 4 queue entries per pool
    // wave: batches) the arena of pre-decoded Zstd sequences
    if (!ensure(ctx, ctx->zlit, (size_t)INFLATE_POOL * (128 * 1024 + 64))) return ctx->fail(SB_ERR_EXTERNAL, "hipMalloc(zlit) failed");
    // (the lane-per-frame record arena only for calls that can hold >= 4 x INFLATE_POOL frames of 16 KiB, and only in a
    // context that has met Zstd pages: LZ4 / plain / Dict-only readers never pay for it)
    const bool zrec_wanted = pages_bytes >= (48ull << 20) && (ctx->zb_mode == 1 || ctx->zstd_recent);
    uint64_t zb_block_cap = 0, zb_lit_cap = 0, zb_rec_cap = 0;
    if (zb_on) {
        // blocks: libzstd's are 128 KiB of content (sub-blocks of a few KiB when it splits them); literals: at most the
        // output (a Huffman stream expands <= 8 x); records: one per >= 3 output bytes, in practice one per >= 2 stream bytes.
        // A frame that does not fit is decoded by the one-wave path.
        uint64_t out_bytes = 0;
        for (uint64_t i = 0; i < n; i++) {
            const uint64_t rows = cols[i].rows;
            out_bytes += sizes_only ? rows * 8 + 64 : cols[i].values_capacity + (is_binary_t(cols[i].physical_type) ? cols[i].offsets_capacity : 0) + rows * 8 + 64;
        }
        // Sized from the OUTPUT, not from the stream: a 128 KiB block of repetitive data is a few hundred stream bytes, RLE
        // literals expand 1 byte to 128 KiB, RLE / repeat-mode tables spend well under a byte per sequence.
        zb_block_cap = std::min<uint64_t>(std::max<uint64_t>(pages_bytes / 2048, out_bytes / 8192) + 2 * job_cap + 64, 1u << 23);
        // ... with a ceiling all the same: a C2-shaped read (4 GB out of 250 MB of pages) asked for ~20 GB.  Literals at most
        // 8 x the pages + 256 MB (what Huffman streams expand to; blocks of RLE literals beyond that overflow the pool), records
        // at most 2^28 (3 GB; a cap by the stream's bytes is wrong: small integers in repeat mode are several sequences per stream
        // byte); what does not fit goes to the frame-serial decoder (ZbCounts, tests/test_gpu_zstd_blocks.py).
        zb_lit_cap = std::min<uint64_t>(out_bytes, 8 * pages_bytes + (256ull << 20)) + 16 * zb_block_cap + (1u << 16);
        zb_rec_cap = std::min<uint64_t>(std::min<uint64_t>(4 * pages_bytes, out_bytes / 3), 1ull << 28) + (1u << 14);
        if (ctx->zb_pool_div > 1) {
            zb_block_cap = std::max<uint64_t>(zb_block_cap / ctx->zb_pool_div, 4);
            zb_lit_cap = std::max<uint64_t>(zb_lit_cap / ctx->zb_pool_div, 4096);
            zb_rec_cap = std::max<uint64_t>(zb_rec_cap / ctx->zb_pool_div, 64);
        }
        if (!ensure(ctx, ctx->zb_blocks, zb_block_cap * (sizeof(ZbBlock) + 16)) || !ensure(ctx, ctx->zb_lit, zb_lit_cap + 64) ||
            !ensure(ctx, ctx->zb_rec, zb_rec_cap * 12 + 16))
            return ctx->fail(SB_ERR_EXTERNAL, "hipMalloc(zstd block pools) failed");
    }
    if (zrec_wanted && !ensure(ctx, ctx->zrec, (size_t)INFLATE_POOL * ZREC_PER_WAVE * 8))
        return ctx->fail(SB_ERR_EXTERNAL, "hipMalloc(zrec) failed");

    uint8_t* tb = ctx->tables.p;
    hipError_t e = hipMemcpyAsync(tb, slot->host, upload_bytes, hipMemcpyHostToDevice, s);
    if (e != hipSuccess) return check_hip(ctx, e, "table upload");

    DecodeArgs a;
    a.zlit = ctx->zlit.p;
    a.zrec = zrec_wanted ? (uint64_t*)ctx->zrec.p : nullptr;
    a.cols = (const ColDesc*)(tb + o_cols);
    a.tasks = (const PageTask*)(tb + o_tasks);
    a.descs = (PageDesc*)(tb + o_descs);
    a.tiles = (TileTask*)(tb + o_tiles);
    a.scratch = ctx->scratch.p;
    a.status = ctx->d_status;
    a.jobs_a = (InflateJob*)(tb + o_jobs_a);
    a.jobs_b = (InflateJob*)(tb + o_jobs_b);
    a.jobs_z = want_z ? (InflateJob*)(tb + o_jobs_z) : nullptr;
    a.job_counts = (uint32_t*)(tb + o_counts);
    a.n_pages = (uint32_t)P;
    a.n_cols = (uint32_t)n;
    a.n_tiles = (uint32_t)T;
    // LZ4 blocks for the workgroup decoder: in a call with few blocks every block of 16 KiB and more (a lone wave's latency
    // is what the call waits for); in a call that fills the one-wave pool several times over only the blocks of 128 KiB and
    // more (64 KiB pages of incompressible values — the reference's bench shape — stay with the one-wave copy path)
    const uint32_t big_min = 2 * P >= 4096 ? 2 * LZ4_BIG_MIN : LZ4_BIG_MIN / 4;
    a.lz4_big_min = max_page_len >= big_min ? big_min : 0xFFFFFFFFu;
    // LZ4 blocks of megabytes (a one-page column): block-parallel (sb_lz4_giant.h) — tables and entries in a pool of their own
    memset(&a.lzg, 0, sizeof a.lzg);
    a.lzg_chunks = a.lzg_wins = a.lzg_rounds = a.lzg_jobs = 0;
    a.lzg_skipped = 0;
    // the inflate kernels of queue A / the tile kernel of primitives are left out when the last read interval queued nothing
    // for them (C2: four launches that found nothing to do, ~30 us of a 0.9 ms read); k_plan asks for the replay otherwise
    a.read_skips = 0;
    if (!ctx->no_hints && !ctx->in_freq_pass && !sizes_only) {
        if (ctx->qa_idle >= 2) a.read_skips |= RSKIP_QUEUE_A;
        if (ctx->tiles_idle >= 2) a.read_skips |= RSKIP_TILES;
    }
    a.zb_skipped = (!zb_on && ctx->zb_mode == 2 && !sizes_only && max_page_len >= (1u << 20)) ? 1u : 0u;
    if (!sizes_only && max_page_len >= LZG_MIN && ctx->lzg_state == 2 && !ctx->no_hints) {
        // the context's last intervals met no LZ4 block of megabytes: no pool, no launches; k_inflate_lz4_big leaves such
        // a block alone and asks for the replay (it used to walk it with one workgroup: 0.8 s for 68 MB)
        a.lzg_skipped = 1;
        ctx->lzg_long_pages = true;
    } else if (!sizes_only && max_page_len >= LZG_MIN) {
        // the pool holds what the (at most LZG_JOBS) picked blocks need: 16 bytes per compressed byte and 4 per output byte of
        // the largest candidate pages — not of every page of the call (128 plain 1 M-row columns pinned 27 GB that way)
        uint64_t out_max = 0;
        std::vector<std::pair<uint64_t, uint64_t>> cand;   // (page bytes, output bytes of its column's share)
        for (uint64_t i = 0; i < n; i++) {
            const uint64_t o = std::max<uint64_t>(cols[i].values_capacity, (cols[i].rows + 1) * 8);
            out_max = std::max(out_max, o);
            for (uint64_t k = 0; k < cols[i].n_pages; k++)
                if (cols[i].metas[k].length >= LZG_MIN) cand.push_back({cols[i].metas[k].length, o + (is_binary_t(cols[i].physical_type) ? (cols[i].rows + 1) * 8 : 0)});
        }
        std::sort(cand.begin(), cand.end(), [](const std::pair<uint64_t, uint64_t>& x, const std::pair<uint64_t, uint64_t>& y) { return x.first + x.second > y.first + y.second; });
        uint64_t pages_sel = 0, out_sel = 0;
        for (size_t q = 0; q < cand.size() && q < 2 * LZG_JOBS; q++) {   // (a page holds up to two blocks: queue A and queue B)
            pages_sel += cand[q].first;
            out_sel += cand[q].second;
        }
        const uint64_t pool = pages_sel * 16 + pages_sel / 512 + LZG_JOBS * (uint64_t)LZG_LITS * 16 + out_sel * 4 + out_sel / 2048 + (LZG_JOBS + 1) * (6 * 256 + (uint64_t)LZG_CH * 8 + 4096) +
                              LZG_JOBS * sizeof(LzgJob) + 1024;
        if (ensure(ctx, ctx->lzg_pool, pool)) {
            a.lzg.jobs = (LzgJob*)ctx->lzg_pool.p;
            a.lzg.njobs = (uint32_t*)(ctx->lzg_pool.p + LZG_JOBS * sizeof(LzgJob));
            const uint64_t head = (LZG_JOBS * sizeof(LzgJob) + 64 + 255) & ~255ull;
            a.lzg.pool = ctx->lzg_pool.p + head;
            a.lzg.pool_bytes = pool - head;
            a.lzg.st = ctx->d_status;
            a.lzg_chunks = (uint32_t)((max_page_len + LZG_CH - 1) / LZG_CH);
            a.lzg_jobs = (uint32_t)std::min<uint64_t>(lzg_pages, LZG_JOBS);
            ctx->lzg_long_pages = true;
            a.lzg_wins = (uint32_t)std::min<uint64_t>((out_max + LZG_WIN - 1) / LZG_WIN, 0x7FFFFFFFu);
            uint32_t bits = 1;
            while ((1ull << bits) < out_max + 1 && bits < 32) bits++;
            a.lzg_rounds = bits / 4 + 2;   // (a launch of k_lzg_jump is LZG_PASSES passes; a chain halves per pass at least — in practice a launch or two)
        }
    }
    a.rle_parts = rle_parts;
    a.rle_sums = rle_parts > 1 ? (uint64_t*)(tb + o_rle) : nullptr;
    a.bp_guess = rle_parts > 1 ? (uint32_t*)(tb + o_bpg) : nullptr;
    a.zs_hdr = a.zs_segs = nullptr;
    a.zs_seg_cap = 0;
    if (zs_on) {
        a.zs_hdr = (uint32_t*)(tb + o_zs);
        a.zs_segs = (uint32_t*)(tb + o_zs + 256);
        a.zs_seg_cap = (uint32_t)std::min<uint64_t>(zs_seg_cap, 0x7FFFFFFFu);
        (void)hipMemsetAsync(tb + o_zs, 0, 256 + zs_seg_cap * 32, s);
    }
    memset(&a.zb, 0, sizeof a.zb);
    if (zb_on) {
        a.zb.blocks = (ZbBlock*)ctx->zb_blocks.p;
        a.zb.lists = (uint32_t*)(ctx->zb_blocks.p + zb_block_cap * sizeof(ZbBlock));
        a.zb.frames = (ZbFrame*)(tb + o_zb_frames);
        a.zb.lit = ctx->zb_lit.p;
        a.zb.rec = (uint64_t*)ctx->zb_rec.p;
        a.zb.counters = (uint32_t*)(tb + o_zb_counts);
        a.zb.block_cap = (uint32_t)zb_block_cap;
        a.zb.frame_cap = (uint32_t)std::min<uint64_t>(job_cap, 0x7FFFFFFFu);
        a.zb.lit_cap = zb_lit_cap;
        a.zb.rec_cap = zb_rec_cap;
        a.zb.min_csize = ctx->zb_min_csize;
        a.zb.wg_exec = ctx->zb_wg_exec;
        a.zb.stats = ctx->zb_stats;
        a.zb.kinds = &ctx->d_status->kinds;
    }
    a.freq_log = nullptr;
    a.freq_count = nullptr;
    a.freq_cap = 0;
    a.no_freq = ctx->in_freq_pass ? 1u : 0u;
    if (!sizes_only && !ctx->in_freq_pass && P) {  // room for every page of this call to be a Freq page
        sb_ctx::FreqLog* log = ctx->freq_logs.empty() ? nullptr : &ctx->freq_logs.back();
        if (!log || (uint64_t)log->reserved + P > log->cap) {
            sb_ctx::FreqLog nl;
            nl.cap = (uint32_t)std::max<uint64_t>(8192, 2 * P);
            if (hipMalloc((void**)&nl.dev, 16 + (size_t)nl.cap * sizeof(FreqEntry)) != hipSuccess)
                return ctx->fail(SB_ERR_EXTERNAL, "hipMalloc(freq log) failed");
            (void)hipMemsetAsync(nl.dev, 0, 16, s);
            ctx->freq_logs.push_back(nl);
            log = &ctx->freq_logs.back();
        }
        log->reserved += (uint32_t)P;
        a.freq_count = (uint32_t*)log->dev;
        a.freq_log = (FreqEntry*)(log->dev + 16);
        a.freq_cap = log->cap;
    }
    uint64_t* d_vlen = (uint64_t*)(tb + o_vlen);

    a.sizes_only = sizes_only ? 1u : 0u;
    a.defer_payloads = (!sizes_only && any_binary) ? 1u : 0u;
    a.job_cap_a = a.job_cap_b = (uint32_t)job_cap;
    if (sizes_only) {
        if (P) launch_parse_sizes(ctx, a, d_vlen);
    } else {
        // bitmaps are assembled with OR at page seams: start from zero
        for (uint64_t i = 0; i < n; i++) {
            const ColDesc& d = hc[i];
            if (d.bits_aligned) continue;  // no bitmap word is shared between pages: plain stores only
            if (d.nullable && d.validity && d.rows) (void)hipMemsetAsync(d.validity, 0, (d.rows + 31) / 32 * 4, s);
            if (d.ptype == SB_TYPE_BOOLEAN && d.values && d.rows) (void)hipMemsetAsync(d.values, 0, (d.rows + 31) / 32 * 4, s);
        }
        if (P) launch_decode(ctx, a, any_binary, any_prim, d_vlen);
    }
    e = hipGetLastError();
    if (e != hipSuccess) return check_hip(ctx, e, "decode launch");

    // results: values_len per column
    uint8_t* hv = slot->host + upload_bytes;
    if (P && (any_binary || sizes_only)) {
        e = hipMemcpyAsync(hv, d_vlen, n * sizeof(uint64_t), hipMemcpyDeviceToHost, s);
        if (e != hipSuccess) return check_hip(ctx, e, "values_len readback");
    } else {
        for (uint64_t i = 0; i < n; i++) {  // fixed-width columns: rows * width (booleans: bitmap bytes)
            const uint64_t v = !P ? 0 : hc[i].ptype == SB_TYPE_BOOLEAN ? (hc[i].rows + 7) / 8 : hc[i].rows * hc[i].width;
            memcpy(hv + i * sizeof(uint64_t), &v, sizeof v);
        }
    }
    (void)hipEventRecord(slot->done, s);
    slot->in_flight = true;
    for (uint64_t i = 0; i < n; i++) {
        Pending pd;
        pd.kind = Pending::READ_COL;
        pd.user = &cols[i];
        pd.host = hv + i * sizeof(uint64_t);
        pd.n = 0;
        ctx->pending.push_back(pd);
        if (mem == SB_MEM_HOST && !sizes_only) {
            const sb_column_read& c = cols[i];
            const uint32_t w = type_width(c.physical_type);
            if (c.is_nullable && c.rows) ctx->copybacks.push_back({c.validity, dev_validity[i], (size_t)((c.rows + 7) / 8)});
            if (is_binary_t(c.physical_type)) {
                ctx->copybacks.push_back({c.offsets, dev_offsets[i], (size_t)((c.rows + 1) * w)});
                ctx->copybacks.push_back({c.values, dev_values[i], (size_t)c.values_capacity, &cols[i].values_len});  // (set just before, from `pending`)
            } else if (c.physical_type == SB_TYPE_BOOLEAN) {
                ctx->copybacks.push_back({c.values, dev_values[i], (size_t)((c.rows + 7) / 8)});
            } else if (c.physical_type != SB_TYPE_NULL) {
                ctx->copybacks.push_back({c.values, dev_values[i], (size_t)(c.rows * w)});
            }
        }
    }
    return SB_OK;
}

// SB_MEM_HOST calls of many columns are cut into groups: while group g + 1's pages travel to the device, group g's Arrow
// buffers travel back on the copy stream — PCIe's two directions are independent, and a call that ran them one after the
// other (all copies in, kernels, all copies out at the synchronize) used half of the link.  Fixed-size outputs are sent as
// soon as the group's kernels are done; values of binary columns (length known with the results) at the synchronize.
static uint64_t host_groups(sb_ctx* ctx, uint64_t n, uint64_t bytes) {
    if (n < 4 || bytes < (32ull << 20)) return 1;
    return std::min<uint64_t>(ctx->host_groups_max, n / 2);
}
int32_t sb_read_columns(sb_ctx* ctx, sb_column_read* cols, uint64_t n, int32_t mem) {
    int32_t rc = SB_OK;
    uint64_t groups = 1;
    if (ctx && cols && mem == SB_MEM_HOST && n >= 4) {
        uint64_t bytes = 0;
        for (uint64_t i = 0; i < n; i++) bytes += cols[i].pages_len + cols[i].values_capacity;
        groups = host_groups(ctx, n, bytes);
    }
    hipStream_t cs = groups > 1 ? ctx->copy_stream_get() : nullptr;
    if (!cs) {
        rc = read_columns_impl(ctx, cols, n, mem, false);
    } else {
        const uint64_t per = (n + groups - 1) / groups;
        for (uint64_t g0 = 0; g0 < n && rc == SB_OK; g0 += per) {
            const size_t cb0 = ctx->copybacks.size();
            rc = read_columns_impl(ctx, cols + g0, std::min<uint64_t>(per, n - g0), mem, false);
            if (rc != SB_OK) break;
            hipEvent_t ev = ctx->next_pipe_event();
            if (!ev || hipEventRecord(ev, ctx->stream) != hipSuccess || hipStreamWaitEvent(cs, ev, 0) != hipSuccess) continue;   // (copied at the synchronize)
            for (size_t k = cb0; k < ctx->copybacks.size(); k++) {
                auto& cb = ctx->copybacks[k];
                if (cb.used || !cb.n) continue;
                if (hipMemcpyAsync(cb.host, cb.dev, cb.n, hipMemcpyDeviceToHost, cs) == hipSuccess) cb.issued = true;
            }
        }
    }
    if (rc == SB_OK && ctx && n && !ctx->in_replay) ctx->calls.push_back(sb_ctx::Call{0, cols, n, sb_write_options{}, mem});
    return rc;
}

int32_t sb_read_columns_sizes(sb_ctx* ctx, sb_column_read* cols, uint64_t n, int32_t mem) {
    int32_t rc = read_columns_impl(ctx, cols, n, mem, true);
    if (rc != SB_OK) return rc;
    return sb_ctx_synchronize(ctx);
}

}  // extern "C"
