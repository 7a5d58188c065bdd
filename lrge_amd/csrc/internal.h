// internal.h -- shared host-side structures of liblrge_hip (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <chrono>
#include <cstring>
#include <iterator>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/lrge_hip.h"

typedef uint64_t u64;
typedef uint32_t u32;
typedef int64_t i64;
typedef int32_t i32;
typedef uint16_t u16;
typedef uint8_t u8;

#define WAVE 64

// Effective minimap2 parameters for one preset (mm2:options.c).
struct Preset {
    int k, w, hpc;
    int bw, max_gap, max_skip, max_iter, min_cnt, min_sc;
    int min_mid_occ, max_mid_occ;
    float mid_occ_frac, q_occ_frac;
    float pen_gap, pen_skip;  // chain_gap_scale*0.01*k evaluated in double then narrowed
};

inline Preset make_preset(int preset) {
    Preset p;
    p.w = 5; p.max_gap = 5000; p.max_skip = 25; p.max_iter = 5000; p.min_cnt = 3; p.min_sc = 100;
    p.min_mid_occ = 10; p.max_mid_occ = 1000000; p.mid_occ_frac = 2e-4f; p.q_occ_frac = 0.01f;
    if (preset == LRGE_PRESET_AVA_PB) { p.k = 19; p.hpc = 1; p.bw = 500; }
    else { p.k = 15; p.hpc = 0; p.bw = 2000; }
    p.pen_gap = (float)((double)0.8f * 0.01 * (double)p.k);
    p.pen_skip = (float)((double)0.0f * 0.01 * (double)p.k);
    return p;
}

// Device memory of a context: an arena allocator (the role of minimap2's kalloc / ThreadLocalBuffer, thread_buf.rs:7-42).
// hipMalloc costs ~25 ms per GB on this platform (measured: 321 GB of requests = 9.4 s of a 10.3 s H. sapiens-scale index
// build when every odd-sized request went to the runtime), and hipFree synchronises the device.  So memory is taken from
// the runtime in a few large SEGMENTS that grow geometrically, and requests are served from them by best fit with
// splitting; released blocks coalesce with their free neighbours.  Nothing goes back to the runtime before the context
// dies, except under memory pressure (trim(): wholly idle segments).  A released block is reusable at once: every user
// is ordered on the context's streams (internal.h: Scratch; side streams are joined with events before a block is
// released).
struct DevPool {
    struct Blk { size_t size; bool used; int seg; };
    std::map<char *, Blk> blks;                         // every block of every segment, by address
    std::multimap<size_t, char *> free_by_size;         // the free ones, for best fit
    struct Seg { char *base; size_t size; };
    std::vector<Seg> segs;
    size_t total = 0;                                   // bytes held from the runtime (all segments)
    size_t in_use = 0;
    long fail_every = 0, misses = 0;
    bool fail_always = false;          // test hook (DEBUG_ALLOC_FAIL_ALWAYS): every request is refused
    size_t seg_max = (size_t)32 << 30; // largest segment asked of the runtime at once (option POOL_SEG_MAX_MB: several contexts sharing one GPU keep it small, so that what one of them holds between two turns is not padded to 32 GB)
    // bookkeeping for option VERBOSE: what the device allocator itself cost
    double ms_malloc = 0, ms_free = 0; u64 n_malloc = 0, n_free = 0, n_trim = 0, bytes_malloc = 0;
    static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    static constexpr size_t kAlign = 256, kMinSeg = (size_t)64 << 20, kSplitMin = (size_t)64 << 10;

    void unfree(std::map<char *, Blk>::iterator it) {
        auto r = free_by_size.equal_range(it->second.size);
        for (auto f = r.first; f != r.second; ++f) if (f->second == it->first) { free_by_size.erase(f); return; }
    }
    void *take(std::map<char *, Blk>::iterator it, size_t bytes) {
        unfree(it);
        const size_t rest = it->second.size - bytes;
        if (rest >= kSplitMin) {
            it->second.size = bytes;
            char *q = it->first + bytes;
            blks[q] = Blk{rest, false, it->second.seg};
            free_by_size.emplace(rest, q);
        }
        it->second.used = true;
        in_use += it->second.size;
        return it->first;
    }
    void *alloc(size_t bytes, hipError_t *err) {
        if (fail_always) { *err = hipErrorOutOfMemory; return nullptr; }
        if (bytes == 0) bytes = kAlign;
        bytes = (bytes + kAlign - 1) & ~(kAlign - 1);
        auto f = free_by_size.lower_bound(bytes);
        if (f != free_by_size.end()) return take(blks.find(f->second), bytes);
        // a new segment: at least what is held already (geometric growth -> few, large segments that split well), never
        // less than the request; if the runtime cannot give that much, exactly the request, after returning idle segments
        // test hook (option DEBUG_ALLOC_FAIL_EVERY, per context, set by lrge_hip_ctx_set_option only): every n-th miss asks
        // for an impossible size first, i.e. takes the genuine failure + retry path
        const bool sabotage = fail_every > 0 && (++misses % fail_every) == 0;
        size_t want = std::max(bytes, std::max(kMinSeg, std::min(total, seg_max)));
        want = (want + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
        void *p = nullptr;
        const double t0 = now_ms();
        hipError_t e = hipMalloc(&p, sabotage ? ((size_t)1 << 60) : want);
        if (e != hipSuccess && !sabotage && want > bytes) {      // not that much left: exactly the request
            (void)hipGetLastError(); want = (bytes + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1); e = hipMalloc(&p, want);
        }
        if (e != hipSuccess) {  // give idle segments back and retry once
            (void)hipGetLastError();          // the failed attempt must not linger as the "last error" of a later launch check
            trim();
            e = hipMalloc(&p, want);
        }
        ms_malloc += now_ms() - t0; ++n_malloc;
        if (e != hipSuccess) { (void)hipGetLastError(); *err = e; return nullptr; }
        bytes_malloc += want;
        segs.push_back(Seg{(char *)p, want});
        total += want;
        auto it = blks.emplace((char *)p, Blk{want, false, (int)segs.size() - 1}).first;
        free_by_size.emplace(want, (char *)p);
        return take(it, bytes);
    }
    void release(void *p) {
        if (!p) return;
        auto it = blks.find((char *)p);
        if (it == blks.end() || !it->second.used) return;
        it->second.used = false;
        in_use -= it->second.size;
        // coalesce with the free neighbours of the same segment
        auto nx = std::next(it);
        if (nx != blks.end() && !nx->second.used && nx->second.seg == it->second.seg && nx->first == it->first + it->second.size) {
            unfree(nx); it->second.size += nx->second.size; blks.erase(nx);
        }
        if (it != blks.begin()) {
            auto pv = std::prev(it);
            if (!pv->second.used && pv->second.seg == it->second.seg && pv->first + pv->second.size == it->first) {
                unfree(pv); pv->second.size += it->second.size; blks.erase(it); it = pv;
            }
        }
        free_by_size.emplace(it->second.size, it->first);
    }
    // (round 2 gave multi-gigabyte staging blocks back to the runtime at once; inside an arena a released block serves any
    // later request, so this is release())
    void free_now(void *p) { release(p); }
    size_t cap_of(void *p) const { auto it = blks.find((char *)p); return it == blks.end() ? 0 : it->second.size; }
    // bytes held but not in use: what requests can be served from without asking the runtime (fragmentation aside).
    // `total` also counts the blocks in use -- a resident 150 GB index is not available memory
    size_t idle() const { return total - in_use; }
    // wholly idle segments go back to the runtime (memory pressure only: hipFree synchronises the device)
    void trim() {
        const double t0 = now_ms(); ++n_trim;
        for (size_t si = 0; si < segs.size(); ++si) {
            if (!segs[si].base) continue;
            auto it = blks.find(segs[si].base);
            if (it == blks.end() || it->second.used || it->second.size != segs[si].size) continue;
            unfree(it); blks.erase(it);
            (void)hipFree(segs[si].base); ++n_free;
            total -= segs[si].size;
            segs[si].base = nullptr; segs[si].size = 0;
        }
        ms_free += now_ms() - t0;
    }
    void destroy() {
        for (auto &sg : segs) if (sg.base) (void)hipFree(sg.base);
        segs.clear(); blks.clear(); free_by_size.clear(); total = 0; in_use = 0;
    }
};

struct TimerRec;
struct Uploader;
struct UploadJob;
struct lrge_hip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;           // side stream: k_chain_lpg runs beside k_chain_hw (both are latency-, not throughput-bound)
    // Upload path (lrge_hip_seqset_upload*): host -> device copies and the 2-bit pack run on their own stream, so that a
    // set's transfer overlaps whatever the main stream does for another set (the queries travel while the target index
    // is built).  A pageable source is staged through two pinned buffers filled by a few host threads.
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_gate = nullptr;            // main stream -> copy stream ordering at the start of an upload
    char *stage[2] = {nullptr, nullptr}; hipEvent_t stage_ev[2] = {nullptr, nullptr}; size_t stage_cap = 0;
    // host-side pack (host_pack.h): the uploader thread of the context and its two pinned chunk buffers (packed words + masks)
    Uploader *uploader = nullptr; char *hp_stage[2] = {nullptr, nullptr}; hipEvent_t hp_ev[2] = {nullptr, nullptr}; size_t hp_words = 0;
    // pinned arena for the per-read arrays of an upload (offsets, lengths, ranks, chunk and block maps): they are laid out
    // in it back to back and travel as ONE transfer (six copies from pageable vectors cost ~0.1 ms each of host time in
    // front of every index build).  Bump allocation; rewound when no upload is in flight.
    char *meta_pin = nullptr; size_t meta_cap = 0, meta_used = 0; int meta_inflight = 0;
    hipEvent_t ev_meta = nullptr;            // behind the last transfer out of the arena: waited for before a rewound arena is written again
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t ev_presk = nullptr;           // where the side stream may start a presketch (host_sketch.inl: presketch_prepare)
    std::string err;
    DevPool pool;
    // options: LRGE_HIP_<NAME> from the environment at creation, lrge_hip_ctx_set_option afterwards (never read per call)
    std::map<std::string, std::string> opts;
    const char *opt(const char *name) const { auto it = opts.find(name); return it == opts.end() ? nullptr : it->second.c_str(); }
    u64 opt_u64(const char *name, u64 dflt) const { const char *v = opt(name); return (v && *v) ? strtoull(v, nullptr, 10) : dflt; }
    float ms[LRGE_T_N];
    u64 counters[LRGE_C_N];
    int n_cu = 256;
    bool lsort_ok[3] = {false, false, false};   // which k_seg_sort_local variants this device can launch
    struct lrge_hip_seqset *presk_pending = nullptr; int presk_preset = -1;   // lrge_hip_seqset_presketch request
    struct PreSketch *presk_prepared = nullptr; struct lrge_hip_seqset *presk_prepared_set = nullptr;   // memory taken, kernels not yet queued
    int timer_level = 1;                     // see StageTimer
    u64 shard_stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // last sharded index build (lrge_hip_last_shard_stats)
    bool qshard_fresh = false;               // lrge_hip_seqset_presketch_sharded has just left its exchange volumes in shard_stats: the target-sharded build that follows keeps them
    u64 part_hint_total = 0, part_hint_bases = 0; int part_hint_preset = -1;   // the part size a build of this job had to fall back to (lrge_hip_index_build)
    double kept_ratio = 0.5; bool kept_seen = false;   // survivors per seed hit of the dead-pair filter in this context's recent batches (the batch planner's memory estimate)
    bool ts_build = false;                   // inside lrge_hip_index_build_tsharded: a partitioned local index leaves its occurrence statistics to the collective pass
    // Small device->host reads (totals, censuses, per-read vectors).  hipMemcpyAsync into pageable memory is a blocking
    // staged copy, one round trip EACH; through this pinned area several reads queue up behind the kernels and cost
    // one round trip at the following d2h_sync(), which also moves the bytes to where the caller wants them.
    char *pin = nullptr; size_t pin_cap = 0, pin_used = 0;
    struct PinItem { void *dst; size_t off, bytes; };
    std::vector<PinItem> pin_items;
    hipError_t d2h(void *dst, const void *src, size_t bytes, hipStream_t st) {
        const size_t need = (bytes + 63) & ~(size_t)63;
        if (!pin || pin_used + need > pin_cap) return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st);   // too large: direct
        const hipError_t e = hipMemcpyAsync(pin + pin_used, src, bytes, hipMemcpyDeviceToHost, st);
        pin_items.push_back(PinItem{dst, pin_used, bytes});
        pin_used += need;
        return e;
    }
    hipError_t d2h_sync(hipStream_t st) {
        const hipError_t e = hipStreamSynchronize(st);
        if (e == hipSuccess) for (const PinItem &it : pin_items) memcpy(it.dst, pin + it.off, it.bytes);
        pin_items.clear(); pin_used = 0;
        return e;
    }
    std::vector<struct TimerRec> timers;     // pending event pairs of the current call
    std::vector<hipEvent_t> event_pool;      // recycled events
    hipEvent_t get_event() {
        if (!event_pool.empty()) { hipEvent_t e = event_pool.back(); event_pool.pop_back(); return e; }
        hipEvent_t e = nullptr; (void)hipEventCreate(&e); return e;
    }
    void resolve_timers();                   // call after the stream has been synchronised
};

struct Scratch;
// A streamed read set sketched ahead of time on the side stream (lrge_hip_seqset_presketch): launched by the next
// index build right behind its own sketch, so that this VALU-bound work runs beside the index's memory-bound sort and
// table passes; consumed (once) by the next overlap call that streams the set against an index of the same preset.
struct PreSketch {
    int preset = -1;
    u64 *x = nullptr, *y = nullptr;     // worst-case sized (one minimizer per base)
    u32 *mz_off = nullptr, *d_total = nullptr;
    u32 *cnt = nullptr, *bs = nullptr;  // per-chunk counts and the scan's block sums (taken with the rest, before the launch)
    Scratch *sc = nullptr;              // owns every allocation of the launch until the consumer is done with them
    hipEvent_t ev_start = nullptr, ev_done = nullptr;
};

struct lrge_hip_seqset {
    lrge_hip_ctx *ctx;
    PreSketch *presk = nullptr;
    u64 uid = 0, parent_uid = 0; // identity of the set (of a view's parent) for the life of the process: addresses are recycled
    bool is_view = false;       // reads [r0, r1) of another set: shares its device arrays, owns only d_cs
    const lrge_hip_seqset *parent = nullptr;   // (of a view)
    // a view made while its parent's upload (host-side pack, chunk gates: host_pack.h) is still in flight waits for the GATE that
    // covers its last word, not for the whole set: the first part of a partitioned index is built -- the first view of a streamed set
    // is mapped -- while the later reads are still being packed and sent (seqset_view / seqset_ready)
    std::shared_ptr<UploadJob> view_job; int view_gate = -1; lrge_hip_seqset *view_root = nullptr;
    u32 n = 0;
    u64 total_bases = 0;
    u64 n_words = 0;            // 32-base words in the packed image (reads start on a word)
    u32 max_len = 0;
    bool has_empty = false;
    bool has_rank = false;
    bool dup_rank = false;      // two reads share a rank (duplicate identifier)
    // device
    u64 *d_pack = nullptr;      // 2-bit codes, 32 bases per word, base i at bits [2i,2i+1]
    u32 *d_nmask = nullptr;     // 1 bit per base: non-ACGTU
    u64 *d_woff = nullptr;      // [n+1] word offset of each read
    u32 *d_len = nullptr;       // [n]
    u32 *d_rank = nullptr;      // [n]
    u32 *d_cs = nullptr;        // [n+1] sketch chunk map: first 128-base chunk of each read
    u64 n_chunks = 0;
    // upload in flight on ctx->copy_stream: consumers order themselves behind ev_ready (seqset_ready); the staging
    // blocks below go back to the pool then
    bool pooled = false;        // device arrays come from the context's pool (not hipMalloc)
    bool pending = false;
    hipEvent_t ev_ready = nullptr;
    void *stg_ascii = nullptr, *stg_boff = nullptr, *stg_blk = nullptr;
    void *d_meta = nullptr;     // pooled sets: ONE device block behind d_woff, d_len, d_rank, d_cs, stg_boff, stg_blk
    bool meta_arena = false;    // the upload's per-read arrays sit in the context's pinned arena until the set is ready
    std::vector<u64> h_boff; std::vector<u32> h_blk;
    std::shared_ptr<UploadJob> job;   // host-side pack + chunked transfer running on the context's uploader thread (ev_ready is recorded by it)
    std::vector<u32> h_cs;
    // host copies needed for planning
    std::vector<u64> h_woff;
    std::vector<u32> h_len;
    std::vector<u32> h_rank;
};

struct lrge_hip_index {
    lrge_hip_ctx *ctx;
    const lrge_hip_seqset *seqs;
    int preset_id;
    Preset P;
    u64 n_mz = 0, n_keys = 0;   // what lrge_hip_index_stats reports (a restricted build: those of the WHOLE target set)
    u64 n_entries = 0;          // entries resident in d_pos / d_skey (a restricted build holds fewer than n_mz)
    int mid_occ = 0;
    u64 *d_pos = nullptr;       // [n_mz] y values grouped by key, ascending within a key
    u64 *d_skey = nullptr;      // [n_mz] sorted keys (kept for index_dump / tests)
    u64 *d_ht = nullptr;        // ordered open-addressing table of {key, start<<24 | min(count, 2^24-1)} pairs (k_index.h)
    u32 pk_pos1 = 0, pk_ybits = 0;   // packed entries (d_skey == d_pos): hash << pk_ybits | rid << pk_pos1 | (pos << 1 | strand)
    std::vector<u32> h_seg_start;    // segment-packed entries (k_prims.h: index_sort_segpacked): the hash field holds the low 2k - 8 - seg_e bits of the hash's significance
    u32 seg_e = 0;                   // string, the top 8 + seg_e bits are the number of the segment [h_seg_start[s], h_seg_start[s + 1]) the entry lies in: 256 << seg_e segments (k_index.h: seg_hash)
    u64 ht_cap = 0;             // home slots are [0, ht_cap); slack slots follow
    u32 ht_fix = 0;             // how ht_home() stretches the partial top byte of the hash (k_index.h); fixed at build time
    u64 ht_slots = 0;           // ht_cap + slack
    // A target set too large for one index (more than LRGE_HIP_PART_BASES bases: the 2^32-entry limits) is indexed in
    // parts over views of the set.  The occurrence statistics are global (k_part_global_occ), so the parts together
    // behave exactly like one index; a part's own d_* arrays are used as above, the container's are null.
    lrge_hip_seqset *owned_seqs = nullptr;          // a sharded build's description of the whole target set (freed with the index)
    const lrge_hip_seqset *restrict_set = nullptr;   // lrge_hip_index_build_for: the one set that may be streamed against this index
    u64 restrict_uid = 0;       // (its identity: a freed set's address may be reused by another one)
    std::vector<lrge_hip_index *> parts;
    std::vector<lrge_hip_seqset *> part_sets;
    std::vector<u32> part_r0;
};

#define LRGE_SET_ERR(ctx, ...)                                  \
    do {                                                        \
        char _b[512];                                           \
        snprintf(_b, sizeof(_b), __VA_ARGS__);                  \
        (ctx)->err = _b;                                        \
    } while (0)

#define HIPCHK(ctx, call)                                                                      \
    do {                                                                                       \
        hipError_t _e = (call);                                                                \
        if (_e != hipSuccess) {                                                                \
            LRGE_SET_ERR(ctx, "HIP error %s at %s:%d (%s)", hipGetErrorString(_e), __FILE__,   \
                         __LINE__, #call);                                                     \
            return LRGE_ERR_DEVICE;                                                            \
        }                                                                                      \
    } while (0)

#define KCHK(ctx) HIPCHK(ctx, hipGetLastError())

// RAII list of pool allocations released at scope exit.
struct Scratch {
    lrge_hip_ctx *ctx;
    std::vector<void *> ptrs;
    size_t max_bytes = 0;       // test hook (OverlapRun::batch, DEBUG_BATCH_ALLOC_MAX_BYTES): larger requests are refused, as a fuller device would
    explicit Scratch(lrge_hip_ctx *c) : ctx(c) {}
    ~Scratch() { for (void *p : ptrs) ctx->pool.release(p); }
    template <typename T> T *get(size_t n) {
        hipError_t e = hipErrorOutOfMemory;
        void *p = (max_bytes && n * sizeof(T) > max_bytes) ? nullptr : ctx->pool.alloc(n * sizeof(T), &e);
        if (!p) { LRGE_SET_ERR(ctx, "device allocation of %zu bytes failed: %s", n * sizeof(T), hipGetErrorString(e)); return nullptr; }
        ptrs.push_back(p);
        return (T *)p;
    }
    void drop(void *p) {
        for (size_t i = 0; i < ptrs.size(); ++i) if (ptrs[i] == p) { ptrs.erase(ptrs.begin() + i); break; }
        ctx->pool.release(p);
    }
    // hand ownership to the caller (not released at scope exit)
    void keep(void *p) {
        for (size_t i = 0; i < ptrs.size(); ++i) if (ptrs[i] == p) { ptrs.erase(ptrs.begin() + i); break; }
    }
};

#define ALLOC_OR_FAIL(var, sc, T, n)                  \
    T *var = (sc).get<T>(n);                          \
    if (!var) return LRGE_ERR_DEVICE;

// Stage timing without extra synchronisation: a StageTimer records a HIP event pair on the ctx stream;
// the pairs are turned into milliseconds once, by resolve_timers(), after the call's final sync.
struct TimerRec { int slot; hipEvent_t a, b; };

struct StageTimer {
    lrge_hip_ctx *ctx;
    int slot;
    hipStream_t st;
    hipEvent_t a = nullptr, b = nullptr;
    bool stopped = false;
    StageTimer(lrge_hip_ctx *c, int s, hipStream_t on = nullptr);
    void stop();
    ~StageTimer() { if (!stopped) stop(); }
};

static inline u32 ceil_log2_u64(u64 v) { u32 b = 0; while (b < 64 && (1ULL << b) < v) ++b; return b; }
static inline u64 div_up(u64 a, u64 b) { return (a + b - 1) / b; }

// Timer levels (lrge_hip_set_timer_level / LRGE_HIP_TIMERS): 0 = the call total and the chain stage only (what a
// caller that just wants results should pay: two event records per timer are host work between launches), 1 = every
// stage (default), 2 = also one pair around every k_rs_scatter and every k_sketch_direct launch.
inline int timer_slot_level(int slot) {
    return (slot == LRGE_T_TOTAL || slot == LRGE_T_CHAIN || slot == LRGE_T_CHAIN_LPG) ? 0 : (slot == LRGE_T_RS_SCATTER || slot == LRGE_T_K_SKETCH) ? 2 : 1;
}
inline StageTimer::StageTimer(lrge_hip_ctx *c, int s, hipStream_t on) : ctx(c), slot(s), st(on ? on : c->stream) {
    if (timer_slot_level(s) > ctx->timer_level) { stopped = true; return; }     // not recorded
    a = ctx->get_event(); b = ctx->get_event();
    (void)hipEventRecord(a, st);
}
inline void StageTimer::stop() {
    if (stopped) return;
    stopped = true;
    (void)hipEventRecord(b, st);
    ctx->timers.push_back(TimerRec{slot, a, b});
}
inline void lrge_hip_ctx::resolve_timers() {
    for (auto &t : timers) {
        (void)hipEventSynchronize(t.b);
        float ms = 0;
        if (hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess) this->ms[t.slot] += ms;
        event_pool.push_back(t.a); event_pool.push_back(t.b);
    }
    timers.clear();
}
