// lrge_hip.hip -- extern "C" entry points of liblrge_hip.so (see include/lrge_hip.h) and the host
// orchestration of the kernels in k_*.h.  gfx950 only; no CPU fallback: every entry point that
// computes fails with LRGE_ERR_DEVICE when no HIP device is usable.
#include <algorithm>
#include <cmath>
#include <chrono>
#include <cstdlib>
#include <memory>
#include <set>
#include <mutex>
#include <thread>

#include "internal.h"
#include "k_prims.h"
#include "k_sketch.h"
#include "k_index.h"
#include "k_seed.h"
#include "k_chain.h"
#include "k_chain_common.h"
#include "k_chain_hw.h"
#include "k_chain_lpg.h"
#include "comm.h"
#include "host_pack.h"
#include "k_restrict.h"
#include "k_route.h"
#include "../../include/lrge_rand.hpp"

static thread_local std::string g_last_error;  // failures that happen before a ctx exists
static std::mutex g_live_mu;
static std::set<lrge_hip_ctx *> g_live_ctx;   // contexts that have not been destroyed
static std::atomic<u64> g_seqset_uid{1};

extern char **environ;

// Tuning / test options of a context (internal.h: lrge_hip_ctx::opts).  Read from the environment ONCE, when the context
// is created (LRGE_HIP_<NAME>=value; INTEGRATION.md section 6 lists them), and settable afterwards through
// lrge_hip_ctx_set_option.  The DEBUG_* options change results or inject failures on purpose (parity tests drive rarely
// taken paths with them): they are never read from the environment, only the explicit call sets them.
static void load_env_options(lrge_hip_ctx *ctx) {
    for (char **e = environ; e && *e; ++e) {
        if (strncmp(*e, "LRGE_HIP_", 9) != 0 || strncmp(*e, "LRGE_HIP_DEBUG_", 15) == 0) continue;
        const char *eq = strchr(*e, '=');
        if (!eq) continue;
        ctx->opts[std::string(*e + 9, (size_t)(eq - (*e + 9)))] = std::string(eq + 1);
    }
}

extern "C" int lrge_hip_ctx_set_option(lrge_hip_ctx *ctx, const char *name, const char *value) {
    if (!ctx || !name || !*name) return LRGE_ERR_INVALID;
    if (value) ctx->opts[name] = value; else ctx->opts.erase(name);
    if (!strcmp(name, "DEBUG_ALLOC_FAIL_EVERY")) { ctx->pool.fail_every = value ? atol(value) : 0; ctx->pool.misses = 0; }
    if (!strcmp(name, "DEBUG_ALLOC_FAIL_ALWAYS")) ctx->pool.fail_always = value != nullptr;
    if (!strcmp(name, "TIMERS") && value) ctx->timer_level = atoi(value);
    return LRGE_OK;
}

extern "C" void lrge_hip_seqset_free(lrge_hip_seqset *s);
extern "C" const char *lrge_hip_version(void) { return "lrge_hip 0.1.0 (gfx950)"; }

extern "C" int lrge_hip_device_count(int *n) {
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) { *n = 0; g_last_error = hipGetErrorString(e); return LRGE_ERR_DEVICE; }
    *n = c;
    return LRGE_OK;
}

// The two streams of a context must not share a hardware queue: the chain kernels (and the presketch, the largest
// LDS sort class, the query-occurrence check) run side by side on them, and streams that the runtime multiplexes onto
// one queue serialise.  That happens in processes that own other streams -- measured under torch.distributed + RCCL:
// chain stage 2.3 -> 4.1 ms -- and neither a stream priority nor a creation order guarantees otherwise.  So the side
// stream is chosen by measurement: two ~40 us spin kernels, one per stream, must finish in clearly less than the sum.
__global__ void k_spin_ticks(long long ticks, u32 *sink) {
    const long long t0 = wall_clock64();               // constant 100 MHz counter
    while (wall_clock64() - t0 < ticks) { }
    if (sink && threadIdx.x == 1024) *sink = 1;         // (never true: keeps the loop observable)
}
static hipError_t pick_side_stream(lrge_hip_ctx *ctx) {
    ctx->stream2 = nullptr;
    if (ctx->opt("NO_STREAM_PROBE")) return hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking);
    const long long ticks = 4000;                        // 40 us
    auto run_pair = [&](hipStream_t a, hipStream_t b) -> double {
        (void)hipStreamSynchronize(a); if (b) (void)hipStreamSynchronize(b);
        const auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(k_spin_ticks, dim3(1), dim3(64), 0, a, ticks, (u32 *)nullptr);
        if (b) hipLaunchKernelGGL(k_spin_ticks, dim3(1), dim3(64), 0, b, ticks, (u32 *)nullptr);
        (void)hipStreamSynchronize(a); if (b) (void)hipStreamSynchronize(b);
        return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    };
    (void)run_pair(ctx->stream, nullptr);                // warm-up: code object load, first launch
    double single = 1e30;
    for (int i = 0; i < 3; ++i) single = std::min(single, run_pair(ctx->stream, nullptr));
    std::vector<hipStream_t> rejected;
    hipError_t err = hipSuccess;
    for (int attempt = 0; attempt < 12; ++attempt) {
        hipStream_t cand = nullptr;
        if ((err = hipStreamCreateWithFlags(&cand, hipStreamNonBlocking)) != hipSuccess) break;
        (void)run_pair(ctx->stream, cand);
        double both = 1e30;
        for (int i = 0; i < 3; ++i) both = std::min(both, run_pair(ctx->stream, cand));
        if (both < 1.5 * single || attempt == 11) { ctx->stream2 = cand; break; }   // concurrent (or nothing better to be had)
        rejected.push_back(cand);                          // kept alive until the choice is made: it holds its queue slot
    }
    for (hipStream_t r : rejected) (void)hipStreamDestroy(r);
    (void)hipGetLastError();
    if (!ctx->stream2 && err == hipSuccess) err = hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking);
    return err;
}

extern "C" int lrge_hip_ctx_create(int device, lrge_hip_ctx **out) {
    *out = nullptr;
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess || c <= 0 || device < 0 || device >= c) {
        g_last_error = "no usable HIP device";
        return LRGE_ERR_DEVICE;
    }
    lrge_hip_ctx *ctx = new lrge_hip_ctx();
    ctx->device = device;
    load_env_options(ctx);
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
        pick_side_stream(ctx) != hipSuccess || hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_gate, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_presk, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming) != hipSuccess) {
        g_last_error = "hipSetDevice/hipStreamCreate failed";
        delete ctx;
        return LRGE_ERR_DEVICE;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->n_cu = prop.multiProcessorCount;
    if (hipHostMalloc((void **)&ctx->pin, 1u << 20, hipHostMallocDefault) == hipSuccess) ctx->pin_cap = 1u << 20;
    else { ctx->pin = nullptr; (void)hipGetLastError(); }       // reads fall back to pageable copies
    // segment-local sort variants: the two larger ones need more than the default 64 KB of LDS per workgroup
    ctx->lsort_ok[0] = true;
    ctx->lsort_ok[1] = ctx->lsort_ok[0] && hipFuncSetAttribute((const void *)k_seg_sort_local<512, 16, LSORT_DB>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                               (int)LSORT_BYTES(512, 16, LSORT_DB)) == hipSuccess;
    ctx->lsort_ok[2] = ctx->lsort_ok[0] && hipFuncSetAttribute((const void *)k_seg_sort_local<1024, 16, LSORT_DB>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                               (int)LSORT_BYTES(1024, 16, LSORT_DB)) == hipSuccess;
    // (the keys-only siblings used by the index sort: same LDS footprints)
    if (hipFuncSetAttribute((const void *)k_seg_sort_keys<512, 16, LSORT_DB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LSORT_BYTES(512, 16, LSORT_DB)) != hipSuccess) ctx->lsort_ok[1] = false;
    if (hipFuncSetAttribute((const void *)k_seg_sort_keys<1024, 16, LSORT_DB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LSORT_BYTES(1024, 16, LSORT_DB)) != hipSuccess) ctx->lsort_ok[2] = false;
    (void)hipGetLastError();
    ctx->resolve_timers();
    memset(ctx->ms, 0, sizeof(ctx->ms));
    memset(ctx->counters, 0, sizeof(ctx->counters));
    if (ctx->opt("TIMERS")) ctx->timer_level = atoi(ctx->opt("TIMERS"));
    { std::lock_guard<std::mutex> g(g_live_mu); g_live_ctx.insert(ctx); }
    *out = ctx;
    return LRGE_OK;
}

static void pool_report(lrge_hip_ctx *ctx, const char *where) {
    if (!ctx->opt("VERBOSE")) return;
    const DevPool &P = ctx->pool;
    size_t n_seg = 0; for (const auto &sg : P.segs) n_seg += sg.base != nullptr;
    fprintf(stderr, "[lrge_hip] pool at %s: %zu blocks in %zu segments, %.1f GB held (%.1f idle); so far %llu hipMalloc (%.1f GB, %.0f ms), %llu trims freeing %llu segments (%.0f ms)\n",
            where, P.blks.size(), n_seg, P.total / 1073741824.0, P.idle() / 1073741824.0, (unsigned long long)P.n_malloc, P.bytes_malloc / 1073741824.0, P.ms_malloc,
            (unsigned long long)P.n_trim, (unsigned long long)P.n_free, P.ms_free);
}

extern "C" void lrge_hip_ctx_destroy(lrge_hip_ctx *ctx) {
    if (!ctx) return;
    pool_report(ctx, "ctx_destroy");
    { std::lock_guard<std::mutex> g(g_live_mu); g_live_ctx.erase(ctx); }
    (void)hipSetDevice(ctx->device);
    delete ctx->uploader; ctx->uploader = nullptr;      // (drains its queue)
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipStreamSynchronize(ctx->copy_stream);
    for (int b = 0; b < 2; ++b) { if (ctx->hp_stage[b]) (void)hipHostFree(ctx->hp_stage[b]); if (ctx->hp_ev[b]) (void)hipEventDestroy(ctx->hp_ev[b]); }
    ctx->resolve_timers();
    for (int b = 0; b < 2; ++b) { if (ctx->stage[b]) (void)hipHostFree(ctx->stage[b]); if (ctx->stage_ev[b]) (void)hipEventDestroy(ctx->stage_ev[b]); }
    (void)hipEventDestroy(ctx->ev_gate); (void)hipStreamDestroy(ctx->copy_stream);
    for (hipEvent_t e : ctx->event_pool) (void)hipEventDestroy(e);
    ctx->pool.destroy();
    if (ctx->pin) (void)hipHostFree(ctx->pin);
    if (ctx->ev_meta) (void)hipEventDestroy(ctx->ev_meta);
    if (ctx->meta_pin) (void)hipHostFree(ctx->meta_pin);
    (void)hipStreamSynchronize(ctx->stream2);
    (void)hipEventDestroy(ctx->ev_fork); (void)hipEventDestroy(ctx->ev_join); (void)hipEventDestroy(ctx->ev_presk);
    (void)hipStreamDestroy(ctx->stream2);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" const char *lrge_hip_last_error(const lrge_hip_ctx *ctx) { return ctx ? ctx->err.c_str() : g_last_error.c_str(); }

extern "C" int lrge_hip_set_timer_level(lrge_hip_ctx *ctx, int level) {
    if (!ctx || level < 0 || level > 2) return LRGE_ERR_INVALID;
    ctx->timer_level = level;
    return LRGE_OK;
}

extern "C" int lrge_hip_last_timings(const lrge_hip_ctx *ctx, float ms[LRGE_T_N]) {
    if (!ctx) return LRGE_ERR_INVALID;
    memcpy(ms, ctx->ms, sizeof(ctx->ms));
    return LRGE_OK;
}
extern "C" int lrge_hip_last_counters(const lrge_hip_ctx *ctx, uint64_t c[LRGE_C_N]) {
    if (!ctx) return LRGE_ERR_INVALID;
    memcpy(c, ctx->counters, sizeof(ctx->counters));
    return LRGE_OK;
}

// ------------------------------------------------------------------------------------------
// read sets
// ------------------------------------------------------------------------------------------
// Name ranks are positions in the sorted union of the names that meet in a call, i.e. small dense integers: duplicates and
// intersections are found with one bitmap pass instead of a sort per upload (a sort of 100 000 ranks was ~1 ms of host time
// in front of every index build).  Sparse rank values (a caller's own numbering) fall back to sorting.
static bool ranks_have_duplicate(const std::vector<u32> &r) {
    if (r.size() < 2) return false;
    u32 mx = 0;
    for (u32 v : r) mx = v > mx ? v : mx;
    if ((u64)mx <= 64ull * r.size() + 1024) {
        std::vector<u64> bits(((size_t)mx >> 6) + 1, 0);
        for (u32 v : r) { u64 &w = bits[v >> 6]; const u64 m = 1ULL << (v & 63); if (w & m) return true; w |= m; }
        return false;
    }
    std::vector<u32> t(r);
    std::sort(t.begin(), t.end());
    for (size_t i = 1; i < t.size(); ++i) if (t[i] == t[i - 1]) return true;
    return false;
}
static bool ranks_intersect(const std::vector<u32> &a, const std::vector<u32> &b) {
    if (a.empty() || b.empty()) return false;
    const std::vector<u32> &small = a.size() <= b.size() ? a : b, &large = a.size() <= b.size() ? b : a;
    u32 mx = 0;
    for (u32 v : large) mx = v > mx ? v : mx;
    if ((u64)mx <= 64ull * large.size() + 1024) {
        std::vector<u64> bits(((size_t)mx >> 6) + 1, 0);
        for (u32 v : large) bits[v >> 6] |= 1ULL << (v & 63);
        for (u32 v : small) if (v <= mx && (bits[v >> 6] >> (v & 63)) & 1) return true;
        return false;
    }
    std::vector<u32> t(large);
    std::sort(t.begin(), t.end());
    for (u32 v : small) if (std::binary_search(t.begin(), t.end(), v)) return true;
    return false;
}

extern "C" int lrge_hip_host_alloc(size_t bytes, void **out) {
    if (!out) return LRGE_ERR_INVALID;
    *out = nullptr;
    const hipError_t e = hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) { (void)hipGetLastError(); g_last_error = hipGetErrorString(e); return LRGE_ERR_DEVICE; }
    return LRGE_OK;
}
extern "C" void lrge_hip_host_free(void *p) { if (p) (void)hipHostFree(p); }

// a host-side pack still running on the uploader thread: its last act is to record the set's ev_ready, so nobody may wait
// for that event on the device before the job has finished on the host
static int seqset_job_wait(lrge_hip_ctx *ctx, lrge_hip_seqset *s) {
    if (!s->job) return LRGE_OK;
    std::string e;
    const int rc = s->job->wait(&e);
    for (hipEvent_t g : s->job->gate_ev) ctx->event_pool.push_back(g);
    s->job->gate_ev.clear();
    s->job.reset();
    if (rc) { ctx->err = e; return rc; }
    return LRGE_OK;
}

// Every consumer of a set's device arrays calls this first: work queued on the main stream after it runs behind the
// set's upload; the staging blocks of the upload return to the pool (recycled in main-stream order from here on).
static int seqset_ready(lrge_hip_ctx *ctx, const lrge_hip_seqset *cs) {
    lrge_hip_seqset *s = const_cast<lrge_hip_seqset *>(cs);
    if (!s->pending) return LRGE_OK;
    { int jrc = seqset_job_wait(ctx, s); if (jrc) return jrc; }
    HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, s->ev_ready, 0));
    s->pending = false;
    if (s->meta_arena) { s->meta_arena = false; if (--ctx->meta_inflight == 0) ctx->meta_used = 0; }
    ctx->pool.release(s->stg_ascii);      // (an arena block: whatever its size, it serves any later request)
    s->stg_ascii = nullptr;       // (stg_boff / stg_blk live inside the set's meta block)
    return LRGE_OK;
}

// pageable source -> pinned staging buffer with a few host threads (one thread moves ~10 GB/s, PCIe Gen5 x16 ~55)
static void parallel_memcpy(char *dst, const char *src, size_t n) {
    const size_t kMin = (size_t)4 << 20;
    const unsigned nt = (unsigned)std::min<size_t>(8, std::max<size_t>(1, n / kMin));
    if (nt <= 1) { memcpy(dst, src, n); return; }
    std::vector<std::thread> th;
    const size_t per = (n / nt + 63) & ~(size_t)63;
    for (unsigned t = 1; t < nt; ++t) {
        const size_t o = std::min(n, per * t), e = std::min(n, per * (t + 1));
        if (e > o) th.emplace_back([=] { memcpy(dst + o, src + o, e - o); });
    }
    memcpy(dst, src, std::min(n, per));
    for (auto &x : th) x.join();
}

static int seqset_upload_impl(lrge_hip_ctx *ctx, const char *bases, const uint64_t *offsets, uint32_t n, const uint32_t *name_rank,
                              bool async, lrge_hip_seqset **out) {
    if (!ctx || !out || (n && (!bases || !offsets))) return LRGE_ERR_INVALID;
    *out = nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    HIPCHK(ctx, hipSetDevice(ctx->device));
    ctx->pin_items.clear(); ctx->pin_used = 0;      // reads an earlier, failed call may have left queued
    std::unique_ptr<lrge_hip_seqset, void (*)(lrge_hip_seqset *)> guard(new lrge_hip_seqset(), lrge_hip_seqset_free);
    lrge_hip_seqset *s = guard.get();
    s->ctx = ctx; s->n = n; s->pooled = true; s->uid = g_seqset_uid.fetch_add(1);
    // one pass over the offsets: word offsets, lengths, sketch chunk map (read -> first chunk, fixed for the life of the set)
    s->h_woff.resize((size_t)n + 1); s->h_len.resize(n ? n : 1); s->h_cs.resize((size_t)n + 1);
    u64 w = 0, nc = 0;
    {
        u64 *hw = s->h_woff.data(); u32 *hl = s->h_len.data(), *hc = s->h_cs.data();
        u32 max_len = 0; bool has_empty = false;
        for (u32 i = 0; i < n; ++i) {
            const u64 d = offsets[i + 1] - offsets[i];
            if (offsets[i + 1] < offsets[i] || d >= (1ULL << 31)) {
                LRGE_SET_ERR(ctx, "read %u: bad offsets or length >= 2^31", i); return LRGE_ERR_INVALID;
            }
            const u32 len = (u32)d;
            hw[i] = w; hl[i] = len; hc[i] = (u32)nc;
            w += (len + 31) / 32; nc += (len + SK_CHUNK - 1) / SK_CHUNK;
            has_empty |= len == 0;
            max_len = len > max_len ? len : max_len;
        }
        hw[n] = w; hc[n] = (u32)nc;
        s->max_len = max_len; s->has_empty = has_empty;
    }
    s->n_words = w; s->n_chunks = nc;
    s->total_bases = n ? offsets[n] - offsets[0] : 0;
    if (name_rank) {
        s->has_rank = true;
        s->h_rank.assign(name_rank, name_rank + n);
        s->dup_rank = ranks_have_duplicate(s->h_rank);
    }
    const u64 n_blk = div_up(w, PACK_WORDS);
    hipError_t e = hipSuccess;
    auto alloc = [&](size_t bytes) -> void * { return ctx->pool.alloc(bytes, &e); };
    const size_t nw = (size_t)(w ? w : 1);
    s->d_pack = (u64 *)alloc(nw * 8); s->d_nmask = (u32 *)alloc(nw * 4);
    // the per-read arrays: one device block, one host image, one transfer
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t o_woff = 0, o_boff = o_woff + al(((size_t)n + 1) * 8), o_blk = o_boff + al(((size_t)n + 1) * 8);
    const size_t o_cs = o_blk + al((size_t)(n_blk + 1) * 4), o_len = o_cs + al(((size_t)n + 1) * 4);
    const size_t o_rank = o_len + al((size_t)(n ? n : 1) * 4), meta_bytes = o_rank + al((size_t)(n ? n : 1) * 4);
    s->d_meta = alloc(meta_bytes);
    if (!s->d_pack || !s->d_nmask || !s->d_meta) {
        LRGE_SET_ERR(ctx, "seqset_upload: device allocation failed: %s", hipGetErrorString(e)); return LRGE_ERR_DEVICE;
    }
    char *dm = (char *)s->d_meta;
    s->d_woff = (u64 *)(dm + o_woff); s->stg_boff = dm + o_boff; s->stg_blk = dm + o_blk;
    s->d_cs = (u32 *)(dm + o_cs); s->d_len = (u32 *)(dm + o_len); s->d_rank = (u32 *)(dm + o_rank);
    // where do the bases live?  device memory (no copy at all), pinned host memory (one DMA), pageable host memory (staged)
    const char *src = n ? bases + offsets[0] : nullptr;
    int kind = 2;                                         // 0 device, 1 pinned host, 2 pageable host
    if (src && s->total_bases) {
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, src) == hipSuccess) {
            if (at.type == hipMemoryTypeDevice) kind = 0; else if (at.type == hipMemoryTypeHost) kind = 1;
        } else (void)hipGetLastError();
    }
    const u8 *d_ascii = (const u8 *)src;
    // a set that starts in host memory is packed on the host and travels packed (host_pack.h); option NO_HOST_PACK sends the
    // ASCII and packs on the device as rounds 1-2 did
    const bool host_pack = kind != 0 && s->total_bases > 0 && !ctx->opt("NO_HOST_PACK");
    if (kind != 0 && s->total_bases && !host_pack) {
        s->stg_ascii = alloc(s->total_bases);
        if (!s->stg_ascii) { LRGE_SET_ERR(ctx, "seqset_upload: device allocation failed: %s", hipGetErrorString(e)); return LRGE_ERR_DEVICE; }
        d_ascii = (const u8 *)s->stg_ascii;
    }
    if (!s->ev_ready) s->ev_ready = ctx->get_event();
    hipStream_t cs = ctx->copy_stream;
    // the blocks just taken from the pool may still be in use by work queued on the main stream
    HIPCHK(ctx, hipEventRecord(ctx->ev_gate, ctx->stream));
    HIPCHK(ctx, hipStreamWaitEvent(cs, ctx->ev_gate, 0));
    s->pending = true;                                     // (from here on seqset_free drains the copy stream first)
    {
        if (!ctx->meta_pin && hipHostMalloc((void **)&ctx->meta_pin, (size_t)32 << 20, hipHostMallocDefault) == hipSuccess) ctx->meta_cap = (size_t)32 << 20;
        else if (!ctx->meta_pin) (void)hipGetLastError();
        char *hm = nullptr;
        if (ctx->meta_pin && ctx->meta_used + meta_bytes <= ctx->meta_cap) {
            // a rewound arena: every consumer of the earlier uploads has ordered itself behind them on the DEVICE
            // (seqset_ready); the host must not overwrite the bytes before the last transfer has actually read them
            // (it almost always has: ~2 us)
            if (ctx->meta_used == 0 && ctx->ev_meta) HIPCHK(ctx, hipEventSynchronize(ctx->ev_meta));
            hm = ctx->meta_pin + ctx->meta_used; ctx->meta_used += meta_bytes; ++ctx->meta_inflight; s->meta_arena = true;
        }
        auto put = [&](size_t off, const void *src_, size_t bytes) -> hipError_t {
            if (hm) { memcpy(hm + off, src_, bytes); return hipSuccess; }
            return hipMemcpyAsync(dm + off, src_, bytes, hipMemcpyHostToDevice, cs);       // (arena full: piecewise, from the set's own vectors)
        };
        HIPCHK(ctx, put(o_woff, s->h_woff.data(), ((size_t)n + 1) * 8));
        {   // k_pack's two maps exist for the upload only: base offset of every read relative to the first, and the read that
            // holds the first word of every block -- produced where they travel from (the arena; the set's own vectors when
            // it is full, because an asynchronous copy reads them after this call has returned)
            u64 *boff; u32 *blk;
            if (hm) { boff = (u64 *)(hm + o_boff); blk = (u32 *)(hm + o_blk); }
            else { s->h_boff.resize((size_t)n + 1); s->h_blk.resize((size_t)n_blk + 1); boff = s->h_boff.data(); blk = s->h_blk.data(); }
            const u64 o0 = n ? offsets[0] : 0;
            for (u32 i = 0; i <= n; ++i) boff[i] = n ? offsets[i] - o0 : 0;
            const u64 *hw = s->h_woff.data();
            u32 r = 0;
            for (u64 bq = 0; bq < n_blk; ++bq) {
                const u64 w0 = bq * PACK_WORDS;
                while (r + 1 < n && hw[r + 1] <= w0) ++r;
                blk[bq] = r;
            }
            blk[n_blk] = 0;
            if (!hm) {
                HIPCHK(ctx, put(o_boff, boff, ((size_t)n + 1) * 8));
                HIPCHK(ctx, put(o_blk, blk, (size_t)(n_blk + 1) * 4));
            }
        }
        if (s->n_chunks < (1ULL << 32)) HIPCHK(ctx, put(o_cs, s->h_cs.data(), ((size_t)n + 1) * 4));
        if (n) {
            HIPCHK(ctx, put(o_len, s->h_len.data(), (size_t)n * 4));
            if (name_rank) HIPCHK(ctx, put(o_rank, s->h_rank.data(), (size_t)n * 4));
        }
        if (hm) {
            HIPCHK(ctx, hipMemcpyAsync(dm, hm, meta_bytes, hipMemcpyHostToDevice, cs));
            if (!ctx->ev_meta) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_meta, hipEventDisableTiming));
            HIPCHK(ctx, hipEventRecord(ctx->ev_meta, cs));
        }
    }
    if (host_pack) {
        // pinned chunk buffers + the uploader thread, once per context
        const size_t CH = (size_t)ctx->opt_u64("HOST_PACK_CHUNK_WORDS", (u64)2 << 20);      // 64 Mbases per chunk
        if (!ctx->hp_stage[0] || ctx->hp_words != CH) {
            for (int b = 0; b < 2; ++b) {
                if (ctx->hp_stage[b]) { HIPCHK(ctx, hipStreamSynchronize(cs)); (void)hipHostFree(ctx->hp_stage[b]); ctx->hp_stage[b] = nullptr; }
                HIPCHK(ctx, hipHostMalloc((void **)&ctx->hp_stage[b], CH * 12, hipHostMallocDefault));
                if (!ctx->hp_ev[b]) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->hp_ev[b], hipEventDisableTiming));
                HIPCHK(ctx, hipEventRecord(ctx->hp_ev[b], cs));
            }
            ctx->hp_words = CH;
        }
        if (!ctx->uploader) {
            ctx->uploader = new Uploader();
            if (!ctx->opt("HOST_PACK_NO_PIN")) ctx->uploader->cpus = hp_gpu_node_cpus(ctx->device);     // the GPU's own NUMA node
            const u32 hw = ctx->uploader->cpus.empty() ? std::max(2u, std::thread::hardware_concurrency()) : (u32)ctx->uploader->cpus.size() * 2;
            ctx->uploader->pool.start((u32)ctx->opt_u64("HOST_PACK_THREADS", std::min<u32>(32, std::max<u32>(2, hw / 2))) - 1, ctx->uploader->cpus);
            if (ctx->opt("VERBOSE")) fprintf(stderr, "[lrge_hip] host-side pack: %zu worker threads on %zu CPUs of the GPU's NUMA node\n", ctx->uploader->pool.th.size(), ctx->uploader->cpus.size());
        }
        s->h_boff.resize((size_t)n + 1);
        { const u64 o0 = offsets[0]; for (u32 i = 0; i <= n; ++i) s->h_boff[i] = offsets[i] - o0; }
        auto job = std::make_shared<UploadJob>();
        s->job = job;
        for (u64 w0 = 0; w0 < w; w0 += CH) { job->gate_ev.push_back(ctx->get_event()); job->gate_w1.push_back(std::min<u64>(w, w0 + CH)); }
        hipEvent_t ev_ready = s->ev_ready;
        const int device = ctx->device;
        const u64 n_words = w;
        u64 *d_pack = s->d_pack; u32 *d_nmask = s->d_nmask;
        const u64 *boff = s->h_boff.data(), *woff = s->h_woff.data();
        Uploader *up = ctx->uploader;
        char **stage = ctx->hp_stage; hipEvent_t *sev = ctx->hp_ev;
        const u8 *hsrc = (const u8 *)src;
        const bool verbose = ctx->opt("VERBOSE") != nullptr;
        auto work = [=]() {
            hipError_t e = hipSetDevice(device);
            int b = 0;
            const double t_job = DevPool::now_ms(); double t_pack = 0, t_wait = 0;
            for (u64 w0 = 0; w0 < n_words && e == hipSuccess; w0 += CH, b ^= 1) {
                const u64 w1 = std::min<u64>(n_words, w0 + CH), nw = w1 - w0;
                const double t0 = DevPool::now_ms();
                e = hipEventSynchronize(sev[b]);                      // the DMA that last read this buffer
                if (e != hipSuccess) break;
                const double t1 = DevPool::now_ms(); t_wait += t1 - t0;
                u64 *hp = (u64 *)stage[b]; u32 *hm = (u32 *)(stage[b] + CH * 8);
                const u32 n_tasks = (u32)std::min<u64>(256, std::max<u64>(1, nw / 16384));
                up->pool.parallel_for(n_tasks, [=](u32 t) {
                    const u64 a = w0 + nw * t / n_tasks, z = w0 + nw * (t + 1) / n_tasks;
                    hp_pack_range(hsrc, boff, woff, n, a, z, hp + (a - w0), hm + (a - w0));
                });
                t_pack += DevPool::now_ms() - t1;
                e = hipMemcpyAsync(d_pack + w0, hp, nw * 8, hipMemcpyHostToDevice, cs);
                if (e == hipSuccess) e = hipMemcpyAsync(d_nmask + w0, hm, nw * 4, hipMemcpyHostToDevice, cs);
                if (e == hipSuccess) e = hipEventRecord(sev[b], cs);
                if (e == hipSuccess) { e = hipEventRecord(job->gate_ev[(size_t)(w0 / CH)], cs); if (e == hipSuccess) job->gate_recorded(); }
            }
            if (e == hipSuccess) e = hipEventRecord(ev_ready, cs);
            if (verbose) fprintf(stderr, "[lrge_hip] host-side pack of %llu words: job %.2f ms on the uploader thread (packing %.2f ms, waiting for a chunk buffer %.2f ms)\n",
                                 (unsigned long long)n_words, DevPool::now_ms() - t_job, t_pack, t_wait);
            job->finish(e == hipSuccess ? LRGE_OK : LRGE_ERR_DEVICE, e == hipSuccess ? std::string() : std::string("host-side pack / upload: ") + hipGetErrorString(e));
        };
        // a pinned source stays valid until the set is consumed (the contract of the async form): the job runs in the
        // background.  A pageable source may change as soon as this call returns, and the blocking form waits anyway.
        if (async && kind == 1) up->submit(work);
        else { up->submit(work); const int jrc = seqset_job_wait(ctx, s); if (jrc) return jrc; }
    } else if (kind == 1) {
        HIPCHK(ctx, hipMemcpyAsync(s->stg_ascii, src, s->total_bases, hipMemcpyHostToDevice, cs));
    } else if (kind == 2 && s->total_bases) {
        if (!ctx->stage_cap) {       // both buffers and both events, or nothing (a half-made pair would fail every later upload)
            const size_t cap = (size_t)64 << 20;
            char *bufs[2] = {nullptr, nullptr}; hipEvent_t evs[2] = {nullptr, nullptr};
            hipError_t se = hipSuccess;
            for (int b = 0; b < 2 && se == hipSuccess; ++b) {
                se = hipHostMalloc((void **)&bufs[b], cap, hipHostMallocDefault);
                if (se == hipSuccess) se = hipEventCreateWithFlags(&evs[b], hipEventDisableTiming);
                if (se == hipSuccess) se = hipEventRecord(evs[b], cs);
            }
            if (se != hipSuccess) {
                for (int b = 0; b < 2; ++b) { if (bufs[b]) (void)hipHostFree(bufs[b]); if (evs[b]) (void)hipEventDestroy(evs[b]); }
                (void)hipGetLastError();
                LRGE_SET_ERR(ctx, "seqset_upload: pinned staging buffers: %s", hipGetErrorString(se));
                return LRGE_ERR_DEVICE;
            }
            for (int b = 0; b < 2; ++b) { ctx->stage[b] = bufs[b]; ctx->stage_ev[b] = evs[b]; }
            ctx->stage_cap = cap;
        }
        int b = 0;
        for (u64 o = 0; o < s->total_bases; o += ctx->stage_cap, b ^= 1) {
            const size_t len = (size_t)std::min<u64>(ctx->stage_cap, s->total_bases - o);
            HIPCHK(ctx, hipEventSynchronize(ctx->stage_ev[b]));          // the DMA that last read this buffer
            parallel_memcpy(ctx->stage[b], src + o, len);
            HIPCHK(ctx, hipMemcpyAsync((char *)s->stg_ascii + o, ctx->stage[b], len, hipMemcpyHostToDevice, cs));
            HIPCHK(ctx, hipEventRecord(ctx->stage_ev[b], cs));
        }
    }
    if (w && !host_pack) {
        // (timed only in the blocking form: a pending event pair would make the next call's timer resolution wait for
        // this upload on the host)
        std::unique_ptr<StageTimer> t(async ? nullptr : new StageTimer(ctx, LRGE_T_PACK, cs));
        hipLaunchKernelGGL(k_pack, dim3((u32)n_blk), dim3(PACK_THREADS), 0, cs, d_ascii, (const u64 *)s->stg_boff, s->d_woff,
                           (const u32 *)s->stg_blk, n, w, s->d_pack, s->d_nmask);
        KCHK(ctx);
    }
    if (!host_pack) HIPCHK(ctx, hipEventRecord(s->ev_ready, cs));      // (a host-side pack records it at the end of its job)
    // async: the per-read arrays travel from the set's own host copies (they live as long as the set); only `bases`
    // must stay valid, and only when it is pinned host memory (a pageable source has been copied out by now)
    if (!async) {
        HIPCHK(ctx, hipStreamSynchronize(cs));
        ctx->resolve_timers();
        // the set is complete: its 1 B/base ASCII staging block goes back now, not when somebody consumes the set
        s->pending = false;
        if (s->meta_arena) { s->meta_arena = false; if (--ctx->meta_inflight == 0) ctx->meta_used = 0; }
        ctx->pool.release(s->stg_ascii); s->stg_ascii = nullptr;
    }
    if (ctx->opt("VERBOSE")) fprintf(stderr, "[lrge_hip] upload of %u reads: %.3f ms of host time\n", n,
                                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
    *out = guard.release();
    return LRGE_OK;
}

extern "C" int lrge_hip_seqset_upload(lrge_hip_ctx *ctx, const char *bases, const uint64_t *offsets, uint32_t n,
                                      const uint32_t *name_rank, lrge_hip_seqset **out) {
    return seqset_upload_impl(ctx, bases, offsets, n, name_rank, false, out);
}
extern "C" int lrge_hip_seqset_upload_async(lrge_hip_ctx *ctx, const char *bases, const uint64_t *offsets, uint32_t n,
                                            const uint32_t *name_rank, lrge_hip_seqset **out) {
    return seqset_upload_impl(ctx, bases, offsets, n, name_rank, true, out);
}
extern "C" int lrge_hip_seqset_wait(lrge_hip_seqset *s) {
    if (!s) return LRGE_ERR_INVALID;
    lrge_hip_ctx *ctx = s->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    { int jrc = seqset_job_wait(ctx, s); if (jrc) return jrc; }
    if (s->pending) HIPCHK(ctx, hipEventSynchronize(s->ev_ready));
    return LRGE_OK;
}

static void presketch_drop_prepared(lrge_hip_ctx *ctx);
static void presketch_discard(lrge_hip_seqset *s) {
    lrge_hip_ctx *ctx = s->ctx;
    if (ctx->presk_pending == s) ctx->presk_pending = nullptr;
    if (ctx->presk_prepared_set == s) presketch_drop_prepared(ctx);
    if (!s->presk) return;
    (void)hipStreamSynchronize(ctx->stream2);          // its kernels may still be running
    delete s->presk->sc;
    ctx->event_pool.push_back(s->presk->ev_start); ctx->event_pool.push_back(s->presk->ev_done);
    delete s->presk;
    s->presk = nullptr;
}

extern "C" void lrge_hip_seqset_free(lrge_hip_seqset *s) {
    if (!s) return;
    bool ctx_alive;
    { std::lock_guard<std::mutex> g(g_live_mu); ctx_alive = g_live_ctx.count(s->ctx) != 0; }
    if (ctx_alive) { (void)hipSetDevice(s->ctx->device); presketch_discard(s); }   // (a set that outlives its context only owns its own arrays)
    if (s->is_view) { (void)hipFree(s->d_cs); delete s; return; }                   // a view owns its chunk map only
    if (s->pooled) {
        if (ctx_alive) {       // (a destroyed context has already freed its pool)
            lrge_hip_ctx *ctx = s->ctx;
            if (s->job) (void)seqset_job_wait(ctx, s);
            if (s->pending) (void)hipStreamSynchronize(ctx->copy_stream);          // an upload nobody consumed
            DevPool &P = ctx->pool;
            if (s->meta_arena && --ctx->meta_inflight == 0) ctx->meta_used = 0;
            P.release(s->d_pack); P.release(s->d_nmask); P.release(s->d_meta); P.release(s->stg_ascii);
            if (s->ev_ready) ctx->event_pool.push_back(s->ev_ready);
        }
    } else {
        (void)hipFree(s->d_pack); (void)hipFree(s->d_nmask); (void)hipFree(s->d_woff); (void)hipFree(s->d_len); (void)hipFree(s->d_rank); (void)hipFree(s->d_cs);
    }
    delete s;
}
extern "C" uint32_t lrge_hip_seqset_size(const lrge_hip_seqset *s) { return s ? s->n : 0; }

// ------------------------------------------------------------------------------------------
// sketch driver
// ------------------------------------------------------------------------------------------
struct SketchOut {
    u64 *x = nullptr, *y = nullptr;   // pool memory (owned by the caller's Scratch)
    u32 *mz_off = nullptr;            // [n+1] per-read offsets
    u64 n = 0;
};

// pk_ybits != 0 (index only): packed 8-byte entries in o->x, o->y stays null (k_sketch.h PK)
template <int K, int W, bool HPC>
static int sketch_launch(lrge_hip_ctx *ctx, Scratch &sc, const lrge_hip_seqset *s, bool index_keys, SketchOut *o, u32 pk_pos1, u32 pk_ybits,
                         std::vector<u32> *h_mzoff, bool gated = false) {
    // gated: the caller has NOT waited for the set's upload (seqset_ready): this function does, as late as it can
    if (s->n_chunks >= (1ULL << 32)) { LRGE_SET_ERR(ctx, "read set too large for one sketch launch"); return LRGE_ERR_TOO_MANY; }
    u32 n_chunks = (u32)s->n_chunks;
    const u32 *d_cs = s->d_cs;           // chunk map, uploaded with the set
    const bool pk = index_keys && pk_ybits;
    ALLOC_OR_FAIL(d_cnt, sc, u32, (size_t)n_chunks + 1);
    ALLOC_OR_FAIL(d_total, sc, u32, 2);  // [1] = overflow flag of the one-pass form
    ALLOC_OR_FAIL(d_mzoff, sc, u32, (size_t)s->n + 1);
    ChunkMap cm{d_cs, s->n};
    const dim3 sgrid((u32)div_up(n_chunks, SK_THREADS));
    // One pass (k_sketch_direct into per-chunk slots, then k_sketch_compact) when the slots fit comfortably; the
    // two-pass form (count, scan, write) otherwise, when a chunk overflows its slot, or on request.
    const u64 slot_bytes = (u64)n_chunks * SK_CAP * 8 * (pk ? 1 : 2);
    size_t mfree = (size_t)64 << 30, mtot = 0;
    if (slot_bytes > ((u64)4 << 30)) (void)hipMemGetInfo(&mfree, &mtot);       // (small sets: no need to ask)
    bool one_pass = n_chunks && !ctx->opt("SKETCH_TWO_PASS") && slot_bytes < ((u64)mfree + ctx->pool.idle()) / 4;
    const char *cap_env = ctx->opt("DEBUG_SK_CAP");                      // tests: force the overflow fallback
    const u32 sk_cap = cap_env ? (u32)std::min<u64>(strtoull(cap_env, nullptr, 10), SK_CAP) : (u32)SK_CAP;
    u64 *tx = nullptr, *ty = nullptr;
    if (one_pass) {
        tx = sc.get<u64>((size_t)n_chunks * SK_CAP);
        ty = pk ? nullptr : sc.get<u64>((size_t)n_chunks * SK_CAP);
        if (!tx || (!pk && !ty)) { if (tx) sc.drop(tx); if (ty) sc.drop(ty); tx = ty = nullptr; one_pass = false; (void)hipGetLastError(); }
    }
    u32 tot_ovf[2] = {0, 0};
    for (int pass = 0; pass < 2; ++pass) {       // second round only after a slot overflow
        HIPCHK(ctx, hipMemsetAsync(d_total, 0, 8, ctx->stream));
        if (n_chunks) {
            if (one_pass && gated && pass == 0) {
                // the set's upload is still in flight (host-side pack, chunk after chunk): the sketch chunks that lie wholly inside
                // the words of upload chunk j run behind gate j, while the later chunks are still being packed and sent
                lrge_hip_seqset *ms = const_cast<lrge_hip_seqset *>(s);
                std::shared_ptr<UploadJob> job = ms->job;
                u32 c_prev = 0;
                const size_t ng = job->gate_w1.size();
                for (size_t j = 0; j < ng; ++j) {
                    if (!job->wait_gate((int)j)) break;                          // (the job failed: seqset_ready below reports it)
                    HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, job->gate_ev[j], 0));
                    const u64 w1 = job->gate_w1[j];
                    u32 c_end = n_chunks;
                    if (w1 < s->n_words) {
                        const u32 r = (u32)(std::upper_bound(s->h_woff.begin(), s->h_woff.end(), w1) - s->h_woff.begin()) - 1;
                        const u64 avail = w1 - s->h_woff[r];                      // words of read r that have arrived: 4 per 128-base chunk
                        c_end = s->h_cs[r] + (u32)std::min<u64>(avail / (SK_CHUNK / 32), (u64)(s->h_cs[r + 1] - s->h_cs[r]));
                    }
                    if (c_end > c_prev) {
                        const dim3 g((u32)div_up(c_end - c_prev, SK_THREADS));
                        if (pk) hipLaunchKernelGGL((k_sketch_direct<K, W, HPC, true, true>), g, dim3(SK_THREADS), 0, ctx->stream, s->d_pack, s->d_nmask,
                                                   s->d_woff, s->d_len, cm, c_end, d_cnt, d_total + 1, tx, ty, pk_pos1, pk_ybits, sk_cap, c_prev);
                        else hipLaunchKernelGGL((k_sketch_direct<K, W, HPC, true, false>), g, dim3(SK_THREADS), 0, ctx->stream, s->d_pack,
                                                s->d_nmask, s->d_woff, s->d_len, cm, c_end, d_cnt, d_total + 1, tx, ty, 0u, 0u, sk_cap, c_prev);
                        KCHK(ctx);
                        c_prev = c_end;
                    }
                }
                int rr = seqset_ready(ctx, s); if (rr) return rr;
                if (c_prev < n_chunks) {                                          // (whatever a failed / odd gate sequence left)
                    const dim3 g((u32)div_up(n_chunks - c_prev, SK_THREADS));
                    if (pk) hipLaunchKernelGGL((k_sketch_direct<K, W, HPC, true, true>), g, dim3(SK_THREADS), 0, ctx->stream, s->d_pack, s->d_nmask,
                                               s->d_woff, s->d_len, cm, n_chunks, d_cnt, d_total + 1, tx, ty, pk_pos1, pk_ybits, sk_cap, c_prev);
                    else hipLaunchKernelGGL((k_sketch_direct<K, W, HPC, true, false>), g, dim3(SK_THREADS), 0, ctx->stream, s->d_pack,
                                            s->d_nmask, s->d_woff, s->d_len, cm, n_chunks, d_cnt, d_total + 1, tx, ty, 0u, 0u, sk_cap, c_prev);
                    KCHK(ctx);
                }
            } else if (one_pass) {
                if (gated && pass == 0) { int rr = seqset_ready(ctx, s); if (rr) return rr; }
                if (pk) hipLaunchKernelGGL((k_sketch_direct<K, W, HPC, true, true>), sgrid, dim3(SK_THREADS), 0, ctx->stream, s->d_pack, s->d_nmask,
                                           s->d_woff, s->d_len, cm, n_chunks, d_cnt, d_total + 1, tx, ty, pk_pos1, pk_ybits, sk_cap);
                else if (index_keys) hipLaunchKernelGGL((k_sketch_direct<K, W, HPC, true, false>), sgrid, dim3(SK_THREADS), 0, ctx->stream, s->d_pack,
                                                        s->d_nmask, s->d_woff, s->d_len, cm, n_chunks, d_cnt, d_total + 1, tx, ty, 0u, 0u, sk_cap);
                else hipLaunchKernelGGL((k_sketch_direct<K, W, HPC, false, false>), sgrid, dim3(SK_THREADS), 0, ctx->stream, s->d_pack, s->d_nmask,
                                        s->d_woff, s->d_len, cm, n_chunks, d_cnt, d_total + 1, tx, ty, 0u, 0u, sk_cap);
            } else {
                if (gated && pass == 0) { int rr = seqset_ready(ctx, s); if (rr) return rr; }
                hipLaunchKernelGGL((k_sketch_count<K, W, HPC>), sgrid, dim3(SK_THREADS), 0, ctx->stream, s->d_pack, s->d_nmask, s->d_woff, s->d_len, cm,
                                   n_chunks, d_cnt);
            }
            KCHK(ctx);
            int rc = scan_exclusive_u32(ctx, sc, d_cnt, d_cnt, n_chunks, d_total);
            if (rc) return rc;
        }
        // per-read offsets follow from the chunk scan alone: they travel to the host with the total, in the one sync
        hipLaunchKernelGGL(k_read_mz_offsets, dim3((u32)div_up((u64)s->n + 1, 256)), dim3(256), 0, ctx->stream, d_cs, d_cnt, s->n,
                           n_chunks, d_total, d_mzoff);
        KCHK(ctx);
        HIPCHK(ctx, ctx->d2h(tot_ovf, d_total, 8, ctx->stream));
        if (h_mzoff) {
            h_mzoff->resize((size_t)s->n + 1);
            HIPCHK(ctx, ctx->d2h(h_mzoff->data(), d_mzoff, ((size_t)s->n + 1) * 4, ctx->stream));
        }
        HIPCHK(ctx, ctx->d2h_sync(ctx->stream));
        if (!(one_pass && tot_ovf[1])) break;
        one_pass = false;                        // a chunk held more than SK_CAP minimizers: redo in two passes
        sc.drop(tx); if (ty) sc.drop(ty); tx = ty = nullptr;
    }
    const u32 total = tot_ovf[0];
    ALLOC_OR_FAIL(dx, sc, u64, (size_t)total + 1);
    u64 *dy = nullptr;
    if (!pk) { dy = sc.get<u64>((size_t)total + 1); if (!dy) return LRGE_ERR_DEVICE; }
    if (n_chunks && one_pass) {
        const dim3 cgrid((u32)div_up(div_up(n_chunks, 64), 4));
        if (pk) hipLaunchKernelGGL(k_sketch_compact<false>, cgrid, dim3(256), 0, ctx->stream, tx, ty, d_cnt, d_total, n_chunks, dx, dy);
        else hipLaunchKernelGGL(k_sketch_compact<true>, cgrid, dim3(256), 0, ctx->stream, tx, ty, d_cnt, d_total, n_chunks, dx, dy);
        KCHK(ctx);
        sc.drop(tx); if (ty) sc.drop(ty);
    } else if (n_chunks) {
        if (pk)
            hipLaunchKernelGGL((k_sketch_write<K, W, HPC, true, true>), sgrid, dim3(SK_THREADS), 0,
                               ctx->stream, s->d_pack, s->d_nmask, s->d_woff, s->d_len, cm, n_chunks, d_cnt, dx, dy, pk_pos1, pk_ybits);
        else if (index_keys)
            hipLaunchKernelGGL((k_sketch_write<K, W, HPC, true, false>), sgrid, dim3(SK_THREADS), 0,
                               ctx->stream, s->d_pack, s->d_nmask, s->d_woff, s->d_len, cm, n_chunks, d_cnt, dx, dy, 0u, 0u);
        else
            hipLaunchKernelGGL((k_sketch_write<K, W, HPC, false, false>), sgrid, dim3(SK_THREADS), 0,
                               ctx->stream, s->d_pack, s->d_nmask, s->d_woff, s->d_len, cm, n_chunks, d_cnt, dx, dy, 0u, 0u);
        KCHK(ctx);
    }
    // (no sync: everything runs in order on ctx->stream; scratch is recycled in stream order)
    sc.drop(d_cnt); sc.drop(d_total);
    o->x = dx; o->y = dy; o->mz_off = d_mzoff; o->n = total;
    return LRGE_OK;
}

static int sketch_device(lrge_hip_ctx *ctx, Scratch &sc, const lrge_hip_seqset *s, int preset, bool index_keys, SketchOut *o,
                         u32 pk_pos1 = 0, u32 pk_ybits = 0, std::vector<u32> *h_mzoff = nullptr) {
    // A set whose host-side pack is still running on the uploader thread (chunk gates: host_pack.h) is sketched chunk by chunk
    // behind its transfer -- index sketches of the non-HPC preset only (an HPC step may read a homopolymer run past its chunk,
    // i.e. words that have not arrived; a streamed set's upload hides behind the index build anyway).  option NO_GATED_SKETCH: wait first.
    const bool gated = index_keys && preset != LRGE_PRESET_AVA_PB && s->pending && s->job && !s->job->gate_ev.empty() && s->n_words != 0 &&
                       !s->is_view && !ctx->opt("NO_GATED_SKETCH") && s->n_chunks != 0 && s->n_chunks < (1ULL << 32);
    int rc = gated ? LRGE_OK : seqset_ready(ctx, s);
    if (rc) return rc;
    StageTimer t(ctx, LRGE_T_SKETCH);
    rc = (preset == LRGE_PRESET_AVA_PB) ? sketch_launch<19, 5, true>(ctx, sc, s, index_keys, o, pk_pos1, pk_ybits, h_mzoff, false)
                                            : sketch_launch<15, 5, false>(ctx, sc, s, index_keys, o, pk_pos1, pk_ybits, h_mzoff, gated);
    t.stop();
    return rc;
}

// ---- presketch: the streamed set's minimizers, computed on the side stream with no host round trip ----
// Two steps, because the device arena recycles blocks in the order of the MAIN stream: everything the side stream will touch
// is allocated where it forks (presketch_prepare: nothing released by the index build after that point can be handed to it),
// the kernels may be queued later (presketch_launch_prepared).
static int presketch_alloc(lrge_hip_ctx *ctx, const lrge_hip_seqset *s, PreSketch *p) {
    if (s->n_chunks >= (1ULL << 32) || s->total_bases + 1 >= (1ULL << 32)) return LRGE_ERR_TOO_MANY;
    Scratch &sc = *p->sc;
    const u64 nb = div_up(s->n_chunks, SCAN_TILE);
    if (nb > 8192) return LRGE_ERR_TOO_MANY;                   // (single-level scan with the caller's block sums)
    ALLOC_OR_FAIL(d_cnt, sc, u32, (size_t)s->n_chunks + 1);
    ALLOC_OR_FAIL(d_bs, sc, u32, (size_t)nb + 2);
    ALLOC_OR_FAIL(d_total, sc, u32, 1);
    ALLOC_OR_FAIL(d_mzoff, sc, u32, (size_t)s->n + 1);
    // the count is not known on the host when the write pass is queued: room for one minimizer per base
    ALLOC_OR_FAIL(dx, sc, u64, (size_t)s->total_bases + 1);
    ALLOC_OR_FAIL(dy, sc, u64, (size_t)s->total_bases + 1);
    p->cnt = d_cnt; p->bs = d_bs; p->x = dx; p->y = dy; p->mz_off = d_mzoff; p->d_total = d_total;
    return LRGE_OK;
}

template <int K, int W, bool HPC>
static int presketch_launch(lrge_hip_ctx *ctx, const lrge_hip_seqset *s, PreSketch *p, hipStream_t st) {
    Scratch &sc = *p->sc;
    const u32 n_chunks = (u32)s->n_chunks;
    ChunkMap cm{s->d_cs, s->n};
    const dim3 sgrid((u32)div_up(n_chunks, SK_THREADS));
    // Two passes here, not the one-pass form of sketch_launch: this runs beside the index's memory-bound sort passes, and
    // a second VALU-bound pass overlaps with them where the one-pass form's streaming compaction competes (measured:
    // the sort loses what the sketch gains).
    if (n_chunks) {
        hipLaunchKernelGGL((k_sketch_count<K, W, HPC>), sgrid, dim3(SK_THREADS), 0, st, s->d_pack, s->d_nmask, s->d_woff, s->d_len, cm, n_chunks, p->cnt);
        KCHK(ctx);
        int rc = scan_exclusive_u32(ctx, sc, p->cnt, p->cnt, n_chunks, p->d_total, st, true, p->bs);
        if (rc) return rc;
    } else {
        HIPCHK(ctx, hipMemsetAsync(p->d_total, 0, 4, st));
    }
    hipLaunchKernelGGL(k_read_mz_offsets, dim3((u32)div_up((u64)s->n + 1, 256)), dim3(256), 0, st, s->d_cs, p->cnt, s->n, n_chunks, p->d_total,
                       p->mz_off);
    KCHK(ctx);
    if (n_chunks) {
        hipLaunchKernelGGL((k_sketch_write<K, W, HPC, false, false>), sgrid, dim3(SK_THREADS), 0, st, s->d_pack, s->d_nmask, s->d_woff, s->d_len, cm,
                           n_chunks, p->cnt, p->x, p->y, 0u, 0u);
        KCHK(ctx);
    }
    return LRGE_OK;
}

static void presketch_drop_prepared(lrge_hip_ctx *ctx) {
    PreSketch *p = ctx->presk_prepared;
    if (!p) return;
    ctx->presk_prepared = nullptr; ctx->presk_prepared_set = nullptr;
    delete p->sc;                                        // (nothing has been queued on these blocks)
    ctx->event_pool.push_back(p->ev_start); ctx->event_pool.push_back(p->ev_done);
    delete p;
}

// Called by the index build right after its own sketch has been queued on ctx->stream: marks the point of the main stream
// the side stream starts from and takes the memory of the streamed set's sketch.
// indexed_bases: size of the set whose index build would hide the sketch.  A streamed set several times larger than the
// indexed one (the inverse strategy on a big job: 3 Gbases streamed against a 150 Mbase index) finds nothing to hide behind --
// the two VALU-bound sketches and the small sort just share the chip -- so the hint is ignored there and the overlap call
// sketches in line (C5/10 inverse: 95 -> 89 ms per step).
static int presketch_prepare(lrge_hip_ctx *ctx, u64 indexed_bases) {
    presketch_drop_prepared(ctx);
    lrge_hip_seqset *s = ctx->presk_pending;
    if (!s) return LRGE_OK;
    ctx->presk_pending = nullptr;
    if (s->total_bases > 2 * indexed_bases && !ctx->opt("PRESKETCH_ALWAYS")) return LRGE_OK;
    if (s->total_bases > ctx->opt_u64("STREAM_BASES", 4000000000ull)) return LRGE_OK;   // streamed in views: sketched per view
    if (s->presk) presketch_discard(s);
    PreSketch *p = new PreSketch();
    p->preset = ctx->presk_preset;
    p->sc = new Scratch(ctx);
    p->ev_start = ctx->get_event(); p->ev_done = ctx->get_event();
    ctx->presk_prepared = p; ctx->presk_prepared_set = s;
    // behind the index sketch (both are VALU-bound; the point is to run beside the passes that follow it)
    if (presketch_alloc(ctx, s, p) != LRGE_OK || hipEventRecord(ctx->ev_presk, ctx->stream) != hipSuccess) {
        (void)hipGetLastError();
        presketch_drop_prepared(ctx);                    // not fatal: the overlap call sketches the set itself
    }
    return LRGE_OK;
}

// Queues the prepared sketch on the side stream.  May block on the HOST until the set's upload job (host-side pack) is over,
// which is why the index build calls it only once it has nothing more of its own to queue that could run meanwhile.
static int presketch_launch_prepared(lrge_hip_ctx *ctx) {
    PreSketch *p = ctx->presk_prepared; lrge_hip_seqset *s = ctx->presk_prepared_set;
    if (!p) return LRGE_OK;
    hipError_t e = hipStreamWaitEvent(ctx->stream2, ctx->ev_presk, 0);
    // an upload of the set still in flight: only the side stream waits for it -- the main stream goes on with the index
    // (its own seqset_ready comes with the overlap call, which also returns the staging blocks to the pool)
    if (s->job && seqset_job_wait(ctx, s) != LRGE_OK) { presketch_drop_prepared(ctx); return LRGE_OK; }
    if (e == hipSuccess && s->pending) e = hipStreamWaitEvent(ctx->stream2, s->ev_ready, 0);
    if (e == hipSuccess) e = hipEventRecord(p->ev_start, ctx->stream2);
    int rc = LRGE_OK;
    if (e == hipSuccess) {
        rc = p->preset == LRGE_PRESET_AVA_PB ? presketch_launch<19, 5, true>(ctx, s, p, ctx->stream2)
                                             : presketch_launch<15, 5, false>(ctx, s, p, ctx->stream2);
        if (rc == LRGE_OK) e = hipEventRecord(p->ev_done, ctx->stream2);
    }
    if (e != hipSuccess || rc != LRGE_OK) {      // not fatal: the overlap call sketches the set itself
        (void)hipStreamSynchronize(ctx->stream2);
        (void)hipGetLastError();
        presketch_drop_prepared(ctx);
        return LRGE_OK;
    }
    ctx->presk_prepared = nullptr; ctx->presk_prepared_set = nullptr;
    s->presk = p;
    return LRGE_OK;
}

static int presketch_start_pending(lrge_hip_ctx *ctx, u64 indexed_bases) {
    int rc = presketch_prepare(ctx, indexed_bases);
    return rc ? rc : presketch_launch_prepared(ctx);
}

extern "C" int lrge_hip_seqset_presketch(lrge_hip_ctx *ctx, lrge_hip_seqset *s, int preset) {
    if (!ctx || !s || s->ctx != ctx) return LRGE_ERR_INVALID;
    if (preset != LRGE_PRESET_AVA_ONT && preset != LRGE_PRESET_AVA_PB) { LRGE_SET_ERR(ctx, "Preset not found: %d", preset); return LRGE_ERR_INVALID; }
    ctx->presk_pending = s; ctx->presk_preset = preset;
    return LRGE_OK;
}

extern "C" int lrge_hip_sketch_dump(lrge_hip_ctx *ctx, const lrge_hip_seqset *s, int preset, uint64_t *x, uint64_t *y,
                                    uint64_t cap, uint64_t *n_out) {
    if (!ctx || !s || !n_out) return LRGE_ERR_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    ctx->pin_items.clear(); ctx->pin_used = 0;      // reads an earlier, failed call may have left queued
    Scratch sc(ctx);
    SketchOut o;
    int rc = sketch_device(ctx, sc, s, preset, false, &o);
    if (rc) return rc;
    *n_out = o.n;
    u64 m = o.n < cap ? o.n : cap;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));   // blocking copies below run on the null stream
    if (m && x) HIPCHK(ctx, hipMemcpy(x, o.x, m * 8, hipMemcpyDeviceToHost));
    if (m && y) HIPCHK(ctx, hipMemcpy(y, o.y, m * 8, hipMemcpyDeviceToHost));
    return LRGE_OK;
}


// ------------------------------------------------------------------------------------------
// index
// ------------------------------------------------------------------------------------------
extern "C" void lrge_hip_index_free(lrge_hip_index *ix);
struct IndexFree { void operator()(lrge_hip_index *ix) const { lrge_hip_index_free(ix); } };
typedef std::unique_ptr<lrge_hip_index, IndexFree> IndexGuard;     // every early return releases what the index holds so far

// Restricted build, fast form: the key-set test inside the one-pass sketch (k_sketch_restrict).  *done = false when the
// slots do not fit or a chunk overflowed its slot: the caller then takes the general form (full sketch, first sort pass,
// filter sweeps).  On success o->x [, o->y] hold the kept entries (o->n of them), *hashes / *n_hashes the owned hashes.
template <int K, int W, bool HPC>
static int sketch_restrict_launch(lrge_hip_ctx *ctx, Scratch &sc, const lrge_hip_seqset *s, bool pk, u32 pk_pos1, u32 pk_ybits, KeySet ks,
                                  u32 rank, u32 world, SketchOut *o, u64 **hashes, u64 *n_hashes, bool *done) {
    *done = false;
    if (s->n_chunks >= (1ULL << 32) || s->n_chunks == 0 || ctx->opt("SKETCH_TWO_PASS")) return LRGE_OK;
    const u32 n_chunks = (u32)s->n_chunks;
    const u64 slot_bytes = (u64)n_chunks * SK_CAP * 8 * (pk ? 2 : 3);
    size_t mfree = (size_t)64 << 30, mtot = 0;
    if (slot_bytes > ((u64)4 << 30)) (void)hipMemGetInfo(&mfree, &mtot);
    if (slot_bytes >= ((u64)mfree + ctx->pool.idle()) / 4) return LRGE_OK;
    const u32 sk_cap = ctx->opt("DEBUG_SK_CAP") ? (u32)std::min<u64>(ctx->opt_u64("DEBUG_SK_CAP", SK_CAP), SK_CAP) : (u32)SK_CAP;
    u64 *tx = sc.get<u64>((size_t)n_chunks * SK_CAP), *ty = pk ? nullptr : sc.get<u64>((size_t)n_chunks * SK_CAP);
    u64 *th = sc.get<u64>((size_t)n_chunks * SK_CAP);
    auto drop_slots = [&]() { if (tx) sc.drop(tx); if (ty) sc.drop(ty); if (th) sc.drop(th); };
    if (!tx || (!pk && !ty) || !th) { drop_slots(); (void)hipGetLastError(); return LRGE_OK; }
    ALLOC_OR_FAIL(ck, sc, u32, (size_t)n_chunks + 1); ALLOC_OR_FAIL(co, sc, u32, (size_t)n_chunks + 1); ALLOC_OR_FAIL(d_tot, sc, u32, 3);
    HIPCHK(ctx, hipMemsetAsync(d_tot, 0, 12, ctx->stream));
    ChunkMap cm{s->d_cs, s->n};
    const dim3 sgrid((u32)div_up(n_chunks, SK_THREADS));
    if (pk) hipLaunchKernelGGL((k_sketch_restrict<K, W, HPC, true>), sgrid, dim3(SK_THREADS), 0, ctx->stream, s->d_pack, s->d_nmask, s->d_woff, s->d_len, cm,
                               n_chunks, ck, co, d_tot + 2, tx, ty, th, pk_pos1, pk_ybits, sk_cap, ks, rank, world);
    else hipLaunchKernelGGL((k_sketch_restrict<K, W, HPC, false>), sgrid, dim3(SK_THREADS), 0, ctx->stream, s->d_pack, s->d_nmask, s->d_woff, s->d_len, cm,
                            n_chunks, ck, co, d_tot + 2, tx, ty, th, 0u, 0u, sk_cap, ks, rank, world);
    KCHK(ctx);
    int rc = scan_exclusive_u32(ctx, sc, ck, ck, n_chunks, d_tot); if (rc) return rc;
    rc = scan_exclusive_u32(ctx, sc, co, co, n_chunks, d_tot + 1); if (rc) return rc;
    u32 tot[3] = {0, 0, 0};
    HIPCHK(ctx, ctx->d2h(tot, d_tot, 12, ctx->stream));
    HIPCHK(ctx, ctx->d2h_sync(ctx->stream));
    if (tot[2]) { drop_slots(); sc.drop(ck); sc.drop(co); sc.drop(d_tot); return LRGE_OK; }    // a slot overflowed: general form
    ALLOC_OR_FAIL(dx, sc, u64, (size_t)tot[0] + 1);
    u64 *dy = nullptr;
    if (!pk) { dy = sc.get<u64>((size_t)tot[0] + 1); if (!dy) return LRGE_ERR_DEVICE; }
    ALLOC_OR_FAIL(dh, sc, u64, (size_t)tot[1] + 1);
    const dim3 cgrid((u32)div_up(div_up(n_chunks, 64), 4));
    if (pk) hipLaunchKernelGGL(k_sketch_compact<false>, cgrid, dim3(256), 0, ctx->stream, tx, ty, ck, d_tot, n_chunks, dx, dy);
    else hipLaunchKernelGGL(k_sketch_compact<true>, cgrid, dim3(256), 0, ctx->stream, tx, ty, ck, d_tot, n_chunks, dx, dy);
    KCHK(ctx);
    hipLaunchKernelGGL(k_sketch_compact<false>, cgrid, dim3(256), 0, ctx->stream, th, (const u64 *)nullptr, co, d_tot + 1, n_chunks, dh, (u64 *)nullptr);
    KCHK(ctx);
    drop_slots(); sc.drop(ck); sc.drop(co); sc.drop(d_tot);
    o->x = dx; o->y = dy; o->mz_off = nullptr; o->n = tot[0];
    *hashes = dh; *n_hashes = tot[1];
    *done = true;
    return LRGE_OK;
}

// A restricted build (lrge_hip_index_build_for, k_restrict.h): the index holds the entries of the keys that occur in
// `restrict_to`'s minimizers, its statistics (mid_occ, key and minimizer totals) are those of the whole target set.
struct IndexBuildOpts {
    lrge_hip_seqset *restrict_to = nullptr; lrge_hip_comm *comm = nullptr;
    // sharded target sketch (lrge_hip_index_build_sharded, k_route.h): this rank's contiguous share of the target reads, whose
    // first read is read `shard_first` of the whole set (`targets` then describes the whole set: lengths and names, no bases)
    const lrge_hip_seqset *shard = nullptr; u32 shard_first = 0;
};

// Work counters of the last sharded build on a context (exchange volumes, for the projection tables of DESIGN.md section 7)
struct ShardStats { u64 keyset_bytes = 0, entries_sketched = 0, entries_sent = 0, entries_recv = 0, hashes_sent = 0, hashes_recv = 0; };
static thread_local ShardStats g_shard_stats;

// A collective call must fail on every rank when it fails on one: a rank that leaves early (any `return` of the macros
// above) still enters the agreement all-reduce the healthy ranks run right before the first data collective, through this
// guard's destructor; the healthy path calls agree() itself.
struct CollectiveGuard {
    lrge_hip_comm *c; hipStream_t st; bool armed = false;
    ~CollectiveGuard() { if (armed && c) (void)comm_agree(c, LRGE_ERR_DEVICE, st); }
    int agree() { const bool was = armed; armed = false; return (was && c) ? comm_agree(c, LRGE_OK, st) : LRGE_OK; }
};

// The three exchanges of a sharded build (k_route.h).  On success so->x [, so->y] hold this rank's kept entries in the order
// the one index would hold them (so->n of them), *own_hashes / *n_own the hashes of the keys this rank owns.  Collective:
// a failure on one rank fails the call on every rank (status words ride in the small vectors; comm_agree before the
// exchanges that follow large allocations).
static int sharded_collect(lrge_hip_ctx *ctx, Scratch &sc, const Preset &P, int preset, bool pk, u32 pk_pos1, u32 pk_ybits,
                           const IndexBuildOpts *ro, SketchOut *so, u64 **own_hashes, u64 *n_own) {
    lrge_hip_comm *c = ro->comm;
    const int W = c->world, me = c->rank;
    lrge_hip_seqset *S = ro->restrict_to;
    const lrge_hip_seqset *Tsh = ro->shard;
    hipStream_t st = ctx->stream;
    g_shard_stats = ShardStats();
    int rc = LRGE_OK;
    // option VERBOSE: time this rank spent in each phase, the waits for the other ranks (local transport) taken out
    double t_mark = DevPool::now_ms(), w_mark = c->wait_ms;
    auto mark = [&](const char *what) {
        if (!ctx->opt("VERBOSE")) return;
        (void)hipStreamSynchronize(st);
        const double now = DevPool::now_ms();
        fprintf(stderr, "[lrge_hip] rank %d sharded build: %-28s %7.3f ms (+ %.3f ms waiting)\n", me, what, (now - t_mark) - (c->wait_ms - w_mark), c->wait_ms - w_mark);
        t_mark = now; w_mark = c->wait_ms;
    };
    // ---- (1) one agreed key-set size: all ranks' streamed base counts (and whether anybody has failed already) ----
    std::vector<u64> hv((size_t)W + 1, 0);
    u64 *d_sz = sc.get<u64>((size_t)W + 1);
    hv[(size_t)me] = S->total_bases; hv[(size_t)W] = d_sz ? 0 : 1;
    if (!d_sz) { rc = comm_agree(c, LRGE_ERR_DEVICE, st); return rc ? rc : LRGE_ERR_DEVICE; }
    HIPCHK(ctx, hipMemcpyAsync(d_sz, hv.data(), hv.size() * 8, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipStreamSynchronize(st));          // (hv is reused below)
    rc = comm_allreduce_sum(c, d_sz, hv.size(), 8, st); if (rc) return rc;
    HIPCHK(ctx, hipMemcpyAsync(hv.data(), d_sz, hv.size() * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    if (hv[(size_t)W]) { LRGE_SET_ERR(ctx, "sharded index build: another rank failed"); return LRGE_ERR_DEVICE; }
    u64 max_bases = 1;
    for (int r = 0; r < W; ++r) max_bases = std::max(max_bases, hv[(size_t)r]);
    const u64 bloom_bits = ctx->opt_u64("SHARD_BLOOM_BITS", 4);      // filter bits per streamed base (~3-4 minimizers per 16 bits)
    u64 n_words = 1ULL << 14;
    while (n_words < (1ULL << 31) && n_words * 64 < bloom_bits * max_bases) n_words <<= 1;
    g_shard_stats.keyset_bytes = n_words * 8;
    mark("sizes all-reduce");
    // ---- (2) local: the streamed set's sketch + this rank's key set on the side stream, beside the target shard's sketch ----
    KeySet ks{nullptr, n_words - 1, 0, (u32)(2 * P.k), ceil_log2_u64(n_words)};
    u64 *gathered = nullptr, *inter = nullptr;
    SketchOut raw;
    auto local1 = [&]() -> int {
        if (!S->presk || S->presk->preset != preset) {
            ctx->presk_pending = S; ctx->presk_preset = preset;
            int r = presketch_start_pending(ctx, ~0ULL >> 2); if (r) return r;
        }
        if (!S->presk) { LRGE_SET_ERR(ctx, "index_build_sharded: the streamed set is too large to restrict an index to (it is streamed in views)"); return LRGE_ERR_TOO_MANY; }
        ks.bits = sc.get<u64>(n_words); gathered = sc.get<u64>(n_words * (u64)W); inter = sc.get<u64>(n_words * (u64)(W <= 8 ? 8 : 16));
        if (!ks.bits || !gathered || !inter) return LRGE_ERR_DEVICE;
        HIPCHK(ctx, hipMemsetAsync(ks.bits, 0, n_words * 8, ctx->stream2));
        hipLaunchKernelGGL(k_keyset_build, dim3((u32)div_up(S->total_bases + 1, 256)), dim3(256), 0, ctx->stream2, S->presk->x, S->presk->d_total, ks);
        KCHK(ctx);
        HIPCHK(ctx, hipEventRecord(ctx->ev_join, ctx->stream2));
        // the shard's own sketch runs on the main stream meanwhile
        int r = sketch_device(ctx, sc, Tsh, preset, true, &raw, pk ? pk_pos1 : 0, pk_ybits, nullptr); if (r) return r;
        sc.drop(raw.mz_off);
        if (raw.n && ro->shard_first) {     // read index inside the shard -> index in the whole target set
            if (pk) hipLaunchKernelGGL(k_add_u64, dim3((u32)div_up(raw.n, 256)), dim3(256), 0, st, raw.x, raw.n, (u64)ro->shard_first << pk_pos1);
            else hipLaunchKernelGGL(k_add_u64, dim3((u32)div_up(raw.n, 256)), dim3(256), 0, st, raw.y, raw.n, (u64)ro->shard_first << 32);
            KCHK(ctx);
        }
        HIPCHK(ctx, hipStreamWaitEvent(st, ctx->ev_join, 0));
        return LRGE_OK;
    };
    rc = local1();
    mark("sketches + key set");
    rc = comm_agree(c, rc, st); if (rc) return rc;
    g_shard_stats.entries_sketched = raw.n;
    mark("agree");
    rc = comm_allgather(c, ks.bits, n_words * 8, gathered, st); if (rc) return rc;
    mark("key-set all-gather");
    if (W <= 8) hipLaunchKernelGGL(k_keyset_interleave<8>, dim3((u32)div_up(n_words, 256)), dim3(256), 0, st, gathered, n_words, (u32)W, inter);
    else hipLaunchKernelGGL(k_keyset_interleave<16>, dim3((u32)div_up(n_words, 256)), dim3(256), 0, st, gathered, n_words, (u32)W, inter);
    KCHK(ctx);
    // ---- (3) local: which ranks ask for every entry, who owns its hash; counts per destination ----
    const u64 Mr = raw.n;
    if (Mr >= (1ULL << 32)) { LRGE_SET_ERR(ctx, "sharded index build: this rank's target share yields %llu minimizers (limit 2^32)", (unsigned long long)Mr); }
    RouteArgs A; A.x = raw.x; A.y = pk ? nullptr : raw.y; A.n = Mr; A.kshift = pk ? pk_ybits : 0;
    A.ks = KeySetAll{inter, n_words - 1, (u32)W}; A.n_tiles = (u32)std::max<u64>(1, div_up(Mr, RF_TILE));
    u32 *flags = nullptr, *cnt = nullptr, *d_tot = nullptr;
    std::vector<u64> mine((size_t)2 * W + 1, 0), matrix(((size_t)2 * W + 1) * (size_t)W, 0);
    auto local2 = [&]() -> int {
        if (Mr >= (1ULL << 32)) return LRGE_ERR_TOO_MANY;
        flags = sc.get<u32>(Mr + 1); cnt = sc.get<u32>((u64)2 * W * A.n_tiles); d_tot = sc.get<u32>((size_t)2 * W);
        if (!flags || !cnt || !d_tot) return LRGE_ERR_DEVICE;
        if (W <= 8) hipLaunchKernelGGL(k_route_count<8>, dim3(A.n_tiles), dim3(RF_THREADS), 0, st, A, flags, cnt);
        else hipLaunchKernelGGL(k_route_count<16>, dim3(A.n_tiles), dim3(RF_THREADS), 0, st, A, flags, cnt);
        KCHK(ctx);
        hipLaunchKernelGGL(k_route_scan, dim3((u32)(2 * W)), dim3(1024), 0, st, cnt, A.n_tiles, d_tot);
        KCHK(ctx);
        std::vector<u32> tot((size_t)2 * W);
        HIPCHK(ctx, hipMemcpyAsync(tot.data(), d_tot, tot.size() * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
        for (int q = 0; q < 2 * W; ++q) mine[(size_t)q] = tot[(size_t)q];
        return LRGE_OK;
    };
    mine[(size_t)2 * W] = local2() ? 1 : 0;
    mark("interleave + route count");
    const int rc2 = mine[(size_t)2 * W] ? LRGE_ERR_DEVICE : LRGE_OK;
    // ---- (4) everybody learns every (source, destination) count (and whether a rank has failed) ----
    {
        u64 *d_mine = sc.get<u64>(mine.size()), *d_all = sc.get<u64>(matrix.size());
        rc = comm_agree(c, (d_mine && d_all) ? LRGE_OK : LRGE_ERR_DEVICE, st); if (rc) return rc;
        HIPCHK(ctx, hipMemcpyAsync(d_mine, mine.data(), mine.size() * 8, hipMemcpyHostToDevice, st));
        rc = comm_allgather(c, d_mine, mine.size() * 8, d_all, st); if (rc) return rc;
        HIPCHK(ctx, hipMemcpyAsync(matrix.data(), d_all, matrix.size() * 8, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
        sc.drop(d_mine); sc.drop(d_all);
    }
    mark("counts all-gather");
    const size_t row = (size_t)2 * W + 1;
    for (int r = 0; r < W; ++r) if (matrix[(size_t)r * row + 2 * W]) { if (!rc2) LRGE_SET_ERR(ctx, "sharded index build: rank %d failed", r); return LRGE_ERR_DEVICE; }
    // send / receive offsets (elements) of the two all-to-alls
    std::vector<u64> ks_off((size_t)W + 1, 0), kr_off((size_t)W + 1, 0), os_off((size_t)W + 1, 0), or_off((size_t)W + 1, 0);
    for (int d = 0; d < W; ++d) {
        ks_off[(size_t)d + 1] = ks_off[(size_t)d] + mine[(size_t)d];
        os_off[(size_t)d + 1] = os_off[(size_t)d] + mine[(size_t)W + d];
        kr_off[(size_t)d + 1] = kr_off[(size_t)d] + matrix[(size_t)d * row + (size_t)me];
        or_off[(size_t)d + 1] = or_off[(size_t)d] + matrix[(size_t)d * row + (size_t)W + (size_t)me];
    }
    const u64 n_ks = ks_off[(size_t)W], n_kr = kr_off[(size_t)W], n_os = os_off[(size_t)W], n_or = or_off[(size_t)W];
    g_shard_stats.entries_sent = n_ks - mine[(size_t)me]; g_shard_stats.entries_recv = n_kr - mine[(size_t)me];
    g_shard_stats.hashes_sent = n_os - mine[(size_t)W + me]; g_shard_stats.hashes_recv = n_or - mine[(size_t)W + me];
    // ---- (5) local: send buffers grouped by destination (order-preserving), receive buffers ----
    u64 *sx = nullptr, *sy = nullptr, *sh = nullptr, *rx = nullptr, *ry = nullptr, *rh = nullptr;
    u32 *sh32 = nullptr, *rh32 = nullptr;
    const bool narrow = 2 * P.k <= 32 && !ctx->opt("SHARD_WIDE_HASHES");     // k = 15: the hashes of the second exchange travel as 4 bytes
    auto local3 = [&]() -> int {
        if (n_kr >= (1ULL << 32)) { LRGE_SET_ERR(ctx, "index limited to < 2^32 minimizers (this rank would hold %llu)", (unsigned long long)n_kr); return LRGE_ERR_TOO_MANY; }
        sx = sc.get<u64>(n_ks + 1); rx = sc.get<u64>(n_kr + 1); rh = sc.get<u64>(n_or + 1);
        if (narrow) { sh32 = sc.get<u32>(n_os + 1); rh32 = sc.get<u32>(n_or + 1); } else sh = sc.get<u64>(n_os + 1);
        if (!pk) { sy = sc.get<u64>(n_ks + 1); ry = sc.get<u64>(n_kr + 1); }
        if (!sx || !rx || !rh || (narrow ? (!sh32 || !rh32) : !sh) || (!pk && (!sy || !ry))) return LRGE_ERR_DEVICE;
        RouteBases B;
        for (int d = 0; d < ROUTE_MAX_WORLD; ++d) { B.keep[d] = d < W ? ks_off[(size_t)d] : 0; B.own[d] = d < W ? os_off[(size_t)d] : 0; }
        if (Mr) { hipLaunchKernelGGL(k_route_write, dim3(A.n_tiles), dim3(RF_THREADS), 0, st, A, flags, cnt, B, sx, sy, sh, sh32); KCHK(ctx); }
        return LRGE_OK;
    };
    rc = local3();
    mark("route write");
    rc = comm_agree(c, rc, st); if (rc) return rc;
    mark("agree");
    // ---- (6) the exchanges ----
    rc = comm_alltoallv(c, sx, ks_off.data(), rx, kr_off.data(), 8, st); if (rc) return rc;
    if (!pk) { rc = comm_alltoallv(c, sy, ks_off.data(), ry, kr_off.data(), 8, st); if (rc) return rc; }
    if (narrow) {
        rc = comm_alltoallv(c, sh32, os_off.data(), rh32, or_off.data(), 4, st); if (rc) return rc;
        if (n_or) { hipLaunchKernelGGL(k_u32_to_u64, dim3((u32)div_up(n_or, 256)), dim3(256), 0, st, rh32, n_or, rh); KCHK(ctx); }
    } else { rc = comm_alltoallv(c, sh, os_off.data(), rh, or_off.data(), 8, st); if (rc) return rc; }
    HIPCHK(ctx, hipStreamSynchronize(st));        // (the offset vectors are locals; the local transport has synchronised already)
    mark("all-to-alls");
    sc.drop(raw.x); if (raw.y) sc.drop(raw.y);
    sc.drop(flags); sc.drop(cnt); sc.drop(d_tot); sc.drop(sx); if (sh) sc.drop(sh); if (sh32) sc.drop(sh32); if (rh32) sc.drop(rh32); if (sy) sc.drop(sy);
    sc.drop(ks.bits); sc.drop(gathered); sc.drop(inter); sc.drop(d_sz);
    so->x = rx; so->y = ry; so->mz_off = nullptr; so->n = n_kr;
    *own_hashes = rh; *n_own = n_or;
    const u64 ss[8] = {g_shard_stats.keyset_bytes, g_shard_stats.entries_sketched, g_shard_stats.entries_sent, g_shard_stats.entries_recv,
                       g_shard_stats.hashes_sent, g_shard_stats.hashes_recv, (u64)(pk ? 8 : 16) | (u64)(narrow ? 4 : 8) << 8, n_kr};
    memcpy(ctx->shard_stats, ss, sizeof ss);
    return LRGE_OK;
}


static int index_build_one(lrge_hip_ctx *ctx, const lrge_hip_seqset *targets, int preset, lrge_hip_index **out, const IndexBuildOpts *ro = nullptr) {
    *out = nullptr;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    ctx->pin_items.clear(); ctx->pin_used = 0;      // reads an earlier, failed call may have left queued
    ctx->resolve_timers();
    memset(ctx->ms, 0, sizeof(ctx->ms));
    memset(ctx->counters, 0, sizeof(ctx->counters));
    StageTimer t_total(ctx, LRGE_T_TOTAL);
    Scratch sc(ctx);
    Preset P = make_preset(preset);
    // test-only overrides of two chaining heuristics, so that parity tests can drive the rarely taken
    // paths (no max_skip break -> candidates beyond the register window; tight max_iter clamp)
    P.max_skip = (int)ctx->opt_u64("DEBUG_MAX_SKIP", (u64)P.max_skip);
    P.max_iter = (int)ctx->opt_u64("DEBUG_MAX_ITER", (u64)P.max_iter);
    // Index entries are packed into one u64 -- hash << ybits | rid << pos1 | (pos << 1 | strand) -- whenever that
    // fits (2k + bits(rid) + bits(pos) + 1 <= 64: ava-ont always in practice, ava-pb for small read sets): half the
    // bytes through the sort, the table build and the lookups, and 8 instead of 16 bytes per entry resident in HBM.
    const u32 pk_pos1 = std::max<u32>(1, ceil_log2_u64((u64)targets->max_len + 1)) + 1;
    const u32 pk_rid = std::max<u32>(1, ceil_log2_u64((u64)targets->n + 1));
    const bool pk = 2 * (u32)P.k + pk_rid + pk_pos1 <= 64 && !ctx->opt_u64("NO_PACKED_INDEX", 0);
    const u32 pk_ybits = pk ? pk_rid + pk_pos1 : 0;
    SketchOut so;
    int rc = LRGE_OK;
    KeySet ks{nullptr, 0, 0, 0, 0};
    const bool sharded = ro && ro->shard;
    CollectiveGuard cg{ro ? ro->comm : nullptr, ctx->stream};
    cg.armed = ro && ro->comm && !sharded;        // (a sharded build agrees inside sharded_collect first)
    if (ro && ro->restrict_to && !sharded) {
        // the streamed set's sketch and the key set built from it go to the side stream FIRST, so that they run beside
        // the target sketch below; the main stream meets them (ev_join) where the entries are filtered
        lrge_hip_seqset *S = ro->restrict_to;
        if (!S->presk || S->presk->preset != preset) {
            ctx->presk_pending = S; ctx->presk_preset = preset;
            rc = presketch_start_pending(ctx, ~0ULL >> 2);      // (the restricted build NEEDS the streamed set's minimizers)
            if (rc) return rc;
        }
        if (!S->presk) { LRGE_SET_ERR(ctx, "index_build_for: the streamed set is too large to restrict an index to (it is streamed in views)"); return LRGE_ERR_TOO_MANY; }
        // The entries are tested AFTER the first LSD pass of the index sort has grouped them by the top digit of the hash
        // (below), and a key's bit lives in the slice of the set that belongs to its top digit: a group's tests stay
        // inside 1/64 .. 1/256 of the set (k = 15: 2 MB of the 128 MB bitmap), i.e. in L2, instead of one random line
        // from the Infinity Cache per entry (measured at C4: 8 ms per sweep over 244 M entries without the grouping).
        const int passes_ = (2 * P.k + 7) / 8;
        ks.top_shift = 8u * (u32)(passes_ - 1);
        const u32 top_bits = (u32)(2 * P.k) - ks.top_shift;
        u64 n_words;
        if (2 * P.k <= 33) { ks.direct = 1; n_words = std::max<u64>(1, (1ULL << (2 * P.k)) >> 6); }
        else { n_words = 1ULL << 20; while (n_words < (1ULL << 31) && n_words * 64 < 8 * (S->total_bases + 1)) n_words <<= 1; }
        ks.word_mask = n_words - 1;
        ks.low_bits = ceil_log2_u64(n_words) - top_bits;
        ks.bits = sc.get<u64>(n_words);
        if (!ks.bits) return LRGE_ERR_DEVICE;
        HIPCHK(ctx, hipMemsetAsync(ks.bits, 0, n_words * 8, ctx->stream2));
        hipLaunchKernelGGL(k_keyset_build, dim3((u32)div_up(S->total_bases + 1, 256)), dim3(256), 0, ctx->stream2, S->presk->x, S->presk->d_total, ks);
        KCHK(ctx);
        HIPCHK(ctx, hipEventRecord(ctx->ev_join, ctx->stream2));
    }
    // a restricted build counts its 1/world share of the hash space (the rest comes through the communicator)
    u32 own_rank = 0, own_world = 1;
    if (ro && ro->restrict_to) {
        own_rank = ro->comm ? (u32)ro->comm->rank : 0; own_world = ro->comm ? (u32)ro->comm->world : 1;
        if (!ro->comm && ctx->opt("DEBUG_OWN_SHARE")) {
            // timing emulation of ONE rank of a world on a 1-GPU box ("world,rank"): this rank counts its share of the hash
            // space and nobody supplies the rest, so the statistics (mid_occ) are incomplete and the results invalid
            unsigned w_ = 1, r_ = 0;
            if (sscanf(ctx->opt("DEBUG_OWN_SHARE"), "%u,%u", &w_, &r_) == 2 && w_ >= 1 && r_ < w_) { own_world = w_; own_rank = r_; }
        }
    }
    bool fused = false; u64 *own_hashes = nullptr; u64 n_own = 0;
    struct PreparedGuard { lrge_hip_ctx *c; ~PreparedGuard() { presketch_drop_prepared(c); } } prepared_guard{ctx};   // (an error between the two steps)
    if (sharded) {
        // this rank sketches its own share of the targets; key sets, kept entries and owned hashes travel (k_route.h)
        StageTimer t(ctx, LRGE_T_INDEX_RESTRICT);
        rc = sharded_collect(ctx, sc, P, preset, pk, pk ? pk_pos1 : 0, pk_ybits, ro, &so, &own_hashes, &n_own);
        t.stop();
        if (rc) return rc;
        cg.armed = ro->comm != nullptr;
        fused = true;                              // (so holds exactly the entries this rank's index keeps)
    }
    // (measured at C4: with a world of 2 the key set is so dense that the sweeps of the general form are the faster way)
    if (!sharded && ro && ro->restrict_to && !ctx->opt("RESTRICT_SWEEPS") && (own_world >= 4 || ctx->opt("RESTRICT_FUSED"))) {
        // fast form: the key-set test inside the target sketch (needs the key set first: the main stream meets the side
        // stream here instead of after the sketch)
        rc = seqset_ready(ctx, targets);
        if (rc) return rc;
        HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
        StageTimer t(ctx, LRGE_T_SKETCH);
        rc = (preset == LRGE_PRESET_AVA_PB)
                 ? sketch_restrict_launch<19, 5, true>(ctx, sc, targets, pk, pk ? pk_pos1 : 0, pk_ybits, ks, own_rank, own_world, &so, &own_hashes, &n_own, &fused)
                 : sketch_restrict_launch<15, 5, false>(ctx, sc, targets, pk, pk ? pk_pos1 : 0, pk_ybits, ks, own_rank, own_world, &so, &own_hashes, &n_own, &fused);
        t.stop();
        if (rc) return rc;
    }
    if (!fused) {
        rc = sketch_device(ctx, sc, targets, preset, true, &so, pk ? pk_pos1 : 0, pk_ybits);
        if (rc) return rc;
        sc.drop(so.mz_off);
    }
    u64 M = so.n;
    if (M >= (1ULL << 32)) { LRGE_SET_ERR(ctx, "index limited to < 2^32 minimizers (got %llu)", (unsigned long long)M); return LRGE_ERR_TOO_MANY; }

    // ---- restricted build: keep the entries the streamed reads can ask for, count ALL keys for the statistics ----
    bool have_global = false; u64 g_distinct = 0, g_mz = 0; int g_mid_occ = 0;
    int pass_from = 0;      // LSD passes of the index sort already done
    if (ro && ro->restrict_to) {
        u64 *sh = own_hashes; u64 Ms = n_own;
        if (!fused) {   // general form: first pass of the index sort over ALL entries: groups them by the top digit of the hash (see the key set above)
            StageTimer t(ctx, LRGE_T_INDEX_SORT);
            ALLOC_OR_FAIL(k1, sc, u64, M + 1);
            if (pk) {
                u64 *rk;
                rc = radix_sort_keys(ctx, sc, so.x, k1, M, (int)pk_ybits, 2 * P.k, &rk, /*reverse_digits=*/true, 0, 1);
                if (rc) return rc;
                sc.drop(rk == so.x ? k1 : so.x);
                so.x = rk;
            } else {
                ALLOC_OR_FAIL(v1, sc, u64, M + 1);
                u64 *rk, *rv;
                rc = radix_sort_pairs(ctx, sc, so.x, so.y, k1, v1, M, 0, 2 * P.k, &rk, &rv, /*reverse_digits=*/true, nullptr, 0, 0, 1);
                if (rc) return rc;
                sc.drop(rk == so.x ? k1 : so.x); sc.drop(rv == so.y ? v1 : so.y);
                so.x = rk; so.y = rv;
            }
            pass_from = 1;
            t.stop();
        }
        StageTimer t(ctx, LRGE_T_INDEX_RESTRICT);
        if (!fused) {
        HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
        RestrictArgs A;
        A.x = so.x; A.y = pk ? nullptr : so.y; A.n = M; A.kshift = pk ? pk_ybits : 0; A.ks = ks;
        A.rank = own_rank; A.world = own_world;
        const u32 nb = (u32)div_up(M, RF_TILE);
        ALLOC_OR_FAIL(bc_keep, sc, u32, (size_t)nb + 1); ALLOC_OR_FAIL(bc_own, sc, u32, (size_t)nb + 1); ALLOC_OR_FAIL(d_tot, sc, u32, 2);
        ALLOC_OR_FAIL(d_flags, sc, u32, (size_t)nb * RF_THREADS + 1);
        u32 tot[2] = {0, 0};
        if (nb) {
            hipLaunchKernelGGL(k_restrict_count, dim3(nb), dim3(RF_THREADS), 0, ctx->stream, A, bc_keep, bc_own, d_flags);
            KCHK(ctx);
            rc = scan_exclusive_u32(ctx, sc, bc_keep, bc_keep, nb, d_tot); if (rc) return rc;
            rc = scan_exclusive_u32(ctx, sc, bc_own, bc_own, nb, d_tot + 1); if (rc) return rc;
            HIPCHK(ctx, ctx->d2h(tot, d_tot, 8, ctx->stream));
            HIPCHK(ctx, ctx->d2h_sync(ctx->stream));
        }
        const u64 Mk = tot[0];
        Ms = tot[1];
        ALLOC_OR_FAIL(kx, sc, u64, Mk + 1);
        u64 *ky = nullptr;
        if (!pk) { ky = sc.get<u64>(Mk + 1); if (!ky) return LRGE_ERR_DEVICE; }
        sh = sc.get<u64>(Ms + 1);
        if (!sh) return LRGE_ERR_DEVICE;
        if (nb) {
            hipLaunchKernelGGL(k_restrict_write, dim3(nb), dim3(RF_THREADS), 0, ctx->stream, A, bc_keep, bc_own, d_flags, kx, ky, sh);
            KCHK(ctx);
        }
        sc.drop(so.x); if (so.y) sc.drop(so.y);
        sc.drop(bc_keep); sc.drop(bc_own); sc.drop(d_tot); sc.drop(d_flags);
        so.x = kx; so.y = ky; M = Mk;
        }
        // occurrence statistics of the owned share of the hash space
        const u32 max_bin_ = (u32)P.max_mid_occ + 1;
        ALLOC_OR_FAIL(sh2, sc, u64, Ms + 1);
        u64 *rs_ = nullptr;
        rc = radix_sort_keys(ctx, sc, sh, sh2, Ms, 0, 2 * P.k, &rs_, /*reverse_digits=*/true, pass_from, -1); if (rc) return rc;   // (they arrive grouped by the top digit too)
        ALLOC_OR_FAIL(starts, sc, u32, Ms + 2); ALLOC_OR_FAIL(d_nr, sc, u32, 1);
        rc = compact_heads_async(ctx, sc, rs_, Ms, 0, starts, d_nr); if (rc) return rc;
        ALLOC_OR_FAIL(d_hist, sc, u32, (size_t)max_bin_ + 2);
        HIPCHK(ctx, hipMemsetAsync(d_hist, 0, ((size_t)max_bin_ + 2) * 4, ctx->stream));
        if (Ms) {
            hipLaunchKernelGGL(k_occ_hist_runs, dim3((u32)std::min<u64>(div_up(Ms, 256), (u64)ctx->n_cu * 8)), dim3(256), 0, ctx->stream, starts, d_nr, Ms, d_hist, max_bin_);
            KCHK(ctx);
        }
        const u32 head = std::min<u32>(4096, max_bin_ + 1);
        ALLOC_OR_FAIL(d_vec, sc, u64, (size_t)head + 2);
        hipLaunchKernelGGL(k_stats_pack, dim3((u32)div_up(head, 256)), dim3(256), 0, ctx->stream, d_nr, Ms, d_hist, head, d_vec);
        KCHK(ctx);
        rc = cg.agree(); if (rc) return rc;       // every rank got this far, or none goes on
        if (ro->comm) { rc = comm_allreduce_sum(ro->comm, d_vec, (size_t)head + 2, 8, ctx->stream); if (rc) return rc; }
        std::vector<u64> hv((size_t)head + 2);
        HIPCHK(ctx, hipMemcpyAsync(hv.data(), d_vec, hv.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        g_distinct = hv[0]; g_mz = hv[1];
        // mm_idx_cal_max_occ + mm_mapopt_update clamps over the distinct keys of the whole target set (same arithmetic as below)
        int thres = INT32_MAX;
        if (g_distinct) {
            const u64 kth = (u64)((1. - (double)P.mid_occ_frac) * (double)g_distinct);
            u64 cum = 0; u32 v = max_bin_; bool found = false;
            for (u32 b = 0; b < head; ++b) { cum += hv[2 + b]; if (cum > kth) { v = b; found = true; break; } }
            if (!found && head < max_bin_ + 1) {      // the k-th count lies beyond the head bins: the whole histogram travels
                ALLOC_OR_FAIL(d_full, sc, u64, (size_t)max_bin_ + 1);
                hipLaunchKernelGGL(k_u32_to_u64, dim3((u32)div_up((u64)max_bin_ + 1, 256)), dim3(256), 0, ctx->stream, d_hist, (u64)max_bin_ + 1, d_full);
                KCHK(ctx);
                if (ro->comm) { rc = comm_allreduce_sum(ro->comm, d_full, (size_t)max_bin_ + 1, 8, ctx->stream); if (rc) return rc; }
                std::vector<u64> full((size_t)max_bin_ + 1);
                HIPCHK(ctx, hipMemcpyAsync(full.data(), d_full, full.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
                HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
                cum = 0;
                for (u32 b = 0; b <= max_bin_; ++b) { cum += full[b]; if (cum > kth) { v = b; break; } }
                sc.drop(d_full);
            }
            thres = (int)v + 1;
        }
        if (thres < P.min_mid_occ) thres = P.min_mid_occ;
        if (P.max_mid_occ > P.min_mid_occ && thres > P.max_mid_occ) thres = P.max_mid_occ;
        g_mid_occ = thres; have_global = true;
        sc.drop(sh); sc.drop(sh2); sc.drop(starts); sc.drop(d_nr); sc.drop(d_hist); sc.drop(d_vec); sc.drop(ks.bits);
        t.stop();
    }

    u64 *skey = so.x, *spos = so.y;
    bool seg_packed = false; u32 kshift_t = pk_ybits; u32 *d_seg_start = nullptr; std::vector<u32> h_seg_start;
    {
        StageTimer t(ctx, LRGE_T_INDEX_SORT);
        ALLOC_OR_FAIL(k1, sc, u64, M + 1);
        if (pk) {
            u64 *rk = so.x;
            bool hybrid = false;
            // two most-significant-digit passes, then the rest inside LDS (k_prims.h: index_sort_hybrid) where the entries suit it
            if (pass_from == 0) { rc = index_sort_hybrid(ctx, sc, so.x, k1, M, (int)pk_ybits, 2 * P.k, &rk, &hybrid); if (rc) return rc; }
            if (!hybrid) rc = radix_sort_keys(ctx, sc, so.x, k1, M, (int)pk_ybits, 2 * P.k, &rk, /*reverse_digits=*/true, pass_from, -1);   // see k_index.h
            if (rc) return rc;
            skey = rk; spos = rk;
            sc.drop(rk == so.x ? k1 : so.x);
        } else {
            ALLOC_OR_FAIL(v1, sc, u64, M + 1);
            // the pair layout, segment-packed (k_prims.h: index_sort_segpacked): behind the first digit the low hash byte is implied
            // and the rest of the entry fits one word -- fewer bytes through the remaining passes, 8 bytes per entry resident
            const u32 yb_p = pk_rid + pk_pos1;
            if (pass_from == 0 && 2 * (u32)P.k - 8 + yb_p <= 64 && 2 * P.k > 16 && !ctx->opt("NO_SEG_PACK") && M >= ctx->opt_u64("SEG_PACK_MIN", 1ULL << 22)) {
                u64 *rk = nullptr;
                rc = index_sort_segpacked(ctx, sc, so.x, so.y, k1, v1, M, 2 * P.k, yb_p, pk_pos1, &rk, &d_seg_start, &h_seg_start);
                if (rc) return rc;
                seg_packed = true; kshift_t = yb_p;
                skey = rk; spos = rk;
                sc.drop(rk == so.x ? so.y : so.x); sc.drop(k1); sc.drop(v1);
            } else {
            u64 *rk, *rv;
            rc = radix_sort_pairs(ctx, sc, so.x, so.y, k1, v1, M, 0, 2 * P.k, &rk, &rv, /*reverse_digits=*/true, nullptr, 0, pass_from, -1);   // see k_index.h
            if (rc) return rc;
            // (no sync: everything runs in order on ctx->stream; scratch is recycled in stream order)
            skey = rk; spos = rv;
            sc.drop(rk == so.x ? k1 : so.x);
            sc.drop(rv == so.y ? v1 : so.y);
            }
        }
        t.stop();
    }
    if (!fused && !ctx->opt("NO_PRESKETCH")) {
        // The streamed set's sketch goes to the side stream here, beside the table build (its memory is taken here too: the arena
        // recycles in main-stream order).  It is VALU-bound at the full issue rate, so it hides little wherever it runs -- beside
        // the first sort passes (rounds 2-3) those went from 0.43 + 0.86 to 1.23 + 2.17 ms, beside the run-head and placement
        // passes these go from 3.2 to 5.5 ms: ~0.7 of its 2.9 ms either way (C4) -- but here the host never has to wait for the
        // set's upload job with nothing queued behind it.
        rc = presketch_start_pending(ctx, targets->total_bases);
        if (rc) return rc;
    }
    const bool pk_t = pk || seg_packed;          // what the table build and the lookups see: one packed word per entry

    lrge_hip_index *ix = new lrge_hip_index();
    IndexGuard ix_guard(ix);
    ix->ctx = ctx; ix->seqs = targets; ix->preset_id = preset; ix->P = P; ix->n_mz = M; ix->n_entries = M;
    u32 n_runs = 0;
    const u32 max_bin = (u32)P.max_mid_occ + 1;
    std::vector<u32> occ;
    {
        StageTimer t(ctx, LRGE_T_INDEX_TABLE);
        const u32 ht_fix = ctx->opt("HT_NO_FIX") ? 0u : ht_fix_with_power(P.k, (u32)ctx->opt_u64("HT_POWER", 3));   // (HT_POWER: exponent of the distribution correction, 0 = linear stretch only; measured 2-4 alike, mean displacement 0.30 slots at 3)     // (option HT_NO_FIX: the clustered homes of rounds 1-2, for A/B runs)
        u32 *d_runstart = nullptr;
        if (M) {
            rc = compact_heads(ctx, sc, skey, M, kshift_t, &d_runstart, &n_runs, d_seg_start, seg_packed ? 256u : 0u);    // runs of equal hash
            if (rc) return rc;
        }
#ifndef HT_CAP_NUM
#define HT_CAP_NUM 2       // home slots per distinct key = HT_CAP_NUM / HT_CAP_DEN
#define HT_CAP_DEN 1
#endif
        // a part of a partitioned index (a target set of tens of gigabases) gets 1.25 instead of 2 slots per key: the
        // tables of all parts are resident together and memory, not probe length (+15 % lookup time), is what binds there
        u64 cap = targets->is_view ? (u64)n_runs * 5 / 4 : (u64)n_runs * HT_CAP_NUM / HT_CAP_DEN;
        if (cap < 1024) cap = 1024;
        if (cap + n_runs >= (1ULL << 32)) { LRGE_SET_ERR(ctx, "index limited to < 2^32/3 distinct minimizers (got %u)", n_runs); return LRGE_ERR_TOO_MANY; }
        ix->ht_cap = cap; ix->ht_fix = ht_fix;
        ix->n_keys = n_runs;
        u32 *d_occ = sc.get<u32>((size_t)max_bin + 5);     // [max_bin + 1] = overflow flag, then (8-byte aligned) the u64 sum of displacements
        if (!d_occ) return LRGE_ERR_DEVICE;
        u64 *ht = nullptr;
        occ.assign((size_t)max_bin + 1, 0);
        const size_t head_bins = std::min<size_t>(4096, (size_t)max_bin + 1);
        // slack behind cap: displaced keys at the very end of the table do not wrap.  n_runs / 16 is far more
        // than linear probing at load 1/2 ever needs; if it were not, the second attempt (n_runs + 1) always fits.
        for (int attempt = 0; attempt < 2; ++attempt) {
            const u64 slack = attempt == 0 ? std::max<u64>((u64)n_runs / 16, 4096) : (u64)n_runs + 1;
            const u64 n_slots = cap + slack;
            ix->ht_slots = n_slots;
            ht = sc.get<u64>(2 * n_slots);
            if (!ht) return LRGE_ERR_DEVICE;
            // the placement kernel writes every slot itself (entries and empty ones) unless told otherwise (option HT_MEMSET: clear
            // first, then 16-byte entry stores -- the form of rounds 1-2, for A/B runs)
            const bool fused_fill = n_runs != 0 && !ctx->opt("HT_MEMSET");
            if (!fused_fill) HIPCHK(ctx, hipMemsetAsync(ht, 0xFF, 2 * n_slots * 8, ctx->stream));   // key = HT_EMPTY
            HIPCHK(ctx, hipMemsetAsync(d_occ, 0, ((size_t)max_bin + 5) * 4, ctx->stream));
            if (n_runs) {
                const u32 n_tiles = (u32)div_up(n_runs, PLACE_TILE);
                u32 *bmax = sc.get<u32>((size_t)n_tiles + 1);
                if (!bmax) return LRGE_ERR_DEVICE;
                hipLaunchKernelGGL(k_place_reduce, dim3(n_tiles), dim3(PLACE_THREADS), 0, ctx->stream, skey, d_runstart, n_runs, cap, bmax, kshift_t, ht_fix, (const u32 *)d_seg_start);
                KCHK(ctx);
                hipLaunchKernelGGL(k_place_scan, dim3(1), dim3(1024), 0, ctx->stream, bmax, n_tiles);
                KCHK(ctx);
                hipLaunchKernelGGL(k_place_apply, dim3(std::min<u32>(n_tiles, (u32)ctx->n_cu * 8)), dim3(PLACE_THREADS), 0, ctx->stream,
                                   skey, d_runstart, n_runs, M, cap, n_slots, bmax, ht, d_occ, max_bin, d_occ + max_bin + 1, kshift_t, ht_fix,
                                   fused_fill ? bmax + n_tiles : (u32 *)nullptr, pk_t ? (const u64 *)nullptr : (const u64 *)spos, pk_t ? pk_pos1 : 0u,
                                   ctx->opt("NO_INLINE_SINGLETONS") ? 0u : 1u, (const u32 *)d_seg_start);
                KCHK(ctx);
                if (fused_fill) {
                    hipLaunchKernelGGL(k_fill_tail, dim3((u32)std::min<u64>(div_up(n_slots - cap / 2, 256), (u64)ctx->n_cu * 8)), dim3(256), 0, ctx->stream, ht, n_slots, bmax + n_tiles);
                    KCHK(ctx);
                }
                sc.drop(bmax);
            }
            // the k-th smallest occurrence count almost always sits in the first few bins: fetch 16 KB of
            // the histogram first, the whole 4 MB only if the prefix does not reach the k-th element
            u32 overflow = 0; u64 disp_sum = 0;
            HIPCHK(ctx, ctx->d2h(occ.data(), d_occ, head_bins * 4, ctx->stream));
            HIPCHK(ctx, ctx->d2h(&overflow, d_occ + max_bin + 1, 4, ctx->stream));
            HIPCHK(ctx, ctx->d2h(&disp_sum, d_occ + max_bin + 2 + ((max_bin + 2) & 1), 8, ctx->stream));
            HIPCHK(ctx, ctx->d2h_sync(ctx->stream));
            ctx->counters[LRGE_C_TABLE_DISP_SUM] = disp_sum;
            if (!overflow) break;
            sc.drop(ht); ht = nullptr;
            if (attempt == 1) { LRGE_SET_ERR(ctx, "index table placement overflowed%s", ""); return LRGE_ERR_DEVICE; }
        }
        {
            const u32 kth = n_runs ? (u32)((1. - (double)P.mid_occ_frac) * (double)n_runs) : 0;
            u64 cum = 0;
            for (size_t b = 0; b < head_bins; ++b) cum += occ[b];
            if (n_runs && cum <= kth) {
                HIPCHK(ctx, hipMemcpyAsync(occ.data(), d_occ, ((size_t)max_bin + 1) * 4, hipMemcpyDeviceToHost, ctx->stream));
                HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
            }
        }
        sc.drop(d_occ);
        if (d_runstart) sc.drop(d_runstart);
        ix->d_ht = ht; sc.keep(ht);
        t.stop();
    }
    // mm_idx_cal_max_occ + mm_mapopt_update clamps (mm2:index.c, mm2:options.c; aligner.rs:189)
    {
        int thres;
        if (n_runs == 0) thres = INT32_MAX;
        else {
            u32 kth = (u32)((1. - (double)P.mid_occ_frac) * (double)n_runs);
            u64 cum = 0; u32 v = max_bin;
            for (u32 b = 0; b <= max_bin; ++b) { cum += occ[b]; if (cum > kth) { v = b; break; } }
            thres = (int)v + 1;
        }
        if (thres < P.min_mid_occ) thres = P.min_mid_occ;
        if (P.max_mid_occ > P.min_mid_occ && thres > P.max_mid_occ) thres = P.max_mid_occ;
        ix->mid_occ = thres;
    }
    if (have_global) {      // restricted build: what mm_idx_stat / mm_idx_cal_max_occ report for the whole target set
        ix->mid_occ = g_mid_occ; ix->n_keys = g_distinct; ix->n_mz = g_mz;
        ix->restrict_set = ro->restrict_to; ix->restrict_uid = ro->restrict_to->uid;
    }
    // the sorted hashes of the (hash, y) pair layout are only read again by index_dump (tests); a part of a partitioned index
    // cannot be dumped and is short of memory, so it gives them back (8 of its 16 bytes per minimizer)
    ix->d_pos = spos; sc.keep(spos);
    if (skey != spos && targets->is_view) { ix->d_skey = nullptr; }          // stays with `sc`: released at scope exit
    else { ix->d_skey = skey; if (skey != spos) sc.keep(skey); }
    ix->pk_pos1 = pk_t ? pk_pos1 : 0; ix->pk_ybits = kshift_t;
    if (seg_packed) { ix->h_seg_start = h_seg_start; sc.drop(d_seg_start); }     // (the device copy served the table build; the dump needs the host copy)
    t_total.stop();
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->resolve_timers();
    pool_report(ctx, "index_build_one");
    *out = ix_guard.release();
    return LRGE_OK;
}

// Reads [r0, r1) of `s` as a set of its own: the packed image, the masks and the per-read arrays are shared (word offsets
// are absolute), only the sketch chunk map is rebuilt so that chunk ids start at 0.
static int seqset_view(lrge_hip_ctx *ctx, const lrge_hip_seqset *s, u32 r0, u32 r1, lrge_hip_seqset **out) {
    int rrc = seqset_ready(ctx, s);
    if (rrc) return rrc;
    lrge_hip_seqset *v = new lrge_hip_seqset();
    v->ctx = ctx; v->is_view = true; v->n = r1 - r0; v->parent = s->parent ? s->parent : s;
    v->uid = g_seqset_uid.fetch_add(1); v->parent_uid = s->parent ? s->parent_uid : s->uid;
    v->has_rank = s->has_rank; v->dup_rank = s->dup_rank;
    v->d_pack = s->d_pack; v->d_nmask = s->d_nmask; v->d_woff = s->d_woff + r0; v->d_len = s->d_len + r0;
    v->d_rank = s->d_rank ? s->d_rank + r0 : nullptr;
    v->h_woff.assign(s->h_woff.begin() + r0, s->h_woff.begin() + r1 + 1);
    v->h_len.assign(s->h_len.begin() + r0, s->h_len.begin() + r1);
    if (v->h_len.empty()) v->h_len.push_back(0);
    if (s->has_rank) {
        v->h_rank.assign(s->h_rank.begin() + r0, s->h_rank.begin() + r1);
    }
    v->h_cs.resize((size_t)v->n + 1);
    for (u32 i = 0; i <= v->n; ++i) v->h_cs[i] = s->h_cs[r0 + i] - s->h_cs[r0];
    v->n_chunks = v->h_cs[v->n];
    for (u32 i = r0; i < r1; ++i) {
        v->total_bases += s->h_len[i];
        if (s->h_len[i] > v->max_len) v->max_len = s->h_len[i];
        if (s->h_len[i] == 0) v->has_empty = true;
    }
    v->n_words = s->h_woff[r1] - s->h_woff[r0];
    hipError_t e = hipMalloc((void **)&v->d_cs, ((size_t)v->n + 1) * 4);
    if (e == hipSuccess) e = hipMemcpy(v->d_cs, v->h_cs.data(), ((size_t)v->n + 1) * 4, hipMemcpyHostToDevice);
    if (e != hipSuccess) { LRGE_SET_ERR(ctx, "seqset view: %s", hipGetErrorString(e)); if (v->d_cs) (void)hipFree(v->d_cs); delete v; return LRGE_ERR_DEVICE; }
    *out = v;
    return LRGE_OK;
}

// mm_idx_reader_read with batch_size = max (aligner.rs:112-122) makes ONE index whatever the size of the target file.
// Here a target set above LRGE_HIP_PART_BASES bases (default 4e9: the 2^32-entry limits of one part) is indexed in parts
// over views of the set; the occurrence statistics are then taken over all parts together (k_part_global_occ), mid_occ
// from that global histogram, and a key that is too frequent globally is marked so in every part (k_part_drop) -- the
// parts answer every lookup exactly as the one index would.
extern "C" int lrge_hip_index_build(lrge_hip_ctx *ctx, const lrge_hip_seqset *targets, int preset, lrge_hip_index **out) {
    if (!ctx || !targets || !out) return LRGE_ERR_INVALID;
    if (preset != LRGE_PRESET_AVA_ONT && preset != LRGE_PRESET_AVA_PB) { LRGE_SET_ERR(ctx, "Preset not found: %d", preset); return LRGE_ERR_INVALID; }
    *out = nullptr;
    const u64 part_bases = ctx->opt_u64("PART_BASES", 4000000000ull);
    if (targets->total_bases <= part_bases || targets->n < 2 || targets->is_view) return index_build_one(ctx, targets, preset, out);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // cut by reads, every part at most part_bases bases (a single longer read gets a part of its own)
    std::vector<u32> cuts{0};
    u64 acc = 0;
    for (u32 r = 0; r < targets->n; ++r) {
        if (acc && acc + targets->h_len[r] > part_bases) { cuts.push_back(r); acc = 0; }
        acc += targets->h_len[r];
    }
    cuts.push_back(targets->n);
    const int np = (int)cuts.size() - 1;
    if (np > MAX_INDEX_PARTS) { LRGE_SET_ERR(ctx, "target set needs %d index parts (limit %d)", np, MAX_INDEX_PARTS); return LRGE_ERR_TOO_MANY; }
    lrge_hip_index *top = new lrge_hip_index();
    IndexGuard top_guard(top);
    top->ctx = ctx; top->seqs = targets; top->preset_id = preset;
    float ms_acc[LRGE_T_N]; u64 cn_acc[LRGE_C_N];
    memset(ms_acc, 0, sizeof ms_acc); memset(cn_acc, 0, sizeof cn_acc);
    for (int p = 0; p < np; ++p) {
        lrge_hip_seqset *v = nullptr;
        int rc = seqset_view(ctx, targets, cuts[p], cuts[p + 1], &v);
        if (rc) return rc;
        top->part_sets.push_back(v); top->part_r0.push_back(cuts[p]);
        lrge_hip_index *ixp = nullptr;
        rc = index_build_one(ctx, v, preset, &ixp);
        if (rc) return rc;
        top->parts.push_back(ixp);
        top->n_mz += ixp->n_mz;
        for (int i = 0; i < LRGE_T_N; ++i) ms_acc[i] += ctx->ms[i];
        for (int i = 0; i < LRGE_C_N; ++i) cn_acc[i] += ctx->counters[i];
    }
    top->P = top->parts[0]->P;
    // ---- global occurrence statistics ----
    ctx->resolve_timers();
    memset(ctx->ms, 0, sizeof(ctx->ms));
    StageTimer t_glob(ctx, LRGE_T_INDEX_TABLE);
    const Preset &P = top->P;
    const u32 max_bin = (u32)P.max_mid_occ + 1;
    Scratch sc(ctx);
    ALLOC_OR_FAIL(d_hist, sc, u32, (size_t)max_bin + 1);
    unsigned long long *d_nd = (unsigned long long *)sc.get<u64>(1);
    if (!d_nd) return LRGE_ERR_DEVICE;
    HIPCHK(ctx, hipMemsetAsync(d_hist, 0, ((size_t)max_bin + 1) * 4, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(d_nd, 0, 8, ctx->stream));
    PartTables T; T.n = np; T.fix = top->parts[0]->ht_fix;
    for (int p = 0; p < np; ++p) { T.ht[p] = top->parts[p]->d_ht; T.cap[p] = top->parts[p]->ht_cap; }
    // every slot's global count stays resident between the two sweeps (4 bytes per slot) unless memory is short
    std::vector<u32 *> gsum((size_t)np, nullptr);
    if (!ctx->opt("PART_NO_GSUM")) {
        for (int p = 0; p < np; ++p) {
            gsum[(size_t)p] = sc.get<u32>(top->parts[p]->ht_slots);
            if (!gsum[(size_t)p]) { (void)hipGetLastError(); for (int q = 0; q < p; ++q) { sc.drop(gsum[(size_t)q]); gsum[(size_t)q] = nullptr; } ctx->err.clear(); break; }
        }
    }
    const bool have_gsum = np > 0 && gsum[(size_t)np - 1] != nullptr;
    for (int p = 0; p < np; ++p) {
        const u64 ns = top->parts[p]->ht_slots;
        hipLaunchKernelGGL(k_part_global_occ, dim3((u32)std::min<u64>(div_up(ns, 256), (u64)ctx->n_cu * 16)), dim3(256), 0, ctx->stream, top->parts[p]->d_ht, ns, T, p,
                           d_hist, max_bin, d_nd, have_gsum ? gsum[(size_t)p] : (u32 *)nullptr);
        KCHK(ctx);
    }
    std::vector<u32> occ((size_t)max_bin + 1);
    unsigned long long n_distinct = 0;
    HIPCHK(ctx, hipMemcpyAsync(occ.data(), d_hist, ((size_t)max_bin + 1) * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(&n_distinct, d_nd, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    top->n_keys = n_distinct;
    {   // mm_idx_cal_max_occ + mm_mapopt_update clamps, over the distinct keys of all parts (same arithmetic as index_build_one)
        int thres;
        if (n_distinct == 0) thres = INT32_MAX;
        else {
            const u64 kth = (u64)((1. - (double)P.mid_occ_frac) * (double)n_distinct);
            u64 cum = 0; u32 v = max_bin;
            for (u32 b = 0; b <= max_bin; ++b) { cum += occ[b]; if (cum > kth) { v = b; break; } }
            thres = (int)v + 1;
        }
        if (thres < P.min_mid_occ) thres = P.min_mid_occ;
        if (P.max_mid_occ > P.min_mid_occ && thres > P.max_mid_occ) thres = P.max_mid_occ;
        top->mid_occ = thres;
    }
    for (int p = 0; p < np; ++p) {
        const u64 ns = top->parts[p]->ht_slots;
        hipLaunchKernelGGL(k_part_drop, dim3((u32)div_up(ns, 256)), dim3(256), 0, ctx->stream, top->parts[p]->d_ht, ns, T, p, (u32)top->mid_occ,
                           have_gsum ? (const u32 *)gsum[(size_t)p] : (const u32 *)nullptr);
        KCHK(ctx);
        top->parts[p]->mid_occ = top->mid_occ;
    }
    t_glob.stop();
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->resolve_timers();
    ms_acc[LRGE_T_INDEX_TABLE] += ctx->ms[LRGE_T_INDEX_TABLE]; ms_acc[LRGE_T_TOTAL] += ctx->ms[LRGE_T_INDEX_TABLE];
    memcpy(ctx->ms, ms_acc, sizeof ms_acc); memcpy(ctx->counters, cn_acc, sizeof cn_acc);
    *out = top_guard.release();
    return LRGE_OK;
}

extern "C" int lrge_hip_index_build_for(lrge_hip_ctx *ctx, const lrge_hip_seqset *targets, int preset, lrge_hip_seqset *streamed,
                                        lrge_hip_comm *comm, lrge_hip_index **out) {
    if (!ctx || !targets || !out) return LRGE_ERR_INVALID;
    if (!streamed && !comm) return lrge_hip_index_build(ctx, targets, preset, out);
    if (preset != LRGE_PRESET_AVA_ONT && preset != LRGE_PRESET_AVA_PB) { LRGE_SET_ERR(ctx, "Preset not found: %d", preset); return LRGE_ERR_INVALID; }
    *out = nullptr;
    if (!streamed) { LRGE_SET_ERR(ctx, "index_build_for: a communicator needs the streamed set of this rank"); return LRGE_ERR_INVALID; }
    if (streamed->ctx != ctx || targets->ctx != ctx || (comm && comm->ctx != ctx)) { LRGE_SET_ERR(ctx, "index_build_for: sets / communicator belong to another context"); return LRGE_ERR_INVALID; }
    if (targets->total_bases > ctx->opt_u64("PART_BASES", 4000000000ull)) {
        LRGE_SET_ERR(ctx, "index_build_for: target sets above PART_BASES bases (a partitioned index) are not implemented for restricted builds");
        return LRGE_ERR_TOO_MANY;
    }
    IndexBuildOpts ro; ro.restrict_to = streamed; ro.comm = comm;
    return index_build_one(ctx, targets, preset, out, &ro);
}

// A read set known by its lengths and names only (its bases live elsewhere: on the other ranks of a sharded build).
// It can stand where an index's target set is consulted for lengths and name ranks; it cannot be sketched.
static int seqset_describe(lrge_hip_ctx *ctx, const uint32_t *lens, uint32_t n, const uint32_t *name_rank, lrge_hip_seqset **out) {
    *out = nullptr;
    std::unique_ptr<lrge_hip_seqset, void (*)(lrge_hip_seqset *)> guard(new lrge_hip_seqset(), lrge_hip_seqset_free);
    lrge_hip_seqset *s = guard.get();
    s->ctx = ctx; s->n = n; s->pooled = true; s->uid = g_seqset_uid.fetch_add(1);
    s->h_len.assign(lens, lens + n);
    if (s->h_len.empty()) s->h_len.push_back(0);
    for (u32 i = 0; i < n; ++i) {
        if (lens[i] >= (1u << 31)) { LRGE_SET_ERR(ctx, "read %u: length >= 2^31", i); return LRGE_ERR_INVALID; }
        s->total_bases += lens[i]; s->max_len = std::max(s->max_len, lens[i]); s->has_empty |= lens[i] == 0;
    }
    if (name_rank) { s->has_rank = true; s->h_rank.assign(name_rank, name_rank + n); s->dup_rank = ranks_have_duplicate(s->h_rank); }
    hipError_t e = hipSuccess;
    const size_t nb = (((size_t)(n ? n : 1) * 4) + 255) & ~(size_t)255;
    s->d_meta = ctx->pool.alloc(2 * nb, &e);
    if (!s->d_meta) { LRGE_SET_ERR(ctx, "seqset_describe: device allocation failed: %s", hipGetErrorString(e)); return LRGE_ERR_DEVICE; }
    s->d_len = (u32 *)s->d_meta; s->d_rank = (u32 *)((char *)s->d_meta + nb);
    if (n) {
        HIPCHK(ctx, hipMemcpyAsync(s->d_len, s->h_len.data(), (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
        if (name_rank) HIPCHK(ctx, hipMemcpyAsync(s->d_rank, s->h_rank.data(), (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    *out = guard.release();
    return LRGE_OK;
}

extern "C" int lrge_hip_index_build_sharded(lrge_hip_ctx *ctx, const uint32_t *all_target_lens, const uint32_t *all_target_ranks, uint32_t n_targets,
                                            const lrge_hip_seqset *target_shard, uint32_t shard_first, int preset, lrge_hip_seqset *streamed,
                                            lrge_hip_comm *comm, lrge_hip_index **out) {
    if (!ctx || !out || !all_target_lens || !target_shard || !streamed || !comm) return LRGE_ERR_INVALID;
    *out = nullptr;
    // (argument errors below are rank-local by nature -- every rank passes the same job -- so they return before any collective)
    if (preset != LRGE_PRESET_AVA_ONT && preset != LRGE_PRESET_AVA_PB) { LRGE_SET_ERR(ctx, "Preset not found: %d", preset); return LRGE_ERR_INVALID; }
    if (streamed->ctx != ctx || target_shard->ctx != ctx || comm->ctx != ctx) { LRGE_SET_ERR(ctx, "index_build_sharded: sets / communicator belong to another context"); return LRGE_ERR_INVALID; }
    if (comm->world > ROUTE_MAX_WORLD) { LRGE_SET_ERR(ctx, "index_build_sharded: at most %d ranks", ROUTE_MAX_WORLD); return LRGE_ERR_INVALID; }
    if ((u64)shard_first + target_shard->n > n_targets) { LRGE_SET_ERR(ctx, "index_build_sharded: the shard [%u, %u) lies outside the %u target reads", shard_first, shard_first + target_shard->n, n_targets); return LRGE_ERR_INVALID; }
    for (u32 i = 0; i < target_shard->n; ++i)
        if (target_shard->h_len[i] != all_target_lens[shard_first + i]) { LRGE_SET_ERR(ctx, "index_build_sharded: read %u of the shard does not have the length of target read %u", i, shard_first + i); return LRGE_ERR_INVALID; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    lrge_hip_seqset *meta = nullptr;
    int rc = seqset_describe(ctx, all_target_lens, n_targets, all_target_ranks, &meta);
    if (rc) { (void)comm_agree(comm, rc, ctx->stream); return rc; }          // (the others are entering the build's first collective)
    IndexBuildOpts ro; ro.restrict_to = streamed; ro.comm = comm; ro.shard = target_shard; ro.shard_first = shard_first;
    rc = index_build_one(ctx, meta, preset, out, &ro);
    if (rc) { lrge_hip_seqset_free(meta); return rc; }
    (*out)->owned_seqs = meta;
    return LRGE_OK;
}

extern "C" int lrge_hip_last_shard_stats(const lrge_hip_ctx *ctx, uint64_t out[8]) {
    if (!ctx || !out) return LRGE_ERR_INVALID;
    memcpy(out, ctx->shard_stats, sizeof(ctx->shard_stats));
    return LRGE_OK;
}

extern "C" void lrge_hip_index_free(lrge_hip_index *ix) {
    if (!ix) return;
    if (!ix->parts.empty() || !ix->part_sets.empty()) {
        for (lrge_hip_index *p : ix->parts) lrge_hip_index_free(p);
        for (lrge_hip_seqset *v : ix->part_sets) lrge_hip_seqset_free(v);
        delete ix;
        return;
    }
    bool ctx_alive;
    { std::lock_guard<std::mutex> g(g_live_mu); ctx_alive = g_live_ctx.count(ix->ctx) != 0; }
    if (ix->owned_seqs) lrge_hip_seqset_free(ix->owned_seqs);
    if (ctx_alive) {     // (a destroyed context has already freed its pool: an index that outlives it owns nothing)
        ix->ctx->pool.release(ix->d_pos); if (ix->d_skey && ix->d_skey != ix->d_pos) ix->ctx->pool.release(ix->d_skey);
        ix->ctx->pool.release(ix->d_ht);
    }
    delete ix;
}

extern "C" int lrge_hip_index_stats(const lrge_hip_index *ix, uint64_t *n_minimizers, uint64_t *n_keys, int32_t *mid_occ) {
    if (!ix) return LRGE_ERR_INVALID;
    if (n_minimizers) *n_minimizers = ix->n_mz;
    if (n_keys) *n_keys = ix->n_keys;
    if (mid_occ) *mid_occ = ix->mid_occ;
    return LRGE_OK;
}

extern "C" int lrge_hip_index_dump(lrge_hip_ctx *ctx, const lrge_hip_index *ix, uint64_t *keys, uint64_t *pos, uint64_t cap,
                                   uint64_t *n_out) {
    if (!ctx || !ix || !n_out) return LRGE_ERR_INVALID;
    if (!ix->parts.empty()) { LRGE_SET_ERR(ctx, "index_dump: not implemented for a partitioned index"); return LRGE_ERR_TOO_MANY; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    ctx->pin_items.clear(); ctx->pin_used = 0;      // reads an earlier, failed call may have left queued
    *n_out = ix->n_entries;
    u64 m = ix->n_entries < cap ? ix->n_entries : cap;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));   // blocking copies below run on the null stream
    // the device keeps the stream ordered by the byte-reversed hash (k_index.h); the dump presents it in
    // ascending hash order, lists ascending in y, i.e. the order mm_idx_get users see (debug / test entry point)
    std::vector<u64> hk(ix->n_entries), hp(ix->n_entries);
    if (ix->n_entries) {
        HIPCHK(ctx, hipMemcpy(hk.data(), ix->d_skey, ix->n_entries * 8, hipMemcpyDeviceToHost));
        HIPCHK(ctx, hipMemcpy(hp.data(), ix->d_pos, ix->n_entries * 8, hipMemcpyDeviceToHost));
    }
    if (ix->pk_ybits) {   // packed entries -> (hash, y)
        const u64 ym = (1ULL << ix->pk_ybits) - 1, pm = (1ULL << ix->pk_pos1) - 1;
        for (u64 i = 0; i < ix->n_entries; ++i) {
            const u64 e = hk[i], yb = e & ym;
            hk[i] = e >> ix->pk_ybits; hp[i] = (yb >> ix->pk_pos1) << 32 | (yb & pm);
        }
        if (!ix->h_seg_start.empty())        // segment-packed: the low hash byte is the number of the entry's segment
            for (u32 sgm = 0; sgm < 256; ++sgm)
                for (u64 i = ix->h_seg_start[sgm]; i < ix->h_seg_start[sgm + 1]; ++i) hk[i] = hk[i] << 8 | sgm;
    }
    std::vector<u32> ord(ix->n_entries);
    for (u64 i = 0; i < ix->n_entries; ++i) ord[i] = (u32)i;
    std::stable_sort(ord.begin(), ord.end(), [&](u32 a, u32 b) { return hk[a] < hk[b]; });
    for (u64 i = 0; i < m; ++i) {
        if (keys) keys[i] = hk[ord[i]];
        if (pos) pos[i] = hp[ord[i]];
    }
    return LRGE_OK;
}

// ------------------------------------------------------------------------------------------
// overlap core
// ------------------------------------------------------------------------------------------
enum { MODE_TWOSET = 0, MODE_INVERSE = 1, MODE_AVA = 2 };

// The streamed set's sketch, kept across the parts of a partitioned index (every part sees the same queries: sketched once, not
// once per part -- 8 x 12.7 ms at full-size C5)
struct SketchCache { std::unique_ptr<Scratch> sc; SketchOut so; std::vector<u32> h_mzoff; bool valid = false; };

struct OverlapJob {
    int mode;
    int dual;                       // 1: NO_DUAL cleared, 0: set
    lrge_hip_params prm;
    // outputs (host)
    u32 *counts = nullptr;          // size: nq (twoset) or n_indexed (inverse / ava)
    u32 *has_map = nullptr;
    lrge_hip_chain *chains = nullptr; u64 chain_cap = 0; u64 *n_chains = nullptr;
    // anchors of one query instead of chaining
    bool dump_anchors = false; u32 dump_query = 0; u64 *ax = nullptr, *ay = nullptr; u64 acap = 0; u64 *an = nullptr;
    // per-query PAF statistics instead of chaining
    bool paf_stats = false; i32 *rep_len = nullptr; u64 *sum_span = nullptr; u32 *n_kept = nullptr;
    // one part of a partitioned index (the entry points loop over the parts)
    u32 rid_base = 0;                                   // first read of the part in the whole indexed set
    const lrge_hip_seqset *indexed_top = nullptr;       // all-vs-all: the whole indexed set (counts are keyed by it)
    u32 *d_hc_acc = nullptr; bool hc_last = true;       // paf_stats: occurrence counts accumulated over the parts (device)
    const u32 *d_hc_global = nullptr;                   // chain records: those counts, complete (a seed's rank among the KEPT seeds
                                                        // of its query -- n_seeds / dv -- counts seeds kept in ANY part)
    SketchCache *qcache = nullptr;                      // the streamed set's sketch, shared by the parts' runs
};


// Split of the size-sorted group list between the two chain kernels, from the size census of k_group_count (hn / ha:
// groups and anchors per class of GSZ_W anchors).  Groups above T anchors -> k_chain_hw (~0.55 us per anchor of latency,
// ~93 VALU instructions per anchor), the rest -> k_chain_lpg (~4.1 us per anchor of the LONGEST group of a wavefront,
// ~21 VALU per anchor).  Both run side by side; the stage takes about
//   max(T * t_lpg, n_longest * t_hw, VALU work / issue rate of the chip)
// and T (a multiple of GSZ_W) minimises that estimate -- measured constants of this kernel pair on MI355X.
// `fixed` != LPG_MAX_AUTO pins T (LRGE_HIP_LPG_MAX / LRGE_HIP_CHAIN=hw|lpg).
struct ChainSplit { u32 T, n_big; unsigned long long a_big; int top; };
static ChainSplit choose_chain_split(const u32 *hn, const unsigned long long *ha, unsigned long long a_chained, u32 fixed, int n_cu) {
    ChainSplit r; r.T = fixed; r.n_big = 0; r.a_big = 0; r.top = -1;
    for (int b = 0; b < GSZ_BINS; ++b) if (hn[b]) r.top = b;
    if (fixed == LPG_MAX_AUTO) {
        // measured constants of this kernel pair on MI355X.  `rate` is the wave64 VALU instruction rate the chip sustains for
        // k_chain_lpg at its residency (1.25 wavefronts per SIMD, bounded by LDS) -- 422 G/s measured at C4; a shape with four
        // wavefronts per workgroup and twice the residency was measured too (round 2): every step took 1.4x as long and the
        // stage was slower or equal on C2, C4 and C5/10 alike, because the stage is bound by T * t_lpg, not by throughput
        const double t_lpg = 4.1e-6, t_hw = 0.55e-6, c_lpg = 21.0, c_hw = 93.0;
        const double rate = 0.8 * (double)n_cu * 4 * 2.1e9 / 4.0;
        double best = 1e30, a_le = 0;    // a_le: anchors in classes <= b
        r.T = 0;
        for (int b = -1; b < GSZ_BINS - 1; ++b) {      // T = (b + 1) * GSZ_W: classes 0..b go to k_chain_lpg
            if (b >= 0) a_le += (double)ha[b];
            const double a_hw = (double)a_chained - a_le;
            const double crit_lpg = b >= 0 ? (double)std::min<int>(b + 1, r.top + 1) * GSZ_W * t_lpg : 0.0;
            const double crit_hw = a_hw > 0 ? (double)(r.top + 1) * GSZ_W * t_hw : 0.0;
            const double est = std::max(std::max(crit_lpg, crit_hw), (a_le * c_lpg + a_hw * c_hw) / rate);
            if (est < best - 1e-9) { best = est; r.T = (u32)(b + 1) * GSZ_W; }
            if (b >= r.top) break;
        }
    }
    // groups above T: whole classes (class b = (b*W, (b+1)*W]); a pinned T that is no class edge counts by class floor --
    // any split point of the sorted list is valid, only the balance depends on it
    for (int b = 0; b < GSZ_BINS; ++b)
        if ((u64)b * GSZ_W >= (u64)r.T) { r.n_big += hn[b]; r.a_big += ha[b]; }
    return r;
}

// One overlap call = one OverlapRun: the state every stage shares lives here, the stages are its methods
// (prepare -> seeds -> plan -> batch x N -> finish); a stage returns RUN_DONE when the call is complete early
// (empty sets, statistics-only or anchor-dump runs).
enum { RUN_DONE = 1 };

struct OverlapRun {
    lrge_hip_ctx *ctx; const lrge_hip_index *ix; const lrge_hip_seqset *Q; OverlapJob &job;
    Scratch sc;
    // outputs on the device
    u32 n_out = 0; u32 *d_qmap = nullptr, *d_counts = nullptr, *d_hasmap = nullptr;
    unsigned long long *d_nchains = nullptr; lrge_hip_chain *d_chains = nullptr;
    bool need_rank = true;      // seed ranks (krank) are wanted by this run's anchors
    // seeds: query minimizers, their index lookups, per-query anchor totals
    SketchOut so; std::vector<u32> h_mzoff, h_qtot; u64 Mq = 0; SeedParams sp;
    std::unique_ptr<Scratch> presk_sc;   // memory of a consumed presketch (released with the run)
    u64 *hs = nullptr;                   // where every seed's list lives: start in pos[], or HT_INLINE | y (k_index.h)
    u32 *hc = nullptr, *hn = nullptr, *hv = nullptr, *krank = nullptr, *aoff_all = nullptr;
    // batch plan
    u64 batch_cap = 0; KeyLayout kl; u32 max_bits_q = 0, min_n = 0; ChainParams cp;
    std::vector<SegTile> h_tiles;   // per batch; lives until the batch's next host sync (the async H2D copy reads it)
    std::vector<SegDesc> h_local[3];
    u32 n_local_items = 0;         // anchors of the batch sorted by k_seg_sort_local

    OverlapRun(lrge_hip_ctx *c, const lrge_hip_index *i, const lrge_hip_seqset *q, OverlapJob &j) : ctx(c), ix(i), Q(q), job(j), sc(c) {}
    int prepare();                              // output buffers, shard map, empty-set shortcut
    int seeds();                                // K1 sketch, K3 lookup, K4a query-occurrence filter, hit counts
    int plan();                                 // batch size, key layout, chaining parameters
    int batch(u32 q0, u32 q1, u64 A);           // K4 expand, sort, K5 groups, K6 chain, K7 count for queries [q0, q1)
    int finish();                               // results to the host
    void plan_anchor_sort(u32 q0, u32 q1, bool packed);          // which queries sort inside LDS, tiles for the rest
    int dump_sorted_anchors(const u64 *skey, const u64 *sval, u64 A);   // lrge_hip_anchors_dump: one query's anchors, mm2 encoding
};

int OverlapRun::prepare() {
    const lrge_hip_seqset *T = ix->seqs; const Preset &P = ix->P; const u32 nq = Q->n, nt = T->n;
    (void)T; (void)P; (void)nq; (void)nt;
    const lrge_hip_seqset *I = (job.mode == MODE_AVA && job.indexed_top) ? job.indexed_top : T;   // what the counts are keyed by
    n_out = job.mode == MODE_TWOSET ? nq : (job.mode == MODE_AVA ? I->n : nt);
    if (job.mode == MODE_AVA && Q != I) {
        // a shard of the reads as queries: counts stay keyed by indexed read, so every query needs the index of the
        // read with the same name (= the same rank) in the indexed set
        const u32 ni = I->n;
        std::vector<std::pair<u32, u32>> byrank(ni);
        for (u32 i = 0; i < ni; ++i) byrank[i] = {I->h_rank[i], i};
        std::sort(byrank.begin(), byrank.end());
        std::vector<u32> qm(nq);
        for (u32 q = 0; q < nq; ++q) {
            auto it = std::lower_bound(byrank.begin(), byrank.end(), std::make_pair(Q->h_rank[q], 0u));
            if (it == byrank.end() || it->first != Q->h_rank[q]) { LRGE_SET_ERR(ctx, "all-vs-all shard: read %u is not in the indexed set", q); return LRGE_ERR_INVALID; }
            qm[q] = it->second;
        }
        d_qmap = sc.get<u32>((size_t)nq + 1);
        if (!d_qmap) return LRGE_ERR_DEVICE;
        HIPCHK(ctx, hipMemcpyAsync(d_qmap, qm.data(), (size_t)nq * 4, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));   // qm is a local
    }
    d_counts = sc.get<u32>((size_t)n_out + 1); d_hasmap = sc.get<u32>((size_t)nq + 1);
    if (!d_counts || !d_hasmap) return LRGE_ERR_DEVICE;
    HIPCHK(ctx, hipMemsetAsync(d_counts, 0, ((size_t)n_out + 1) * 4, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(d_hasmap, 0, ((size_t)nq + 1) * 4, ctx->stream));
    if (job.n_chains) {
        d_nchains = (unsigned long long *)sc.get<u64>(1);
        d_chains = sc.get<lrge_hip_chain>(job.chain_cap ? job.chain_cap : 1);
        if (!d_nchains || !d_chains) return LRGE_ERR_DEVICE;
        HIPCHK(ctx, hipMemsetAsync(d_nchains, 0, 8, ctx->stream));
    }
    ctx->counters[LRGE_C_QUERY_BASES] = Q->total_bases;
    if (nq == 0 || nt == 0) {
        if (job.counts) memset(job.counts, 0, (size_t)n_out * 4);
        if (job.has_map) memset(job.has_map, 0, (size_t)nq * 4);
        if (job.n_chains) *job.n_chains = 0;
        if (job.an) *job.an = 0;
        if (job.paf_stats) { memset(job.rep_len, 0, (size_t)nq * 4); memset(job.sum_span, 0, (size_t)nq * 8); memset(job.n_kept, 0, (size_t)nq * 4); }
        return RUN_DONE;
    }
    return LRGE_OK;

}

int OverlapRun::seeds() {
    const lrge_hip_seqset *T = ix->seqs; const Preset &P = ix->P; const u32 nq = Q->n, nt = T->n;
    (void)T; (void)P; (void)nq; (void)nt;
    // ---- 1. sketch the queries ----
    int rc = LRGE_OK;
    if (Q->presk && Q->presk->preset == ix->preset_id) {
        // sketched ahead on the side stream (lrge_hip_seqset_presketch): wait for it on the device, fetch the count and
        // the per-read offsets in the one round trip the in-line sketch pays too, and keep its memory until the call ends
        PreSketch *p = Q->presk;
        const_cast<lrge_hip_seqset *>(Q)->presk = nullptr;
        presk_sc.reset(p->sc);
        HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, p->ev_done, 0));
        u32 total = 0;
        h_mzoff.resize((size_t)Q->n + 1);
        HIPCHK(ctx, ctx->d2h(&total, p->d_total, 4, ctx->stream));
        HIPCHK(ctx, ctx->d2h(h_mzoff.data(), p->mz_off, ((size_t)Q->n + 1) * 4, ctx->stream));
        HIPCHK(ctx, ctx->d2h_sync(ctx->stream));
        so.x = p->x; so.y = p->y; so.mz_off = p->mz_off; so.n = total;
        ctx->timers.push_back(TimerRec{LRGE_T_SKETCH, p->ev_start, p->ev_done});   // both have completed; resolved with the call's timers
        delete p;
        if (job.qcache) {     // (the other parts of a partitioned index reuse it)
            job.qcache->sc = std::move(presk_sc); job.qcache->so = so; job.qcache->h_mzoff = h_mzoff; job.qcache->valid = true;
        }
    } else if (job.qcache && job.qcache->valid) {
        so = job.qcache->so; h_mzoff = job.qcache->h_mzoff;
    } else {
        if (job.qcache && !job.qcache->sc) job.qcache->sc.reset(new Scratch(ctx));
        rc = sketch_device(ctx, job.qcache ? *job.qcache->sc : sc, Q, ix->preset_id, false, &so, 0, 0, &h_mzoff);
        if (rc) return rc;
        if (job.qcache) { job.qcache->so = so; job.qcache->h_mzoff = h_mzoff; job.qcache->valid = true; }
    }
    Mq = so.n;
    ctx->counters[LRGE_C_QUERY_MINIMIZERS] = Mq;
    if (Mq >= (1ULL << 32)) { LRGE_SET_ERR(ctx, "query set limited to < 2^32 minimizers"); return LRGE_ERR_TOO_MANY; }

    // ---- 2. lookup ----
    sp.ht = ix->d_ht; sp.ht_cap = ix->ht_cap; sp.ht_fix = ix->ht_fix; sp.pos = ix->d_pos; sp.pk_pos1 = ix->pk_pos1; sp.pk_ybits = ix->pk_ybits;
    sp.t_len = T->d_len; sp.t_rank = T->d_rank; sp.q_len = Q->d_len; sp.q_rank = Q->d_rank;
    sp.mid_occ = ix->mid_occ;
    sp.check_names = (Q->has_rank && T->has_rank) ? 1 : 0;   // qname == NULL in minimap2 skips skip_seed entirely
    if (sp.check_names && job.dual) {
        // with --dual=yes skip_seed only ever fires for a query that IS one of the indexed reads (same name, same
        // length, same position).  Ranks are positions in the sorted union of names, so if no rank occurs in both
        // sets (the two-set strategies) no hit can be skipped and the per-hit name checks are dropped altogether.
        const bool shared = ranks_intersect(Q->h_rank, T->h_rank);
        if (!shared) sp.check_names = 0;
    }
    sp.no_dual = job.dual ? 0 : 1;
    // without name checks every kept hit survives skip_seed: hv IS hn, and k_lookup fills it (k_seed_counts only runs again
    // if the exact query occurrence filter had to change hc)
    const bool counts_in_lookup = !sp.check_names && !ctx->opt("COUNTS_AFTER_LOOKUP");   // (option: the separate pass, for A/B runs)
    hs = sc.get<u64>(Mq + 1); hc = sc.get<u32>(Mq + 1); hn = sc.get<u32>(Mq + 1); hv = counts_in_lookup ? hn : sc.get<u32>(Mq + 1); krank = sc.get<u32>(Mq + 1);
    u32 *d_qtot = sc.get<u32>((size_t)nq + 1);
    aoff_all = sc.get<u32>(Mq + 1);
    if (!hs || !hc || !hn || !hv || !krank || !d_qtot || !aoff_all) return LRGE_ERR_DEVICE;
    h_qtot.assign((size_t)nq + 1, 0);
    if (Mq) {
        StageTimer t(ctx, LRGE_T_LOOKUP), tk(ctx, LRGE_T_K_LOOKUP);
        hipLaunchKernelGGL(k_lookup, dim3((u32)div_up(Mq, 256)), dim3(256), 0, ctx->stream, so.x, Mq, sp, hs, hc, counts_in_lookup ? hn : (u32 *)nullptr);
        KCHK(ctx);
        tk.stop(); t.stop();
        ctx->counters[LRGE_C_LOOKUP_LAUNCHES] += 1;
    }

    // ---- 3. query occurrence filter (mm_seed_mz_flt) ----
    // minimap2 applies it before the lookup; the result is the same afterwards, restricted to the
    // minimizers present in the index: every occurrence of a value x in one query gets the same lookup
    // result, so the per-query multiplicity of x is fully visible inside that subset, and absent values
    // contribute nothing whether removed or not.  A removed minimizer is marked absent (hc = 0).
    // The exact filter (two radix sorts + a run-length mark) as a callable: it only runs when the conservative
    // pre-check k_qocc_check cannot rule it out, or when LRGE_HIP_QOCC_EXACT forces it (tests).
    const u32 *d_qsel = nullptr;      // per-query verdicts of the pre-check (null: the exact pass takes every query)
    auto run_exact_qocc = [&]() -> int {
        StageTimer t(ctx, LRGE_T_QFILTER);
        ALLOC_OR_FAIL(flag, sc, u32, Mq); ALLOC_OR_FAIL(fpos, sc, u32, Mq); ALLOC_OR_FAIL(d_ns, sc, u32, 1);
        hipLaunchKernelGGL(k_flag_present_sel, dim3((u32)div_up(Mq, 256)), dim3(256), 0, ctx->stream, hc, so.y, d_qsel, Mq, flag);
        KCHK(ctx);
        rc = scan_exclusive_u32(ctx, sc, flag, fpos, Mq, d_ns);
        if (rc) return rc;
        u32 Ms = 0;
        HIPCHK(ctx, ctx->d2h(&Ms, d_ns, 4, ctx->stream));
        HIPCHK(ctx, ctx->d2h_sync(ctx->stream));
        if (Ms > (u32)ix->mid_occ) {
            ALLOC_OR_FAIL(ka, sc, u64, Ms); ALLOC_OR_FAIL(va, sc, u64, Ms);
            ALLOC_OR_FAIL(kb, sc, u64, Ms); ALLOC_OR_FAIL(vb, sc, u64, Ms);
            hipLaunchKernelGGL(k_qocc_keys, dim3((u32)div_up(Mq, 256)), dim3(256), 0, ctx->stream, so.x, so.y, flag, fpos, Mq, ka, va);
            KCHK(ctx);
            u64 *rk, *rv;
            rc = radix_sort_pairs(ctx, sc, ka, va, kb, vb, Ms, 0, 2 * P.k + 8, &rk, &rv);   // by x
            if (rc) return rc;
            u64 *ok = (rk == ka) ? kb : ka, *ov = (rv == va) ? vb : va;
            u64 *rk2, *rv2;
            // then (stable) by query id held in bits [32, 32+bits) of the value: swap roles
            rc = radix_sort_pairs(ctx, sc, rv, rk, ov, ok, Ms, 32, (int)ceil_log2_u64((u64)nq + 1), &rk2, &rv2);
            if (rc) return rc;
            hipLaunchKernelGGL(k_qocc_mark, dim3((u32)div_up(Ms, 256)), dim3(256), 0, ctx->stream, rv2 /* x */, rk2 /* (q,idx) */,
                               (u64)Ms, so.mz_off, ix->mid_occ, P.q_occ_frac, hc);
            KCHK(ctx);
            sc.drop(ka); sc.drop(va); sc.drop(kb); sc.drop(vb);
        }
        sc.drop(flag); sc.drop(fpos); sc.drop(d_ns);
        t.stop();
        return LRGE_OK;
    };
    bool qocc_possible = false;
    bool hc_changed = false;       // by run_exact_qocc: k_lookup's own kept counts are stale then
    if (Mq > 0 && P.q_occ_frac > 0.0f && ix->mid_occ > 0)   // only queries with more minimizers than mid_occ can be affected
        for (u32 q = 0; q < nq && !qocc_possible; ++q) qocc_possible = (i64)(h_mzoff[q + 1] - h_mzoff[q]) > (i64)ix->mid_occ;
    u32 *d_qf = nullptr; u32 qf = 0; bool qf_on_side = false;
    if (qocc_possible) {
        if (ctx->opt("QOCC_EXACT")) { rc = run_exact_qocc(); if (rc) return rc; hc_changed = true; }
        else {
            // cheap conservative check, on the side stream beside the hit counting below (both only read the lookup
            // results); its verdict travels to the host with the next sync (no extra round trip)
            d_qf = sc.get<u32>((size_t)nq + 1);     // [0] any query, [1 + q] query q
            if (!d_qf) return LRGE_ERR_DEVICE;
            d_qsel = d_qf;
            HIPCHK(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
            HIPCHK(ctx, hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
            StageTimer t(ctx, LRGE_T_QFILTER, ctx->stream2);
            HIPCHK(ctx, hipMemsetAsync(d_qf, 0, ((size_t)nq + 1) * 4, ctx->stream2));
            hipLaunchKernelGGL(k_qocc_check, dim3(nq), dim3(256), 0, ctx->stream2, so.x, hc, so.mz_off, nq, ix->mid_occ, d_qf);
            KCHK(ctx);
            t.stop();
            HIPCHK(ctx, hipEventRecord(ctx->ev_join, ctx->stream2));
            qf_on_side = true;
        }
    }
    if (job.paf_stats) {   // per-query seed statistics only (rl, avg_k ingredients)
        if (d_qf) {
            if (qf_on_side) { HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0)); qf_on_side = false; }
            HIPCHK(ctx, ctx->d2h(&qf, d_qf, 4, ctx->stream));
            HIPCHK(ctx, ctx->d2h_sync(ctx->stream));
            if (qf) { rc = run_exact_qocc(); if (rc) return rc; }
        }
        const u32 *hc_stats = hc;
        if (job.d_hc_acc) {      // one part of a partitioned index: the statistics need the counts over all parts
            if (Mq) { hipLaunchKernelGGL(k_hc_accumulate, dim3((u32)div_up(Mq, 256)), dim3(256), 0, ctx->stream, hc, Mq, (u32)ix->mid_occ, job.d_hc_acc); KCHK(ctx); }
            if (!job.hc_last) { HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); return RUN_DONE; }
            hc_stats = job.d_hc_acc;
        }
        ALLOC_OR_FAIL(d_rl, sc, i32, (size_t)nq); ALLOC_OR_FAIL(d_ss, sc, u64, (size_t)nq); ALLOC_OR_FAIL(d_nk, sc, u32, (size_t)nq);
        hipLaunchKernelGGL(k_query_paf_stats, dim3((u32)div_up(nq, 64)), dim3(64), 0, ctx->stream, so.x, so.y, hc_stats, so.mz_off, nq, ix->mid_occ,
                           d_rl, d_ss, d_nk);
        KCHK(ctx);
        HIPCHK(ctx, hipMemcpyAsync(job.rep_len, d_rl, (size_t)nq * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(job.sum_span, d_ss, (size_t)nq * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(job.n_kept, d_nk, (size_t)nq * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        return RUN_DONE;
    }
    auto run_counts = [&]() -> int {
        StageTimer t(ctx, LRGE_T_LOOKUP);
        if (Mq) {
            if (!counts_in_lookup || hc_changed) {
                hipLaunchKernelGGL(k_seed_counts, dim3((u32)div_up(Mq, 256)), dim3(256), 0, ctx->stream, so.y, Mq, sp, hs, hc, hn, hv);
                KCHK(ctx);
            }
            // rank of every kept seed inside its query (= its index in minimap2's mini_pos[]): only chain records carry it
            // (mm_est_err's dv); a count-only run packs its anchors without it (OverlapRun::batch) and skips the flag + scan
            const u32 bits_rpos_ = std::max<u32>(1, ceil_log2_u64((u64)T->max_len + 1)), bits_rid_ = std::max<u32>(1, ceil_log2_u64((u64)T->n));
            const u32 bits_qy_ = std::max<u32>(1, ceil_log2_u64((u64)Q->max_len + 1));
            need_rank = d_chains || job.dump_anchors || bits_rpos_ + 1 + bits_rid_ + bits_qy_ + 9 > 64 || ctx->opt_u64("NO_PACKED", 0);
            if (need_rank) {
                ALLOC_OR_FAIL(kflag, sc, u32, Mq);
                if (job.d_hc_global) hipLaunchKernelGGL(k_flag_kept, dim3((u32)div_up(Mq, 256)), dim3(256), 0, ctx->stream, job.d_hc_global, Mq, (u32)ix->mid_occ, kflag);
                else hipLaunchKernelGGL(k_flag_nonzero, dim3((u32)div_up(Mq, 256)), dim3(256), 0, ctx->stream, hn, Mq, kflag);
                KCHK(ctx);
                rc = scan_exclusive_u32(ctx, sc, kflag, krank, Mq, krank + Mq);
                if (rc) return rc;
                sc.drop(kflag);
            }
        } else {
            HIPCHK(ctx, hipMemsetAsync(krank, 0, 4, ctx->stream));
        }
        // ONE scan of the surviving-hit counts over all query minimizers: the per-query totals are differences of it, and every
        // batch's k_expand reads its output offsets from it (relative to the batch's first minimizer; all modulo 2^32, so a job
        // with more than 2^32 anchors is fine as long as a batch -- at most 2^30 -- and a query stay below)
        if (Mq) {
            rc = scan_exclusive_u32(ctx, sc, hv, aoff_all, Mq, aoff_all + Mq);
            if (rc) return rc;
        } else HIPCHK(ctx, hipMemsetAsync(aoff_all, 0, 4, ctx->stream));
        hipLaunchKernelGGL(k_query_totals_from_scan, dim3((u32)div_up(nq, 256)), dim3(256), 0, ctx->stream, aoff_all, so.mz_off, nq, d_qtot);
        KCHK(ctx);
        HIPCHK(ctx, ctx->d2h(h_qtot.data(), d_qtot, (size_t)nq * 4, ctx->stream));
        if (d_qf) {
            if (qf_on_side) { HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0)); qf_on_side = false; }
            HIPCHK(ctx, ctx->d2h(&qf, d_qf, 4, ctx->stream));
        }
        HIPCHK(ctx, ctx->d2h_sync(ctx->stream));
        t.stop();
        return LRGE_OK;
    };
    rc = run_counts();
    if (rc) return rc;
    if (d_qf && qf) {   // the pre-check could not rule the filter out: apply it, then count again
        d_qf = nullptr;
        rc = run_exact_qocc(); if (rc) return rc;
        hc_changed = true;
        rc = run_counts(); if (rc) return rc;
    }
    return LRGE_OK;
}

int OverlapRun::plan() {
    const lrge_hip_seqset *T = ix->seqs; const Preset &P = ix->P; const u32 nq = Q->n, nt = T->n;
    (void)T; (void)P; (void)nq; (void)nt;
    // ---- 4. batches ----
    // Anchors per batch.  Every batch pays the latency of its longest chain group once (the chain kernels are
    // bound by it), so batches are as large as memory allows: ~64 B of scratch per anchor, at most half of the
    // free HBM, at most 2^30 anchors (positions are 32-bit).
    batch_cap = 1ULL << 30;
    {
        size_t mfree = 0, mtotal = 0;
        if (hipMemGetInfo(&mfree, &mtotal) == hipSuccess) {
            const u64 by_mem = ((u64)mfree + ctx->pool.idle()) / 2 / 64;     // the pool's idle blocks are reusable too (not the ones in use: a resident index)
            if (by_mem < batch_cap) batch_cap = by_mem;
        }
        if (batch_cap < (1ULL << 20)) batch_cap = 1ULL << 20;
    }
    batch_cap = ctx->opt_u64("BATCH_ANCHORS", batch_cap);
    kl.bits_rpos = std::max<u32>(1, ceil_log2_u64((u64)T->max_len + 1));
    kl.bits_rid = std::max<u32>(1, ceil_log2_u64((u64)nt));
    max_bits_q = 63 - (kl.bits_rpos + 1 + kl.bits_rid);
    min_n = std::max<u32>((u32)P.min_cnt, (u32)div_up((u64)P.min_sc, P.hpc ? 255 : (u64)P.k));
    cp.max_dist_x = std::max(P.max_gap, P.bw); cp.max_dist_y = std::max(P.max_gap, P.bw);
    cp.bw = P.bw; cp.max_skip = P.max_skip; cp.max_iter = P.max_iter; cp.min_cnt = P.min_cnt; cp.min_sc = P.min_sc;
    cp.max_drop = P.bw; cp.pen_gap = P.pen_gap; cp.pen_skip = P.pen_skip;
    cp.remove_internal = job.prm.remove_internal ? (job.mode == MODE_INVERSE ? 2 : 1) : 0;
    cp.max_overhang_ratio = job.prm.max_overhang_ratio;
    cp.want_all = (job.n_chains != nullptr || cp.remove_internal) ? 1 : 0;
    cp.q_len = Q->d_len; cp.t_len = T->d_len;
    return LRGE_OK;
}

// The expansion emits the anchors query by query, so only (target, strand, position) need sorting, inside every
// query's segment.  Packed (count-only) runs sort the segments that fit a workgroup's LDS there (k_seg_sort_local,
// capacity classes 2048 / 8192 / 16384 anchors); everything else is cut into RS_TILE tiles for the segmented
// global passes (SegTile, k_prims.h), whose scanned histogram is offset by the items sorted locally (delta).
void OverlapRun::plan_anchor_sort(u32 q0, u32 q1, bool packed) {
    h_tiles.clear();
    for (auto &v : h_local) v.clear();
    u32 off = 0, tb = 0, &n_local = n_local_items;
    n_local = 0;
    const bool local_ok = !ctx->opt("NO_LOCAL_SORT");
    const int local_max = ctx->opt("LOCAL_SORT_MAX") ? atoi(ctx->opt("LOCAL_SORT_MAX")) : 2;   // largest class sorted in LDS
    for (u32 q = q0; q < q1; ++q) {
        const u32 c = h_qtot[q];
        if (packed && c) {
            const int cls = c <= 2048 ? 0 : c <= 8192 ? 1 : c <= 16384 ? 2 : 3;
            if (cls < 3 && cls <= local_max && local_ok && ctx->lsort_ok[cls]) { h_local[cls].push_back(SegDesc{off, c, q - q0, 0}); off += c; n_local += c; continue; }
        }
        const u32 nt_q = (u32)div_up((u64)c, RS_TILE);
        for (u32 lt = 0; lt < nt_q; ++lt) {
            SegTile t; t.start = off + lt * RS_TILE; t.len = std::min<u32>(RS_TILE, c - lt * RS_TILE);
            t.hbase = 256u * tb + lt; t.hstride = nt_q; t.seg = q - q0; t.delta = n_local;
            h_tiles.push_back(t);
        }
        off += c; tb += nt_q;
    }
}

int OverlapRun::dump_sorted_anchors(const u64 *skey, const u64 *sval, u64 A) {
    *job.an = A;
    u64 m = A < job.acap ? A : job.acap;
    std::vector<u64> hk(m), hvv(m);
    if (m) {
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));   // blocking copies below run on the null stream
        HIPCHK(ctx, hipMemcpy(hk.data(), skey, m * 8, hipMemcpyDeviceToHost));
        HIPCHK(ctx, hipMemcpy(hvv.data(), sval, m * 8, hipMemcpyDeviceToHost));
    }
    const u64 rmask = (1ULL << kl.bits_rpos) - 1;
    // back to minimap2's mm128 anchor encoding and array order: the device orders groups
    // (target, strand) so that both strands of a pair are adjacent, minimap2 orders them
    // (strand, target); a stable re-sort by x keeps the order inside every group.
    std::vector<std::pair<u64, u64>> tmp(m);
    for (u64 i = 0; i < m; ++i) {
        u64 k = hk[i];
        u64 rev = (k >> kl.sh_rev()) & 1, rid = (k >> kl.sh_rid()) & ((1ULL << kl.bits_rid) - 1);
        tmp[i] = {rev << 63 | rid << 32 | (k & rmask), hvv[i] & AVAL_LOW_MASK};   // drop the seed rank
    }
    std::stable_sort(tmp.begin(), tmp.end(), [](const std::pair<u64, u64> &a, const std::pair<u64, u64> &b) { return a.first < b.first; });
    for (u64 i = 0; i < m; ++i) { job.ax[i] = tmp[i].first; job.ay[i] = tmp[i].second; }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return RUN_DONE;
}

int OverlapRun::batch(u32 q0, u32 q1, u64 A) {
    const lrge_hip_seqset *T = ix->seqs; const Preset &P = ix->P; const u32 nq = Q->n, nt = T->n;
    (void)T; (void)P; (void)nq; (void)nt;
    int rc = 0;
    const u64 mb = h_mzoff[q0], me = h_mzoff[q1];
    if (A == 0 || me == mb) return LRGE_OK;
    ctx->counters[LRGE_C_ANCHORS] += A;
    Scratch bsc(ctx);
    u64 *akey, *aval, *akey2, *aval2, *skey, *sval;
    // count-only runs carry one packed u64 per anchor through the expansion and the sort (k_prims.h UnpackParams);
    // chain records (PAF) need the seed rank as well and keep the (key, value) pairs
    const u32 bits_qy = std::max<u32>(1, ceil_log2_u64((u64)Q->max_len + 1));
    const bool packed = !d_chains && !job.dump_anchors && kl.sh_q() + bits_qy + 9 <= 64 && !ctx->opt_u64("NO_PACKED", 0);
    {
        StageTimer t(ctx, LRGE_T_EXPAND);
        const u32 *aoff = aoff_all;
        // (+8: k_chain_lpg streams anchors in 16-byte pairs and may read one element past the last group)
        akey = bsc.get<u64>(A + 8); aval = bsc.get<u64>(A + 8); akey2 = bsc.get<u64>(A + 8); aval2 = bsc.get<u64>(A + 8);
        if (!aoff || !akey || !aval || !akey2 || !aval2) return LRGE_ERR_DEVICE;
        hipLaunchKernelGGL(k_expand, dim3((u32)div_up(me - mb, 256)), dim3(256), 0, ctx->stream, so.x, so.y, mb, me, sp, hs, hn, aoff,
                           need_rank ? krank : (const u32 *)nullptr, so.mz_off, q0, kl, akey, aval, packed ? bits_qy : 0u);
        KCHK(ctx);
        // (no sync: everything runs in order on ctx->stream; scratch is recycled in stream order)
        t.stop();
    }
    {
        StageTimer t(ctx, LRGE_T_ANCHOR_SORT);
        plan_anchor_sort(q0, q1, packed);
        SegTile *d_tiles = (SegTile *)bsc.get<u32>(h_tiles.size() * (sizeof(SegTile) / 4) + 4);
        if (!d_tiles) return LRGE_ERR_DEVICE;
        HIPCHK(ctx, hipMemcpyAsync(d_tiles, h_tiles.data(), h_tiles.size() * sizeof(SegTile), hipMemcpyHostToDevice, ctx->stream));
        if (packed) {
            UnpackParams up; up.sb = kl.sh_q(); up.bits_qy = bits_qy; up.sh_q = kl.sh_q(); up.dmask = 255;
            // segments that fit a workgroup's LDS are sorted there in one kernel (k_seg_sort_local: 8 B in, 16 B out
            // per anchor); only the larger ones take the tiled global passes
            // the classes touch disjoint segments: the largest class runs on the side stream beside the others and the
            // tiled passes (fork / join with events), so that its one-block-per-CU tail does not stand alone
            const bool side = !h_local[2].empty() && (!h_local[1].empty() || !h_tiles.empty()) && !ctx->opt("LSORT_SERIAL");
            SegDesc *d_seg[3] = {nullptr, nullptr, nullptr};
            for (int cls = 0; cls < 3; ++cls) {
                if (h_local[cls].empty()) continue;
                d_seg[cls] = (SegDesc *)bsc.get<u32>(h_local[cls].size() * 4);
                if (!d_seg[cls]) return LRGE_ERR_DEVICE;
                HIPCHK(ctx, hipMemcpyAsync(d_seg[cls], h_local[cls].data(), h_local[cls].size() * sizeof(SegDesc), hipMemcpyHostToDevice, ctx->stream));
            }
            if (side) {
                HIPCHK(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
                HIPCHK(ctx, hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
            }
            const int nbits = (int)kl.sh_q();
            if (d_seg[2]) {
                hipLaunchKernelGGL((k_seg_sort_local<1024, 16, LSORT_DB>), dim3((u32)h_local[2].size()), dim3(1024), LSORT_BYTES(1024, 16, LSORT_DB), side ? ctx->stream2 : ctx->stream,
                                   akey, aval, aval2, d_seg[2], up, nbits);
                KCHK(ctx);
                if (side) HIPCHK(ctx, hipEventRecord(ctx->ev_join, ctx->stream2));
            }
            if (d_seg[1]) {
                hipLaunchKernelGGL((k_seg_sort_local<512, 16, LSORT_DB>), dim3((u32)h_local[1].size()), dim3(512), LSORT_BYTES(512, 16, LSORT_DB), ctx->stream, akey, aval, aval2, d_seg[1], up, nbits);
                KCHK(ctx);
            }
            if (d_seg[0]) {
                hipLaunchKernelGGL((k_seg_sort_local<256, 8, 8>), dim3((u32)h_local[0].size()), dim3(256), LSORT_BYTES(256, 8, 8), ctx->stream, akey, aval, aval2, d_seg[0], up, nbits);
                KCHK(ctx);
            }
            rc = radix_sort_packed_seg(ctx, bsc, akey, akey2, aval, aval2, A, (int)kl.sh_q(), d_tiles, (u32)h_tiles.size(), up, A - n_local_items);
            if (rc) return rc;
            if (side) HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
            skey = aval; sval = aval2;
            bsc.drop((u32 *)d_tiles);
            bsc.drop(akey); bsc.drop(akey2);
        } else {
            rc = radix_sort_pairs(ctx, bsc, akey, aval, akey2, aval2, A, 0, (int)(kl.bits_rpos + 1 + kl.bits_rid), &skey, &sval, false,
                                  d_tiles, (u32)h_tiles.size());
            if (rc) return rc;
            bsc.drop((u32 *)d_tiles);
            // (no sync: everything runs in order on ctx->stream; scratch is recycled in stream order)
            bsc.drop(skey == akey ? akey2 : akey);
            bsc.drop(sval == aval ? aval2 : aval);
        }
        t.stop();
    }
    if (job.dump_anchors) return dump_sorted_anchors(skey, sval, A);
    // groups.  The size-sorted list of the groups worth chaining is split: groups above lpg_max anchors go to k_chain_hw
    // (short latency per anchor), the rest to k_chain_lpg (64 groups per wavefront).  The split is chosen per batch from
    // the size census of the groups (see below); option LPG_MAX pins it, CHAIN=hw|lpg forces one kernel.
    const char *cm = ctx->opt("CHAIN");
    u32 lpg_max = LPG_MAX_AUTO;
    if (const char *e = ctx->opt("LPG_MAX")) lpg_max = (u32)strtoul(e, nullptr, 10);
    if (cm && !strcmp(cm, "hw")) lpg_max = 0;
    if (cm && !strcmp(cm, "lpg")) lpg_max = 0xFFFFFFFFu;
    if (lpg_max && (cp.want_all || d_chains) && !(cm && !strcmp(cm, "lpg"))) lpg_max = 0;   // records: wave-wide backtrack anyway
    if (cp.max_iter < LPG_W) lpg_max = 0;    // (debug knob only) k_chain_lpg assumes every window slot is a candidate
    u32 n_big = 0, lpg_split = 0;
    u32 G = 0; u32 *gstart, *gflags, *hw_list = nullptr;
    u32 n_chained = 0; unsigned long long a_chained = 0, a_big = 0;
    {
        StageTimer t(ctx, LRGE_T_GROUP);
        u32 *d_G = bsc.get<u32>(1);
        {
            // group starts into an upper-bound block (one entry per anchor): the group count stays on the device
            // until it travels to the host together with the size census -- one round trip instead of two
            gstart = bsc.get<u32>((size_t)A + 1);
            if (!gstart || !d_G) return LRGE_ERR_DEVICE;
            rc = compact_heads_async(ctx, bsc, skey, A, kl.bits_rpos, gstart, d_G);   // runs of equal (query, target, strand)
            if (rc) return rc;
        }
        {
            // groups worth chaining, sorted by size (largest first) so that k_chain_hw pairs equals
            u32 *d_cnt = bsc.get<u32>(4 + GSZ_BINS);
            unsigned long long *d_anch = (unsigned long long *)bsc.get<u64>(2 + GSZ_BINS);
            if (!d_cnt || !d_anch) return LRGE_ERR_DEVICE;
            HIPCHK(ctx, hipMemsetAsync(d_cnt, 0, (4 + GSZ_BINS) * 4, ctx->stream));
            HIPCHK(ctx, hipMemsetAsync(d_anch, 0, (2 + GSZ_BINS) * 8, ctx->stream));
            hipLaunchKernelGGL(k_group_count, dim3((u32)std::min<u64>(div_up(A, 4096), (u64)ctx->n_cu * 8)), dim3(256), 0, ctx->stream, gstart, d_G, A, min_n,
                               d_cnt, d_anch, d_cnt + 4, d_anch + 2);
            KCHK(ctx);
            u32 h_cnt[4 + GSZ_BINS]; unsigned long long h_anch[2 + GSZ_BINS];
            HIPCHK(ctx, ctx->d2h(&G, d_G, 4, ctx->stream));
            HIPCHK(ctx, ctx->d2h(h_cnt, d_cnt, sizeof(h_cnt), ctx->stream));
            HIPCHK(ctx, ctx->d2h(h_anch, d_anch, sizeof(h_anch), ctx->stream));
            HIPCHK(ctx, ctx->d2h_sync(ctx->stream));
            gflags = bsc.get<u32>((size_t)G + 1);
            if (!gflags) return LRGE_ERR_DEVICE;
            HIPCHK(ctx, hipMemsetAsync(gflags, 0, ((size_t)G + 1) * 4, ctx->stream));
            n_chained = h_cnt[0]; a_chained = h_anch[0];
            {
                const ChainSplit sp_ = choose_chain_split(h_cnt + 4, h_anch + 2, a_chained, lpg_max, ctx->n_cu);
                n_big = sp_.n_big; a_big = sp_.a_big; lpg_split = sp_.T;
                ctx->counters[LRGE_C_LPG_SPLIT] = lpg_split;
                if (ctx->opt("VERBOSE"))
                    fprintf(stderr, "[lrge_hip] batch: %u groups chained, %llu anchors, largest class %d (<= %d anchors), split T=%u -> hw %u groups / %llu anchors\n",
                            n_chained, a_chained, sp_.top, (sp_.top + 1) * GSZ_W, sp_.T, n_big, a_big);
            }
            if (n_chained) {
                u64 *k0 = bsc.get<u64>(n_chained), *v0 = bsc.get<u64>(n_chained), *k1 = bsc.get<u64>(n_chained), *v1 = bsc.get<u64>(n_chained);
                hw_list = bsc.get<u32>(n_chained);
                if (!k0 || !v0 || !k1 || !v1 || !hw_list) return LRGE_ERR_DEVICE;
                hipLaunchKernelGGL(k_group_fill, dim3((u32)div_up(G, GB_CHUNK)), dim3(256), 0, ctx->stream, gstart, G, A, min_n, d_cnt + 1, k0, v0);
                KCHK(ctx);
                u64 *rk, *rv;
                rc = radix_sort_pairs(ctx, bsc, k0, v0, k1, v1, n_chained, 0, 16, &rk, &rv);   // keys: 65535 - min(n, 65535)
                if (rc) return rc;
                hipLaunchKernelGGL(k_vals_to_u32, dim3((u32)div_up(n_chained, 256)), dim3(256), 0, ctx->stream, rv, n_chained, hw_list);
                KCHK(ctx);
                bsc.drop(k0); bsc.drop(v0); bsc.drop(k1); bsc.drop(v1);
            }
            bsc.drop(d_cnt); bsc.drop(d_anch);
        }
        t.stop();
    }
    ctx->counters[LRGE_C_GROUPS] += G;
    {
        GroupOut go; go.flags = gflags; go.chains = d_chains; go.n_chains = d_nchains; go.chain_cap = job.chain_cap; go.rid_base = job.rid_base;
        {
            if (n_chained) {
                StageTimer t(ctx, LRGE_T_CHAIN);
                HwChainArgs ha;
                ha.akey = skey; ha.aval = sval; ha.gstart = gstart; ha.n_groups = G; ha.n_anchors = A; ha.list = hw_list; ha.n_list = n_big;
                ha.grec = bsc.get<u64>(A); ha.tmark = bsc.get<u32>(A);
                ha.prio = (u32)ctx->opt_u64("HW_PRIO", 0);
                if (!ha.grec || !ha.tmark) return LRGE_ERR_DEVICE;
                HIPCHK(ctx, hipMemsetAsync(ha.tmark, 0, A * 4, ctx->stream));
                // the list is sorted by min(n, 65535) descending, so [0, n_big) are exactly the groups above lpg_max
                // the two kernels touch disjoint groups; k_chain_lpg goes to the side stream so that its long
                // wavefronts run beside k_chain_hw's (fork / join with events, no host sync)
                const bool both = n_big && n_chained > n_big;
                if (both) {
                    HIPCHK(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
                    HIPCHK(ctx, hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
                }
                if (n_chained > n_big) {
                    LpgChainArgs la;
                    la.akey = skey; la.aval = sval; la.gstart = gstart; la.n_groups = G; la.n_anchors = A;
                    la.list = hw_list + n_big; la.n_list = n_chained - n_big; la.grec = ha.grec; la.tmark = ha.tmark;
                    la.prio = (u32)ctx->opt_u64("LPG_PRIO", 3);
                    // 1024: on clean input (C2) no group is given up -- redoing even one 500-anchor group costs 0.3 ms of
                    // critical path; on a repeat-rich genome (synth c2_repeats) 64 would be ~1.7x faster still
                    la.slow_budget = (u32)ctx->opt_u64("LPG_SLOW_BUDGET", 1024);
                    la.slow_entries = (u32)ctx->opt_u64("LPG_SLOW_ENTRIES", 4);
                    la.redo_list = bsc.get<u32>((size_t)la.n_list + 1); la.redo_count = bsc.get<u32>(1);
                    if (!la.redo_list || !la.redo_count) return LRGE_ERR_DEVICE;
                    HIPCHK(ctx, hipMemsetAsync(la.redo_count, 0, 4, both ? ctx->stream2 : ctx->stream));
                    StageTimer tl(ctx, LRGE_T_CHAIN_LPG, both ? ctx->stream2 : ctx->stream);
                    const bool pentab = cp.pen_skip == 0.0f && cp.bw >= 0 && cp.bw + 2 <= 8192 && !ctx->opt("LPG_NOTAB");
                    const bool fastreach = cp.max_iter >= 64 && !ctx->opt("LPG_EXACT_REACH");
                    const dim3 lgrid((la.n_list + 64 * LPG_WAVES - 1) / (64 * LPG_WAVES)), lblock(64 * LPG_WAVES);
                    const size_t lds_ring = (size_t)LPG_WAVES * LPG_RING_BYTES;
                    const size_t lds_tab = (((size_t)cp.bw + 2) * 4 + 15) / 16 * 16 + lds_ring;
                    hipStream_t lst = both ? ctx->stream2 : ctx->stream;
                    if (pentab && fastreach) hipLaunchKernelGGL((k_chain_lpg<true, true>), lgrid, lblock, lds_tab, lst, la, cp, go);
                    else if (pentab) hipLaunchKernelGGL((k_chain_lpg<true, false>), lgrid, lblock, lds_tab, lst, la, cp, go);
                    else if (fastreach) hipLaunchKernelGGL((k_chain_lpg<false, true>), lgrid, lblock, lds_ring, lst, la, cp, go);
                    else hipLaunchKernelGGL((k_chain_lpg<false, false>), lgrid, lblock, lds_ring, lst, la, cp, go);
                    KCHK(ctx);
                    tl.stop();
                    ctx->counters[LRGE_C_CHAIN_LAUNCHES] += 1;
                    ctx->counters[LRGE_C_LPG_LAUNCHES] += 1;
                    ctx->counters[LRGE_C_LPG_ANCHORS] += a_chained - a_big;
                    {   // the groups k_chain_lpg gave up (slow-path budget), on the same stream right behind it -- beside
                        // k_chain_hw's tail.  Usually none: then this is an empty launch.  Their number only exists on the
                        // device: as many wavefronts as the chip holds stride the list.
                        HwChainArgs hr = ha;
                        hr.list = la.redo_list; hr.n_list = 0; hr.prio = 0;
                        const u32 redo_grid = (u32)std::min<u64>(((u64)la.n_list + 1) / 2, (u64)ctx->n_cu * 32);
                        hipLaunchKernelGGL(k_chain_hw_redo, dim3(std::max<u32>(redo_grid, 1)), dim3(64), 0, both ? ctx->stream2 : ctx->stream, hr, cp, go, la.redo_count);
                        KCHK(ctx);
                        if (ctx->opt("VERBOSE")) {
                            u32 nr = 0;
                            HIPCHK(ctx, hipMemcpyAsync(&nr, la.redo_count, 4, hipMemcpyDeviceToHost, both ? ctx->stream2 : ctx->stream));
                            HIPCHK(ctx, hipStreamSynchronize(both ? ctx->stream2 : ctx->stream));
                            fprintf(stderr, "[lrge_hip] k_chain_lpg handed %u of %u groups to k_chain_hw_redo\n", nr, la.n_list);
                        }
                    }
                }
                if (n_big) {
                    hipLaunchKernelGGL(k_chain_hw, dim3((n_big + 1) / 2), dim3(64), 0, ctx->stream, ha, cp, go);
                    KCHK(ctx);
                    ctx->counters[LRGE_C_CHAIN_LAUNCHES] += 1;
                }
                if (both) {
                    HIPCHK(ctx, hipEventRecord(ctx->ev_join, ctx->stream2));
                    HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
                }
                t.stop();
                ctx->counters[LRGE_C_CHAIN_ANCHORS] += a_chained;
                ctx->counters[LRGE_C_GROUPS_CHAINED] += n_chained;
            }
        }
    }
    {
        StageTimer t(ctx, LRGE_T_COUNT);
        CountParams cnp; cnp.kl = kl; cnp.q0 = q0; cnp.mode = job.mode;
        cnp.q_rank = Q->has_rank ? Q->d_rank : nullptr; cnp.t_rank = T->has_rank ? T->d_rank : nullptr;
        cnp.t_dup = T->dup_rank ? 1 : 0;
        cnp.q_map = d_qmap; cnp.rid_base = job.rid_base;
        if (n_chained) {
            hipLaunchKernelGGL(k_count, dim3((u32)div_up(n_chained, 256)), dim3(256), 0, ctx->stream, skey, gstart, gflags, hw_list, n_chained, cnp, d_counts, d_hasmap);
            KCHK(ctx);
        }
        // (no sync: everything runs in order on ctx->stream; scratch is recycled in stream order)
        t.stop();
    }
    return LRGE_OK;
}

int OverlapRun::finish() {
    const lrge_hip_seqset *T = ix->seqs; const Preset &P = ix->P; const u32 nq = Q->n, nt = T->n;
    (void)T; (void)P; (void)nq; (void)nt;
    if (job.counts) HIPCHK(ctx, ctx->d2h(job.counts, d_counts, (size_t)n_out * 4, ctx->stream));
    if (job.has_map) HIPCHK(ctx, ctx->d2h(job.has_map, d_hasmap, (size_t)nq * 4, ctx->stream));
    HIPCHK(ctx, ctx->d2h_sync(ctx->stream));
    if (job.n_chains) {
        unsigned long long nchn = 0;
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));   // blocking copies below run on the null stream
        HIPCHK(ctx, hipMemcpy(&nchn, d_nchains, 8, hipMemcpyDeviceToHost));
        *job.n_chains = nchn;
        u64 m = nchn < job.chain_cap ? nchn : job.chain_cap;
        if (m && job.chains) HIPCHK(ctx, hipMemcpy(job.chains, d_chains, m * sizeof(lrge_hip_chain), hipMemcpyDeviceToHost));
    }
    if (job.an && job.dump_anchors) *job.an = 0;
    return LRGE_OK;
}

static int run_overlap(lrge_hip_ctx *ctx, const lrge_hip_index *ix, const lrge_hip_seqset *Q, OverlapJob &job) {

    HIPCHK(ctx, hipSetDevice(ctx->device));
    ctx->pin_items.clear(); ctx->pin_used = 0;      // reads an earlier, failed call may have left queued
    ctx->resolve_timers();
    memset(ctx->ms, 0, sizeof(ctx->ms));
    memset(ctx->counters, 0, sizeof(ctx->counters));
    if (Q->has_empty && !job.dump_anchors) {  // aligner.rs:214-216 -> LrgeError::MapError aborts the run
        LRGE_SET_ERR(ctx, "Error mapping read: Sequence is empty");
        return LRGE_ERR_MAP;
    }
    { int rrc = seqset_ready(ctx, Q); if (rrc) return rrc; rrc = seqset_ready(ctx, ix->seqs); if (rrc) return rrc; }
    StageTimer t_total(ctx, LRGE_T_TOTAL);
    OverlapRun R(ctx, ix, Q, job);
    auto done = [&](int rc) -> int {            // common exit: total time, drain the stream, resolve the stage timers
        if (rc == RUN_DONE) rc = LRGE_OK;
        t_total.stop();
        const hipError_t e = hipStreamSynchronize(ctx->stream);
        ctx->resolve_timers();
        pool_report(ctx, "run_overlap");
        if (rc == LRGE_OK && e != hipSuccess) { LRGE_SET_ERR(ctx, "stream: %s", hipGetErrorString(e)); return LRGE_ERR_DEVICE; }
        return rc;
    };
    int rc = R.prepare();
    if (rc) return done(rc);
    rc = R.seeds();
    if (rc) return done(rc);
    rc = R.plan();
    if (rc) return done(rc);
    const u32 nq = Q->n;
    u32 q0 = job.dump_anchors ? job.dump_query : 0;
    const u32 q_end = job.dump_anchors ? job.dump_query + 1 : nq;
    while (q0 < q_end) {
        u32 q1 = q0; u64 A = 0;
        while (q1 < q_end && (q1 - q0) < (1u << std::min<u32>(R.max_bits_q, 24)) && (q1 == q0 || A + R.h_qtot[q1] <= R.batch_cap)) { A += R.h_qtot[q1]; ++q1; }
        if (A >= (1ULL << 32)) { LRGE_SET_ERR(ctx, "query %u alone yields %llu anchors (limit 2^32)", q0, (unsigned long long)A); return done(LRGE_ERR_TOO_MANY); }
        ctx->counters[LRGE_C_BATCHES] += 1;
        R.kl.bits_q = std::max<u32>(1, ceil_log2_u64((u64)(q1 - q0)));
        R.cp.kl = R.kl; R.cp.q0 = q0;
        rc = R.batch(q0, q1, A);
        if (rc) return done(rc);
        q0 = q1;
    }
    return done(R.finish());
}

static int check_common(lrge_hip_ctx *ctx, const lrge_hip_index *ix, const lrge_hip_seqset *q, bool parts_ok = false) {
    if (!ctx) return LRGE_ERR_INVALID;
    if (!ix) { LRGE_SET_ERR(ctx, "No index"); return LRGE_ERR_MAP; }   // aligner.rs:210-212
    if (!q) { LRGE_SET_ERR(ctx, "null read set"); return LRGE_ERR_INVALID; }
    if (ix->ctx != ctx || q->ctx != ctx) { LRGE_SET_ERR(ctx, "index / read set belong to another context"); return LRGE_ERR_INVALID; }
    if (ix->restrict_set && q->uid != ix->restrict_uid && q->parent_uid != ix->restrict_uid) {
        LRGE_SET_ERR(ctx, "this index was built for one streamed set (lrge_hip_index_build_for): only that set may be streamed against it");
        return LRGE_ERR_INVALID;
    }
    if (!ix->parts.empty() && !parts_ok) {
        LRGE_SET_ERR(ctx, "the index is partitioned (%zu parts, target set above PART_BASES bases): this entry point is not implemented for it", ix->parts.size());
        return LRGE_ERR_TOO_MANY;
    }
    return LRGE_OK;
}

// A streamed set above LRGE_HIP_STREAM_BASES bases (default 4e9: < 2^32 minimizers per pass) goes through in views of
// at most that many bases.  The streamed reads are independent of each other (twoset.rs:266-334, :485-565), so the passes
// simply follow one another: per-read outputs land at the view's offset, per-indexed-read counts add up.
static u64 stream_limit(const lrge_hip_ctx *ctx) { return ctx->opt_u64("STREAM_BASES", 4000000000ull); }
static std::vector<u32> stream_cuts(const lrge_hip_seqset *s) {
    std::vector<u32> cuts{0};
    const u64 lim = stream_limit(s->ctx);
    u64 acc = 0;
    for (u32 r = 0; r < s->n; ++r) {
        if (acc && acc + s->h_len[r] > lim) { cuts.push_back(r); acc = 0; }
        acc += s->h_len[r];
    }
    cuts.push_back(s->n);
    return cuts;
}
struct StageAcc {      // timings / counters of a call made of several passes
    float ms[LRGE_T_N]; u64 cn[LRGE_C_N];
    StageAcc() { memset(ms, 0, sizeof ms); memset(cn, 0, sizeof cn); }
    void add(const lrge_hip_ctx *ctx) {
        for (int i = 0; i < LRGE_T_N; ++i) ms[i] += ctx->ms[i];
        for (int i = 0; i < LRGE_C_N; ++i) cn[i] = i == LRGE_C_LPG_SPLIT ? ctx->counters[i] : cn[i] + ctx->counters[i];
    }
    void store(lrge_hip_ctx *ctx) const { memcpy(ctx->ms, ms, sizeof ms); memcpy(ctx->counters, cn, sizeof cn); }
};

// two-set forward against one (unpartitioned) index, the queries in views if there are too many of them
static int twoset_one_index(lrge_hip_ctx *ctx, const lrge_hip_index *ix, const lrge_hip_seqset *queries, const OverlapJob &job, StageAcc &acc) {
    if (queries->total_bases <= stream_limit(ctx) || queries->n < 2) {
        OverlapJob j = job;
        int rc = run_overlap(ctx, ix, queries, j);
        acc.add(ctx);
        return rc;
    }
    const std::vector<u32> cuts = stream_cuts(queries);
    for (size_t v = 0; v + 1 < cuts.size(); ++v) {
        lrge_hip_seqset *view = nullptr;
        int rc = seqset_view(ctx, queries, cuts[v], cuts[v + 1], &view);
        if (rc) return rc;
        OverlapJob j = job;
        if (j.counts) j.counts += cuts[v];
        if (j.has_map) j.has_map += cuts[v];
        rc = run_overlap(ctx, ix, view, j);
        acc.add(ctx);
        lrge_hip_seqset_free(view);
        if (rc) return rc;
    }
    return LRGE_OK;
}

extern "C" int lrge_hip_overlap_twoset(lrge_hip_ctx *ctx, const lrge_hip_index *ix, const lrge_hip_seqset *queries,
                                       const lrge_hip_params *p, uint32_t *counts, uint32_t *has_mapping) {
    int rc = check_common(ctx, ix, queries, /*parts_ok=*/true);
    if (rc) return rc;
    OverlapJob job; job.mode = MODE_TWOSET; job.dual = 1;
    job.prm = p ? *p : lrge_hip_params{0, 0.2f};
    job.counts = counts; job.has_map = has_mapping;
    StageAcc acc;
    if (ix->parts.empty()) {
        rc = twoset_one_index(ctx, ix, queries, job, acc);
        acc.store(ctx);
        return rc;
    }
    // partitioned index: the parts hold disjoint target reads, so a query's distinct-target count is the sum over the
    // parts and it has a mapping if it has one in any part; every part sees the same queries and the global mid_occ
    const u32 nq = queries->n;
    std::vector<u32> c((size_t)nq + 1), h((size_t)nq + 1);
    if (counts) std::fill(counts, counts + nq, 0u);
    if (has_mapping) std::fill(has_mapping, has_mapping + nq, 0u);
    SketchCache qcache;
    const bool cache_ok = queries->total_bases <= stream_limit(ctx) || queries->n < 2;     // (in views every view is sketched per part)
    for (const lrge_hip_index *part : ix->parts) {
        OverlapJob pj = job;
        if (cache_ok) pj.qcache = &qcache;
        pj.counts = c.data(); pj.has_map = h.data();
        rc = twoset_one_index(ctx, part, queries, pj, acc);
        if (rc) return rc;
        for (u32 q = 0; q < nq; ++q) { if (counts) counts[q] += c[q]; if (has_mapping) has_mapping[q] |= h[q]; }
    }
    acc.store(ctx);
    return LRGE_OK;
}

// inverse against one (unpartitioned) index, the streamed set in views if it is too large; counts has ix->seqs->n entries
static int inverse_one_index(lrge_hip_ctx *ctx, const lrge_hip_index *ix, const lrge_hip_seqset *streamed, const OverlapJob &job, uint32_t *counts,
                             StageAcc &acc) {
    if (streamed->total_bases <= stream_limit(ctx) || streamed->n < 2) {
        OverlapJob j = job;
        j.counts = counts;
        int rc = run_overlap(ctx, ix, streamed, j);
        acc.add(ctx);
        return rc;
    }
    // the streamed (target) set in views: every streamed read adds one to the indexed reads it hits (twoset.rs:520-523)
    const u32 n_ix = ix->seqs->n;
    std::vector<u32> c((size_t)n_ix + 1);
    if (counts) std::fill(counts, counts + n_ix, 0u);
    const std::vector<u32> cuts = stream_cuts(streamed);
    for (size_t v = 0; v + 1 < cuts.size(); ++v) {
        lrge_hip_seqset *view = nullptr;
        int rc = seqset_view(ctx, streamed, cuts[v], cuts[v + 1], &view);
        if (rc) return rc;
        OverlapJob j = job;
        j.counts = c.data();
        rc = run_overlap(ctx, ix, view, j);
        acc.add(ctx);
        lrge_hip_seqset_free(view);
        if (rc) return rc;
        if (counts) for (u32 i = 0; i < n_ix; ++i) counts[i] += c[i];
    }
    return LRGE_OK;
}

extern "C" int lrge_hip_overlap_inverse(lrge_hip_ctx *ctx, const lrge_hip_index *ix, const lrge_hip_seqset *streamed,
                                        const lrge_hip_params *p, uint32_t *counts) {
    int rc = check_common(ctx, ix, streamed, /*parts_ok=*/true);
    if (rc) return rc;
    if (ix->seqs->dup_rank) { LRGE_SET_ERR(ctx, "Duplicate read identifier in the indexed set"); return LRGE_ERR_DUPLICATE_ID; }
    OverlapJob job; job.mode = MODE_INVERSE; job.dual = 1;
    job.prm = p ? *p : lrge_hip_params{0, 0.2f};
    StageAcc acc;
    if (ix->parts.empty()) {
        rc = inverse_one_index(ctx, ix, streamed, job, counts, acc);
        acc.store(ctx);
        return rc;
    }
    // partitioned index: the parts hold disjoint indexed reads, every part sees all streamed reads and the global mid_occ --
    // a part's counts are the counts of its reads
    for (size_t pi = 0; pi < ix->parts.size(); ++pi) {
        rc = inverse_one_index(ctx, ix->parts[pi], streamed, job, counts ? counts + ix->part_r0[pi] : nullptr, acc);
        if (rc) return rc;
    }
    acc.store(ctx);
    return LRGE_OK;
}

extern "C" int lrge_hip_overlap_ava(lrge_hip_ctx *ctx, const lrge_hip_index *ix, const lrge_hip_seqset *reads,
                                    const lrge_hip_params *p, uint32_t *counts) {
    int rc = check_common(ctx, ix, reads, /*parts_ok=*/true);
    if (rc) return rc;
    if (ix->seqs != reads && !(ix->seqs->has_rank && reads->has_rank)) {
        LRGE_SET_ERR(ctx, "all-vs-all over a shard of the reads needs name ranks on both sets"); return LRGE_ERR_INVALID;
    }
    if (reads->dup_rank || ix->seqs->dup_rank) { LRGE_SET_ERR(ctx, "Duplicate read identifier"); return LRGE_ERR_DUPLICATE_ID; }
    OverlapJob job; job.mode = MODE_AVA; job.dual = 0;
    job.prm = p ? *p : lrge_hip_params{0, 0.2f};
    job.counts = counts;
    const bool in_views = reads->total_bases > stream_limit(ctx) && reads->n >= 2;
    if (ix->parts.empty() && !in_views) return run_overlap(ctx, ix, reads, job);
    // A partitioned index: every part sees all reads as queries; a pair is found in the part that holds its larger-named
    // read (NO_DUAL), and both of its counts live in the one vector keyed by the whole set.  A read set above STREAM_BASES
    // bases (ava.rs:165-366 has no such limit) goes through in views like the streamed set of the two-set strategies: a view
    // is a shard of the reads, and the shards' contributions add up (see the header).
    if (!(ix->seqs->has_rank && reads->has_rank)) { LRGE_SET_ERR(ctx, "all-vs-all against a partitioned index / over more than STREAM_BASES bases needs name ranks"); return LRGE_ERR_INVALID; }
    const u32 n_all = ix->seqs->n;
    std::vector<u32> c((size_t)n_all + 1);
    if (counts) std::fill(counts, counts + n_all, 0u);
    StageAcc acc;
    const std::vector<u32> cuts = in_views ? stream_cuts(reads) : std::vector<u32>{0, reads->n};
    const size_t n_parts = ix->parts.empty() ? 1 : ix->parts.size();
    for (size_t v = 0; v + 1 < cuts.size(); ++v) {
        lrge_hip_seqset *view = nullptr;
        if (in_views) { rc = seqset_view(ctx, reads, cuts[v], cuts[v + 1], &view); if (rc) return rc; }
        for (size_t pi = 0; pi < n_parts; ++pi) {
            OverlapJob j = job;
            j.counts = c.data(); j.indexed_top = ix->seqs;
            if (!ix->parts.empty()) j.rid_base = ix->part_r0[pi];
            rc = run_overlap(ctx, ix->parts.empty() ? ix : ix->parts[pi], in_views ? view : reads, j);
            acc.add(ctx);
            if (rc) break;
            if (counts) for (u32 i = 0; i < n_all; ++i) counts[i] += c[i];
        }
        if (view) lrge_hip_seqset_free(view);
        if (rc) return rc;
    }
    acc.store(ctx);
    return LRGE_OK;
}

extern "C" int lrge_hip_chains(lrge_hip_ctx *ctx, const lrge_hip_index *ix, const lrge_hip_seqset *queries, int dual,
                               lrge_hip_chain *out, uint64_t cap, uint64_t *n_out) {
    int rc = check_common(ctx, ix, queries, /*parts_ok=*/true);
    if (rc) return rc;
    if (!n_out) return LRGE_ERR_INVALID;
    OverlapJob job; job.mode = MODE_TWOSET; job.dual = dual ? 1 : 0;
    job.prm = lrge_hip_params{0, 0.2f};
    if (ix->parts.empty()) {
        job.chains = out; job.chain_cap = out ? cap : 0; job.n_chains = n_out;
        return run_overlap(ctx, ix, queries, job);
    }
    // partitioned index: the chains of a query onto the reads of one part are found in that part; records carry the
    // read's index in the whole set (rid_base).  n_seeds spans the query's KEPT seeds, and kept is a property of the
    // whole index: a first sweep over the parts sums every query minimizer's occurrence count (k_hc_accumulate)
    if (queries->total_bases + 1 >= (1ULL << 32)) { LRGE_SET_ERR(ctx, "chains against a partitioned index: query set too large"); return LRGE_ERR_TOO_MANY; }
    Scratch sc(ctx);
    ALLOC_OR_FAIL(d_acc, sc, u32, (size_t)queries->total_bases + 1);
    HIPCHK(ctx, hipMemsetAsync(d_acc, 0, ((size_t)queries->total_bases + 1) * 4, ctx->stream));
    for (size_t pi = 0; pi < ix->parts.size(); ++pi) {
        OverlapJob j = job;
        j.paf_stats = true; j.d_hc_acc = d_acc; j.hc_last = false;      // (accumulate only)
        rc = run_overlap(ctx, ix->parts[pi], queries, j);
        if (rc) return rc;
    }
    u64 total = 0;
    StageAcc acc;
    for (size_t pi = 0; pi < ix->parts.size(); ++pi) {
        OverlapJob j = job;
        j.d_hc_global = d_acc;
        u64 n_part = 0;
        const u64 room = (out && cap > total) ? cap - total : 0;
        j.chains = room ? out + total : nullptr; j.chain_cap = room; j.n_chains = &n_part; j.rid_base = ix->part_r0[pi];
        rc = run_overlap(ctx, ix->parts[pi], queries, j);
        acc.add(ctx);
        if (rc) return rc;
        total += n_part;
    }
    acc.store(ctx);
    *n_out = total;
    return LRGE_OK;
}

extern "C" int lrge_hip_paf_stats(lrge_hip_ctx *ctx, const lrge_hip_index *ix, const lrge_hip_seqset *queries, int32_t *rep_len,
                                  uint64_t *sum_span, uint32_t *n_kept) {
    int rc = check_common(ctx, ix, queries, /*parts_ok=*/true);
    if (rc) return rc;
    if (!rep_len || !sum_span || !n_kept) return LRGE_ERR_INVALID;
    OverlapJob job; job.mode = MODE_TWOSET; job.dual = 1;
    job.prm = lrge_hip_params{0, 0.2f};
    job.paf_stats = true; job.rep_len = rep_len; job.sum_span = sum_span; job.n_kept = n_kept;
    if (queries->n == 0) return LRGE_OK;
    if (ix->parts.empty()) return run_overlap(ctx, ix, queries, job);
    // partitioned index: a seed is kept / repetitive by its occurrence count over ALL parts (k_hc_accumulate); the last
    // part's pass turns the accumulated counts into rl / avg_k
    if (queries->total_bases + 1 >= (1ULL << 32)) { LRGE_SET_ERR(ctx, "paf_stats against a partitioned index: query set too large"); return LRGE_ERR_TOO_MANY; }
    Scratch sc(ctx);
    ALLOC_OR_FAIL(d_acc, sc, u32, (size_t)queries->total_bases + 1);      // (one minimizer per base at most)
    HIPCHK(ctx, hipMemsetAsync(d_acc, 0, ((size_t)queries->total_bases + 1) * 4, ctx->stream));
    for (size_t pi = 0; pi < ix->parts.size(); ++pi) {
        OverlapJob j = job;
        j.d_hc_acc = d_acc; j.hc_last = pi + 1 == ix->parts.size();
        rc = run_overlap(ctx, ix->parts[pi], queries, j);
        if (rc) return rc;
    }
    return LRGE_OK;
}

extern "C" int lrge_hip_anchors_dump(lrge_hip_ctx *ctx, const lrge_hip_index *ix, const lrge_hip_seqset *queries, int dual,
                                     uint32_t query, uint64_t *x, uint64_t *y, uint64_t cap, uint64_t *n_out) {
    int rc = check_common(ctx, ix, queries);
    if (rc) return rc;
    if (!n_out || query >= queries->n) return LRGE_ERR_INVALID;
    OverlapJob job; job.mode = MODE_TWOSET; job.dual = dual ? 1 : 0;
    job.prm = lrge_hip_params{0, 0.2f};
    job.dump_anchors = true; job.dump_query = query; job.ax = x; job.ay = y; job.acap = (x && y) ? cap : 0; job.an = n_out;
    *n_out = 0;
    return run_overlap(ctx, ix, queries, job);
}

// ------------------------------------------------------------------------------------------
// communicators (comm.h)
// ------------------------------------------------------------------------------------------
extern "C" int lrge_hip_comm_unique_id(void *id128) {
    if (!id128) return LRGE_ERR_INVALID;
    std::lock_guard<std::mutex> g(g_rccl_mu);
    if (!g_rccl.load()) { g_last_error = g_rccl.err; return LRGE_ERR_DEVICE; }
    lrge_ncclUniqueId id;
    const int r = g_rccl.GetUniqueId(&id);
    if (r != 0) { g_last_error = std::string("ncclGetUniqueId: ") + g_rccl.GetErrorString(r); return LRGE_ERR_DEVICE; }
    memcpy(id128, id.internal, 128);
    return LRGE_OK;
}

extern "C" int lrge_hip_comm_create(lrge_hip_ctx *ctx, int rank, int world, const void *id128, lrge_hip_comm **out) {
    if (!ctx || !out || world < 1 || rank < 0 || rank >= world || !id128) return LRGE_ERR_INVALID;
    *out = nullptr;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    { std::lock_guard<std::mutex> g(g_rccl_mu); if (!g_rccl.load()) { LRGE_SET_ERR(ctx, "%s", g_rccl.err.c_str()); return LRGE_ERR_DEVICE; } }
    lrge_ncclUniqueId id;
    memcpy(id.internal, id128, 128);
    std::unique_ptr<lrge_hip_comm> c(new lrge_hip_comm());
    c->ctx = ctx; c->rank = rank; c->world = world;
    NCCLCHK(ctx, g_rccl.CommInitRank(&c->nccl, world, id, rank));
    *out = c.release();
    return LRGE_OK;
}

extern "C" int lrge_hip_comm_local_group_create(int world, void **grp) {
    if (!grp || world < 1) return LRGE_ERR_INVALID;
    *grp = new LocalGroup(world);
    return LRGE_OK;
}
extern "C" void lrge_hip_comm_local_group_destroy(void *grp) { delete (LocalGroup *)grp; }

extern "C" int lrge_hip_comm_create_local(lrge_hip_ctx *ctx, int rank, void *grp, lrge_hip_comm **out) {
    LocalGroup *g = (LocalGroup *)grp;
    if (!ctx || !out || !g || rank < 0 || rank >= g->world) return LRGE_ERR_INVALID;
    lrge_hip_comm *c = new lrge_hip_comm();
    c->ctx = ctx; c->rank = rank; c->world = g->world; c->grp = g;
    *out = c;
    return LRGE_OK;
}

extern "C" int lrge_hip_comm_create_host(lrge_hip_ctx *ctx, int rank, int world, lrge_hip_host_allreduce_fn allreduce,
                                         lrge_hip_host_allgather_fn allgather, void *user, lrge_hip_comm **out) {
    if (!ctx || !out || world < 1 || rank < 0 || rank >= world || !allreduce || !allgather) return LRGE_ERR_INVALID;
    lrge_hip_comm *c = new lrge_hip_comm();
    c->ctx = ctx; c->rank = rank; c->world = world; c->cb_allreduce = allreduce; c->cb_allgather = allgather; c->cb_user = user;
    *out = c;
    return LRGE_OK;
}

extern "C" void lrge_hip_comm_destroy(lrge_hip_comm *c) {
    if (!c) return;
    if (c->nccl) {
        // (a communicator that outlives its context -- as lrge_hip_index_free / _seqset_free tolerate too -- must not touch it)
        bool ctx_alive;
        { std::lock_guard<std::mutex> g(g_live_mu); ctx_alive = g_live_ctx.count(c->ctx) != 0; }
        if (ctx_alive) { (void)hipSetDevice(c->ctx->device); (void)hipStreamSynchronize(c->ctx->stream); }
        (void)g_rccl.CommDestroy(c->nccl);
    }
    delete c;
}
extern "C" int lrge_hip_comm_rank(const lrge_hip_comm *c) { return c ? c->rank : -1; }
extern "C" int lrge_hip_comm_world(const lrge_hip_comm *c) { return c ? c->world : 0; }

// how many ranks RCCL itself sees in this communicator (ncclCommCount): 0 for the local / host transports
extern "C" int lrge_hip_comm_rccl_ranks(const lrge_hip_comm *c, int *n) {
    if (!c || !n) return LRGE_ERR_INVALID;
    *n = 0;
    if (!c->nccl) return LRGE_OK;
    if (g_rccl.CommCount(c->nccl, n) != 0) { *n = 0; return LRGE_ERR_DEVICE; }
    return LRGE_OK;
}
extern "C" int lrge_hip_comm_local_group_serialize(void *grp, int on) {
    if (!grp) return LRGE_ERR_INVALID;
    ((LocalGroup *)grp)->serialize = on != 0;
    return LRGE_OK;
}
extern "C" int lrge_hip_comm_local_turn(lrge_hip_comm *c, int begin) {
    if (!c) return LRGE_ERR_INVALID;
    comm_turn(c, begin != 0);
    return LRGE_OK;
}
extern "C" double lrge_hip_comm_busy_ms(lrge_hip_comm *c, int reset) {
    if (!c) return 0.0;
    const double v = c->busy_ms;
    if (reset) c->busy_ms = 0;
    return v;
}

// host-buffer form of the variable-size all-to-all (the library itself uses the device form inside lrge_hip_index_build_sharded)
extern "C" int lrge_hip_comm_alltoallv(lrge_hip_comm *c, const void *send, const uint64_t *send_off, void *recv, const uint64_t *recv_off,
                                       size_t elem_bytes) {
    if (!c || !send_off || !recv_off || elem_bytes == 0) return LRGE_ERR_INVALID;
    lrge_hip_ctx *ctx = c->ctx;
    const int W = c->world;
    const u64 ns = send_off[W], nr = recv_off[W];
    if ((ns && !send) || (nr && !recv)) return LRGE_ERR_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    Scratch sc(ctx);
    char *ds = sc.get<char>(ns * elem_bytes + 1), *dr = sc.get<char>(nr * elem_bytes + 1);
    int rc = comm_agree(c, (ds && dr) ? LRGE_OK : LRGE_ERR_DEVICE, ctx->stream);
    if (rc) return rc;
    if (ns) HIPCHK(ctx, hipMemcpyAsync(ds, send, ns * elem_bytes, hipMemcpyHostToDevice, ctx->stream));
    rc = comm_alltoallv(c, ds, send_off, dr, recv_off, elem_bytes, ctx->stream);
    if (rc) return rc;
    if (nr) HIPCHK(ctx, hipMemcpyAsync(recv, dr, nr * elem_bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return LRGE_OK;
}

// host-buffer forms of the two collectives that close a step
extern "C" int lrge_hip_comm_allreduce_u32(lrge_hip_comm *c, uint32_t *inout, size_t n) {
    if (!c || (n && !inout)) return LRGE_ERR_INVALID;
    lrge_hip_ctx *ctx = c->ctx;
    if (c->world == 1 || n == 0) return LRGE_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    Scratch sc(ctx);
    ALLOC_OR_FAIL(d, sc, u32, n);
    HIPCHK(ctx, hipMemcpyAsync(d, inout, n * 4, hipMemcpyHostToDevice, ctx->stream));
    int rc = comm_allreduce_sum(c, d, n, 4, ctx->stream);
    if (rc) return rc;
    HIPCHK(ctx, hipMemcpyAsync(inout, d, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return LRGE_OK;
}

extern "C" int lrge_hip_comm_allgather(lrge_hip_comm *c, const void *send, size_t bytes, void *recv) {
    if (!c || (bytes && (!send || !recv))) return LRGE_ERR_INVALID;
    lrge_hip_ctx *ctx = c->ctx;
    if (bytes == 0) return LRGE_OK;
    if (c->world == 1) { memcpy(recv, send, bytes); return LRGE_OK; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    Scratch sc(ctx);
    ALLOC_OR_FAIL(ds, sc, char, bytes);
    ALLOC_OR_FAIL(dr, sc, char, bytes * (size_t)c->world);
    HIPCHK(ctx, hipMemcpyAsync(ds, send, bytes, hipMemcpyHostToDevice, ctx->stream));
    int rc = comm_allgather(c, ds, bytes, dr, ctx->stream);
    if (rc) return rc;
    HIPCHK(ctx, hipMemcpyAsync(recv, dr, bytes * (size_t)c->world, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return LRGE_OK;
}

// ------------------------------------------------------------------------------------------
// estimates
// ------------------------------------------------------------------------------------------
extern "C" int lrge_hip_estimates(lrge_hip_ctx *ctx, const uint32_t *counts, const uint32_t *read_lens, uint32_t n,
                                  float avg_target_len, uint64_t n_target_reads, uint32_t overlap_thresh, float *out) {
    if (!ctx || (n && (!counts || !read_lens || !out))) return LRGE_ERR_INVALID;
    if (n == 0) return LRGE_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    ctx->pin_items.clear(); ctx->pin_used = 0;      // reads an earlier, failed call may have left queued
    Scratch sc(ctx);
    ALLOC_OR_FAIL(dc, sc, u32, n); ALLOC_OR_FAIL(dl, sc, u32, n); ALLOC_OR_FAIL(d_out, sc, float, n);
    HIPCHK(ctx, hipMemcpyAsync(dc, counts, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(dl, read_lens, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    // `n_target_reads as f32`, `2.0 * ovlap_thresh as f32` (estimate.rs:153-156)
    float nt = (float)n_target_reads, two_thr = 2.0f * (float)overlap_thresh;
    hipLaunchKernelGGL(k_estimate, dim3((u32)div_up(n, 256)), dim3(256), 0, ctx->stream, dc, dl, n, avg_target_len, nt, two_thr, d_out);
    KCHK(ctx);
    HIPCHK(ctx, ctx->d2h(out, d_out, (size_t)n * 4, ctx->stream));
    HIPCHK(ctx, ctx->d2h_sync(ctx->stream));
    return LRGE_OK;
}

// estimate.rs:80-132.  f32 arithmetic, no contraction (this TU is built with -ffp-contract=off).  `d` must hold the
// order statistics idx and idx + 1 at their sorted positions (the caller selects them; a full sort is not needed).
static size_t quantile_index(size_t n, float q) {
    volatile float pos = q * (float)(n - 1);
    return (size_t)floorf(pos);
}
static bool quantile_f32(const std::vector<float> &d, float q, float *out) {
    if (d.empty()) return false;
    size_t n = d.size();
    volatile float pos = q * (float)(n - 1);
    size_t idx = (size_t)floorf(pos);
    volatile float frac = pos - (float)idx;
    if (idx + 1 < n) {
        volatile float lo = d[idx] * (1.0f - frac);
        volatile float hi = d[idx + 1] * frac;
        *out = lo + hi;
    } else *out = d[idx];
    return true;
}

extern "C" int lrge_hip_median(const float *estimates, uint64_t n, int finite, int has_lower, float lower_q, int has_upper,
                               float upper_q, float out[3], int ok[3]) {
    if (!out || !ok || (n && !estimates)) return LRGE_ERR_INVALID;
    ok[0] = ok[1] = ok[2] = 0; out[0] = out[1] = out[2] = 0.f;
    if (!has_lower && has_upper) return LRGE_ERR_INVALID;  // the reference panics here (estimate.rs:109)
    if ((has_lower && !(lower_q >= 0.f && lower_q <= 1.f)) || (has_upper && !(upper_q >= 0.f && upper_q <= 1.f)))
        return LRGE_ERR_INVALID;                           // "Quantile must be between 0.0 and 1.0"
    std::vector<float> v;
    v.reserve(n);
    // kept values, and on the way a histogram over the top bits of their patterns: for non-negative floats the bit pattern
    // orders like the value, so the bin that holds an order statistic is known after one pass
    constexpr int kShift = 17, kBins = 1 << (31 - kShift);
    std::vector<u32> hist((size_t)kBins + 1, 0);
    bool radix_ok = true;
    v.resize(n);
    size_t nv = 0;
    for (u64 i = 0; i < n; ++i) {
        const float e = estimates[i];
        u32 b; memcpy(&b, &e, 4);
        if (finite && (b & 0x7F800000u) == 0x7F800000u) continue;      // infinity or NaN
        if ((b >> 31) || e != e) radix_ok = false; else ++hist[b >> kShift];
        v[nv++] = e;
    }
    v.resize(nv);
    if (v.empty()) return LRGE_OK;
    // the reference sorts the whole vector (estimate.rs:90-95); only the (at most six) order statistics the three
    // quantiles read are needed, and an order statistic does not depend on how ties are arranged
    std::vector<size_t> need;
    auto want = [&](float q) { const size_t i = quantile_index(v.size(), q); need.push_back(i); if (i + 1 < v.size()) need.push_back(i + 1); };
    want(0.5f);
    if (has_lower) want(lower_q);
    if (has_upper) want(upper_q);
    std::sort(need.begin(), need.end());
    need.erase(std::unique(need.begin(), need.end()), need.end());
    if (radix_ok) {
        // gather the (few) bins that hold a needed rank, select inside them, and put each statistic at its index of `v`
        // (quantile_f32 below reads v[idx] and v[idx + 1] only)
        std::vector<u32> cum((size_t)kBins + 1, 0);
        for (int b = 0; b < kBins; ++b) cum[(size_t)b + 1] = cum[b] + hist[b];
        std::vector<int> bin_of(need.size());
        std::vector<int> bins;
        for (size_t k = 0; k < need.size(); ++k) {
            const int b = (int)(std::upper_bound(cum.begin(), cum.end(), (u32)need[k]) - cum.begin()) - 1;
            bin_of[k] = b;
            if (bins.empty() || bins.back() != b) bins.push_back(b);       // (need is ascending, so are the bins)
        }
        std::vector<std::vector<float>> members(bins.size());
        for (size_t t = 0; t < bins.size(); ++t) members[t].reserve(hist[bins[t]]);
        std::vector<int8_t> slot_of((size_t)kBins, (int8_t)-1);      // (16 K bins: ~200 of 50 000 clustered estimates per bin)
        for (size_t t = 0; t < bins.size(); ++t) slot_of[bins[t]] = (int8_t)t;
        for (const float e : v) {
            u32 b; memcpy(&b, &e, 4);
            const int t = slot_of[b >> kShift];
            if (t >= 0) members[(size_t)t].push_back(e);
        }
        std::vector<float> stat(need.size());
        for (size_t k = 0; k < need.size(); ++k) {
            const size_t t = (size_t)(std::find(bins.begin(), bins.end(), bin_of[k]) - bins.begin());
            std::vector<float> &m = members[t];
            const size_t r = need[k] - cum[bin_of[k]];
            std::nth_element(m.begin(), m.begin() + r, m.end());
            stat[k] = m[r];
        }
        for (size_t k = 0; k < need.size(); ++k) v[need[k]] = stat[k];
    } else {
        // negative values or NaNs (finite == 0): comparison-based selection.  The middle one of the needed order statistics
        // first, then the rest inside the halves it leaves: every later selection works on a fraction of the vector
        struct Sel {
            static void run(std::vector<float> &v, const std::vector<size_t> &need, size_t a, size_t b, size_t lo, size_t hi) {
                if (a >= b) return;
                const size_t m = (a + b) / 2, i = need[m];
                std::nth_element(v.begin() + lo, v.begin() + i, v.begin() + hi);
                run(v, need, a, m, lo, i);
                run(v, need, m + 1, b, i + 1, hi);
            }
        };
        Sel::run(v, need, 0, need.size(), 0, v.size());
    }
    ok[1] = quantile_f32(v, 0.5f, &out[1]);
    if (has_lower) ok[0] = quantile_f32(v, lower_q, &out[0]);
    if (has_upper) ok[2] = quantile_f32(v, upper_q, &out[2]);
    return LRGE_OK;
}

extern "C" int lrge_hip_unique_random_set(uint64_t k, uint32_t n, int has_seed, uint64_t seed, uint32_t *out) {
    if (k > n || (k && !out)) return LRGE_ERR_INVALID;
    std::vector<uint32_t> v = lrge::unique_random_set((size_t)k, n, has_seed ? std::optional<uint64_t>(seed) : std::nullopt);
    std::copy(v.begin(), v.end(), out);
    return LRGE_OK;
}

extern "C" int lrge_hip_chacha_block(const uint32_t key[8], uint64_t counter, int rounds, uint32_t out[16]) {
    if (!key || !out || rounds <= 0 || (rounds & 1)) return LRGE_ERR_INVALID;
    lrge::rand09::chacha_block(key, counter, 0, rounds, out);
    return LRGE_OK;
}
