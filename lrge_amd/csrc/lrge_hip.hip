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
#include "k_sketch_tile.h"
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
#include "k_tshard.h"
#include "../../include/lrge_rand.hpp"
#include "../../include/lrge_io.hpp"

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
    // POOL_TRIM: an action, not a setting -- wholly idle arena segments go back to the runtime now (a caller that is about to place
    // something large of its own beside the context: bench.py between its two clocks)
    if (!strcmp(name, "POOL_TRIM")) { ctx->opts.erase(name); if (value) { (void)hipSetDevice(ctx->device); (void)hipDeviceSynchronize(); ctx->pool.trim(); } return LRGE_OK; }
    if (!strcmp(name, "POOL_SEG_MAX_MB")) ctx->pool.seg_max = value ? (size_t)std::max<u64>(64, strtoull(value, nullptr, 10)) << 20 : (size_t)32 << 30;
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

// ---- the entry points, by family (one translation unit: the kernels of k_*.h are static / templates) ----
#include "host_seqset.inl"
#include "host_sketch.inl"
#include "host_index_collective.inl"
#include "host_index_build.inl"
#include "host_index_parts.inl"
#include "host_tshard.inl"
#include "host_qshard.inl"
#include "host_overlap_seeds.inl"
#include "host_overlap_batch.inl"
#include "host_overlap_api.inl"
#include "host_comm.inl"
#include "host_estimate.inl"
