// comm.h -- the collectives of the multi-GPU path behind the C ABI (lrge_hip_comm_*, include/lrge_hip.h).
//
// The overlap path needs exactly two exchanges: a SUM all-reduce of small integer
// vectors (the global minimizer-occurrence histogram of a sharded index build; the per-indexed-read count vectors of the
// all-vs-all and inverse strategies, ava.rs:300-301 / twoset.rs:520-523) and an all-gather of the per-read estimate
// vectors (twoset.rs:319-331's `estimates` vector).  Two transports:
//   * RCCL over xGMI, one process per GPU: ncclAllReduce / ncclAllGather on the context's stream.  librccl is bound at
//     run time (dlopen), so a single-GPU user of liblrge_hip.so needs no RCCL at all;
//   * "local": the ranks are threads of ONE process, each with its own context (its own GPU, or several contexts on one
//     GPU in the tests); buffers meet in host memory behind a barrier.  This is what a Rust host that drives the GPUs
//     from its rayon pool would use, and it lets a 1-GPU box run every rank of a world for real.
#pragma once
#include <dlfcn.h>

#include <condition_variable>
#include <mutex>

#include "internal.h"

// ---- the slice of rccl.h this file uses (/opt/rocm/include/rccl/rccl.h; bound with dlsym) ----
typedef struct { char internal[128]; } lrge_ncclUniqueId;
typedef void *lrge_ncclComm_t;
enum { LRGE_NCCL_UINT32 = 3, LRGE_NCCL_UINT64 = 5, LRGE_NCCL_FLOAT32 = 7, LRGE_NCCL_UINT8 = 1, LRGE_NCCL_SUM = 0 };
struct RcclApi {
    void *lib = nullptr;
    int (*GetUniqueId)(lrge_ncclUniqueId *) = nullptr;
    int (*CommInitRank)(lrge_ncclComm_t *, int, lrge_ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(lrge_ncclComm_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, lrge_ncclComm_t, hipStream_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, lrge_ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string err;
    bool load() {
        if (lib) return true;
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) { lib = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (lib) break; }
        if (!lib) { err = std::string("cannot load librccl: ") + dlerror(); return false; }
        bool ok = true;
        auto sym = [&](const char *s) { void *p = dlsym(lib, s); if (!p) { ok = false; err = std::string("librccl lacks ") + s; } return p; };
        GetUniqueId = (decltype(GetUniqueId))sym("ncclGetUniqueId");
        CommInitRank = (decltype(CommInitRank))sym("ncclCommInitRank");
        CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
        AllReduce = (decltype(AllReduce))sym("ncclAllReduce");
        AllGather = (decltype(AllGather))sym("ncclAllGather");
        GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
        if (!ok) { dlclose(lib); lib = nullptr; }
        return ok;
    }
};
static RcclApi g_rccl;
static std::mutex g_rccl_mu;

// ---- local transport: threads of one process ----
struct LocalGroup {
    int world;
    std::mutex mu; std::condition_variable cv;
    int arrived = 0; u64 gen = 0;
    std::vector<const void *> slot;
    explicit LocalGroup(int w) : world(w), slot((size_t)w, nullptr) {}
    void barrier() {
        std::unique_lock<std::mutex> lk(mu);
        const u64 g = gen;
        if (++arrived == world) { arrived = 0; ++gen; cv.notify_all(); }
        else cv.wait(lk, [&] { return gen != g; });
    }
};

struct lrge_hip_comm {
    lrge_hip_ctx *ctx = nullptr;
    int rank = 0, world = 1;
    lrge_ncclComm_t nccl = nullptr;      // RCCL transport
    LocalGroup *grp = nullptr;           // local transport
    // host transport: the caller's own collectives (MPI, gloo, ...) on host buffers; the library stages through the host
    lrge_hip_host_allreduce_fn cb_allreduce = nullptr; lrge_hip_host_allgather_fn cb_allgather = nullptr; void *cb_user = nullptr;
    std::vector<char> hbuf;              // host staging of the local / host transports
};

#define NCCLCHK(ctx, call)                                                                                          \
    do {                                                                                                            \
        const int _r = (call);                                                                                      \
        if (_r != 0) {                                                                                              \
            LRGE_SET_ERR(ctx, "RCCL error %s at %s:%d (%s)", g_rccl.GetErrorString ? g_rccl.GetErrorString(_r) : "?", __FILE__, __LINE__, #call); \
            return LRGE_ERR_DEVICE;                                                                                 \
        }                                                                                                           \
    } while (0)

// In-place SUM all-reduce of n elements of `esz` bytes (4: u32, 8: u64) in DEVICE memory, ordered on `st`.
static int comm_allreduce_sum(lrge_hip_comm *c, void *dbuf, size_t n, int esz, hipStream_t st) {
    lrge_hip_ctx *ctx = c->ctx;
    if (c->world == 1 || n == 0) return LRGE_OK;
    if (c->nccl) {
        NCCLCHK(ctx, g_rccl.AllReduce(dbuf, dbuf, n, esz == 8 ? LRGE_NCCL_UINT64 : LRGE_NCCL_UINT32, LRGE_NCCL_SUM, c->nccl, st));
        return LRGE_OK;
    }
    c->hbuf.resize(n * (size_t)esz);
    HIPCHK(ctx, hipMemcpyAsync(c->hbuf.data(), dbuf, n * (size_t)esz, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    if (c->cb_allreduce) {
        if (c->cb_allreduce(c->cb_user, c->hbuf.data(), n, esz) != 0) { LRGE_SET_ERR(ctx, "host communicator: all-reduce callback failed"); return LRGE_ERR_DEVICE; }
        HIPCHK(ctx, hipMemcpyAsync(dbuf, c->hbuf.data(), n * (size_t)esz, hipMemcpyHostToDevice, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
        return LRGE_OK;
    }
    LocalGroup *g = c->grp;
    g->slot[(size_t)c->rank] = c->hbuf.data();
    g->barrier();
    std::vector<char> sum(n * (size_t)esz, 0);
    for (int r = 0; r < c->world; ++r) {
        if (esz == 8) { const u64 *p = (const u64 *)g->slot[(size_t)r]; u64 *o = (u64 *)sum.data(); for (size_t i = 0; i < n; ++i) o[i] += p[i]; }
        else { const u32 *p = (const u32 *)g->slot[(size_t)r]; u32 *o = (u32 *)sum.data(); for (size_t i = 0; i < n; ++i) o[i] += p[i]; }
    }
    g->barrier();                          // everyone has read every slot: the staging buffers may change again
    HIPCHK(ctx, hipMemcpyAsync(dbuf, sum.data(), n * (size_t)esz, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipStreamSynchronize(st)); // (`sum` is a local)
    return LRGE_OK;
}

// All-gather of `bytes` bytes per rank between DEVICE buffers (recv holds world * bytes), ordered on `st`.
static int comm_allgather(lrge_hip_comm *c, const void *dsend, size_t bytes, void *drecv, hipStream_t st) {
    lrge_hip_ctx *ctx = c->ctx;
    if (bytes == 0) return LRGE_OK;
    if (c->world == 1) { HIPCHK(ctx, hipMemcpyAsync(drecv, dsend, bytes, hipMemcpyDeviceToDevice, st)); return LRGE_OK; }
    if (c->nccl) { NCCLCHK(ctx, g_rccl.AllGather(dsend, drecv, bytes, LRGE_NCCL_UINT8, c->nccl, st)); return LRGE_OK; }
    c->hbuf.resize(bytes);
    HIPCHK(ctx, hipMemcpyAsync(c->hbuf.data(), dsend, bytes, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    if (c->cb_allgather) {
        std::vector<char> all(bytes * (size_t)c->world);
        if (c->cb_allgather(c->cb_user, c->hbuf.data(), bytes, all.data()) != 0) { LRGE_SET_ERR(ctx, "host communicator: all-gather callback failed"); return LRGE_ERR_DEVICE; }
        HIPCHK(ctx, hipMemcpyAsync(drecv, all.data(), all.size(), hipMemcpyHostToDevice, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
        return LRGE_OK;
    }
    LocalGroup *g = c->grp;
    g->slot[(size_t)c->rank] = c->hbuf.data();
    g->barrier();
    std::vector<char> all(bytes * (size_t)c->world);
    for (int r = 0; r < c->world; ++r) memcpy(all.data() + bytes * (size_t)r, g->slot[(size_t)r], bytes);
    g->barrier();
    HIPCHK(ctx, hipMemcpyAsync(drecv, all.data(), all.size(), hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    return LRGE_OK;
}
