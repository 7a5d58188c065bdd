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
    int (*CommAbort)(lrge_ncclComm_t) = nullptr;        // (optional symbol)
    int (*AllReduce)(const void *, void *, size_t, int, int, lrge_ncclComm_t, hipStream_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, lrge_ncclComm_t, hipStream_t) = nullptr;
    int (*Send)(const void *, size_t, int, int, lrge_ncclComm_t, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, lrge_ncclComm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*CommCount)(const lrge_ncclComm_t, int *) = nullptr;
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
        CommAbort = (decltype(CommAbort))dlsym(lib, "ncclCommAbort");
        AllReduce = (decltype(AllReduce))sym("ncclAllReduce");
        AllGather = (decltype(AllGather))sym("ncclAllGather");
        Send = (decltype(Send))sym("ncclSend");
        Recv = (decltype(Recv))sym("ncclRecv");
        GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
        CommCount = (decltype(CommCount))sym("ncclCommCount");
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
    std::vector<const u64 *> slot2;      // all-to-all: every rank's send offsets (elements), world + 1 of them
    bool aborted = false;                // a rank failed between two barriers: everybody leaves with an error
    // Timing emulation of a world on ONE GPU: with `serialize` the ranks take turns -- a rank computes only while it holds
    // the token and hands it over whenever it waits at a barrier, so its kernels never share the GPU with another rank's
    // and the time it holds the token is the time its share of the job takes on a GPU of its own (link transfers aside).
    bool serialize = false; std::mutex token;
    // trim_on_yield: a rank that hands the GPU over gives its IDLE arena segments back to the runtime first -- eight contexts' arenas,
    // each sized as if it owned the 288 GB, do not fit one GPU; the time inside hipMalloc / hipFree (25 ms per GB: a warm arena on a
    // GPU of its own pays none of it after the first step) is kept out of the rank's busy time
    bool trim_on_yield = false;
    explicit LocalGroup(int w) : world(w), slot((size_t)w, nullptr), slot2((size_t)w, nullptr) {}
    // false: the group was aborted (by this call's peer or earlier) -- the collective fails on every rank instead of
    // leaving the others blocked on a rank that will never arrive
    bool barrier() {
        std::unique_lock<std::mutex> lk(mu);
        if (aborted) return false;
        const u64 g = gen;
        if (++arrived == world) { arrived = 0; ++gen; cv.notify_all(); }
        else cv.wait(lk, [&] { return gen != g || aborted; });
        return !aborted;
    }
    void abort() {
        std::lock_guard<std::mutex> lk(mu);
        aborted = true; arrived = 0; ++gen;
        cv.notify_all();
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
    // RCCL transport: a small device block taken ONCE at creation for the host-buffer collectives below (the status-carrying
    // vectors of a collective build), so that a rank that has just run out of memory can still JOIN a collective to say so
    char *d_small = nullptr; static constexpr size_t kSmall = (size_t)1 << 20;
    bool aborted = false;                // lrge_hip_comm_abort: every later collective of this communicator fails at once
    bool in_turn = false; double busy_ms = 0, t_acquired = 0, alloc_ms0 = 0;     // (serialized local groups)
    double wait_ms = 0;                   // wall time spent inside barriers of the local transport (waiting for the other ranks)
    // option RCCL_WORLD1 (read when an RCCL communicator is created): a world of ONE still goes through librccl for every collective --
    // ncclAllReduce / ncclAllGather of one rank, send / receive pairs with itself inside a group -- instead of the world-1 shortcuts, so
    // that a 1-GPU box executes the RCCL branches of this file (staging, groups, abort) with the shapes the sharded builds use.
    // rccl_ops counts the librccl data-path calls made (lrge_hip_comm_rccl_ops).
    bool force = false; u64 rccl_ops = 0;
    double standin_ms = 0;                // (local transport) time inside the device-to-device copies that stand in for link transfers of comm_allgatherv: part of busy_ms, reported beside it
};

static void comm_turn(lrge_hip_comm *c, bool begin) {
    if (!c || !c->grp || !c->grp->serialize) return;
    if (begin) {
        c->grp->token.lock(); c->in_turn = true;
        c->alloc_ms0 = c->ctx->pool.ms_malloc + c->ctx->pool.ms_free;
        c->t_acquired = DevPool::now_ms();
        return;
    }
    if (!c->in_turn) return;
    (void)hipStreamSynchronize(c->ctx->stream); (void)hipStreamSynchronize(c->ctx->stream2); (void)hipStreamSynchronize(c->ctx->copy_stream);
    c->busy_ms += (DevPool::now_ms() - c->t_acquired) - (c->grp->trim_on_yield ? (c->ctx->pool.ms_malloc + c->ctx->pool.ms_free) - c->alloc_ms0 : 0.0);
    if (c->grp->trim_on_yield) c->ctx->pool.trim();
    c->in_turn = false;
    c->grp->token.unlock();
}
// barrier of the local transport; a serialized group hands the GPU to another rank while this one waits
static bool grp_barrier(lrge_hip_comm *c) {
    const bool turn = c->grp->serialize && c->in_turn;
    const double t0 = DevPool::now_ms();
    if (turn) comm_turn(c, false);
    const bool ok = c->grp->barrier();
    if (turn) comm_turn(c, true);
    c->wait_ms += DevPool::now_ms() - t0;
    return ok;
}

// HIP call inside a collective of the local transport: a failure wakes the ranks waiting at the barrier
#define HIPCHK_GRP(c, call)                                                                                         \
    do {                                                                                                            \
        hipError_t _e = (call);                                                                                     \
        if (_e != hipSuccess) {                                                                                     \
            LRGE_SET_ERR((c)->ctx, "HIP error %s at %s:%d (%s)", hipGetErrorString(_e), __FILE__, __LINE__, #call); \
            if ((c)->grp) (c)->grp->abort();                                                                        \
            return LRGE_ERR_DEVICE;                                                                                 \
        }                                                                                                           \
    } while (0)

// a communicator this rank has aborted (lrge_hip_comm_abort) takes part in nothing any more
#define COMM_LIVE(c)                                                                                                \
    do { if ((c)->aborted && ((c)->world > 1 || (c)->force)) { LRGE_SET_ERR((c)->ctx, "communicator was aborted"); return LRGE_ERR_DEVICE; } } while (0)

#define NCCLCHK(ctx, call)                                                                                          \
    do {                                                                                                            \
        const int _r = (call);                                                                                      \
        if (_r != 0) {                                                                                              \
            LRGE_SET_ERR(ctx, "RCCL error %s at %s:%d (%s)", g_rccl.GetErrorString ? g_rccl.GetErrorString(_r) : "?", __FILE__, __LINE__, #call); \
            return LRGE_ERR_DEVICE;                                                                                 \
        }                                                                                                           \
    } while (0)

static inline bool comm_solo(const lrge_hip_comm *c) { return c->world == 1 && !c->force; }      // the world-1 shortcuts apply

// In-place SUM all-reduce of n elements of `esz` bytes (4: u32, 8: u64) in DEVICE memory, ordered on `st`.
static int comm_allreduce_sum(lrge_hip_comm *c, void *dbuf, size_t n, int esz, hipStream_t st) {
    lrge_hip_ctx *ctx = c->ctx;
    COMM_LIVE(c);
    if (comm_solo(c) || n == 0) return LRGE_OK;
    if (c->nccl) {
        ++c->rccl_ops;
        NCCLCHK(ctx, g_rccl.AllReduce(dbuf, dbuf, n, esz == 8 ? LRGE_NCCL_UINT64 : LRGE_NCCL_UINT32, LRGE_NCCL_SUM, c->nccl, st));
        return LRGE_OK;
    }
    c->hbuf.resize(n * (size_t)esz);
    HIPCHK_GRP(c, hipMemcpyAsync(c->hbuf.data(), dbuf, n * (size_t)esz, hipMemcpyDeviceToHost, st));
    HIPCHK_GRP(c, hipStreamSynchronize(st));
    if (c->cb_allreduce) {
        if (c->cb_allreduce(c->cb_user, c->hbuf.data(), n, esz) != 0) { LRGE_SET_ERR(ctx, "host communicator: all-reduce callback failed"); return LRGE_ERR_DEVICE; }
        HIPCHK(ctx, hipMemcpyAsync(dbuf, c->hbuf.data(), n * (size_t)esz, hipMemcpyHostToDevice, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
        return LRGE_OK;
    }
    LocalGroup *g = c->grp;
    g->slot[(size_t)c->rank] = c->hbuf.data();
    if (!grp_barrier(c)) { LRGE_SET_ERR(ctx, "local communicator: another rank failed"); return LRGE_ERR_DEVICE; }
    std::vector<char> sum(n * (size_t)esz, 0);
    for (int r = 0; r < c->world; ++r) {
        if (esz == 8) { const u64 *p = (const u64 *)g->slot[(size_t)r]; u64 *o = (u64 *)sum.data(); for (size_t i = 0; i < n; ++i) o[i] += p[i]; }
        else { const u32 *p = (const u32 *)g->slot[(size_t)r]; u32 *o = (u32 *)sum.data(); for (size_t i = 0; i < n; ++i) o[i] += p[i]; }
    }
    if (!grp_barrier(c)) { LRGE_SET_ERR(ctx, "local communicator: another rank failed"); return LRGE_ERR_DEVICE; }   // everyone has read every slot: the staging buffers may change again
    HIPCHK(ctx, hipMemcpyAsync(dbuf, sum.data(), n * (size_t)esz, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipStreamSynchronize(st)); // (`sum` is a local)
    return LRGE_OK;
}

// All-gather of `bytes` bytes per rank between DEVICE buffers (recv holds world * bytes), ordered on `st`.
static int comm_allgather(lrge_hip_comm *c, const void *dsend, size_t bytes, void *drecv, hipStream_t st) {
    lrge_hip_ctx *ctx = c->ctx;
    COMM_LIVE(c);
    if (bytes == 0) return LRGE_OK;
    if (comm_solo(c)) { HIPCHK(ctx, hipMemcpyAsync(drecv, dsend, bytes, hipMemcpyDeviceToDevice, st)); return LRGE_OK; }
    if (c->nccl) { ++c->rccl_ops; NCCLCHK(ctx, g_rccl.AllGather(dsend, drecv, bytes, LRGE_NCCL_UINT8, c->nccl, st)); return LRGE_OK; }
    if (c->grp && bytes >= ((size_t)64 << 10)) {
        // threads of one process, a large payload (the key sets of a sharded build): every rank copies the others' buffers
        // device to device instead of meeting in pageable host memory
        LocalGroup *g = c->grp;
        HIPCHK_GRP(c, hipStreamSynchronize(st));                   // my buffer is complete before anybody reads it
        g->slot[(size_t)c->rank] = dsend;
        if (!grp_barrier(c)) { LRGE_SET_ERR(ctx, "local communicator: another rank failed"); return LRGE_ERR_DEVICE; }
        for (int r = 0; r < c->world; ++r)
            HIPCHK_GRP(c, hipMemcpyAsync((char *)drecv + bytes * (size_t)r, g->slot[(size_t)r], bytes, hipMemcpyDefault, st));
        HIPCHK_GRP(c, hipStreamSynchronize(st));
        if (!grp_barrier(c)) { LRGE_SET_ERR(ctx, "local communicator: another rank failed"); return LRGE_ERR_DEVICE; }
        return LRGE_OK;
    }
    c->hbuf.resize(bytes);
    HIPCHK_GRP(c, hipMemcpyAsync(c->hbuf.data(), dsend, bytes, hipMemcpyDeviceToHost, st));
    HIPCHK_GRP(c, hipStreamSynchronize(st));
    if (c->cb_allgather) {
        std::vector<char> all(bytes * (size_t)c->world);
        if (c->cb_allgather(c->cb_user, c->hbuf.data(), bytes, all.data()) != 0) { LRGE_SET_ERR(ctx, "host communicator: all-gather callback failed"); return LRGE_ERR_DEVICE; }
        HIPCHK(ctx, hipMemcpyAsync(drecv, all.data(), all.size(), hipMemcpyHostToDevice, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
        return LRGE_OK;
    }
    LocalGroup *g = c->grp;
    g->slot[(size_t)c->rank] = c->hbuf.data();
    if (!grp_barrier(c)) { LRGE_SET_ERR(ctx, "local communicator: another rank failed"); return LRGE_ERR_DEVICE; }
    std::vector<char> all(bytes * (size_t)c->world);
    for (int r = 0; r < c->world; ++r) memcpy(all.data() + bytes * (size_t)r, g->slot[(size_t)r], bytes);
    if (!grp_barrier(c)) { LRGE_SET_ERR(ctx, "local communicator: another rank failed"); return LRGE_ERR_DEVICE; }
    HIPCHK(ctx, hipMemcpyAsync(drecv, all.data(), all.size(), hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    return LRGE_OK;
}

// ---- small collectives on HOST vectors (sizes, counts, statistics -- each carrying a status word) ----
// Same wire shape as the device-buffer forms above (RCCL: the same ncclAllReduce / ncclAllGather; host callbacks: the same callback;
// local: the same two barriers), but no allocation and, off RCCL, no device round trip: a rank whose allocation or kernel has failed
// can always take part, which is what makes failure collective (CollectiveGuard, host_index_collective.inl).
static int comm_small_stage(lrge_hip_comm *c, size_t bytes, Scratch &sc, char **d) {
    if (bytes <= lrge_hip_comm::kSmall && c->d_small) { *d = c->d_small; return LRGE_OK; }
    *d = sc.get<char>(bytes);
    return *d ? LRGE_OK : LRGE_ERR_DEVICE;
}

// RCCL transport: a rank that cannot even stage its vector cannot join the collective either -- it aborts its communicator
// (ncclCommAbort) so that nothing of this rank lingers in the group, and fails; its peers' pending operations end with their own
// processes (the launcher tears the job down when one process fails: ADVICE r04).  Untested on hardware: 1-GPU leases only.
static int comm_rccl_give_up(lrge_hip_comm *c, const char *what) {
    LRGE_SET_ERR(c->ctx, "RCCL transport: %s failed before the collective could be entered; communicator aborted", what);
    c->aborted = true;
    if (c->nccl && g_rccl.CommAbort) { (void)g_rccl.CommAbort(c->nccl); c->nccl = nullptr; }
    return LRGE_ERR_DEVICE;
}

// In-place SUM all-reduce of n elements of `esz` bytes (4: u32, 8: u64) in HOST memory.
static int comm_allreduce_sum_host(lrge_hip_comm *c, void *hbuf, size_t n, int esz, hipStream_t st) {
    lrge_hip_ctx *ctx = c->ctx;
    COMM_LIVE(c);
    if (comm_solo(c) || n == 0) return LRGE_OK;
    const size_t bytes = n * (size_t)esz;
    if (c->nccl) {
        ++c->rccl_ops;
        Scratch sc(ctx); char *d = nullptr;
        if (comm_small_stage(c, bytes, sc, &d)) return comm_rccl_give_up(c, "the staging allocation");
        if (hipMemcpyAsync(d, hbuf, bytes, hipMemcpyHostToDevice, st) != hipSuccess) { (void)hipGetLastError(); return comm_rccl_give_up(c, "the staging copy"); }
        NCCLCHK(ctx, g_rccl.AllReduce(d, d, n, esz == 8 ? LRGE_NCCL_UINT64 : LRGE_NCCL_UINT32, LRGE_NCCL_SUM, c->nccl, st));
        HIPCHK(ctx, hipMemcpyAsync(hbuf, d, bytes, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
        return LRGE_OK;
    }
    if (c->cb_allreduce) {
        if (c->cb_allreduce(c->cb_user, hbuf, n, esz) != 0) { LRGE_SET_ERR(ctx, "host communicator: all-reduce callback failed"); return LRGE_ERR_DEVICE; }
        return LRGE_OK;
    }
    LocalGroup *g = c->grp;
    g->slot[(size_t)c->rank] = hbuf;
    if (!grp_barrier(c)) { LRGE_SET_ERR(ctx, "local communicator: another rank failed"); return LRGE_ERR_DEVICE; }
    std::vector<char> sum(bytes, 0);
    for (int r = 0; r < c->world; ++r) {
        if (esz == 8) { const u64 *p = (const u64 *)g->slot[(size_t)r]; u64 *o = (u64 *)sum.data(); for (size_t i = 0; i < n; ++i) o[i] += p[i]; }
        else { const u32 *p = (const u32 *)g->slot[(size_t)r]; u32 *o = (u32 *)sum.data(); for (size_t i = 0; i < n; ++i) o[i] += p[i]; }
    }
    if (!grp_barrier(c)) { LRGE_SET_ERR(ctx, "local communicator: another rank failed"); return LRGE_ERR_DEVICE; }   // everyone has read every slot
    memcpy(hbuf, sum.data(), bytes);
    return LRGE_OK;
}

// All-gather of `bytes` bytes per rank between HOST buffers (hrecv holds world * bytes).
static int comm_allgather_host(lrge_hip_comm *c, const void *hsend, size_t bytes, void *hrecv, hipStream_t st) {
    lrge_hip_ctx *ctx = c->ctx;
    COMM_LIVE(c);
    if (bytes == 0) return LRGE_OK;
    if (comm_solo(c)) { memcpy(hrecv, hsend, bytes); return LRGE_OK; }
    if (c->nccl) {
        ++c->rccl_ops;
        Scratch sc(ctx); char *d = nullptr;
        if (comm_small_stage(c, bytes * ((size_t)c->world + 1), sc, &d)) return comm_rccl_give_up(c, "the staging allocation");
        if (hipMemcpyAsync(d, hsend, bytes, hipMemcpyHostToDevice, st) != hipSuccess) { (void)hipGetLastError(); return comm_rccl_give_up(c, "the staging copy"); }
        NCCLCHK(ctx, g_rccl.AllGather(d, d + bytes, bytes, LRGE_NCCL_UINT8, c->nccl, st));
        HIPCHK(ctx, hipMemcpyAsync(hrecv, d + bytes, bytes * (size_t)c->world, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
        return LRGE_OK;
    }
    if (c->cb_allgather) {
        if (c->cb_allgather(c->cb_user, hsend, bytes, hrecv) != 0) { LRGE_SET_ERR(ctx, "host communicator: all-gather callback failed"); return LRGE_ERR_DEVICE; }
        return LRGE_OK;
    }
    LocalGroup *g = c->grp;
    g->slot[(size_t)c->rank] = hsend;
    if (!grp_barrier(c)) { LRGE_SET_ERR(ctx, "local communicator: another rank failed"); return LRGE_ERR_DEVICE; }
    for (int r = 0; r < c->world; ++r) memcpy((char *)hrecv + bytes * (size_t)r, g->slot[(size_t)r], bytes);
    if (!grp_barrier(c)) { LRGE_SET_ERR(ctx, "local communicator: another rank failed"); return LRGE_ERR_DEVICE; }
    return LRGE_OK;
}

// Failure made collective: every rank contributes whether it failed so far (rc != 0); all ranks leave with an error if any
// did -- so that a rank whose allocation or kernel failed does not leave the others blocked in the next collective.
// One u32 all-reduce on a host word: no allocation, and no device round trip off RCCL.
static int comm_agree(lrge_hip_comm *c, int rc, hipStream_t st) {
    if (!c || comm_solo(c)) return rc;
    lrge_hip_ctx *ctx = c->ctx;
    if (c->grp && rc) { c->grp->abort(); return rc; }                    // (threads of one process: wake the others directly)
    std::string mine = rc ? ctx->err : std::string();
    u32 tot = rc ? 1u : 0u;
    const int r2 = comm_allreduce_sum_host(c, &tot, 1, 4, st);
    if (r2) { if (rc) ctx->err = mine; return rc ? rc : r2; }
    if (rc) { ctx->err = mine; return rc; }
    if (tot) { LRGE_SET_ERR(ctx, "collective call: %u other rank(s) failed", tot); return LRGE_ERR_DEVICE; }
    return LRGE_OK;
}

// Variable-size all-to-all between DEVICE buffers: rank r sends elements [soff[d], soff[d + 1]) of `dsend` to rank d and
// receives rank s's share at element roff[s] of `drecv` (soff / roff: world + 1 prefix sums, in elements of `esz` bytes;
// the receive counts come from an all-gather of the send counts, which the caller has done).  Ordered on `st`.
//   RCCL: one ncclSend / ncclRecv pair per peer inside a group (point-to-point over xGMI: every pair has its own link);
//   local: the ranks are threads of one process -- after a barrier every rank copies its share straight out of the other
//          ranks' send buffers (device-to-device; hipMemcpyPeer semantics between two GPUs of the process);
//   host callbacks: an all-gather of the padded send buffers through the caller's collective, each rank picks its slices
//          (the fallback transport: correct, not fast).
static int comm_alltoallv(lrge_hip_comm *c, const void *dsend, const u64 *soff, void *drecv, const u64 *roff, size_t esz, hipStream_t st) {
    lrge_hip_ctx *ctx = c->ctx;
    COMM_LIVE(c);
    const int W = c->world, me = c->rank;
    if (comm_solo(c)) {
        const u64 n = soff[1] - soff[0];
        if (n) HIPCHK(ctx, hipMemcpyAsync((char *)drecv + roff[0] * esz, (const char *)dsend + soff[0] * esz, n * esz, hipMemcpyDeviceToDevice, st));
        return LRGE_OK;
    }
    if (c->nccl) {
        const bool self = W == 1;        // (option RCCL_WORLD1: the rank's own share travels through a send / receive pair too)
        ++c->rccl_ops;
        NCCLCHK(ctx, g_rccl.GroupStart());
        int gerr = 0;      // a failed Send / Recv must not leave the group open on this thread: later RCCL calls would queue silently
        for (int p = 0; p < W && !gerr; ++p) {
            const u64 ns = soff[p + 1] - soff[p], nr = roff[p + 1] - roff[p];
            if (p == me && !self) continue;
            if (ns) gerr = g_rccl.Send((const char *)dsend + soff[p] * esz, ns * esz, LRGE_NCCL_UINT8, p, c->nccl, st);
            if (nr && !gerr) gerr = g_rccl.Recv((char *)drecv + roff[p] * esz, nr * esz, LRGE_NCCL_UINT8, p, c->nccl, st);
        }
        const int gend = g_rccl.GroupEnd();
        if (gerr || gend) {
            LRGE_SET_ERR(ctx, "RCCL error %s in the all-to-all's send / receive group", g_rccl.GetErrorString ? g_rccl.GetErrorString(gerr ? gerr : gend) : "?");
            return LRGE_ERR_DEVICE;
        }
        const u64 n = soff[me + 1] - soff[me];
        if (n && !self) HIPCHK(ctx, hipMemcpyAsync((char *)drecv + roff[me] * esz, (const char *)dsend + soff[me] * esz, n * esz, hipMemcpyDeviceToDevice, st));
        return LRGE_OK;
    }
    if (c->grp) {
        LocalGroup *g = c->grp;
        HIPCHK_GRP(c, hipStreamSynchronize(st));                   // my send buffer is complete before anybody reads it
        g->slot[(size_t)me] = dsend; g->slot2[(size_t)me] = soff;
        if (!grp_barrier(c)) { LRGE_SET_ERR(ctx, "local communicator: another rank failed"); return LRGE_ERR_DEVICE; }
        for (int s = 0; s < W; ++s) {
            const u64 *so = g->slot2[(size_t)s];
            const u64 n = so[me + 1] - so[me];
            if (n) HIPCHK_GRP(c, hipMemcpyAsync((char *)drecv + roff[s] * esz, (const char *)g->slot[(size_t)s] + so[me] * esz, n * esz, hipMemcpyDefault, st));
        }
        HIPCHK_GRP(c, hipStreamSynchronize(st));
        if (!grp_barrier(c)) { LRGE_SET_ERR(ctx, "local communicator: another rank failed"); return LRGE_ERR_DEVICE; }   // everybody has read: the send buffers may change
        return LRGE_OK;
    }
    // host callbacks: all-gather of [world + 1 offsets | padded payload]
    u64 mx = soff[W];
    {
        std::vector<u64> mine((size_t)W, 0), all((size_t)W * W, 0);
        mine[(size_t)me] = soff[W];
        Scratch sc(ctx);
        u64 *dv = sc.get<u64>((size_t)W);
        if (!dv) return LRGE_ERR_DEVICE;
        HIPCHK(ctx, hipMemcpyAsync(dv, mine.data(), (size_t)W * 8, hipMemcpyHostToDevice, st));
        int rc = comm_allreduce_sum(c, dv, (size_t)W, 8, st); if (rc) return rc;
        HIPCHK(ctx, hipMemcpyAsync(mine.data(), dv, (size_t)W * 8, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
        for (int r = 0; r < W; ++r) mx = std::max(mx, mine[(size_t)r]);
    }
    const size_t hdr = ((size_t)W + 1) * 8, blk = hdr + (size_t)mx * esz;
    std::vector<char> sendh(blk, 0), allh(blk * (size_t)W);
    memcpy(sendh.data(), soff, hdr);
    if (soff[W]) HIPCHK(ctx, hipMemcpyAsync(sendh.data() + hdr, dsend, soff[W] * esz, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    if (c->cb_allgather(c->cb_user, sendh.data(), blk, allh.data()) != 0) { LRGE_SET_ERR(ctx, "host communicator: all-gather callback failed"); return LRGE_ERR_DEVICE; }
    for (int s = 0; s < W; ++s) {
        const char *b = allh.data() + blk * (size_t)s;
        const u64 *so = (const u64 *)b;
        const u64 n = so[me + 1] - so[me];
        if (n) HIPCHK(ctx, hipMemcpyAsync((char *)drecv + roff[s] * esz, b + hdr + so[me] * esz, n * esz, hipMemcpyHostToDevice, st));
    }
    HIPCHK(ctx, hipStreamSynchronize(st));      // (`allh` is a local)
    return LRGE_OK;
}

// Variable-size all-gather between DEVICE buffers, k arrays at once: of array j rank r contributes the elements that land at
// [roff[r], roff[r + 1]) of g[j].recv on every rank (roff: world + 1 prefix sums of the contributions, in elements of esz bytes, known to
// every rank: the caller has all-gathered the counts).  Ordered on `st`.  ONE synchronisation for all k arrays.  The three transports of
// comm_alltoallv: RCCL = one send / receive pair per peer and array inside ONE group (every pair of GPUs has its own xGMI link;
// ncclAllGather wants equal sizes), local = one barrier pair and device-to-device copies out of the other ranks' buffers (their
// duration is kept in standin_ms: on a node the links deliver into HBM, there is no copy to pay), host callbacks = an all-gather of
// the arrays padded to the longest contribution.
struct GatherV { const void *send; void *recv; const u64 *roff; size_t esz; };
static int comm_allgatherv(lrge_hip_comm *c, const GatherV *g, int k, hipStream_t st) {
    lrge_hip_ctx *ctx = c->ctx;
    COMM_LIVE(c);
    const int W = c->world, me = c->rank;
    if (comm_solo(c) || c->nccl) {
        const bool self = W == 1 && !comm_solo(c);        // (option RCCL_WORLD1: the rank's own share travels through a send / receive pair)
        if (W > 1 || self) {
            ++c->rccl_ops;
            NCCLCHK(ctx, g_rccl.GroupStart());
            int gerr = 0;      // (a failed Send / Recv must not leave the group open on this thread)
            for (int j = 0; j < k && !gerr; ++j) {
                const u64 n_mine = g[j].roff[me + 1] - g[j].roff[me];
                for (int p = 0; p < W && !gerr; ++p) {
                    if (p == me && !self) continue;
                    const u64 nr = g[j].roff[p + 1] - g[j].roff[p];
                    if (n_mine) gerr = g_rccl.Send(g[j].send, n_mine * g[j].esz, LRGE_NCCL_UINT8, p, c->nccl, st);
                    if (nr && !gerr) gerr = g_rccl.Recv((char *)g[j].recv + g[j].roff[p] * g[j].esz, nr * g[j].esz, LRGE_NCCL_UINT8, p, c->nccl, st);
                }
            }
            const int gend = g_rccl.GroupEnd();
            if (gerr || gend) {
                LRGE_SET_ERR(ctx, "RCCL error %s in the all-gather's send / receive group", g_rccl.GetErrorString ? g_rccl.GetErrorString(gerr ? gerr : gend) : "?");
                return LRGE_ERR_DEVICE;
            }
        }
        for (int j = 0; j < k; ++j) {
            const u64 n_mine = g[j].roff[me + 1] - g[j].roff[me];
            if (n_mine && !self && (const char *)g[j].send != (char *)g[j].recv + g[j].roff[me] * g[j].esz)
                HIPCHK(ctx, hipMemcpyAsync((char *)g[j].recv + g[j].roff[me] * g[j].esz, g[j].send, n_mine * g[j].esz, hipMemcpyDeviceToDevice, st));
        }
        return LRGE_OK;
    }
    if (c->grp) {
        LocalGroup *grp = c->grp;
        HIPCHK_GRP(c, hipStreamSynchronize(st));                   // my buffers are complete before anybody reads them
        grp->slot[(size_t)me] = g;
        if (!grp_barrier(c)) { LRGE_SET_ERR(ctx, "local communicator: another rank failed"); return LRGE_ERR_DEVICE; }
        const double t0 = DevPool::now_ms();
        for (int s = 0; s < W; ++s) {
            const GatherV *gs = (const GatherV *)grp->slot[(size_t)s];
            for (int j = 0; j < k; ++j) {
                const u64 n = g[j].roff[s + 1] - g[j].roff[s];
                char *dst = (char *)g[j].recv + g[j].roff[s] * g[j].esz;
                if (n && dst != (const char *)gs[j].send) HIPCHK_GRP(c, hipMemcpyAsync(dst, gs[j].send, n * g[j].esz, hipMemcpyDefault, st));
            }
        }
        HIPCHK_GRP(c, hipStreamSynchronize(st));
        c->standin_ms += DevPool::now_ms() - t0;
        if (!grp_barrier(c)) { LRGE_SET_ERR(ctx, "local communicator: another rank failed"); return LRGE_ERR_DEVICE; }   // everybody has read: the buffers may change
        return LRGE_OK;
    }
    // host callbacks: per array, an all-gather of the buffers padded to the longest contribution
    for (int j = 0; j < k; ++j) {
        u64 mx = 0;
        for (int r = 0; r < W; ++r) mx = std::max(mx, g[j].roff[r + 1] - g[j].roff[r]);
        if (mx == 0) continue;
        const size_t esz = g[j].esz, blk = (size_t)mx * esz;
        const u64 n_mine = g[j].roff[me + 1] - g[j].roff[me];
        std::vector<char> sendh(blk, 0), allh(blk * (size_t)W);
        if (n_mine) HIPCHK(ctx, hipMemcpyAsync(sendh.data(), g[j].send, n_mine * esz, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
        if (c->cb_allgather(c->cb_user, sendh.data(), blk, allh.data()) != 0) { LRGE_SET_ERR(ctx, "host communicator: all-gather callback failed"); return LRGE_ERR_DEVICE; }
        for (int s = 0; s < W; ++s) {
            const u64 n = g[j].roff[s + 1] - g[j].roff[s];
            if (n) HIPCHK(ctx, hipMemcpyAsync((char *)g[j].recv + g[j].roff[s] * esz, allh.data() + blk * (size_t)s, n * esz, hipMemcpyHostToDevice, st));
        }
        HIPCHK(ctx, hipStreamSynchronize(st));      // (`allh` is a local)
    }
    return LRGE_OK;
}
