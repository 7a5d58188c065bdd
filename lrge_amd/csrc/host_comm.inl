// host_comm.inl -- part of lrge_hip.hip (one translation unit; included there, in this order): communicators (comm.h) behind the C ABI.
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
    c->ctx = ctx; c->rank = rank; c->world = world; c->force = ctx->opt("RCCL_WORLD1") != nullptr;
    NCCLCHK(ctx, g_rccl.CommInitRank(&c->nccl, world, id, rank));
    if (hipMalloc((void **)&c->d_small, lrge_hip_comm::kSmall) != hipSuccess) { (void)hipGetLastError(); c->d_small = nullptr; }   // (then the pool serves)
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
    if (c->nccl || c->d_small) {
        // (a communicator that outlives its context -- as lrge_hip_index_free / _seqset_free tolerate too -- must not touch it)
        bool ctx_alive;
        { std::lock_guard<std::mutex> g(g_live_mu); ctx_alive = g_live_ctx.count(c->ctx) != 0; }
        if (ctx_alive) { (void)hipSetDevice(c->ctx->device); (void)hipStreamSynchronize(c->ctx->stream); }
        if (c->d_small) (void)hipFree(c->d_small);
        if (c->nccl) (void)g_rccl.CommDestroy(c->nccl);      // (an aborted communicator is gone already)
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
// librccl data-path calls made through this communicator so far (tests: the RCCL branches really ran)
extern "C" int lrge_hip_comm_rccl_ops(const lrge_hip_comm *c, uint64_t *n) {
    if (!c || !n) return LRGE_ERR_INVALID;
    *n = c->rccl_ops;
    return LRGE_OK;
}
extern "C" int lrge_hip_comm_local_group_serialize(void *grp, int on) {
    if (!grp) return LRGE_ERR_INVALID;
    ((LocalGroup *)grp)->serialize = on != 0;
    ((LocalGroup *)grp)->trim_on_yield = on == 2;       // (2: the ranks also hand their idle arena segments back between turns)
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
extern "C" double lrge_hip_comm_standin_ms(lrge_hip_comm *c, int reset) {
    if (!c) return 0.0;
    const double v = c->standin_ms;
    if (reset) c->standin_ms = 0;
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

// host-buffer forms of the two collectives that close a step.  They work on the caller's host vectors directly
// (comm_*_host: no device allocation, and off RCCL no device round trip), so a rank that has just run out of device memory can
// still JOIN the collective that closes its step -- with a status word in its vector, if the caller wants the failure collective
// (lrge_amd/parallel.py, integration/liblrge_hip_shim.rs).
extern "C" int lrge_hip_comm_allreduce_u32(lrge_hip_comm *c, uint32_t *inout, size_t n) {
    if (!c || (n && !inout)) return LRGE_ERR_INVALID;
    lrge_hip_ctx *ctx = c->ctx;
    if (comm_solo(c) || n == 0) return LRGE_OK;
    if (c->nccl) HIPCHK(ctx, hipSetDevice(ctx->device));
    return comm_allreduce_sum_host(c, inout, n, 4, ctx->stream);
}

extern "C" int lrge_hip_comm_allgather(lrge_hip_comm *c, const void *send, size_t bytes, void *recv) {
    if (!c || (bytes && (!send || !recv))) return LRGE_ERR_INVALID;
    lrge_hip_ctx *ctx = c->ctx;
    if (bytes == 0) return LRGE_OK;
    if (comm_solo(c)) { memcpy(recv, send, bytes); return LRGE_OK; }
    if (c->nccl) HIPCHK(ctx, hipSetDevice(ctx->device));
    return comm_allgather_host(c, send, bytes, recv, ctx->stream);
}

// A rank that cannot go on (an error between two collectives: a failed upload before a collective index build, a failed overlap
// call before the all-reduce that closes the step) calls this instead of leaving its peers waiting for it for ever:
//   local transport  -- the group is aborted: every rank blocked in, or later entering, a collective of the group returns
//                       LRGE_ERR_DEVICE ("another rank failed");
//   RCCL             -- ncclCommAbort on this rank's communicator; the peers' pending RCCL operations end when their own processes
//                       abort (the launcher's job: torch.distributed.run tears the group down when one process fails);
//   host callbacks   -- the caller owns the collectives; nothing to do here.
// Afterwards every collective on `c` fails at once; lrge_hip_comm_destroy is still to be called.
extern "C" int lrge_hip_comm_abort(lrge_hip_comm *c) {
    if (!c) return LRGE_ERR_INVALID;
    c->aborted = true;
    if (c->grp) c->grp->abort();
    else if (c->nccl && g_rccl.CommAbort) { (void)g_rccl.CommAbort(c->nccl); c->nccl = nullptr; }
    return LRGE_OK;
}
