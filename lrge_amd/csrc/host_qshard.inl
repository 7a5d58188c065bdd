// host_qshard.inl -- part of lrge_hip.hip (one translation unit; included there, in this order): lrge_hip_seqset_presketch_sharded, the
// streamed set's sketch made ONCE per world instead of once per rank.
//
// The forward strategy with the targets sharded (host_tshard.inl) has every rank map ALL queries against its share of the index, so
// every rank used to sketch all queries: K1 is VALU-bound, 18-20 ms of a rank's 147 at H. sapiens scale and world 8 (round 5) -- work
// that does not shrink with the world.  twoset.rs:266-334 maps the queries independently of each other and mm2:map.c collect_minimizers
// sketches a query from its own bases alone, so WHERE a query is sketched is free: rank r sketches the r-th share of the reads (cut by
// bases, from the lengths every rank holds) and the minimizers -- (x, y) pairs in read order, exactly the stream one rank's sketch of
// the whole set yields -- are all-gathered: 16 bytes per minimizer (374 M x 16 B = 6 GB at full-size C5: 0.75 GB per xGMI link and
// rank).  The result is attached to the set as its presketch (host_sketch.inl: PreSketch) and consumed by the next overlap call.
//
// A collective call: the sequence is fixed, a rank that fails joins the next one in its own shape with the status word set
// (CollectiveGuard, host_index_collective.inl).
//   Q1 all-gather  u64[2]       minimizers of this rank's share, status
//   QA agreement                (receive buffers taken on every rank)
//   Q2-Q4 all-gather-v          x (u64), y (u64), per-read offsets (u32)
//   QB agreement                (everything arrived on every rank: nobody leaves with LRGE_OK alone)
__global__ void k_add_rid_base(u64 *__restrict__ y, u64 n, u32 r0) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] += (u64)r0 << 32;
}
__global__ void k_add_u32(u32 *__restrict__ a, u32 n, u32 add) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] += add;
}

// reads [b[r], b[r + 1]) go to rank r: the cut closest below r / W of the bases (every rank computes the same cuts from the same lengths)
static std::vector<u32> qshard_cuts(const lrge_hip_seqset *s, int W) {
    std::vector<u32> b((size_t)W + 1, s->n);
    b[0] = 0;
    u64 acc = 0; int next = 1;
    for (u32 r = 0; r < s->n && next < W; ++r) {
        while (next < W && acc * (u64)W >= (u64)next * s->total_bases) b[(size_t)next++] = r;
        acc += s->h_len[r];
    }
    return b;
}

extern "C" int lrge_hip_seqset_presketch_sharded(lrge_hip_ctx *ctx, lrge_hip_seqset *s, int preset, lrge_hip_comm *comm) {
    if (!ctx || !s || !comm) return LRGE_ERR_INVALID;
    // (argument errors are rank-local by nature -- every rank passes the same job -- so they return before any collective)
    if (preset != LRGE_PRESET_AVA_ONT && preset != LRGE_PRESET_AVA_PB) { LRGE_SET_ERR(ctx, "Preset not found: %d", preset); return LRGE_ERR_INVALID; }
    if (s->ctx != ctx || comm->ctx != ctx) { LRGE_SET_ERR(ctx, "presketch_sharded: set / communicator belong to another context"); return LRGE_ERR_INVALID; }
    if (s->is_view) { LRGE_SET_ERR(ctx, "presketch_sharded: a whole set, not a view"); return LRGE_ERR_INVALID; }
    if (s->total_bases + 1 >= (1ULL << 32)) { LRGE_SET_ERR(ctx, "presketch_sharded: a set above 2^32 bases is streamed in views (sketched per view)"); return LRGE_ERR_TOO_MANY; }
    const int W = comm->world, me = comm->rank;
    hipStream_t st = ctx->stream;
    CollectiveGuard cg{comm, st};
    cg.expect(CollectiveGuard::ALLGATHER_U64, 2, 1);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    presketch_discard(s);
    const std::vector<u32> cut = qshard_cuts(s, W);
    const u32 r0 = cut[(size_t)me], r1 = cut[(size_t)me + 1];
    // ---- this rank's share ----
    Scratch lsc(ctx);                     // the share's own sketch: gone when the call returns
    SketchOut so;
    lrge_hip_seqset *view = nullptr;
    struct ViewGuard { lrge_hip_seqset *&v; ~ViewGuard() { if (v) lrge_hip_seqset_free(v); } } view_guard{view};
    hipEvent_t ev_start = ctx->get_event(), ev_done = ctx->get_event();
    struct EvGuard { lrge_hip_ctx *c; hipEvent_t &a, &b; ~EvGuard() { if (a) c->event_pool.push_back(a); if (b) c->event_pool.push_back(b); } } ev_guard{ctx, ev_start, ev_done};
    auto local1 = [&]() -> int {
        if (shard_fail_at(ctx, 20)) return LRGE_ERR_DEVICE;
        int rc = seqset_ready(ctx, s); if (rc) return rc;
        HIPCHK(ctx, hipEventRecord(ev_start, st));
        if (r1 > r0) {
            rc = seqset_view(ctx, s, r0, r1, &view); if (rc) return rc;
            rc = sketch_device(ctx, lsc, view, preset, false, &so); if (rc) return rc;
            if (so.n) {
                hipLaunchKernelGGL(k_add_rid_base, dim3((u32)div_up(so.n, 256)), dim3(256), 0, st, so.y, so.n, r0);
                KCHK(ctx);
            }
        }
        return LRGE_OK;
    };
    int rc = local1();
    const bool failed1 = rc != LRGE_OK;
    std::vector<u64> mine(2, 0), all((size_t)2 * W, 0);
    mine[0] = failed1 ? 0 : so.n; mine[1] = failed1 ? 1 : 0;
    cg.disarm();
    rc = comm_allgather_host(comm, mine.data(), 16, all.data(), st); if (rc) return rc;
    std::vector<u64> off((size_t)W + 1, 0), roff_reads((size_t)W + 1, 0);
    for (int r = 0; r < W; ++r) {
        if (all[(size_t)2 * r + 1]) { if (!failed1) LRGE_SET_ERR(ctx, "sharded presketch: rank %d failed", r); return LRGE_ERR_DEVICE; }
        off[(size_t)r + 1] = off[(size_t)r] + all[(size_t)2 * r];
        roff_reads[(size_t)r + 1] = cut[(size_t)r + 1];
    }
    const u64 M = off[(size_t)W];
    if (M >= (1ULL << 32)) { LRGE_SET_ERR(ctx, "query set limited to < 2^32 minimizers"); return LRGE_ERR_TOO_MANY; }      // (the same verdict on every rank)
    // ---- the whole set's sketch: this rank's copy ----
    std::unique_ptr<PreSketch> p(new PreSketch());
    p->preset = preset;
    p->sc = new Scratch(ctx);
    struct ScGuard { std::unique_ptr<PreSketch> &p; ~ScGuard() { if (p) { delete p->sc; p->sc = nullptr; } } } sc_guard{p};
    cg.expect(CollectiveGuard::AGREE);
    auto local2 = [&]() -> int {
        if (shard_fail_at(ctx, 21)) return LRGE_ERR_DEVICE;
        Scratch &sc = *p->sc;
        p->x = sc.get<u64>(M + 1); p->y = sc.get<u64>(M + 1); p->mz_off = sc.get<u32>((size_t)s->n + 1); p->d_total = sc.get<u32>(1);
        if (!p->x || !p->y || !p->mz_off || !p->d_total) return LRGE_ERR_DEVICE;
        // my share's per-read offsets start at my place in the whole stream
        if (r1 > r0 && off[(size_t)me]) { hipLaunchKernelGGL(k_add_u32, dim3((u32)div_up(r1 - r0, 256)), dim3(256), 0, st, so.mz_off, r1 - r0, (u32)off[(size_t)me]); KCHK(ctx); }
        return LRGE_OK;
    };
    rc = local2();
    cg.disarm();
    rc = comm_agree(comm, rc, st); if (rc) return rc;                                                   // QA
    cg.expect(CollectiveGuard::AGREE);
    auto tail = [&]() -> int {
        int r2 = comm_allgatherv(comm, so.x, p->x, off.data(), 8, st); if (r2) return r2;              // Q2
        r2 = comm_allgatherv(comm, so.y, p->y, off.data(), 8, st); if (r2) return r2;                  // Q3
        r2 = comm_allgatherv(comm, so.mz_off, p->mz_off, roff_reads.data(), 4, st); if (r2) return r2; // Q4
        if (shard_fail_at(ctx, 22)) return LRGE_ERR_DEVICE;
        hipLaunchKernelGGL(k_store_u32, dim3(1), dim3(1), 0, st, p->mz_off + s->n, (u32)M);
        hipLaunchKernelGGL(k_store_u32, dim3(1), dim3(1), 0, st, p->d_total, (u32)M);
        KCHK(ctx);
        HIPCHK(ctx, hipEventRecord(ev_done, st));
        HIPCHK(ctx, hipStreamSynchronize(st));            // (the share's sketch is released when this function returns)
        return LRGE_OK;
    };
    rc = tail();
    cg.disarm();
    rc = comm_agree(comm, rc, st); if (rc) return rc;                                                   // QB
    g_shard_stats = ShardStats();
    const u64 ss[8] = {0, 0, (M - mine[0]), (M - mine[0]), 0, 0, (u64)16 | (u64)8 << 8, 0};             // (entries_sent / _recv slots: minimizers of 16 bytes)
    memcpy(ctx->shard_stats, ss, sizeof ss); ctx->qshard_fresh = true;
    p->ev_start = ev_start; p->ev_done = ev_done; ev_start = nullptr; ev_done = nullptr;
    s->presk = p.release();
    return LRGE_OK;
}
