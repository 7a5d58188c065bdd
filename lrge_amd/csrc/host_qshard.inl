// host_qshard.inl -- part of lrge_hip.hip (one translation unit; included there, in this order): lrge_hip_seqset_presketch_sharded, the
// streamed set's sketch made ONCE per world instead of once per rank.
//
// The forward strategy with the targets sharded (host_tshard.inl) has every rank map ALL queries against its share of the index, so
// every rank used to sketch all queries: K1 is VALU-bound, 18-20 ms of a rank's 147 at H. sapiens scale and world 8 (round 5) -- work
// that does not shrink with the world.  twoset.rs:266-334 maps the queries independently of each other and mm2:map.c collect_minimizers
// sketches a query from its own bases alone, so WHERE a query is sketched is free: rank r sketches the r-th share of the reads (cut by
// bases, from the lengths every rank holds) and the minimizers are all-gathered, in read order: exactly the stream one rank's sketch
// of the whole set yields.  What travels per minimizer is ONE word whenever it fits -- x = hash << 8 | span (2k + 8 bits) over
// pos << 1 | strand; the read id follows from the per-read offsets, which travel too (4 bytes per read) -- 374 M x 8 B = 3 GB at
// full-size C5: 0.37 GB per xGMI link and rank; reads longer than the word allows (2k + 8 + bits(pos) + 1 > 64) send (x, y) pairs.
// The result is attached to the set as its presketch (host_sketch.inl: PreSketch) and consumed by the next overlap call.
//
// A collective call: the sequence is fixed, a rank that fails joins the next one in its own shape with the status word set
// (CollectiveGuard, host_index_collective.inl).
//   Q1 all-gather  u64[2]       minimizers of this rank's share, status
//   QA agreement                (receive buffers taken on every rank)
//   Q2 all-gather-v             the words [or x and y], the per-read offsets: one synchronisation
//   QB agreement                (everything arrived and was unpacked on every rank: nobody leaves with LRGE_OK alone)
__global__ void k_qs_rid_base(u64 *__restrict__ y, u64 n, u32 r0) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] += (u64)r0 << 32;
}
__global__ void k_qs_pack(u64 *__restrict__ x, const u64 *__restrict__ y, u64 n, u32 pbits) {      // x[i] <- x[i] << pbits | (pos << 1 | strand)
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = x[i] << pbits | (y[i] & ((1ULL << pbits) - 1));
}
__global__ void k_add_u32(u32 *__restrict__ a, u32 n, u32 add) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] += add;
}
// one wavefront per read: its words -> (x, y) pairs, in place for x
__global__ __launch_bounds__(256) void k_qs_unpack(u64 *__restrict__ x, u64 *__restrict__ y, const u32 *__restrict__ mz_off, u32 n_reads, u32 pbits) {
    const u32 r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_reads) return;
    const u32 b = mz_off[r], e = mz_off[r + 1];
    const u64 m = (1ULL << pbits) - 1;
    for (u32 i = b + lane_id(); i < e; i += 64) { const u64 w = x[i]; x[i] = w >> pbits; y[i] = (u64)r << 32 | (w & m); }
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
    const Preset P = make_preset(preset);
    const std::vector<u32> cut = qshard_cuts(s, W);
    const u32 r0 = cut[(size_t)me], r1 = cut[(size_t)me + 1];
    // one word per minimizer when x (2k + 8 bits) and pos << 1 | strand fit it (option QSHARD_PAIRS: always pairs; the tests run both)
    const u32 pbits = std::max<u32>(1, ceil_log2_u64((u64)s->max_len + 1)) + 1;
    const bool words = 2 * (u32)P.k + 8 + pbits <= 64 && !ctx->opt("QSHARD_PAIRS");
    // ---- this rank's share, and room for everybody's ----
    Scratch lsc(ctx);                     // the share's own sketch: gone when the call returns
    SketchOut so;
    lrge_hip_seqset *view = nullptr;
    struct ViewGuard { lrge_hip_seqset *&v; ~ViewGuard() { if (v) lrge_hip_seqset_free(v); } } view_guard{view};
    hipEvent_t ev_start = ctx->get_event(), ev_done = ctx->get_event();
    struct EvGuard { lrge_hip_ctx *c; hipEvent_t &a, &b; ~EvGuard() { if (a) c->event_pool.push_back(a); if (b) c->event_pool.push_back(b); } } ev_guard{ctx, ev_start, ev_done};
    std::unique_ptr<PreSketch> p(new PreSketch());
    p->preset = preset;
    p->sc = new Scratch(ctx);
    struct ScGuard { std::unique_ptr<PreSketch> &p; ~ScGuard() { if (p) { delete p->sc; p->sc = nullptr; } } } sc_guard{p};
    auto take = [&](u64 n_) -> int {
        Scratch &sc = *p->sc;
        p->x = sc.get<u64>(n_ + 1); p->y = sc.get<u64>(n_ + 1);
        return (p->x && p->y) ? LRGE_OK : LRGE_ERR_DEVICE;
    };
    auto local1 = [&]() -> int {
        if (shard_fail_at(ctx, 20)) return LRGE_ERR_DEVICE;
        int rc = seqset_ready(ctx, s); if (rc) return rc;
        HIPCHK(ctx, hipEventRecord(ev_start, st));
        if (r1 > r0) {
            rc = seqset_view(ctx, s, r0, r1, &view); if (rc) return rc;
            rc = sketch_device(ctx, lsc, view, preset, false, &so); if (rc) return rc;
            if (so.n) {
                if (words) hipLaunchKernelGGL(k_qs_pack, dim3((u32)div_up(so.n, 256)), dim3(256), 0, st, so.x, so.y, so.n, pbits);
                else hipLaunchKernelGGL(k_qs_rid_base, dim3((u32)div_up(so.n, 256)), dim3(256), 0, st, so.y, so.n, r0);
                KCHK(ctx);
            }
        }
        p->mz_off = p->sc->get<u32>((size_t)s->n + 1); p->d_total = p->sc->get<u32>(1);
        if (!p->mz_off || !p->d_total) return LRGE_ERR_DEVICE;
        return LRGE_OK;
    };
    int rc = local1();
    const bool failed1 = rc != LRGE_OK;
    std::vector<u64> mine(2, 0), all((size_t)2 * W, 0);
    mine[0] = failed1 ? 0 : so.n; mine[1] = failed1 ? 1 : 0;
    cg.disarm();
    rc = comm_allgather_host(comm, mine.data(), 16, all.data(), st); if (rc) return rc;                 // Q1
    std::vector<u64> off((size_t)W + 1, 0), roff_reads((size_t)W + 1, 0);
    for (int r = 0; r < W; ++r) {
        if (all[(size_t)2 * r + 1]) { if (!failed1) LRGE_SET_ERR(ctx, "sharded presketch: rank %d failed", r); return LRGE_ERR_DEVICE; }
        off[(size_t)r + 1] = off[(size_t)r] + all[(size_t)2 * r];
        roff_reads[(size_t)r + 1] = cut[(size_t)r + 1];
    }
    const u64 M = off[(size_t)W];
    if (M >= (1ULL << 32)) { LRGE_SET_ERR(ctx, "query set limited to < 2^32 minimizers"); return LRGE_ERR_TOO_MANY; }      // (the same verdict on every rank)
    // the receive buffers, sized exactly (a worst-case or estimated size held across the index build that follows is what N ranks sharing
    // one GPU -- the emulation -- cannot afford: 8 x 10 GB), and a word of agreement on them
    cg.expect(CollectiveGuard::AGREE);
    rc = shard_fail_at(ctx, 21) ? LRGE_ERR_DEVICE : take(M);
    cg.disarm();
    rc = comm_agree(comm, rc, st); if (rc) return rc;                                                   // QA
    cg.expect(CollectiveGuard::AGREE);
    auto tail = [&]() -> int {
        // my share's per-read offsets start at my place in the whole stream
        if (r1 > r0 && off[(size_t)me]) { hipLaunchKernelGGL(k_add_u32, dim3((u32)div_up(r1 - r0, 256)), dim3(256), 0, st, so.mz_off, r1 - r0, (u32)off[(size_t)me]); KCHK(ctx); }
        GatherV g[3] = {{so.x, p->x, off.data(), 8}, {so.mz_off, p->mz_off, roff_reads.data(), 4}, {so.y, p->y, off.data(), 8}};
        int r2 = comm_allgatherv(comm, g, words ? 2 : 3, st); if (r2) return r2;                        // Q2
        if (shard_fail_at(ctx, 22)) return LRGE_ERR_DEVICE;
        hipLaunchKernelGGL(k_store_u32, dim3(1), dim3(1), 0, st, p->mz_off + s->n, (u32)M);
        hipLaunchKernelGGL(k_store_u32, dim3(1), dim3(1), 0, st, p->d_total, (u32)M);
        if (words && s->n) hipLaunchKernelGGL(k_qs_unpack, dim3((u32)div_up(s->n, 4)), dim3(256), 0, st, p->x, p->y, p->mz_off, s->n, pbits);
        KCHK(ctx);
        HIPCHK(ctx, hipEventRecord(ev_done, st));
        HIPCHK(ctx, hipStreamSynchronize(st));            // (the share's sketch is released when this function returns)
        return LRGE_OK;
    };
    rc = tail();
    cg.disarm();
    rc = comm_agree(comm, rc, st); if (rc) return rc;                                                   // QB
    g_shard_stats = ShardStats();
    // (entries_sent / _recv slots: minimizers this rank's sketch sent to each of the others / received; [6] low byte: bytes each)
    const u64 ss[8] = {0, 0, (u64)(W - 1) * mine[0], (M - mine[0]), 0, 0, (u64)(words ? 8 : 16) | (u64)8 << 8, 0};
    memcpy(ctx->shard_stats, ss, sizeof ss); ctx->qshard_fresh = true;
    p->ev_start = ev_start; p->ev_done = ev_done; ev_start = nullptr; ev_done = nullptr;
    s->presk = p.release();
    return LRGE_OK;
}
