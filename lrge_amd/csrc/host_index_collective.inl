// host_index_collective.inl -- part of lrge_hip.hip (one translation unit; included there, in this order): what the collective index builds share: the restricted sketch launch (k_restrict.h), IndexBuildOpts, CollectiveGuard (a failing rank joins the next collective in its shape), and sharded_collect -- the three exchanges of the query-sharded build with a sharded target sketch (k_route.h).
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

// A collective call must fail on every rank when it fails on one, and nobody may hang.  The builds below are a fixed sequence of
// collectives; a rank that leaves early (any `return` of the macros above) is, at that moment, owed to ONE of them -- the next
// status-carrying collective its healthy peers will enter.  The guard knows which (expect() is called as the sequence advances)
// and its destructor JOINS that collective in its own shape with the status word set: an all-reduce of n u64 with a 1 at
// `status_at`, an all-gather of n u64 per rank likewise, or the one-word agreement.  The joins work on host vectors
// (comm.h: comm_*_host), so they need no allocation and -- off RCCL -- no working device.  Threads of one process simply abort
// the group's barrier.  (a failing rank used to enter a ONE-word agreement while its peers were in the (W + 1)-word
// sizes all-reduce.)
struct CollectiveGuard {
    lrge_hip_comm *c; hipStream_t st;
    enum Next { NONE = 0, AGREE, ALLREDUCE_U64, ALLGATHER_U64 };
    int next = NONE; size_t n = 0, status_at = 0;
    void expect(int k, size_t n_ = 0, size_t at = 0) { next = k; n = n_; status_at = at; }
    void disarm() { next = NONE; }
    void join_failed() {
        if (!c || comm_solo(c) || next == NONE) { next = NONE; return; }
        const int k = next; next = NONE;
        const std::string mine = c->ctx->err;            // (the join must not overwrite this rank's own error text)
        if (c->grp) c->grp->abort();
        else if (k == AGREE) (void)comm_agree(c, LRGE_ERR_DEVICE, st);
        else if (k == ALLREDUCE_U64) { std::vector<u64> v(n, 0); v[status_at] = 1; (void)comm_allreduce_sum_host(c, v.data(), n, 8, st); }
        else { std::vector<u64> v(n, 0), all(n * (size_t)c->world, 0); v[status_at] = 1; (void)comm_allgather_host(c, v.data(), n * 8, all.data(), st); }
        c->ctx->err = mine;
    }
    ~CollectiveGuard() { join_failed(); }
};
// test hook (option DEBUG_SHARD_FAIL_AT = stage number, set by lrge_hip_ctx_set_option only): this rank fails at that stage of a
// collective build, as an allocation or a kernel would
static bool shard_fail_at(lrge_hip_ctx *ctx, int stage) {
    if ((int)ctx->opt_u64("DEBUG_SHARD_FAIL_AT", 0) != stage) return false;
    LRGE_SET_ERR(ctx, "injected failure at stage %d of the collective index build", stage);
    return true;
}
// words of the statistics all-reduce that closes the collective part of a restricted / sharded build: [distinct, minimizers,
// head bins..., status]
static size_t stats_vec_words(const Preset &P) { return (size_t)std::min<u32>(4096, (u32)P.max_mid_occ + 2) + 3; }

// The three exchanges of a sharded build (k_route.h).  On success so->x [, so->y] hold this rank's kept entries in the order
// the one index would hold them (so->n of them), *own_hashes / *n_own the hashes of the keys this rank owns.  Collective:
// a failure on one rank fails the call on every rank (status words ride in the small vectors; comm_agree before the
// exchanges that follow large allocations).
static int sharded_collect(lrge_hip_ctx *ctx, Scratch &sc, const Preset &P, int preset, bool pk, u32 pk_pos1, u32 pk_ybits,
                           const IndexBuildOpts *ro, SketchOut *so, u64 **own_hashes, u64 *n_own, CollectiveGuard &cg) {
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
    // (the caller armed the guard for this very all-reduce: lrge_hip_index_build_sharded)
    std::vector<u64> hv((size_t)W + 1, 0);
    hv[(size_t)me] = S->total_bases; hv[(size_t)W] = shard_fail_at(ctx, 1) ? 1 : 0;
    const bool failed1 = hv[(size_t)W] != 0;
    cg.disarm();
    rc = comm_allreduce_sum_host(c, hv.data(), hv.size(), 8, st); if (rc) return rc;
    if (hv[(size_t)W]) { if (!failed1) LRGE_SET_ERR(ctx, "sharded index build: another rank failed"); return LRGE_ERR_DEVICE; }
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
        if (shard_fail_at(ctx, 2)) return LRGE_ERR_DEVICE;
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
    cg.expect(CollectiveGuard::AGREE);               // (large allocations behind us or failed: one word says which, before the key sets travel)
    rc = local1();
    mark("sketches + key set");
    cg.disarm();
    rc = comm_agree(c, rc, st); if (rc) return rc;
    g_shard_stats.entries_sketched = raw.n;
    mark("agree");
    rc = comm_allgather(c, ks.bits, n_words * 8, gathered, st); if (rc) return rc;
    mark("key-set all-gather");
    cg.expect(CollectiveGuard::ALLGATHER_U64, (size_t)2 * W + 1, (size_t)2 * W);      // the counts all-gather of (4)
    // ---- (3) local: which ranks ask for every entry, who owns its hash; counts per destination ----
    const u64 Mr = raw.n;
    if (Mr >= (1ULL << 32)) { LRGE_SET_ERR(ctx, "sharded index build: this rank's target share yields %llu minimizers (limit 2^32)", (unsigned long long)Mr); }
    RouteArgs A; A.x = raw.x; A.y = pk ? nullptr : raw.y; A.n = Mr; A.kshift = pk ? pk_ybits : 0;
    A.ks = KeySetAll{inter, n_words - 1, (u32)W}; A.n_tiles = (u32)std::max<u64>(1, div_up(Mr, RF_TILE));
    u32 *flags = nullptr, *cnt = nullptr, *d_tot = nullptr;
    std::vector<u64> mine((size_t)2 * W + 1, 0), matrix(((size_t)2 * W + 1) * (size_t)W, 0);
    auto local2 = [&]() -> int {
        if (shard_fail_at(ctx, 3)) return LRGE_ERR_DEVICE;
        if (W <= 8) hipLaunchKernelGGL(k_keyset_interleave<8>, dim3((u32)div_up(n_words, 256)), dim3(256), 0, st, gathered, n_words, (u32)W, inter);
        else hipLaunchKernelGGL(k_keyset_interleave<16>, dim3((u32)div_up(n_words, 256)), dim3(256), 0, st, gathered, n_words, (u32)W, inter);
        KCHK(ctx);
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
    // ---- (4) everybody learns every (source, destination) count (and whether a rank has failed): host vectors, no allocation ----
    cg.disarm();
    rc = comm_allgather_host(c, mine.data(), mine.size() * 8, matrix.data(), st); if (rc) return rc;
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
        if (shard_fail_at(ctx, 4)) return LRGE_ERR_DEVICE;
        if (n_kr >= (1ULL << 32)) { LRGE_SET_ERR(ctx, "index limited to < 2^32 minimizers (this rank would hold %llu)", (unsigned long long)n_kr); return LRGE_ERR_TOO_MANY; }
        sx = sc.get<u64>(n_ks + 1); rx = sc.get<u64>(n_kr + 1); rh = sc.get<u64>(n_or + 1);
        if (narrow) { sh32 = sc.get<u32>(n_os + 1); rh32 = sc.get<u32>(n_or + 1); } else sh = sc.get<u64>(n_os + 1);
        if (!pk) { sy = sc.get<u64>(n_ks + 1); ry = sc.get<u64>(n_kr + 1); }
        if (!sx || !rx || !rh || (narrow ? (!sh32 || !rh32) : !sh) || (!pk && (!sy || !ry))) return LRGE_ERR_DEVICE;
        RouteBases B;
        for (int d = 0; d < ROUTE_MAX_WORLD; ++d) { B.keep[d] = d < W ? ks_off[(size_t)d] : 0; B.own[d] = d < W ? os_off[(size_t)d] : 0; }
        if (Mr) { hipLaunchKernelGGL(k_route_write, dim3(A.n_tiles), dim3(RF_THREADS), 0, st, A, flags, cnt, B, sx, sy, sh, sh32); KCHK(ctx); }
        // the raw sketch has been read for the last time: its blocks serve this rank's later requests (recycled in stream order) --
        // at H. sapiens scale 15 GB per rank that need not stay resident across the exchanges
        sc.drop(raw.x); raw.x = nullptr; if (raw.y) { sc.drop(raw.y); raw.y = nullptr; }
        sc.drop(flags); flags = nullptr; sc.drop(cnt); cnt = nullptr; sc.drop(d_tot); d_tot = nullptr;
        return LRGE_OK;
    };
    cg.expect(CollectiveGuard::AGREE);               // (the send / receive buffers are the build's largest allocations)
    rc = local3();
    mark("route write");
    cg.disarm();
    rc = comm_agree(c, rc, st); if (rc) return rc;
    mark("agree");
    // ---- (6) the exchanges ----
    rc = comm_alltoallv(c, sx, ks_off.data(), rx, kr_off.data(), 8, st); if (rc) return rc;
    if (!pk) { rc = comm_alltoallv(c, sy, ks_off.data(), ry, kr_off.data(), 8, st); if (rc) return rc; }
    if (narrow) {
        rc = comm_alltoallv(c, sh32, os_off.data(), rh32, or_off.data(), 4, st); if (rc) return rc;
        if (n_or) { hipLaunchKernelGGL(k_u32_to_u64, dim3((u32)div_up(n_or, 256)), dim3(256), 0, st, rh32, n_or, rh); KCHK(ctx); }
    } else { rc = comm_alltoallv(c, sh, os_off.data(), rh, or_off.data(), 8, st); if (rc) return rc; }
    // from here to the statistics all-reduce of index_build_one a rank that fails owes its peers THAT collective
    cg.expect(CollectiveGuard::ALLREDUCE_U64, stats_vec_words(P), stats_vec_words(P) - 1);
    if (shard_fail_at(ctx, 5)) return LRGE_ERR_DEVICE;
    HIPCHK(ctx, hipStreamSynchronize(st));        // (the offset vectors are locals; the local transport has synchronised already)
    mark("all-to-alls");
    if (raw.x) sc.drop(raw.x); if (raw.y) sc.drop(raw.y);
    if (flags) sc.drop(flags); if (cnt) sc.drop(cnt); if (d_tot) sc.drop(d_tot); sc.drop(sx); if (sh) sc.drop(sh); if (sh32) sc.drop(sh32); if (rh32) sc.drop(rh32); if (sy) sc.drop(sy);
    sc.drop(ks.bits); sc.drop(gathered); sc.drop(inter);
    so->x = rx; so->y = ry; so->mz_off = nullptr; so->n = n_kr;
    *own_hashes = rh; *n_own = n_or;
    const u64 ss[8] = {g_shard_stats.keyset_bytes, g_shard_stats.entries_sketched, g_shard_stats.entries_sent, g_shard_stats.entries_recv,
                       g_shard_stats.hashes_sent, g_shard_stats.hashes_recv, (u64)(pk ? 8 : 16) | (u64)(narrow ? 4 : 8) << 8, n_kr};
    memcpy(ctx->shard_stats, ss, sizeof ss);
    return LRGE_OK;
}


