// host_index.inl -- part of lrge_hip.hip (one translation unit; included there, in this order): index build: plain, restricted (k_restrict.h), sharded over ranks (k_route.h), partitioned; statistics, dump, release.
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
        if (!c || c->world == 1 || next == NONE) { next = NONE; return; }
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
    // the collective this rank owes its peers if it fails now: a sharded build opens with the sizes all-reduce of sharded_collect,
    // a replicated-sketch build (lrge_hip_index_build_for with a communicator) has ONE collective, the statistics all-reduce
    if (ro && ro->comm) {
        if (sharded) cg.expect(CollectiveGuard::ALLREDUCE_U64, (size_t)ro->comm->world + 1, (size_t)ro->comm->world);
        else cg.expect(CollectiveGuard::ALLREDUCE_U64, stats_vec_words(P), stats_vec_words(P) - 1);
    }
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
        rc = sharded_collect(ctx, sc, P, preset, pk, pk ? pk_pos1 : 0, pk_ybits, ro, &so, &own_hashes, &n_own, cg);
        t.stop();
        if (rc) return rc;
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
        // plain build of packed entries: the sort's first pass reads the sketch's per-chunk slots, no compaction in between
        // (k_prims.h: radix_sort_keys_first_pass_from_slots; option NO_SLOT_SORT: compact first, rounds 1-3)
        const bool keep_slots = pk && !ro && !ctx->opt("NO_SLOT_SORT");
        rc = sketch_device(ctx, sc, targets, preset, true, &so, pk ? pk_pos1 : 0, pk_ybits, nullptr, keep_slots);
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
        // the statistics of the whole target set: ONE all-reduce of [distinct, minimizers, head bins, status] on a host vector.  The
        // status word is what used to be a one-word agreement in front of it: every rank got this far, or none goes on
        std::vector<u64> hv((size_t)head + 3, 0);
        HIPCHK(ctx, hipMemcpyAsync(hv.data(), d_vec, ((size_t)head + 2) * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        if (shard_fail_at(ctx, 6)) return LRGE_ERR_DEVICE;
        cg.disarm();
        if (ro->comm) {
            rc = comm_allreduce_sum_host(ro->comm, hv.data(), hv.size(), 8, ctx->stream); if (rc) return rc;
            if (hv[(size_t)head + 2]) { LRGE_SET_ERR(ctx, "collective index build: %llu other rank(s) failed", (unsigned long long)hv[(size_t)head + 2]); return LRGE_ERR_DEVICE; }
        }
        g_distinct = hv[0]; g_mz = hv[1];
        // mm_idx_cal_max_occ + mm_mapopt_update clamps over the distinct keys of the whole target set (same arithmetic as below)
        int thres = INT32_MAX;
        if (g_distinct) {
            const u64 kth = (u64)((1. - (double)P.mid_occ_frac) * (double)g_distinct);
            u64 cum = 0; u32 v = max_bin_; bool found = false;
            for (u32 b = 0; b < head; ++b) { cum += hv[2 + b]; if (cum > kth) { v = b; found = true; break; } }
            if (!found && head < max_bin_ + 1) {      // the k-th count lies beyond the head bins: the whole histogram travels
                // (every rank takes this branch or none does: it follows from the reduced vector.  One more allocation in front of a
                // collective, so one word of agreement first)
                cg.expect(CollectiveGuard::AGREE);
                u64 *d_full = sc.get<u64>((size_t)max_bin_ + 1);
                int arc = d_full ? LRGE_OK : LRGE_ERR_DEVICE;
                if (d_full) {
                    hipLaunchKernelGGL(k_u32_to_u64, dim3((u32)div_up((u64)max_bin_ + 1, 256)), dim3(256), 0, ctx->stream, d_hist, (u64)max_bin_ + 1, d_full);
                    if (hipGetLastError() != hipSuccess) arc = LRGE_ERR_DEVICE;
                }
                cg.disarm();
                if (ro->comm) { rc = comm_agree(ro->comm, arc, ctx->stream); if (rc) return rc; }
                else if (arc) return arc;
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
        if (pk && so.slots) {
            ALLOC_OR_FAIL(k0, sc, u64, M + 1);
            rc = radix_sort_keys_first_pass_from_slots(ctx, sc, so.slots, so.offs, so.n_chunks, (u32)SK_CAP, k1, M, (int)pk_ybits, 2 * P.k, /*reverse_digits=*/true);
            if (rc) return rc;
            sc.drop(so.slots); sc.drop(so.offs);          // (recycled in stream order)
            u64 *rk = k1;
            rc = radix_sort_keys(ctx, sc, k1, k0, M, (int)pk_ybits, 2 * P.k, &rk, /*reverse_digits=*/true, 1, -1);
            if (rc) return rc;
            skey = rk; spos = rk;
            sc.drop(rk == k1 ? k0 : k1);
        } else if (pk) {
            u64 *rk = so.x;
            rc = radix_sort_keys(ctx, sc, so.x, k1, M, (int)pk_ybits, 2 * P.k, &rk, /*reverse_digits=*/true, pass_from, -1);   // see k_index.h
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
        rc = presketch_start_pending(ctx, targets->total_bases, /*may_block=*/!targets->is_view);
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
        if (const char *o = ctx->opt("HT_SLOTS_X100")) cap = (u64)n_runs * std::max<u64>(110, strtoull(o, nullptr, 10)) / 100;    // (several contexts sharing one GPU: memory binds there too)
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
    // A parent whose upload is still in flight (host-side pack + chunked transfer, host_pack.h) is NOT waited for: the view
    // remembers the chunk gate that covers its last word and its consumers wait for that one (seqset_ready).  Everything a view
    // is made of here comes from the parent's host-side arrays, which exist from the moment the upload call returned.
    lrge_hip_seqset *root = const_cast<lrge_hip_seqset *>(s->parent ? s->parent : s);
    const bool gate_ok = root->pending && root->job && !root->job->gate_ev.empty() && !ctx->opt("NO_VIEW_GATES");
    if (!gate_ok) { const int rrc = seqset_ready(ctx, s); if (rrc) return rrc; }
    lrge_hip_seqset *v = new lrge_hip_seqset();
    v->ctx = ctx; v->is_view = true; v->n = r1 - r0; v->parent = root;
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
    if (gate_ok) {
        const std::vector<u64> &gw = root->job->gate_w1;          // (word offsets are absolute in views too)
        const u64 w_end = s->h_woff[r1];
        size_t j = (size_t)(std::lower_bound(gw.begin(), gw.end(), w_end) - gw.begin());
        if (j >= gw.size()) j = gw.size() - 1;
        v->view_job = root->job; v->view_gate = (int)j; v->view_root = root;
    }
    // the chunk map: a pool block filled on the main stream from the view's own host copy (which lives as long as the view)
    hipError_t e = hipSuccess;
    v->d_cs = (u32 *)ctx->pool.alloc(((size_t)v->n + 1) * 4, &e);
    if (v->d_cs) e = hipMemcpyAsync(v->d_cs, v->h_cs.data(), ((size_t)v->n + 1) * 4, hipMemcpyHostToDevice, ctx->stream);
    if (!v->d_cs || e != hipSuccess) { LRGE_SET_ERR(ctx, "seqset view: %s", hipGetErrorString(e)); if (v->d_cs) ctx->pool.release(v->d_cs); delete v; (void)hipGetLastError(); return LRGE_ERR_DEVICE; }
    *out = v;
    return LRGE_OK;
}

// mm_idx_reader_read with batch_size = max (aligner.rs:112-122) makes ONE index whatever the size of the target file.
// Here a target set above ONE_INDEX_BASES (4e9 bases: the 32-bit entry counts and base offsets of one build) is indexed in
// parts over views of the set; the occurrence statistics are then taken over all parts together (k_part_global_occ), mid_occ
// from that global histogram, and a key that is too frequent globally is marked so in every part (k_part_drop) -- the
// parts answer every lookup exactly as the one index would.
// How many parts: every part costs the queries one more round of lookups and one more batch of anchors with its chain-latency
// floor, so as few as the limits allow -- a part's minimizers must stay below 2^32 (with a margin: ~0.25 per base with HPC,
// ~0.34 without) and its sort must fit the free HBM.  Full-size C5 (30 Gbases): 8 parts of 4e9 bases 1.47 s per step, 4 parts
// 1.33 s, 3 parts 1.24 s, 2 parts 1.20 s, the same counts every time.  Option PART_BASES pins the size; a part that turns out
// too large for either limit makes the build start over with parts of half the size.
#define ONE_INDEX_BASES 4000000000ull
static u64 auto_part_bases(lrge_hip_ctx *ctx, const lrge_hip_seqset *targets, int preset) {
    const bool hpc = preset == LRGE_PRESET_AVA_PB;
    const double density = hpc ? 0.27 : 0.36;                       // minimizers per base, rounded up
    u64 by_limit = (u64)(3.4e9 / density);                           // < 2^32 entries with ~25 % to spare
    by_limit = ctx->opt_u64("DEBUG_PART_LIMIT_BASES", by_limit);      // (tests: parts on small sets)
    const u64 floor_bases = std::min<u64>(ONE_INDEX_BASES / 4, by_limit);
    size_t mfree = 0, mtot = 0;
    if (hipMemGetInfo(&mfree, &mtot) == hipSuccess) {
        // the sort's two buffers of 16-byte pairs (8-byte packed entries where they fit), the resident entries and the table
        const double per_base = density * 48.0;
        const u64 by_mem = (u64)(((double)mfree + (double)ctx->pool.idle()) * 0.6 / per_base);
        if (by_mem < by_limit) by_limit = by_mem;
    } else (void)hipGetLastError();
    if (by_limit < floor_bases) by_limit = floor_bases;
    // parts of equal size
    const u64 np = div_up(targets->total_bases, by_limit);
    return div_up(targets->total_bases, np) + targets->max_len;
}

static int index_build_parts(lrge_hip_ctx *ctx, const lrge_hip_seqset *targets, int preset, u64 part_bases, lrge_hip_index **out);

extern "C" int lrge_hip_index_build(lrge_hip_ctx *ctx, const lrge_hip_seqset *targets, int preset, lrge_hip_index **out) {
    if (!ctx || !targets || !out) return LRGE_ERR_INVALID;
    if (preset != LRGE_PRESET_AVA_ONT && preset != LRGE_PRESET_AVA_PB) { LRGE_SET_ERR(ctx, "Preset not found: %d", preset); return LRGE_ERR_INVALID; }
    *out = nullptr;
    const bool pinned = ctx->opt("PART_BASES") != nullptr;
    const u64 one_index = ctx->opt_u64("DEBUG_ONE_INDEX_BASES", ONE_INDEX_BASES);
    if (targets->total_bases <= (pinned ? ctx->opt_u64("PART_BASES", one_index) : one_index) || targets->n < 2 || targets->is_view)
        return index_build_one(ctx, targets, preset, out);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    u64 part_bases = pinned ? ctx->opt_u64("PART_BASES", ONE_INDEX_BASES) : auto_part_bases(ctx, targets, preset);
    const int fail_first = (int)ctx->opt_u64("DEBUG_PART_FAIL_ATTEMPTS", 0);          // (tests: the start-over path)
    for (int attempt = 0;; ++attempt) {
        int rc;
        if (attempt < fail_first) { LRGE_SET_ERR(ctx, "index limited to < 2^32 minimizers (injected)"); rc = LRGE_ERR_TOO_MANY; }
        else rc = index_build_parts(ctx, targets, preset, part_bases, out);
        // a part with >= 2^32 minimizers, or one the memory could not hold: smaller parts (the failed attempt released everything)
        if ((rc != LRGE_ERR_TOO_MANY && rc != LRGE_ERR_DEVICE) || pinned || attempt >= 3 || (part_bases <= ONE_INDEX_BASES / 4 && attempt >= fail_first)) return rc;
        if (ctx->opt("VERBOSE")) fprintf(stderr, "[lrge_hip] index parts of %llu bases failed (%s): trying half\n", (unsigned long long)part_bases, ctx->err.c_str());
        (void)hipGetLastError();
        ctx->pool.trim();
        part_bases /= 2;
    }
}

static int index_build_parts(lrge_hip_ctx *ctx, const lrge_hip_seqset *targets, int preset, u64 part_bases, lrge_hip_index **out) {
    // cut by reads, every part at most part_bases bases (a single longer read gets a part of its own)
    std::vector<u32> cuts{0};
    u64 acc = 0;
    for (u32 r = 0; r < targets->n; ++r) {
        if (acc && acc + targets->h_len[r] > part_bases) { cuts.push_back(r); acc = 0; }
        acc += targets->h_len[r];
    }
    cuts.push_back(targets->n);
    const int np = (int)cuts.size() - 1;
    if (np > MAX_INDEX_PARTS) { LRGE_SET_ERR(ctx, "target set needs %d index parts (limit %d)", np, MAX_INDEX_PARTS); return LRGE_ERR_INVALID; }
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
    if (ctx->ts_build) {       // one rank's shard of a target-sharded build: the statistics are taken over ALL ranks' tables (host_tshard.inl)
        memcpy(ctx->ms, ms_acc, sizeof ms_acc); memcpy(ctx->counters, cn_acc, sizeof cn_acc);
        *out = top_guard.release();
        return LRGE_OK;
    }
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
    if (targets->total_bases > ctx->opt_u64("PART_BASES", ONE_INDEX_BASES)) {
        LRGE_SET_ERR(ctx, "index_build_for: target sets above 4e9 bases (a partitioned index) are not implemented for restricted builds");
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
    // from here on a failure is owed to the build's first collective, the (world + 1)-word sizes all-reduce of sharded_collect
    CollectiveGuard eg{comm, ctx->stream};
    eg.expect(CollectiveGuard::ALLREDUCE_U64, (size_t)comm->world + 1, (size_t)comm->world);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    lrge_hip_seqset *meta = nullptr;
    int rc = seqset_describe(ctx, all_target_lens, n_targets, all_target_ranks, &meta);
    if (rc || shard_fail_at(ctx, 7)) { if (meta) lrge_hip_seqset_free(meta); return rc ? rc : LRGE_ERR_DEVICE; }
    eg.disarm();                 // (index_build_one arms its own guard for the same collective)
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
