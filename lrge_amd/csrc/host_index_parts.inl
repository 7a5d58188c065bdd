// host_index_parts.inl -- part of lrge_hip.hip (one translation unit; included there, in this order): views of a read set, the partitioned index with global occurrence statistics, and the index entry points of the C ABI (build, build_for, build_sharded, free, stats, dump).
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
    // a job of this size that had to start over with smaller parts last time (the estimate above was too kind: e.g. ASCII reads resident
    // beside an ava-ont job) starts with the size that worked -- a failed attempt costs its allocations, a trim and the runtime's
    // hipMalloc again: 6 s per step instead of 1.3 at full-size C5 ava-ont on the resident clock (round 5)
    if (!pinned && ctx->part_hint_total == targets->total_bases && ctx->part_hint_preset == preset && ctx->part_hint_bases && ctx->part_hint_bases < part_bases)
        part_bases = ctx->part_hint_bases;
    const int fail_first = (int)ctx->opt_u64("DEBUG_PART_FAIL_ATTEMPTS", 0);          // (tests: the start-over path)
    for (int attempt = 0;; ++attempt) {
        int rc;
        if (attempt < fail_first) { LRGE_SET_ERR(ctx, "index limited to < 2^32 minimizers (injected)"); rc = LRGE_ERR_TOO_MANY; }
        else rc = index_build_parts(ctx, targets, preset, part_bases, out);
        // a part with >= 2^32 minimizers, or one the memory could not hold: smaller parts (the failed attempt released everything)
        if (rc == LRGE_OK && attempt > 0 && !pinned) { ctx->part_hint_total = targets->total_bases; ctx->part_hint_preset = preset; ctx->part_hint_bases = part_bases; }
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
            for (u32 sgm = 0; sgm + 1 < (u32)ix->h_seg_start.size(); ++sgm)
                for (u64 i = ix->h_seg_start[sgm]; i < ix->h_seg_start[sgm + 1]; ++i) hk[i] = seg_hash(hk[i], sgm, ix->seg_e, 2 * (u32)ix->P.k);
    }
    std::vector<u32> ord(ix->n_entries);
    for (u64 i = 0; i < ix->n_entries; ++i) ord[i] = (u32)i;
    // (by hash, then by y: the wave-dense sketch leaves the entries of equal hash inside one 8 192-base window in emission order, not in
    // position order -- nothing on the device depends on that order, the dump presents the reference's)
    std::sort(ord.begin(), ord.end(), [&](u32 a, u32 b) { return hk[a] != hk[b] ? hk[a] < hk[b] : hp[a] < hp[b]; });
    for (u64 i = 0; i < m; ++i) {
        if (keys) keys[i] = hk[ord[i]];
        if (pos) pos[i] = hp[ord[i]];
    }
    return LRGE_OK;
}

