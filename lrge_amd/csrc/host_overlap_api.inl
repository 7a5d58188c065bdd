// host_overlap_api.inl -- part of lrge_hip.hip (one translation unit; included there, in this order): run_overlap (one call = one OverlapRun over its batches) and the entry points: two-set, inverse, all-vs-all, chains, PAF statistics, anchor dump -- over one index, its parts, and views of the streamed set.
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
    int shrinks = 0;
    while (q0 < q_end) {
        u32 q1 = q0; u64 A = 0;
        while (q1 < q_end && (q1 - q0) < (1u << std::min<u32>(R.max_bits_q, 24)) && (q1 == q0 || A + R.h_qtot[q1] <= R.batch_cap)) { A += R.h_qtot[q1]; ++q1; }
        if (A >= (1ULL << 32)) { LRGE_SET_ERR(ctx, "query %u alone yields %llu anchors (limit 2^32)", q0, (unsigned long long)A); return done(LRGE_ERR_TOO_MANY); }
        u64 cn_before[LRGE_C_N];                   // (a batch that is retried in halves must leave no trace in the counters: ADVICE r05 -- anchors_kept
        memcpy(cn_before, ctx->counters, sizeof cn_before);   //  was added before the allocation that failed, and the bench's figures derive from it)
        ctx->counters[LRGE_C_BATCHES] += 1;
        R.kl.bits_q = std::max<u32>(1, ceil_log2_u64((u64)(q1 - q0)));
        R.cp.kl = R.kl; R.cp.q0 = q0;
        rc = R.batch(q0, q1, A);
        if (rc == LRGE_ERR_DEVICE && q1 - q0 > 1 && shrinks < 6 && ctx->err.compare(0, 17, "device allocation") == 0 && !ctx->opt("NO_BATCH_RETRY")) {
            // The batch's scratch did not fit after all (the plan budgets 48 B per anchor out of 4/5 of the free HBM; other users of
            // the device, a fragmented arena): nothing of the batch has reached the counts yet (k_count is its last launch and
            // needs no memory), so drain both streams, give idle segments back and take the same queries in smaller batches.
            (void)hipStreamSynchronize(ctx->stream); (void)hipStreamSynchronize(ctx->stream2); (void)hipGetLastError();
            ctx->pool.trim();
            memcpy(ctx->counters, cn_before, sizeof cn_before);
            R.batch_cap = std::max<u64>(A / 2, 1024);
            ++shrinks;
            if (ctx->opt("VERBOSE")) fprintf(stderr, "[lrge_hip] batch of %llu anchors did not fit (%s): batches of at most %llu from here\n", (unsigned long long)A, ctx->err.c_str(), (unsigned long long)R.batch_cap);
            ctx->err.clear();
            continue;
        }
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
    float ms[LRGE_T_N]; u64 cn[LRGE_C_N]; u64 parts = 0;
    StageAcc() { memset(ms, 0, sizeof ms); memset(cn, 0, sizeof cn); }
    void add(const lrge_hip_ctx *ctx) {
        for (int i = 0; i < LRGE_T_N; ++i) ms[i] += ctx->ms[i];
        for (int i = 0; i < LRGE_C_N; ++i) cn[i] = i == LRGE_C_LPG_SPLIT ? ctx->counters[i] : cn[i] + ctx->counters[i];
    }
    void store(lrge_hip_ctx *ctx) const { memcpy(ctx->ms, ms, sizeof ms); memcpy(ctx->counters, cn, sizeof cn); ctx->counters[LRGE_C_INDEX_PARTS] = parts; }
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
    // parts and it has a mapping if it has one in any part; every part sees the same queries and the global mid_occ.
    // The reference counts distinct target NAMES (twoset.rs:286-317) and never rejects a duplicate identifier in this mode: a name
    // that two reads of ONE part share is counted once (k_count's t_dup walk), a name shared across PARTS would be counted once per
    // part -- refused instead of counted wrongly (the target-sharded multi-GPU form has the same limit: lrge_hip_index_build_tsharded)
    if (ix->seqs && ix->seqs->dup_rank) {
        std::vector<std::pair<u32, u32>> rp;      // (name rank, part)
        for (size_t pi = 0; pi < ix->parts.size(); ++pi)
            for (u32 r : ix->parts[pi]->seqs->h_rank) rp.emplace_back(r, (u32)pi);
        std::sort(rp.begin(), rp.end());
        for (size_t i = 1; i < rp.size(); ++i)
            if (rp[i].first == rp[i - 1].first && rp[i].second != rp[i - 1].second) {
                LRGE_SET_ERR(ctx, "Duplicate read identifier across the parts of a partitioned index (target set above PART_BASES bases): distinct target names cannot be counted part by part");
                return LRGE_ERR_DUPLICATE_ID;
            }
    }
    const u32 nq = queries->n;
    acc.parts = ix->parts.size();
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
    acc.parts = ix->parts.size();
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

